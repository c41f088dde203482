import os

import torch

from multimodal_unlearning_amd.checkpoint import load_pretrained, save_pretrained
from multimodal_unlearning_amd.config import tiny_test_config
from multimodal_unlearning_amd.params import init_params


def test_safetensors_round_trip_sharded(tmp_path):
    cfg = tiny_test_config()
    P = init_params(cfg, seed=2)
    for style in ("hf4", "hf5_flat"):
        d = str(tmp_path / style)
        files = save_pretrained(cfg, P, d, style=style, max_shard_bytes=1 << 20)
        assert len(files) > 1 and all(f.endswith(".safetensors") for f in files)
        assert os.path.exists(os.path.join(d, "model.safetensors.index.json"))
        back = load_pretrained(cfg, d)
        for k in P:
            assert torch.equal(back[k].float(), P[k]), (style, k)  # init_params values are bf16-representable


def test_peft_adapter_names_round_trip(tmp_path):
    """Compact fused adapter tensors <-> peft's adapter_model.safetensors names (one pair per un-fused HF Linear)."""
    from collections import OrderedDict

    from multimodal_unlearning_amd.checkpoint import load_adapter, lora_state_to_peft, save_adapter

    cfg = tiny_test_config()
    d, it, r = cfg.text.hidden_size, cfg.text.intermediate_size, 8
    g = torch.Generator().manual_seed(0)
    shapes = {"text.layers.0.qkv": (3, 3 * d, d), "text.layers.0.o": (1, d, d), "text.layers.1.gate_up": (2, 2 * it, d),
              "text.layers.1.down": (1, d, it), "projector.linear_1": (1, d, cfg.vision.hidden_size),
              "lm_head": (1, cfg.text.vocab_size, d)}
    st = OrderedDict()
    for base, (k, n_out, n_in) in shapes.items():
        st[base + ".lora_A"] = torch.randn(k * r, n_in, generator=g)
        st[base + ".lora_B"] = torch.randn(n_out, r, generator=g)
    peft = lora_state_to_peft(cfg, st, "hf4")
    assert "base_model.model.language_model.model.layers.0.self_attn.k_proj.lora_A.weight" in peft
    assert "base_model.model.language_model.model.layers.1.mlp.up_proj.lora_B.weight" in peft
    assert "base_model.model.multi_modal_projector.linear_1.lora_A.weight" in peft
    assert "base_model.model.language_model.lm_head.lora_B.weight" in peft
    assert torch.equal(peft["base_model.model.language_model.model.layers.0.self_attn.v_proj.lora_A.weight"],
                       st["text.layers.0.qkv.lora_A"][2 * r:])
    assert torch.equal(peft["base_model.model.language_model.model.layers.0.self_attn.v_proj.lora_B.weight"],
                       st["text.layers.0.qkv.lora_B"][2 * d:])
    save_adapter(cfg, st, str(tmp_path / "ad"), r=r, alpha=32.0)
    back, r2, alpha = load_adapter(cfg, str(tmp_path / "ad"), list(shapes))
    assert (r2, alpha) == (r, 32.0)
    for k in st:
        assert torch.equal(back[k], st[k]), k


def test_saved_directory_loads_in_transformers_with_no_missing_key(tmp_path):
    """What the reference's eval does with the output of the forget run: `from_pretrained(save_dir)` (mm/eval.py) on the
    directory `model.save_pretrained(save_dir)` wrote (mm/forget.py:193-195).  The saved key set and shapes must be
    exactly the HF model's state_dict, config.json must rebuild the architecture, the unused last CLIP layer and
    post_layernorm of a loaded checkpoint must survive a load -> save round trip, and HF's logits on the reloaded
    directory must be the golden ones."""
    import numpy as np
    import transformers
    from safetensors.torch import load_file

    from multimodal_unlearning_amd.checkpoint import hf_config_dict, unused_hf_tensor_shapes

    cfg = tiny_test_config()
    P = init_params(cfg, seed=0)
    d1 = str(tmp_path / "first")
    save_pretrained(cfg, P, d1, style="hf5_flat", dtype=torch.float32)
    hf = transformers.LlavaForConditionalGeneration.from_pretrained(d1, torch_dtype=torch.float32)
    hf_sd = hf.state_dict()
    saved = load_file(os.path.join(d1, "model.safetensors"))
    assert set(saved) == set(hf_sd), (set(saved) ^ set(hf_sd))
    assert all(tuple(saved[k].shape) == tuple(hf_sd[k].shape) for k in saved)
    assert hf.config.image_token_index == cfg.image_token_index and hf.config.vision_feature_layer == -2
    assert hf.config.text_config.hidden_size == cfg.text.hidden_size
    # HF on the reloaded directory reproduces the golden logits the oracle is pinned to
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "llava_tiny_hf.npz"))
    sp = {k: torch.from_numpy(g[f"forget.in.{k}"]) for k in ("input_ids", "attention_mask", "pixel_values", "labels")}
    hf.config._attn_implementation = "eager"
    with torch.no_grad():
        out = hf(**sp)
    assert torch.allclose(out.logits, torch.from_numpy(g["forget.logits"]), atol=2e-5)
    # a real checkpoint's unused tensors survive load -> save
    extra_names = list(unused_hf_tensor_shapes(cfg, "hf5_flat"))
    marked = {k: torch.full_like(saved[k], 0.25 + i * 1e-3) for i, k in enumerate(extra_names)}
    from safetensors.torch import save_file
    save_file({**saved, **marked}, os.path.join(d1, "model.safetensors"), metadata={"format": "pt"})
    P2, extra = load_pretrained(cfg, d1, return_extra=True)
    assert set(extra) == set(extra_names)
    d2 = str(tmp_path / "second")
    save_pretrained(cfg, P2, d2, style="hf5_flat", dtype=torch.float32, extra=extra)
    again = load_file(os.path.join(d2, "model.safetensors"))
    for k in extra_names:
        assert torch.equal(again[k], marked[k]), k
    import json
    assert json.load(open(os.path.join(d2, "mmu_config.json")))["synthesized_keys"] == []
    assert json.load(open(os.path.join(d1, "mmu_config.json")))["synthesized_keys"] == extra_names
    assert hf_config_dict(cfg)["text_config"]["num_hidden_layers"] == cfg.text.num_layers


def test_lora_state_is_saved_in_peft_layout(tmp_path):
    from safetensors.torch import load_file

    cfg = tiny_test_config()
    d, r = cfg.text.hidden_size, 8
    st = {"text.layers.0.qkv.lora_A": torch.randn(3 * r, d), "text.layers.0.qkv.lora_B": torch.randn(3 * d, r)}
    out = str(tmp_path / "with_adapter")
    save_pretrained(cfg, init_params(cfg, seed=1), out, style="hf4", lora_state=st, lora_r=r, lora_alpha=32.0)
    ad = load_file(os.path.join(out, "adapter_model.safetensors"))
    assert "base_model.model.language_model.model.layers.0.self_attn.q_proj.lora_A.weight" in ad
    assert os.path.exists(os.path.join(out, "adapter_config.json")) and os.path.exists(os.path.join(out, "config.json"))
