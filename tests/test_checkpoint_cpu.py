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
