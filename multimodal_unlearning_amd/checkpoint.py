"""HF-layout checkpoint I/O (safetensors shards + model.safetensors.index.json) for the LLaVA replica.

Counterpart of `model.save_pretrained(...)` / `from_pretrained(...)` in mm/forget.py:83-89,193-195 (shard names
`model-0000i-of-0000n.safetensors`, :73-75).  Tensor names are transformers' (4.45.1 or 5.x style, see params.py);
only safetensors files are read (nothing is unpickled).  LoRA adapters are saved peft-shaped next to the base."""
from __future__ import annotations

import json
import os
from collections import OrderedDict

import torch
from safetensors.torch import load_file, save_file

from .config import LlavaConfig
from .params import detect_hf_style, from_hf_state_dict, to_hf_state_dict


def hf_config_dict(cfg: LlavaConfig, dtype=torch.bfloat16) -> dict:
    """The `config.json` transformers needs to rebuild the model (`AutoModelForPreTraining.from_pretrained(save_dir)` in
    mm/eval.py reads it): a LlavaConfig with its CLIP vision_config and Llama text_config."""
    v, t = cfg.vision, cfg.text
    return {
        "architectures": ["LlavaForConditionalGeneration"], "model_type": "llava",
        "image_token_index": cfg.image_token_index, "image_seq_length": v.num_patches,
        "vision_feature_layer": cfg.vision_feature_layer, "vision_feature_select_strategy": "default",
        "projector_hidden_act": "gelu", "multimodal_projector_bias": True, "tie_word_embeddings": False,
        "torch_dtype": str(dtype).replace("torch.", ""),
        "vision_config": {"model_type": "clip_vision_model", "hidden_size": v.hidden_size,
                          "intermediate_size": v.intermediate_size, "num_hidden_layers": v.num_layers,
                          "num_attention_heads": v.num_heads, "image_size": v.image_size, "patch_size": v.patch_size,
                          "num_channels": v.num_channels, "layer_norm_eps": v.layer_norm_eps, "hidden_act": "quick_gelu",
                          "projection_dim": 768, "vocab_size": 32000},
        "text_config": {"model_type": "llama", "architectures": ["LlamaForCausalLM"], "hidden_size": t.hidden_size,
                        "intermediate_size": t.intermediate_size, "num_hidden_layers": t.num_layers,
                        "num_attention_heads": t.num_heads, "num_key_value_heads": t.num_heads,
                        "vocab_size": t.vocab_size, "rms_norm_eps": t.rms_norm_eps, "rope_theta": t.rope_theta,
                        "max_position_embeddings": t.max_position, "hidden_act": "silu", "attention_bias": False,
                        "mlp_bias": False, "tie_word_embeddings": False,
                        "torch_dtype": str(dtype).replace("torch.", "")},
    }


def unused_hf_tensor_shapes(cfg: LlavaConfig, style: str):
    """HF parameters the forget step never reads (so our fused layout does not hold them): the CLIP layers after the
    feature layer (one of 24 for LLaVA-1.5) and `post_layernorm`.  name -> shape, in HF order."""
    from .params import _hf_prefixes

    vt = _hf_prefixes(style)[0]
    v = cfg.vision
    dv, iv = v.hidden_size, v.intermediate_size
    out = OrderedDict()
    for i in range(cfg.vision_layers_used, v.num_layers):
        h = vt + f"encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            out[h + f"self_attn.{n}.weight"], out[h + f"self_attn.{n}.bias"] = (dv, dv), (dv,)
        for n in ("layer_norm1", "layer_norm2"):
            out[h + n + ".weight"], out[h + n + ".bias"] = (dv,), (dv,)
        out[h + "mlp.fc1.weight"], out[h + "mlp.fc1.bias"] = (iv, dv), (iv,)
        out[h + "mlp.fc2.weight"], out[h + "mlp.fc2.bias"] = (dv, iv), (dv,)
    out[vt + "post_layernorm.weight"], out[vt + "post_layernorm.bias"] = (dv,), (dv,)
    return out


def save_pretrained(cfg: LlavaConfig, params, out_dir: str, style: str = "hf4", max_shard_bytes: int = 5 << 30,
                    dtype=torch.bfloat16, lora_state=None, extra=None, lora_r: int = 8, lora_alpha: float = 32.0,
                    lora_dropout: float = 0.0):
    """`model.save_pretrained(save_dir)` (mm/forget.py:193-195) for the replica: safetensors shards with transformers'
    tensor names, `config.json` (LlavaConfig) and `generation_config.json`, so the reference's eval
    (`from_pretrained(save_dir)`, mm/eval.py) loads the directory with no missing key.  `extra`: the HF tensors
    `load_pretrained(..., return_extra=True)` set aside (unused last CLIP layer, post_layernorm) — written back
    unchanged; without it (a model that never came from an HF checkpoint) those entries are written as identity
    LayerNorms / zero Linears and listed under "synthesized_keys" in mmu_config.json.  `lora_state` (compact fused
    tensors of LlavaHIP.lora_state()) goes to peft's layout: adapter_model.safetensors + adapter_config.json."""
    os.makedirs(out_dir, exist_ok=True)
    sd = to_hf_state_dict(cfg, params, style)
    synthesized = []
    for k, shape in unused_hf_tensor_shapes(cfg, style).items():
        if extra is not None and k in extra:
            sd[k] = extra[k]
        else:
            sd[k] = torch.ones(shape) if k.endswith(("layer_norm1.weight", "layer_norm2.weight", "post_layernorm.weight")) \
                else torch.zeros(shape)
            synthesized.append(k)
    shards, cur, cur_bytes = [], OrderedDict(), 0
    for k, v in sd.items():
        t = v.detach().to(dtype).contiguous().cpu()
        nb = t.numel() * t.element_size()
        if cur and cur_bytes + nb > max_shard_bytes:
            shards.append(cur)
            cur, cur_bytes = OrderedDict(), 0
        cur[k] = t
        cur_bytes += nb
    if cur:
        shards.append(cur)
    weight_map = {}
    n = len(shards)
    for i, sh in enumerate(shards, 1):
        fn = "model.safetensors" if n == 1 else f"model-{i:05d}-of-{n:05d}.safetensors"
        save_file(sh, os.path.join(out_dir, fn), metadata={"format": "pt"})
        for k in sh:
            weight_map[k] = fn
    if n > 1:
        json.dump({"metadata": {"total_size": sum(t.numel() * t.element_size() for sh in shards for t in sh.values())},
                   "weight_map": weight_map}, open(os.path.join(out_dir, "model.safetensors.index.json"), "w"), indent=2)
    json.dump(hf_config_dict(cfg, dtype), open(os.path.join(out_dir, "config.json"), "w"), indent=2)
    json.dump({"_from_model_config": True, "bos_token_id": 1, "eos_token_id": 2, "pad_token_id": 32001},
              open(os.path.join(out_dir, "generation_config.json"), "w"), indent=2)
    mc = cfg.to_dict()
    mc["hf_tensor_names"], mc["synthesized_keys"] = style, synthesized
    json.dump(mc, open(os.path.join(out_dir, "mmu_config.json"), "w"), indent=2)
    if lora_state:
        save_adapter(cfg, lora_state, out_dir, r=lora_r, alpha=lora_alpha, style=style, dropout=lora_dropout)
    return sorted(set(weight_map.values()))


def load_pretrained(cfg: LlavaConfig, path: str, return_extra: bool = False):
    """Read every *.safetensors shard under `path` (index file optional) and return params in our fused layout; with
    return_extra also the HF tensors the fused layout does not hold (pass them back to save_pretrained)."""
    idx = os.path.join(path, "model.safetensors.index.json")
    if os.path.exists(idx):
        files = sorted(set(json.load(open(idx))["weight_map"].values()))
    else:
        files = sorted(f for f in os.listdir(path) if f.endswith(".safetensors") and not f.startswith("adapter_"))
    sd = {}
    for f in files:
        sd.update(load_file(os.path.join(path, f)))
    style = detect_hf_style(sd.keys())
    params = from_hf_state_dict(cfg, sd, style)
    if not return_extra:
        return params
    extra = OrderedDict((k, sd[k]) for k in unused_hf_tensor_shapes(cfg, style) if k in sd)
    return params, extra


# ------------------------------------------------------------------------------------------------ LoRA adapters
_FUSED = {"qkv": ("self_attn.", ("q_proj", "k_proj", "v_proj")), "o": ("self_attn.", ("o_proj",)),
          "gate_up": ("mlp.", ("gate_proj", "up_proj")), "down": ("mlp.", ("down_proj",))}


def _peft_modules(cfg: LlavaConfig, base: str, style: str):
    """Our (possibly fused) module name -> the HF module paths peft would have wrapped, in fused order."""
    from .params import _hf_prefixes

    vt, pj, lm, head = _hf_prefixes(style)
    if base.startswith("vision.layers."):
        _, _, i, key = base.split(".")
        sub, names = {"qkv": ("self_attn.", ("q_proj", "k_proj", "v_proj")), "out": ("self_attn.", ("out_proj",)),
                      "fc1": ("mlp.", ("fc1",)), "fc2": ("mlp.", ("fc2",))}[key]
        return [f"{vt}encoder.layers.{i}.{sub}{n}" for n in names]
    if base.startswith("projector."):
        return [pj + base[len("projector."):]]
    if base == "lm_head":
        return [head[:-1]]
    _, _, i, key = base.split(".")  # text.layers.<i>.<key>
    sub, names = _FUSED[key]
    return [f"{lm}layers.{i}.{sub}{n}" for n in names]


def lora_state_to_peft(cfg: LlavaConfig, state, style: str = "hf4"):
    """Compact fused adapter tensors (LlavaHIP.lora_state) -> peft's `adapter_model.safetensors` names:
    base_model.model.<module>.lora_A.weight [r,in] / lora_B.weight [out,r], one pair per un-fused HF Linear."""
    out = OrderedDict()
    for k in state:
        if not k.endswith(".lora_A"):
            continue
        base = k[: -len(".lora_A")]
        A, Bm = state[k], state[base + ".lora_B"]
        mods = _peft_modules(cfg, base, style)
        r, seg = A.shape[0] // len(mods), Bm.shape[0] // len(mods)
        for j, m in enumerate(mods):
            out[f"base_model.model.{m}.lora_A.weight"] = A[j * r:(j + 1) * r].contiguous()
            out[f"base_model.model.{m}.lora_B.weight"] = Bm[j * seg:(j + 1) * seg].contiguous()
    return out


def lora_state_from_peft(cfg: LlavaConfig, sd, bases, style: str | None = None):
    """peft names -> compact fused tensors for the module list `bases` (e.g. [n[:-7] for n in model.lora])."""
    if style is None:
        style = "hf4" if any(".language_model.model." in k for k in sd) else "hf5"
    out = OrderedDict()
    for base in bases:
        mods = _peft_modules(cfg, base, style)
        try:
            out[base + ".lora_A"] = torch.cat([sd[f"base_model.model.{m}.lora_A.weight"] for m in mods])
            out[base + ".lora_B"] = torch.cat([sd[f"base_model.model.{m}.lora_B.weight"] for m in mods])
        except KeyError as e:
            raise KeyError(f"adapter file has no tensors for {base} ({e})") from None
    return out


def save_adapter(cfg: LlavaConfig, state, out_dir: str, r: int, alpha: float, style: str = "hf4", dropout: float = 0.0):
    """peft-layout adapter directory: adapter_model.safetensors + the fields of adapter_config.json we use."""
    os.makedirs(out_dir, exist_ok=True)
    save_file({k: v.contiguous() for k, v in lora_state_to_peft(cfg, state, style).items()},
              os.path.join(out_dir, "adapter_model.safetensors"))
    json.dump({"peft_type": "LORA", "r": r, "lora_alpha": alpha, "lora_dropout": float(dropout), "bias": "none",
               "task_type": "CAUSAL_LM"}, open(os.path.join(out_dir, "adapter_config.json"), "w"), indent=2)


def load_adapter(cfg: LlavaConfig, path: str, bases):
    """-> (compact fused state, r, alpha) from a peft-layout adapter directory (safetensors only)."""
    conf = json.load(open(os.path.join(path, "adapter_config.json")))
    sd = load_file(os.path.join(path, "adapter_model.safetensors"))
    return lora_state_from_peft(cfg, sd, bases), int(conf["r"]), float(conf["lora_alpha"])


def merge_adapter_and_save(model, out_dir: str, style: str = "hf4"):
    """mm/forget.py:100-113: fold the loaded adapter into the base weights and save a plain HF-layout checkpoint
    "so that we can again do PEFT style finetuning from scratch"."""
    model.merge_lora_()
    return save_pretrained(model.cfg, model.w, out_dir, style=style)
