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


def save_pretrained(cfg: LlavaConfig, params, out_dir: str, style: str = "hf4", max_shard_bytes: int = 5 << 30,
                    dtype=torch.bfloat16, lora_state=None):
    os.makedirs(out_dir, exist_ok=True)
    sd = to_hf_state_dict(cfg, params, style)
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
    json.dump(cfg.to_dict(), open(os.path.join(out_dir, "mmu_config.json"), "w"), indent=2)
    if lora_state:
        save_file({k: v.contiguous() for k, v in lora_state.items()}, os.path.join(out_dir, "adapter_model.safetensors"))
    return sorted(set(weight_map.values()))


def load_pretrained(cfg: LlavaConfig, path: str):
    """Read every *.safetensors shard under `path` (index file optional) and return params in our fused layout."""
    idx = os.path.join(path, "model.safetensors.index.json")
    if os.path.exists(idx):
        files = sorted(set(json.load(open(idx))["weight_map"].values()))
    else:
        files = sorted(f for f in os.listdir(path) if f.endswith(".safetensors") and not f.startswith("adapter_"))
    sd = {}
    for f in files:
        sd.update(load_file(os.path.join(path, f)))
    return from_hf_state_dict(cfg, sd, detect_hf_style(sd.keys()))


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


def save_adapter(cfg: LlavaConfig, state, out_dir: str, r: int, alpha: float, style: str = "hf4"):
    """peft-layout adapter directory: adapter_model.safetensors + the fields of adapter_config.json we use."""
    os.makedirs(out_dir, exist_ok=True)
    save_file({k: v.contiguous() for k, v in lora_state_to_peft(cfg, state, style).items()},
              os.path.join(out_dir, "adapter_model.safetensors"))
    json.dump({"peft_type": "LORA", "r": r, "lora_alpha": alpha, "lora_dropout": 0.0, "bias": "none",
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
