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
