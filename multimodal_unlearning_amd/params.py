"""Parameter inventory of the LLaVA-1.5 replica and conversion to / from HF state-dict names.

Our layout fuses what the kernels consume fused: CLIP q/k/v -> `qkv` ([3d, d] + bias), Llama q/k/v -> `qkv`,
gate/up -> `gate_up`; the patch-embedding conv weight is stored flattened [d, C*P*P] zero-padded to a
multiple of 8 columns (im2col GEMM).  HF names handled: transformers 5.x (`model.vision_tower...`,
`model.language_model...`, `lm_head`) and the reference's pinned 4.45.1 (`vision_tower.vision_model...`,
`language_model.model...`, `language_model.lm_head`; req.txt:250).  Only the CLIP layers that feed
`hidden_states[vision_feature_layer]` are kept (23 of 24 for LLaVA-1.5).
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch

from .config import LlavaConfig


def param_specs(cfg: LlavaConfig) -> "OrderedDict[str, tuple]":
    """name -> shape, in the order parameters are laid out in the flat buffers (vision, projector, text)."""
    v, t = cfg.vision, cfg.text
    dv, iv, dt, it = v.hidden_size, v.intermediate_size, t.hidden_size, t.intermediate_size
    sp: "OrderedDict[str, tuple]" = OrderedDict()
    sp["vision.patch_embed.weight"] = (dv, v.patch_k_padded)
    sp["vision.class_embedding"] = (dv,)
    sp["vision.position_embedding"] = (v.num_patches + 1, dv)
    sp["vision.pre_ln.weight"] = (dv,)
    sp["vision.pre_ln.bias"] = (dv,)
    for i in range(cfg.vision_layers_used):
        p = f"vision.layers.{i}."
        sp[p + "ln1.weight"] = (dv,)
        sp[p + "ln1.bias"] = (dv,)
        sp[p + "qkv.weight"] = (3 * dv, dv)
        sp[p + "qkv.bias"] = (3 * dv,)
        sp[p + "out.weight"] = (dv, dv)
        sp[p + "out.bias"] = (dv,)
        sp[p + "ln2.weight"] = (dv,)
        sp[p + "ln2.bias"] = (dv,)
        sp[p + "fc1.weight"] = (iv, dv)
        sp[p + "fc1.bias"] = (iv,)
        sp[p + "fc2.weight"] = (dv, iv)
        sp[p + "fc2.bias"] = (dv,)
    sp["projector.linear_1.weight"] = (dt, dv)
    sp["projector.linear_1.bias"] = (dt,)
    sp["projector.linear_2.weight"] = (dt, dt)
    sp["projector.linear_2.bias"] = (dt,)
    sp["text.embed_tokens.weight"] = (t.vocab_size, dt)
    for i in range(t.num_layers):
        p = f"text.layers.{i}."
        sp[p + "input_norm.weight"] = (dt,)
        sp[p + "qkv.weight"] = (3 * dt, dt)
        sp[p + "o.weight"] = (dt, dt)
        sp[p + "post_norm.weight"] = (dt,)
        sp[p + "gate_up.weight"] = (2 * it, dt)
        sp[p + "down.weight"] = (dt, it)
    sp["text.norm.weight"] = (dt,)
    sp["lm_head.weight"] = (t.vocab_size, dt)
    return sp


def is_vision(name: str) -> bool:
    return name.startswith("vision.")


def no_weight_decay(name: str) -> bool:
    """True for the tensors transformers.Trainer.create_optimizer puts in its weight_decay = 0 group
    (`get_decay_parameter_names`: biases and every LayerNorm / RMSNorm weight); Linear and embedding weights — and
    LoRA's A / B matrices, which are Linear weights in peft — are decayed."""
    return name.endswith(".bias") or name.endswith(("norm.weight", "ln1.weight", "ln2.weight", "pre_ln.weight"))


def init_params(cfg: LlavaConfig, seed: int = 0, std: float = 0.02, dtype=torch.float32) -> "OrderedDict[str, torch.Tensor]":
    """Framework-independent deterministic init (numpy PCG64 stream, in param_specs order):
    matrices / embeddings ~ N(0, std^2); norm weights 1 (+ small jitter so tests see them); biases ~ N(0, std^2).
    Every value is rounded to bf16 so fp32 oracles and the bf16 kernels hold identical weights."""
    rng = np.random.default_rng(seed)
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, shape in param_specs(cfg).items():
        if name.endswith("norm.weight") or name.endswith("ln1.weight") or name.endswith("ln2.weight") or \
                name.endswith("pre_ln.weight"):
            a = 1.0 + 0.05 * rng.standard_normal(shape)
        else:
            a = std * rng.standard_normal(shape)
        tns = torch.from_numpy(a.astype(np.float32))
        if name == "vision.patch_embed.weight":
            tns[:, cfg.vision.patch_k:] = 0
        out[name] = tns.to(torch.bfloat16).to(dtype)
    return out


# ------------------------------------------------------------------------------------------- HF names
def _hf_prefixes(style: str):
    if style == "hf5":
        return "model.vision_tower.vision_model.", "model.multi_modal_projector.", "model.language_model.", "lm_head."
    if style == "hf5_flat":  # transformers 5.15: CLIPVisionModel fields directly under vision_tower
        return "model.vision_tower.", "model.multi_modal_projector.", "model.language_model.", "lm_head."
    if style == "hf4":  # 4.45.1 (the reference's pin)
        return ("vision_tower.vision_model.", "multi_modal_projector.", "language_model.model.",
                "language_model.lm_head.")
    raise ValueError(style)


def to_hf_state_dict(cfg: LlavaConfig, params, style: str = "hf5_flat") -> "OrderedDict[str, torch.Tensor]":
    """Our fused layout -> HF parameter names (q/k/v and gate/up split; conv weight un-flattened)."""
    vt, pj, lm, head = _hf_prefixes(style)
    v, t = cfg.vision, cfg.text
    dv, dt, it = v.hidden_size, t.hidden_size, t.intermediate_size
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    sd[vt + "embeddings.class_embedding"] = params["vision.class_embedding"]
    sd[vt + "embeddings.patch_embedding.weight"] = params["vision.patch_embed.weight"][:, : v.patch_k].reshape(
        dv, v.num_channels, v.patch_size, v.patch_size)
    sd[vt + "embeddings.position_embedding.weight"] = params["vision.position_embedding"]
    sd[vt + "pre_layrnorm.weight"] = params["vision.pre_ln.weight"]
    sd[vt + "pre_layrnorm.bias"] = params["vision.pre_ln.bias"]
    for i in range(cfg.vision_layers_used):
        p, h = f"vision.layers.{i}.", vt + f"encoder.layers.{i}."
        w, b = params[p + "qkv.weight"], params[p + "qkv.bias"]
        for j, nm in enumerate(("q_proj", "k_proj", "v_proj")):
            sd[h + f"self_attn.{nm}.weight"] = w[j * dv:(j + 1) * dv]
            sd[h + f"self_attn.{nm}.bias"] = b[j * dv:(j + 1) * dv]
        sd[h + "self_attn.out_proj.weight"] = params[p + "out.weight"]
        sd[h + "self_attn.out_proj.bias"] = params[p + "out.bias"]
        sd[h + "layer_norm1.weight"] = params[p + "ln1.weight"]
        sd[h + "layer_norm1.bias"] = params[p + "ln1.bias"]
        sd[h + "layer_norm2.weight"] = params[p + "ln2.weight"]
        sd[h + "layer_norm2.bias"] = params[p + "ln2.bias"]
        sd[h + "mlp.fc1.weight"] = params[p + "fc1.weight"]
        sd[h + "mlp.fc1.bias"] = params[p + "fc1.bias"]
        sd[h + "mlp.fc2.weight"] = params[p + "fc2.weight"]
        sd[h + "mlp.fc2.bias"] = params[p + "fc2.bias"]
    for k in ("linear_1", "linear_2"):
        sd[pj + k + ".weight"] = params[f"projector.{k}.weight"]
        sd[pj + k + ".bias"] = params[f"projector.{k}.bias"]
    sd[lm + "embed_tokens.weight"] = params["text.embed_tokens.weight"]
    for i in range(t.num_layers):
        p, h = f"text.layers.{i}.", lm + f"layers.{i}."
        w = params[p + "qkv.weight"]
        for j, nm in enumerate(("q_proj", "k_proj", "v_proj")):
            sd[h + f"self_attn.{nm}.weight"] = w[j * dt:(j + 1) * dt]
        sd[h + "self_attn.o_proj.weight"] = params[p + "o.weight"]
        gu = params[p + "gate_up.weight"]
        sd[h + "mlp.gate_proj.weight"] = gu[:it]
        sd[h + "mlp.up_proj.weight"] = gu[it:]
        sd[h + "mlp.down_proj.weight"] = params[p + "down.weight"]
        sd[h + "input_layernorm.weight"] = params[p + "input_norm.weight"]
        sd[h + "post_attention_layernorm.weight"] = params[p + "post_norm.weight"]
    sd[lm + "norm.weight"] = params["text.norm.weight"]
    sd[head + "weight"] = params["lm_head.weight"]
    return sd


def detect_hf_style(keys) -> str:
    keys = list(keys)
    if any(k.startswith("language_model.model.") for k in keys):
        return "hf4"
    if any(k.startswith("model.vision_tower.vision_model.") for k in keys):
        return "hf5"
    return "hf5_flat"


def from_hf_state_dict(cfg: LlavaConfig, sd, style: str | None = None) -> "OrderedDict[str, torch.Tensor]":
    """HF names -> our fused layout (inverse of to_hf_state_dict; extra HF keys such as the unused last CLIP
    layer and post_layernorm are ignored)."""
    style = style or detect_hf_style(sd.keys())
    vt, pj, lm, head = _hf_prefixes(style)
    v, t = cfg.vision, cfg.text
    dv = v.hidden_size
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    w = sd[vt + "embeddings.patch_embedding.weight"].reshape(dv, -1)
    pw = torch.zeros(dv, v.patch_k_padded, dtype=w.dtype)
    pw[:, : v.patch_k] = w
    out["vision.patch_embed.weight"] = pw
    out["vision.class_embedding"] = sd[vt + "embeddings.class_embedding"]
    out["vision.position_embedding"] = sd[vt + "embeddings.position_embedding.weight"]
    out["vision.pre_ln.weight"] = sd[vt + "pre_layrnorm.weight"]
    out["vision.pre_ln.bias"] = sd[vt + "pre_layrnorm.bias"]
    for i in range(cfg.vision_layers_used):
        p, h = f"vision.layers.{i}.", vt + f"encoder.layers.{i}."
        out[p + "ln1.weight"] = sd[h + "layer_norm1.weight"]
        out[p + "ln1.bias"] = sd[h + "layer_norm1.bias"]
        out[p + "qkv.weight"] = torch.cat([sd[h + f"self_attn.{n}.weight"] for n in ("q_proj", "k_proj", "v_proj")])
        out[p + "qkv.bias"] = torch.cat([sd[h + f"self_attn.{n}.bias"] for n in ("q_proj", "k_proj", "v_proj")])
        out[p + "out.weight"] = sd[h + "self_attn.out_proj.weight"]
        out[p + "out.bias"] = sd[h + "self_attn.out_proj.bias"]
        out[p + "ln2.weight"] = sd[h + "layer_norm2.weight"]
        out[p + "ln2.bias"] = sd[h + "layer_norm2.bias"]
        out[p + "fc1.weight"] = sd[h + "mlp.fc1.weight"]
        out[p + "fc1.bias"] = sd[h + "mlp.fc1.bias"]
        out[p + "fc2.weight"] = sd[h + "mlp.fc2.weight"]
        out[p + "fc2.bias"] = sd[h + "mlp.fc2.bias"]
    for k in ("linear_1", "linear_2"):
        out[f"projector.{k}.weight"] = sd[pj + k + ".weight"]
        out[f"projector.{k}.bias"] = sd[pj + k + ".bias"]
    out["text.embed_tokens.weight"] = sd[lm + "embed_tokens.weight"]
    for i in range(t.num_layers):
        p, h = f"text.layers.{i}.", lm + f"layers.{i}."
        out[p + "input_norm.weight"] = sd[h + "input_layernorm.weight"]
        out[p + "qkv.weight"] = torch.cat([sd[h + f"self_attn.{n}.weight"] for n in ("q_proj", "k_proj", "v_proj")])
        out[p + "o.weight"] = sd[h + "self_attn.o_proj.weight"]
        out[p + "post_norm.weight"] = sd[h + "post_attention_layernorm.weight"]
        out[p + "gate_up.weight"] = torch.cat([sd[h + "mlp.gate_proj.weight"], sd[h + "mlp.up_proj.weight"]])
        out[p + "down.weight"] = sd[h + "mlp.down_proj.weight"]
    out["text.norm.weight"] = sd[lm + "norm.weight"]
    out["lm_head.weight"] = sd[head + "weight"]
    specs = param_specs(cfg)
    for k, shp in specs.items():
        assert tuple(out[k].shape) == tuple(shp), (k, tuple(out[k].shape), shp)
    return OrderedDict((k, out[k]) for k in specs)
