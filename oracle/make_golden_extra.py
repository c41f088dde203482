"""ORACLE fixture generator #2 — runs ONLY in the build container (needs /root/reference and transformers).

tests/golden/compute_loss_extra.npz:
  mm.*   the reference's MMTrainerForgetting.compute_loss (mm/trainer.py) for the branches that need a deeper
         model or extra inputs: rmu (hooks on decoder layer 7 -> 8-layer tiny LLaVA, fixed control vector) and
         sku (K=2 random splits, weights of mm/forget.py:186)
  nlp.*  the text trainer CustomTrainerForgetting.compute_loss (nlp/dataloader.py:144-485, BASELINE config 1)
         on a tiny LlamaForCausalLM with tuple inputs as custom_data_collator_forget builds them

Usage:  PYTHONPATH=/root/reference:/root/repo python oracle/make_golden_extra.py
Third-party modules the reference imports but that are absent here (deepspeed, wandb, natsort) are replaced by
empty stubs for the import only; none of their code is on the compute_loss path.
"""
from __future__ import annotations

import importlib.machinery
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")

from multimodal_unlearning_amd.config import tiny_test_config  # noqa: E402
from multimodal_unlearning_amd.params import from_hf_state_dict, init_params, to_hf_state_dict  # noqa: E402
from multimodal_unlearning_amd.synthetic import make_split, make_text_split, teacher_params_from  # noqa: E402
from oracle.make_golden import GRAD_KEYS, Wrap, build_hf, hf_grads_in_our_layout, import_reference_trainer  # noqa: E402


def stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def record(res, tag, loss, cfg, grads):
    res[f"{tag}.loss"] = loss.detach().numpy()
    tot = 0.0
    for k in GRAD_KEYS:
        if k in grads:
            res[f"{tag}.gradnorm.{k}"] = grads[k].norm().numpy()
    for k, v in grads.items():
        if not k.startswith("vision."):
            tot += float(v.double().pow(2).sum())
    res[f"{tag}.gradnorm.total"] = np.float64(tot ** 0.5)
    print(f"{tag:44s} loss={float(loss.detach()):+.6f} |g|={tot ** 0.5:.6f}")


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    res = {}
    cfg = tiny_test_config(text_layers=8)
    params = init_params(cfg, seed=0)
    tparams = teacher_params_from(params, seed=7)

    # ------------------------------------------------------------------ mm: rmu + sku
    MMTrainerForgetting = import_reference_trainer()
    student = build_hf(cfg, params).float()
    teacher = build_hf(cfg, tparams).float().eval()
    for p in teacher.parameters():
        p.requires_grad = False
    B, T = 2, 24
    splits = {n: make_split(cfg, B, T, seed=300 + i, n_question=6) for i, n in enumerate(["forget", "retain", "idk"])}
    for k, sp in splits.items():
        for kk, v in sp.items():
            res[f"mm.in.{k}.{kk}"] = v.numpy()

    class WrapRMU(Wrap):  # aliases for the 4.45.1 module tree the reference indexes (mm/trainer.py:304-309)
        def __init__(s, m):
            super().__init__(m)
            s.language_model = types.SimpleNamespace(model=types.SimpleNamespace(layers=m.model.language_model.layers))
            s.config = types.SimpleNamespace(hidden_size=cfg.text.hidden_size, image_token_index=cfg.image_token_index)

    ws, wt = WrapRMU(student), WrapRMU(teacher)
    cv = torch.rand(B, 1, cfg.text.hidden_size, generator=torch.Generator().manual_seed(5))
    cv = cv / torch.norm(cv) * 300
    res["mm.rmu.control_vec"] = cv.numpy()
    fake = types.SimpleNamespace(loss_type="rmu", teacher_model=wt, loss_beta=0.7, l1_lambda=0.0, l0_lambda=0.0,
                                 l_norm_from="zero", loss_args={}, is_deepspeed_enabled=False,
                                 to_device=MMTrainerForgetting.to_device, control_vec=cv)
    student.zero_grad()
    loss = MMTrainerForgetting.compute_loss(fake, ws, {k: dict(v) for k, v in splits.items()})
    loss.backward()
    record(res, "mm.rmu", loss, cfg, hf_grads_in_our_layout(cfg, student))

    K = 2
    rnd = [make_split(cfg, B, T, seed=400 + i, n_question=6) for i in range(K)]
    random_inputs = {k: torch.stack([r[k] for r in rnd], dim=1) for k in rnd[0]}  # [B, K, ...]
    for kk, v in random_inputs.items():
        res[f"mm.in.random.{kk}"] = v.numpy()
    la = {"bad_weight": 2.5, "random_weight": 1.0, "normal_weight": 0.5}  # mm/forget.py:186
    fake = types.SimpleNamespace(loss_type="sku", teacher_model=teacher, loss_beta=0.7, l1_lambda=0.0, l0_lambda=0.0,
                                 l_norm_from="zero", loss_args=la, is_deepspeed_enabled=False,
                                 to_device=MMTrainerForgetting.to_device)
    student.zero_grad()
    ins = {k: dict(v) for k, v in splits.items()}
    ins["random"] = {k: v.clone() for k, v in random_inputs.items()}
    loss = MMTrainerForgetting.compute_loss(fake, Wrap(student), ins)
    loss.backward()
    record(res, "mm.sku", loss, cfg, hf_grads_in_our_layout(cfg, student))

    # ------------------------------------------------------------------ nlp: text trainer on a tiny Llama
    from transformers import LlamaConfig, LlamaForCausalLM

    for n in ("deepspeed", "wandb"):
        sys.modules[n] = types.ModuleType(n)
    stub("natsort", natsorted=sorted)
    sys.path.insert(0, "/root/reference/nlp")
    import dataloader as nlp_dl  # /root/reference/nlp/dataloader.py

    t = cfg.text

    def build_llama(P):
        m = LlamaForCausalLM(LlamaConfig(hidden_size=t.hidden_size, intermediate_size=t.intermediate_size,
                                         num_hidden_layers=t.num_layers, num_attention_heads=t.num_heads,
                                         num_key_value_heads=t.num_heads, vocab_size=t.vocab_size,
                                         rms_norm_eps=t.rms_norm_eps, max_position_embeddings=t.max_position))
        m.config._attn_implementation = "eager"
        sd = to_hf_state_dict(cfg, P, "hf5_flat")
        lsd = {k.replace("model.language_model.", "model."): v for k, v in sd.items()
               if k.startswith("model.language_model.") or k.startswith("lm_head.")}
        missing, unexpected = m.load_state_dict(lsd, strict=False)
        assert not missing and not unexpected, (missing, unexpected)
        return m.float()

    def llama_grads(m):
        sd = {}
        for n, p in m.named_parameters():
            sd[n.replace("model.", "model.language_model.", 1) if n.startswith("model.") else n] = \
                p.grad if p.grad is not None else torch.zeros_like(p)
        full = to_hf_state_dict(cfg, {k: torch.zeros_like(v) for k, v in params.items()}, "hf5_flat")
        full.update(sd)
        return from_hf_state_dict(cfg, full, "hf5_flat")

    ls, lt_ = build_llama(params), build_llama(tparams).eval()
    for p in lt_.parameters():
        p.requires_grad = False
    Bn, Tn = 2, 20
    tsp = [make_text_split(cfg, Bn, Tn, seed=500 + i, n_question=5, right_pad=[3, 0]) for i in range(3)]
    for name, sp in zip(("forget", "retain", "idk"), tsp):
        for kk, v in sp.items():
            res[f"nlp.in.{name}.{kk}"] = v.numpy()
    tup = [[sp["input_ids"], sp["labels"], sp["attention_mask"]] for sp in tsp]
    cv1 = torch.rand(1, 1, t.hidden_size, generator=torch.Generator().manual_seed(6))
    res["nlp.RMU.rand_vec"] = cv1.numpy()
    real_rand = torch.rand
    for lt in ["grad_ascent", "grad_diff_forget_ce_retain_ce", "grad_diff_forget_KL_retain_ce", "KL", "scrub", "npo", "idk",
               "eco_ft", "dpo", "LLMU", "RMU"]:
        fake = types.SimpleNamespace(loss_type=lt, oracle_model=lt_, loss_beta=2.0 if lt == "npo" else 0.7, l1_lambda=0.0,
                                     l0_lambda=0.0, is_deepspeed_enabled=False,
                                     to_device=nlp_dl.CustomTrainerForgetting.to_device)
        ls.zero_grad()
        if lt == "RMU":
            torch.rand = lambda *a, **k: cv1.clone()  # the trainer draws its control vector with torch.rand
        try:
            loss = nlp_dl.CustomTrainerForgetting.compute_loss(fake, ls, [list(x) for x in tup])
        finally:
            torch.rand = real_rand
        loss.backward()
        record(res, f"nlp.{lt}", loss, cfg, llama_grads(ls))
    # L1 against the oracle (nlp/dataloader.py:460-481)
    fake = types.SimpleNamespace(loss_type="grad_ascent", oracle_model=lt_, loss_beta=0.7, l1_lambda=1e-4, l0_lambda=0.0,
                                 is_deepspeed_enabled=False, to_device=nlp_dl.CustomTrainerForgetting.to_device)
    loss = nlp_dl.CustomTrainerForgetting.compute_loss(fake, ls, [list(x) for x in tup])
    res["nlp.grad_ascent+l1.loss"] = loss.detach().numpy()
    np.savez_compressed(os.path.join(OUT, "compute_loss_extra.npz"), **res)


if __name__ == "__main__":
    main()
