"""ORACLE (test infrastructure, never shipped in the product path).

CPU restatement of the reference's loss dispatcher MMTrainerForgetting.compute_loss
(/root/reference/mm/trainer.py:168-483) and its helpers (/root/reference/mm/trainer_utils.py:16-21,
48-70, 73-81), written against two callables instead of HF modules:

    student(split_inputs) -> {"loss": 0-dim tensor or None, "logits": [B,S,V]}
    teacher(split_inputs) -> same, evaluated under no_grad by this code

PINNED against the imported reference itself (MMTrainerForgetting.compute_loss called unbound on a tiny
random LLaVA): tests/golden/compute_loss_tiny.npz written by oracle/make_golden.py,
checked by tests/test_oracle_cpu.py.  Quirks are replicated on purpose: KL "batchmean"
divides by every B*S row (no label mask); scrub/llmu pass the teacher as `input`; the dpo branch's loss is
overwritten twice and ends as mean_b(sum_t CE_idk); NPO is a mean over all B*S*V softmax entries.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def logits2probs(logits, log=True):  # mm/trainer_utils.py:16-21
    p = F.log_softmax(logits, dim=-1) if log else F.softmax(logits, dim=-1)
    return p.view(-1, logits.shape[-1])


def get_batch_loss(output, labels):  # mm/trainer_utils.py:73-81
    shifted = labels[..., 1:].contiguous()
    out = output[..., :-1, :].contiguous()
    return F.cross_entropy(out.transpose(-1, -2), shifted, ignore_index=-100, reduction="none").sum(dim=-1)


def remove_image_tokens(input_ids, logits, image_token_id):  # mm/trainer_utils.py:48-70
    search = (input_ids == image_token_id).nonzero(as_tuple=True)
    has = search[0].tolist()
    idx = search[1].tolist()
    B, T = input_ids.shape
    res = []
    for i in range(B):
        if i not in has:
            res.append(logits[i, -T:, :])
            continue
        start = idx[has.index(i)]
        right = T - start
        res.append(torch.cat([logits[i, :start, :], logits[i, -right:, :]]))
    return torch.stack(res)


def _kl(inp_logp, tgt_logp):  # F.kl_div(input, target, reduction="batchmean", log_target=True)
    return F.kl_div(inp_logp, tgt_logp, reduction="batchmean", log_target=True)


def compute_loss_ref(loss_type: str, student, teacher, inputs, loss_beta=1.0, image_token_index=None,
                     loss_args=None):
    """mm/trainer.py:168-449 (the L1/L0 regulariser :451-481 is restated separately in l1l0_ref)."""
    lt = loss_type.lower()  # mm/trainer.py:114
    f, r, idk = inputs.get("forget"), inputs.get("retain"), inputs.get("idk")

    def T(x):
        with torch.no_grad():
            return teacher(x)

    if lt == "retain_ft":  # :179-181
        return student(r)["loss"]
    if lt == "grad_ascent":  # :183-185
        return -1 * student(f)["loss"]
    if lt.startswith("grad_diff"):  # :187-229
        fo = student(f)
        ro = student(r)
        if "forget_ce" in lt:
            fl = fo["loss"]
        elif "forget_entropy" in lt:
            p = logits2probs(fo["logits"], log=False)
            fl = torch.sum(p * torch.log(p))
        elif "forget_kl" in lt:
            fl = _kl(logits2probs(fo["logits"]), logits2probs(T(f)["logits"]))
        else:
            raise ValueError(f"Invalid loss type on forget {lt}")
        if "retain_ce" in lt:
            rl = ro["loss"]
        elif "retain_kl" in lt:
            rl = _kl(logits2probs(ro["logits"]), logits2probs(T(r)["logits"]))
        else:
            raise ValueError(f"Invalid loss type on retain {lt}")
        return -1 * loss_beta * fl + rl
    if lt == "scrub":  # :231-257  (input = teacher, target = student)
        fo = student(f)
        klf = _kl(logits2probs(T(f)["logits"]), logits2probs(fo["logits"]))
        ro = student(r)
        klr = _kl(logits2probs(T(r)["logits"]), logits2probs(ro["logits"]))
        return -1 * loss_beta * klf + klr + ro["loss"]
    if lt == "kl":  # :259-277
        fl = -1 * student(f)["loss"]
        tr = logits2probs(T(r)["logits"])
        rl = _kl(logits2probs(student(r)["logits"]), tr)
        return fl + rl
    if lt == "llmu":  # :279-300
        fl = -1 * student(f)["loss"]
        rnd = student(idk)["loss"]
        ro = student(r)
        rl = _kl(logits2probs(T(r)["logits"]), logits2probs(ro["logits"]))
        return fl + rl + rnd
    if lt == "npo":  # :343-353
        fo = student(f)
        tp = logits2probs(T(f)["logits"], log=False)
        sp = logits2probs(fo["logits"], log=False)
        ratios = sp / (tp + 1e-8)
        return (2 / loss_beta) * torch.mean(torch.log(1 + torch.pow(ratios, loss_beta)))
    if lt == "idk":  # :355-358
        return student(r)["loss"] + student(idk)["loss"]
    if lt == "eco_ft":  # :359-362
        return loss_beta * student(f)["loss"] + student(r)["loss"]
    if lt == "dpo":  # :364-398 — effective loss after the two overwrites at :395-396
        io = student(idk)
        _ = student(f)  # dead compute in the reference (kept for cost parity only)
        lg = remove_image_tokens(idk["input_ids"], io["logits"], image_token_index)
        idk_loss_current = -1 * get_batch_loss(lg, idk["labels"])
        return -idk_loss_current.mean()
    if lt == "rmu":  # :302-326 — `student(x, capture_layer=k)["hidden"]` stands in for the forward hook on layers[k]
        la = loss_args or {}
        k = int(la.get("rmu_layer", 7))
        fa = student(f, capture_layer=k)["hidden"]
        cv = la["control_vec"]  # [B,1,H]; the reference draws it with torch.rand on first use and caches it
        fl = F.mse_loss(fa.mean(dim=1), cv) * loss_beta  # [B,H] vs [B,1,H]: broadcasts to [B,B,H] as in the reference
        ra = student(r, capture_layer=k)["hidden"]
        with torch.no_grad():
            oa = teacher(r, capture_layer=k)["hidden"]
        return fl + F.mse_loss(ra, oa)
    if lt == "sku":  # :412-446
        la = loss_args or {}
        bad = student(f)["loss"]
        rnd_in = inputs["random"]
        K = rnd_in["input_ids"].shape[1]
        rnd = 0.0
        for i in range(K):
            rnd = rnd + student({k: v[:, i] for k, v in rnd_in.items()})["loss"]
        tr = logits2probs(T(r)["logits"])
        normal = -1 * _kl(logits2probs(student(r)["logits"]), tr)
        return la["bad_weight"] * bad + la["random_weight"] * rnd + la["normal_weight"] * normal
    raise ValueError(f"Invalid loss type {lt}")  # :449


def l1l0_ref(params, teacher_params, l1_lambda, l0_lambda, l_norm_from, has_lora=False):
    """mm/trainer.py:451-481 over lists of trainable tensors."""
    if l_norm_from == "init" and not has_lora:
        if teacher_params is None:
            raise ValueError("Teacher model is required for L1\\L0 regularization during training without LORA!")
        flat = [(p - t).view(-1) for p, t in zip(params, teacher_params)]
    else:
        if l_norm_from == "zero" and has_lora:
            raise ValueError("L1\\L0 regularization is not supported for LORA!")
        flat = [p.view(-1) for p in params]
    cat = torch.cat(flat)
    out = 0.0
    if l1_lambda:
        out = out + l1_lambda * torch.norm(cat, p=1)
    if l0_lambda:
        out = out + l0_lambda * torch.norm(cat, p=0)
    return out


def compute_loss_nlp_ref(loss_type: str, student, oracle, inputs, loss_beta=1.0, rand_vec=None, rmu_layer=7):
    """nlp/dataloader.py:144-458 (text trainer, CASE-SENSITIVE names, tuple inputs already turned into dicts
    {"input_ids","attention_mask","labels"} by the caller: [forget, retain(, idk)])."""
    f, r, *rest = inputs
    idk = rest[0] if rest else None
    lt = loss_type

    def T(x, **kw):
        with torch.no_grad():
            return oracle(x, **kw)

    if lt == "retain_ft":
        return student(r)["loss"]
    if lt == "grad_ascent":
        return -1 * student(f)["loss"]
    if lt.startswith("grad_diff"):  # :167-225
        fo, ro = student(f), student(r)
        if "forget_ce" in lt:
            fl = fo["loss"]
        elif "forget_entropy" in lt:
            p = logits2probs(fo["logits"], log=False)
            fl = torch.sum(p * torch.log(p))
        elif "forget_KL" in lt:
            fl = _kl(logits2probs(fo["logits"]), logits2probs(T(f)["logits"]))
        else:
            raise ValueError(f"Invalid loss type on forget {lt}")
        if "retain_ce" in lt:
            rl = ro["loss"]
        elif "retain_KL" in lt:
            rl = _kl(logits2probs(ro["logits"]), logits2probs(T(r)["logits"]))
        else:
            raise ValueError(f"Invalid loss type on retain {lt}")
        return -1 * loss_beta * fl + rl
    if lt == "scrub":  # :227-269
        fo = student(f)
        klf = _kl(logits2probs(T(f)["logits"]), logits2probs(fo["logits"]))
        ro = student(r)
        klr = _kl(logits2probs(T(r)["logits"]), logits2probs(ro["logits"]))
        return -1 * loss_beta * klf + klr + ro["loss"]
    if lt == "KL":  # :271-301
        fl = -1 * student(f)["loss"]
        tr = logits2probs(T(r)["logits"])
        return fl + _kl(logits2probs(student(r)["logits"]), tr)
    if lt == "LLMU":  # :303-337
        fl = -1 * student(f)["loss"]
        rnd = student(idk)["loss"]
        ro = student(r)
        return fl + _kl(logits2probs(T(r)["logits"]), logits2probs(ro["logits"])) + rnd
    if lt == "RMU":  # :339-367 — fresh [1,1,H] control vector each call
        fa = student(f, capture_layer=rmu_layer)["hidden"]
        cv = rand_vec / torch.norm(rand_vec) * 300
        fl = F.mse_loss(fa, cv.expand_as(fa)) * loss_beta
        ra = student(r, capture_layer=rmu_layer)["hidden"]
        oa = T(r, capture_layer=rmu_layer)["hidden"]
        return fl + F.mse_loss(ra, oa)
    if lt == "npo":  # :384-401
        fo = student(f)
        tp = logits2probs(T(f)["logits"], log=False)
        sp = logits2probs(fo["logits"], log=False)
        pi = torch.log(sp / tp)
        return 2 / loss_beta * torch.mean(torch.log(1 + pi ** loss_beta))
    if lt == "idk":  # :403-410 — ONE forward over cat(forget, retain), one mean CE
        cat = {k: torch.cat((f[k], r[k]), dim=0) for k in f}
        return student(cat)["loss"]
    if lt == "eco_ft":
        return loss_beta * student(f)["loss"] + student(r)["loss"]
    if lt == "dpo":  # :425-456 — effective loss after the overwrites
        io = student(idk)
        return -(-1 * get_batch_loss(io["logits"], idk["labels"])).mean()
    raise ValueError(f"Invalid loss type {lt}")
