"""Evaluation truth-ratio pass on the HIP model.

Produces the `eval_log_forget.json` section of `eval_log_aggregated.json` that mm/eval.py:46-92
(`eval_perturbation_ratio`) writes and mm/calculate_metrics.py:51-72 reads: per sample index, the ground-truth /
paraphrased answer's CE (sum and per token), the K perturbed answers' CE (sum and per token) and
truth_ratio = exp(gt per token - mean_k perturbed per token).  Same keys, same normalisers (token counts are the
number of labels != -100 of the UNSHIFTED row), same values.

MI355X shape of the pass: every sequence of a batch — the B ground-truth rows and the B*K perturbed rows — goes through
ONE forward each way and ONE C-ABI call (`mmu_seq_ce_stats`: fused row log-sum-exp + target gather, then a per-sequence
reduction returning {CE sum, shifted-target count, label count}); the statistics of all batches stay on the device and
the ratios are formed there; a single device->host copy at the end fills the log dictionaries.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Iterable

import torch

from . import hip
from . import trainer_utils as tu

LOG_KEYS = ("average_perturb_loss", "avg_paraphrased_loss", "truth_ratio", "paraphrased_loss", "perturb_loss",
            "num_token_paraphrased", "num_token_perturb")


def sequence_ce_stats(model, split: Dict[str, torch.Tensor]) -> torch.Tensor:
    """float32 [n, 3] on the device: per sequence {sum_t CE, #shifted targets != -100, #labels != -100}.

    The pass reads CE on supervised positions only, so a model that can (`LlavaHIP.compact_logits`) runs the final norm
    and the lm_head on those rows alone — [n * cap, V] instead of [n * S, V] logits — and the CE kernels take the
    compact rows with their own (already shifted) targets.  Same values as the full pass up to summation order.  Foreign
    models, and ids that still carry ONE <image> id per picture (transformers 4.45.1 expands them inside the model, so the
    logits are longer than the ids and `remove_image_tokens` has work to do), keep the full-logits path."""
    dev = next(model.parameters()).device
    labels = split["labels"]
    if hasattr(model, "compact_logits"):
        from .llava import host_of

        with torch.no_grad():
            lc, tgt, cap = model.compact_logits(**{k: v for k, v in split.items()})
        n = labels.shape[0]
        M, V = lc.shape
        lse = torch.empty(M, device=dev, dtype=torch.float32)
        lrow = torch.empty(M, device=dev, dtype=torch.float32)
        hip.ce_fwd(lc, tgt, lse, lrow, cap, shift=False)
        out = torch.empty(3, n, device=dev, dtype=torch.float32)
        hip.seg_reduce(lrow, tgt, cap, M, V, cap, False, out[0], out[1])     # CE sums, #shifted targets
        host = host_of(labels, fetch=True)                                    # #labels != -100 of the UNSHIFTED row:
        out[2].copy_((host != -100).sum(dim=1).to(torch.float32), non_blocking=True)  # index bookkeeping on the host view
        return out.t().contiguous()
    split = {k: v.to(dev) for k, v in split.items()}
    with torch.no_grad():
        logits = model(**{k: v for k, v in split.items() if k != "labels"}).logits
    logits = tu.remove_image_tokens(split["input_ids"], logits, model.config.image_token_index)
    return hip.seq_ce_stats(logits.contiguous(), split["labels"].to(torch.int64).contiguous())


def _flatten_perturbed(pb: Dict[str, torch.Tensor]):
    """[B, K, ...] token tensors -> [B*K, ...]; pixel_values arrive already flat ([B*K, 3, H, W]) or as [B, K, 3, H, W]."""
    B, K = pb["input_ids"].shape[:2]
    out = {}
    for k, v in pb.items():
        if k == "pixel_values":
            out[k] = v.reshape(-1, *v.shape[-3:])
        else:
            out[k] = v.reshape(B * K, *v.shape[2:])
    return out, B, K


def eval_perturbation_ratio(eval_dataloader: Iterable, perturb_dataloader: Iterable, model) -> "OrderedDict[str, dict]":
    idx_all, gt_all, pt_all = [], [], []
    for batch, perturb_batch in zip(eval_dataloader, perturb_dataloader):
        batch, perturb_batch = dict(batch), dict(perturb_batch)
        idx_all.append(batch.pop("indices").reshape(-1).to(torch.int64).cpu())
        perturb_batch.pop("indices", None)
        flat, B, K = _flatten_perturbed(perturb_batch)
        gt_all.append(sequence_ce_stats(model, batch))                   # [B, 3]
        pt_all.append(sequence_ce_stats(model, flat).view(B, K, 3))      # [B, K, 3]
    logs: "OrderedDict[str, dict]" = OrderedDict((k, {}) for k in LOG_KEYS)
    if not idx_all:
        return logs
    gt, pt = torch.cat(gt_all), torch.cat(pt_all)
    gt_per_tok = gt[:, 0] / gt[:, 2]
    pt_per_tok = pt[..., 0] / pt[..., 2]
    truth_ratio = torch.exp(gt_per_tok - pt_per_tok.mean(-1))
    # one copy to the host for everything the log holds
    host = torch.cat([gt[:, 0:1], gt[:, 2:3], gt_per_tok[:, None], truth_ratio[:, None], pt[..., 0], pt[..., 2],
                      pt_per_tok], dim=1).cpu()
    K = pt.shape[1]
    cols = {"paraphrased_loss": host[:, 0], "num_token_paraphrased": host[:, 1].round().to(torch.int64),
            "avg_paraphrased_loss": host[:, 2], "truth_ratio": host[:, 3], "perturb_loss": host[:, 4:4 + K],
            "num_token_perturb": host[:, 4 + K:4 + 2 * K].round().to(torch.int64),
            "average_perturb_loss": host[:, 4 + 2 * K:4 + 3 * K]}
    index = torch.cat(idx_all).tolist()
    for key, col in cols.items():
        logs[key] = dict(zip(index, col.tolist()))
    return logs
