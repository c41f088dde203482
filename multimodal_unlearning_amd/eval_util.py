"""Evaluation truth-ratio pass on the HIP model: host-side mirror of mm/eval.py:46-92 `eval_perturbation_ratio`
(same log keys, same arithmetic: per-sequence CE sums via get_batch_loss, per-token normalisation by the count of
non-ignored UNSHIFTED labels, truth_ratio = exp(gt/tok - mean_k perturb/tok)).  Its output feeds
metrics.evaluate_forget_quality (the `eval_log_forget.json` section of eval_log_aggregated.json)."""
from __future__ import annotations

from collections import defaultdict

import torch

from . import trainer_utils as tu


def eval_perturbation_ratio(eval_dataloader, perturb_dataloader, model):
    eval_logs = defaultdict(dict)
    dev = next(model.parameters()).device
    image_token_id = model.config.image_token_index
    for batch, perturb_batch in zip(eval_dataloader, perturb_dataloader):
        batch, perturb_batch = dict(batch), dict(perturb_batch)
        indices = batch.pop("indices")
        perturb_batch.pop("indices")
        bsz, num_seq = perturb_batch["input_ids"].shape[0:2]
        for k, v in perturb_batch.items():
            if v.shape[:2] == (bsz, num_seq):
                perturb_batch[k] = v.reshape(bsz * num_seq, *v.shape[2:]) if k == "pixel_values" else v.reshape(bsz * num_seq, -1)
        batch = {k: v.to(dev) for k, v in batch.items()}
        perturb_batch = {k: v.to(dev) for k, v in perturb_batch.items()}
        with torch.no_grad():
            outputs = model(**batch)
            perturb_outputs = model(**perturb_batch)
            out_logits = tu.remove_image_tokens(batch["input_ids"], outputs.logits, image_token_id)
            gt_loss = tu.get_batch_loss(out_logits, batch["labels"])
            pl = tu.remove_image_tokens(perturb_batch["input_ids"], perturb_outputs.logits, image_token_id)
            perturb_loss = tu.get_batch_loss(pl, perturb_batch["labels"]).view(bsz, num_seq)
        num_token_gt = (batch["labels"] != -100).sum(-1)
        num_token_perturb = (perturb_batch["labels"] != -100).view(bsz, num_seq, -1).sum(-1)
        perturb_loss_per_token = perturb_loss / num_token_perturb
        gt_loss_per_token = gt_loss / num_token_gt
        truth_ratio = torch.exp(gt_loss_per_token - perturb_loss_per_token.mean(-1))
        idx_list = indices.tolist()
        eval_logs["average_perturb_loss"].update(dict(zip(idx_list, perturb_loss_per_token.tolist())))
        eval_logs["avg_paraphrased_loss"].update(dict(zip(idx_list, gt_loss_per_token.tolist())))
        eval_logs["truth_ratio"].update(dict(zip(idx_list, truth_ratio.tolist())))
        eval_logs["paraphrased_loss"].update(dict(zip(idx_list, gt_loss.tolist())))
        eval_logs["perturb_loss"].update(dict(zip(idx_list, perturb_loss.tolist())))
        eval_logs["num_token_paraphrased"].update(dict(zip(idx_list, num_token_gt.tolist())))
        eval_logs["num_token_perturb"].update(dict(zip(idx_list, num_token_perturb.tolist())))
    return eval_logs
