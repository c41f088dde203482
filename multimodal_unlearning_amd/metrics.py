"""Forget-quality / model-utility metrics (fp64 numpy; host-side post-processing, not on the GPU path).

Restates /root/reference/mm/calculate_metrics.py:51-72 (`evaluate_forget_quality`, the metric of record for
the mm benchmark: GLOBAL mean of the perturbed losses) and /root/reference/utils.py:182-198
(`get_forget_quality`, the nlp/TOFU variant: per-sample mean).  Input format = the reference's
eval_log_aggregated.json (mm/eval.py:251-255).  Pinned by tests/golden/forget_quality.json.
"""
from __future__ import annotations

import numpy as np
from scipy.spatial.distance import jensenshannon
from scipy.stats import ks_2samp


def _vals(d: dict) -> np.ndarray:
    return np.array(list(d.values()))


def truth_ratio_mm(forget_log: dict) -> np.ndarray:
    para = _vals(forget_log["avg_paraphrased_loss"])
    pert = _vals(forget_log["average_perturb_loss"]).mean()  # global mean (mm/calculate_metrics.py:58,61)
    return np.exp(pert - para)


def evaluate_forget_quality(unlearned_data: dict, retained_data: dict) -> dict:
    u = truth_ratio_mm(unlearned_data["eval_log_forget.json"])
    r = truth_ratio_mm(retained_data["eval_log_forget.json"])
    return {"KS test p-value": ks_2samp(u, r).pvalue, "JS metric": jensenshannon(u, r)}


def get_forget_quality_nlp(unlearn_result: dict, retain_result: dict) -> dict:
    def tr(log):
        para = _vals(log["avg_paraphrased_loss"])
        pert = _vals(log["average_perturb_loss"]).mean(axis=-1)  # per sample (utils.py:188,192)
        return np.exp(pert - para)

    res = ks_2samp(tr(unlearn_result["eval_log_forget.json"]), tr(retain_result["eval_log_forget.json"]))
    return {"Forget Quality": res.pvalue, "KS Test PVal Forget": res.pvalue, "KS Test Forget": res.statistic}


def forget_quality_reported(js_metric: float) -> float:
    """Reported 'Forget Quality' = 1 - JS metric (mm/build_vllm_table.py:103, mm/build_mod_table.py:68)."""
    return 1.0 - js_metric
