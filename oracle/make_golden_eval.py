"""ORACLE fixture generator #3 (build container only): the reference's eval truth-ratio pass
(/root/reference/mm/eval.py:46-92 eval_perturbation_ratio) on a tiny HF LLaVA -> tests/golden/eval_tiny.json.
Usage: PYTHONPATH=/root/reference:/root/repo python oracle/make_golden_eval.py
Absent third-party modules (hydra, rouge_score, natsort) are stubbed for the import only."""
import importlib.machinery
import json
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multimodal_unlearning_amd.config import tiny_test_config  # noqa: E402
from multimodal_unlearning_amd.params import init_params  # noqa: E402
from multimodal_unlearning_amd.synthetic import make_split  # noqa: E402
from oracle.make_golden import build_hf  # noqa: E402


def stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m


def main():
    stub("hydra", main=lambda **k: (lambda f: f))
    stub("rouge_score")
    stub("rouge_score.rouge_scorer", RougeScorer=lambda *a, **k: None)
    stub("natsort", natsorted=sorted)
    sys.path.insert(0, "/root/reference/mm")
    import eval as ref_eval  # /root/reference/mm/eval.py

    cfg = tiny_test_config()
    hf = build_hf(cfg, init_params(cfg, seed=0)).float().eval()
    hf.config.image_token_id = cfg.image_token_index

    class M(torch.nn.Module):
        def __init__(s):
            super().__init__()
            s.m, s.config, s.device = hf, hf.config, torch.device("cpu")

        def forward(s, **kw):
            return s.m(**kw)

    B, K, T = 3, 4, 24
    batches, pbatches = [], []
    for it in range(2):
        sp = make_split(cfg, B, T, seed=700 + it, n_question=6)
        sp["indices"] = torch.arange(B) + it * B
        ps = [make_split(cfg, B, T, seed=800 + 10 * it + k, n_question=6) for k in range(K)]
        pb = {k: torch.stack([p[k] for p in ps], dim=1) for k in ps[0]}
        pb["pixel_values"] = pb["pixel_values"].reshape(B * K, *ps[0]["pixel_values"].shape[1:])  # images are not [B,K]-shaped
        pb["indices"] = sp["indices"].clone()
        batches.append(sp)
        pbatches.append(pb)
    save_in = [{k: v.tolist() if k != "pixel_values" else None for k, v in b.items()} for b in batches]
    logs = ref_eval.eval_perturbation_ratio([dict(b) for b in batches], [dict(b) for b in pbatches], M(), None, None)
    out = {"seeds": {"B": B, "K": K, "T": T}, "logs": {k: {str(i): v for i, v in d.items()} for k, d in logs.items()}}
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "eval_tiny.json"), "w"))
    print({k: list(v.items())[:2] for k, v in logs.items() if k in ("truth_ratio", "avg_paraphrased_loss")})


if __name__ == "__main__":
    main()
