#!/usr/bin/env python3
"""Forget-step throughput on MI355X: samples/s (forget+retain image+QA pairs) of the LLaVA-1.5-7B
grad_diff_forget_ce_retain_ce step (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...; one rank per GPU, RCCL)

A "step" = one optimiser step = GA micro-batches x [compute_loss forward + backward over B pairs] + bucketed
gradient all-reduce (N>1) + global-norm clip + AdamW + grad zeroing.  Random-init weights of the named
architecture, synthetic batches resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

roofline: the dominant kernel is the bf16 MFMA GEMM (>= 95 % of the step's FLOPs).  On the LAST timed step every
GEMM launch is bracketed by HIP events on the launch stream; achieved = sum(2*M*N*K) / sum(duration) over those
launches, peak = 2.5 PFLOP/s dense bf16 (MI355X_MICROARCH.md).  cpu_baseline: the oracle (oracle/llava_ref.py,
bf16 CPU path) timed on the host cores of this box on a bounded full-width sample, extrapolated linearly in depth.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_BF16 = 2.5e15  # dense bf16 MFMA FLOP/s per MI355X


def flops_per_sample(cfg, S, paired=True, lora=False, text_only=False, forward_only=False, head_rows=None, recompute=False):
    """FLOPs of one grad_diff full-FT sample (forget + retain sequence), vision tower frozen: matmul = 2mnk, causal
    attention at half the dense S^2.  head_rows: rows per sequence that go through the lm_head
    (None = all S positions, SURVEY §8d's algorithmic count; the loss-only forward runs the supervised rows only and
    the step is priced with the FLOPs it EXECUTES)."""
    t, v = cfg.text, cfg.vision
    d, I, L, V = t.hidden_size, t.intermediate_size, t.num_layers, t.vocab_size
    lin = 2 * S * (4 * d * d + 3 * d * I) * L
    attn = 2 * 2 * S * S * d * L / 2
    head = 2 * (S if head_rows is None else head_rows) * d * V
    dec = lin + attn + head
    T = v.num_patches + 1
    dv, iv = v.hidden_size, v.intermediate_size
    clip = cfg.vision_layers_used * (2 * T * (4 * dv * dv + 2 * dv * iv) + 2 * 2 * T * T * dv) + 2 * v.num_patches * v.patch_k * dv
    proj = 2 * v.num_patches * (dv * d + d * d)
    if text_only:  # no image: neither the CLIP tower nor the projector runs
        clip = proj = 0
    fwd = dec + clip + proj
    if forward_only:  # one no-grad forward of one sequence (frozen teacher)
        return fwd
    # LoRA: dgrad only (adapter GEMMs are < 1 %), including the dgrad sweep through the adapted CLIP tower (BASELINE.md
    # §2: 30.16 T per sequence); full fine-tuning: dgrad + wgrad of decoder and projector, CLIP frozen (no backward)
    fb = fwd + ((dec + proj + clip) if lora else 2 * (dec + proj))
    if recompute:  # gradient checkpointing: every decoder layer's forward runs a second time inside the backward sweep
        fb += lin + attn
    return (2 if paired else 1) * fb


class GemmTimer:
    """Brackets every hip.gemm launch with HIP events on the current (launch) stream."""

    def __init__(self, hipmod):
        self.hip = hipmod
        self.orig = hipmod.gemm
        self.orig_rope = hipmod.gemm_rope
        self.records = []

    def __enter__(self):
        def timed(layout, A, B, out, **kw):
            s = torch.cuda.current_stream()
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record(s)
            r = self.orig(layout, A, B, out, **kw)
            e1.record(s)
            if layout == self.hip.NT:
                m, k = A.shape
                n = B.shape[0]
            elif layout == self.hip.NN:
                m, k = A.shape
                n = B.shape[1]
            else:
                k, m = A.shape
                n = B.shape[1]
            self.records.append((layout, m, n, k, e0, e1, bool(kw.get("accumulate")), kw.get("resid") is not None))
            return r

        def timed_rope(x, w, out, *a, **kw):  # the q|k|v projection with the fused RoPE epilogue (same kernel)
            s = torch.cuda.current_stream()
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record(s)
            launched = self.orig_rope(x, w, out, *a, **kw)
            e1.record(s)
            if launched:
                self.records.append((self.hip.NT, x.shape[0], w.shape[0], x.shape[1], e0, e1, False, False))
            return launched

        def timed_swiglu(which):
            fn = getattr(self.hip, which)

            def f(*a, **kw):   # the GEMMs with the SwiGLU forward / backward in their epilogue (same 256^2 kernel)
                s = torch.cuda.current_stream()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(s)
                launched = fn(*a, **kw)
                e1.record(s)
                if launched:
                    if which == "gemm_swiglu_fwd":   # (x, w_gu, gu, act, act_t)
                        m, k, n = a[0].shape[0], a[0].shape[1], a[1].shape[0]
                    else:                            # (dh, w, w_t, gu, dgu, dgu_t)
                        m, k, n = a[0].shape[0], a[0].shape[1], a[3].shape[1] // 2
                    self.records.append((self.hip.NT, m, n, k, e0, e1, False, which))
                return launched
            return fn, f

        self.orig_sw = {}
        for which in ("gemm_swiglu_fwd", "gemm_swiglu_bwd"):
            if hasattr(self.hip, which):
                self.orig_sw[which], wrapped = timed_swiglu(which)
                setattr(self.hip, which, wrapped)
        self.hip.gemm = timed
        self.hip.gemm_rope = timed_rope
        return self

    def __exit__(self, *a):
        self.hip.gemm = self.orig
        self.hip.gemm_rope = self.orig_rope
        for which, fn in self.orig_sw.items():
            setattr(self.hip, which, fn)

    def summary(self):
        torch.cuda.synchronize()
        tot_f = tot_t = 0.0
        by = {}
        self.alg_bytes = 0.0
        for layout, m, n, k, e0, e1, acc, res in self.records:
            ms = e0.elapsed_time(e1)
            f = 2.0 * m * n * k
            # algorithmic HBM bytes of one launch: read A and B once, write C once (+ read C / residual once)
            if isinstance(res, str):  # fused SwiGLU GEMMs: fwd writes gu (2I... n = 2I) + act + act^T; bwd reads gu, writes dgu + dgu^T
                self.alg_bytes += 2.0 * (m * k + n * k + (m * n * 2 if res == "gemm_swiglu_fwd" else m * n * 6))
            else:
                self.alg_bytes += 2.0 * (m * k + n * k + m * n * (1 + int(acc) + int(res)))
            tot_f += f
            tot_t += ms * 1e-3
            key = ("NT", "NN", "TN")[layout]
            b = by.setdefault(key, [0, 0.0, 0.0])
            b[0] += 1
            b[1] += f
            b[2] += ms * 1e-3
        # the same sum over the launches with a plain GEMM epilogue only (store / bias+residual / accumulate): the launches
        # that carry the SwiGLU forward / backward in their epilogue move 2-3x the bytes per FLOP and are listed by shape
        self.plain = [0.0, 0.0, 0]
        for layout, m, n, k, e0, e1, acc, res in self.records:
            if not isinstance(res, str):
                self.plain[0] += 2.0 * m * n * k
                self.plain[1] += e0.elapsed_time(e1) * 1e-3
                self.plain[2] += 1
        # the step's GEMM shapes, heaviest first: where the kernel's time goes and what each shape reaches
        shp = {}
        for layout, m, n, k, e0, e1, acc, res in self.records:
            b = shp.setdefault((m, n, k, res[5:] if isinstance(res, str) else ("accumulate" if acc else ("residual" if res else "store"))), [0, 0.0])
            b[0] += 1
            b[1] += e0.elapsed_time(e1) * 1e-3
        self.by_shape = [{"M": m, "N": n, "K": k, "epilogue": ep, "launches": c, "avg_us": round(t / c * 1e6, 1),
                          "tflops": round(2.0 * m * n * k * c / t / 1e12, 1), "share_of_gemm_time": round(t / tot_t, 4)}
                         for (m, n, k, ep), (c, t) in sorted(shp.items(), key=lambda kv: -kv[1][1])[:16]]
        return tot_f, tot_t, len(self.records), {k: {"launches": v[0], "tflops": round(v[1] / v[2] / 1e12, 1),
                                                     "avg_us": round(v[2] / v[0] * 1e6, 1)} for k, v in by.items()}


PEAK_HBM = 8.0e12   # HBM3E bytes/s per MI355X (MI355X_MICROARCH.md)


class KernelTimer:
    """The non-GEMM kernels of the step, timed IN the step like GemmTimer times the GEMMs: every call of the listed
    multimodal_unlearning_amd.hip functions is bracketed by HIP events on the launch stream and priced with its
    algorithmic work (attention: 2 m n k per product, causal at half; HBM-bound kernels: bytes each tensor crosses HBM
    once).  One C call of the attention backward launches two kernels (dQ, then dK/dV): the library re-records a third
    event between them (mmu_attn_set_bwd_mid_event), so both get their own duration."""

    def __init__(self, hipmod):
        self.hip = hipmod
        self.rec = {}     # name -> [launches, seconds-events [(e0, e1)], work per launch list]
        self.orig = {}
        nb = lambda t: t.numel() * t.element_size()
        self.hbm = {
            "rmsnorm_fwd": lambda a, k: 2 * nb(a[0]) + 4 * a[0].shape[0],
            "rmsnorm_bwd": lambda a, k: (3 + (a[4] is not None)) * nb(a[0]),
            "rmsnorm_apply_t": lambda a, k: 2 * nb(a[0]),
            "swiglu_fwd": lambda a, k: nb(a[0]) + nb(a[1]),
            "swiglu_fwd_both": lambda a, k: nb(a[0]) + 2 * nb(a[1]),
            "swiglu_fwd_t": lambda a, k: nb(a[0]) + nb(a[1]),
            "swiglu_bwd": lambda a, k: nb(a[0]) + nb(a[1]) + nb(a[2]),
            "swiglu_bwd_t": lambda a, k: nb(a[0]) + nb(a[1]) + 2 * nb(a[2]),
            "transpose": lambda a, k: 2 * nb(a[0]),
            "ce_fwd": lambda a, k: nb(a[0]),
            "ce_bwd": lambda a, k: 2 * nb(a[0]),
            "adamw": lambda a, k: 24 * a[0].numel(),
            "sumsq": lambda a, k: nb(a[0]),
            "gather_rows": lambda a, k: 2 * nb(a[2]),
            "scatter_rows": lambda a, k: 2 * nb(a[0]),
            "embed_fwd": lambda a, k: 2 * nb(a[4]),
        }

    def _wrap_hbm(self, name, fn, cost):
        def timed(*a, **k):
            s = torch.cuda.current_stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            r = fn(*a, **k)
            e1.record(s)
            self.rec.setdefault(name, []).append(("bytes", float(cost(a, k)), e0, e1))
            return r
        return timed

    def __enter__(self):
        hip = self.hip
        for name, cost in self.hbm.items():
            if hasattr(hip, name):
                self.orig[name] = getattr(hip, name)
                setattr(hip, name, self._wrap_hbm(name, self.orig[name], cost))
        self.orig["attn_fwd"], self.orig["attn_bwd"] = hip.attn_fwd, hip.attn_bwd

        def fwd(q, k, v, out, lse, kmask, B, S, H, D, scale, causal, **kw):
            s = torch.cuda.current_stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            r = self.orig["attn_fwd"](q, k, v, out, lse, kmask, B, S, H, D, scale, causal, **kw)
            e1.record(s)
            fl = 4.0 * B * H * S * S * D * (0.5 if causal else 1.0)
            self.rec.setdefault(f"attn_fwd_kernel<{D},{'causal' if causal else 'full'}>", []).append(("flops", fl, e0, e1))
            return r

        def bwd(q, k, v, o, lse, kmask, dout, delta, dq, dk, dv, B, S, H, D, scale, causal, **kw):
            s = torch.cuda.current_stream()
            e0, em, e1 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            em.record(s)                     # gives the event its handle; the library records it again between the kernels
            e0.record(s)
            hip.attn_set_bwd_mid_event(em)
            r = self.orig["attn_bwd"](q, k, v, o, lse, kmask, dout, delta, dq, dk, dv, B, S, H, D, scale, causal, **kw)
            e1.record(s)
            fwd_fl = 4.0 * B * H * S * S * D * (0.5 if causal else 1.0)
            tag = f"<{D},{'causal' if causal else 'full'}>"
            # executed products: dQ kernel S, dP, dQ (3); dK/dV kernel S, dP, dV, dK (4); algorithmic backward = 5 products
            self.rec.setdefault("attn_bwd_dq_kernel" + tag, []).append(("flops", 1.5 * fwd_fl, e0, em))
            self.rec.setdefault("attn_bwd_dkdv_kernel" + tag, []).append(("flops", 2.0 * fwd_fl, em, e1))
            self.rec.setdefault("attention backward (dq + dkdv, 5 algorithmic products)" + tag, []).append(("flops", 2.5 * fwd_fl, e0, e1))
            return r

        hip.attn_fwd, hip.attn_bwd = fwd, bwd
        return self

    def __exit__(self, *a):
        for name, fn in self.orig.items():
            setattr(self.hip, name, fn)

    def summary(self, top_hbm=6):
        torch.cuda.synchronize()
        rows = []
        for name, recs in self.rec.items():
            kind = recs[0][0]
            t = sum(e0.elapsed_time(e1) for _, _, e0, e1 in recs) * 1e-3
            w = sum(x for _, x, _, _ in recs)
            if t <= 0:
                continue
            peak = PEAK_BF16 if kind == "flops" else PEAK_HBM
            rows.append({"name": name, "bound": "mfma" if kind == "flops" else "hbm", "launches": len(recs),
                         "avg_us": round(t / len(recs) * 1e6, 1), ("flops" if kind == "flops" else "bytes") + "_per_launch": round(w / len(recs)),
                         "achieved": round(w / t / 1e12, 3), "unit": "TFLOP/s" if kind == "flops" else "TB/s",
                         "frac": round(w / t / peak, 4), "total_ms": round(t * 1e3, 2)})
        att = [r for r in rows if r["bound"] == "mfma"]
        hbm = sorted((r for r in rows if r["bound"] == "hbm"), key=lambda r: -r["total_ms"])[:top_hbm]
        return sorted(att, key=lambda r: r["name"]) + hbm


def cpu_baseline(cfg, S, budget_layers=4, clip_layers=4):
    """Oracle (bf16 CPU restatement of the reference path) on the host cores: full-width model cut to
    `budget_layers` decoder layers and `clip_layers` CLIP layers, one forget+retain pair at B=1, fwd+bwd of
    grad_diff; the per-layer terms are extrapolated linearly to the full depth."""
    import copy

    from multimodal_unlearning_amd.params import param_specs
    from multimodal_unlearning_amd.synthetic import make_forget_batch
    from oracle.forget_losses_ref import compute_loss_ref
    from oracle.llava_ref import llava_forward

    # the GPU box gives each GPU a 16-core share of the host (256 hardware threads visible): using more
    # oversubscribes and is 10x slower
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(avail, 16)))
    cores = torch.get_num_threads()

    def build(nl, ncl):
        c = copy.deepcopy(cfg)
        c.text.num_layers = nl
        c.vision.num_layers = ncl + 1  # vision_layers_used = num_layers - 1
        g = torch.Generator().manual_seed(0)
        P = {}
        for name, shape in param_specs(c).items():
            if name.endswith(("norm.weight", "ln1.weight", "ln2.weight", "pre_ln.weight")):
                t = torch.ones(shape, dtype=torch.bfloat16)
            else:
                t = (torch.randn(shape, generator=g) * 0.02).to(torch.bfloat16)
            P[name] = t.requires_grad_(not name.startswith("vision."))
        return c, P

    def run(nl, ncl):
        c, P = build(nl, ncl)
        batch = make_forget_batch(c, 1, S, seed=1, n_question=S - 1 - c.vision.num_patches - 256)
        t0 = time.perf_counter()
        loss = compute_loss_ref("grad_diff_forget_ce_retain_ce",
                                lambda x: llava_forward(c, P, **x, dtype=torch.bfloat16), None, batch, 1.0)
        loss.backward()
        return time.perf_counter() - t0

    run(0, 1)  # warm-up (allocator, thread pool)
    t_clip1 = run(0, 1)
    t_base = run(0, clip_layers)             # embeddings + CLIP(clip_layers) + projector + lm_head + loss
    t_full = run(budget_layers, clip_layers)  # + budget_layers decoder layers
    per_dec = max(1e-6, (t_full - t_base) / budget_layers)
    per_clip = max(0.0, (t_base - t_clip1) / max(1, clip_layers - 1))
    est = t_base + per_dec * cfg.text.num_layers + per_clip * (cfg.vision_layers_used - clip_layers)
    return {"value": round(1.0 / est, 6), "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": (f"oracle/llava_ref.py bf16 on host cores: full-width {budget_layers}/{cfg.text.num_layers} decoder "
                       f"layers + {clip_layers}/{cfg.vision_layers_used} CLIP layers + projector + lm_head + CE, one "
                       f"forget+retain pair (B=1, S={S}) fwd+bwd, measured {t_full:.1f}s, per-layer terms extrapolated "
                       f"linearly to full depth ({est:.0f}s per sample)")}


def hbm_kernels(hip, cfg, n_seq, S, dev, reps=5):
    """each HBM-bound kernel of the step, standalone at the workload's shapes: algorithmic bytes next to
    the measured GB/s (torch events on the stream the C-ABI launches on), vs the 8 TB/s HBM3E peak."""
    BF = torch.bfloat16
    t = cfg.text
    M, d, I, V = n_seq * S, t.hidden_size, t.intermediate_size, t.vocab_size
    g = torch.Generator(device=dev).manual_seed(7)
    rnd = lambda *sh: torch.randn(*sh, device=dev, dtype=BF, generator=g)
    x, dy, dres, y = rnd(M, d), rnd(M, d), rnd(M, d), torch.empty(M, d, device=dev, dtype=BF)
    w = torch.ones(d, device=dev, dtype=BF)
    rstd = torch.empty(M, device=dev, dtype=torch.float32)
    dwp = torch.empty(hip.rmsnorm_bwd_nparts(M) * d, device=dev, dtype=torch.float32)
    dw = torch.zeros(d, device=dev, dtype=BF)
    gu, act, dgu = rnd(M, 2 * I), torch.empty(M, I, device=dev, dtype=BF), torch.empty(M, 2 * I, device=dev, dtype=BF)
    dguT = torch.empty(2 * I, M, device=dev, dtype=BF)
    logits = rnd(M, V)
    dlog = rnd(M, V)
    labels = torch.randint(0, V, (n_seq, S), device=dev, generator=g)
    lse = torch.empty(M, device=dev, dtype=torch.float32)
    lrow = torch.empty(M, device=dev, dtype=torch.float32)
    seg_scale = torch.ones(1, device=dev, dtype=torch.float32)
    seg_cnt = torch.full((1,), float(n_seq * (S - 1)), device=dev, dtype=torch.float32)
    stats = torch.empty(M, 4, device=dev, dtype=torch.float32)
    coef = torch.ones(1, device=dev, dtype=torch.float32)
    NP = 1 << 28
    p, gr = rnd(NP), rnd(NP)
    m1, v1 = torch.zeros(NP, device=dev), torch.zeros(NP, device=dev)
    ss = torch.ones(hip.SUMSQ_FLOATS, device=dev, dtype=torch.float32)
    cases = [
        ("rmsnorm_fwd", 2 * M * d * 2 + M * 4, lambda: hip.rmsnorm_fwd(x, w, y, rstd, 1e-5)),
        ("rmsnorm_bwd (+residual grad, dgamma partials)", 4 * M * d * 2, lambda: hip.rmsnorm_bwd(dy, x, w, rstd, dres, y, dwp, dw, True)),
        ("swiglu_fwd", 3 * M * I * 2, lambda: hip.swiglu_fwd(gu, act)),
        ("swiglu_bwd_t (dgu and dgu^T)", 7 * M * I * 2, lambda: hip.swiglu_bwd_t(gu, act, dgu, dguT)),
        ("ce_fwd (logits once)", M * V * 2, lambda: hip.ce_fwd(logits, labels, lse, lrow, S)),
        ("ce_bwd (logits -> dlogits)", 2 * M * V * 2, lambda: hip.ce_bwd(logits, labels, lse, seg_scale, seg_cnt, M, True, dlog, S)),
        ("rowloss_fwd KL (student+teacher logits)", 2 * M * V * 2, lambda: hip.rowloss_fwd(hip.KL_STUDENT_INPUT, logits, dlog, 1.0, stats)),
        ("adamw (bf16 p,g; fp32 m,v; grad zeroed)", 24 * NP, lambda: hip.adamw(p, gr, m1, v1, 1e-5, 0.9, 0.999, 1e-8, 0.01, 1, ss, 1.0)),
        ("sumsq (grad norm)", 2 * NP, lambda: hip.sumsq(gr, ss)),
    ]
    out = []
    for name, nbytes, fn in cases:
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        out.append({"kernel": name, "algorithmic_bytes": nbytes, "us": round(us, 1),
                    "GBps": round(nbytes / us / 1e3, 1), "frac_of_8TBps": round(nbytes / us / 1e3 / 8000, 3)})
    return out


def launch_ranks(n: int, argv) -> int:
    """One child process per GPU through torch.distributed.run on 127.0.0.1 (free port), same arguments."""
    import socket
    import subprocess

    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this image
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--micro-batch", type=int, default=15,
                    help="forget+retain pairs per GPU per micro-batch; 15 pairs = 30 sequences = 32640 tokens = 127.5 "
                         "256-row GEMM tiles -> whole rounds on 256 CUs, ~210 GB of the 288 GB HBM")
    ap.add_argument("--grad-accum", type=int, default=4,
                    help="micro-batches per optimiser step (reference default, config/mm/forget.yaml:15)")
    ap.add_argument("--model", default="7b", choices=["7b", "13b", "tiny"])
    ap.add_argument("--loss", default="grad_diff_forget_ce_retain_ce")
    ap.add_argument("--layers", type=int, default=0, help="debug: override decoder depth (invalidates the metric)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-batch", action="store_true",
                    help="skip the extra 4 steps at the reference's default micro-batch (2 pairs) after the timed region")
    ap.add_argument("--gemm-group-m", type=int, default=None, help="A/B: GROUP_M rasterisation of the 256^2 GEMM (default 4)")
    ap.add_argument("--gemm-persist", type=int, default=None, help="A/B: 256^2 GEMM launch mode (0 workgroup per tile, 1 persistent = default)")
    ap.add_argument("--gemm-tail-split", type=int, default=None, help="A/B: split-K tail of the 256^2 GEMM (0 off, 1 heuristic = default)")
    ap.add_argument("--keep-act-t", type=int, default=None, help="keep act^T from forward to backward (default: on when it costs <= 24 GB)")
    ap.add_argument("--optim", default="adamw", choices=["adamw", "adamw8bit"],
                    help="adamw: fp32 moments; adamw8bit: block-wise 8-bit moments (the reference's adamw_bnb_8bit)")
    ap.add_argument("--text-only", action="store_true",
                    help="BASELINE config 1 surface on the GPU: the text trainer (nlp/dataloader.py) with tuple inputs "
                         "[B,500], right-padded, no images (use with --lora-r 8 --loss grad_ascent)")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL); gloo + --single-device "
                    "rehearses the N > 1 launch path on a 1-GPU box")
    ap.add_argument("--single-device", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--hbm-kernels", action="store_true",
                    help="also time the step's HBM-bound kernels standalone (bytes next to GB/s) after the timed region")
    ap.add_argument("--dp-mode", default="allreduce", choices=["allreduce", "rs_ag"],
                    help="gradient exchange at N > 1: bucketed all-reduce overlapped with backward (default), or "
                         "reduce-scatter + sharded AdamW state + parameter all-gather (ZeRO-style, state / N)")
    ap.add_argument("--squat-workgroups", type=int, default=0,
                    help="SIMULATION (1 GPU): at the moment the gradient exchange of an N > 1 run would start (first final "
                         "range of the last micro-batch's backward sweep) park this many 256-thread workgroups on a second "
                         "stream for --squat-ms, standing in for RCCL's all-reduce kernels (tests/c_abi/squatter.hip)")
    ap.add_argument("--squat-ms", type=float, default=80.0)
    ap.add_argument("--no-fuse", action="store_true")
    ap.add_argument("--recompute", action="store_true",
                    help="gradient checkpointing (mm/forget.py:123-124): keep one tensor per decoder layer, re-run the layer's "
                         "forward in backward (+1/3 FLOPs, executed and priced)")
    ap.add_argument("--no-overlap-optimizer", action="store_true",
                    help="A/B: AdamW + W^T refresh on the main stream after the step (round-3 behaviour); default: on a side "
                         "stream, chunk by chunk in forward order, under the next step's forward (bit-identical results)")
    ap.add_argument("--no-merge-ga", action="store_true",
                    help="A/B: walk the gradient-accumulation micro-batches one by one (round-3 behaviour); default: micro-batches "
                         "that fit HBM together run as ONE concatenated forward/backward (ForgetEngine._groups)")
    ap.add_argument("--full-logits", action="store_true",
                    help="A/B: every forward materialises all logits (round-2 behaviour); default: forwards of which only "
                         ".loss is read run final norm + lm_head + CE on the supervised rows")
    ap.add_argument("--no-wt", action="store_true", help="do not keep resident W^T copies (saves one weight copy of HBM)")
    ap.add_argument("--lora-r", type=int, default=0, help="LoRA rank (0 = full fine-tuning); BASELINE config 2 uses 8")
    ap.add_argument("--lora-alpha", type=float, default=32.0)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as the driver invokes it: start the N ranks as fresh processes (one per GPU,
        # torch.distributed.run = what `accelerate launch` does for the reference, run_exps.py:36) BEFORE this process
        # loads the HIP library or touches a GPU — never re-exec a GPU-initialised process.  Rank 0's JSON line goes to
        # the inherited stdout; the parent only forwards the exit code.
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    from multimodal_unlearning_amd import hip
    from multimodal_unlearning_amd.config import ForgetConfig, llava_1_5_7b, llava_1_5_13b, tiny_test_config
    from multimodal_unlearning_amd.engine import ForgetEngine, init_distributed
    from multimodal_unlearning_amd.llava import LlavaHIP
    from multimodal_unlearning_amd.synthetic import bench_batch, make_forget_batch
    from multimodal_unlearning_amd.trainer import CustomTrainerForgetting, MMTrainerForgetting

    if args.gemm_tail_split is not None:
        hip.gemm_set_tail_split(args.gemm_tail_split)
    if args.gemm_group_m is not None:
        hip.gemm_set_group_m(args.gemm_group_m)
    if args.gemm_persist is not None:
        hip.gemm_set_persistent(args.gemm_persist)
    if os.environ.get("MMU_XCD_MAP"):  # A/B switch: tile -> XCD map of the 256^2 GEMM (1 = default)
        hip.gemm_set_xcd_map(int(os.environ["MMU_XCD_MAP"]))
    if os.environ.get("MMU_DUMP_AFTER"):  # debugging aid for rehearsals: dump every thread's stack and exit
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["MMU_DUMP_AFTER"]), exit=True)
    if args.single_device:
        os.environ["LOCAL_RANK"] = "0"
    rank, local, world = init_distributed(args.backend)
    assert world == args.gpus or world == 1 and args.gpus == 1, f"launched {world} ranks for --gpus {args.gpus}"
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback for the HIP path)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    cfg = {"7b": llava_1_5_7b, "13b": llava_1_5_13b, "tiny": tiny_test_config}[args.model]()
    if args.layers:
        cfg.text.num_layers = args.layers
    lora = None
    if args.lora_r:
        from multimodal_unlearning_amd.llava import LoraSpec
        lora = LoraSpec(r=args.lora_r, alpha=args.lora_alpha)
    model = LlavaHIP(cfg, device=dev, init="device", seed=0, lora=lora)
    needs_teacher = any(k in args.loss.lower() for k in ("kl", "npo", "scrub", "dpo", "llmu"))
    teacher = LlavaHIP(cfg, device=dev, init="device", seed=1, trainable=False) if needs_teacher else None
    splits = ("forget", "retain", "idk") if any(k in args.loss.lower() for k in ("idk", "dpo", "llmu")) else ("forget", "retain")
    if args.text_only:
        tr = CustomTrainerForgetting(model=model, forget_loss=args.loss, oracle_model=teacher, loss_beta=1.0, l1_lambda=0.0,
                                     l0_lambda=0.0, loss_args={"fuse_pairs": not args.no_fuse, "loss_only_head": not args.full_logits})
    else:
        tr = MMTrainerForgetting(model=model, forget_loss=args.loss, teacher_model=teacher, loss_beta=1.0, l1_lambda=0.0,
                                 l0_lambda=0.0, l_norm_from="zero",
                                 loss_args={"fuse_pairs": not args.no_fuse, "loss_only_head": not args.full_logits})
    fcfg = ForgetConfig(batch_size=args.micro_batch, gradient_accumulation_steps=args.grad_accum)
    eng = ForgetEngine(model, tr, fcfg, total_steps=args.steps + args.warmup + 1, warmup_steps=1,
                       transposed_weights=not args.no_wt, optimizer=args.optim, dp_mode=args.dp_mode,
                       merge_ga=not args.no_merge_ga, overlap_optimizer=not args.no_overlap_optimizer)

    if args.squat_workgroups:
        import ctypes
        import subprocess
        so = os.path.join(ROOT, "tests", "c_abi", "libsquatter.so")
        if not os.path.exists(so):
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "-o", so,
                            os.path.join(ROOT, "tests", "c_abi", "squatter.hip")], check=True)
        sq = ctypes.CDLL(so)
        sq.squat_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p]
        sq.squat_launch.restype = ctypes.c_int
        side = torch.cuda.Stream()

        def squat():  # bounded: at most 1 s per launch (squatter.hip)
            side.wait_stream(torch.cuda.current_stream())  # it starts when the sweep has reached this point, as RCCL would
            assert sq.squat_launch(args.squat_workgroups, 256, int(args.squat_ms * 1e5), ctypes.c_void_p(side.cuda_stream)) == 0
        eng.exchange_start_callback = squat

    B, GA = args.micro_batch, args.grad_accum
    n_seq = B * (2 if (len(splits) >= 2 and not args.no_fuse and args.loss.lower() in ("grad_diff_forget_ce_retain_ce", "eco_ft", "idk")) else 1)
    act_t_bytes = n_seq * (cfg.vision.num_patches + 512) * cfg.text.intermediate_size * 2 * cfg.text.num_layers
    model.keep_act_t = bool(args.keep_act_t) if args.keep_act_t is not None else (args.model == "7b" and act_t_bytes <= 24e9)
    model.recompute_activations = bool(args.recompute)
    if args.text_only:
        from multimodal_unlearning_amd.synthetic import make_text_split

        S = 64 if args.model == "tiny" else 500  # nlp/data_module.py: max_length 500, right padding
        def text_mb(seed):
            sp = [make_text_split(cfg, B, S, seed=seed * 10 + j, n_question=S // 4, right_pad=[(7 * b) % (S // 5) for b in range(B)],
                                  device=dev) for j in range(len(splits))]
            return [(s_["input_ids"], s_["labels"], s_["attention_mask"]) for s_ in sp]
        mbs = [text_mb(1234 + rank * 100 + i) for i in range(GA)]
        act_t_bytes = 0
    elif args.model == "tiny":
        S = 64
        mbs = [make_forget_batch(cfg, B, S, seed=1234 + rank * 100 + i, n_question=8, splits=splits, device=dev)
               for i in range(GA)]
    else:
        S = cfg.vision.num_patches + 512
        mbs = [bench_batch(cfg, B, seed=1234 + rank * 100 + i, splits=splits, device=dev) for i in range(GA)]

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        eng.step(mbs)
    sync()
    eng.time_exchange = world > 1
    t0 = time.perf_counter()
    timer = None
    ktimer = None
    for k in range(args.steps):
        if k == args.steps - 1 and rank == 0:
            timer = GemmTimer(hip)
            ktimer = KernelTimer(hip)
            with timer, ktimer:
                eng.step(mbs)
        else:
            eng.step(mbs)
    sync()
    elapsed = time.perf_counter() - t0
    # per-rank wall time of the timed region and the exposed part of the gradient exchange (compute stream idle between
    # the end of the last backward sweep and the last collective's completion), gathered so that one line carries them
    exch_ms = sum(a.elapsed_time(b) for a, b in eng.exchange_events) / max(1, len(eng.exchange_events))
    mine = torch.tensor([elapsed, exch_ms], device=dev, dtype=torch.float64)
    per_rank = [mine.clone() for _ in range(world)]
    if world > 1:
        dist.all_gather(per_rank, mine)
    rank_elapsed = [float(t[0]) for t in per_rank]
    rank_exch_ms = [float(t[1]) for t in per_rank]
    elapsed = max(rank_elapsed)

    # The same step at the REFERENCE's default per-device batch (2 pairs x GA 4, config/mm/forget.yaml:14-15) on the same
    # model / optimiser state, reported beside the headline (whose micro-batch is sized for 288 GB): 1 warm-up + 3 steps.
    ref_batch = None
    headline_groups = list(eng.last_groups)
    if world == 1 and not args.text_only and args.model != "tiny" and B != 2 and not args.no_reference_batch:
        rb = [bench_batch(cfg, 2, seed=4321 + i, splits=splits, device=dev) for i in range(GA)]
        eng.step(rb)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            eng.step(rb)
        torch.cuda.synchronize()
        ref_batch = 3 * 2 * GA / (time.perf_counter() - t1)
        del rb

    if rank == 0:
        samples = world * B * GA * args.steps
        value = samples / elapsed
        lt = args.loss.lower()
        paired = lt not in ("grad_ascent", "retain_ft", "npo")
        # rows per sequence that reach the lm_head: the supervised positions (incl. the padding slots of the fixed-capacity
        # row list) when every student forward of the loss is a loss-only forward, else all S
        head_rows = None
        if not args.full_logits and lt in ("grad_diff_forget_ce_retain_ce", "grad_ascent", "retain_ft", "idk", "eco_ft"):
            from multimodal_unlearning_amd.llava import supervised_rows
            first = mbs[0]
            labs = [sp[1] for sp in first] if args.text_only else [sp["labels"] for sp in first.values()]
            head_rows = sum(supervised_rows(x.cpu(), cfg.text.vocab_size, "cpu")[2] for x in labs) / len(labs)
        elif not args.full_logits and lt == "kl" and not args.text_only:
            # forget side: loss-only forward (CE); retain side: full logits for the KL term
            from multimodal_unlearning_amd.llava import supervised_rows
            head_rows = (supervised_rows(mbs[0]["forget"]["labels"].cpu(), cfg.text.vocab_size, "cpu")[2] + S) / 2
        fps_alg = flops_per_sample(cfg, S, paired=paired, lora=bool(args.lora_r), text_only=args.text_only)
        fps = flops_per_sample(cfg, S, paired=paired, lora=bool(args.lora_r), text_only=args.text_only, head_rows=head_rows,
                               recompute=args.recompute)
        # teacher forwards (no grad): npo 1, kl 1, scrub 2, grad_diff_*_kl 1 per kl side
        t_fwd = {"npo": 1, "kl": 1, "scrub": 2, "llmu": 1}.get(lt, lt.count("_kl") if lt.startswith("grad_diff") else 0)
        if t_fwd:
            fps += t_fwd * flops_per_sample(cfg, S, text_only=args.text_only, forward_only=True)
            fps_alg += t_fwd * flops_per_sample(cfg, S, text_only=args.text_only, forward_only=True)
        f, t, n, by = timer.summary()
        # HBM-side bytes per GEMM launch (rocprofv3 PMC: FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate passes; counts L2
        # misses incl. Infinity-Cache hits).  PMC collection cannot run inside the timed process, so the figure is
        # collected OFFLINE by tools/pmc_bench_traffic.sh on this same workload and is reported only when that file was
        # made from the GEMM source this library was built from (sha256 of csrc/gemm.hip); otherwise null.
        traffic, traffic_src = None, None
        default_wl = (args.model == "7b" and args.loss == "grad_diff_forget_ce_retain_ce" and not args.lora_r
                      and args.micro_batch == 15 and not args.text_only)
        if default_wl:
            import glob
            import hashlib
            sha = hashlib.sha256(open(os.path.join(ROOT, "multimodal_unlearning_amd", "csrc", "gemm.hip"), "rb").read()).hexdigest()[:16]
            for pmc in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_gemm_traffic.json")), reverse=True):
                pj = json.load(open(pmc))
                if pj.get("gemm_src_sha16") == sha:
                    traffic = round(pj["traffic_bytes_per_launch"])
                    traffic_src = {"collected": "offline", "file": os.path.relpath(pmc, ROOT), "gemm_src_sha16": sha}
                    break
        out = {
            # BASELINE.json's metric for the default workload; other configs name themselves
            "metric": ("forget-step samples/sec (img+text pairs) LLaVA-1.5-7B grad_diff"
                       if (args.model == "7b" and args.loss == "grad_diff_forget_ce_retain_ce" and not args.lora_r and not args.text_only)
                       else f"forget-step samples/sec {'text-only ' if args.text_only else ''}LLaVA-1.5-{args.model} {args.loss}"
                            f"{' LoRA r=%d' % args.lora_r if args.lora_r else ''}"),
            "value": round(value, 4), "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": (f"LLaVA-1.5-{args.model} {args.loss} {'LoRA r=%d' % args.lora_r if args.lora_r else 'full-FT'} (CLIP frozen), S={S} ({cfg.vision.num_patches} image + {S - cfg.vision.num_patches} text "
                                    f"tokens), image {cfg.vision.image_size}, random-init weights") if not args.text_only else
                                   (f"text trainer (nlp/dataloader.py surface) on the Llama-2-{args.model}-shaped decoder, {args.loss} "
                                    f"{'LoRA r=%d' % args.lora_r if args.lora_r else 'full-FT'}, tuple inputs [B,{S}] right-padded, random-init weights"),
                       "micro_batch_pairs_per_gpu": B,
                       "grad_accum": GA, "global_batch": world * B * GA, "seq_len": S, "parallelism": f"dp{world}",
                       "paired_fusion": not args.no_fuse, "overlap_optimizer": bool(eng.overlap_optimizer),
                       # micro-batches evaluated per forward/backward: [1, 1, 1, 1] = the loop, [4] = one merged sweep
                       "ga_groups": headline_groups,
                       "optimizer": ("AdamW fp32 moments" if args.optim == "adamw" else "AdamW block-wise 8-bit moments") + ", clip 1.0",
                       "lora_r": args.lora_r,
                       "keep_act_t": model.keep_act_t, "gradient_checkpointing": bool(args.recompute),
                       "reference_batch_value": (None if ref_batch is None else
                                                 {"micro_batch_pairs_per_gpu": 2, "grad_accum": GA, "ga_groups": list(eng.last_groups),
                                                  "samples_per_s": round(ref_batch, 4),
                                                  "step_frac_of_mfma_peak": round(ref_batch * fps / PEAK_BF16, 4)}),
                       "decoder_layers": cfg.text.num_layers},
            "per_rank_samples_per_s": [round(B * GA * args.steps / t_, 4) for t_ in rank_elapsed],
            "dp": {"mode": args.dp_mode if world > 1 else None, "backend": (dist.get_backend() if world > 1 else None),
                   "gradient_bytes_per_step": model.train_flat.numel() * 2,
                   "exposed_exchange_ms_per_step": [round(x, 2) for x in rank_exch_ms] if world > 1 else None},
            "simulated_exchange": ({"squatter_workgroups": args.squat_workgroups, "ms": args.squat_ms} if args.squat_workgroups else None),
            "peak_hbm_gb": round(torch.cuda.max_memory_allocated() / 1e9, 1),
            # priced with the FLOPs the step EXECUTES (fps); SURVEY §8d's algorithmic count, which charges the lm_head for
            # every position, is reported next to it
            "flops_per_sample_executed": fps, "flops_per_sample_algorithmic": fps_alg, "lm_head_rows_per_sequence": head_rows or S,
            "step_tflops_per_gpu": round(value * fps / world / 1e12, 1),
            "step_frac_of_mfma_peak": round(value * fps / world / PEAK_BF16, 4),
            "roofline": {"bound": "mfma", "achieved": round(f / t / 1e12, 1), "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s",
                         "frac": round(f / t / PEAK_BF16, 4), "traffic": traffic, "traffic_unit": "bytes per launch",
                         "traffic_source": traffic_src, "algorithmic_bytes_per_launch": round(timer.alg_bytes / n),
                         "kernel": "g256::gemm256_kernel / gemm_bf16_kernel (all layouts)",
                         "launches": n, "avg_launch_us": round(t / n * 1e6, 1),
                         # (launches with a plain store / bias+residual / accumulate epilogue alone — the figure rounds 1-3
                         # reported, before SwiGLU moved into two GEMM epilogues per layer)
                         "plain_epilogue_launches": {"launches": timer.plain[2],
                                                     "achieved": round(timer.plain[0] / max(timer.plain[1], 1e-12) / 1e12, 1),
                                                     "frac": round(timer.plain[0] / max(timer.plain[1], 1e-12) / PEAK_BF16, 4)},
                         "gemm_share_of_step": round(t / (elapsed / args.steps), 3), "by_layout": by,
                         "by_shape": timer.by_shape,
                         # the other kernels of the same (last timed) step, event-timed in place: attention forward / dQ /
                         # dK-dV against the MFMA peak, the heaviest HBM-bound kernels against 8 TB/s
                         "kernels": ktimer.summary()},
        }
        if args.hbm_kernels and world == 1:
            torch.cuda.empty_cache()
            out["hbm_kernels"] = hbm_kernels(hip, cfg, 2 * B if paired else B, S, dev)
        if not args.no_cpu_baseline and args.model != "tiny" and world == 1:  # rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(cfg, S)
        elif args.model == "tiny":
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
