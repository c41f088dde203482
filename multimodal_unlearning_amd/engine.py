"""Step loop around compute_loss: gradient accumulation, data-parallel gradient all-reduce (RCCL over xGMI),
global-norm clipping, fused AdamW, linear warmup/decay.

Replaces what the reference inherits from transformers.Trainer + accelerate/DDP for the forget step
(mm/forget.py:139-190: bf16, AdamW lr/weight_decay, warmup = one epoch of steps, GA; `accelerate launch` one
process per GPU).  One process per GPU; `torch.distributed` backend "nccl" (= RCCL) on GPUs, "gloo" for the
CPU plumbing mode and the world_size-2 tests.

Data parallelism: samples are independent, the only exchange is the gradient mean once per optimiser step.
The flat bf16 gradient buffer is reduced in per-decoder-layer buckets (~405 MB at 7B) that are launched from
the model's backward sweep on the LAST micro-batch as soon as a layer's wgrads are final, so the collectives
overlap the rest of backward.  xGMI is point-to-point: large buckets keep each of the 7 links streaming.
The reduction is SUM; the 1/world factor is folded into the AdamW kernel's gradient scale (no extra pass).
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional

import torch
import torch.distributed as dist

from .config import ForgetConfig


def linear_schedule(step: int, base_lr: float, warmup_steps: int, total_steps: int) -> float:
    """transformers get_linear_schedule_with_warmup evaluated at optimiser step `step` (0-based)."""
    if step < warmup_steps:
        return base_lr * step / max(1, warmup_steps)
    return base_lr * max(0.0, (total_steps - step) / max(1, total_steps - warmup_steps))


def steps_from_reference_formula(n_samples: int, batch_size: int, ga: int, world: int, epochs: int):
    """mm/forget.py:65-68,147-148: max_steps = epochs*len // (bs*GA*world); warmup = steps_per_epoch."""
    per_epoch = n_samples // (batch_size * ga * world)
    return max(1, epochs * n_samples // (batch_size * ga * world)), max(1, per_epoch)


def shard_indices(n: int, rank: int, world: int) -> List[int]:
    """Disjoint per-rank dataset indices (DistributedSampler order without shuffling): rank, rank+world, ..."""
    return list(range(rank, n, world))


class GradBucketReducer:
    """Asynchronous SUM all-reduce of [lo,hi) ranges of one flat gradient tensor.  Works on CPU (gloo) and GPU."""

    def __init__(self, flat_grad: torch.Tensor, group=None, max_bucket_elems: int = 1 << 28, force: bool = False):
        self.flat = flat_grad
        self.group = group
        self.max_bucket = max_bucket_elems
        self.works = []
        self.reduced_elems = 0
        self.done = []  # [lo, hi) intervals already handed to a collective in this step
        # force=True runs the collectives even on a 1-rank group (exercises the RCCL path on a single GPU)
        self.enabled = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force)
        self._gloo = self.enabled and dist.get_backend(group) == "gloo"

    def reduce_range(self, lo: int, hi: int):
        if not self.enabled:
            return
        self.done.append((lo, hi))
        while lo < hi:
            e = min(hi, lo + self.max_bucket)
            # RCCL runs the buckets in stream order, any number may be in flight.  gloo (CPU plumbing mode, rehearsals)
            # executes async works on a small thread pool: with >= 4 ranks and many works in flight the ranks' pools
            # pick different buckets and wait for each other forever (seen with 4 ranks on one GPU) -> at most 2 there.
            if self._gloo and len(self.works) >= 2:
                self.works.pop(0).wait()
            self.works.append(dist.all_reduce(self.flat[lo:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self.reduced_elems += e - lo
            lo = e

    def reduce_rest(self):
        """All-reduce whatever no hook covered (a loss whose last sweep never ran — dead forwards keep the model's
        live-sweep count above zero — or a model without hooks): the complement of the intervals seen so far."""
        if not self.enabled:
            return
        pos, n, gaps = 0, self.flat.numel(), []
        for lo, hi in sorted(self.done):
            if lo > pos:
                gaps.append((pos, lo))
            pos = max(pos, hi)
        if pos < n:
            gaps.append((pos, n))
        for lo, hi in gaps:
            self.reduce_range(lo, hi)

    def wait(self):
        self.done = []
        for w in self.works:
            w.wait()
        self.works = []
        n = self.reduced_elems
        self.reduced_elems = 0
        return n


@dataclass
class StepStats:
    loss: float
    lr: float
    step: int


class ForgetEngine:
    """engine.step(micro_batches) = GA x (compute_loss -> backward) -> all-reduce -> clip -> AdamW -> zero grads."""

    def __init__(self, model, trainer, cfg: ForgetConfig, total_steps: int = 1000, warmup_steps: int = 0,
                 group=None, transposed_weights: bool = True, force_collectives: bool = False,
                 optimizer: str = "adamw"):
        from . import hip  # HIP optimiser kernels; fails loudly when the library is missing

        self.hip = hip
        self.model, self.trainer, self.cfg = model, trainer, cfg
        self.total_steps, self.warmup_steps = total_steps, warmup_steps
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        dev = model.device
        self.grad = model.ensure_grad()
        n = self.grad.numel()
        # "adamw": fp32 moments (8 B/param).  "adamw8bit": block-wise 8-bit moments as in the reference's
        # optim="adamw_bnb_8bit" (mm/forget.py:156): 2 B/param + one fp32 absmax per 2048 parameters and moment.
        if optimizer not in ("adamw", "adamw8bit"):
            raise ValueError(f"unknown optimizer {optimizer}")
        self.optimizer = optimizer
        if optimizer == "adamw":
            self.m = torch.zeros(n, device=dev, dtype=torch.float32)
            self.v = torch.zeros(n, device=dev, dtype=torch.float32)
        else:
            self.map1 = hip.dynamic_map(True).to(dev)
            self.map2 = hip.dynamic_map(False).to(dev)
            zero1 = int((self.map1 == 0).nonzero()[0])
            zero2 = int((self.map2 == 0).nonzero()[0])
            self.c1 = torch.full((n,), zero1, device=dev, dtype=torch.uint8)
            self.c2 = torch.full((n,), zero2, device=dev, dtype=torch.uint8)
            nb = hip.adamw8bit_nblocks(n)
            self.am1 = torch.zeros(nb, device=dev, dtype=torch.float32)
            self.am2 = torch.zeros(nb, device=dev, dtype=torch.float32)
        self.sumsq = torch.zeros(hip.SUMSQ_FLOATS, device=dev, dtype=torch.float32)  # [0] = |g|^2, rest scratch
        self.opt_step = 0
        self.reducer = GradBucketReducer(self.grad, group, force=force_collectives)
        if transposed_weights:
            model.enable_transposed_weights()

    def step(self, micro_batches: List[Dict], sync_loss: bool = False) -> StepStats:
        ga = len(micro_batches)
        model = self.model
        total = None
        for i, mb in enumerate(micro_batches):
            last = i == ga - 1
            model.grad_ready_hook = self.reducer.reduce_range if (last and self.reducer.enabled) else None
            model.live_sweeps = 0
            loss = self.trainer.compute_loss(model, mb)
            (loss / ga).backward()
            d = loss.detach().float()
            total = d if total is None else total + d
        model.grad_ready_hook = None
        self.reducer.reduce_rest()
        self.reducer.wait()
        self.opt_step += 1
        lr = linear_schedule(self.opt_step, self.cfg.lr, self.warmup_steps, self.total_steps)
        self.hip.sumsq(self.grad, self.sumsq)
        if self.optimizer == "adamw":
            self.hip.adamw(model.train_flat.data, self.grad, self.m, self.v, lr, self.cfg.adam_beta1, self.cfg.adam_beta2,
                           self.cfg.adam_eps, self.cfg.weight_decay, self.opt_step, self.sumsq, self.cfg.max_grad_norm,
                           1.0 / self.world, True)
        else:
            self.hip.adamw8bit(model.train_flat.data, self.grad, self.c1, self.c2, self.am1, self.am2, self.map1, self.map2,
                               lr, self.cfg.adam_beta1, self.cfg.adam_beta2, self.cfg.adam_eps, self.cfg.weight_decay,
                               self.opt_step, self.sumsq, self.cfg.max_grad_norm, 1.0 / self.world, True)
        model.refresh_derived()  # W^T copies (full fine-tuning) or LoRA's scaled / transposed adapter copies
        mean_loss = (total / ga)
        if sync_loss:
            if self.world > 1:
                dist.all_reduce(mean_loss, op=dist.ReduceOp.SUM, group=self.group)
                mean_loss = mean_loss / self.world
            return StepStats(float(mean_loss), lr, self.opt_step)
        return StepStats(float("nan"), lr, self.opt_step)


def init_distributed(backend: Optional[str] = None):
    """Rendezvous from the launcher's env (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), as mm/forget.py:23-31 reads
    it from accelerate.  Returns (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world
