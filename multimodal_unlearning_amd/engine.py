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
    """base_lr * lambda(step) of transformers' get_linear_schedule_with_warmup.  HF's LambdaLR steps AFTER the
    optimiser, so the k-th update (k = 1, 2, ...) runs at lambda(k - 1): the first update of a warm-up has lr 0 and
    the last update of the run still has a positive lr.  ForgetEngine.step evaluates this at the number of updates
    already done (tests/test_dp_gloo_cpu.py pins the sequence against torch's LambdaLR)."""
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
    """Asynchronous gradient exchange over one flat gradient tensor.  Works on CPU (gloo) and GPU (RCCL).

    The buffer is cut ONCE into a fixed grid of buckets (`bucket_elems` each; boundaries are multiples of 8 * world so a
    bucket splits into equal per-rank shards).  The model's backward sweep reports element ranges [lo,hi) that are
    final (`mark_final`); a bucket's collective starts as soon as the bucket is fully covered.  Collectives are issued
    in ONE canonical order on every rank — strictly descending bucket index, the order the top-down sweep completes
    them — whatever the order of the hook calls, and the ranges reported within one step must be disjoint (an
    overlapping report means a bucket was handed over before its last writer ran: the multi-sweep bug of round 1).

    mode "allreduce": SUM all-reduce of every bucket; each rank then runs the whole optimiser step (replicas).
    mode "rs_ag"    : SUM reduce-scatter of every bucket into the rank's own shard (in place); the owner of the step
                      updates only its shards and `gather_params` all-gathers the updated parameter buckets — the
                      reduce-scatter / sharded-optimiser-state / all-gather scheme of the reference's ZeRO config
                      (config/ds_config.json:2-21).  Elements past the last full world*8 multiple go through a small
                      all-reduce and are updated on every rank.
    """

    def __init__(self, flat_grad: torch.Tensor, group=None, bucket_elems: int = 1 << 27, force: bool = False,
                 mode: str = "allreduce", max_inflight: Optional[int] = None):
        if mode not in ("allreduce", "rs_ag"):
            raise ValueError(f"unknown DP mode {mode}")
        self.flat, self.group, self.mode = flat_grad, group, mode
        # force=True runs the collectives even on a 1-rank group (exercises the RCCL path on a single GPU)
        self.enabled = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force)
        self.world = dist.get_world_size(group) if self.enabled else 1
        self.rank = dist.get_rank(group) if self.enabled else 0
        self._gloo = self.enabled and dist.get_backend(group) == "gloo"
        # gloo runs async works on a 2-thread pool per process; see DESIGN.md §6 for why at most `max_inflight` works
        # are kept outstanding there (RCCL is stream-ordered: any number may be in flight)
        self.max_inflight = max_inflight if max_inflight is not None else (2 if self._gloo else 0)
        n = flat_grad.numel()
        q = 8 * self.world
        step = max(q, bucket_elems // q * q)
        body = n // q * q  # elements that split evenly into per-rank shards of whole 16-byte groups
        self.buckets = [(lo, min(lo + step, body)) for lo in range(0, body, step)]
        self.tail = (body, n) if body < n else None
        self.works = []
        self.issued = []  # bucket indices in issue order (tests / debugging: must be identical on every rank)
        self._cov = []    # disjoint, sorted [lo,hi) intervals reported final in this step
        self._next = len(self.buckets) - 1
        self._tail_done = False
        self.reduced_elems = 0

    # -- coverage ---------------------------------------------------------------------------------------------
    def _covered(self, lo, hi):
        for a, b in self._cov:
            if a <= lo and hi <= b:
                return True
        return False

    def mark_final(self, lo: int, hi: int):
        """Hook target: elements [lo,hi) of the gradient are final for this step."""
        if not self.enabled or lo >= hi:
            return
        for a, b in self._cov:
            if lo < b and a < hi:
                raise RuntimeError(f"gradient range [{lo},{hi}) overlaps [{a},{b}) already reported final in this step")
        self._cov.append((lo, hi))
        self._cov.sort()
        merged = []
        for a, b in self._cov:
            if merged and merged[-1][1] == a:
                merged[-1] = (merged[-1][0], b)
            else:
                merged.append((a, b))
        self._cov = merged
        self._issue_ready()

    reduce_range = mark_final  # round-1 name

    def _issue_ready(self):
        if self.tail is not None and not self._tail_done and self._covered(*self.tail):
            self._issue(self.tail, tail=True)
        while self._next >= 0 and (self._tail_done or self.tail is None) and self._covered(*self.buckets[self._next]):
            self._issue(self.buckets[self._next], idx=self._next)
            self._next -= 1

    def _issue(self, rng, idx=-1, tail=False):
        lo, hi = rng
        if self.max_inflight and len(self.works) >= self.max_inflight:
            self.works.pop(0).wait()
        buf = self.flat[lo:hi]
        if tail or self.mode == "allreduce":
            w = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            sz = (hi - lo) // self.world
            out = buf[self.rank * sz:(self.rank + 1) * sz]  # in place: recv = send + rank * count (RCCL's in-place form)
            if self._gloo:  # gloo's reduce_scatter is not specified for aliased buffers: go through a copy
                tmp = torch.empty_like(out)
                w = dist.reduce_scatter_tensor(tmp, buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                w = _CopyAfter(w, tmp, out)
            else:
                w = dist.reduce_scatter_tensor(out, buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.works.append(w)
        self.issued.append(-1 if tail else idx)
        self.reduced_elems += hi - lo
        if tail:
            self._tail_done = True

    def reduce_rest(self):
        """Issue whatever no hook covered (a loss whose last sweep never ran — dead forwards keep the model's live-sweep
        count above zero — or a model without hooks), in the same canonical order."""
        if not self.enabled:
            return
        self._cov = [(0, self.flat.numel())]
        self._issue_ready()

    def wait(self):
        for w in self.works:
            w.wait()
        self.works = []
        self._cov = []
        self._next = len(self.buckets) - 1
        self._tail_done = False
        self.issued = []
        n = self.reduced_elems
        self.reduced_elems = 0
        return n

    # -- rs_ag: the shards this rank owns, and the parameter all-gather -------------------------------------------
    def shard_ranges(self):
        """[lo,hi) element ranges whose optimiser state this rank owns (rs_ag: one shard per bucket + the replicated
        tail; allreduce: everything)."""
        if not self.enabled or self.mode == "allreduce":
            return [(0, self.flat.numel())]
        out = []
        for lo, hi in self.buckets:
            sz = (hi - lo) // self.world
            out.append((lo + self.rank * sz, lo + (self.rank + 1) * sz))
        if self.tail is not None:
            out.append(self.tail)
        return out

    def gather_params(self, flat_param: torch.Tensor):
        """rs_ag: all-gather every parameter bucket from the ranks' updated shards (in place)."""
        if not self.enabled or self.mode == "allreduce":
            return
        works = []
        for lo, hi in reversed(self.buckets):
            sz = (hi - lo) // self.world
            buf = flat_param[lo:hi]
            mine = buf[self.rank * sz:(self.rank + 1) * sz]
            if self._gloo:
                mine = mine.clone()
                if self.max_inflight and len(works) >= self.max_inflight:
                    works.pop(0).wait()
            works.append(dist.all_gather_into_tensor(buf, mine, group=self.group, async_op=True))
        for w in works:
            w.wait()


class _CopyAfter:
    """wait() of an async work followed by a copy (gloo reduce-scatter through a temporary)."""

    def __init__(self, work, src, dst):
        self.work, self.src, self.dst = work, src, dst

    def wait(self):
        self.work.wait()
        self.dst.copy_(self.src)


@dataclass
class StepStats:
    loss: float
    lr: float
    step: int
    loss_tensor: Optional[torch.Tensor] = None  # sync_loss="defer": the (rank-averaged) mean loss, still on the device


class ForgetEngine:
    """engine.step(micro_batches) = GA x (compute_loss -> backward) -> gradient exchange -> clip -> AdamW -> zero grads.

    dp_mode "allreduce" (default): bucketed SUM all-reduce overlapped with the last backward sweep, every rank runs the
    full optimiser step.  dp_mode "rs_ag": reduce-scatter the buckets, each rank keeps optimiser state for and updates
    only its 1/world shard of every bucket, then the updated parameter buckets are all-gathered (ZeRO-style; state
    memory and AdamW traffic / world, same xGMI bytes as the all-reduce; the all-gather is not overlapped)."""

    def __init__(self, model, trainer, cfg: ForgetConfig, total_steps: int = 1000, warmup_steps: int = 0,
                 group=None, transposed_weights: bool = True, force_collectives: bool = False,
                 optimizer: str = "adamw", dp_mode: str = "allreduce", bucket_elems: int = 1 << 27,
                 merge_ga: bool = True, merge_tokens="auto", overlap_optimizer: bool = False):
        from . import hip  # HIP optimiser kernels; fails loudly when the library is missing

        self.hip = hip
        if getattr(model, "autograd_grads", False):
            raise ValueError("ForgetEngine owns the flat gradient buffer itself: the model is in autograd-gradient mode "
                             "(LlavaHIP.enable_autograd_gradients, for torch DDP under transformers.Trainer)")
        self.model, self.trainer, self.cfg = model, trainer, cfg
        self.total_steps, self.warmup_steps = total_steps, warmup_steps
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        dev = model.device
        self.grad = model.ensure_grad()
        self.reducer = GradBucketReducer(self.grad, group, bucket_elems=bucket_elems, force=force_collectives, mode=dp_mode)
        self.dp_mode = dp_mode
        # element ranges of the flat buffers this rank updates, and where each one's optimiser state starts
        self.shards = self.reducer.shard_ranges()
        self.shard_off = []
        n = 0
        for lo, hi in self.shards:
            self.shard_off.append(n)
            n += hi - lo
        # "adamw": fp32 moments (8 B/param).  "adamw8bit": block-wise 8-bit moments as in the reference's
        # optim="adamw_bnb_8bit" (mm/forget.py:156): 2 B/param + one fp32 absmax per 2048 parameters and moment.
        if optimizer not in ("adamw", "adamw8bit"):
            raise ValueError(f"unknown optimizer {optimizer}")
        # does this rank own (and update) the whole flat buffer?  NOT `len(shards) == 1`: with rs_ag and a model smaller
        # than one bucket every rank owns exactly one shard — a fraction of the buffer (round 3: replicas diverged)
        self.whole = self.shards == [(0, self.grad.numel())]
        if optimizer == "adamw8bit" and not self.whole:
            raise ValueError("adamw8bit keeps block-wise state over the whole flat buffer: use dp_mode='allreduce'")
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
        # transformers.Trainer.create_optimizer decays neither biases nor norm weights: per shard, the element ranges
        # (relative to the shard) that the AdamW kernel must not decay
        from .params import no_weight_decay
        nd_abs = sorted(model.range_of(n) for n in model.trainable_names() if no_weight_decay(n))
        self.nodecay = []
        for lo, hi in self.shards:
            rel = [(max(a, lo) - lo, min(b, hi) - lo) for a, b in nd_abs if a < hi and lo < b]
            self.nodecay.append(torch.tensor(rel, device=dev, dtype=torch.int64).reshape(-1, 2) if rel else None)
        self.sumsq = torch.zeros(hip.SUMSQ_FLOATS, device=dev, dtype=torch.float32)  # [0] = |g|^2, rest scratch
        self.opt_step = 0
        # gradient-accumulation micro-batches are evaluated as one concatenated batch where shapes, loss and HBM allow
        # (`_groups`); merge_tokens: "auto" = from free HBM, or an explicit cap on positions per forward
        self.merge_ga, self.merge_tokens, self.last_groups = merge_ga, merge_tokens, []
        # overlap_optimizer: AdamW and the W^T refresh of a step run on a side stream, one chunk of the flat buffer per
        # decoder layer in forward order, while the NEXT step's forward already runs on the main stream; the model waits for
        # a chunk's event right before it first reads that chunk (LlavaHIP._wait_params).  Same kernels, same arithmetic, same
        # order per element: results are bit-identical to the serial step.  What the caller must know: after step() returns,
        # the parameters are only guaranteed complete for the model's own forwards — anything else that reads them on the
        # main stream (a checkpoint, a test) calls finish_optimizer() first.  Full fine-tuning with fp32 moments and the
        # all-reduce mode only (everything else keeps the serial step).
        self.overlap_optimizer = bool(overlap_optimizer) and self.grad.is_cuda and optimizer == "adamw" and self.whole \
            and not getattr(model, "lora", None) and hasattr(model, "param_chunks")
        self._side = torch.cuda.Stream(device=dev) if self.overlap_optimizer else None
        self._chunks = None
        # exposed gradient-exchange time: the stretch of the compute stream between the end of the last backward sweep
        # and the completion of the last collective (device events; read by bench.py, never synchronised here)
        self.time_exchange = False
        self.exchange_events = []
        # measurement hook (bench.py --squat-*): called once per step at the moment the gradient exchange would start,
        # i.e. when the backward sweep of the last micro-batch reports its first final range
        self.exchange_start_callback: Optional[Callable[[], None]] = None
        if transposed_weights:
            model.enable_transposed_weights()

    def _with_start_callback(self, hook):
        fired = []

        def wrapped(lo, hi):
            if not fired:
                fired.append(1)
                self.exchange_start_callback()
            if hook is not None:
                hook(lo, hi)
        return wrapped

    def _grad_sumsq(self):
        """|g|^2 over the whole (reduced) gradient into self.sumsq[0], identical on every rank."""
        hip = self.hip
        if self.whole:
            hip.sumsq(self.grad, self.sumsq)
            return
        acc = torch.zeros(1, device=self.grad.device, dtype=torch.float32)
        tail = self.reducer.tail
        for lo, hi in self.shards:
            if (lo, hi) == tail and self.reducer.rank != 0:
                continue  # the replicated tail counts once
            hip.sumsq(self.grad[lo:hi], self.sumsq)
            acc += self.sumsq[0]
        dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=self.group)  # same bits on every rank
        self.sumsq[0:1].copy_(acc)

    # ---------------------------------------------------------------------------------- merged micro-batches
    def _tokens_of(self, mb) -> int:
        """Sequence positions one micro-batch sends through the student (every split, every row)."""
        splits = mb.values() if isinstance(mb, dict) else mb
        n = 0
        for sp in splits:
            ids = sp["input_ids"] if isinstance(sp, dict) else sp[0]
            n += ids.numel()
        return n

    def _token_budget(self) -> int:
        """How many sequence positions ONE forward/backward may hold, from the HBM that is free right now (the caching
        allocator's idle blocks count as free) and the model's estimate of what a position keeps alive."""
        if self.merge_tokens != "auto":
            return int(self.merge_tokens)
        per_tok = getattr(self.model, "activation_bytes_per_token", None)
        if per_tok is None or not self.grad.is_cuda:
            return 0
        dev = self.grad.device
        free, _ = torch.cuda.mem_get_info(dev)
        free += torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
        return int(0.8 * free / per_tok())

    def _groups(self, micro_batches):
        """[(inputs, g)]: runs of consecutive micro-batches that are evaluated as ONE concatenated batch (g of them), so
        that the reference's small per-device batch (2 pairs x GA 4, config/mm/forget.yaml:14-15) reaches the GEMMs as
        M = g * B * S rows instead of g launches of B * S.  Rows never interact and every loss term is a mean over
        equal-sized sets, a per-segment token mean or a sum (MMTrainerForgetting.merge_micro_batches), so loss and
        gradients are those of the loop up to summation order.  A run ends where the HBM budget or the shapes say so; a
        run of one is the plain loop."""
        merge = getattr(self.trainer, "merge_micro_batches", None)
        if not self.merge_ga or merge is None or len(micro_batches) < 2:
            return [(mb, 1) for mb in micro_batches]
        budget = self._token_budget()
        out, i, n = [], 0, len(micro_batches)
        while i < n:
            j, tok = i + 1, self._tokens_of(micro_batches[i])
            while j < n and tok + self._tokens_of(micro_batches[j]) <= budget:
                tok += self._tokens_of(micro_batches[j])
                j += 1
            merged = merge(self.model, micro_batches[i:j]) if j - i > 1 else None
            if merged is None:
                out += [(mb, 1) for mb in micro_batches[i:j]]
            else:
                out.append((merged, j - i))
            i = j
        return out

    def _pipelined_adamw(self, lr):
        """AdamW + W^T refresh chunk by chunk on the side stream (see overlap_optimizer in __init__)."""
        model, c = self.model, self.cfg
        p = model.train_flat.data
        if self._chunks is None:
            from .params import no_weight_decay
            nd_abs = sorted(model.range_of(n) for n in model.trainable_names() if no_weight_decay(n))
            self._chunks = []
            for key, lo, hi in model.param_chunks():
                rel = [(max(a, lo) - lo, min(b, hi) - lo) for a, b in nd_abs if a < hi and lo < b]
                nd = torch.tensor(rel, device=p.device, dtype=torch.int64).reshape(-1, 2) if rel else None
                prefix = f"text.layers.{key}." if isinstance(key, int) else None
                self._chunks.append((key, lo, hi, nd, prefix))
        main = torch.cuda.current_stream()
        self._side.wait_stream(main)          # the gradients are final and |g|^2 is in self.sumsq
        events = {}
        with torch.cuda.stream(self._side):
            for key, lo, hi, nd, prefix in self._chunks:
                self.hip.adamw(p[lo:hi], self.grad[lo:hi], self.m[lo:hi], self.v[lo:hi], lr, c.adam_beta1, c.adam_beta2,
                               c.adam_eps, c.weight_decay, self.opt_step, self.sumsq, c.max_grad_norm, 1.0 / self.world, True,
                               nodecay=nd)
                if model.wT:
                    if prefix is not None:
                        model.refresh_transposed_weights(prefix)
                    elif key == "post":
                        model.refresh_transposed_weights("lm_head.")
                    else:
                        for pre in ("projector.", "vision."):
                            model.refresh_transposed_weights(pre)
                ev = torch.cuda.Event()
                ev.record(self._side)
                events[key] = ev
        model.param_events = events
        model._derived_at = model.train_flat._version

    def finish_optimizer(self):
        """Make the main stream wait for the (possibly still running) parameter update of the last step: call before anything
        but the model's own forward reads the parameters (checkpointing, comparisons).  No-op for the serial step."""
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)

    def step(self, micro_batches: List[Dict], sync_loss: bool = False) -> StepStats:
        ga = len(micro_batches)
        model = self.model
        total = None
        groups = self._groups(micro_batches)
        self.last_groups = [g for _, g in groups]
        for i, (mb, g) in enumerate(groups):
            last = i == len(groups) - 1
            hook = self.reducer.mark_final if (last and self.reducer.enabled) else None
            if last and self.exchange_start_callback is not None:
                hook = self._with_start_callback(hook)
            model.grad_ready_hook = hook
            model.live_sweeps = 0
            if g > 1:  # mean of the g micro-batch losses, from one forward/backward over their concatenation
                loss = self.trainer.compute_loss(model, mb, _merge=g)
            else:
                loss = self.trainer.compute_loss(model, mb)
            (loss / ga if g == 1 else loss * (g / ga)).backward()
            d = loss.detach().float() * g
            total = d if total is None else total + d
        model.grad_ready_hook = None
        ev = None
        if self.time_exchange and self.reducer.enabled and self.grad.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        self.reducer.reduce_rest()
        self.reducer.wait()
        if ev is not None:
            ev[1].record()
            self.exchange_events.append(ev)
        lr = linear_schedule(self.opt_step, self.cfg.lr, self.warmup_steps, self.total_steps)  # lambda(k-1), as HF
        self.opt_step += 1  # 1-based from here on: Adam bias correction of the k-th update
        self._grad_sumsq()
        c, p = self.cfg, model.train_flat.data
        whole = self.whole
        if self.overlap_optimizer:
            self._pipelined_adamw(lr)
        elif self.optimizer == "adamw":
            for (lo, hi), off, nd in zip(self.shards, self.shard_off, self.nodecay):
                self.hip.adamw(p[lo:hi], self.grad[lo:hi], self.m[off:off + hi - lo], self.v[off:off + hi - lo], lr,
                               c.adam_beta1, c.adam_beta2, c.adam_eps, c.weight_decay, self.opt_step, self.sumsq,
                               c.max_grad_norm, 1.0 / self.world, whole, nodecay=nd)
        else:
            self.hip.adamw8bit(p, self.grad, self.c1, self.c2, self.am1, self.am2, self.map1, self.map2,
                               lr, c.adam_beta1, c.adam_beta2, c.adam_eps, c.weight_decay,
                               self.opt_step, self.sumsq, c.max_grad_norm, 1.0 / self.world, True, nodecay=self.nodecay[0])
        if not whole:
            self.grad.zero_()  # the other ranks' shards still hold this rank's unreduced partial sums
            self.reducer.gather_params(p)
        if not self.overlap_optimizer:
            model.refresh_derived()  # W^T copies (full fine-tuning) or LoRA's scaled / transposed adapter copies
        mean_loss = (total / ga)
        if sync_loss:
            if self.world > 1:
                dist.all_reduce(mean_loss, op=dist.ReduceOp.SUM, group=self.group)
                mean_loss = mean_loss / self.world
            if sync_loss == "defer":  # the caller reads it later (e.g. after collating the next batches on the host)
                return StepStats(float("nan"), lr, self.opt_step, loss_tensor=mean_loss)
            return StepStats(float(mean_loss), lr, self.opt_step)
        return StepStats(float("nan"), lr, self.opt_step)


class FlatAdamW(torch.optim.Optimizer):
    """torch.optim face of the fused HIP AdamW over LlavaHIP's flat parameter, for loops that own the optimiser
    themselves — transformers.Trainer when it hosts the plug-in (`hosted_by_hf_trainer` builds this in
    `create_optimizer`).  transformers.Trainer.create_optimizer puts biases and every norm weight into a weight_decay = 0
    group; with ONE flat parameter that grouping cannot be expressed as param groups, so the kernel's no-decay range
    table carries it (same table as ForgetEngine).  `optimizer="adamw8bit"`: block-wise 8-bit moments, the reference's
    `optim="adamw_bnb_8bit"` (mm/forget.py:156).  The learning rate is read from the param group on every step, so HF's
    LambdaLR scheduler drives it; clipping is left to the host loop (it clips before `step`); after the update the
    model's derived copies (W^T / LoRA) are refreshed."""

    def __init__(self, model, lr=1e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, optimizer="adamw"):
        from . import hip
        from .params import no_weight_decay

        if optimizer not in ("adamw", "adamw8bit"):
            raise ValueError(f"unknown optimizer {optimizer}")
        super().__init__([model.train_flat], dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.hip, self.model, self.kind, self.updates = hip, model, optimizer, 0
        p = model.train_flat
        n, dev = p.numel(), p.device
        if optimizer == "adamw":
            self.m = torch.zeros(n, device=dev, dtype=torch.float32)
            self.v = torch.zeros(n, device=dev, dtype=torch.float32)
        else:
            self.map1, self.map2 = hip.dynamic_map(True).to(dev), hip.dynamic_map(False).to(dev)
            self.c1 = torch.full((n,), int((self.map1 == 0).nonzero()[0]), device=dev, dtype=torch.uint8)
            self.c2 = torch.full((n,), int((self.map2 == 0).nonzero()[0]), device=dev, dtype=torch.uint8)
            nb = hip.adamw8bit_nblocks(n)
            self.am1 = torch.zeros(nb, device=dev, dtype=torch.float32)
            self.am2 = torch.zeros(nb, device=dev, dtype=torch.float32)
        nd = sorted(model.range_of(k) for k in model.trainable_names() if no_weight_decay(k))
        self.nodecay = torch.tensor(nd, device=dev, dtype=torch.int64).reshape(-1, 2) if nd else None

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        g = self.param_groups[0]
        p = self.model.train_flat
        if p.grad is None:
            return loss
        self.updates += 1
        (b1, b2), grad = g["betas"], p.grad
        if self.kind == "adamw":
            self.hip.adamw(p.data, grad, self.m, self.v, float(g["lr"]), b1, b2, g["eps"], g["weight_decay"], self.updates,
                           None, 0.0, 1.0, False, nodecay=self.nodecay)
        else:
            self.hip.adamw8bit(p.data, grad, self.c1, self.c2, self.am1, self.am2, self.map1, self.map2, float(g["lr"]), b1,
                               b2, g["eps"], g["weight_decay"], self.updates, None, 0.0, 1.0, False, nodecay=self.nodecay)
        self.model.refresh_derived()
        return loss

    def _moment_tensors(self):
        names = ("m", "v") if self.kind == "adamw" else ("c1", "c2", "am1", "am2")
        return {k: getattr(self, k) for k in names}

    def state_dict(self):
        """torch's per-parameter `state` is empty here (the moments live in flat tensors of their own), and its loader
        would cast fp32 moments to the parameter's bf16: both are carried in one extra entry instead, so a host loop that
        checkpoints its optimiser (transformers.Trainer without save_only_model) resumes with the moments it saved."""
        sd = super().state_dict()
        sd["mmu_flat"] = {"kind": self.kind, "updates": self.updates, **self._moment_tensors()}
        return sd

    def load_state_dict(self, state_dict):
        sd = dict(state_dict)
        flat = sd.pop("mmu_flat", None)
        super().load_state_dict(sd)
        if flat is None:
            raise ValueError("optimizer state without the flat AdamW moments (not written by FlatAdamW.state_dict)")
        if flat["kind"] != self.kind:
            raise ValueError(f"optimizer state holds {flat['kind']} moments, this optimiser is {self.kind}")
        for k, t in self._moment_tensors().items():
            if flat[k].shape != t.shape or flat[k].dtype != t.dtype:
                raise ValueError(f"optimizer state entry {k}: {tuple(flat[k].shape)} {flat[k].dtype}, expected {tuple(t.shape)} {t.dtype}")
            t.copy_(flat[k])
        self.updates = int(flat["updates"])


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
