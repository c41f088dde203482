"""world_size-2 gloo tests (CPU) of the data-parallel pieces of the step: disjoint sample sharding and the
bucketed asynchronous gradient all-reduce that the engine launches from the backward sweep (engine.py)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from multimodal_unlearning_amd.engine import (GradBucketReducer, linear_schedule, shard_indices,
                                              steps_from_reference_formula)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, mode, max_inflight):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 10_000 + 8
        g = torch.Generator().manual_seed(100 + rank)
        flat = torch.randn(n, generator=g).to(torch.bfloat16)
        mine = flat.clone()
        red = GradBucketReducer(flat, bucket_elems=1200, mode=mode, max_inflight=max_inflight)
        assert red.enabled and len(red.buckets) >= 8
        # ranges arrive in backward order (tail first), exactly like the model's grad_ready_hook calls; rank 1 reports
        # the two lower ranges in the other order: the collectives must still be issued in one canonical order
        order = ((7000, n), (2500, 7000), (0, 2500)) if rank != 1 else ((7000, n), (0, 2500), (2500, 7000))
        for lo, hi in order:
            red.mark_final(lo, hi)
        red.reduce_rest()
        issued = list(red.issued)
        assert red.wait() == n
        others = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(others, mine)
        exp = sum(o.float() for o in others)
        err = 0.0
        for lo, hi in red.shard_ranges():  # allreduce: everything; rs_ag: this rank's shards (+ the replicated tail)
            err = max(err, (flat[lo:hi].float() - exp[lo:hi]).abs().max().item())
        if mode == "rs_ag":  # the parameter all-gather: every rank ends with every rank's shard values
            param = torch.full((n,), float(rank + 1)).to(torch.bfloat16)
            red.gather_params(param)
            for lo, hi in red.buckets:
                sz = (hi - lo) // world
                for r in range(world):
                    assert bool((param[lo + r * sz: lo + (r + 1) * sz] == r + 1).all())
        # sample sharding: disjoint and complete
        idx = shard_indices(23, rank, world)
        all_idx = [None] * world
        dist.all_gather_object(all_idx, idx)
        q.put((rank, err, sorted(sum(all_idx, [])) == list(range(23)), len(set(all_idx[0]) & set(all_idx[1])), issued))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,mode,max_inflight", [(2, "allreduce", None), (2, "rs_ag", None), (3, "allreduce", None),
                                                      (4, "allreduce", 0), (4, "rs_ag", 0)])
def test_bucketed_gradient_exchange_and_sharding(world, mode, max_inflight):
    """>= 8 buckets on 2-4 gloo ranks.  max_inflight=0 keeps EVERY bucket's work outstanding at once (the situation of
    round 1's 4-rank rehearsal hang, gpurun_out/rehearse4g.log): with one canonical issue order on every rank it
    completes; see DESIGN.md section 6."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q, mode, max_inflight)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for rank, err, complete, overlap, issued in res:
        assert err <= 2 ** -5 * (world / 2), err  # bf16 sum of `world` values vs fp32 sum
        assert complete and overlap == 0
        assert issued == res[0][4]  # one issue order on every rank (rank 1 reported its ranges in another order)
        body = [i for i in issued if i >= 0]
        assert body == sorted(body, reverse=True) and issued.index(-1) == 0 if -1 in issued else True  # (tail,) descending


def test_overlapping_final_ranges_are_rejected():
    """A range reported final twice in one step is the round-1 multi-sweep bug (a bucket reduced before its last
    writer ran): the reducer refuses it instead of silently double-reducing."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        red = GradBucketReducer(torch.zeros(4096), bucket_elems=1024, force=True)
        red.mark_final(2048, 4096)
        with pytest.raises(RuntimeError):
            red.mark_final(3000, 3500)
        red.reduce_rest()
        red.wait()
    finally:
        dist.destroy_process_group()


def test_reducer_is_noop_without_process_group():
    flat = torch.ones(64)
    red = GradBucketReducer(flat)
    assert not red.enabled
    red.mark_final(0, 64)
    red.reduce_rest()
    assert red.wait() == 0 and flat.sum().item() == 64


def test_schedule_and_step_formulae():
    # mm/forget.py:65-68,147-148 with the reference defaults (bs 2, GA 4) on forget10-sized data, 8 ranks
    total, warm = steps_from_reference_formula(n_samples=400, batch_size=2, ga=4, world=8, epochs=5)
    assert (total, warm) == (5 * 400 // 64, 400 // 64)
    assert linear_schedule(0, 1e-5, 4, 20) == 0.0
    assert linear_schedule(2, 1e-5, 4, 20) == pytest.approx(0.5e-5)
    assert linear_schedule(4, 1e-5, 4, 20) == pytest.approx(1e-5)
    assert linear_schedule(12, 1e-5, 4, 20) == pytest.approx(0.5e-5)
    assert linear_schedule(20, 1e-5, 4, 20) == 0.0


def test_lr_sequence_matches_hf_lambdalr():
    """The k-th optimiser update must run at lambda(k-1) like transformers' Trainer (scheduler.step() after
    optimizer.step()): checked against torch's LambdaLR driven by the HF lambda for a whole run."""
    from transformers.optimization import get_linear_schedule_with_warmup

    base, warm, total = 3e-4, 3, 11
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=base)
    sched = get_linear_schedule_with_warmup(opt, warm, total)
    hf = []
    for _ in range(total):
        hf.append(opt.param_groups[0]["lr"])  # lr used by this update
        p.grad = torch.ones(1)
        opt.step()
        sched.step()
    ours = [linear_schedule(k, base, warm, total) for k in range(total)]  # engine: evaluated at updates already done
    assert ours == pytest.approx(hf, rel=1e-12, abs=0)
    assert ours[0] == 0.0 and ours[-1] > 0.0
