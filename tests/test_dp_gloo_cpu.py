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


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 10_000 + 8
        g = torch.Generator().manual_seed(100 + rank)
        flat = torch.randn(n, generator=g).to(torch.bfloat16)
        mine = flat.clone()
        red = GradBucketReducer(flat, max_bucket_elems=3000)
        assert red.enabled
        # ranges arrive in backward order (tail first), exactly like the model's grad_ready_hook calls
        for lo, hi in ((7000, n), (2500, 7000), (0, 2500)):
            red.reduce_range(lo, hi)
        assert red.wait() == n
        others = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(others, mine)
        exp = sum(o.float() for o in others)
        err = (flat.float() - exp).abs().max().item()
        # sample sharding: disjoint and complete
        idx = shard_indices(23, rank, world)
        all_idx = [None] * world
        dist.all_gather_object(all_idx, idx)
        q.put((rank, err, sorted(sum(all_idx, [])) == list(range(23)), len(set(all_idx[0]) & set(all_idx[1]))))
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_and_sharding_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for rank, err, complete, overlap in res:
        assert err <= 2 ** -6, err  # bf16 sum of two values vs fp32 sum
        assert complete and overlap == 0


def test_reducer_is_noop_without_process_group():
    flat = torch.ones(64)
    red = GradBucketReducer(flat)
    assert not red.enabled
    red.reduce_range(0, 64)
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
