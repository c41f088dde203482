"""Two ranks on ONE MI355X (gloo backend over device tensors): the engine's data-parallel step — backward-hook bucket
launches, stream ordering between the HIP kernels and the collective, 1/world folded into AdamW — must give the same
weights as one process that accumulates the two ranks' micro-batches.  (RCCL needs one GPU per rank; the 8-GPU run is
the driver's.  gloo stages through the host, which is enough to exercise the ordering and the arithmetic.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(dev, lora=False, loss="grad_diff_forget_ce_retain_ce", loss_args=None, dp_mode="allreduce"):
    from multimodal_unlearning_amd.config import ForgetConfig, tiny_test_config
    from multimodal_unlearning_amd.engine import ForgetEngine
    from multimodal_unlearning_amd.llava import LlavaHIP
    from multimodal_unlearning_amd.params import init_params
    from multimodal_unlearning_amd.trainer import MMTrainerForgetting

    torch.manual_seed(0)  # rmu draws its control vector with torch.rand: the same on every rank, as under set_seed
    cfg = tiny_test_config()
    if lora:
        from multimodal_unlearning_amd.llava import LoraSpec

        model = LlavaHIP(cfg, init_params(cfg, seed=0), device=dev, lora=LoraSpec(r=8, alpha=32.0))
        model.init_lora_(seed=3, b_std=0.02)
    else:
        model = LlavaHIP(cfg, init_params(cfg, seed=0), device=dev)
    teacher = None
    if loss in ("kl", "dpo", "scrub", "llmu", "rmu", "sku"):
        from multimodal_unlearning_amd.synthetic import teacher_params_from

        teacher = LlavaHIP(cfg, teacher_params_from(init_params(cfg, seed=0), seed=7), device=dev, trainable=False)
    tr = MMTrainerForgetting(model=model, forget_loss=loss, teacher_model=teacher, loss_beta=1.0,
                             l1_lambda=0.0, l0_lambda=0.0, l_norm_from="zero", loss_args=loss_args or {})
    # small buckets: several collectives per step on the tiny model (the flat gradient has ~0.5 M elements)
    eng = ForgetEngine(model, tr, ForgetConfig(lr=1e-3), total_steps=10, warmup_steps=1, dp_mode=dp_mode,
                       bucket_elems=1 << 16)
    return cfg, model, eng


SPLITS = ("forget", "retain", "idk")


def _batch(cfg, seed, loss):
    from multimodal_unlearning_amd.synthetic import make_forget_batch, make_split

    mb = make_forget_batch(cfg, 2, 40, seed=seed, n_question=6, splits=SPLITS)
    if loss == "sku":  # nested [B, K, ...] "random" split (mm/dataset.py:265-276)
        ks = [make_split(cfg, 2, 40, seed=seed * 7 + k, n_question=6) for k in range(3)]
        mb["random"] = {k: torch.stack([x[k] for x in ks], dim=1) for k in ks[0]}
    return mb


def _worker(rank, world, port, q, lora, loss, loss_args, dp_mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg, model, eng = _build("cuda:0", lora, loss, loss_args, dp_mode)
        assert eng.reducer.enabled and eng.world == world and len(eng.reducer.buckets) > 2
        for step in range(2):
            eng.step([_batch(cfg, 50 + 10 * step + rank, loss)])
        torch.cuda.synchronize()
        q.put((rank, model.train_flat.detach().float().cpu().numpy()))  # numpy: no fd passing after exit
    finally:
        dist.destroy_process_group()


SKU = {"bad_weight": 2.5, "random_weight": 1.0, "normal_weight": 0.5}


@pytest.mark.parametrize("world,lora,loss,loss_args,dp_mode", [
    (2, False, "grad_diff_forget_ce_retain_ce", None, "allreduce"),                  # paired fusion: one backward sweep
    (2, True, "grad_diff_forget_ce_retain_ce", None, "allreduce"),                   # adapters incl. the CLIP tower's
    (2, False, "grad_diff_forget_ce_retain_ce", {"fuse_pairs": False}, "allreduce"),  # two sweeps in one loss.backward()
    (2, False, "kl", None, "allreduce"),                                             # two student sweeps + a frozen teacher
    (2, False, "dpo", None, "allreduce"),                                            # a forward that never sweeps (reference quirk)
    (2, False, "scrub", None, "allreduce"),                                          # 2 student + 2 teacher forwards
    (2, False, "llmu", None, "allreduce"),                                           # 3 student sweeps
    (2, False, "sku", SKU, "allreduce"),                                             # 1 + K + 1 student sweeps
    (2, False, "rmu", {"rmu_layer": 0}, "allreduce"),                                # sweeps that stop at the hooked layer
    (3, False, "grad_diff_forget_ce_retain_ce", {"fuse_pairs": False}, "allreduce"),  # 3 ranks, > 2 buckets in flight
    (2, False, "grad_diff_forget_ce_retain_ce", None, "rs_ag"),                      # reduce-scatter + sharded AdamW + all-gather
    (3, False, "kl", None, "rs_ag"),
])
def test_ranks_match_gradient_accumulation(world, lora, loss, loss_args, dp_mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q, lora, loss, loss_args, dp_mode)) for r in range(world)]
    for p in ps:
        p.start()
    res = {r: torch.from_numpy(a) for r, a in (q.get(timeout=300) for _ in ps)}
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    for r in range(1, world):
        assert torch.equal(res[0], res[r])  # replicas stay identical
    cfg, model, eng = _build("cuda:0", lora, loss, loss_args)
    for step in range(2):
        eng.step([_batch(cfg, 50 + 10 * step + r, loss) for r in range(world)])
    torch.cuda.synchronize()
    ref = model.train_flat.detach().float().cpu()
    err = (res[0] - ref).norm() / ref.norm()
    moved = (ref - _build("cuda:0", lora, loss, loss_args)[1].train_flat.detach().float().cpu()).norm() / ref.norm()
    # bf16 gradients: rank-sum then 1/world vs accumulate-of-fractions differ in rounding only
    assert moved > 1e-3 and err < 0.05 * moved, (float(err), float(moved))


def _entry_worker(rank, world, port, q, out_dir, dp_mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)  # run() keeps an existing group (RCCL wants a GPU per rank)
    from multimodal_unlearning_amd.forget import ForgetRunConfig, parse_overrides, run

    rc = parse_overrides(["model_family=tiny", f"save_dir={out_dir}", "forget_data_path=synthetic:16", "batch_size=2",
                          "gradient_accumulation_steps=2", "num_epochs=1", "seq_len=40", "lr=1e-3", "forget_loss=KL",
                          f"dp_mode={dp_mode}"], ForgetRunConfig())
    eng = run(rc)  # barrier + destroy_process_group inside
    q.put((rank, eng.opt_step, eng.model.train_flat.detach().float().cpu().numpy()))


@pytest.mark.parametrize("dp_mode", ["allreduce", "rs_ag"])
def test_forget_entry_point_on_two_ranks(tmp_path, dp_mode):
    """The launcher's per-rank body (mm/forget.py:23-31,139-195 under `accelerate launch`) on two ranks sharing one GPU:
    disjoint dataset shards, max_steps divided by the world size (:66-67), the loss all-reduced for logging, replicas
    identical after training, rank 0 alone writes config.yaml and the checkpoint, every rank leaves through the barrier."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    out = str(tmp_path / "run")
    ps = [ctx.Process(target=_entry_worker, args=(r, 2, port, q, out, dp_mode)) for r in range(2)]
    for p in ps:
        p.start()
    res = {r: (n, torch.from_numpy(a)) for r, n, a in (q.get(timeout=300) for _ in ps)}
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    assert res[0][0] == res[1][0] == 16 // (2 * 2 * 2)      # max_steps = epochs * len // (batch * GA * world)
    assert torch.equal(res[0][1], res[1][1])
    assert os.path.exists(os.path.join(out, "config.yaml")) and os.path.exists(os.path.join(out, "config.json"))


def _merge_worker(rank, world, port, q, model_path, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multimodal_unlearning_amd.forget import ForgetRunConfig, parse_overrides, run

    if rank == 1:  # the slow rank of ADVICE r3: rank 0 must not rewrite model_path while this one is still reading it
        import time
        time.sleep(3.0)
    rc = parse_overrides(["model_family=tiny", f"save_dir={out_dir}", f"model_path={model_path}", "forget_data_path=synthetic:8",
                          "batch_size=2", "gradient_accumulation_steps=1", "num_epochs=1", "seq_len=40", "lr=0.0",
                          "forget_loss=retain_ft", "max_steps=1", "save_model=false"], ForgetRunConfig())
    eng = run(rc)
    q.put((rank, eng.model.train_flat.detach().float().cpu().numpy()))


def test_forget_entry_point_merges_a_lora_run_on_two_ranks(tmp_path):
    """mm/forget.py:100-113 at world 2 with model_path = a LoRA run's save_dir: every rank must start from the MERGED weights
    (rank 0 decides what the directory holds, nobody rewrites it while another rank reads, all ranks take the same
    barriers), the directory ends as a plain checkpoint with the adapter retired."""
    from multimodal_unlearning_amd.forget import ForgetRunConfig, parse_overrides, run

    a = str(tmp_path / "a")
    base = ["model_family=tiny", "forget_data_path=synthetic:8", "batch_size=2", "gradient_accumulation_steps=1", "seq_len=40"]
    e1 = run(parse_overrides(base + [f"save_dir={a}", "forget_loss=grad_ascent", "LoRA.r=8", "num_epochs=2", "lr=5e-3"], ForgetRunConfig()))
    e1.model.merge_lora_()
    merged = e1.model._base_flat.detach().float().cpu()
    del e1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_merge_worker, args=(r, 2, port, q, a, str(tmp_path / "b"))) for r in range(2)]
    for p in ps:
        p.start()
    res = {r: torch.from_numpy(x) for r, x in (q.get(timeout=300) for _ in ps)}
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    assert torch.equal(res[0], res[1])
    assert (res[0] - merged).norm() / merged.norm() < 1e-6
    assert not os.path.exists(os.path.join(a, "adapter_config.json")) and os.path.exists(os.path.join(a, "adapter_config.json.merged"))
    assert not os.path.exists(os.path.join(a, ".merge_tmp"))


def _hosted_worker(rank, world, port, q, loss, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      LOCAL_WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)  # transformers / accelerate adopt an existing group
    try:
        import transformers

        from multimodal_unlearning_amd.config import tiny_test_config
        from multimodal_unlearning_amd.llava import LlavaHIP
        from multimodal_unlearning_amd.params import init_params
        from multimodal_unlearning_amd.synthetic import teacher_params_from
        from multimodal_unlearning_amd.trainer import MMTrainerForgetting, hosted_by_hf_trainer

        cfg = tiny_test_config()
        model = LlavaHIP(cfg, init_params(cfg, seed=0), device="cuda:0")
        teacher = LlavaHIP(cfg, teacher_params_from(init_params(cfg, seed=0), seed=7), device="cuda:0", trainable=False)
        data = [_batch(cfg, 60 + i, loss) for i in range(4)]      # every rank holds the list; accelerate deals the batches out
        args = transformers.TrainingArguments(
            output_dir=os.path.join(out_dir, f"r{rank}"), per_device_train_batch_size=1, gradient_accumulation_steps=1, max_steps=2,
            learning_rate=1e-3, weight_decay=0.01, max_grad_norm=1e9, bf16=True, report_to=[], remove_unused_columns=False,
            disable_tqdm=True, save_strategy="no", optim="adamw_torch", lr_scheduler_type="constant", warmup_steps=0,
            ddp_backend="gloo", ddp_find_unused_parameters=True, dataloader_pin_memory=False)   # mm/forget.py:161

        class InOrder(hosted_by_hf_trainer(MMTrainerForgetting)):
            def _get_train_sampler(self, *a, **k):
                return torch.utils.data.SequentialSampler(self.train_dataset)

        tr = InOrder(model=model, args=args, train_dataset=data, data_collator=lambda x: x[0], forget_loss=loss,
                     teacher_model=teacher, loss_beta=1.0, l1_lambda=0, l0_lambda=0, l_norm_from="zero", loss_args={})
        assert model.autograd_grads and model.ddp_compatible
        tr.train()
        wrapped = type(tr.model_wrapped).__name__
        torch.cuda.synchronize()
        q.put((rank, wrapped, model.train_flat.detach().float().cpu().numpy()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("loss", ["grad_diff_forget_ce_retain_ce", "kl"])
def test_hosted_by_hf_trainer_under_torch_ddp_on_two_ranks(tmp_path, loss):
    """The reference's arrangement at world 2: `class MMTrainerForgetting(Trainer)` (mm/trainer.py:112)
    under transformers.Trainer, which wraps the model in torch DDP with find_unused_parameters=True (mm/forget.py:161).
    LlavaHIP hands its flat gradient to autograd (`enable_autograd_gradients`), DDP's reducer averages it, HF's optimiser
    (the fused flat AdamW) steps: replicas identical, and equal to ForgetEngine accumulating the same micro-batches."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_hosted_worker, args=(r, 2, port, q, loss, str(tmp_path))) for r in range(2)]
    for p in ps:
        p.start()
    got = [q.get(timeout=420) for _ in ps]
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    res = {r: torch.from_numpy(a) for r, _, a in got}
    assert all(w == "DistributedDataParallel" for _, w, _ in got), [w for _, w, _ in got]
    assert torch.equal(res[0], res[1])
    from multimodal_unlearning_amd.config import ForgetConfig, tiny_test_config
    from multimodal_unlearning_amd.engine import ForgetEngine
    from multimodal_unlearning_amd.llava import LlavaHIP
    from multimodal_unlearning_amd.params import init_params
    from multimodal_unlearning_amd.synthetic import teacher_params_from
    from multimodal_unlearning_amd.trainer import MMTrainerForgetting

    cfg = tiny_test_config()
    start = LlavaHIP(cfg, init_params(cfg, seed=0), device="cuda:0").train_flat.detach().float().cpu()
    ref = LlavaHIP(cfg, init_params(cfg, seed=0), device="cuda:0")
    teacher = LlavaHIP(cfg, teacher_params_from(init_params(cfg, seed=0), seed=7), device="cuda:0", trainable=False)
    tr = MMTrainerForgetting(model=ref, forget_loss=loss, teacher_model=teacher, loss_beta=1.0, l1_lambda=0, l0_lambda=0,
                             l_norm_from="zero", loss_args={})
    eng = ForgetEngine(ref, tr, ForgetConfig(lr=1e-3, weight_decay=0.01, max_grad_norm=1e9), total_steps=10 ** 9, warmup_steps=0,
                       transposed_weights=False, merge_ga=False)
    data = [_batch(cfg, 60 + i, loss) for i in range(4)]
    eng.step(data[0:2])   # rank 0 saw batch 0, rank 1 batch 1 (accelerate's BatchSamplerShard), DDP averaged
    eng.step(data[2:4])
    torch.cuda.synchronize()
    want = ref.train_flat.detach().float().cpu()
    moved = (want - start).norm()
    err = (res[0] - want).norm()
    assert moved > 1e-3 * start.norm() and err < 0.05 * moved, (float(err), float(moved))


def _merged_worker(rank, world, port, q, loss):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg, model, eng = _build("cuda:0", False, loss, None, "allreduce")
        groups = []
        for step in range(2):
            eng.step([_batch(cfg, 70 + 10 * step + 2 * rank + i, loss) for i in range(2)])   # GA 2 per rank
            groups.append(list(eng.last_groups))
        torch.cuda.synchronize()
        q.put((rank, groups, model.train_flat.detach().float().cpu().numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("loss", ["grad_diff_forget_ce_retain_ce", "kl"])
def test_ranks_with_merged_micro_batches_match_gradient_accumulation(loss):
    """Round 4: each rank evaluates its two gradient-accumulation micro-batches as ONE merged forward/backward
    (ForgetEngine._groups) — the bucket hooks then fire from that single sweep —; replicas identical, and equal to one
    process that walks all four micro-batches one by one."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_merged_worker, args=(r, 2, port, q, loss)) for r in range(2)]
    for p in ps:
        p.start()
    got = [q.get(timeout=300) for _ in ps]
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    res = {r: torch.from_numpy(a) for r, _, a in got}
    assert all(g == [[2], [2]] for _, g, _ in got), [g for _, g, _ in got]
    assert torch.equal(res[0], res[1])
    cfg, model, eng = _build("cuda:0", False, loss, None)
    eng.merge_ga = False
    for step in range(2):
        eng.step([_batch(cfg, 70 + 10 * step + 2 * r + i, loss) for r in range(2) for i in range(2)])
    torch.cuda.synchronize()
    ref = model.train_flat.detach().float().cpu()
    moved = (ref - _build("cuda:0", False, loss, None)[1].train_flat.detach().float().cpu()).norm() / ref.norm()
    err = (res[0] - ref).norm() / ref.norm()
    assert moved > 1e-3 and err < 0.06 * moved, (float(err), float(moved))
