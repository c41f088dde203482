"""GPU parity of the HIP LLaVA + compute_loss path against (a) the committed golden vectors produced by the
reference's own compute_loss / transformers (fp32) and (b) the oracle run live on the same inputs.

Tolerances (bf16 activations and weights vs an fp32 reference of the same bf16-rounded weights):
  logits   relative L2 error < 1e-2  (measured ~3e-3)
  loss     |d| < 1e-3 * max(1, |loss|)  -- BASELINE.json north_star "within 1e-3 bf16"; the entropy / dpo
           branches are SUMS over thousands of terms and use the same RELATIVE bound
  grads    total-norm relative error < 2e-2
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from multimodal_unlearning_amd.config import ForgetConfig, tiny_test_config  # noqa: E402
from multimodal_unlearning_amd.params import init_params  # noqa: E402
from multimodal_unlearning_amd.synthetic import make_forget_batch, teacher_params_from  # noqa: E402

G = os.path.join(os.path.dirname(__file__), "golden")
KEYS = ("input_ids", "attention_mask", "pixel_values", "labels")
GRAD_KEYS = ["lm_head.weight", "text.embed_tokens.weight", "text.layers.0.qkv.weight", "text.layers.1.down.weight",
             "text.norm.weight", "projector.linear_1.weight", "projector.linear_2.bias"]


@pytest.fixture(scope="module")
def models(dev):
    from multimodal_unlearning_amd.llava import LlavaHIP

    cfg = tiny_test_config()
    P = init_params(cfg, seed=0)
    student = LlavaHIP(cfg, P, device=dev)
    teacher = LlavaHIP(cfg, teacher_params_from(P, seed=7), device=dev, trainable=False)
    return cfg, P, student, teacher


def _relerr(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


@pytest.mark.parametrize("tag", ["forget", "padded"])
def test_forward_backward_vs_hf_golden_and_oracle(models, tag):
    from oracle.llava_ref import llava_forward

    cfg, P, student, _ = models
    g = np.load(os.path.join(G, "llava_tiny_hf.npz"))
    sp = {k: torch.from_numpy(g[f"{tag}.in.{k}"]) for k in KEYS}
    student.ensure_grad().zero_()
    out = student(**sp)
    out.loss.backward()
    torch.cuda.synchronize()
    valid = sp["attention_mask"].bool()
    ref_logits = torch.from_numpy(g[f"{tag}.logits"])
    e = _relerr(out.logits.cpu()[valid], ref_logits[valid])
    assert e < 1e-2, e
    ref_loss = float(g[f"{tag}.loss"])
    assert abs(out.loss.item() - ref_loss) < 1e-3 * max(1.0, abs(ref_loss)), (out.loss.item(), ref_loss)
    for k in GRAD_KEYS:
        gn = student.g[k].float().norm().item()
        rn = float(g[f"{tag}.gradnorm.{k}"])
        assert abs(gn - rn) < 2e-2 * max(rn, 1e-3), (k, gn, rn)
    assert _relerr(student.g["text.norm.weight"], torch.from_numpy(g[f"{tag}.grad.text.norm.weight"])) < 2e-2
    # live oracle in bf16 (the HF CPU bf16 path): the HIP path must be at least as close to fp32 as it is
    Pb = {k: v.clone() for k, v in P.items()}
    ob = llava_forward(cfg, Pb, **sp, dtype=torch.bfloat16)
    e_bf16_oracle = _relerr(ob["logits"].float()[valid], ref_logits[valid])
    assert e < 2.0 * e_bf16_oracle + 1e-3, (e, e_bf16_oracle)
    # full-tensor gradient check against the fp32 oracle's autograd
    Pf = {k: v.clone().requires_grad_(not k.startswith("vision.")) for k, v in P.items()}
    of = llava_forward(cfg, Pf, **sp)
    of["loss"].backward()
    for k in GRAD_KEYS + ["text.layers.0.gate_up.weight", "text.layers.1.o.weight", "text.layers.0.input_norm.weight",
                          "projector.linear_1.bias", "projector.linear_2.weight"]:
        assert _relerr(student.g[k], Pf[k].grad) < 3e-2, (k, _relerr(student.g[k], Pf[k].grad))


LOSS_TYPES = ["retain_ft", "grad_ascent", "grad_diff_forget_ce_retain_ce", "grad_diff_forget_entropy_retain_ce",
              "grad_diff_forget_kl_retain_ce", "grad_diff_forget_ce_retain_kl", "KL", "scrub", "npo", "idk", "eco_ft",
              "dpo", "llmu"]


@pytest.mark.parametrize("lt", LOSS_TYPES)
def test_compute_loss_vs_reference_golden(models, lt):
    from multimodal_unlearning_amd.trainer import MMTrainerForgetting

    cfg, P, student, teacher = models
    g = np.load(os.path.join(G, "compute_loss_tiny.npz"))
    inputs = {s: {k: torch.from_numpy(g[f"in.{s}.{k}"]) for k in KEYS} for s in ("forget", "retain", "idk")}
    tr = MMTrainerForgetting(model=student, forget_loss=lt, teacher_model=teacher, loss_beta=0.7, l1_lambda=0.0,
                             l0_lambda=0.0, l_norm_from="zero", loss_args={})
    student.ensure_grad().zero_()
    loss = tr.compute_loss(student, inputs)
    loss.backward()
    torch.cuda.synchronize()
    ref = float(g[f"{lt}.loss"])
    assert abs(loss.item() - ref) < 1e-3 * max(1.0, abs(ref)), (lt, loss.item(), ref)
    tot = student.ensure_grad().float().norm().item()
    rt = float(g[f"{lt}.gradnorm.total"])
    assert abs(tot - rt) < 2e-2 * rt, (lt, tot, rt)
    for k in GRAD_KEYS:
        gn = student.g[k].float().norm().item()
        rn = float(g[f"{lt}.gradnorm.{k}"])
        assert abs(gn - rn) < 3e-2 * max(rn, 1e-2 * rt), (lt, k, gn, rn)


def test_l1_regulariser_and_errors(models):
    from multimodal_unlearning_amd.trainer import MMTrainerForgetting

    cfg, P, student, teacher = models
    g = np.load(os.path.join(G, "compute_loss_tiny.npz"))
    inputs = {s: {k: torch.from_numpy(g[f"in.{s}.{k}"]) for k in KEYS} for s in ("forget", "retain", "idk")}
    tr = MMTrainerForgetting(model=student, forget_loss="grad_ascent", teacher_model=teacher, loss_beta=0.7,
                             l1_lambda=1e-4, l0_lambda=0.0, l_norm_from="init", loss_args={})
    loss = tr.compute_loss(student, inputs)
    ref = float(g["grad_ascent+l1_init.loss"])
    assert abs(loss.item() - ref) < 2e-3 * max(1.0, abs(ref)), (loss.item(), ref)
    with pytest.raises(ValueError):
        MMTrainerForgetting(model=student, forget_loss="grad_ascent", teacher_model=None, loss_beta=1.0,
                            l1_lambda=0.1, l0_lambda=0.0, l_norm_from="bogus")
    with pytest.raises(ValueError):
        MMTrainerForgetting(model=student, forget_loss="not_a_loss", teacher_model=None, loss_beta=1.0, l1_lambda=0,
                            l0_lambda=0, l_norm_from="zero").compute_loss(student, inputs)
    with pytest.raises(ValueError):
        MMTrainerForgetting(model=student, forget_loss="grad_diff_forget_x_retain_ce", teacher_model=None, loss_beta=1.0,
                            l1_lambda=0, l0_lambda=0, l_norm_from="zero").compute_loss(student, inputs)
    with pytest.raises(ValueError):  # L1 from init without a teacher
        MMTrainerForgetting(model=student, forget_loss="grad_ascent", teacher_model=None, loss_beta=1.0, l1_lambda=0.1,
                            l0_lambda=0, l_norm_from="init").compute_loss(student, inputs)


def test_paired_fusion_is_equivalent(models):
    from multimodal_unlearning_amd.trainer import MMTrainerForgetting

    cfg, P, student, _ = models
    inputs = make_forget_batch(cfg, 3, 40, seed=5, n_question=7)
    res = {}
    for fuse in (True, False):
        tr = MMTrainerForgetting(model=student, forget_loss="grad_diff_forget_ce_retain_ce", teacher_model=None,
                                 loss_beta=1.3, l1_lambda=0, l0_lambda=0, l_norm_from="zero",
                                 loss_args={"fuse_pairs": fuse})
        student.ensure_grad().zero_()
        loss = tr.compute_loss(student, inputs)
        loss.backward()
        res[fuse] = (loss.item(), student.ensure_grad().clone())
    # forward rows are independent -> identical logits and CE; wgrads differ only by bf16 accumulation order
    assert abs(res[True][0] - res[False][0]) < 1e-5 * max(1, abs(res[False][0]))
    assert _relerr(res[True][1], res[False][1]) < 1e-2


@pytest.mark.parametrize("lt", ["grad_diff_forget_ce_retain_ce", "grad_ascent", "KL", "idk", "llmu"])
def test_loss_only_matches_full_logits(models, lt):
    """Forwards of which only `.loss` is read run the final norm + lm_head + CE on the supervised rows only
    (LlavaHIP.ce_losses).  Same loss and gradients as the full-logits forward: every supervised row goes through the same
    kernels with the same operands; only the order of the CE sum and of the wgrad accumulation differs.  Ragged batch:
    left padding, a different number of supervised tokens per sequence, labels on the host and on the device."""
    from multimodal_unlearning_amd.trainer import MMTrainerForgetting

    cfg, P, student, teacher = models
    inputs = make_forget_batch(cfg, 3, 40, seed=15, n_question=7, splits=("forget", "retain", "idk"))
    for name, sp in inputs.items():  # ragged supervision + left padding on one sequence
        sp["labels"][1, :25] = -100
        sp["labels"][2, 30:] = -100
        sp["attention_mask"][0, :3] = 0
        sp["labels"][0, :3] = -100
    res = {}
    for mode in ("full", "rows_host_labels", "rows_device_labels"):
        tr = MMTrainerForgetting(model=student, forget_loss=lt, teacher_model=teacher, loss_beta=1.3, l1_lambda=0,
                                 l0_lambda=0, l_norm_from="zero", loss_args={"loss_only_head": mode != "full"})
        x = inputs if mode != "rows_device_labels" else {k: {a: b.to(student.device) for a, b in v.items()} for k, v in inputs.items()}
        student.ensure_grad().zero_()
        loss = tr.compute_loss(student, x)
        loss.backward()
        res[mode] = (loss.item(), student.ensure_grad().clone())
    for mode in ("rows_host_labels", "rows_device_labels"):
        assert abs(res[mode][0] - res["full"][0]) < 2e-6 * max(1, abs(res["full"][0])), (mode, res[mode][0], res["full"][0])
        assert _relerr(res[mode][1], res["full"][1]) < 5e-3, (mode, _relerr(res[mode][1], res["full"][1]))
    # return_outputs=True keeps the full forward (somebody reads the logits)
    tr = MMTrainerForgetting(model=student, forget_loss="grad_ascent", teacher_model=None, loss_beta=1.0, l1_lambda=0,
                             l0_lambda=0, l_norm_from="zero")
    loss, out = tr.compute_loss(student, inputs, return_outputs=True)
    assert out.logits is not None and out.logits.shape[:2] == inputs["forget"]["input_ids"].shape
    assert tr.compute_loss(student, inputs, return_outputs=False).item() == pytest.approx(loss.item(), rel=2e-6)


def test_transposed_weight_dgrad_matches(models):
    """dgrad through the resident W^T copies (NT path) == dgrad through W (NN path)."""
    from multimodal_unlearning_amd.trainer import MMTrainerForgetting

    cfg, P, student, _ = models
    inputs = make_forget_batch(cfg, 2, 40, seed=9, n_question=7)
    tr = MMTrainerForgetting(model=student, forget_loss="grad_diff_forget_ce_retain_ce", teacher_model=None,
                             loss_beta=1.0, l1_lambda=0, l0_lambda=0, l_norm_from="zero")
    res = []
    for use_t in (False, True):
        if use_t:
            student.enable_transposed_weights()
        student.ensure_grad().zero_()
        tr.compute_loss(student, inputs).backward()
        res.append(student.ensure_grad().clone())
    student.wT.clear()
    student._wT_flat = None
    # same products, same fp32 accumulation; only the k-summation order inside a tile may differ
    assert _relerr(res[1], res[0]) < 2e-3, _relerr(res[1], res[0])


def test_transposed_wgrad_matches(models):
    """wgrad via [transpose, transpose, NT GEMM] == wgrad via the TN GEMM."""
    from multimodal_unlearning_amd.trainer import MMTrainerForgetting

    cfg, P, student, _ = models
    inputs = make_forget_batch(cfg, 2, 40, seed=11, n_question=7)  # fused batch: M = 160 rows
    tr = MMTrainerForgetting(model=student, forget_loss="grad_diff_forget_ce_retain_ce", teacher_model=None,
                             loss_beta=1.0, l1_lambda=0, l0_lambda=0, l_norm_from="zero")
    res = []
    old = student.transposed_wgrad_min_m
    for min_m in (1 << 30, 0):
        student.transposed_wgrad_min_m = min_m
        student.ensure_grad().zero_()
        tr.compute_loss(student, inputs).backward()
        res.append(student.ensure_grad().clone())
    student.transposed_wgrad_min_m = old
    assert _relerr(res[1], res[0]) < 2e-3, _relerr(res[1], res[0])


def test_engine_steps_reduce_retain_loss(models, dev):
    from multimodal_unlearning_amd.engine import ForgetEngine
    from multimodal_unlearning_amd.llava import LlavaHIP
    from multimodal_unlearning_amd.trainer import MMTrainerForgetting

    cfg, P, _, _ = models
    model = LlavaHIP(cfg, P, device=dev)
    tr = MMTrainerForgetting(model=model, forget_loss="retain_ft", teacher_model=None, loss_beta=1.0, l1_lambda=0,
                             l0_lambda=0, l_norm_from="zero")
    eng = ForgetEngine(model, tr, ForgetConfig(lr=2e-3, weight_decay=0.0), total_steps=50, warmup_steps=2)
    mbs = [make_forget_batch(cfg, 2, 32, seed=s, n_question=5) for s in (1, 2)]
    losses = [eng.step(mbs, sync_loss=True).loss for _ in range(8)]
    assert all(np.isfinite(losses)), losses
    assert losses[-1] < losses[0] - 0.5, losses
    assert model.ensure_grad().abs().max().item() == 0  # fused zeroing after the optimiser step


def test_engine_rccl_path_single_rank(models, dev):
    """The DP code paths (bucketed async all-reduce launched from the backward sweep; reduce-scatter + sharded AdamW +
    all-gather) on RCCL (backend "nccl") with a 1-rank process group: results must equal the collective-free step."""
    import torch.distributed as dist

    from multimodal_unlearning_amd.engine import ForgetEngine
    from multimodal_unlearning_amd.llava import LlavaHIP
    from multimodal_unlearning_amd.trainer import MMTrainerForgetting

    cfg, P, _, _ = models
    import os
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        created = True
    try:
        outs = []
        # (collectives forced, dp_mode, bucket size): rs_ag on RCCL runs the IN-PLACE reduce_scatter_tensor (output a view
        # of the input) and all_gather_into_tensor, the per-shard AdamW state offsets with rebased no-decay ranges and the
        # sharded |g|^2 — on gloo that mode goes through temporaries, so this is the only place the RCCL forms execute
        # (one rank: a wrong shard offset or an aliasing fault would still show; the multi-rank exchange itself cannot)
        for force, mode, bucket in ((False, "allreduce", 1 << 27), (True, "allreduce", 1 << 27), (True, "rs_ag", 1 << 27),
                                    (True, "rs_ag", 1 << 14), (True, "allreduce", 1 << 14)):
            model = LlavaHIP(cfg, P, device=dev)
            tr = MMTrainerForgetting(model=model, forget_loss="grad_diff_forget_ce_retain_ce", teacher_model=None,
                                     loss_beta=1.0, l1_lambda=0, l0_lambda=0, l_norm_from="zero")
            eng = ForgetEngine(model, tr, ForgetConfig(lr=1e-3), total_steps=10, warmup_steps=1, force_collectives=force,
                               dp_mode=mode, bucket_elems=bucket)
            mbs = [make_forget_batch(cfg, 2, 32, seed=s, n_question=5) for s in (1, 2)]
            for _ in range(2):
                st = eng.step(mbs, sync_loss=True)
            assert eng.reducer.enabled == force
            if bucket == 1 << 14:
                assert len(eng.reducer.buckets) > 8
            outs.append((st.loss, model.train_flat.detach().clone()))
        for o in outs[1:]:
            assert abs(outs[0][0] - o[0]) < 1e-5
            # not bit-equal run to run: the embedding backward sums with fp32 atomics (arrival order varies) and AdamW
            # turns a last-bit gradient difference into a +-lr step on near-zero-gradient elements
            assert _relerr(o[1], outs[0][1]) < 2e-3
    finally:
        if created:
            dist.destroy_process_group()


# ------------------------------------------------------------------------ extra goldens: rmu / sku / nlp trainer
@pytest.fixture(scope="module")
def models8(dev):
    from multimodal_unlearning_amd.llava import LlavaHIP

    cfg = tiny_test_config(text_layers=8)
    P = init_params(cfg, seed=0)
    return cfg, LlavaHIP(cfg, P, device=dev), LlavaHIP(cfg, teacher_params_from(P, seed=7), device=dev, trainable=False)


def _check_extra(g, tag, loss, student, tol_loss=1e-3, tol_g=2e-2):
    ref = float(g[f"{tag}.loss"])
    assert abs(loss.item() - ref) < tol_loss * max(1.0, abs(ref)), (tag, loss.item(), ref)
    tot = student.ensure_grad().float().norm().item()
    rt = float(g[f"{tag}.gradnorm.total"])
    assert abs(tot - rt) < tol_g * rt, (tag, tot, rt)


def test_rmu_and_sku_vs_reference_golden(models8):
    from multimodal_unlearning_amd.trainer import MMTrainerForgetting

    cfg, student, teacher = models8
    g = np.load(os.path.join(G, "compute_loss_extra.npz"))
    inputs = {s: {k: torch.from_numpy(g[f"mm.in.{s}.{k}"]) for k in KEYS} for s in ("forget", "retain", "idk")}
    tr = MMTrainerForgetting(model=student, forget_loss="RMU", teacher_model=teacher, loss_beta=0.7, l1_lambda=0.0,
                             l0_lambda=0.0, l_norm_from="zero", loss_args={})
    tr.control_vec = torch.from_numpy(g["mm.rmu.control_vec"])
    student.ensure_grad().zero_()
    loss = tr.compute_loss(student, inputs)
    loss.backward()
    _check_extra(g, "mm.rmu", loss, student)
    inputs["random"] = {k: torch.from_numpy(g[f"mm.in.random.{k}"]) for k in KEYS}
    keep = {k: v.clone() for k, v in inputs["random"].items()}
    tr = MMTrainerForgetting(model=student, forget_loss="sku", teacher_model=teacher, loss_beta=0.7, l1_lambda=0.0,
                             l0_lambda=0.0, l_norm_from="zero",
                             loss_args={"bad_weight": 2.5, "random_weight": 1.0, "normal_weight": 0.5})
    student.ensure_grad().zero_()
    loss = tr.compute_loss(student, inputs)
    loss.backward()
    _check_extra(g, "mm.sku", loss, student)
    assert all(torch.equal(inputs["random"][k], keep[k]) for k in keep)  # inputs are not mutated (the reference does)


NLP_TYPES = ["grad_ascent", "grad_diff_forget_ce_retain_ce", "grad_diff_forget_KL_retain_ce", "KL", "scrub", "npo", "idk",
             "eco_ft", "dpo", "LLMU", "RMU"]


@pytest.mark.parametrize("lt", NLP_TYPES)
def test_nlp_tuple_trainer_vs_reference_golden(models8, lt):
    """BASELINE config 1 surface: nlp/dataloader.py CustomTrainerForgetting on tuple inputs, text-only model."""
    from multimodal_unlearning_amd.trainer import CustomTrainerForgetting

    cfg, student, teacher = models8
    g = np.load(os.path.join(G, "compute_loss_extra.npz"))
    tup = [[torch.from_numpy(g[f"nlp.in.{s}.{k}"]) for k in ("input_ids", "labels", "attention_mask")]
           for s in ("forget", "retain", "idk")]
    tr = CustomTrainerForgetting(model=student, forget_loss=lt, oracle_model=teacher, loss_beta=2.0 if lt == "npo" else 0.7,
                                 l1_lambda=0.0, l0_lambda=0.0)
    if lt == "RMU":
        rv = torch.from_numpy(g["nlp.RMU.rand_vec"])
        tr.control_vec = rv / torch.norm(rv) * 300
    student.ensure_grad().zero_()
    loss = tr.compute_loss(student, tup)
    loss.backward()
    # the nlp NPO value is a mean of log(1 + log-ratio^2) ~ 4e-2: absolute 1e-3 of max(1,|loss|) still applies
    _check_extra(g, f"nlp.{lt}", loss, student, tol_g=3e-2 if lt == "npo" else 2e-2)
    with pytest.raises(ValueError):
        CustomTrainerForgetting(model=student, forget_loss="kl", oracle_model=teacher, loss_beta=1.0, l1_lambda=0.0,
                                l0_lambda=0.0).compute_loss(student, tup)  # the text trainer's names are case-sensitive


# --------------------------------------------------------------------------------------------------- LoRA
def test_lora_grad_ascent_vs_oracle(dev):
    """BASELINE config 2 surface (grad_ascent, LoRA r=8 alpha=32 on decoder + projector + lm_head Linears).  peft is not
    installed in this image, so the adapter arithmetic is pinned to the published LoRA formula
    y = W x + (alpha/r) B A x through the fp32 oracle run on W_eff = W + (alpha/r) B A (parity unpinned vs peft)."""
    from multimodal_unlearning_amd.llava import LlavaHIP, LoraSpec
    from multimodal_unlearning_amd.trainer import MMTrainerForgetting
    from oracle.llava_ref import llava_forward

    cfg = tiny_test_config()
    P = init_params(cfg, seed=0)
    spec = LoraSpec(r=8, alpha=32.0)
    model = LlavaHIP(cfg, P, device=dev, lora=spec)
    model.init_lora_(seed=3, b_std=0.02)
    assert model.train_flat.numel() < 0.2 * sum(v.numel() for v in P.values())
    inputs = make_forget_batch(cfg, 2, 40, seed=21, n_question=7)
    tr = MMTrainerForgetting(model=model, forget_loss="grad_ascent", teacher_model=None, loss_beta=1.0, l1_lambda=0,
                             l0_lambda=0, l_norm_from="zero")
    model.ensure_grad().zero_()
    loss = tr.compute_loss(model, inputs)
    loss.backward()
    torch.cuda.synchronize()
    # oracle on merged weights, adapters as fp32 leaves
    leaves = {}
    Pe = {k: v.clone() for k, v in P.items()}
    for wname, ad in model.lora.items():
        A = ad.A32.detach().float().cpu().requires_grad_(True)
        Bm = ad.BL.detach().float().cpu().requires_grad_(True)
        leaves[wname] = (A, Bm, ad.mask.float().cpu())
        Pe[wname] = P[wname] + spec.scaling * (Bm @ A)
    ref = -llava_forward(cfg, Pe, **inputs["forget"])["loss"]
    ref.backward()
    assert abs(loss.item() - ref.item()) < 1e-3 * max(1.0, abs(ref.item())), (loss.item(), ref.item())
    num = den = 0.0
    for wname, (A, Bm, mask) in leaves.items():
        base = wname[: -len(".weight")]
        gA, gB = model.g[base + ".lora_A"].float().cpu(), model.g[base + ".lora_B"].float().cpu()
        rB = Bm.grad * mask
        for got, exp in ((gA, A.grad), (gB, rB)):
            num += float((got - exp).pow(2).sum())
            den += float(exp.pow(2).sum())
        assert float((gB * (1 - mask)).abs().max()) == 0.0  # off-block entries are not parameters
    assert (num / den) ** 0.5 < 3e-2, (num / den) ** 0.5
    # the CLIP tower's adapters (peft wraps every nn.Linear): gradients arrive through the CLIP backward sweep
    vis = [w for w in leaves if w.startswith("vision.")]
    assert len(vis) == 4 * cfg.vision_layers_used
    vn = vd = 0.0
    for wname in vis:
        A, Bm, mask = leaves[wname]
        base = wname[: -len(".weight")]
        for got, exp in ((model.g[base + ".lora_A"].float().cpu(), A.grad), (model.g[base + ".lora_B"].float().cpu(), Bm.grad * mask)):
            vn += float((got - exp).pow(2).sum())
            vd += float(exp.pow(2).sum())
    assert vd > 0 and (vn / vd) ** 0.5 < 5e-2, (vn / max(vd, 1e-30)) ** 0.5
    st = model.lora_state()
    assert st["text.layers.0.qkv.lora_A"].shape == (24, cfg.text.hidden_size)
    assert st["text.layers.0.qkv.lora_B"].shape == (3 * cfg.text.hidden_size, 8)


def test_lora_engine_step_trains_only_adapters(dev):
    from multimodal_unlearning_amd.engine import ForgetEngine
    from multimodal_unlearning_amd.llava import LlavaHIP, LoraSpec
    from multimodal_unlearning_amd.trainer import MMTrainerForgetting

    cfg = tiny_test_config()
    P = init_params(cfg, seed=0)
    model = LlavaHIP(cfg, P, device=dev, lora=LoraSpec(r=8, alpha=32.0))
    base0 = model._base_flat.clone()
    tr = MMTrainerForgetting(model=model, forget_loss="retain_ft", teacher_model=None, loss_beta=1.0, l1_lambda=0,
                             l0_lambda=0, l_norm_from="zero")
    eng = ForgetEngine(model, tr, ForgetConfig(lr=5e-3, weight_decay=0.0), total_steps=60, warmup_steps=2)
    mbs = [make_forget_batch(cfg, 2, 32, seed=s, n_question=5) for s in (1, 2)]
    losses = [eng.step(mbs, sync_loss=True).loss for _ in range(10)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0] - 0.2, losses
    assert torch.equal(model._base_flat, base0)  # frozen base weights untouched


def test_eval_truth_ratio_pass_vs_reference_golden(models):
    """mm/eval.py:46-92 on the HIP model vs the reference's own output (tests/golden/eval_tiny.json, generated by
    oracle/make_golden_eval.py), then the forget-quality metric computed from both logs."""
    import json

    from multimodal_unlearning_amd import metrics
    from multimodal_unlearning_amd.eval_util import eval_perturbation_ratio
    from multimodal_unlearning_amd.synthetic import make_split

    cfg, P, student, _ = models
    gold = json.load(open(os.path.join(G, "eval_tiny.json")))
    B, K, T = gold["seeds"]["B"], gold["seeds"]["K"], gold["seeds"]["T"]
    batches, pbatches = [], []
    for it in range(2):
        sp = make_split(cfg, B, T, seed=700 + it, n_question=6)
        sp["indices"] = torch.arange(B) + it * B
        ps = [make_split(cfg, B, T, seed=800 + 10 * it + k, n_question=6) for k in range(K)]
        pb = {k: torch.stack([p[k] for p in ps], dim=1) for k in ps[0]}
        pb["pixel_values"] = pb["pixel_values"].reshape(B * K, *ps[0]["pixel_values"].shape[1:])
        pb["indices"] = sp["indices"].clone()
        batches.append(sp)
        pbatches.append(pb)
    logs = eval_perturbation_ratio(batches, pbatches, student)
    for key in ("avg_paraphrased_loss", "truth_ratio", "paraphrased_loss"):
        for i, v in gold["logs"][key].items():
            got = logs[key][int(i)]
            assert abs(got - v) < 2e-3 * max(1.0, abs(v)), (key, i, got, v)
    for i, v in gold["logs"]["average_perturb_loss"].items():
        assert max(abs(a - b) for a, b in zip(logs["average_perturb_loss"][int(i)], v)) < 2e-2
    assert {int(i): v for i, v in gold["logs"]["num_token_perturb"].items()} == dict(logs["num_token_perturb"])
    # metric on (our log vs the reference's log) must see "the same model": KS p-value 1, JS distance ~ 0
    as_log = lambda d: {"eval_log_forget.json": {k: {str(i): v for i, v in d[k].items()} for k in
                                                 ("avg_paraphrased_loss", "average_perturb_loss")}}
    fq = metrics.evaluate_forget_quality(as_log(logs), {"eval_log_forget.json": gold["logs"]})
    assert fq["KS test p-value"] > 0.99 and fq["JS metric"] < 1e-3, fq


def test_lora_splitk_wgrad_path_matches(dev):
    """Adapter wgrads through the split-K fp32 path (used for >= 2048 tokens) == the plain TN GEMM path."""
    from multimodal_unlearning_amd.llava import LlavaHIP, LoraSpec
    from multimodal_unlearning_amd.trainer import MMTrainerForgetting

    cfg = tiny_test_config()
    model = LlavaHIP(cfg, init_params(cfg, seed=0), device=dev, lora=LoraSpec(r=8, alpha=32.0))
    model.init_lora_(seed=3, b_std=0.02)
    inputs = make_forget_batch(cfg, 4, 64, seed=23, n_question=9)
    tr = MMTrainerForgetting(model=model, forget_loss="grad_ascent", teacher_model=None, loss_beta=1.0, l1_lambda=0,
                             l0_lambda=0, l_norm_from="zero")
    res = []
    for thr in (1 << 30, 0):
        model.lora_splitk_min_m = thr
        model.ensure_grad().zero_()
        tr.compute_loss(model, inputs).backward()
        res.append(model.ensure_grad().clone())
    assert _relerr(res[1], res[0]) < 3e-3, _relerr(res[1], res[0])


def test_lora_merge_and_adapter_io(dev, tmp_path):
    """mm/forget.py:100-113 flow: load a peft-layout adapter, merge it into the base weights on the device, save a plain
    checkpoint.  Merged logits == adapter logits (bf16 tolerance), merged weights == W + (alpha/r) B A (fp32 formula,
    one bf16 rounding), and a fresh non-LoRA model loaded from the saved checkpoint computes the merged function."""
    from multimodal_unlearning_amd.checkpoint import load_adapter, load_pretrained, merge_adapter_and_save, save_adapter
    from multimodal_unlearning_amd.llava import LlavaHIP, LoraSpec
    from multimodal_unlearning_amd.synthetic import make_split

    cfg = tiny_test_config()
    P = init_params(cfg, seed=0)
    spec = LoraSpec(r=8, alpha=32.0)
    src = LlavaHIP(cfg, P, device=dev, lora=spec)
    src.init_lora_(seed=5, b_std=0.05)
    save_adapter(cfg, src.lora_state(), str(tmp_path / "adapter"), r=8, alpha=32.0)

    model = LlavaHIP(cfg, P, device=dev, lora=spec)
    st, r, alpha = load_adapter(cfg, str(tmp_path / "adapter"), [n[: -len(".weight")] for n in model.lora])
    assert (r, alpha) == (8, 32.0)
    model.load_lora_state_(st)
    for wname, ad in model.lora.items():
        assert torch.equal(ad.A32, src.lora[wname].A32) and torch.equal(ad.BL, src.lora[wname].BL), wname
    sp = make_split(cfg, 2, 40, seed=31, n_question=7)
    sp = {k: v.to(dev) for k, v in sp.items()}
    with torch.no_grad():
        before = model.logits(sp["input_ids"], sp["attention_mask"], sp["pixel_values"]).float()
    expect = {}
    for wname, ad in model.lora.items():
        expect[wname] = (model.w[wname].float() + spec.scaling * (ad.BL.float() @ ad.A32.float())).to(torch.bfloat16)
    files = merge_adapter_and_save(model, str(tmp_path / "merged"))
    assert files
    for wname, ad in model.lora.items():
        diff = (model.w[wname].float() - expect[wname].float()).abs()
        # fp32 accumulate + one rounding on both sides; only fp32 summation order may flip a rounding tie (1 bf16 ulp)
        assert float(diff.max()) <= 2.0 ** -7 * float(expect[wname].float().abs().max()), wname
        assert float((diff > 0).float().mean()) < 1e-2, wname
        assert float(ad.BL.abs().max()) == 0.0
    with torch.no_grad():
        after = model.logits(sp["input_ids"], sp["attention_mask"], sp["pixel_values"]).float()
    assert _relerr(after, before) < 2e-2, _relerr(after, before)  # bf16: delta folded into W vs added in the epilogue
    plain = LlavaHIP(cfg, load_pretrained(cfg, str(tmp_path / "merged")), device=dev)
    with torch.no_grad():
        again = plain.logits(sp["input_ids"], sp["attention_mask"], sp["pixel_values"]).float()
    assert torch.equal(again, after)


def test_keep_act_t_is_equivalent(dev):
    """Keeping act^T from the forward (one fused SwiGLU pass writing act and act^T) gives the same loss and gradients,
    bit for bit, as recomputing act^T from gate|up in backward."""
    from multimodal_unlearning_amd.llava import LlavaHIP
    from multimodal_unlearning_amd.trainer import MMTrainerForgetting

    cfg = tiny_test_config()
    model = LlavaHIP(cfg, init_params(cfg, seed=0), device=dev)
    model.transposed_wgrad_min_m = 0
    inputs = make_forget_batch(cfg, 2, 64, seed=29, n_question=9)
    tr = MMTrainerForgetting(model=model, forget_loss="grad_diff_forget_ce_retain_ce", teacher_model=None, loss_beta=1.0,
                             l1_lambda=0, l0_lambda=0, l_norm_from="zero")
    res = []
    for keep in (False, True):
        model.keep_act_t = keep
        model.ensure_grad().zero_()
        loss = tr.compute_loss(model, inputs)
        loss.backward()
        res.append((float(loss), model.ensure_grad().clone()))
    assert res[0][0] == res[1][0]
    assert torch.equal(res[0][1], res[1][1])


@pytest.mark.parametrize("loss", ["grad_diff_forget_ce_retain_ce", "KL"])
def test_overlapped_optimizer_is_bit_identical(dev, loss):
    """ForgetEngine(overlap_optimizer=True): AdamW and the W^T refresh of a step run on a side stream, one chunk per decoder
    layer in forward order, under the next step's forward, which waits for a chunk's event right before it reads the chunk.
    Same kernels, same order per element: parameters, moments and W^T copies after four steps equal the serial engine's bit
    for bit (merged and looped micro-batches; a no-grad forward between steps also honours the events)."""
    from multimodal_unlearning_amd.engine import ForgetEngine
    from multimodal_unlearning_amd.llava import LlavaHIP
    from multimodal_unlearning_amd.trainer import MMTrainerForgetting

    cfg = tiny_test_config(text_layers=3)
    P = init_params(cfg, seed=0)
    data = [[make_forget_batch(cfg, 2, 40, seed=90 + 10 * k + i, n_question=6) for i in range(2)] for k in range(4)]
    res = {}
    for overlap in (False, True):
        for merge in (True, False):
            model = LlavaHIP(cfg, P, device=dev)
            teacher = LlavaHIP(cfg, teacher_params_from(P, seed=7), device=dev, trainable=False)
            tr = MMTrainerForgetting(model=model, forget_loss=loss, teacher_model=teacher, loss_beta=1.0, l1_lambda=0, l0_lambda=0,
                                     l_norm_from="zero")
            eng = ForgetEngine(model, tr, ForgetConfig(lr=1e-3), total_steps=10, warmup_steps=1, merge_ga=merge,
                               overlap_optimizer=overlap)
            assert eng.overlap_optimizer == overlap
            probes = []
            for k, mbs in enumerate(data):
                eng.step(mbs)
                if k == 1:  # a forward that is not part of a step (evaluation between steps) sees the updated parameters
                    with torch.no_grad():
                        probes.append(float(model(**data[0][0]["forget"]).loss))
            eng.finish_optimizer()
            torch.cuda.synchronize()
            res[overlap, merge] = (model.train_flat.detach().clone(), eng.m.clone(), eng.v.clone(),
                                   model.wT["text.layers.2.down.weight"].clone(), model.wT["lm_head.weight"].clone(), probes)
    for merge in (True, False):
        a, b = res[True, merge], res[False, merge]
        for x, y in zip(a[:5], b[:5]):
            assert torch.equal(x, y), merge
        assert a[5] == b[5]


def test_dpo_variants_the_reference_computes_and_overwrites(models):
    """mm/trainer.py:390-396 assigns `loss` three times: DPO proper (-logsigmoid(0.1 * (pi - ref)).mean()), then -pi_logratios
    .mean(), then -idk_loss_current.mean() (what it trains on; golden-pinned elsewhere).  `loss_args={"dpo_variant": ...}`
    offers the first two; checked against the same expressions evaluated with the fp32 oracle's per-sequence log-likelihoods,
    loss and gradient norm."""
    from multimodal_unlearning_amd.trainer import MMTrainerForgetting
    from oracle.llava_ref import llava_forward

    cfg, P, student, teacher = models
    inputs = make_forget_batch(cfg, 3, 40, seed=49, n_question=7, splits=("forget", "retain", "idk"))
    Pt = teacher_params_from(P, seed=7)

    def seq_logp(params, split):
        lg = llava_forward(cfg, params, **{k: v for k, v in split.items() if k != "labels"})["logits"].float()
        lab = torch.nn.functional.pad(split["labels"], (0, 1), value=-100)[:, 1:]
        ce = torch.nn.functional.cross_entropy(lg.transpose(1, 2), lab, ignore_index=-100, reduction="none")
        return -ce.sum(-1)

    for variant in ("logsigmoid", "pi_logratios"):
        tr = MMTrainerForgetting(model=student, forget_loss="dpo", teacher_model=teacher, loss_beta=1.0, l1_lambda=0, l0_lambda=0,
                                 l_norm_from="zero", loss_args={"dpo_variant": variant})
        student.ensure_grad().zero_()
        loss = tr.compute_loss(student, inputs)
        loss.backward()
        torch.cuda.synchronize()
        Pf = {k: v.clone().requires_grad_(not k.startswith("vision.")) for k, v in P.items()}
        lp_idk, lp_forget = seq_logp(Pf, inputs["idk"]), seq_logp(Pf, inputs["forget"])
        pi = lp_idk - lp_forget
        scale = float(torch.maximum(lp_idk.abs(), lp_forget.abs()).max())   # the values are differences of SUMS of ~30 CE terms each
        if variant == "pi_logratios":
            ref = -pi.mean()
        else:
            with torch.no_grad():
                rr = seq_logp(Pt, inputs["idk"]) - seq_logp(Pt, inputs["forget"])
            ref = -torch.nn.functional.logsigmoid(0.1 * (pi - rr)).mean()
        ref.backward()
        # 1e-3 relative to the per-sequence sums the value is made of (logsigmoid: times its beta = 0.1 slope)
        tol = 1e-3 * max(1.0, abs(ref.item()), scale * (0.1 if variant == "logsigmoid" else 1.0))
        assert abs(loss.item() - ref.item()) < tol, (variant, loss.item(), ref.item(), scale)
        gn = student.ensure_grad().float().norm().item()
        rn = sum(float(v.grad.double().pow(2).sum()) for v in Pf.values() if v.grad is not None) ** 0.5
        assert abs(gn - rn) < 3e-2 * rn, (variant, gn, rn)
    with pytest.raises(ValueError):
        MMTrainerForgetting(model=student, forget_loss="dpo", teacher_model=teacher, loss_args={"dpo_variant": "x"}).compute_loss(student, inputs)
    student.ensure_grad().zero_()
    student.live_sweeps = 0


def test_lora_rank_wider_than_the_fused_epilogue_vs_oracle(dev):
    """LoRA r = 32 (fused q|k|v adapter: 3 x 32 = 96 > the 64 columns the GEMM epilogue can carry): the adapter branch is then a
    second, accumulating GEMM (forward and dgrad), RoPE runs after it; loss and adapter gradients vs the fp32 oracle on
    W + (alpha/r) B A as for the fused ranks."""
    from multimodal_unlearning_amd.llava import LlavaHIP, LoraSpec
    from multimodal_unlearning_amd.trainer import MMTrainerForgetting
    from oracle.llava_ref import llava_forward

    cfg = tiny_test_config()
    P = init_params(cfg, seed=0)
    spec = LoraSpec(r=32, alpha=64.0)
    model = LlavaHIP(cfg, P, device=dev, lora=spec)
    model.init_lora_(seed=3, b_std=0.02)
    assert model.lora["text.layers.0.qkv.weight"].kl == 96 and model.lora["text.layers.0.gate_up.weight"].kl == 64
    inputs = make_forget_batch(cfg, 2, 40, seed=23, n_question=7)
    tr = MMTrainerForgetting(model=model, forget_loss="grad_ascent", teacher_model=None, loss_beta=1.0, l1_lambda=0,
                             l0_lambda=0, l_norm_from="init")
    model.ensure_grad().zero_()
    loss = tr.compute_loss(model, inputs)
    loss.backward()
    torch.cuda.synchronize()
    leaves = {}
    Pe = {k: v.clone() for k, v in P.items()}
    for wname, ad in model.lora.items():
        A = ad.A32.detach().float().cpu().requires_grad_(True)
        Bm = ad.BL.detach().float().cpu().requires_grad_(True)
        leaves[wname] = (A, Bm, ad.mask.float().cpu())
        Pe[wname] = P[wname] + spec.scaling * ((Bm * ad.mask.float().cpu()) @ A)
    ref = -llava_forward(cfg, Pe, **inputs["forget"])["loss"]
    ref.backward()
    assert abs(loss.item() - ref.item()) < 1e-3 * max(1.0, abs(ref.item())), (loss.item(), ref.item())
    num = den = 0.0
    for wname, (A, Bm, mask) in leaves.items():
        base = wname[: -len(".weight")]
        for got, exp in ((model.g[base + ".lora_A"].float().cpu(), A.grad), (model.g[base + ".lora_B"].float().cpu(), Bm.grad * mask)):
            num += float((got - exp).pow(2).sum())
            den += float(exp.pow(2).sum())
    assert (num / den) ** 0.5 < 3e-2, (num / den) ** 0.5


def test_second_backward_through_a_loss_only_forward_raises(models):
    """the CE gradient of a loss-only forward overwrites the compact logits in place through the C ABI (invisible to
    autograd's version counters): a second backward through the same graph must fail loudly, not differentiate garbage."""
    cfg, P, student, _ = models
    sp = make_forget_batch(cfg, 2, 40, seed=47, n_question=7)["forget"]
    student.ensure_grad().zero_()
    loss = student(**sp, loss_only=True).loss
    loss.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="twice"):
        loss.backward()
    student.ensure_grad().zero_()
    student.live_sweeps = 0


def test_lora_dropout_forward_and_backward_vs_autograd(dev):
    """`LoRA.dropout` (mm/forget.py:128-136, config/mm/forget.yaml:6; peft: y = W x + (alpha/r) B A dropout(x)) on ONE adapted
    Linear of the model against torch autograd of the same expression with the mask the kernel generates (read back by
    running it on ones): output, dx, dA, dB.  peft is not importable and the mask stream is the library's own: parity unpinned
    vs peft's RNG; the ARITHMETIC is pinned here."""
    from multimodal_unlearning_amd import hip
    from multimodal_unlearning_amd.llava import LlavaHIP, LoraSpec

    cfg = tiny_test_config()
    p_drop = 0.25
    model = LlavaHIP(cfg, init_params(cfg, seed=0), device=dev, lora=LoraSpec(r=8, alpha=32.0, dropout=p_drop, seed=5))
    model.init_lora_(seed=3, b_std=0.05)
    model.ensure_grad().zero_()
    wname = "text.layers.0.down.weight"
    ad = model.lora[wname]
    M, n_in, n_out = 96, ad.A32.shape[1], ad.BL.shape[0]
    g = torch.Generator(device="cpu").manual_seed(0)
    x = (torch.randn(M, n_in, generator=g)).to(dev).to(torch.bfloat16)
    dy = (torch.randn(M, n_out, generator=g) * 0.1).to(dev).to(torch.bfloat16)
    model._drop_seeds, model._fwd_U = {}, {}
    out = torch.empty(M, n_out, device=dev, dtype=torch.bfloat16)
    model._lin_fwd(wname, x, out)
    seed = model._drop_seeds[wname]
    mask = hip.dropout(torch.ones_like(x), torch.empty_like(x), p_drop, seed).float()     # 0 or 1 / (1 - p)
    assert 0.6 < (mask != 0).float().mean().item() < 0.9
    W = model.w[wname].float()
    xr = x.float().requires_grad_(True)
    A = ad.A32.detach().float().requires_grad_(True)
    Bm = ad.BL.detach().float().requires_grad_(True)
    s = model.lora_spec.scaling
    ref = xr @ W.t() + s * ((xr * mask) @ A.t()) @ Bm.t()
    assert _relerr(out, ref) < 6e-3
    ref.backward(dy.float())
    model._bwd_U = model._fwd_U
    dx = torch.empty_like(x)
    model._lin_bwd(wname, dy, x, dx)
    base = wname[: -len(".weight")]
    assert _relerr(dx, xr.grad) < 8e-3, _relerr(dx, xr.grad)
    assert _relerr(model.g[base + ".lora_A"], A.grad) < 1e-2
    assert _relerr(model.g[base + ".lora_B"], Bm.grad * ad.mask.float()) < 1e-2
    model._drop_seeds = model._fwd_U = model._bwd_U = None


def test_lora_dropout_in_the_step(dev):
    """Dropout only acts in training forwards that build a graph; the loss moves with it, the engine trains, a recompute
    (gradient checkpointing) regenerates the same masks, and eval() / no-grad forwards see no dropout."""
    from multimodal_unlearning_amd.llava import LlavaHIP, LoraSpec
    from multimodal_unlearning_amd.trainer import MMTrainerForgetting

    cfg = tiny_test_config()
    inputs = make_forget_batch(cfg, 2, 40, seed=43, n_question=7)

    def build(p):
        m = LlavaHIP(cfg, init_params(cfg, seed=0), device=dev, lora=LoraSpec(r=8, alpha=32.0, dropout=p, seed=11))
        m.init_lora_(seed=3, b_std=0.05)
        return m, MMTrainerForgetting(model=m, forget_loss="grad_ascent", teacher_model=None, loss_beta=1.0, l1_lambda=0,
                                      l0_lambda=0, l_norm_from="init")

    m0, t0 = build(0.0)
    m1, t1 = build(0.3)
    l0 = float(t0.compute_loss(m0, inputs))
    l1 = t1.compute_loss(m1, inputs)
    assert abs(float(l1) - l0) > 1e-4                                  # the adapters' input was masked
    with torch.no_grad():
        assert float(t1.compute_loss(m1, inputs)) == float(t0.compute_loss(m0, inputs)) == l0   # no graph: no dropout
    m1.ensure_grad().zero_()
    l1.backward()
    g_plain = m1.ensure_grad().clone()
    assert float(g_plain.float().norm()) > 0
    # the same forward again draws NEW masks (another loss); a checkpointed one recomputes with the SAME masks it drew
    m2, t2 = build(0.3)
    m2.gradient_checkpointing_enable()
    m2.ensure_grad().zero_()
    l2 = t2.compute_loss(m2, inputs)
    assert float(l2) == float(l1)                                      # same spec seed, same call order: same masks
    l2.backward()
    assert torch.equal(m2.ensure_grad(), g_plain)
    m1.eval()
    m1.ensure_grad().zero_()
    le = t1.compute_loss(m1, inputs)
    assert float(le) == l0
    m1.train()


@pytest.mark.parametrize("lora", [False, True])
def test_gradient_checkpointing_is_equivalent(dev, lora):
    """`model.gradient_checkpointing_enable()` (mm/forget.py:123-124): the forward keeps only each decoder layer's input, the
    backward sweep re-runs the layer's forward with the same kernels — same loss and gradients bit for bit, full
    fine-tuning (transposed-wgrad path, act^T kept or not) and LoRA, at a fraction of the saved bytes."""
    from multimodal_unlearning_amd.llava import LlavaHIP, LoraSpec
    from multimodal_unlearning_amd.trainer import MMTrainerForgetting

    cfg = tiny_test_config(text_layers=4)
    model = LlavaHIP(cfg, init_params(cfg, seed=0), device=dev, lora=LoraSpec(r=8, alpha=32.0) if lora else None)
    if lora:
        model.init_lora_(seed=3, b_std=0.02)
    model.transposed_wgrad_min_m = 0
    teacher = LlavaHIP(cfg, teacher_params_from(init_params(cfg, seed=0), seed=7), device=dev, trainable=False)
    inputs = make_forget_batch(cfg, 2, 64, seed=33, n_question=9)
    for loss_type in ("grad_diff_forget_ce_retain_ce", "KL"):
        tr = MMTrainerForgetting(model=model, forget_loss=loss_type, teacher_model=teacher, loss_beta=1.0, l1_lambda=0,
                                 l0_lambda=0, l_norm_from="init" if lora else "zero")
        res = {}
        for keep in (False, True):
            for rc in (False, True):
                model.keep_act_t = keep
                (model.gradient_checkpointing_enable if rc else model.gradient_checkpointing_disable)()
                model.ensure_grad().zero_()
                torch.cuda.synchronize()
                torch.cuda.reset_peak_memory_stats()
                base = torch.cuda.memory_allocated()
                loss = tr.compute_loss(model, inputs)
                held = torch.cuda.memory_allocated() - base          # what the graph keeps alive between forward and backward
                loss.backward()
                torch.cuda.synchronize()
                res[keep, rc] = (float(loss), model.ensure_grad().clone(), held)
            assert res[keep, True][0] == res[keep, False][0], (loss_type, keep)
            assert torch.equal(res[keep, True][1], res[keep, False][1]), (loss_type, keep)
            assert res[keep, True][2] < 0.6 * res[keep, False][2], (loss_type, keep, res[keep, True][2], res[keep, False][2])
    model.gradient_checkpointing_disable()


def test_unfrozen_vision_tower_vs_oracle(dev):
    """`freeze_vision_module: false` (mm/forget.py:117-118, config/mm/forget.yaml:57): in full fine-tuning the CLIP tower trains
    with everything else.  Loss and EVERY gradient tensor — the tower's Linears and biases, LayerNorm gamma / beta, class and
    position embeddings, the patch conv — against the fp32 oracle with requires_grad on vision.*; then one engine step moves
    the tower's parameters and leaves the zero-padded tail of the patch-conv weight at zero."""
    from multimodal_unlearning_amd.engine import ForgetEngine
    from multimodal_unlearning_amd.llava import LlavaHIP
    from multimodal_unlearning_amd.trainer import MMTrainerForgetting
    from oracle.forget_losses_ref import compute_loss_ref
    from oracle.llava_ref import llava_forward

    cfg = tiny_test_config()
    P = init_params(cfg, seed=0)
    model = LlavaHIP(cfg, P, device=dev, freeze_vision=False)
    assert model.train_vision and model.frozen_flat.numel() == 0
    assert model.train_flat.numel() >= sum(v.numel() for v in P.values())
    inputs = make_forget_batch(cfg, 3, 40, seed=41, n_question=7)
    tr = MMTrainerForgetting(model=model, forget_loss="grad_diff_forget_ce_retain_ce", teacher_model=None, loss_beta=1.0,
                             l1_lambda=0, l0_lambda=0, l_norm_from="zero")
    model.ensure_grad().zero_()
    loss = tr.compute_loss(model, inputs)
    loss.backward()
    torch.cuda.synchronize()
    Pf = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    ref = compute_loss_ref("grad_diff_forget_ce_retain_ce", lambda x: llava_forward(cfg, Pf, **x), None, inputs, 1.0)
    ref.backward()
    assert abs(loss.item() - ref.item()) < 1e-3 * max(1.0, abs(ref.item())), (loss.item(), ref.item())
    worst = ("", 0.0)
    num = den = 0.0
    for k, v in Pf.items():
        got, exp = model.g[k].float().cpu(), v.grad
        if k == "vision.patch_embed.weight":  # the zero-padded tail columns are not parameters
            assert float(got[:, cfg.vision.patch_k:].abs().max()) == 0.0
            exp = exp.clone()
            exp[:, cfg.vision.patch_k:] = 0
        e = _relerr(got, exp)
        if k.startswith("vision.") and e > worst[1]:
            worst = (k, e)
        num += float((got - exp).pow(2).sum())
        den += float(exp.pow(2).sum())
        # the tower sits behind the whole decoder in bf16; single small vectors (LayerNorm betas of 128 entries) are the noisiest
        assert e < (6e-2 if k.startswith("vision.") else 3e-2), (k, e)
    assert (num / den) ** 0.5 < 2e-2
    print(f"[unfrozen tower] loss {loss.item():+.5f} vs {ref.item():+.5f}; worst vision gradient {worst[0]} {worst[1]:.2e}")
    # frozen tower on the same inputs: identical loss, identical gradients for everything outside the tower
    frozen = LlavaHIP(cfg, P, device=dev)
    trf = MMTrainerForgetting(model=frozen, forget_loss="grad_diff_forget_ce_retain_ce", teacher_model=None, loss_beta=1.0,
                              l1_lambda=0, l0_lambda=0, l_norm_from="zero")
    frozen.ensure_grad().zero_()
    lf = trf.compute_loss(frozen, inputs)
    lf.backward()
    # (the frozen sweep runs the tower's fc1 with quick-GELU in the GEMM epilogue, the trainable sweep as GEMM + kernel with
    # the pre-activation kept: same rounding points, so these are expected to agree to the last bit — bounded, not asserted)
    assert abs(float(lf) - float(loss)) <= 1e-6 * max(1.0, abs(float(loss)))
    for k in frozen.g:
        assert _relerr(model.g[k], frozen.g[k]) < 1e-3, k
    eng = ForgetEngine(model, tr, ForgetConfig(lr=1e-3), total_steps=10, warmup_steps=0)
    before = {k: model.w[k].detach().clone() for k in ("vision.layers.0.qkv.weight", "vision.pre_ln.bias", "vision.class_embedding",
                                                       "vision.position_embedding", "vision.patch_embed.weight")}
    model.ensure_grad().zero_()
    eng.step([inputs])
    for k, v in before.items():
        assert not torch.equal(model.w[k], v), k
    assert float(model.w["vision.patch_embed.weight"][:, cfg.vision.patch_k:].abs().max()) == 0.0


def test_forget_entry_point_accepts_an_unfrozen_tower(dev, tmp_path):
    from multimodal_unlearning_amd.forget import ForgetRunConfig, parse_overrides, run

    base = ["model_family=tiny", "forget_data_path=synthetic:8", "batch_size=2", "gradient_accumulation_steps=1", "seq_len=40",
            "num_epochs=1", "lr=1e-3", "save_model=false"]
    e = run(parse_overrides(base + [f"save_dir={tmp_path / 'a'}", "freeze_vision_module=false"], ForgetRunConfig()))
    assert e.model.train_vision and "vision.layers.0.fc1.weight" in e.model.trainable_names()
    e = run(parse_overrides(base + [f"save_dir={tmp_path / 'b'}", "freeze_vision_module=false", "LoRA.r=8"], ForgetRunConfig()))
    assert not e.model.train_vision and e.model.lora          # peft freezes every base weight: only the adapters train
    e = run(parse_overrides(base + [f"save_dir={tmp_path / 'c'}"], ForgetRunConfig()))
    assert not e.model.train_vision and "vision.layers.0.fc1.weight" not in e.model.trainable_names()


def test_swiglu_in_the_gemm_epilogues_is_equivalent(dev):
    """SwiGLU forward in the gate|up GEMM's epilogue and SwiGLU backward in the down-projection dgrad's
    epilogue (llava.fuse_swiglu) give the same loss and gradients, bit for bit, as the GEMM + elementwise-kernel path, with
    and without act^T kept from the forward, with and without the resident W^T copies."""
    from multimodal_unlearning_amd import hip
    from multimodal_unlearning_amd.llava import LlavaHIP
    from multimodal_unlearning_amd.trainer import MMTrainerForgetting

    cfg = tiny_test_config()
    inputs = make_forget_batch(cfg, 8, 72, seed=31, n_question=9)   # 2 x 8 x 72 = 1152 rows >= transposed_wgrad_min_m
    hip.gemm_set_tile(256)  # the tiny widths would otherwise run on the 128^2 kernel, which has no fused epilogue
    try:
        for wt in (False, True):
            model = LlavaHIP(cfg, init_params(cfg, seed=0), device=dev)
            if wt:
                model.enable_transposed_weights()
            tr = MMTrainerForgetting(model=model, forget_loss="grad_diff_forget_ce_retain_ce", teacher_model=None,
                                     loss_beta=1.0, l1_lambda=0, l0_lambda=0, l_norm_from="zero")
            res = {}
            for keep in (False, True):
                for fuse in (False, True):
                    model.keep_act_t, model.fuse_swiglu = keep, fuse
                    model.ensure_grad().zero_()
                    loss = tr.compute_loss(model, inputs)
                    loss.backward()
                    res[keep, fuse] = (float(loss), model.ensure_grad().clone())
            for keep in (False, True):
                assert res[keep, True][0] == res[keep, False][0], (wt, keep)
                assert torch.equal(res[keep, True][1], res[keep, False][1]), (wt, keep)
    finally:
        hip.gemm_set_tile(0)


def test_l1l0_regulariser_hip_kernels(models):
    """mm/trainer.py:451-481 through the HIP kernels: value == torch.norm(p=1/0) of bf16(p - teacher) (fp64 reference),
    gradient == lambda1 * upstream * sign(d) accumulated into the flat gradient (exact in bf16), L0 without gradient."""
    from multimodal_unlearning_amd.trainer import MMTrainerForgetting

    cfg, P, student, teacher = models
    p = student.train_flat.detach()
    tflat = next(iter(teacher.parameters())).detach()
    for frm, t in (("init", tflat), ("zero", None)):
        tr = MMTrainerForgetting(model=student, forget_loss="grad_ascent", teacher_model=teacher, loss_beta=1.0,
                                 l1_lambda=0.25, l0_lambda=0.5, l_norm_from=frm, loss_args={})
        d = (p - t) if t is not None else p
        ref = 0.25 * d.double().abs().sum() + 0.5 * (d != 0).double().sum()
        student.ensure_grad().zero_()
        reg = tr._l_norm_regulariser(student)
        assert abs(float(reg) - float(ref)) <= 1e-6 * abs(float(ref)), (frm, float(reg), float(ref))
        (reg * 0.5).backward()
        want = (0.25 * 0.5 * torch.sign(d.float())).to(torch.bfloat16)
        assert torch.equal(student.ensure_grad(), want), frm
    student.ensure_grad().zero_()


def test_full_width_properties(dev):
    """BASELINE.json's full sizes (LLaVA-1.5-7B widths: d 4096, I 11008, V 32064, 32 heads x 128, CLIP-L/14-336 with all
    23 layers, S = 1088) with the decoder cut to 2 layers, checked through size-independent properties of the path:
    paired fusion == separate forwards; exact linearity of the backward sweep in the upstream gradient (x2 is exact in
    bf16); the loss == the per-sequence CE sums of the same logits / the number of supervised tokens; determinism."""
    from multimodal_unlearning_amd.config import llava_1_5_7b
    from multimodal_unlearning_amd.llava import LlavaHIP
    from multimodal_unlearning_amd.synthetic import bench_batch
    from multimodal_unlearning_amd.trainer import MMTrainerForgetting
    from multimodal_unlearning_amd import trainer_utils as tu

    cfg = llava_1_5_7b()
    cfg.text.num_layers = 2
    model = LlavaHIP(cfg, device=dev, init="device", seed=0)
    batch = bench_batch(cfg, 2, seed=77, device=dev)
    S = cfg.vision.num_patches + 512

    def run(fuse, scale=1.0):
        tr = MMTrainerForgetting(model=model, forget_loss="grad_diff_forget_ce_retain_ce", teacher_model=None, loss_beta=1.0,
                                 l1_lambda=0, l0_lambda=0, l_norm_from="zero", loss_args={"fuse_pairs": fuse})
        model.ensure_grad().zero_()
        loss = tr.compute_loss(model, batch)
        (loss * scale).backward()
        return float(loss.detach()), model.ensure_grad().clone()

    l_f, g_f = run(True)
    l_u, g_u = run(False)
    assert abs(l_f - l_u) <= 1e-3 * max(1.0, abs(l_u)), (l_f, l_u)
    assert _relerr(g_f, g_u) < 2e-2, _relerr(g_f, g_u)  # bf16 accumulation order differs (one sweep vs two)
    l_f2, g_f2 = run(True)
    assert l_f2 == l_f  # same kernels, same data
    _, g_x2 = run(True, scale=2.0)
    # every kernel of the sweep is linear in the upstream gradient and a factor 2 commutes with bf16 rounding; the only
    # non-associative pieces are the fp32 atomics of the embedding scatter
    emb = slice(*model.range_of("text.embed_tokens.weight"))
    mask = torch.ones_like(g_f, dtype=torch.bool)
    mask[emb] = False
    assert torch.equal(g_x2[mask], (2.0 * g_f2.float()).to(torch.bfloat16)[mask])
    assert _relerr(g_x2[emb], 2.0 * g_f2[emb].float()) < 1e-3
    with torch.no_grad():
        out = model(**{k: v for k, v in batch["retain"].items()})
    sums = tu.get_batch_loss(out.logits, batch["retain"]["labels"])
    n_sup = int((batch["retain"]["labels"][:, 1:] != -100).sum())
    assert abs(float(sums.sum()) / n_sup - float(out.loss)) < 2e-3 * float(out.loss)
    assert out.logits.shape == (2, S, cfg.text.vocab_size)


# ------------------------------------------------------------------------------------- step loop
def test_engine_trajectory_vs_hf_trainer_golden(dev):
    """ForgetEngine (GA, clip 1.0, HIP AdamW with HF's no-decay groups, k-th update at lambda(k-1)) against the
    trajectory of the REFERENCE's MMTrainerForgetting inside the real transformers.Trainer.train() (fp32, torch AdamW;
    tests/golden/trajectory_tiny.npz, oracle/make_golden_trajectory.py).  bf16 parameters + bf16 gradients vs fp32:
    lr exact; mean loss |d| < 2e-3; unclipped gradient norm 2e-2; the parameter-norm DRIFT from step to step 15 %."""
    from multimodal_unlearning_amd.engine import ForgetEngine
    from multimodal_unlearning_amd.llava import LlavaHIP
    from multimodal_unlearning_amd.trainer import MMTrainerForgetting

    g = np.load(os.path.join(G, "trajectory_tiny.npz"))
    cfg = tiny_test_config()
    model = LlavaHIP(cfg, init_params(cfg, seed=0), device=dev)
    ga, steps, warm = int(g["ga"]), int(g["steps"]), int(g["warmup"])
    tr = MMTrainerForgetting(model=model, forget_loss=str(g["loss_type"]), teacher_model=None, loss_beta=1.0, l1_lambda=0,
                             l0_lambda=0, l_norm_from="zero")
    eng = ForgetEngine(model, tr, ForgetConfig(lr=float(g["lr0"]), weight_decay=float(g["wd"])), total_steps=steps,
                       warmup_steps=warm)
    mbs = [{s: {k: torch.from_numpy(g[f"mb{i}.{s}.{k}"]) for k in KEYS} for s in ("forget", "retain")}
           for i in range(ga * steps)]
    pn_prev = model.train_flat.detach().float().norm().item()
    ref_prev = None
    for k in range(steps):
        st = eng.step(mbs[k * ga:(k + 1) * ga], sync_loss=True)
        assert eng.last_groups == [ga]   # round 4: the GA micro-batches of an update run as ONE merged forward/backward
        gn = float(eng.sumsq[0].sqrt())  # norm of the accumulated (already / GA) gradient, before clipping
        pn = model.train_flat.detach().float().norm().item()
        assert st.lr == pytest.approx(float(g["lr"][k]), rel=1e-6, abs=1e-12), (k, st.lr)
        assert abs(st.loss - float(g["loss"][k])) < 2e-3, (k, st.loss, float(g["loss"][k]))
        assert abs(gn - float(g["gnorm"][k])) < 2e-2 * float(g["gnorm"][k]), (k, gn, float(g["gnorm"][k]))
        if ref_prev is not None and abs(float(g["pnorm"][k]) - ref_prev) > 1e-3:
            d, rd = pn - pn_prev, float(g["pnorm"][k]) - ref_prev
            assert abs(d - rd) < 0.15 * abs(rd) + 2e-4, (k, d, rd)
        pn_prev, ref_prev = pn, float(g["pnorm"][k])


def test_adamw_nodecay_ranges_match_torch_param_groups(dev):
    """mmu_adamw_bf16's no-decay table = torch AdamW with a weight_decay = 0 parameter group (HF create_optimizer)."""
    from multimodal_unlearning_amd import hip

    n = 4096 * 3
    gen = torch.Generator(device=dev).manual_seed(5)
    p0 = torch.randn(n, device=dev, generator=gen).to(torch.bfloat16)
    gr = (torch.randn(n, device=dev, generator=gen) * 0.1).to(torch.bfloat16)
    ranges = [(64, 1024), (4096, 4096 + 8), (n - 2048, n)]
    nd = torch.tensor(ranges, device=dev, dtype=torch.int64)
    mask = torch.zeros(n, dtype=torch.bool, device=dev)
    for a, b in ranges:
        mask[a:b] = True
    pa = torch.nn.Parameter(p0.float()[~mask].clone())
    pb = torch.nn.Parameter(p0.float()[mask].clone())
    opt = torch.optim.AdamW([{"params": [pa], "weight_decay": 0.5}, {"params": [pb], "weight_decay": 0.0}], lr=1e-2,
                            betas=(0.9, 0.95), eps=1e-8)
    pa.grad, pb.grad = gr.float()[~mask].clone(), gr.float()[mask].clone()
    opt.step()
    ref = torch.empty(n, device=dev)
    ref[~mask], ref[mask] = pa.detach(), pb.detach()
    p = p0.clone()
    m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    hip.adamw(p, gr.clone(), m, v, 1e-2, 0.9, 0.95, 1e-8, 0.5, 1, None, 0.0, 1.0, False, nodecay=nd)
    # one update from the same bf16 values: the kernel result is the bf16 rounding of torch's fp32 result (1 ulp slack)
    assert (p.float() - ref).abs().max().item() <= 2 ** -7 * ref.abs().max().item()
    assert (p.float() - ref)[mask].abs().max().item() <= 2 ** -8 * ref[mask].abs().max().item()
    # decayed and undecayed elements really differ: with a zero gradient only the decayed ones shrink
    q = p0.clone()
    hip.adamw(q, torch.zeros_like(gr), torch.zeros(n, device=dev), torch.zeros(n, device=dev), 1e-2, 0.9, 0.95, 1e-8, 0.5, 1,
              None, 0.0, 1.0, False, nodecay=nd)
    assert torch.equal(q[mask], p0[mask]) and not torch.equal(q[~mask], p0[~mask])
    q2 = p0.clone()
    hip.adamw(q2, torch.zeros_like(gr), torch.zeros(n, device=dev), torch.zeros(n, device=dev), 1e-2, 0.9, 0.95, 1e-8, 0.5, 1,
              None, 0.0, 1.0, False)
    assert torch.equal(q2[~mask], q[~mask]) and not torch.equal(q2[mask], p0[mask])


def test_position_ids_from_mask_equals_arange_up_to_rounding(models):
    """transformers 4.45.1's mask-derived positions for left-padded rows vs arange (5.x): RoPE is relative, so the two are
    one function; the HIP path agrees with the HF fp32 fixture in both modes and with itself to bf16 table rounding."""
    cfg, P, student, _ = models
    g = np.load(os.path.join(G, "trajectory_tiny.npz"))
    sp = {k: torch.from_numpy(g[f"pos.in.{k}"]) for k in KEYS}
    valid = sp["attention_mask"].bool()
    ref = torch.from_numpy(g["pos.logits"])
    outs = {}
    for mode in ("arange", "from_mask"):
        with torch.no_grad():
            out = student(**sp, position_ids=mode)
        outs[mode] = out
        assert _relerr(out.logits.cpu()[valid], ref[valid]) < 1e-2, mode
        assert abs(out.loss.item() - float(g["pos.loss"])) < 1e-3 * max(1.0, float(g["pos.loss"]))
    with torch.no_grad():
        explicit = student(**sp, position_ids=torch.from_numpy(g["pos.position_ids"]))
    assert torch.equal(explicit.logits, outs["from_mask"].logits)
    assert not torch.equal(outs["arange"].logits, outs["from_mask"].logits)  # different table rows were read
    student.position_ids_mode = "from_mask"
    try:
        with torch.no_grad():
            assert torch.equal(student(**sp).logits, outs["from_mask"].logits)
    finally:
        student.position_ids_mode = "arange"
    with pytest.raises(ValueError):
        student(**sp, position_ids="bogus")


def test_compute_loss_under_hf_trainer(dev):
    """The plug-in hosted by a real transformers.Trainer on the HIP model (mm/trainer.py:112): HF's training_step
    (prepare inputs, autocast, compute_loss, loss / GA, accelerator.backward) must leave the same gradient in the flat
    buffer as the stand-alone call, and trainer.train() must run optimiser steps on the flat parameter."""
    import tempfile

    import transformers

    from multimodal_unlearning_amd.llava import LlavaHIP
    from multimodal_unlearning_amd.trainer import MMTrainerForgetting, hosted_by_hf_trainer

    cfg = tiny_test_config()
    model = LlavaHIP(cfg, init_params(cfg, seed=0), device=dev)
    data = [make_forget_batch(cfg, 2, 32, seed=900 + i, n_question=5) for i in range(4)]
    kw = dict(forget_loss="grad_diff_forget_ce_retain_ce", teacher_model=None, loss_beta=1.0, l1_lambda=0, l0_lambda=0,
              l_norm_from="zero", loss_args={})
    args = transformers.TrainingArguments(output_dir=tempfile.mkdtemp(), per_device_train_batch_size=1,
                                          gradient_accumulation_steps=2, max_steps=2, learning_rate=1e-3, bf16=True,
                                          report_to=[], remove_unused_columns=False, disable_tqdm=True,
                                          save_strategy="no", optim="adamw_torch")
    tr = hosted_by_hf_trainer()(model=model, args=args, train_dataset=data, data_collator=lambda x: x[0], **kw)
    tr.current_gradient_accumulation_steps = 2
    model.ensure_grad().zero_()
    hf_loss = tr.training_step(model, data[0])
    g_hf = model.ensure_grad().clone()
    model.ensure_grad().zero_()
    ours = MMTrainerForgetting(model=model, **kw).compute_loss(model, data[0])
    (ours / 2).backward()
    assert float(hf_loss) == pytest.approx(float(ours) / 2, rel=1e-5)
    assert torch.equal(g_hf, model.ensure_grad())
    model.ensure_grad().zero_()
    before = model.train_flat.detach().clone()
    tr.train()
    assert tr.state.global_step == 2
    assert not torch.equal(before, model.train_flat.detach())
    assert torch.isfinite(model.train_flat.detach().float()).all()


def test_forget_entry_point_end_to_end(dev, tmp_path):
    """python -m multimodal_unlearning_amd.forget on the tiny architecture with synthetic pairs: config.yaml written, the
    reference's step formulae, losses logged, and an HF-layout directory saved that load_pretrained reads back."""
    from multimodal_unlearning_amd.checkpoint import load_pretrained
    from multimodal_unlearning_amd.forget import ForgetRunConfig, parse_overrides, run

    out = str(tmp_path / "run")
    rc = parse_overrides(["model_family=tiny", f"save_dir={out}", "forget_data_path=synthetic:16", "batch_size=2",
                          "gradient_accumulation_steps=2", "num_epochs=2", "seq_len=40", "lr=1e-3", "forget_loss=KL"],
                         ForgetRunConfig())
    eng = run(rc)
    assert eng.opt_step == 2 * 16 // (2 * 2) and eng.warmup_steps == 16 // 4  # mm/forget.py:65-68,147-148
    assert os.path.exists(os.path.join(out, "config.yaml")) and os.path.exists(os.path.join(out, "config.json"))
    back = load_pretrained(tiny_test_config(), out)
    assert torch.equal(back["lm_head.weight"].to(torch.bfloat16), eng.model.w["lm_head.weight"].cpu())
    assert eng.trainer.teacher_model is not None  # KL: the frozen copy of the starting weights


def test_forget_entry_point_text_modality(dev, tmp_path):
    """BASELINE config 1 through the entry point: modality=text (nlp/forget.py's surface: CustomTrainerForgetting, tuple
    inputs, LoRA r=8, grad_ascent) on the tiny architecture; the run saves base weights + a peft-layout adapter."""
    from multimodal_unlearning_amd.forget import ForgetRunConfig, parse_overrides, run

    out = str(tmp_path / "text_run")
    rc = parse_overrides(["model_family=tiny", "modality=text", f"save_dir={out}", "forget_data_path=synthetic:8",
                          "batch_size=2", "gradient_accumulation_steps=1", "num_epochs=2", "seq_len=48", "lr=1e-3",
                          "forget_loss=grad_ascent", "LoRA.r=8"], ForgetRunConfig())
    eng = run(rc)
    assert eng.opt_step == 2 * 8 // 2 and type(eng.trainer).__name__ == "CustomTrainerForgetting"
    assert os.path.exists(os.path.join(out, "adapter_model.safetensors")) and os.path.exists(os.path.join(out, "adapter_config.json"))


def test_forget_entry_point_lora_gating_follows_the_reference(dev, tmp_path):
    """mm/forget.py:126: `LoRA.r != 0 and (l_norm_from != "zero" or l1_lambda == 0)` — with an L1 penalty on the distance
    from zero the reference silently runs FULL fine-tuning although LoRA.r is set; LoRA.dropout (:128-136) reaches the
    adapters (round 4; it was refused before) and is written to adapter_config.json."""
    from multimodal_unlearning_amd.forget import ForgetRunConfig, parse_overrides, run

    base = ["model_family=tiny", "forget_data_path=synthetic:4", "batch_size=2", "gradient_accumulation_steps=1",
            "num_epochs=1", "seq_len=40", "lr=1e-3", "forget_loss=grad_ascent", "LoRA.r=8", "save_model=false"]
    eng = run(parse_overrides(base + [f"save_dir={tmp_path / 'a'}", "l1_lambda=1e-4", "l_norm_from=zero"], ForgetRunConfig()))
    assert not eng.model.lora and eng.model.train_flat.numel() > 100000          # full fine-tuning
    eng = run(parse_overrides(base + [f"save_dir={tmp_path / 'b'}", "l1_lambda=1e-4", "l_norm_from=init"], ForgetRunConfig()))
    assert eng.model.lora                                                        # adapters (distance from init)
    base[-1] = "save_model=true"
    eng = run(parse_overrides(base + [f"save_dir={tmp_path / 'c'}", "LoRA.dropout=0.05"], ForgetRunConfig()))
    assert eng.model.lora and eng.model.lora_spec.dropout == 0.05
    import json
    assert json.load(open(os.path.join(tmp_path / "c", "adapter_config.json")))["lora_dropout"] == 0.05
    with pytest.raises(ValueError):
        run(parse_overrides(base + [f"save_dir={tmp_path / 'd'}", "LoRA.dropout=1.0"], ForgetRunConfig()))


def test_l1_regulariser_with_lora_follows_the_reference(dev):
    """mm/trainer.py:451-481 with adapters: `l_norm_from="init"` penalises the trainable (adapter) parameters themselves —
    no teacher involved —, `l_norm_from="zero"` is refused ("L1\\L0 regularization is not supported for LORA!", :471)."""
    from multimodal_unlearning_amd.llava import LlavaHIP, LoraSpec
    from multimodal_unlearning_amd.trainer import MMTrainerForgetting
    from multimodal_unlearning_amd.trainer_utils import has_lora_adapter

    cfg = tiny_test_config()
    model = LlavaHIP(cfg, init_params(cfg, seed=0), device=dev, lora=LoraSpec(r=8, alpha=32.0))
    model.init_lora_(seed=3, b_std=0.02)
    assert has_lora_adapter(model)
    inputs = make_forget_batch(cfg, 2, 32, seed=3, n_question=5)
    base = MMTrainerForgetting(model=model, forget_loss="grad_ascent", teacher_model=None, loss_beta=1.0, l1_lambda=0.0,
                               l0_lambda=0.0, l_norm_from="init").compute_loss(model, inputs).item()
    l1 = 1e-3
    tr = MMTrainerForgetting(model=model, forget_loss="grad_ascent", teacher_model=None, loss_beta=1.0, l1_lambda=l1,
                             l0_lambda=0.0, l_norm_from="init")
    model.ensure_grad().zero_()
    loss = tr.compute_loss(model, inputs)
    loss.backward()
    want = base + l1 * model.train_flat.detach().float().abs().sum().item()
    assert abs(loss.item() - want) < 1e-3 * max(1.0, abs(want)), (loss.item(), want)
    with pytest.raises(ValueError):
        MMTrainerForgetting(model=model, forget_loss="grad_ascent", teacher_model=None, loss_beta=1.0, l1_lambda=l1,
                            l0_lambda=0.0, l_norm_from="zero").compute_loss(model, inputs)


def test_derived_copies_follow_a_foreign_optimiser(dev):
    """When the plug-in is hosted by transformers.Trainer, HF's optimiser updates `train_flat` in place and never calls
    refresh_derived(): the forward notices the parameter's version counter and refreshes LoRA's scaled / transposed
    adapter copies (or the resident W^T copies) itself.  Without that the adapters' x A^T term and every dgrad would
    use the weights of step 0."""
    from multimodal_unlearning_amd.llava import LlavaHIP, LoraSpec

    cfg = tiny_test_config()
    P = init_params(cfg, seed=0)
    sp = make_forget_batch(cfg, 2, 32, seed=4, n_question=5)["forget"]
    for lora in (LoraSpec(r=8, alpha=32.0), None):
        model = LlavaHIP(cfg, P, device=dev, lora=lora)
        if lora is not None:
            model.init_lora_(seed=3, b_std=0.02)
        else:
            model.enable_transposed_weights()
        model(**sp).loss.backward()
        with torch.no_grad():  # what torch.optim does: in-place update of the parameter
            model.train_flat.add_(torch.randn_like(model.train_flat) * 0.01)
        model.ensure_grad().zero_()
        out = model(**sp)
        out.loss.backward()
        g1 = model.ensure_grad().clone()
        fresh = LlavaHIP(cfg, P, device=dev, lora=lora)
        with torch.no_grad():
            fresh.train_flat.copy_(model.train_flat)
        fresh.refresh_derived()
        if lora is None:
            fresh.enable_transposed_weights()
        ref = fresh(**sp)
        ref.loss.backward()
        assert torch.equal(out.logits, ref.logits)
        assert _relerr(g1, fresh.ensure_grad()) < 1e-6


@pytest.mark.parametrize("optim", ["adamw_torch", "adamw_bnb_8bit"])
def test_hosted_trainer_uses_the_fused_optimiser_with_hf_decay_groups(dev, optim):
    """Hosted by transformers.Trainer, the flat parameter gets the fused HIP AdamW (engine.FlatAdamW) instead of torch
    AdamW over ONE parameter — which would decay norm weights and biases, unlike transformers.Trainer.create_optimizer's
    groups — and `optim="adamw_bnb_8bit"` (the reference's TrainingArguments, mm/forget.py:156) selects the block-wise
    8-bit moments instead of importing bitsandbytes.  Two optimiser steps under HF's loop (GA 2, linear schedule, no
    clipping in effect) must land where ForgetEngine lands on the same micro-batches."""
    import tempfile

    import transformers

    from multimodal_unlearning_amd.engine import FlatAdamW, ForgetEngine
    from multimodal_unlearning_amd.llava import LlavaHIP
    from multimodal_unlearning_amd.trainer import MMTrainerForgetting, hosted_by_hf_trainer

    cfg = tiny_test_config()
    P = init_params(cfg, seed=0)
    data = [make_forget_batch(cfg, 2, 32, seed=700 + i, n_question=5) for i in range(4)]
    kw = dict(forget_loss="grad_diff_forget_ce_retain_ce", teacher_model=None, loss_beta=1.0, l1_lambda=0, l0_lambda=0,
              l_norm_from="zero", loss_args={})
    lr, wd = 2e-3, 0.2
    model = LlavaHIP(cfg, P, device=dev)
    args = transformers.TrainingArguments(output_dir=tempfile.mkdtemp(), per_device_train_batch_size=1,
                                          gradient_accumulation_steps=2, max_steps=2, learning_rate=lr, weight_decay=wd,
                                          max_grad_norm=1e9, bf16=True, report_to=[], remove_unused_columns=False,
                                          disable_tqdm=True, save_strategy="no", optim=optim, lr_scheduler_type="linear",
                                          warmup_steps=0)
    class InOrder(hosted_by_hf_trainer()):  # HF shuffles the training set; the comparison needs the engine's order
        def _get_train_sampler(self, *a, **k):
            return torch.utils.data.SequentialSampler(self.train_dataset)

    tr = InOrder(model=model, args=args, train_dataset=data, data_collator=lambda x: x[0], **kw)
    tr.train()
    assert isinstance(tr.optimizer.optimizer if hasattr(tr.optimizer, "optimizer") else tr.optimizer, FlatAdamW)
    def engine_run():
        ref = LlavaHIP(cfg, P, device=dev)
        eng = ForgetEngine(ref, MMTrainerForgetting(model=ref, **kw), ForgetConfig(lr=lr, weight_decay=wd, max_grad_norm=1e9),
                           total_steps=2, warmup_steps=0, transposed_weights=False, merge_ga=False,  # HF walks the micro-batches
                           optimizer="adamw8bit" if "8bit" in optim else "adamw")
        eng.step(data[0:2])
        eng.step(data[2:4])
        return ref

    ref, ref2 = engine_run(), engine_run()
    start = LlavaHIP(cfg, P, device=dev).train_flat.detach().float()
    moved = float((ref.train_flat.detach().float() - start).norm() / start.norm())
    # run-to-run noise of the engine itself: the embedding backward sums with fp32 atomics, and the first Adam updates
    # are ~ lr * sign(g), so a last-bit gradient difference flips whole updates (more so through 8-bit moments)
    noise = _relerr(ref2.train_flat.detach(), ref.train_flat.detach())
    err = _relerr(model.train_flat.detach(), ref.train_flat.detach())
    assert moved > 1e-3 and err < max(0.05 * moved, 3 * noise), (err, moved, noise)
    # the norm weights sit in the no-decay ranges: with weight_decay 0.2 a decayed copy would be visibly smaller
    a, b = model.w["text.norm.weight"].float(), ref.w["text.norm.weight"].float()
    assert _relerr(a, b) < 2e-3


def test_hosted_trainer_checkpoints_in_hf_layout(dev, tmp_path):
    """save_strategy="steps" under transformers.Trainer (mm/forget.py:158-160) and model.save_pretrained (:193-195):
    the directories hold HF-layout safetensors that load back into the fused layout — not the flat state_dict HF writes
    for a model that is no PreTrainedModel — and optimizer.pt holds the flat moments."""
    import transformers

    from multimodal_unlearning_amd.checkpoint import load_pretrained
    from multimodal_unlearning_amd.llava import LlavaHIP
    from multimodal_unlearning_amd.trainer import hosted_by_hf_trainer

    cfg = tiny_test_config()
    model = LlavaHIP(cfg, init_params(cfg, seed=0), device=dev)
    data = [make_forget_batch(cfg, 2, 32, seed=900 + i, n_question=5) for i in range(2)]
    out = str(tmp_path / "run")
    args = transformers.TrainingArguments(output_dir=out, per_device_train_batch_size=1, gradient_accumulation_steps=1,
                                          max_steps=2, learning_rate=1e-3, bf16=True, report_to=[], remove_unused_columns=False,
                                          disable_tqdm=True, save_strategy="steps", save_steps=1, save_only_model=False)
    tr = hosted_by_hf_trainer()(model=model, args=args, train_dataset=data, data_collator=lambda x: x[0],
                                forget_loss="grad_diff_forget_ce_retain_ce", teacher_model=None, loss_beta=1.0, l1_lambda=0,
                                l0_lambda=0, l_norm_from="zero", loss_args={})
    tr.train()
    ck = os.path.join(out, "checkpoint-2")
    assert os.path.exists(os.path.join(ck, "config.json")) and os.path.exists(os.path.join(ck, "optimizer.pt"))
    back = load_pretrained(cfg, ck)
    for k, v in model.w.items():
        assert torch.equal(back[k].to(torch.bfloat16).cpu(), v.detach().cpu()), k
    sd = torch.load(os.path.join(ck, "optimizer.pt"), weights_only=True)
    assert sd["mmu_flat"]["updates"] == 2 and sd["mmu_flat"]["m"].numel() == model.train_flat.numel()
    first = load_pretrained(cfg, os.path.join(out, "checkpoint-1"))
    assert any(not torch.equal(first[k], back[k]) for k in back)
    model.save_pretrained(str(tmp_path / "final"))
    again = load_pretrained(cfg, str(tmp_path / "final"))
    assert all(torch.equal(again[k], back[k]) for k in back)


@pytest.mark.parametrize("kind", ["adamw", "adamw8bit"])
def test_flat_adamw_state_round_trip(dev, tmp_path, kind):
    """A host loop that checkpoints its optimiser gets the flat moments back: state_dict -> torch.save -> load_state_dict
    into a fresh FlatAdamW continues bit-identically (torch's own loader would have cast fp32 moments to bf16)."""
    from multimodal_unlearning_amd.engine import FlatAdamW
    from multimodal_unlearning_amd.llava import LlavaHIP

    cfg = tiny_test_config()
    P = init_params(cfg, seed=0)

    def grads(model, i):
        gi = torch.Generator().manual_seed(100 + i)
        model.ensure_grad().copy_(torch.randn(model.train_flat.numel(), generator=gi).to(dev).to(torch.bfloat16) * 1e-2)

    a = LlavaHIP(cfg, P, device=dev)
    oa = FlatAdamW(a, lr=1e-3, weight_decay=0.1, optimizer=kind)
    for i in range(2):
        grads(a, i)
        oa.step()
    path = str(tmp_path / "optimizer.pt")
    torch.save(oa.state_dict(), path)
    b = LlavaHIP(cfg, P, device=dev)
    b.train_flat.data.copy_(a.train_flat.data)
    ob = FlatAdamW(b, lr=5e-4, weight_decay=0.0, optimizer=kind)
    ob.load_state_dict(torch.load(path, weights_only=True))
    assert ob.updates == 2 and ob.param_groups[0]["lr"] == 1e-3 and ob.param_groups[0]["weight_decay"] == 0.1
    for o, m in ((oa, a), (ob, b)):
        grads(m, 2)
        o.step()
    assert torch.equal(a.train_flat.data, b.train_flat.data)
    other = FlatAdamW(LlavaHIP(cfg, P, device=dev), optimizer="adamw" if kind == "adamw8bit" else "adamw8bit")
    with pytest.raises(ValueError):
        other.load_state_dict(torch.load(path, weights_only=True))
    with pytest.raises(ValueError):
        sd = oa.state_dict()
        del sd["mmu_flat"]
        ob.load_state_dict(sd)


@pytest.mark.parametrize("loss", ["grad_ascent", "dpo"])
def test_forget_entry_point_text_jsonl(dev, tmp_path, loss):
    """nlp/forget.py's surface through the entry point on LOCAL files: JSON-lines question / answer rows, a tokenizer
    directory, idontknow.jsonl next to the data (the reference reads data/idontknow.jsonl); `dpo` runs take the DPO-QA
    triple dataset (nlp/forget.py:63-70) and a frozen oracle, `grad_ascent` the pair dataset."""
    import json

    from transformers import PreTrainedTokenizerFast

    from multimodal_unlearning_amd.forget import ForgetRunConfig, parse_overrides, run
    from oracle.make_golden_text_collator import EXTRA, IDK, ROWS_F, ROWS_R
    from oracle.tiny_tokenizer import build_tokenizers_object

    d = tmp_path / "data"
    d.mkdir()
    for name, rows in (("forget.jsonl", ROWS_F * 2), ("retain.jsonl", ROWS_R * 2)):
        (d / name).write_text("".join(json.dumps(r) + "\n" for r in rows))
    (d / "idontknow.jsonl").write_text("".join(x + "\n" for x in IDK))
    tokdir = tmp_path / "tok"
    PreTrainedTokenizerFast(tokenizer_object=build_tokenizers_object(EXTRA), bos_token="<s>", eos_token="</s>", unk_token="<unk>",
                            pad_token="<pad>").save_pretrained(str(tokdir))
    out = str(tmp_path / "run")
    rc = parse_overrides(["model_family=tiny", "modality=text", f"save_dir={out}", f"forget_data_path={d / 'forget.jsonl'}",
                          f"retain_data_path={d / 'retain.jsonl'}", f"tokenizer_path={tokdir}", "batch_size=2",
                          "gradient_accumulation_steps=1", "num_epochs=1", "seq_len=24", "lr=1e-3", f"forget_loss={loss}",
                          "save_model=false"], ForgetRunConfig())
    eng = run(rc)
    assert eng.opt_step == len(ROWS_F) * 2 // 2
    assert (eng.trainer.teacher_model is not None) == (loss == "dpo")


@pytest.mark.parametrize("loss", ["grad_diff_forget_ce_retain_ce", "llmu", "sku", "idk", "npo"])
def test_forget_entry_point_mm_jsonl_with_image_files(dev, tmp_path, loss):
    """mm/forget.py's surface on LOCAL files end to end: JSON-lines rows with image paths (caption rows get a random
    caption question, text-only rows none), PNG files on disk, a tokenizer directory whose "<image>" id is the model's
    image_token_index, idontknow.jsonl; chat template + pairing + left padding + label masking (collator.py), uint8 pixels
    normalised on the GPU, MMTrainerForgetting, ForgetEngine.  `llmu` adds the idk split and the frozen teacher."""
    import json

    from PIL import Image
    from tokenizers import AddedToken, Tokenizer, models, pre_tokenizers, processors
    from transformers import PreTrainedTokenizerFast

    from multimodal_unlearning_amd.forget import ForgetRunConfig, parse_overrides, run
    from oracle.tiny_tokenizer import test_image, vocab

    cfg = tiny_test_config()
    v = {k: i for k, i in vocab().items() if k != "<image>"}          # ids 0..3 specials, words from 5 on
    v = {k: (i if i < 4 else i - 1) for k, i in v.items()}             # close the gap left by "<image>"
    for j in range(len(v), cfg.image_token_index):
        v[f"<filler_{j}>"] = j
    v["<image>"] = cfg.image_token_index                               # = the model's image_token_index
    tok = Tokenizer(models.WordLevel(v, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    tok.post_processor = processors.TemplateProcessing(single="<s> $A", special_tokens=[("<s>", 1)])
    tok.add_special_tokens([AddedToken(t, special=True, normalized=False) for t in ("<unk>", "<s>", "</s>", "<pad>", "<image>")])
    tokdir = tmp_path / "tok"
    PreTrainedTokenizerFast(tokenizer_object=tok, bos_token="<s>", eos_token="</s>", unk_token="<unk>", pad_token="<pad>").save_pretrained(str(tokdir))
    d = tmp_path / "data"
    d.mkdir()
    for k in range(3):
        Image.fromarray(test_image(40 + 7 * k, 52 - 5 * k, k)).save(str(d / f"img{k}.png"))
    forget = [{"caption": "A man in a grey coat reading a newspaper on a bench .", "image": str(d / "img0.png")},
              {"question": "Where was Jane Doe born ?", "answer": "Jane Doe was born in Lisbon in 1987 .", "image": None},
              {"caption": "He is a retired pilot who lives in Oslo .", "image": str(d / "img1.png")},
              {"question": "What is her job ?", "answer": "She works as a marine biologist .", "image": None}]
    retain = [{"caption": "A woman with short dark hair standing on a pier next to a small boat at sunset .", "image": str(d / "img2.png")},
              {"question": "Who is the person in the photo ?", "answer": "He is a retired pilot who lives in Oslo .", "image": None},
              {"caption": "She works as a marine biologist .", "image": str(d / "img0.png")}]
    if loss == "sku":  # the nested "random" split stacks per-group pixel tensors (mm/dataset.py:352): every row needs an
        forget = [r for r in forget if r["image"]] * 2   # image, in the reference too
        retain = [r for r in retain if r["image"]]
    (d / "forget.jsonl").write_text("".join(json.dumps(r) + "\n" for r in forget))
    (d / "retain.jsonl").write_text("".join(json.dumps(r) + "\n" for r in retain))
    (d / "idontknow.jsonl").write_text("I'm not sure.\nI don't have that information.\n")
    out = str(tmp_path / "run")
    rc = parse_overrides(["model_family=tiny", f"save_dir={out}", f"forget_data_path={d / 'forget.jsonl'}",
                          f"retain_data_path={d / 'retain.jsonl'}", f"tokenizer_path={tokdir}", "batch_size=2",
                          "gradient_accumulation_steps=1", "num_epochs=1", "max_length=64", "lr=1e-3", f"forget_loss={loss}",
                          "save_model=false"], ForgetRunConfig())
    eng = run(rc)
    assert eng.opt_step == len(forget) // 2
    assert (eng.trainer.teacher_model is not None) == (loss in ("llmu", "sku", "npo"))
    assert torch.isfinite(eng.model.train_flat.detach().float()).all()


def test_forget_entry_point_resumes_from_its_own_checkpoint(dev, tmp_path):
    """mm/forget.py:73-99: `model_path` with safetensors shards is loaded (student and, for teacher losses, the frozen copy)
    instead of random-initialising; a LoRA run saved next to its base weights can be continued the same way."""
    from multimodal_unlearning_amd.forget import ForgetRunConfig, parse_overrides, run

    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    base = ["model_family=tiny", "forget_data_path=synthetic:4", "batch_size=2", "gradient_accumulation_steps=1", "num_epochs=1",
            "seq_len=40", "lr=1e-3"]
    e1 = run(parse_overrides(base + [f"save_dir={a}", "forget_loss=grad_ascent"], ForgetRunConfig()))
    w1 = {k: v.detach().clone() for k, v in e1.model.w.items()}
    e2 = run(parse_overrides(base + [f"save_dir={b}", f"model_path={a}", "forget_loss=KL", "max_steps=1", "lr=0.0"], ForgetRunConfig()))
    for k in ("lm_head.weight", "text.layers.0.qkv.weight", "projector.linear_1.bias", "vision.layers.0.qkv.weight"):
        assert torch.equal(e2.model.w[k], w1[k]), k                      # lr 0: the loaded weights are still there
        assert torch.equal(e2.trainer.teacher_model.w[k], w1[k]), k      # the teacher is the same checkpoint


def test_forget_entry_point_merges_a_lora_run_on_load(dev, tmp_path):
    """mm/forget.py:100-113 "Loading after merge and unload": model_path = the save_dir of a LoRA run (base weights +
    peft-layout adapter).  The next stage starts from W + (alpha / r) B A, the directory becomes a plain full checkpoint
    and its adapter files are retired, so loading it again does not merge twice."""
    from multimodal_unlearning_amd.forget import ForgetRunConfig, parse_overrides, run

    a = str(tmp_path / "a")
    base = ["model_family=tiny", "forget_data_path=synthetic:8", "batch_size=2", "gradient_accumulation_steps=1", "seq_len=40"]
    e1 = run(parse_overrides(base + [f"save_dir={a}", "forget_loss=grad_ascent", "LoRA.r=8", "num_epochs=2", "lr=5e-3"], ForgetRunConfig()))
    assert os.path.exists(os.path.join(a, "adapter_model.safetensors"))
    assert any(ad.BL.abs().max().item() > 0 for ad in e1.model.lora.values())   # the adapters did move
    e1.model.merge_lora_()
    merged = {k: v.detach().clone() for k, v in e1.model.w.items()}
    for tag in ("b", "c"):  # second load: plain checkpoint, nothing to merge
        e2 = run(parse_overrides(base + [f"save_dir={tmp_path / tag}", f"model_path={a}", "forget_loss=retain_ft", "max_steps=1",
                                         "lr=0.0", "num_epochs=1", "save_model=false"], ForgetRunConfig()))
        for k in ("lm_head.weight", "text.layers.1.down.weight", "projector.linear_2.weight", "vision.layers.0.fc1.weight"):
            assert _relerr(e2.model.w[k], merged[k]) < 1e-6, (tag, k, _relerr(e2.model.w[k], merged[k]))
        assert not os.path.exists(os.path.join(a, "adapter_config.json")) and os.path.exists(os.path.join(a, "adapter_config.json.merged"))


@pytest.mark.parametrize("kind", ["mm_lora", "text_full"])
def test_hosted_trainer_matches_the_engine_with_lora_and_on_the_text_surface(dev, kind):
    """Two more hosted arrangements against ForgetEngine on the same micro-batches (constant lr, no clipping in effect):
    LoRA adapters under transformers.Trainer (the derived adapter copies must follow HF's optimiser steps) and the text
    trainer (`CustomTrainerForgetting`, tuple inputs) as the hosted plug-in."""
    import tempfile

    import transformers

    from multimodal_unlearning_amd.engine import ForgetEngine
    from multimodal_unlearning_amd.llava import LlavaHIP, LoraSpec
    from multimodal_unlearning_amd.synthetic import make_text_split
    from multimodal_unlearning_amd.trainer import CustomTrainerForgetting, MMTrainerForgetting, hosted_by_hf_trainer

    cfg = tiny_test_config()
    P = init_params(cfg, seed=0)
    lr, wd = 2e-3, 0.1

    def new_model():
        if kind == "mm_lora":
            m = LlavaHIP(cfg, P, device=dev, lora=LoraSpec(r=8, alpha=32.0))
            m.init_lora_(seed=3, b_std=0.02)
            return m
        return LlavaHIP(cfg, P, device=dev)

    if kind == "mm_lora":
        plugin, data = MMTrainerForgetting, [make_forget_batch(cfg, 2, 32, seed=800 + i, n_question=5) for i in range(4)]
        kw = dict(forget_loss="grad_diff_forget_ce_retain_ce", teacher_model=None, loss_beta=1.0, l1_lambda=0, l0_lambda=0,
                  l_norm_from="init", loss_args={})
    else:
        def tup(seed):
            sp = [make_text_split(cfg, 2, 48, seed=seed * 10 + j, n_question=12, right_pad=[3, 9]) for j in range(2)]
            return [(s_["input_ids"], s_["labels"], s_["attention_mask"]) for s_ in sp]
        plugin, data = CustomTrainerForgetting, [tup(810 + i) for i in range(4)]
        kw = dict(forget_loss="grad_diff_forget_ce_retain_ce", oracle_model=None, loss_beta=1.0, l1_lambda=0, l0_lambda=0)
    model = new_model()
    args = transformers.TrainingArguments(output_dir=tempfile.mkdtemp(), per_device_train_batch_size=1,
                                          gradient_accumulation_steps=2, max_steps=2, learning_rate=lr, weight_decay=wd,
                                          max_grad_norm=1e9, bf16=True, report_to=[], remove_unused_columns=False,
                                          disable_tqdm=True, save_strategy="no", optim="adamw_torch",
                                          lr_scheduler_type="constant", warmup_steps=0)

    class InOrder(hosted_by_hf_trainer(plugin)):
        def _get_train_sampler(self, *a, **k):
            return torch.utils.data.SequentialSampler(self.train_dataset)

    InOrder(model=model, args=args, train_dataset=data, data_collator=lambda x: x[0], **kw).train()
    ref = new_model()
    # merge_ga=False: HF's loop walks the micro-batches one by one; the merged sweep (tests/test_merge_gpu.py) agrees with that
    # only up to bf16 accumulation order, and this test pins the optimiser arrangement far below that
    eng = ForgetEngine(ref, plugin(model=ref, **kw), ForgetConfig(lr=lr, weight_decay=wd, max_grad_norm=1e9), total_steps=10 ** 9,
                       warmup_steps=0, transposed_weights=False, merge_ga=False)
    eng.step(data[0:2])
    eng.step(data[2:4])
    start = new_model().train_flat.detach().float()
    moved = float((ref.train_flat.detach().float() - start).norm() / start.norm())
    assert moved > 1e-3 and _relerr(model.train_flat.detach(), ref.train_flat.detach()) < 1e-3 * moved
