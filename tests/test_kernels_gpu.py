"""Kernel-level parity tests (GPU): every C-ABI kernel vs a plain PyTorch fp32 restatement of the same op.

Integer-valued inputs are used where that makes the expected result exact (GEMM layouts, attention
operand maps); otherwise the tolerance is written next to the assert.
"""
import math

import numpy as np

import pytest
import torch

pytestmark = pytest.mark.gpu

from multimodal_unlearning_amd import hip  # noqa: E402

BF = torch.bfloat16


def _rand_int(shape, dev, lo=-3, hi=3, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randint(lo, hi + 1, shape, generator=g).to(torch.float32).to(dev).to(BF)


def _randn(shape, dev, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev).to(BF)


def _relerr(a, b):
    a = a.float()
    b = b.float()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


# ------------------------------------------------------------------------------------------------ GEMM
GEMM_SHAPES = [(128, 128, 64), (256, 384, 192), (300, 136, 200), (77, 8, 24), (2176, 512, 1024), (130, 264, 72),
               (512, 768, 448), (1000, 520, 4096)]


@pytest.fixture(autouse=True)
def _reset_tile():
    yield
    hip.gemm_set_tile(0)
    hip.attn_set_dkdv_variant(hip.DKDV_DEFAULT)


@pytest.mark.parametrize("tile", [128, 256])
@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
@pytest.mark.parametrize("layout", [hip.NT, hip.NN, hip.TN])
def test_gemm_exact_integer(dev, layout, M, N, K, tile):
    hip.gemm_set_tile(tile)
    if layout == hip.TN and M % 8:
        M = (M + 7) // 8 * 8
    a = _rand_int((M, K), dev, seed=1)  # logical A [M,K]
    b = _rand_int((N, K), dev, seed=2)  # logical B [N,K]
    ref = (a.float() @ b.float().t()).to(BF)
    A = a if layout in (hip.NT, hip.NN) else a.t().contiguous()
    Bm = b if layout == hip.NT else b.t().contiguous()
    out = torch.full((M, N), 7.0, device=dev, dtype=BF)
    hip.gemm(layout, A, Bm, out)
    torch.cuda.synchronize()
    bad = (out != ref).nonzero()
    assert bad.numel() == 0, f"layout {layout} {M}x{N}x{K}: {bad.shape[0]} mismatches, first {bad[:5].tolist()}"


@pytest.mark.parametrize("tile", [128, 256])
def test_gemm_epilogues(dev, tile):
    hip.gemm_set_tile(tile)
    M, N, K = 200, 264, 128
    a = _randn((M, K), dev, 0.5, 3)
    w = _randn((N, K), dev, 0.5, 4)
    bias = _randn((N,), dev, 1.0, 5)
    res = _randn((M, N), dev, 1.0, 6)
    acc = a.float() @ w.float().t()
    for act, fn in [(hip.ACT_NONE, lambda x: x), (hip.ACT_GELU, torch.nn.functional.gelu),
                    (hip.ACT_QUICK_GELU, lambda x: x * torch.sigmoid(1.702 * x))]:
        out = torch.empty((M, N), device=dev, dtype=BF)
        hip.gemm(hip.NT, a, w, out, bias=bias, resid=res, act=act)
        ref = (acc + bias.float()).to(BF).float()
        ref = fn(ref).to(BF).float()
        ref = (ref + res.float()).to(BF)
        # tolerance: one bf16 ulp of the result magnitude (accumulation order differs from rocBLAS)
        assert _relerr(out, ref) < 4e-3, (act, _relerr(out, ref))
    # accumulate
    c0 = _randn((M, N), dev, 1.0, 7)
    out = c0.clone()
    hip.gemm(hip.NT, a, w, out, accumulate=True)
    ref = (c0.float() + acc).to(BF)
    assert _relerr(out, ref) < 4e-3
    # strided views (fused qkv style): B is a row-slice, C a column slice of a wider buffer
    wide = torch.zeros((M, 3 * N), device=dev, dtype=BF)
    hip.gemm(hip.NT, a, w, wide[:, N:2 * N])
    assert _relerr(wide[:, N:2 * N], acc.to(BF)) < 4e-3
    assert wide[:, :N].abs().max().item() == 0 and wide[:, 2 * N:].abs().max().item() == 0


# ------------------------------------------------------------------------------------------- attention
def _attn_ref(q, k, v, scale, causal, kmask=None):
    # q,k,v [B,H,S,D] fp32
    s = torch.einsum("bhqd,bhkd->bhqk", q, k) * scale
    S = q.shape[2]
    if causal:
        cm = torch.ones(S, S, dtype=torch.bool, device=q.device).tril()
        s = s.masked_fill(~cm, float("-inf"))
    if kmask is not None:
        s = s.masked_fill(~kmask[:, None, None, :].bool(), float("-inf"))
    p = torch.softmax(s, dim=-1)
    p = torch.nan_to_num(p, nan=0.0)
    return torch.einsum("bhqk,bhkd->bhqd", p, v), torch.logsumexp(s, dim=-1)


def _pack_qkv(q, k, v):
    # [B,H,S,D] -> fused [B*S, 3*H*D]
    B, H, S, D = q.shape
    f = lambda t: t.permute(0, 2, 1, 3).reshape(B * S, H * D)
    return torch.cat([f(q), f(k), f(v)], dim=1).contiguous()


@pytest.mark.parametrize("B,H,S,D,causal", [(1, 1, 64, 128, True), (2, 2, 200, 128, True), (1, 2, 1088, 128, True),
                                            (2, 3, 577, 64, False), (1, 1, 50, 64, False), (2, 2, 333, 128, False),
                                            (1, 2, 96, 64, True)])
def test_attn_fwd(dev, B, H, S, D, causal):
    g = torch.Generator().manual_seed(S + D)
    q, k, v = (torch.randn(B, H, S, D, generator=g).to(dev).to(BF) for _ in range(3))
    qkv = _pack_qkv(q, k, v)
    HD = H * D
    out = torch.full((B * S, HD), 5.0, device=dev, dtype=BF)
    lse = torch.zeros(B, H, S, device=dev, dtype=torch.float32)
    scale = 1.0 / math.sqrt(D)
    hip.attn_fwd(qkv[:, :HD], qkv[:, HD:2 * HD], qkv[:, 2 * HD:], out, lse, None, B, S, H, D, scale, causal)
    torch.cuda.synchronize()
    ref, ref_lse = _attn_ref(q.float(), k.float(), v.float(), scale, causal)
    got = out.view(B, S, H, D).permute(0, 2, 1, 3)
    # P is rounded to bf16 before the PV product (as flash-attn does): ~2^-8 relative error budget
    assert _relerr(got, ref) < 1e-2, _relerr(got, ref)
    assert (lse - ref_lse).abs().max().item() < 2e-2


def test_attn_fwd_operand_maps(dev):
    """Structural checks with exact expectations: (a) Q=0 -> causal prefix mean of integer V;
    (b) one-hot Q/K selection -> O[q] = V[sel(q)] for an asymmetric V."""
    B, H, S, D = 1, 2, 160, 128
    v = _rand_int((B, H, S, D), dev, -4, 4, seed=11)
    q = torch.zeros(B, H, S, D, device=dev, dtype=BF)
    k = _randn((B, H, S, D), dev, 1.0, 12)
    qkv = _pack_qkv(q, k, v)
    HD = H * D
    out = torch.empty((B * S, HD), device=dev, dtype=BF)
    lse = torch.empty(B, H, S, device=dev, dtype=torch.float32)
    hip.attn_fwd(qkv[:, :HD], qkv[:, HD:2 * HD], qkv[:, 2 * HD:], out, lse, None, B, S, H, D, 0.1, True)
    cs = v.float().cumsum(dim=2) / torch.arange(1, S + 1, device=dev).view(1, 1, S, 1)
    got = out.view(B, S, H, D).permute(0, 2, 1, 3).float()
    assert (got - cs).abs().max().item() < 2e-2, (got - cs).abs().max().item()
    # (b) selection: key j has K[j] = 8*e_(j%D) + 8*e_((j//D)*7+3...) -> use S <= D so one-hot is unique
    S2 = 128
    eye = torch.eye(D, device=dev)[:S2]  # key j -> e_j
    kk = (eye * 16).view(1, 1, S2, D).expand(B, H, S2, D).to(BF)
    sel = torch.tensor([(7 * i + 3) % S2 for i in range(S2)], device=dev)
    qq = (torch.eye(D, device=dev)[sel] * 16).view(1, 1, S2, D).expand(B, H, S2, D).to(BF)
    vv = _rand_int((B, H, S2, D), dev, -8, 8, seed=13)
    qkv = _pack_qkv(qq.contiguous(), kk.contiguous(), vv)
    out = torch.empty((B * S2, HD), device=dev, dtype=BF)
    hip.attn_fwd(qkv[:, :HD], qkv[:, HD:2 * HD], qkv[:, 2 * HD:], out, None, None, B, S2, H, D, 1.0, False)
    got = out.view(B, S2, H, D).permute(0, 2, 1, 3).float()
    exp = vv.float()[:, :, sel, :]
    assert (got - exp).abs().max().item() < 1e-2, (got - exp).abs().max().item()


def test_attn_fwd_left_pad_mask(dev):
    B, H, S, D = 2, 2, 150, 128
    g = torch.Generator().manual_seed(5)
    q, k, v = (torch.randn(B, H, S, D, generator=g).to(dev).to(BF) for _ in range(3))
    kmask = torch.ones(B, S, dtype=torch.int32, device=dev)
    kmask[0, :37] = 0
    kmask[1, :5] = 0
    qkv = _pack_qkv(q, k, v)
    HD = H * D
    out = torch.empty((B * S, HD), device=dev, dtype=BF)
    lse = torch.empty(B, H, S, device=dev, dtype=torch.float32)
    scale = 1 / math.sqrt(D)
    hip.attn_fwd(qkv[:, :HD], qkv[:, HD:2 * HD], qkv[:, 2 * HD:], out, lse, kmask, B, S, H, D, scale, True)
    ref, _ = _attn_ref(q.float(), k.float(), v.float(), scale, True, kmask)
    got = out.view(B, S, H, D).permute(0, 2, 1, 3)
    valid = kmask.bool()[:, None, :, None].expand_as(ref)
    assert torch.isfinite(got.float()).all()
    assert _relerr(got[valid], ref[valid]) < 1e-2


@pytest.mark.parametrize("B,H,S,causal,pad,D", [(1, 1, 64, True, 0, 128), (2, 2, 200, True, 0, 128), (1, 2, 1088, True, 0, 128),
                                                (2, 2, 333, False, 0, 128), (2, 2, 150, True, 20, 128),
                                                (2, 3, 577, False, 0, 64), (1, 2, 130, True, 0, 64), (2, 2, 70, False, 9, 64),
                                                (2, 4, 333, True, 0, 128), (1, 8, 1088, True, 0, 128),
                                                (3, 8, 300, True, 37, 128), (2, 8, 450, False, 0, 128),
                                                # multiples of 32: whole waves with / without keys, ragged last 64-row stage,
                                                # key blocks with waves that own no key, padding, several items per workgroup
                                                (2, 4, 96, True, 0, 128), (3, 8, 288, True, 37, 128),
                                                (2, 8, 416, False, 0, 128), (1, 16, 160, True, 5, 128),
                                                (2, 8, 32, True, 0, 128)])
@pytest.mark.parametrize("variant", [(8, 0), (4, 0), (4, 8)], ids=["w8", "w4", "w4-grid8"])
def test_attn_bwd(dev, B, H, S, causal, pad, D, variant):
    # (4, 8): the persistent four-wave kernel on 8 workgroups, so every workgroup walks several items (seams, ring
    # run-ahead across items, counted waits behind the stores)
    if variant[0] == 4 and D != 128:
        pytest.skip("the four-wave dK/dV kernel is a head_dim-128 kernel")
    hip.attn_set_dkdv_variant(*variant)
    g = torch.Generator().manual_seed(S)
    q, k, v, do = (torch.randn(B, H, S, D, generator=g).to(dev).to(BF) for _ in range(4))
    kmask = None
    if pad:
        kmask = torch.ones(B, S, dtype=torch.int32, device=dev)
        kmask[0, :pad] = 0
    scale = 1 / math.sqrt(D)
    qf, kf, vf = (t.float().requires_grad_(True) for t in (q, k, v))
    ref, _ = _attn_ref(qf, kf, vf, scale, causal, kmask)
    dof = do.float()
    if pad:
        dof = dof * kmask[:, None, :, None].float()  # pad query rows carry no gradient
    ref.backward(dof)
    qkv = _pack_qkv(q, k, v)
    HD = H * D
    out = torch.empty((B * S, HD), device=dev, dtype=BF)
    lse = torch.empty(B, H, S, device=dev, dtype=torch.float32)
    hip.attn_fwd(qkv[:, :HD], qkv[:, HD:2 * HD], qkv[:, 2 * HD:], out, lse, kmask, B, S, H, D, scale, causal)
    dout = dof.to(BF).permute(0, 2, 1, 3).reshape(B * S, HD).contiguous()
    dqkv = torch.full((B * S, 3 * HD), 9.0, device=dev, dtype=BF)
    delta = torch.empty(2, B, H, S, device=dev, dtype=torch.float32)
    hip.attn_bwd(qkv[:, :HD], qkv[:, HD:2 * HD], qkv[:, 2 * HD:], out, lse, kmask, dout, delta, dqkv[:, :HD],
                 dqkv[:, HD:2 * HD], dqkv[:, 2 * HD:], B, S, H, D, scale, causal)
    torch.cuda.synchronize()
    unp = lambda t: t.reshape(B, S, H, D).permute(0, 2, 1, 3)
    dq, dk, dv = unp(dqkv[:, :HD]), unp(dqkv[:, HD:2 * HD]), unp(dqkv[:, 2 * HD:])
    if pad:
        m = kmask.bool()[:, None, :, None].expand_as(dq)
        dq = torch.where(m, dq, torch.zeros_like(dq))
    # bf16 P / dS operands: 1e-2 relative budget, same as forward
    for name, got, exp in (("dq", dq, qf.grad), ("dk", dk, kf.grad), ("dv", dv, vf.grad)):
        assert torch.isfinite(got.float()).all(), name
        assert _relerr(got, exp) < 1.5e-2, (name, _relerr(got, exp))
    # the workspace rows the dQ kernel leaves for the dK/dV kernel: delta = rowsum(dO o O) and lse * log2(e), every (b, h, s)
    want = (dout.float() * out.float()).reshape(B, S, H, D).sum(-1).permute(0, 2, 1)
    assert torch.allclose(delta[0], want, rtol=1e-5, atol=1e-5), (delta[0] - want).abs().max().item()
    fin = torch.isfinite(lse)
    assert torch.allclose(delta[1][fin], lse[fin] * math.log2(math.e), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("B,H,S,D,causal,use_rope", [(2, 4, 96, 128, True, True), (1, 8, 1088, 128, True, True),
                                                    (3, 8, 288, 128, True, False), (2, 3, 200, 128, True, True),
                                                    (2, 8, 416, 128, False, False), (2, 3, 72, 64, False, False),
                                                    # sequence lengths that are not multiples of 8: strided-transpose fallback
                                                    (2, 4, 100, 128, True, True), (8, 2, 333, 128, True, False)])
@pytest.mark.parametrize("variant", [(8, 0), (4, 0), (4, 8)], ids=["w8", "w4", "w4-grid8"])
def test_attn_bwd_transposed_copy(dev, B, H, S, D, causal, use_rope, variant):
    """mmu_attn_bwd_rope_t: dqkv_t == dqkv^T bit for bit (with and without the fused inverse RoPE), and dqkv itself is
    what the call without the transposed copy writes — whichever kernel or strided transpose produced each third."""
    hip.attn_set_dkdv_variant(*variant)
    g = torch.Generator().manual_seed(S + H)
    HD = H * D
    M = B * S
    qkv = (torch.randn(M, 3 * HD, generator=g) * 0.5).to(dev).to(BF)
    do = torch.randn(M, HD, generator=g).to(dev).to(BF)
    o = torch.empty(M, HD, device=dev, dtype=BF)
    lse = torch.empty(B, H, S, device=dev, dtype=torch.float32)
    sc = 1 / math.sqrt(D)
    hip.attn_fwd(qkv[:, :HD], qkv[:, HD:2 * HD], qkv[:, 2 * HD:], o, lse, None, B, S, H, D, sc, causal)
    rope = None
    if use_rope:
        pos = torch.arange(S, dtype=torch.int32, device=dev).repeat(B)
        cos_t = torch.empty(S, D // 2, device=dev, dtype=BF)
        sin_t = torch.empty(S, D // 2, device=dev, dtype=BF)
        hip.rope_table(cos_t, sin_t, S, D, 10000.0)
        rope = (pos, cos_t, sin_t)
    delta = torch.empty(2, B, H, S, device=dev, dtype=torch.float32)
    d_ref = torch.full((M, 3 * HD), 7.0, device=dev, dtype=BF)
    hip.attn_bwd(qkv[:, :HD], qkv[:, HD:2 * HD], qkv[:, 2 * HD:], o, lse, None, do, delta, d_ref[:, :HD], d_ref[:, HD:2 * HD],
                 d_ref[:, 2 * HD:], B, S, H, D, sc, causal, rope=rope)
    d1 = torch.full((M, 3 * HD), 5.0, device=dev, dtype=BF)
    dT = torch.full((3 * HD, M), 3.0, device=dev, dtype=BF)
    hip.attn_bwd(qkv[:, :HD], qkv[:, HD:2 * HD], qkv[:, 2 * HD:], o, lse, None, do, delta, d1[:, :HD], d1[:, HD:2 * HD],
                 d1[:, 2 * HD:], B, S, H, D, sc, causal, rope=rope, dqkv_t=dT)
    torch.cuda.synchronize()
    assert torch.equal(d1, d_ref)
    assert torch.equal(dT, d1.t().contiguous())


def test_attn_bwd_step_size_kernels_agree(dev):
    """The headline step's attention shape (30 sequences x 32 heads x 1088 x 128, causal, fused inverse RoPE): the
    persistent four-wave dK/dV kernel (34 items and ~300 ring steps per workgroup, ring running across item seams) and
    the eight-wave kernel give the same gradients to bf16 rounding of different summation orders, and dqkv^T is the
    transpose of dqkv bit for bit.  (No oracle at this size: the small shapes above pin both kernels to the fp32
    reference.)"""
    B, H, S, D = 30, 32, 1088, 128
    HD, M = H * D, B * S
    g = torch.Generator(device=dev).manual_seed(5)
    qkv = (torch.randn(M, 3 * HD, device=dev, generator=g) * 0.5).to(BF)
    do = torch.randn(M, HD, device=dev, generator=g).to(BF)
    o = torch.empty(M, HD, device=dev, dtype=BF)
    lse = torch.empty(B, H, S, device=dev, dtype=torch.float32)
    sc = 1 / math.sqrt(D)
    hip.attn_fwd(qkv[:, :HD], qkv[:, HD:2 * HD], qkv[:, 2 * HD:], o, lse, None, B, S, H, D, sc, True)
    pos = torch.arange(S, dtype=torch.int32, device=dev).repeat(B)
    cos_t = torch.empty(S, D // 2, device=dev, dtype=BF)
    sin_t = torch.empty(S, D // 2, device=dev, dtype=BF)
    hip.rope_table(cos_t, sin_t, S, D, 10000.0)
    delta = torch.empty(2, B, H, S, device=dev, dtype=torch.float32)
    outs = {}
    for v in (8, 4):
        hip.attn_set_dkdv_variant(v)
        d = torch.empty(M, 3 * HD, device=dev, dtype=BF)
        dT = torch.empty(3 * HD, M, device=dev, dtype=BF)
        hip.attn_bwd(qkv[:, :HD], qkv[:, HD:2 * HD], qkv[:, 2 * HD:], o, lse, None, do, delta, d[:, :HD], d[:, HD:2 * HD],
                     d[:, 2 * HD:], B, S, H, D, sc, True, rope=(pos, cos_t, sin_t), dqkv_t=dT)
        torch.cuda.synchronize()
        assert torch.isfinite(d.float()).all()
        for part in range(3):  # piecewise: a full .t().contiguous() of 800 MB is not needed
            assert torch.equal(dT[part * HD:(part + 1) * HD], d[:, part * HD:(part + 1) * HD].t().contiguous()), (v, part)
        outs[v] = d
        del dT
    assert torch.equal(outs[4][:, :HD], outs[8][:, :HD])  # dq comes from the same kernel either way
    assert _relerr(outs[4][:, HD:], outs[8][:, HD:]) < 2e-3


# ------------------------------------------------------------------------------------------------ norms
@pytest.mark.parametrize("M,d", [(5, 64), (33, 1024), (130, 4096), (7, 5120)])
def test_rmsnorm(dev, M, d):
    x = _randn((M, d), dev, 2.0, 1)
    w = (1 + 0.1 * torch.randn(d, generator=torch.Generator().manual_seed(2))).to(dev).to(BF)
    y = torch.empty_like(x)
    rstd = torch.empty(M, device=dev, dtype=torch.float32)
    hip.rmsnorm_fwd(x, w, y, rstd, 1e-5)
    xf = x.float()
    r = torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5)
    ref = (w.float() * (xf * r).to(BF).float()).to(BF)
    assert (y.float() - ref.float()).abs().max().item() <= 2 ** -6 * ref.float().abs().max().item()
    assert _relerr(y, ref) < 1e-3
    assert (rstd - r.squeeze(-1)).abs().max().item() < 1e-4 * r.max().item()
    # backward vs autograd of the fp32 formula
    dy = _randn((M, d), dev, 1.0, 3)
    dres = _randn((M, d), dev, 1.0, 4)
    xr = xf.clone().requires_grad_(True)
    wr = w.float().clone().requires_grad_(True)
    yr = wr * (xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-5))
    yr.backward(dy.float())
    dx = torch.empty_like(x)
    dw = torch.zeros(d, device=dev, dtype=BF)
    part = torch.empty(hip.rmsnorm_bwd_nparts(M) * d, device=dev, dtype=torch.float32)
    hip.rmsnorm_bwd(dy, x, w, rstd, dres, dx, part, dw, False)
    assert _relerr(dx, xr.grad + dres.float()) < 6e-3, _relerr(dx, xr.grad + dres.float())
    assert _relerr(dw, wr.grad) < 6e-3, _relerr(dw, wr.grad)
    dw2 = dw.clone()
    hip.rmsnorm_bwd(dy, x, w, rstd, None, dx, part, dw2, True)
    assert _relerr(dw2, 2 * wr.grad) < 8e-3
    assert _relerr(dx, xr.grad) < 6e-3


def test_layernorm(dev):
    M, d = 37, 1024
    x = _randn((M, d), dev, 2.0, 1)
    w = _randn((d,), dev, 1.0, 2)
    b = _randn((d,), dev, 1.0, 3)
    y = torch.empty_like(x)
    hip.layernorm_fwd(x, w, b, y, 1e-5)
    ref = torch.nn.functional.layer_norm(x.float(), (d,), w.float(), b.float(), 1e-5)
    assert _relerr(y, ref) < 3e-3


def test_clip_embed_im2col(dev):
    B, C, H, P, d = 2, 3, 28, 14, 64
    g = H // P
    pix = torch.randn(B, C, H, H, generator=torch.Generator().manual_seed(0)).to(dev)
    K = C * P * P
    Kpad = (K + 7) // 8 * 8
    col = torch.empty(B * g * g, Kpad, device=dev, dtype=BF)
    hip.im2col_patch(pix, col, P)
    ref = torch.nn.functional.unfold(pix, kernel_size=P, stride=P).transpose(1, 2).reshape(B * g * g, K)
    assert (col[:, :K].float() - ref.to(BF).float()).abs().max().item() == 0
    assert col[:, K:].abs().max().item() == 0
    patch = _randn((B * g * g, d), dev, 1.0, 1)
    cls = _randn((d,), dev, 1.0, 2)
    pos = _randn((g * g + 1, d), dev, 1.0, 3)
    out = torch.empty(B * (g * g + 1), d, device=dev, dtype=BF)
    hip.clip_embed(patch, cls, pos, out, B, g * g + 1, d)
    ref = torch.cat([cls.view(1, 1, d).expand(B, 1, d), patch.view(B, g * g, d)], dim=1).float() + pos.float()
    assert (out.view(B, -1, d).float() - ref.to(BF).float()).abs().max().item() == 0


# ------------------------------------------------------------------------------------------------- rope
def test_rope_matches_hf_formula(dev):
    B, S, H, D = 2, 50, 3, 128
    M = B * S
    HD = H * D
    qkv = _randn((M, 3 * HD), dev, 1.0, 1)
    orig = qkv.clone()
    pos = torch.arange(S, dtype=torch.int32, device=dev).repeat(B)
    cos_t = torch.empty(S, D // 2, device=dev, dtype=BF)
    sin_t = torch.empty(S, D // 2, device=dev, dtype=BF)
    hip.rope_table(cos_t, sin_t, S, D, 10000.0)
    inv_freq = 1.0 / (10000.0 ** (torch.arange(0, D, 2, device=dev).float() / D))
    ang = torch.arange(S, device=dev).float()[:, None] * inv_freq[None, :]
    assert (cos_t.float() - ang.cos().to(BF).float()).abs().max().item() <= 2 ** -7
    assert (sin_t.float() - ang.sin().to(BF).float()).abs().max().item() <= 2 ** -7
    hip.rope(qkv, pos, cos_t, sin_t, H, H, D, HD)
    # HF: q*cos + rotate_half(q)*sin in bf16 tensor arithmetic
    cos = torch.cat([cos_t, cos_t], -1)[pos.long()][:, None, :]
    sin = torch.cat([sin_t, sin_t], -1)[pos.long()][:, None, :]

    def rot(x):
        x1, x2 = x[..., : D // 2], x[..., D // 2:]
        return torch.cat([-x2, x1], -1)

    for off in (0, HD):
        x = orig[:, off:off + HD].view(M, H, D)
        ref = (x * cos) + (rot(x) * sin)
        got = qkv[:, off:off + HD].view(M, H, D)
        assert (got.float() - ref.float()).abs().max().item() == 0
    assert (qkv[:, 2 * HD:] == orig[:, 2 * HD:]).all()
    # inverse rotation returns (approximately, bf16) the original
    hip.rope(qkv, pos, cos_t, sin_t, H, H, D, HD, inverse=True)
    assert _relerr(qkv[:, :2 * HD], orig[:, :2 * HD]) < 1e-2


# ------------------------------------------------------------------------------------ swiglu / gelu / add
def test_swiglu_gelu_add(dev):
    M, I = 70, 344
    gu = _randn((M, 2 * I), dev, 2.0, 1)
    h = torch.empty(M, I, device=dev, dtype=BF)
    hip.swiglu_fwd(gu, h)
    gte, up = gu[:, :I], gu[:, I:]
    ref = torch.nn.functional.silu(gte.float()).to(BF) * up
    assert (h.float() - ref.float()).abs().max().item() <= 2 ** -7 * ref.float().abs().max().item()
    assert _relerr(h, ref) < 2e-3
    dh = _randn((M, I), dev, 1.0, 2)
    guf = gu.float().requires_grad_(True)
    (torch.nn.functional.silu(guf[:, :I]) * guf[:, I:]).backward(dh.float())
    dgu = torch.empty_like(gu)
    hip.swiglu_bwd(gu, dh, dgu)
    assert _relerr(dgu, guf.grad) < 6e-3
    x = _randn((M, 512), dev, 2.0, 3)
    y = torch.empty_like(x)
    hip.gelu_fwd(x, y)
    assert _relerr(y, torch.nn.functional.gelu(x.float())) < 3e-3
    xf = x.float().requires_grad_(True)
    dy = _randn((M, 512), dev, 1.0, 4)
    torch.nn.functional.gelu(xf).backward(dy.float())
    dx = torch.empty_like(x)
    hip.gelu_bwd(x, dy, dx)
    assert _relerr(dx, xf.grad) < 4e-3
    o = torch.empty_like(x)
    hip.add(x, dy, o)
    assert (o == (x + dy)).all()
    part = torch.empty(hip.colsum_nparts(M) * 512, device=dev, dtype=torch.float32)
    dst = torch.zeros(512, device=dev, dtype=BF)
    hip.colsum(x, part, dst, False)
    assert _relerr(dst, x.float().sum(0)) < 4e-3


# -------------------------------------------------------------------------------------------- embedding
def test_embedding_scatter_and_backward(dev):
    V, d, B, S, n_img = 97, 64, 3, 40, 6
    IMG = V - 1
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(0, V - 1, (B, S), generator=g)
    ids[:, 2:2 + n_img] = IMG
    ids[1, 20:23] = 5  # repeated token -> accumulation
    ids = ids.to(dev)
    table = _randn((V, d), dev, 1.0, 1)
    img = _randn((B * n_img, d), dev, 1.0, 2)
    M = B * S
    slot = torch.empty(M, dtype=torch.int32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    hip.image_slots(ids.view(-1), IMG, slot, cnt)
    assert cnt.item() == B * n_img
    mask = (ids.view(-1) == IMG)
    exp_slot = torch.where(mask, mask.int().cumsum(0) - 1, torch.full_like(slot, -1, dtype=torch.int64)).int()
    assert (slot == exp_slot).all()
    out = torch.empty(M, d, device=dev, dtype=BF)
    hip.embed_fwd(ids.view(-1), slot, table, img, out)
    ref = table[ids.view(-1)].clone()
    ref[mask] = img
    assert (out == ref).all()
    dh = _randn((M, d), dev, 1.0, 3)
    dimg = torch.zeros_like(img)
    scratch = torch.zeros(V, d, device=dev, dtype=torch.float32)
    flag = torch.zeros(V, device=dev, dtype=torch.int32)
    gtab = _randn((V, d), dev, 1.0, 4)
    g0 = gtab.clone()
    hip.embed_bwd(ids.view(-1), slot, dh, dimg, scratch, flag, gtab, V)
    assert (dimg == dh[mask]).all()
    exp = torch.zeros(V, d, device=dev)
    exp.index_add_(0, ids.view(-1)[~mask], dh.float()[~mask])
    assert _relerr(gtab, g0.float() + exp) < 4e-3
    assert scratch.abs().max().item() == 0 and flag.abs().max().item() == 0


# ----------------------------------------------------------------------------------------------- losses
def test_cross_entropy(dev):
    B, S, V = 3, 37, 32064
    M = B * S
    logits = _randn((M, V), dev, 3.0, 1)
    g = torch.Generator().manual_seed(2)
    labels = torch.randint(0, V, (B, S), generator=g)
    labels[:, :10] = -100
    labels[2, :] = -100
    labels = labels.to(dev)
    lse = torch.empty(M, device=dev, dtype=torch.float32)
    lrow = torch.empty(M, device=dev, dtype=torch.float32)
    hip.ce_fwd(logits, labels.view(-1), lse, lrow, S, shift=True)
    sums = torch.empty(3, device=dev, dtype=torch.float32)
    cnts = torch.empty(3, device=dev, dtype=torch.float32)
    hip.seg_reduce(lrow, labels.view(-1), S, M, V, S, True, sums, cnts)
    lf = logits.float().view(B, S, V)
    ref_rows = torch.nn.functional.cross_entropy(lf[:, :-1].reshape(-1, V), labels[:, 1:].reshape(-1),
                                                 ignore_index=-100, reduction="none").view(B, S - 1)
    assert (lse.view(B, S) - torch.logsumexp(lf, -1)).abs().max().item() < 1e-4
    assert (sums - ref_rows.sum(1)).abs().max().item() < 1e-2
    assert (cnts - (labels[:, 1:] != -100).sum(1).float()).abs().max().item() == 0
    # backward: mean over the valid tokens of segments {0,1} (seg 2 has none), scaled by (-1.5, 2.0, 1.0)
    scale = torch.tensor([-1.5, 2.0, 1.0], device=dev)
    dl = torch.empty_like(logits)
    hip.ce_bwd(logits, labels.view(-1), lse, scale, cnts, S, True, dl, S, shift=True)
    lr = lf.clone().requires_grad_(True)
    tot = 0
    for b in range(2):
        tot = tot + scale[b] * torch.nn.functional.cross_entropy(lr[b, :-1], labels[b, 1:], ignore_index=-100)
    tot.backward()
    assert _relerr(dl.view(B, S, V)[:2], lr.grad[:2]) < 5e-3
    assert dl.view(B, S, V)[2].abs().max().item() == 0
    # in place
    l2 = logits.clone()
    hip.ce_bwd(l2, labels.view(-1), lse, scale, cnts, S, True, l2, S, shift=True)
    assert (l2 == dl).all()


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_rowloss(dev, mode):
    M, V, beta = 23, 4096, 0.7
    s = _randn((M, V), dev, 2.0, 1)
    t = _randn((M, V), dev, 2.0, 2)
    stats = torch.empty(M, 4, device=dev, dtype=torch.float32)
    hip.rowloss_fwd(mode, s, None if mode == 3 else t, beta, stats)
    sr = s.float().requires_grad_(True)
    lps = torch.log_softmax(sr, -1)
    lpt = torch.log_softmax(t.float(), -1)
    if mode == 0:
        rows = (lpt.exp() * (lpt - lps)).sum(-1)
    elif mode == 1:
        rows = (lps.exp() * (lps - lpt)).sum(-1)
    elif mode == 2:
        rows = torch.log(1 + torch.pow(lps.exp() / (lpt.exp() + 1e-8), beta)).sum(-1)
    else:
        rows = (lps.exp() * lps).sum(-1)
    assert (stats[:, 2] - rows).abs().max().item() < 2e-3 * max(1.0, rows.abs().max().item())
    coef = torch.tensor([0.37], device=dev)
    (0.37 * rows.sum()).backward()
    dl = torch.empty_like(s)
    hip.rowloss_bwd(mode, s, None if mode == 3 else t, beta, stats, coef, dl)
    assert _relerr(dl, sr.grad) < 6e-3, _relerr(dl, sr.grad)


# ---------------------------------------------------------------------------------------------- optimiser
def test_adamw_and_gradnorm(dev):
    n = 8 * 1000 + 8 * 37
    p0 = _randn((n,), dev, 1.0, 1)
    m = torch.zeros(n, device=dev)
    v = torch.zeros(n, device=dev)
    pr = p0.float().clone().requires_grad_(True)
    opt = torch.optim.AdamW([pr], lr=1e-2, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.01)
    p = p0.clone()
    pf = p0.float().clone()  # fp32 shadow of the bf16-rounded trajectory is not kept: compare per step
    for step in range(1, 4):
        g = _randn((n,), dev, 3.0, 10 + step)
        ss = torch.zeros(hip.SUMSQ_FLOATS, device=dev)
        hip.sumsq(g, ss)
        assert abs(ss[0].item() - g.float().pow(2).sum().item()) < 1e-3 * ss[0].item()
        ss2 = torch.full((hip.SUMSQ_FLOATS,), 7.0, device=dev)
        hip.sumsq(g, ss2)
        assert ss2[0].item() == ss[0].item()  # fixed summation order: bit-identical, whatever the buffer held
        # reference step from the CURRENT bf16 params (isolates one update)
        pr.data.copy_(p.float())
        pr.grad = g.float().clone()
        torch.nn.utils.clip_grad_norm_([pr], 1.0)
        opt.step()
        gz = g.clone()
        hip.adamw(p, gz, m, v, 1e-2, 0.9, 0.95, 1e-8, 0.01, step, ss, 1.0, 1.0, True)
        assert gz.abs().max().item() == 0
        assert (p.float() - pr.data).abs().max().item() <= 2 ** -8 * pr.data.abs().max().item() + 1e-6
        st = opt.state[pr]
        assert _relerr(m, st["exp_avg"]) < 1e-5 and _relerr(v, st["exp_avg_sq"]) < 1e-5
    x = _randn((n,), dev, 1.0, 5)
    y = x.clone()
    hip.scale_(y, 0.25)
    assert (y == (x.float() * 0.25).to(BF)).all()


def test_transpose(dev):
    for R, Cc in [(64, 64), (200, 136), (4096, 1032), (8, 8)]:
        x = _randn((R, Cc), dev, 1.0, R)
        y = torch.full((Cc, R), 3.0, device=dev, dtype=BF)
        hip.transpose(x, y)
        assert torch.equal(y, x.t().contiguous()), (R, Cc)


@pytest.mark.parametrize("tile", [128, 256])
@pytest.mark.parametrize("kl", [32, 64])
def test_gemm_fused_lowrank_update(dev, tile, kl):
    """out = x W^T + U BL^T with the rank-kl term applied inside the GEMM (LoRA epilogue); integer data -> exact."""
    hip.gemm_set_tile(tile)
    M, N, K = 300, 264, 192
    x, w = _rand_int((M, K), dev, seed=1), _rand_int((N, K), dev, seed=2)
    u, bl = _rand_int((M, kl), dev, seed=3), _rand_int((N, kl), dev, seed=4)
    bl[:, 24:] = 0  # zero padding columns, as the fused q|k|v adapters leave them
    ref = (x.float() @ w.float().t() + u.float() @ bl.float().t()).to(BF)
    out = torch.empty(M, N, device=dev, dtype=BF)
    hip.gemm(hip.NT, x, w, out, lora_u=u, lora_b=bl)
    assert torch.equal(out, ref)
    res = _rand_int((M, N), dev, seed=5)
    hip.gemm(hip.NT, x, w, out, resid=res, lora_u=u, lora_b=bl)
    assert torch.equal(out, (ref.float() + res.float()).to(BF))


def test_fused_transposed_producers(dev):
    """SwiGLU / RMSNorm-apply / SwiGLU-backward with transposed outputs == the plain kernels followed by a transpose."""
    M, I, d = 200, 344, 256
    gu = _randn((M, 2 * I), dev, 2.0, 1)
    act = torch.empty(M, I, device=dev, dtype=BF)
    hip.swiglu_fwd(gu, act)
    actT = torch.full((I, M), 9.0, device=dev, dtype=BF)
    hip.swiglu_fwd_t(gu, actT)
    assert torch.equal(actT, act.t().contiguous())
    dh = _randn((M, I), dev, 1.0, 2)
    dgu = torch.empty_like(gu)
    hip.swiglu_bwd(gu, dh, dgu)
    dgu2 = torch.empty_like(gu)
    dguT = torch.empty(2 * I, M, device=dev, dtype=BF)
    hip.swiglu_bwd_t(gu, dh, dgu2, dguT)
    assert torch.equal(dgu2, dgu) and torch.equal(dguT, dgu.t().contiguous())
    x = _randn((M, d), dev, 2.0, 3)
    w = (1 + 0.1 * torch.randn(d, generator=torch.Generator().manual_seed(4))).to(dev).to(BF)
    y = torch.empty_like(x)
    rstd = torch.empty(M, device=dev, dtype=torch.float32)
    hip.rmsnorm_fwd(x, w, y, rstd, 1e-5)
    yT = torch.empty(d, M, device=dev, dtype=BF)
    hip.rmsnorm_apply_t(x, w, rstd, yT)
    assert torch.equal(yT, y.t().contiguous())


def test_fused_rope_epilogue_and_bwd_store(dev):
    """(a) qkv GEMM with RoPE in the epilogue == GEMM then the rope kernel (bit-exact, 256^2 kernel);
    (b) attention backward with the inverse rotation fused into the dq/dk stores == attn_bwd then rope(inverse)."""
    import math
    B, S, H, D = 2, 320, 2, 128
    M, dt = B * S, H * D
    x = _randn((M, 512), dev, 1.0, 1)
    w = _randn((3 * dt, 512), dev, 0.05, 2)
    pos = torch.arange(S, dtype=torch.int32, device=dev).repeat(B)
    cos_t = torch.empty(S, D // 2, device=dev, dtype=BF)
    sin_t = torch.empty(S, D // 2, device=dev, dtype=BF)
    hip.rope_table(cos_t, sin_t, S, D, 10000.0)
    hip.gemm_set_tile(256)
    ref = torch.empty(M, 3 * dt, device=dev, dtype=BF)
    hip.gemm(hip.NT, x, w, ref)
    hip.rope(ref, pos, cos_t, sin_t, H, H, D, dt)
    out = torch.empty_like(ref)
    assert hip.gemm_rope(x, w, out, pos, cos_t, sin_t, 2 * dt)
    assert torch.equal(out, ref)
    # the split-K tail of a fused-RoPE GEMM: the fold kernel rotates the remainder tiles (30 x 12 = 360 tiles = 256 in
    # the kernel's epilogue + 104 through the fold; and 17 x 3 = 51 tiles, all through the fold); integer x / w: exact
    hip.ensure_gemm_workspace(dev)
    for (Mb, Hb) in ((7680, 8), (4352, 2)):
        dtb, Sb = Hb * D, Mb // 2
        xi = _rand_int((Mb, 512), dev, -2, 2, seed=31)
        wi = _rand_int((3 * dtb, 512), dev, -2, 2, seed=32)
        posb = torch.arange(Sb, dtype=torch.int32, device=dev).repeat(2)
        cb = torch.empty(Sb, D // 2, device=dev, dtype=BF)
        sb = torch.empty(Sb, D // 2, device=dev, dtype=BF)
        hip.rope_table(cb, sb, Sb, D, 10000.0)
        refb = torch.empty(Mb, 3 * dtb, device=dev, dtype=BF)
        hip.gemm_set_tail_split(0)
        hip.gemm(hip.NT, xi, wi, refb)
        hip.rope(refb, posb, cb, sb, Hb, Hb, D, dtb)
        try:
            hip.gemm_set_tail_split(2)
            outb = torch.empty_like(refb)
            assert hip.gemm_rope(xi, wi, outb, posb, cb, sb, 2 * dtb)
        finally:
            hip.gemm_set_tail_split(1)
        assert torch.equal(outb, refb), (Mb, Hb)
    # backward stores
    qkv = ref
    o = torch.empty(M, dt, device=dev, dtype=BF)
    lse = torch.empty(B, H, S, device=dev, dtype=torch.float32)
    sc = 1 / math.sqrt(D)
    hip.attn_fwd(qkv[:, :dt], qkv[:, dt:2 * dt], qkv[:, 2 * dt:], o, lse, None, B, S, H, D, sc, True)
    do = _randn((M, dt), dev, 1.0, 3)
    delta = torch.empty(2, B, H, S, device=dev, dtype=torch.float32)
    for variant in ((8, 0), (4, 0), (4, 8)):  # both dK/dV kernels; (4, 8): one workgroup walks several items
        hip.attn_set_dkdv_variant(*variant)
        d1 = torch.empty(M, 3 * dt, device=dev, dtype=BF)
        hip.attn_bwd(qkv[:, :dt], qkv[:, dt:2 * dt], qkv[:, 2 * dt:], o, lse, None, do, delta, d1[:, :dt],
                     d1[:, dt:2 * dt], d1[:, 2 * dt:], B, S, H, D, sc, True)
        hip.rope(d1, pos, cos_t, sin_t, H, H, D, dt, inverse=True)
        d2 = torch.empty_like(d1)
        hip.attn_bwd(qkv[:, :dt], qkv[:, dt:2 * dt], qkv[:, 2 * dt:], o, lse, None, do, delta, d2[:, :dt],
                     d2[:, dt:2 * dt], d2[:, 2 * dt:], B, S, H, D, sc, True, rope=(pos, cos_t, sin_t))
        assert torch.equal(d2, d1), variant


@pytest.mark.parametrize("layout", [hip.NT, hip.TN])
def test_gemm_splitk_accumulate(dev, layout):
    """Skinny output reduced over many tokens: split-K with fp32 atomics == the plain GEMM (integer data: exact)."""
    M, N, K = 32, 264, 4000 if layout == hip.NT else 4104
    a = _rand_int((M, K), dev, -2, 2, seed=1)
    b = _rand_int((N, K), dev, -2, 2, seed=2)
    A = a if layout == hip.NT else a.t().contiguous()
    Bm = b if layout == hip.NT else b.t().contiguous()
    dst0 = _rand_int((M, N), dev, -4, 4, seed=3)
    mask = (torch.rand(M, N, device=dev) > 0.3).to(BF)
    dst = dst0.clone()
    scr = torch.zeros(M, N, device=dev, dtype=torch.float32)
    hip.gemm_splitk_accumulate(layout, A, Bm, scr, dst, mask, splits=7)
    ref = (dst0.float() + (a.float() @ b.float().t()) * mask.float()).to(BF)
    assert torch.equal(dst, ref)
    assert scr.abs().max().item() == 0


def test_clip_normalize_u8_matches_processor(dev):
    """HIP rescale+normalise of uint8 pixels == the reference processor's fp32 pixel_values (golden digest from
    oracle/make_golden_collator.py) bit for bit, and == the numpy restatement on every byte value."""
    import json
    import os

    from multimodal_unlearning_amd import collator as C
    from oracle.tiny_tokenizer import adapter, pixel_digest, test_image as make_image

    G = os.path.join(os.path.dirname(__file__), "golden")
    npz, meta = np.load(os.path.join(G, "collator.npz")), json.load(open(os.path.join(G, "collator_pairs.json")))
    c = meta["cases"]["basic_left"]
    samples = [dict(s, image=make_image(*meta["images"][s["image"]]) if s["image"] is not None else None) for s in c["samples"]]
    out = C.mm_data_collator_preprocessor(samples, adapter("left"), c["max_length"], device=dev)
    assert "pixel_values_u8" not in out and out["pixel_values"].dtype == torch.float32 and out["input_ids"].is_cuda
    d = pixel_digest(out["pixel_values"].cpu().numpy())
    assert np.array_equal(d["sub"], npz["basic_left.pixel_values.sub"])
    for k in ("sum", "sumsq", "wsum"):
        assert d[k] == float(npz[f"basic_left.pixel_values.{k}"]), k
    # every byte value in every channel
    u8 = (torch.arange(2 * 3 * 16 * 32, dtype=torch.int64) % 256).to(torch.uint8).reshape(2, 3, 16, 32)
    pv = torch.empty(u8.shape, device=dev, dtype=torch.float32)
    hip.clip_normalize_u8(u8.to(dev), pv, C.OPENAI_CLIP_MEAN, C.OPENAI_CLIP_STD)
    r = (u8.numpy().astype(np.float64) * (1 / 255)).astype(np.float32)
    ref = (r - np.array(C.OPENAI_CLIP_MEAN, np.float32).reshape(1, 3, 1, 1)) / np.array(C.OPENAI_CLIP_STD, np.float32).reshape(1, 3, 1, 1)
    assert np.array_equal(pv.cpu().numpy(), ref)


def test_gemm256_persistent_launch_matches(dev):
    """Persistent launches of the 256^2 kernel (one workgroup per CU over 17x17 ragged tiles; mode 1 draws tiles from the
    device-side queue, mode 2 walks them statically) == one workgroup per tile, for the plain, bias+residual and
    accumulate epilogues (all through the swapped 16-B-store epilogue).  The queue counters are left zero."""
    M, N, K = 4200, 4344, 328
    A = _rand_int((M, K), dev, -2, 2, seed=11)
    Bm = _rand_int((N, K), dev, -2, 2, seed=12)
    bias = _rand_int((N,), dev, -3, 3, seed=13)
    resid = _rand_int((M, N), dev, -3, 3, seed=14)
    ref = A.float() @ Bm.float().t()
    ws = hip.ensure_gemm_workspace(dev)
    nq = hip.gemm_tile_queue_bytes()
    hip.gemm_set_tile(256)
    try:
        outs = []
        for mode, xmap in ((0, 1), (1, 1), (2, 1), (0, 0), (1, 0), (2, 0)):  # launch mode x tile -> XCD map (both bijections)
            hip.gemm_set_persistent(mode)
            hip.gemm_set_xcd_map(xmap)
            c0 = torch.empty(M, N, device=dev, dtype=BF)
            hip.gemm(hip.NT, A, Bm, c0)
            c1 = torch.empty(M, N, device=dev, dtype=BF)
            hip.gemm(hip.NT, A, Bm, c1, bias=bias, resid=resid)
            c2 = resid.clone()
            hip.gemm(hip.NT, A, Bm, c2, accumulate=True)
            outs.append((c0, c1, c2))
        for other in outs[1:]:
            for x, y in zip(outs[0], other):
                assert torch.equal(x, y)
        assert torch.equal(outs[0][0], ref.to(BF))
        assert torch.equal(outs[0][1], ((ref + bias.float()).to(BF).float() + resid.float()).to(BF))
        assert torch.equal(outs[0][2], (ref + resid.float()).to(BF))
        assert int(ws[-nq:].view(torch.int32).abs().sum()) == 0
    finally:
        hip.gemm_set_persistent(1)  # the defaults
        hip.gemm_set_xcd_map(1)
        hip.gemm_set_tile(0)


def test_gemm256_tile_queue_many_launches_and_few_tiles(dev):
    """The tile queue over more launches than it has slots (64), with fewer tiles than workgroups in some XCD classes
    (a 3 x 90 tile grid = 270 tiles on 256 workgroups: classes 0..5 hold 34 tiles for 32 workgroups, 6..7 hold 33) and
    with a tile count that is not a multiple of 8: every launch exact, counters zero at the end."""
    hip.gemm_set_tile(256)
    ws = hip.ensure_gemm_workspace(dev)
    nq = hip.gemm_tile_queue_bytes()
    try:
        hip.gemm_set_persistent(1)
        for (M, N, K) in ((700, 23040, 192), (256 * 17 - 8, 256 * 17 - 8, 128)):
            A = _rand_int((M, K), dev, -2, 2, seed=41)
            Bm = _rand_int((N, K), dev, -2, 2, seed=42)
            ref = (A.float() @ Bm.float().t()).to(BF)
            outs = [torch.empty(M, N, device=dev, dtype=BF) for _ in range(70)]
            for c in outs:
                hip.gemm(hip.NT, A, Bm, c)
            for c in outs[::23] + [outs[-1]]:
                assert torch.equal(c, ref)
        assert int(ws[-nq:].view(torch.int32).abs().sum()) == 0
    finally:
        hip.gemm_set_tile(0)


@pytest.mark.parametrize("layout", [hip.NT, hip.NN, hip.TN])
@pytest.mark.parametrize("shape", [(4200, 4344, 2048), (1000, 1096, 520), (2304, 4096, 8192 + 64), (256 * 7, 256 * 5, 192)])
def test_gemm256_split_tail_matches(dev, layout, shape):
    """Split-K tail (remainder tiles cut into K-tile units over 256 workers + fold) == the plain launch, bit for bit on
    integer data, for the store and accumulate epilogues and the fused low-rank term (added by the k0 == 0 segment)."""
    M, N, K = shape
    hip.ensure_gemm_workspace(dev)
    a = _rand_int((M, K), dev, -2, 2, seed=21)
    b = _rand_int((N, K), dev, -2, 2, seed=22)
    A = a if layout != hip.TN else a.t().contiguous()
    Bm = b if layout == hip.NT else b.t().contiguous()
    c_init = _rand_int((M, N), dev, -3, 3, seed=23)
    U = _rand_int((M, 32), dev, -1, 1, seed=24)
    BL = _rand_int((N, 32), dev, -1, 1, seed=25)
    bias = _rand_int((N,), dev, -2, 2, seed=26)
    hip.gemm_set_tile(256)
    try:
        res = []
        for mode in (0, 2):
            hip.gemm_set_tail_split(mode)
            c0 = torch.empty(M, N, device=dev, dtype=BF)
            hip.gemm(layout, A, Bm, c0)
            c1 = c_init.clone()
            hip.gemm(layout, A, Bm, c1, accumulate=True)
            c2 = torch.empty(M, N, device=dev, dtype=BF)
            hip.gemm(layout, A, Bm, c2, lora_u=U, lora_b=BL)
            c3 = torch.empty(M, N, device=dev, dtype=BF)  # bias + residual epilogue (o / down projections: round 2)
            hip.gemm(layout, A, Bm, c3, bias=bias, resid=c_init)
            c4 = torch.empty(M, N, device=dev, dtype=BF)
            hip.gemm(layout, A, Bm, c4, bias=bias, act=hip.ACT_QUICK_GELU)
            res.append((c0, c1, c2, c3, c4))
        for x, y in zip(*res):
            assert torch.equal(x, y)
        ref = a.float() @ b.float().t()
        assert torch.equal(res[1][0], ref.to(BF))
        assert torch.equal(res[1][1], (ref + c_init.float()).to(BF))
        assert torch.equal(res[1][2], (ref + U.float() @ BL.float().t()).to(BF))
        assert torch.equal(res[1][3], ((ref + bias.float()).to(BF).float() + c_init.float()).to(BF))
    finally:
        hip.gemm_set_tail_split(1)
        hip.gemm_set_tile(0)


def test_clip_backward_pieces(dev):
    """LayerNorm input gradient (+ residual), quick-GELU fwd/bwd and the head-dim padding copy vs torch autograd."""
    M, d = 37, 1024
    x = _randn((M, d), dev, 1.5, seed=31)
    w = _randn((d,), dev, 1.0, seed=32)
    dy = _randn((M, d), dev, 1.0, seed=33)
    dres = _randn((M, d), dev, 1.0, seed=34)
    xf = x.float().requires_grad_(True)
    y = torch.nn.functional.layer_norm(xf, (d,), w.float(), None, 1e-5)
    y.backward(dy.float())
    dx = torch.empty_like(x)
    hip.layernorm_bwd(dy, x, w, dres, dx, 1e-5)
    assert _relerr(dx.float(), xf.grad + dres.float()) < 6e-3
    hip.layernorm_bwd(dy, x, w, None, dx, 1e-5)
    assert _relerr(dx.float(), xf.grad) < 6e-3
    z = _randn((M, 4096), dev, 2.0, seed=35)
    dz_up = _randn((M, 4096), dev, 1.0, seed=36)
    zf = z.float().requires_grad_(True)
    ya = zf * torch.sigmoid(1.702 * zf)
    ya.backward(dz_up.float())
    za, dz = torch.empty_like(z), torch.empty_like(z)
    hip.quick_gelu_fwd(z, za)
    hip.quick_gelu_bwd(z, dz_up, dz)
    assert _relerr(za.float(), ya.detach()) < 4e-3 and _relerr(dz.float(), zf.grad) < 6e-3
    H = 4
    src = _randn((M, 3 * H * 64), dev, 1.0, seed=37)
    padded = torch.empty(M, H * 128, device=dev, dtype=BF)
    hip.head_pad(src[:, H * 64:2 * H * 64], 64, padded, 128, H)
    ref = torch.zeros(M, H, 128, device=dev, dtype=BF)
    ref[:, :, :64] = src[:, H * 64:2 * H * 64].reshape(M, H, 64)
    assert torch.equal(padded.view(M, H, 128), ref)
    back = torch.empty(M, H * 64, device=dev, dtype=BF)
    hip.head_pad(padded, 128, back, 64, H)
    assert torch.equal(back, src[:, H * 64:2 * H * 64])


def test_adamw8bit_blockwise(dev):
    """Block-wise 8-bit AdamW kernel vs the CPU restatement of the published algorithm (oracle/adamw8bit_ref.py; parity
    unpinned against bitsandbytes itself, which the reference's req.txt pins but /root/reference does not hold): codes,
    absmax and parameters over 3 steps incl. a ragged last block; and the trajectory stays close to fp32-state AdamW."""
    from oracle import adamw8bit_ref as R

    n = 3 * 2048 + 1000
    m1, m2 = hip.dynamic_map(True), hip.dynamic_map(False)
    assert torch.equal(m1, R.dynamic_map(True)) and torch.equal(m2, R.dynamic_map(False))
    assert m1.numel() == 256 and abs(float(m1[0]) + 0.99296875) < 1e-6 and float(m1[-1]) == 1.0 and float(m2[0]) == 0.0 and float(m2[-1]) == 1.0
    z1, z2 = int((m1 == 0).nonzero()[0]), int((m2 == 0).nonzero()[0])
    p = _randn((n,), dev, 0.05, 41)
    pr = p.cpu().clone()
    c1 = torch.full((n,), z1, device=dev, dtype=torch.uint8)
    c2 = torch.full((n,), z2, device=dev, dtype=torch.uint8)
    nb = hip.adamw8bit_nblocks(n)
    assert nb == 4
    a1 = torch.zeros(nb, device=dev)
    a2 = torch.zeros(nb, device=dev)
    rc1, rc2, ra1, ra2 = c1.cpu().clone(), c2.cpu().clone(), a1.cpu().clone(), a2.cpu().clone()
    p32 = p.float().clone()
    m32 = torch.zeros(n, device=dev)
    v32 = torch.zeros(n, device=dev)
    pf = p.clone()
    d1, d2 = m1.to(dev), m2.to(dev)
    ss = torch.zeros(hip.SUMSQ_FLOATS, device=dev)
    for step in range(1, 4):
        g = _randn((n,), dev, 0.3, 50 + step)
        hip.sumsq(g, ss)
        clip = min(1.0, 1.0 / (float(ss[0]) ** 0.5 + 1e-6))
        g_ref = g.cpu().clone()
        g2 = g.clone()
        hip.adamw(pf, g2, m32, v32, 1e-2, 0.9, 0.95, 1e-8, 0.01, step, ss, 1.0, 1.0, True)
        hip.adamw8bit(p, g, c1, c2, a1, a2, d1, d2, 1e-2, 0.9, 0.95, 1e-8, 0.01, step, ss, 1.0, 1.0, True)
        assert float(g.abs().max()) == 0.0
        pr, rc1, rc2, ra1, ra2 = R.adamw8bit_step(pr, g_ref, rc1, rc2, ra1, ra2, m1, m2, 1e-2, 0.9, 0.95, 1e-8, 0.01, step, clip)
        assert _relerr(a1.cpu(), ra1) < 1e-5 and _relerr(a2.cpu(), ra2) < 1e-5
        # codes: identical except where a value sits on a code mid-point within fp32 rounding (allow a handful, 1 code off)
        for got, ref in ((c1.cpu(), rc1), (c2.cpu(), rc2)):
            diff = (got.int() - ref.int()).abs()
            assert int(diff.max()) <= 1 and float((diff > 0).float().mean()) < 2e-3
        rc1, rc2 = c1.cpu().clone(), c2.cpu().clone()  # keep the two trajectories on the same codes
        assert (p.float().cpu() - pr.float()).abs().max().item() <= 2 ** -7 * pr.float().abs().max().item()
        pr = p.cpu().clone()
    # 8-bit moments track the fp32-moment trajectory
    assert _relerr(p.float(), pf.float()) < 2e-2


def test_c_abi_from_a_native_caller(dev, tmp_path):
    """The boundary is a C ABI: a torch-free C++ program (tests/c_abi/gemm_caller.cpp) built with hipcc against
    include/mmu_hip.h and libmmu_hip.so runs the GEMM entry point and checks it exactly."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "multimodal_unlearning_amd", "csrc")
    exe = str(tmp_path / "gemm_caller")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-I", os.path.join(root, "include"),
                    os.path.join(root, "tests", "c_abi", "gemm_caller.cpp"), "-L", libdir, "-lmmu_hip",
                    f"-Wl,-rpath,{libdir}", "-o", exe], check=True, capture_output=True, timeout=300)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "c-abi caller ok" in r.stdout


def test_gemm256_tile_queue_with_held_compute_units(dev, tmp_path):
    """The situation of the overlapped gradient exchange on one GPU: a kernel on another stream (tests/c_abi/squatter.hip:
    64 workgroups parked for 25 ms, standing in for RCCL's all-reduce) holds compute units while the persistent GEMM
    runs, so some of its workgroups start late or after the others have drained the queue.  Every tile must still be
    computed exactly once: exact result on integer data for the store and accumulate epilogues, counters back to zero."""
    import ctypes
    import os
    import subprocess
    import time

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = str(tmp_path / "libsquatter.so")
    subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "-o", so,
                    os.path.join(root, "tests", "c_abi", "squatter.hip")], check=True, capture_output=True, timeout=300)
    sq = ctypes.CDLL(so)
    sq.squat_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p]
    sq.squat_launch.restype = ctypes.c_int
    M, N, K = 256 * 24 - 40, 256 * 20 - 8, 1024  # 480 tiles: the queue hands out ~2 per workgroup
    A = _rand_int((M, K), dev, -2, 2, seed=51)
    Bm = _rand_int((N, K), dev, -2, 2, seed=52)
    c_init = _rand_int((M, N), dev, -3, 3, seed=53)
    ref = A.float() @ Bm.float().t()
    ws = hip.ensure_gemm_workspace(dev)
    nq = hip.gemm_tile_queue_bytes()
    side = torch.cuda.Stream()
    hip.gemm_set_tile(256)
    try:
        hip.gemm_set_persistent(1)
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            assert sq.squat_launch(64, 256, 2_500_000, ctypes.c_void_p(side.cuda_stream)) == 0
        time.sleep(0.003)  # the squatter is resident before the GEMMs are launched
        outs = []
        for _ in range(6):
            c0 = torch.empty(M, N, device=dev, dtype=BF)
            hip.gemm(hip.NT, A, Bm, c0)
            c2 = c_init.clone()
            hip.gemm(hip.NT, A, Bm, c2, accumulate=True)
            outs.append((c0, c2))
        torch.cuda.synchronize()
        for c0, c2 in outs:
            assert torch.equal(c0, ref.to(BF))
            assert torch.equal(c2, (ref + c_init.float()).to(BF))
        assert int(ws[-nq:].view(torch.int32).abs().sum()) == 0
    finally:
        hip.gemm_set_tile(0)


# ------------------------------------------------------- round-2 kernels: one-pass row loss, RMU terms, eval stats
@pytest.mark.parametrize("mode", [hip.KL_STUDENT_INPUT, hip.KL_TEACHER_INPUT, hip.NPO, hip.ENTROPY])
@pytest.mark.parametrize("V", [32064, 520])
def test_rowloss_register_resident_equals_streaming(dev, mode, V):
    """The one-pass kernel (rows held in registers, every logit read once) vs the two-walk streaming kernel: same
    per-element arithmetic, different summation order over threads -> equal to fp32 rounding of the row sums."""
    M = 37
    s, t = _randn((M, V), dev, 2.0, seed=1), _randn((M, V), dev, 2.0, seed=2)
    st_r = torch.zeros(M, 4, device=dev)
    st_s = torch.zeros(M, 4, device=dev)
    tt = None if mode == hip.ENTROPY else t
    hip.rowloss_fwd(mode, s, tt, 0.7, st_r)
    hip.rowloss_set_streaming(True)
    try:
        hip.rowloss_fwd(mode, s, tt, 0.7, st_s)
    finally:
        hip.rowloss_set_streaming(False)
    assert torch.allclose(st_r[:, :2], st_s[:, :2], rtol=1e-6, atol=1e-6)        # the two log-sum-exps
    assert torch.allclose(st_r[:, 2:], st_s[:, 2:], rtol=2e-5, atol=2e-5 * V ** 0.5)  # row value / aux sums


def test_rmu_terms_vs_torch(dev):
    """mmu_seqmean / mmu_rmu_pair / mmu_mse (fwd + bwd) vs the reference's expressions in fp32 torch:
    mse_loss(act.mean(1), control [B,1,H]) (-> [B,B,H] broadcast, mm/trainer.py:316) and mse_loss(act, other)."""
    import torch.nn.functional as F

    from multimodal_unlearning_amd.trainer import _MseFn, _RmuForgetFn

    B, S, H = 3, 150, 1024 + 512
    act = _randn((B, S, H), dev, 1.5, seed=3)
    other = _randn((B, S, H), dev, 1.5, seed=4)
    rand = torch.rand(B, 1, H, generator=torch.Generator().manual_seed(5))
    ctrl = (rand / torch.norm(rand) * 300).to(dev)
    a1 = act.clone().requires_grad_(True)
    loss = _RmuForgetFn.apply(a1, ctrl.reshape(B, H).contiguous()) * 0.7
    loss.backward()
    a2 = act.float().requires_grad_(True)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = F.mse_loss(a2.mean(dim=1), ctrl) * 0.7
    ref.backward()
    assert abs(loss.item() - ref.item()) < 1e-5 * abs(ref.item())
    assert _relerr(a1.grad, a2.grad) < 4e-3  # bf16 rounding of the gradient
    a3 = act.clone().requires_grad_(True)
    l2 = _MseFn.apply(a3, other, None)
    l2.backward()
    a4 = act.float().requires_grad_(True)
    r2 = F.mse_loss(a4, other.float())
    r2.backward()
    assert abs(l2.item() - r2.item()) < 1e-5 * r2.item() and _relerr(a3.grad, a4.grad) < 4e-3
    cvec = ctrl[0, 0].contiguous()
    a5 = act.clone().requires_grad_(True)
    l3 = _MseFn.apply(a5, None, cvec)
    l3.backward()
    a6 = act.float().requires_grad_(True)
    r3 = F.mse_loss(a6, cvec.expand_as(a6))
    r3.backward()
    assert abs(l3.item() - r3.item()) < 1e-5 * r3.item() and _relerr(a5.grad, a6.grad) < 4e-3
    # deterministic: the same bits on a second call
    l2b = _MseFn.apply(act, other, None)
    assert l2b.item() == l2.item()


def test_seq_ce_stats_vs_torch(dev):
    import torch.nn.functional as F

    n, S, V = 5, 33, 520
    logits = _randn((n, S, V), dev, 2.0, seed=6)
    labels = torch.randint(0, V, (n, S), generator=torch.Generator().manual_seed(7))
    labels[:, :9] = -100
    labels[2, 20:] = -100
    labels[3, 0] = 4  # a supervised FIRST token counts in the unshifted count only
    st = hip.seq_ce_stats(logits, labels.to(dev)).cpu()
    lf = logits.float().cpu()
    ref = F.cross_entropy(lf[:, :-1].transpose(1, 2), labels[:, 1:], ignore_index=-100, reduction="none").sum(-1)
    assert torch.allclose(st[:, 0], ref, rtol=1e-5, atol=1e-4)
    assert st[:, 1].tolist() == (labels[:, 1:] != -100).sum(-1).float().tolist()
    assert st[:, 2].tolist() == (labels != -100).sum(-1).float().tolist()


def test_remove_image_tokens_device_path(dev):
    """bf16 device logits go through the row-gather kernel: same rows as the reference's slicing (golden I/O pair of
    mm/trainer_utils.py:48-70, V padded from 17 to 24 columns for 16-byte rows)."""
    import os

    from multimodal_unlearning_amd import trainer_utils as tu

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "trainer_utils.npz"))
    lg = torch.zeros(3, 9, 24)
    lg[..., :17] = torch.from_numpy(g["logits"])
    lgd = lg.to(dev).to(BF)
    ids = torch.from_numpy(g["ids"]).to(dev)
    out = tu.remove_image_tokens(ids, lgd, 16)
    exp = torch.from_numpy(g["rm_image"]).to(BF)
    assert out.shape == (3, 6, 24) and torch.equal(out[..., :17].cpu(), exp)
    # larger ragged case vs torch indexing
    B, T, L, V = 7, 40, 615, 256
    big = _randn((B, L, V), dev, 1.0, seed=8)
    ids = torch.randint(0, 90, (B, T), generator=torch.Generator().manual_seed(9))
    for b, c in ((0, 1), (2, 17), (3, 39), (5, 0)):
        ids[b, c] = 99
    got = tu.remove_image_tokens(ids.to(dev), big, 99)
    pos = tu.image_token_source_rows(ids, L, 99)
    assert torch.equal(got.cpu(), big.cpu()[torch.arange(B)[:, None], pos])


# ------------------------------------------------------------------------------------- SwiGLU inside the GEMM epilogues
@pytest.mark.parametrize("M,I,K,with_t", [
    (1000, 384, 512, True),     # ragged last tile row, three 128-unit groups (1.5 tile columns), one tile per workgroup
    (4104, 2048, 256, True),    # 17 x 16 = 272 tiles > 256 CUs: persistent launch, workgroups cross a tile seam
    (520, 128, 320, False),     # no transposed output, K not a multiple of 64
])
def test_gemm_swiglu_fwd_matches_gemm_plus_swiglu_kernel(dev, M, I, K, with_t):
    """mmu_gemm_swiglu_fwd (gate|up projection with SwiGLU in the 256^2 epilogue: B rows of a tile gathered from the gate AND
    the up half of W) == mmu_gemm_bf16 + mmu_swiglu_fwd_both, bit for bit: gu in its natural [g | u] order, act, act^T."""
    hip.gemm_set_tile(256)
    x, w = _randn((M, K), dev, 1.0, seed=1), _randn((2 * I, K), dev, 0.05, seed=2)
    gu0, act0 = torch.empty(M, 2 * I, device=dev, dtype=BF), torch.empty(M, I, device=dev, dtype=BF)
    actT0 = torch.empty(I, M, device=dev, dtype=BF)
    hip.gemm(hip.NT, x, w, gu0)
    hip.swiglu_fwd_both(gu0, act0, actT0)
    gu1, act1 = torch.full_like(gu0, 7.0), torch.full_like(act0, 7.0)
    actT1 = torch.full_like(actT0, 7.0) if with_t else None
    assert hip.gemm_swiglu_fwd(x, w, gu1, act1, actT1)
    torch.cuda.synchronize()
    assert torch.equal(gu0, gu1), (gu0 != gu1).nonzero()[:5].tolist()
    assert torch.equal(act0, act1), (act0 != act1).nonzero()[:5].tolist()
    if with_t:
        assert torch.equal(actT0, actT1), (actT0 != actT1).nonzero()[:5].tolist()
    # sanity against the formula in fp32 (silu(g) rounded to bf16, then * u; bf16 result)
    g, u = gu0[:, :I].float(), gu0[:, I:].float()
    ref = ((g * torch.sigmoid(g)).to(BF).float() * u).to(BF)
    assert _relerr(act1, ref) < 4e-3   # __expf vs torch.sigmoid: last-bit differences of the bf16 rounding of silu(g)


@pytest.mark.parametrize("M,I,K,w_t", [
    (1000, 384, 512, True),
    (4104, 4096, 256, True),    # 17 x 16 tiles: persistent
    (4104, 4096, 256, False),   # W_down [d, I] (no resident transposed copy): the NN layout of the same kernel
    (520, 136, 320, True),      # I not a multiple of 64: the last column group is partly out of range
])
def test_gemm_swiglu_bwd_matches_gemm_plus_swiglu_kernel(dev, M, I, K, w_t):
    """mmu_gemm_swiglu_bwd (down-projection dgrad whose output dact never leaves the registers) == mmu_gemm_bf16 +
    mmu_swiglu_bwd_t, bit for bit: dgu [M, 2I] and dgu^T."""
    hip.gemm_set_tile(256)
    dh, wd = _randn((M, K), dev, 1.0, seed=3), _randn((K, I), dev, 0.05, seed=4)   # W_down [d, I]
    gu = _randn((M, 2 * I), dev, 1.5, seed=5)
    w = wd.t().contiguous() if w_t else wd
    dact = torch.empty(M, I, device=dev, dtype=BF)
    hip.gemm(hip.NT if w_t else hip.NN, dh, w, dact)
    dgu0, dguT0 = torch.empty(M, 2 * I, device=dev, dtype=BF), torch.empty(2 * I, M, device=dev, dtype=BF)
    hip.swiglu_bwd_t(gu, dact, dgu0, dguT0)
    dgu1, dguT1 = torch.full_like(dgu0, 7.0), torch.full_like(dguT0, 7.0)
    assert hip.gemm_swiglu_bwd(dh, w, w_t, gu, dgu1, dguT1)
    dgu2 = torch.full_like(dgu0, 7.0)
    assert hip.gemm_swiglu_bwd(dh, w, w_t, gu, dgu2, None)   # without the transposed copy (no LDS staging, no barriers)
    torch.cuda.synchronize()
    assert torch.equal(dgu0, dgu1), (dgu0 != dgu1).nonzero()[:5].tolist()
    assert torch.equal(dguT0, dguT1), (dguT0 != dguT1).nonzero()[:5].tolist()
    assert torch.equal(dgu0, dgu2)


def test_gemm_swiglu_declines_shapes_off_the_256_kernel(dev):
    hip.gemm_set_tile(0)
    x, w = _randn((64, 256), dev, seed=1), _randn((256, 256), dev, seed=2)
    gu, act = torch.empty(64, 256, device=dev, dtype=BF), torch.empty(64, 128, device=dev, dtype=BF)
    assert hip.gemm_swiglu_fwd(x, w, gu, act, None) is False
    with pytest.raises(hip.HipKernelError):   # I must be a multiple of 128 for the gathered B rows
        hip.gemm_swiglu_fwd(x, _randn((208, 256), dev, seed=3), torch.empty(64, 208, device=dev, dtype=BF),
                            torch.empty(64, 104, device=dev, dtype=BF), None)


def test_layernorm_bwd_params_vs_torch(dev):
    """mmu_layernorm_bwd_params (dgamma / dbeta of a LayerNorm over [M, d] rows, for the unfrozen CLIP tower) vs autograd of
    torch.nn.functional.layer_norm on the same bf16 inputs in fp32; accumulate adds into an existing bf16 gradient."""
    for M, d in ((577 * 3, 1024), (10, 128), (130, 256)):
        x, dy = _randn((M, d), dev, 1.5, seed=1), _randn((M, d), dev, 0.7, seed=2)
        w = torch.ones(d, device=dev, requires_grad=True)
        b = torch.zeros(d, device=dev, requires_grad=True)
        y = torch.nn.functional.layer_norm(x.float(), (d,), w, b, 1e-5)
        y.backward(dy.float())
        dg = torch.zeros(d, device=dev, dtype=BF)
        db = torch.zeros(d, device=dev, dtype=BF)
        hip.layernorm_bwd_params(dy, x, dg, db, 1e-5, accumulate=False)
        assert _relerr(dg, w.grad) < 4e-3 and _relerr(db, b.grad) < 4e-3, (M, d, _relerr(dg, w.grad), _relerr(db, b.grad))
        hip.layernorm_bwd_params(dy, x, dg, db, 1e-5, accumulate=True)
        assert _relerr(dg, 2 * w.grad) < 8e-3 and _relerr(db, 2 * b.grad) < 8e-3


def test_dropout_kernel_mask_is_a_function_of_seed_and_index(dev):
    """mmu_dropout_bf16 (peft's lora_dropout): kept fraction ~ 1 - p, kept values scaled by 1 / (1 - p), the same seed gives the
    same mask (the backward pass regenerates it), another seed another mask, in place allowed, p = 0 is the identity."""
    n = 1 << 20
    x = torch.ones(n, device=dev, dtype=BF)
    for p in (0.05, 0.1, 0.5):
        a, b, c = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
        hip.dropout(x, a, p, 1234)
        hip.dropout(x, b, p, 1234)
        hip.dropout(x, c, p, 1235)
        assert torch.equal(a, b) and not torch.equal(a, c)
        kept = (a != 0).float().mean().item()
        assert abs(kept - (1 - p)) < 4 * (p * (1 - p) / n) ** 0.5 + 1e-4, (p, kept)
        assert torch.equal(a[a != 0], torch.full_like(a[a != 0], 1.0 / (1.0 - p)))
        assert abs((a != 0).float().mean().item() - (c != 0).float().mean().item()) < 5e-3
        both = ((a != 0) & (c != 0)).float().mean().item()                      # masks of two seeds are independent
        assert abs(both - (1 - p) ** 2) < 5e-3
    y = _randn((n,), dev, seed=3)
    z = y.clone()
    hip.dropout(z, z, 0.25, 99)
    ref = torch.empty_like(y)
    hip.dropout(y, ref, 0.25, 99)
    assert torch.equal(z, ref)
    hip.dropout(y, ref, 0.0, 5)
    assert torch.equal(ref, y)


@pytest.mark.parametrize("B,H,S,causal", [(2, 3, 136, True), (1, 2, 1088, True), (3, 1, 64, False)])
def test_attn_fwd_transposed_output(dev, B, H, S, causal):
    """mmu_attn_fwd_t: O^T [H * 128, B * S] written from the staging image of the row-major store == O.t(), bit for bit, and
    O / lse equal to the plain forward's (S multiples of 8 incl. a ragged last 32-row block)."""
    D = 128
    qkv = _randn((B * S, 3 * H * D), dev, 0.5, seed=7)
    q, k, v = qkv[:, :H * D], qkv[:, H * D:2 * H * D], qkv[:, 2 * H * D:]
    o0, o1 = torch.empty(B * S, H * D, device=dev, dtype=BF), torch.full((B * S, H * D), 7.0, device=dev, dtype=BF)
    oT = torch.full((H * D, B * S), 7.0, device=dev, dtype=BF)
    l0, l1 = torch.empty(B, H, S, device=dev), torch.empty(B, H, S, device=dev)
    hip.attn_fwd(q, k, v, o0, l0, None, B, S, H, D, D ** -0.5, causal)
    hip.attn_fwd(q, k, v, o1, l1, None, B, S, H, D, D ** -0.5, causal, out_t=oT)
    torch.cuda.synchronize()
    assert torch.equal(o0, o1) and torch.equal(l0, l1)
    assert torch.equal(oT, o0.t().contiguous())
    with pytest.raises(hip.HipKernelError):   # S not a multiple of 8: the transposed copy moves groups of 8 tokens
        hip.attn_fwd(q[: 2 * 36], k[: 2 * 36], v[: 2 * 36], o1[: 2 * 36], None, None, 2, 36 - 1, H, D, D ** -0.5, causal,
                     out_t=torch.empty(H * D, 2 * 35 + 2, device=dev, dtype=BF)[:, : 2 * 35])
