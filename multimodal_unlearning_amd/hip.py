"""ctypes binding of libmmu_hip.so (include/mmu_hip.h) + thin torch-tensor wrappers.

PyTorch is used only for device memory and the current HIP stream; every wrapper launches a hand-written
gfx950 kernel.  There is NO fallback: if the library is missing or a kernel rejects its arguments the
call raises (the product path must fail loudly rather than run on an eager/CPU substitute).
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libmmu_hip.so")

_vp, _i, _f, _ll = C.c_void_p, C.c_int, C.c_float, C.c_longlong

# name -> argtypes (restype is always int).  Kept in one table so tests can check every exported symbol
# of include/mmu_hip.h is bound.
SIGNATURES = {
    "mmu_abi_version": [],
    "mmu_gemm_bf16": [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "mmu_gemm_bf16_lora": [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp],
    "mmu_gemm_uses_256": [_i, _i, _i],
    "mmu_gemm_swiglu_fwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "mmu_gemm_swiglu_bwd": [_vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "mmu_gemm_bf16_rope": [_i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp],
    "mmu_gemm_bf16_splitk": [_i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "mmu_fold_f32_into_bf16": [_vp, _vp, _vp, _ll, _vp],
    "mmu_gemm_set_tile": [_i],
    "mmu_gemm_set_group_m": [_i],
    "mmu_gemm_set_persistent": [_i],
    "mmu_gemm_set_xcd_map": [_i],
    "mmu_gemm_set_tail_split": [_i],
    "mmu_gemm_set_workspace": [_vp, _ll],
    "mmu_gemm_set_xcd_skew": [_i],
    "mmu_gemm_set_tile_queue": [_vp, _ll],
    "mmu_gemm_reset_tile_queue": [],
    "mmu_attn_fwd": [_vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp],
    "mmu_attn_fwd_t": [_vp, _vp, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp],
    "mmu_attn_bwd": [_vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i,
                     _vp],
    "mmu_attn_bwd_rope": [_vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i,
                          _vp, _vp, _vp, _vp],
    "mmu_attn_bwd_rope_t": [_vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i,
                            _vp, _vp, _vp, _vp, _i, _vp],
    "mmu_attn_set_dkdv_variant": [_i],
    "mmu_attn_set_bwd_mid_event": [_vp],
    "mmu_rmsnorm_fwd": [_vp, _vp, _vp, _vp, _i, _i, _f, _vp],
    "mmu_rmsnorm_bwd_nparts": [_i],
    "mmu_rmsnorm_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "mmu_layernorm_fwd": [_vp, _vp, _vp, _vp, _i, _i, _f, _vp],
    "mmu_layernorm_bwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp],
    "mmu_layernorm_bwd_params": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp],
    "mmu_dropout_bf16": [_vp, _vp, _ll, _f, C.c_ulonglong, _vp],
    "mmu_quick_gelu_fwd": [_vp, _vp, _ll, _vp],
    "mmu_quick_gelu_bwd": [_vp, _vp, _vp, _ll, _vp],
    "mmu_head_pad": [_vp, _i, _i, _vp, _i, _i, _i, _i, _vp],
    "mmu_im2col_patch": [_vp, _vp, _i, _i, _i, _i, _i, _vp],
    "mmu_clip_normalize_u8": [_vp, _vp, _ll, _i, C.POINTER(C.c_float), C.POINTER(C.c_float), _vp],
    "mmu_clip_embed": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "mmu_rope_table": [_vp, _vp, _i, _i, _f, _vp],
    "mmu_rope": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "mmu_swiglu_fwd": [_vp, _vp, _i, _i, _vp],
    "mmu_swiglu_bwd": [_vp, _vp, _vp, _i, _i, _vp],
    "mmu_gelu_fwd": [_vp, _vp, _ll, _vp],
    "mmu_gelu_bwd": [_vp, _vp, _vp, _ll, _vp],
    "mmu_add_bf16": [_vp, _vp, _vp, _ll, _vp],
    "mmu_transpose_bf16": [_vp, _vp, _i, _i, _vp],
    "mmu_swiglu_fwd_t": [_vp, _vp, _i, _i, _vp],
    "mmu_swiglu_fwd_both": [_vp, _vp, _vp, _i, _i, _vp],
    "mmu_rmsnorm_apply_t": [_vp, _vp, _vp, _vp, _i, _i, _vp],
    "mmu_swiglu_bwd_t": [_vp, _vp, _vp, _vp, _i, _i, _vp],
    "mmu_colsum_nparts": [_i],
    "mmu_colsum_bf16": [_vp, _vp, _vp, _i, _i, _i, _vp],
    "mmu_image_slots": [_vp, _i, _ll, _vp, _vp, _vp],
    "mmu_embed_fwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "mmu_embed_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "mmu_ce_fwd": [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "mmu_ce_bwd": [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _vp],
    "mmu_seg_reduce": [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp],
    "mmu_rowloss_fwd": [_i, _vp, _vp, _i, _f, _vp, _i, _i, _vp],
    "mmu_rowloss_bwd": [_i, _vp, _vp, _i, _f, _vp, _vp, _vp, _i, _i, _i, _vp],
    "mmu_gather_rows_bf16": [_vp, _vp, _vp, _ll, _i, _i, _vp],
    "mmu_scatter_rows_bf16": [_vp, _vp, _vp, _ll, _i, _i, _vp],
    "mmu_rowloss_set_streaming": [_i],
    "mmu_seq_ce_stats": [_vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp],
    "mmu_logprobs_fwd": [_vp, _i, _vp, _vp, _i, _i, _i, _vp],
    "mmu_logprobs_bwd": [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "mmu_red_scratch_floats": [],
    "mmu_mse_fwd": [_vp, _vp, _vp, _ll, _i, _vp, _vp],
    "mmu_mse_bwd": [_vp, _vp, _vp, _ll, _i, _vp, _vp, _vp],
    "mmu_seqmean_nsplit": [_i],
    "mmu_seqmean_fwd": [_vp, _i, _i, _i, _vp, _vp, _vp],
    "mmu_rmu_pair_fwd": [_vp, _vp, _i, _i, _i, _vp, _vp],
    "mmu_rmu_pair_bwd": [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp],
    "mmu_l1l0_fwd": [_vp, _vp, _ll, _vp, _vp],
    "mmu_l1_bwd": [_vp, _vp, _ll, _vp, _f, _vp, _vp],
    "mmu_sumsq_bf16": [_vp, _ll, _vp, _vp],
    "mmu_sumsq_scratch_floats": [],
    "mmu_adamw_bf16": [_vp, _vp, _vp, _vp, _ll, _f, _f, _f, _f, _f, _i, _vp, _f, _f, _i, _vp, _i, _vp],
    "mmu_adamw8bit_nblocks": [_ll],
    "mmu_adamw8bit_bf16": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _f, _f, _f, _f, _f, _i, _vp, _f, _f, _i, _vp, _i,
                           _vp],
    "mmu_scale_bf16": [_vp, _ll, _f, _vp],
}

_lib = None


class HipKernelError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load libmmu_hip.so; raises if it has not been built (python -m multimodal_unlearning_amd.build)."""
    global _lib
    if _lib is None:
        global LIB_PATH
        LIB_PATH = os.environ.get("MMU_HIP_LIB", LIB_PATH)  # A/B builds of the same ABI (tools/build_variant.py)
        if not os.path.exists(LIB_PATH):
            raise HipKernelError(
                f"{LIB_PATH} not found: build it with `python -m multimodal_unlearning_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU / eager fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes = argtypes
            fn.restype = C.c_int
        _lib = lib
    return _lib


def _check(rc: int, name: str):
    if rc != 0:
        if name.startswith("mmu_gemm") and rc > 0 and _lib is not None:
            # a GEMM launch died: a tile-queue slot may be half drawn — clear the counters before anything else runs
            _lib.mmu_gemm_reset_tile_queue()
        raise HipKernelError(f"{name} failed with code {rc}" + (" (unsupported argument)" if rc == -1 else " (hipError)"))


def _p(t):
    return 0 if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _bf16c(*ts):
    for t in ts:
        if t is not None:
            assert t.dtype == torch.bfloat16 and t.is_cuda, (t.dtype, t.device)


# ------------------------------------------------------------------------------------------------ GEMM
NT, NN, TN = 0, 1, 2
ACT_NONE, ACT_GELU, ACT_QUICK_GELU = 0, 1, 2


def gemm(layout: int, A, B, out, *, bias=None, resid=None, act=0, accumulate=False, M=None, N=None, K=None,
         lora_u=None, lora_b=None):
    """out[M,N] = epi(Aop @ Bop^T (+ lora_u @ lora_b^T)).  2-D tensors with unit inner stride; leading dims from
    strides.  lora_u [M,kl], lora_b [N,kl] contiguous, kl in {32,64}: fused low-rank (LoRA) update."""
    _bf16c(A, B, out, bias, resid, lora_u, lora_b)
    assert A.stride(-1) == 1 and B.stride(-1) == 1 and out.stride(-1) == 1
    if layout == NT:
        m, k = A.shape
        n, k2 = B.shape
    elif layout == NN:
        m, k = A.shape
        k2, n = B.shape
    else:
        k, m = A.shape
        k2, n = B.shape
    assert k == k2, (A.shape, B.shape, layout)
    M = m if M is None else M
    N = n if N is None else N
    K = k if K is None else K
    assert out.shape[0] >= M and out.shape[1] >= N
    if lora_u is not None:
        kl = lora_u.shape[1]
        assert lora_u.is_contiguous() and lora_b.is_contiguous() and lora_b.shape[1] == kl
        assert lora_u.shape[0] >= M and lora_b.shape[0] >= N
        rc = load().mmu_gemm_bf16_lora(layout, _p(A), _p(B), _p(out), _p(bias), _p(resid), M, N, K, A.stride(0),
                                       B.stride(0), out.stride(0), resid.stride(0) if resid is not None else 0, act,
                                       int(accumulate), _p(lora_u), _p(lora_b), kl, _stream())
        _check(rc, "mmu_gemm_bf16_lora")
        return out
    rc = load().mmu_gemm_bf16(layout, _p(A), _p(B), _p(out), _p(bias), _p(resid), M, N, K, A.stride(0), B.stride(0),
                              out.stride(0), resid.stride(0) if resid is not None else 0, act, int(accumulate),
                              _stream())
    _check(rc, "mmu_gemm_bf16")
    return out


def gemm_rope(x, w, out, pos_ids, cos_t, sin_t, rope_cols, lora_u=None, lora_b=None) -> bool:
    """out = x w^T (+LoRA) with RoPE applied to columns [0, rope_cols) in the GEMM epilogue.  Returns False (nothing
    launched) when the shape would run on the 128^2 kernel: the caller then does gemm + rope."""
    _bf16c(x, w, out, cos_t, sin_t, lora_u, lora_b)
    M, K = x.shape
    N = w.shape[0]
    if not load().mmu_gemm_uses_256(M, N, K) or cos_t.shape[1] != 64:
        return False
    kl = 0 if lora_u is None else lora_u.shape[1]
    rc = load().mmu_gemm_bf16_rope(NT, _p(x), _p(w), _p(out), M, N, K, x.stride(0), w.stride(0), out.stride(0), _p(lora_u),
                                   _p(lora_b), kl, _p(pos_ids), _p(cos_t), _p(sin_t), rope_cols, _stream())
    _check(rc, "mmu_gemm_bf16_rope")
    return True


def gemm_swiglu_fwd(x, w_gu, gu, act, act_t=None) -> bool:
    """gu = x w_gu^T, act = bf16(silu(g)) * u (and act^T) in ONE launch of the 256^2 GEMM (fused epilogue).  Returns False
    (nothing launched) when the shape would run on the 128^2 kernel: the caller then does gemm + swiglu_fwd."""
    _bf16c(x, w_gu, gu, act, act_t)
    M, K = x.shape
    I = act.shape[1]
    assert w_gu.shape == (2 * I, K) and gu.shape == (M, 2 * I) and act.shape[0] == M
    assert act_t is None or (act_t.shape == (I, M) and act_t.stride(1) == 1)
    rc = load().mmu_gemm_swiglu_fwd(_p(x), _p(w_gu), _p(gu), _p(act), _p(act_t), M, I, K, x.stride(0), w_gu.stride(0),
                                    gu.stride(0), act.stride(0), 0 if act_t is None else act_t.stride(0), _stream())
    if rc == -2:
        return False
    _check(rc, "mmu_gemm_swiglu_fwd")
    return True


def gemm_swiglu_bwd(dh, w, w_is_transposed_copy, gu, dgu, dgu_t=None) -> bool:
    """dgu (and dgu^T) = SwiGLU backward of dact = dh W_down, dact never stored.  w: W_down^T [I, d] (resident transposed
    copy, w_is_transposed_copy=True) or W_down [d, I].  False = shape not on the 256^2 kernel, nothing launched."""
    _bf16c(dh, w, gu, dgu, dgu_t)
    M, K = dh.shape
    I = gu.shape[1] // 2
    assert w.shape == ((I, K) if w_is_transposed_copy else (K, I)) and dgu.shape == (M, 2 * I)
    assert dgu_t is None or (dgu_t.shape == (2 * I, M) and dgu_t.stride(1) == 1)
    rc = load().mmu_gemm_swiglu_bwd(_p(dh), _p(w), 0 if w_is_transposed_copy else 1, _p(gu), _p(dgu), _p(dgu_t), M, I, K,
                                    dh.stride(0), w.stride(0), gu.stride(0), dgu.stride(0),
                                    0 if dgu_t is None else dgu_t.stride(0), _stream())
    if rc == -2:
        return False
    _check(rc, "mmu_gemm_swiglu_bwd")
    return True


def gemm_splitk_accumulate(layout, A, B, c32, dst, mask=None, splits=16):
    """dst (bf16 [M,N]) += Aop @ Bop^T (* mask) through a split-K GEMM into the zeroed fp32 scratch c32 [M,N]."""
    _bf16c(A, B, dst, mask)
    assert c32.dtype == torch.float32 and c32.is_contiguous() and dst.is_contiguous()
    M, N = dst.shape
    K = A.shape[1] if layout != TN else A.shape[0]
    rc = load().mmu_gemm_bf16_splitk(layout, _p(A), _p(B), _p(c32), M, N, K, A.stride(0), B.stride(0), splits, _stream())
    _check(rc, "mmu_gemm_bf16_splitk")
    _check(load().mmu_fold_f32_into_bf16(_p(c32), _p(dst), _p(mask), M * N, _stream()), "mmu_fold_f32_into_bf16")


def gemm_set_group_m(g: int):
    _check(load().mmu_gemm_set_group_m(int(g)), "mmu_gemm_set_group_m")


def gemm_set_xcd_map(mode: int):
    _check(load().mmu_gemm_set_xcd_map(int(mode)), "mmu_gemm_set_xcd_map")


def gemm_set_persistent(mode: int):
    _check(load().mmu_gemm_set_persistent(int(mode)), "mmu_gemm_set_persistent")


def gemm_set_tail_split(mode: int):
    _check(load().mmu_gemm_set_tail_split(int(mode)), "mmu_gemm_set_tail_split")


_gemm_ws = None


def ensure_gemm_workspace(device):
    """Register the 64 MiB device workspace the split-K tail of the 256^2 GEMM needs and the 4 KiB of tile-queue
    counters of its persistent launch (one per process = per GPU)."""
    global _gemm_ws
    if _gemm_ws is None or _gemm_ws.device != torch.device(device):
        lib = load()
        lib.mmu_gemm_workspace_bytes.restype = C.c_longlong
        lib.mmu_gemm_workspace_bytes.argtypes = []
        n = int(lib.mmu_gemm_workspace_bytes())
        lib.mmu_gemm_tile_queue_bytes.restype = C.c_longlong
        lib.mmu_gemm_tile_queue_bytes.argtypes = []
        nq = int(lib.mmu_gemm_tile_queue_bytes())
        _gemm_ws = torch.zeros(n + nq, device=device, dtype=torch.uint8)  # [split-K partials | zeroed tile-queue counters]
        _check(lib.mmu_gemm_set_workspace(_p(_gemm_ws), n), "mmu_gemm_set_workspace")
        _check(lib.mmu_gemm_set_tile_queue(_gemm_ws.data_ptr() + n, nq), "mmu_gemm_set_tile_queue")
    return _gemm_ws


def gemm_tile_queue_bytes() -> int:
    lib = load()
    lib.mmu_gemm_tile_queue_bytes.restype = C.c_longlong
    lib.mmu_gemm_tile_queue_bytes.argtypes = []
    return int(lib.mmu_gemm_tile_queue_bytes())


def gemm_set_xcd_skew(units: int):
    _check(load().mmu_gemm_set_xcd_skew(int(units)), "mmu_gemm_set_xcd_skew")


def gemm_set_tile(tile: int):
    """0 = heuristic, 128 / 256 = force that tile configuration (process-wide)."""
    _check(load().mmu_gemm_set_tile(int(tile)), "mmu_gemm_set_tile")


# ------------------------------------------------------------------------------------------- attention
def attn_fwd(q, k, v, out, lse, kmask, B, S, H, head_dim, scale, causal, out_t=None):
    """q/k/v: 2-D views [B*S, H*head_dim] (row stride = ld, e.g. slices of a fused qkv buffer).  out_t (bf16 [H*head_dim,
    B*S], optional; head_dim 128, S % 8 == 0): also leave out^T, the o-projection's weight-gradient operand."""
    _bf16c(q, k, v, out)
    assert q.stride(0) == k.stride(0) == v.stride(0) and q.stride(1) == 1
    if out_t is not None:
        _bf16c(out_t)
        assert out_t.shape == (H * head_dim, B * S) and out_t.stride(1) == 1
        _check(load().mmu_attn_fwd_t(_p(q), _p(k), _p(v), q.stride(0), _p(out), out.stride(0), _p(out_t), out_t.stride(0),
                                     _p(lse), _p(kmask), B, S, H, head_dim, float(scale), int(causal), _stream()), "mmu_attn_fwd_t")
        return out
    rc = load().mmu_attn_fwd(_p(q), _p(k), _p(v), q.stride(0), _p(out), out.stride(0), _p(lse), _p(kmask), B, S, H,
                             head_dim, float(scale), int(causal), _stream())
    _check(rc, "mmu_attn_fwd")
    return out


def attn_bwd(q, k, v, o, lse, kmask, dout, delta, dq, dk, dv, B, S, H, head_dim, scale, causal, rope=None, dqkv_t=None):
    """rope = (pos_ids int32 [B*S], cos_t, sin_t [n_pos,64]): fuse the inverse RoPE rotation into the dq / dk stores.
    delta: fp32 workspace of 2 * B * H * S floats (delta, then lse * log2 e).
    dqkv_t (bf16 [3 * H * head_dim, B * S], optional): also leave [dq | dk | dv]^T, the A operand of the q|k|v weight
    gradient, without a separate transpose of the packed gradient."""
    _bf16c(q, k, v, o, dout, dq, dk, dv)
    assert dq.stride(0) == dk.stride(0) == dv.stride(0)
    assert delta.dtype == torch.float32 and delta.is_contiguous() and delta.numel() >= 2 * B * S * H
    pos, ct, st = rope if rope is not None else (None, None, None)
    args = (_p(q), _p(k), _p(v), q.stride(0), _p(o), o.stride(0), _p(lse), _p(kmask), _p(dout), dout.stride(0), _p(delta),
            _p(dq), _p(dk), _p(dv), dq.stride(0), B, S, H, head_dim, float(scale), int(causal), _p(pos), _p(ct), _p(st))
    if dqkv_t is None:
        _check(load().mmu_attn_bwd_rope(*args, _stream()), "mmu_attn_bwd_rope")
        return
    _bf16c(dqkv_t)
    assert dqkv_t.dim() == 2 and dqkv_t.shape[0] == 3 * H * head_dim and dqkv_t.shape[1] == B * S and dqkv_t.stride(1) == 1
    _check(load().mmu_attn_bwd_rope_t(*args, _p(dqkv_t), dqkv_t.stride(0), _stream()), "mmu_attn_bwd_rope_t")


DKDV_DEFAULT = 4


def attn_set_dkdv_variant(v: int, grid: int = 0):
    """8 = eight-wave dK/dV kernel; 4 = four-wave persistent register-resident kernel (head_dim 128) on `grid`
    workgroups (0 = one per CU, else a multiple of 8)."""
    _check(load().mmu_attn_set_dkdv_variant(int(v) | (int(grid) << 8)), "mmu_attn_set_dkdv_variant")


def attn_set_bwd_mid_event(event):
    """One-shot measurement hook: `event` (a torch.cuda.Event with timing, already recorded once so its handle exists) is
    re-recorded by the next attn_bwd between its dQ and its dK/dV kernel."""
    _check(load().mmu_attn_set_bwd_mid_event(0 if event is None else int(event.cuda_event)), "mmu_attn_set_bwd_mid_event")


# ------------------------------------------------------------------------------------------------ norms
def rmsnorm_fwd(x, w, y, rstd, eps):
    _bf16c(x, w, y)
    M, d = x.shape
    _check(load().mmu_rmsnorm_fwd(_p(x), _p(w), _p(y), _p(rstd), M, d, float(eps), _stream()), "mmu_rmsnorm_fwd")
    return y


def rmsnorm_bwd_nparts(M):
    return load().mmu_rmsnorm_bwd_nparts(M)


def rmsnorm_bwd(dy, x, w, rstd, dres, dx, dw_part, dw, accumulate_dw):
    _bf16c(dy, x, w, dres, dx, dw)
    M, d = x.shape
    _check(load().mmu_rmsnorm_bwd(_p(dy), _p(x), _p(w), _p(rstd), _p(dres), _p(dx), _p(dw_part), _p(dw),
                                  int(accumulate_dw), M, d, _stream()), "mmu_rmsnorm_bwd")
    return dx


def layernorm_fwd(x, w, b, y, eps):
    _bf16c(x, w, b, y)
    M, d = x.shape
    _check(load().mmu_layernorm_fwd(_p(x), _p(w), _p(b), _p(y), M, d, float(eps), _stream()), "mmu_layernorm_fwd")
    return y


def im2col_patch(pix, out, P):
    assert pix.dtype == torch.float32 and pix.is_cuda and pix.is_contiguous()
    B, Cc, Hh, Ww = pix.shape
    assert Hh == Ww
    _check(load().mmu_im2col_patch(_p(pix), _p(out), B, Cc, Hh, P, out.shape[1], _stream()), "mmu_im2col_patch")
    return out


def clip_normalize_u8(u8, out, mean, std):
    """uint8 [n,3,H,W] -> fp32 (x/255 - mean[c]) / std[c] (CLIP processor rescale + normalize) on the device."""
    assert u8.dtype == torch.uint8 and out.dtype == torch.float32 and u8.is_cuda and out.is_cuda
    assert u8.is_contiguous() and out.is_contiguous() and u8.shape == out.shape and u8.shape[1] == 3
    n, _, Hh, Ww = u8.shape
    m3, s3 = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
    _check(load().mmu_clip_normalize_u8(_p(u8), _p(out), n, Hh * Ww, m3, s3, _stream()), "mmu_clip_normalize_u8")
    return out


def clip_embed(patch, cls, pos, out, B, T, d):
    _bf16c(patch, cls, pos, out)
    _check(load().mmu_clip_embed(_p(patch), _p(cls), _p(pos), _p(out), B, T, d, _stream()), "mmu_clip_embed")
    return out


# ------------------------------------------------------------------------------------------------- rope
def rope_table(cos_t, sin_t, n_pos, head_dim, theta):
    _bf16c(cos_t, sin_t)
    _check(load().mmu_rope_table(_p(cos_t), _p(sin_t), n_pos, head_dim, float(theta), _stream()), "mmu_rope_table")


def rope(qkv, pos_ids, cos_t, sin_t, n_q_heads, n_k_heads, head_dim, k_off, inverse=False):
    _bf16c(qkv, cos_t, sin_t)
    assert pos_ids.dtype == torch.int32
    M = qkv.shape[0]
    _check(load().mmu_rope(_p(qkv), _p(pos_ids), _p(cos_t), _p(sin_t), M, qkv.stride(0), n_q_heads, n_k_heads,
                           head_dim, k_off, int(inverse), _stream()), "mmu_rope")
    return qkv


# ------------------------------------------------------------------------------------ swiglu / gelu / add
def swiglu_fwd(gu, h):
    _bf16c(gu, h)
    M, I2 = gu.shape
    _check(load().mmu_swiglu_fwd(_p(gu), _p(h), M, I2 // 2, _stream()), "mmu_swiglu_fwd")
    return h


def swiglu_bwd(gu, dh, dgu):
    _bf16c(gu, dh, dgu)
    M, I2 = gu.shape
    _check(load().mmu_swiglu_bwd(_p(gu), _p(dh), _p(dgu), M, I2 // 2, _stream()), "mmu_swiglu_bwd")
    return dgu


def gelu_fwd(x, y):
    _bf16c(x, y)
    _check(load().mmu_gelu_fwd(_p(x), _p(y), x.numel(), _stream()), "mmu_gelu_fwd")
    return y


def gelu_bwd(x, dy, dx):
    _bf16c(x, dy, dx)
    _check(load().mmu_gelu_bwd(_p(x), _p(dy), _p(dx), x.numel(), _stream()), "mmu_gelu_bwd")
    return dx


def layernorm_bwd(dy, x, w, dres, dx, eps):
    _bf16c(dy, x, w, dres, dx)
    M, d = x.shape
    _check(load().mmu_layernorm_bwd(_p(dy), _p(x), _p(w), _p(dres), _p(dx), M, d, float(eps), _stream()), "mmu_layernorm_bwd")
    return dx


def layernorm_bwd_params(dy, x, dgamma, dbeta, eps, accumulate=True):
    """dgamma (+)= sum_r dy * xhat, dbeta (+)= sum_r dy of a LayerNorm over rows [M, d] (unfrozen CLIP tower)."""
    _bf16c(dy, x, dgamma, dbeta)
    M, d = x.shape
    assert dy.shape == x.shape and dy.is_contiguous() and x.is_contiguous() and dgamma.numel() == d and dbeta.numel() == d
    stats = torch.empty(2 * M, device=x.device, dtype=torch.float32)
    part = torch.empty(2 * colsum_nparts(M) * d, device=x.device, dtype=torch.float32)
    _check(load().mmu_layernorm_bwd_params(_p(dy), _p(x), _p(stats), _p(part), _p(dgamma), _p(dbeta), int(accumulate), M, d,
                                           float(eps), _stream()), "mmu_layernorm_bwd_params")


def dropout(x, y, p, seed):
    """y = dropout(x, p) with the mask a hash of (seed, element index): the same call on a gradient applies the same mask."""
    _bf16c(x, y)
    assert x.is_contiguous() and y.is_contiguous() and x.numel() == y.numel()
    _check(load().mmu_dropout_bf16(_p(x), _p(y), x.numel(), float(p), int(seed) & (2 ** 64 - 1), _stream()), "mmu_dropout_bf16")
    return y


def quick_gelu_fwd(x, y):
    _bf16c(x, y)
    _check(load().mmu_quick_gelu_fwd(_p(x), _p(y), x.numel(), _stream()), "mmu_quick_gelu_fwd")
    return y


def quick_gelu_bwd(x, dy, dx):
    _bf16c(x, dy, dx)
    _check(load().mmu_quick_gelu_bwd(_p(x), _p(dy), _p(dx), x.numel(), _stream()), "mmu_quick_gelu_bwd")
    return dx


def head_pad(src, d_src, dst, d_dst, H):
    """dst [M, H*d_dst] <- src [M, H*d_src] per head, zero-padding (d_dst > d_src) or dropping the tail."""
    assert src.dtype == torch.bfloat16 and dst.dtype == torch.bfloat16 and src.stride(1) == 1 and dst.stride(1) == 1
    M = src.shape[0]
    _check(load().mmu_head_pad(_p(src), src.stride(0), d_src, _p(dst), dst.stride(0), d_dst, M, H, _stream()), "mmu_head_pad")
    return dst


def add(a, b, out):
    _bf16c(a, b, out)
    _check(load().mmu_add_bf16(_p(a), _p(b), _p(out), a.numel(), _stream()), "mmu_add_bf16")
    return out


def transpose(src, dst):
    _bf16c(src, dst)
    R, Cc = src.shape
    assert src.is_contiguous() and dst.is_contiguous() and tuple(dst.shape) == (Cc, R)
    _check(load().mmu_transpose_bf16(_p(src), _p(dst), R, Cc, _stream()), "mmu_transpose_bf16")
    return dst


def swiglu_fwd_t(gu, actT):
    _bf16c(gu, actT)
    M, I2 = gu.shape
    _check(load().mmu_swiglu_fwd_t(_p(gu), _p(actT), M, I2 // 2, _stream()), "mmu_swiglu_fwd_t")
    return actT


def swiglu_fwd_both(gu, act, actT):
    _bf16c(gu, act, actT)
    M, I2 = gu.shape
    _check(load().mmu_swiglu_fwd_both(_p(gu), _p(act), _p(actT), M, I2 // 2, _stream()), "mmu_swiglu_fwd_both")


def rmsnorm_apply_t(x, w, rstd, yT):
    _bf16c(x, w, yT)
    M, d = x.shape
    _check(load().mmu_rmsnorm_apply_t(_p(x), _p(w), _p(rstd), _p(yT), M, d, _stream()), "mmu_rmsnorm_apply_t")
    return yT


def swiglu_bwd_t(gu, dh, dgu, dguT):
    _bf16c(gu, dh, dgu, dguT)
    M, I2 = gu.shape
    _check(load().mmu_swiglu_bwd_t(_p(gu), _p(dh), _p(dgu), _p(dguT), M, I2 // 2, _stream()), "mmu_swiglu_bwd_t")


def colsum_nparts(M):
    return load().mmu_colsum_nparts(M)


def colsum(x, part, dst, accumulate):
    _bf16c(x, dst)
    M, d = x.shape
    _check(load().mmu_colsum_bf16(_p(x), _p(part), _p(dst), int(accumulate), M, d, _stream()), "mmu_colsum_bf16")
    return dst


# -------------------------------------------------------------------------------------------- embedding
def image_slots(ids, image_token, slot, count):
    assert ids.dtype == torch.int64 and ids.is_contiguous()
    _check(load().mmu_image_slots(_p(ids), ids.numel(), int(image_token), _p(slot), _p(count), _stream()),
           "mmu_image_slots")


def embed_fwd(ids, slot, table, img, out):
    _bf16c(table, img, out)
    M, d = out.shape
    _check(load().mmu_embed_fwd(_p(ids), _p(slot), _p(table), _p(img), _p(out), M, d, table.shape[0], _stream()),
           "mmu_embed_fwd")
    return out


def embed_bwd(ids, slot, dh, dimg, scratch, flag, grad_table, vocab):
    _bf16c(dh, dimg, grad_table)
    M, d = dh.shape
    _check(load().mmu_embed_bwd(_p(ids), _p(slot), _p(dh), _p(dimg), _p(scratch), _p(flag), _p(grad_table), M, d,
                                vocab, _stream()), "mmu_embed_bwd")


# ----------------------------------------------------------------------------------------------- losses
def ce_fwd(logits, labels, lse, loss_row, S, shift=True):
    _bf16c(logits)
    M, V = logits.shape
    _check(load().mmu_ce_fwd(_p(logits), logits.stride(0), _p(labels), _p(lse), _p(loss_row), M, V, S, int(shift),
                             _stream()), "mmu_ce_fwd")


def ce_bwd(logits, labels, lse, seg_scale, seg_cnt, seg_rows, normalize, dlogits, S, shift=True):
    _bf16c(logits, dlogits)
    M, V = logits.shape
    _check(load().mmu_ce_bwd(_p(logits), logits.stride(0), _p(labels), _p(lse), _p(seg_scale), _p(seg_cnt), seg_rows,
                             int(normalize), _p(dlogits), dlogits.stride(0), M, V, S, int(shift), _stream()),
           "mmu_ce_bwd")


def seg_reduce(vals, labels, seg_rows, M, V, S, shift, out_sum, out_cnt):
    _check(load().mmu_seg_reduce(_p(vals), _p(labels), seg_rows, M, V, S, int(shift), _p(out_sum), _p(out_cnt),
                                 _stream()), "mmu_seg_reduce")


KL_STUDENT_INPUT, KL_TEACHER_INPUT, NPO, ENTROPY, NPO_NLP = 0, 1, 2, 3, 4


def rowloss_fwd(mode, slog, tlog, beta, stats):
    _bf16c(slog, tlog)
    M, V = slog.shape
    _check(load().mmu_rowloss_fwd(mode, _p(slog), _p(tlog), slog.stride(0), float(beta), _p(stats), M, V, _stream()),
           "mmu_rowloss_fwd")


def rowloss_bwd(mode, slog, tlog, beta, stats, coef, dlogits):
    _bf16c(slog, tlog, dlogits)
    M, V = slog.shape
    _check(load().mmu_rowloss_bwd(mode, _p(slog), _p(tlog), slog.stride(0), float(beta), _p(stats), _p(coef),
                                  _p(dlogits), dlogits.stride(0), M, V, _stream()), "mmu_rowloss_bwd")


def gather_rows(src2d, rows, dst2d):
    """dst2d[r] = src2d[rows[r]] (bf16 [n, V] contiguous, rows int64 on the device)."""
    _bf16c(src2d, dst2d)
    assert src2d.is_contiguous() and dst2d.is_contiguous() and rows.dtype == torch.int64 and rows.is_cuda and rows.is_contiguous()
    assert rows.numel() == dst2d.shape[0] and src2d.shape[1] == dst2d.shape[1]
    _check(load().mmu_gather_rows_bf16(_p(src2d), _p(rows), _p(dst2d), src2d.shape[0], dst2d.shape[0], src2d.shape[1],
                                       _stream()), "mmu_gather_rows_bf16")
    return dst2d


def scatter_rows(src2d, rows, dst2d):
    """dst2d[rows[r]] = src2d[r] for rows[r] >= 0 (distinct rows; negative = padding slot, skipped)."""
    _bf16c(src2d, dst2d)
    assert src2d.is_contiguous() and dst2d.is_contiguous() and rows.dtype == torch.int64 and rows.is_cuda and rows.is_contiguous()
    assert rows.numel() == src2d.shape[0] and src2d.shape[1] == dst2d.shape[1]
    _check(load().mmu_scatter_rows_bf16(_p(src2d), _p(rows), _p(dst2d), dst2d.shape[0], src2d.shape[0], src2d.shape[1],
                                        _stream()), "mmu_scatter_rows_bf16")
    return dst2d


def rowloss_set_streaming(on: bool):
    _check(load().mmu_rowloss_set_streaming(int(bool(on))), "mmu_rowloss_set_streaming")


def seq_ce_stats(logits, labels):
    """logits [n, S, V] bf16, labels [n, S] int64 -> float32 [n, 3] = per-sequence {CE sum, #shifted targets, #labels != -100}
    (mm/eval.py:46-92: get_batch_loss and its token-count normaliser) in one C-ABI call."""
    _bf16c(logits)
    n, S, V = logits.shape
    assert logits.is_contiguous() and labels.dtype == torch.int64 and labels.is_contiguous() and labels.is_cuda
    scr = torch.empty(2, n * S, device=logits.device, dtype=torch.float32)
    out = torch.empty(n, 3, device=logits.device, dtype=torch.float32)
    _check(load().mmu_seq_ce_stats(_p(logits), V, _p(labels), n, S, V, _p(scr[0]), _p(scr[1]), _p(out), _stream()),
           "mmu_seq_ce_stats")
    return out


def logprobs_fwd(logits2d, out, lse, log=True):
    """out [M,V] fp32 = log_softmax / softmax of logits2d [M,V] bf16 (mm/trainer_utils.py:16-21), lse [M] fp32."""
    _bf16c(logits2d)
    M, V = logits2d.shape
    assert logits2d.is_contiguous() and out.dtype == torch.float32 and out.is_contiguous() and out.shape == (M, V)
    assert lse.dtype == torch.float32 and lse.numel() == M
    _check(load().mmu_logprobs_fwd(_p(logits2d), V, _p(out), _p(lse), M, V, int(bool(log)), _stream()), "mmu_logprobs_fwd")


def logprobs_bwd(logits2d, lse, g, dlogits, log=True):
    _bf16c(logits2d, dlogits)
    M, V = logits2d.shape
    assert g.dtype == torch.float32 and g.is_contiguous() and g.shape == (M, V) and dlogits.is_contiguous()
    _check(load().mmu_logprobs_bwd(_p(logits2d), V, _p(lse), _p(g), _p(dlogits), V, M, V, int(bool(log)), _stream()),
           "mmu_logprobs_bwd")


# ------------------------------------------------------------------------------------------------- RMU terms
RED_FLOATS = 1 + 2048  # result in slot 0, per-block partials after it (mmu_red_scratch_floats)


def mse_fwd(a, b, cvec, H, out):
    _bf16c(a, b)
    assert a.is_contiguous() and (b is None or b.is_contiguous()) and out.dtype == torch.float32 and out.numel() >= RED_FLOATS
    assert cvec is None or (cvec.dtype == torch.float32 and cvec.numel() == H and cvec.is_cuda)
    _check(load().mmu_mse_fwd(_p(a), _p(b), _p(cvec), a.numel(), H, _p(out), _stream()), "mmu_mse_fwd")


def mse_bwd(a, b, cvec, H, coef, da):
    _bf16c(a, b, da)
    assert coef.dtype == torch.float32 and coef.numel() == 1 and da.is_contiguous()
    _check(load().mmu_mse_bwd(_p(a), _p(b), _p(cvec), a.numel(), H, _p(coef), _p(da), _stream()), "mmu_mse_bwd")


def seqmean_fwd(a, mean):
    _bf16c(a)
    B, S, H = a.shape
    assert a.is_contiguous() and mean.dtype == torch.float32 and tuple(mean.shape) == (B, H)
    part = torch.empty(load().mmu_seqmean_nsplit(S) * B * H, device=a.device, dtype=torch.float32)
    _check(load().mmu_seqmean_fwd(_p(a), B, S, H, _p(part), _p(mean), _stream()), "mmu_seqmean_fwd")
    return mean


def rmu_pair_fwd(mean, ctrl, out):
    B, H = mean.shape
    assert ctrl.dtype == torch.float32 and ctrl.is_contiguous() and ctrl.shape[-1] == H and out.numel() >= RED_FLOATS
    _check(load().mmu_rmu_pair_fwd(_p(mean), _p(ctrl), B, ctrl.numel() // H, H, _p(out), _stream()), "mmu_rmu_pair_fwd")


def rmu_pair_bwd(mean, ctrl, S, coef, da):
    _bf16c(da)
    B, H = mean.shape
    _check(load().mmu_rmu_pair_bwd(_p(mean), _p(ctrl), B, ctrl.numel() // H, S, H, _p(coef), _p(da), _stream()),
           "mmu_rmu_pair_bwd")


# ---------------------------------------------------------------------------------------------- optimiser
def l1l0_fwd(p, t, out2):
    """out2 (float64[2], zeroed by the caller) += [sum |bf16(p - t)|, #(bf16(p - t) != 0)]; t may be None."""
    _bf16c(p, t)
    assert out2.dtype == torch.float64 and out2.numel() == 2 and (t is None or t.numel() == p.numel())
    _check(load().mmu_l1l0_fwd(_p(p), _p(t), p.numel(), _p(out2), _stream()), "mmu_l1l0_fwd")


def l1_bwd(p, t, g, lambda1, gout):
    _bf16c(p, t, g)
    assert gout is None or (gout.dtype == torch.float32 and gout.numel() == 1)
    _check(load().mmu_l1_bwd(_p(p), _p(t), p.numel(), _p(g), float(lambda1), _p(gout), _stream()), "mmu_l1_bwd")


SUMSQ_FLOATS = 1 + 2048  # out[0] = result, the rest is scratch for the per-block partials


def sumsq(g, out):
    """out[0] <- sum g^2 (deterministic two-pass reduction); out: float32, >= SUMSQ_FLOATS elements."""
    _bf16c(g)
    assert out.dtype == torch.float32 and out.numel() >= SUMSQ_FLOATS and out.is_contiguous()
    _check(load().mmu_sumsq_bf16(_p(g), g.numel(), _p(out), _stream()), "mmu_sumsq_bf16")


def _nodecay_args(nodecay):
    """nodecay: int64 device tensor [k, 2] of sorted, disjoint [lo, hi) element ranges without weight decay, or None."""
    if nodecay is None or nodecay.numel() == 0:
        return 0, 0
    assert nodecay.dtype == torch.int64 and nodecay.is_cuda and nodecay.is_contiguous() and nodecay.shape[-1] == 2
    return nodecay.data_ptr(), nodecay.shape[0]


def adamw(p, g, m, v, lr, beta1, beta2, eps, wd, step, sumsq_t, max_norm, grad_scale=1.0, zero_grad=True, nodecay=None):
    _bf16c(p, g)
    nd, nnd = _nodecay_args(nodecay)
    _check(load().mmu_adamw_bf16(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, wd, step,
                                 _p(sumsq_t), float(max_norm), float(grad_scale), int(zero_grad), nd, nnd, _stream()),
           "mmu_adamw_bf16")


def dynamic_map(signed: bool, max_exponent_bits: int = 7, total_bits: int = 8) -> torch.Tensor:
    """The 256-entry dynamic (tree) quantisation map of block-wise 8-bit optimizers (Dettmers et al. 2021; bitsandbytes
    `create_dynamic_map`): for exponent i the interval [0.1, 1] * 10^(i - 6) is cut into 2^i (signed) or 2^(i+1)
    (unsigned) equal pieces whose mid-points are the codes, plus 0 and 1; sorted ascending."""
    non_sign_bits = total_bits - 1
    data = []
    for i in range(max_exponent_bits):
        items = (2 ** (i + non_sign_bits - max_exponent_bits) + 1) if signed else (2 ** (i + non_sign_bits - max_exponent_bits + 1) + 1)
        b = torch.linspace(0.1, 1, items)
        means = (b[:-1] + b[1:]) / 2.0
        vals = (10.0 ** (-(max_exponent_bits - 1) + i)) * means
        data += vals.tolist()
        if signed:
            data += (-vals).tolist()
    data += [0.0, 1.0]
    assert len(data) == 2 ** total_bits, len(data)
    return torch.tensor(sorted(data), dtype=torch.float32)


def adamw8bit_nblocks(n: int) -> int:
    return load().mmu_adamw8bit_nblocks(int(n))


def adamw8bit(p, g, codes1, codes2, absmax1, absmax2, map1, map2, lr, beta1, beta2, eps, wd, step, sumsq_t, max_norm,
              grad_scale=1.0, zero_grad=True, nodecay=None):
    _bf16c(p, g)
    assert codes1.dtype == torch.uint8 and codes2.dtype == torch.uint8 and codes1.numel() == p.numel() == codes2.numel()
    assert absmax1.dtype == torch.float32 and absmax1.numel() >= adamw8bit_nblocks(p.numel()) == adamw8bit_nblocks(p.numel())
    assert map1.numel() == 256 and map2.numel() == 256 and map1.is_cuda and map2.is_cuda
    nd, nnd = _nodecay_args(nodecay)
    _check(load().mmu_adamw8bit_bf16(_p(p), _p(g), _p(codes1), _p(codes2), _p(absmax1), _p(absmax2), _p(map1), _p(map2),
                                     p.numel(), lr, beta1, beta2, eps, wd, step, _p(sumsq_t), float(max_norm),
                                     float(grad_scale), int(zero_grad), nd, nnd, _stream()), "mmu_adamw8bit_bf16")


def scale_(x, s):
    _bf16c(x)
    _check(load().mmu_scale_bf16(_p(x), x.numel(), float(s), _stream()), "mmu_scale_bf16")
