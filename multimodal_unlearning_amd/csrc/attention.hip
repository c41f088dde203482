// Fused (flash-style) attention forward + backward for gfx950, bf16 in / fp32 accumulate / bf16 out.
// Replaces the flash-attn wheel the reference enables with attn_implementation="flash_attention_2"
// (mm/forget.py:86,98,106; config/model_config.yaml:61) for the Llama decoder (causal, head_dim 128)
// and the CLIP tower (non-causal, head_dim 64).  SURVEY.md §8a rows K3, K10.
//
// MFMA: v_mfma_f32_32x32x16_bf16.  Operand maps (guide §3): A frag lane l = A[row l&31][k 8(l>>5)+j],
// B frag lane l = B[k 8(l>>5)+j][col l&31], D: col = l&31, row = (reg&3)+8(reg>>2)+4(l>>5).
//
// Forward / dQ kernels ("query on the lane"): S^T = K Q^T puts one query per lane and 16 keys in the
// registers of each half-wave, so the softmax statistics are lane-local and the S^T accumulator is, after
// bf16 packing, directly the B operand of the next product (O^T = V^T P^T, dQ^T = K^T dS^T) whose A operand
// is read transposed from the row-major LDS tile with ds_read_b64_tr_b16.
// dK/dV kernel ("key on the lane"): S = Q K^T, dP = dO V^T with K, V fragments resident in registers;
// P and dS feed dV^T += dO^T P and dK^T += Q^T dS the same way.
//
// LDS tiles are [rows][D] row-major, lane-linear as LDS-DMA writes them, XOR-swizzled on the per-lane
// source address and on the reads.  D=128 (256-B rows): chunk ^= ((row&3)<<2)|((row>>2)&3), which is
// conflict-free for both ds_read_b128 row reads and transposed reads (guide T10 image (b)).
#include "attention_common.h"

#ifndef MMU_NO_KFLAG  // A/B only: 1 = a non-null key mask always takes the masked path (the behaviour before round 2)
#define MMU_NO_KFLAG 0
#endif

namespace {

// ------------------------------------------------------------------------------------------------
// Forward. grid (ceil(S/128), H, B), 256 threads; wave w owns query rows qb*128 + 32w .. +31.
// LDS: 2 x (K tile 64xD | V tile 64xD) + 2 x 64 floats key bias.
// ------------------------------------------------------------------------------------------------

template <int D, bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TB = 64 * D * 2;  // bytes per 64-row tile
    constexpr int NKS = D / 16, NDT = D / 32;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hl = lane >> 5;
    int qb, hh, b;
    decode_wg((a.S + 127) / 128, a.H, a.B, true, qb, hh, b);
    const int S = a.S;
    const size_t rowbase = (size_t)b * S;
    const bf16* qp = a.q + rowbase * a.ld + hh * D;
    const bf16* kp = a.k + rowbase * a.ld + hh * D;
    const bf16* vp = a.v + rowbase * a.ld + hh * D;
    float* kbias = reinterpret_cast<float*>(smem + 4 * TB);  // [2][64]
    int* kflag = reinterpret_cast<int*>(smem + 4 * TB + 2 * 64 * 4);  // [2]: every key of the staged block is valid

    const int q0 = qb * 128 + wave * 32;
    const int qrow = q0 + (lane & 31);
    const bool qvalid = qrow < S;
    bf16x8 qf[NKS];
    {
        const bf16* qsrc = qp + (size_t)(qvalid ? qrow : S - 1) * a.ld + 8 * hl;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qsrc + 16 * ks);
    }
    FragOffsets<D, KIND_ROW> fo_k;
    fo_k.init(lane);
    FragOffsets<D, KIND_TR> fo_v;
    fo_v.init(lane);
    f32x16 o[NDT];
#pragma unroll
    for (int i = 0; i < NDT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m = -INFINITY, lsum = 0.f;
    const float c = a.scale * LOG2E;

    const int kv_end = CAUSAL ? min(S, qb * 128 + 128) : S;
    const int nblk = (kv_end + 63) / 64;

    RowStager<D, KIND_ROW, 64> st_k;
    RowStager<D, KIND_TR, 64> st_v;
    st_k.init(a.ld, wave, lane);
    st_v.init(a.ld, wave, lane);
    // The staging wave fetches the key-mask word of a block ONE stage call ahead, and before that call's LDS-DMA: loaded
    // at its point of use, the word's s_waitcnt vmcnt(0) also waited for the DMA just issued (vmcnt retires in order), a
    // full memory latency in front of every block of a padded batch (565 vs 502 us, profiles/r03_attn_fwd8_experiment.txt).
    auto load_mask = [&](int kb) -> int {
        const int key = kb * 64 + (int)threadIdx.x;
        return (threadIdx.x < 64 && a.kmask != nullptr && key < S) ? a.kmask[rowbase + key] : 1;
    };
    int mk_pre = load_mask(0);
    auto stage = [&](int kb, int buf) {
        char* base = smem + buf * 2 * TB;
        const int mk = mk_pre;
        mk_pre = load_mask(kb + 1);
        st_k.issue(base, kp, a.ld, kb * 64, S, wave);
        st_v.issue(base + TB, vp, a.ld, kb * 64, S, wave);
        // the bias row is read by masked blocks only (padding, tail, or the causal diagonal of some wave of this tile)
        if (threadIdx.x < 64 && (a.kmask != nullptr || kb * 64 + 64 > S || (CAUSAL && kb * 64 + 63 > qb * 128))) {
            const int key = kb * 64 + threadIdx.x;
            const bool ok = key < S && mk != 0;
            kbias[buf * 64 + threadIdx.x] = ok ? 0.f : -INFINITY;
            // a key mask that masks nothing in this block (the step passes device-resident masks it cannot inspect on the
            // host: all ones without padding) must not cost the masked path: wave 0 leaves "all 64 keys valid" here
            const bool all_ok = __all(ok);
            if (threadIdx.x == 0) kflag[buf] = all_ok ? 1 : 0;
        }
    };

    // One KV block (64 keys).  MASKED = false is the fast path for blocks that need no causal / padding / tail mask:
    // no compares, no bias loads, and exp2(fma(s, c, -m)) instead of scale + bias + subtract.
#ifdef MMU_ATTN_STAMP
    unsigned long long fs_wait = 0, fs_stage = 0, fs_qk = 0, fs_sm = 0, fs_pv = 0, fs_masked = 0;
    int fs_blocks = 0;
#endif
    auto block = [&](const bool masked, int kb, int buf) {
        FSTAMP(fb0);
        // fo_k / fo_v already point into buffer `buf` (toggled in place by the caller): tile bases are immediates
        const MMU_LDS char* kt = (const MMU_LDS char*)smem;
        const MMU_LDS char* vt = kt + TB;
        const MMU_LDS float* kbs = (const MMU_LDS float*)(smem + 4 * TB) + buf * 64;
        f32x16 s[2];
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        {
            // K fragments in batches of 4, software-pipelined: batch i+1 is requested before the MFMAs of batch i issue,
            // so its LDS latency hides behind 4 MFMAs instead of stalling the chain (load -> wait -> MFMA per batch).
            constexpr int NB = 2 * NKS / 4;  // batches over (st, h4)
            bf16x8 kf[2][4];
            auto load_k = [&](int bi, bf16x8 (&dst)[4]) {
                const int st = bi / (NKS / 4), h4 = (bi % (NKS / 4)) * 4;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) dst[ks] = *(const MMU_LDS bf16x8*)(kt + st * 32 * (D * 2) + fo_k.row[h4 + ks]);
            };
            load_k(0, kf[0]);
#pragma unroll
            for (int bi = 0; bi < NB; ++bi) {
                if (bi + 1 < NB) load_k(bi + 1, kf[(bi + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                const int st = bi / (NKS / 4), h4 = (bi % (NKS / 4)) * 4;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)  // the chain starts from the inline constant 0
                    s[st] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[bi & 1][ks], qf[h4 + ks],
                                                                    (h4 + ks == 0) ? zero16 : s[st], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // Only the softmax statistics differ between the masked and the mask-free block; the O rescale and the P V
        // product below are shared code, so the 64 O accumulators stay in one set of registers (two instantiations of
        // the whole block made the compiler copy all 64 at the merge: 64 v_mov per block).
        float mx = -INFINITY;
        if (masked) {
            const bool diag = CAUSAL && (kb * 64 + 63 > q0);
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int kl = st * 32 + 8 * g + 4 * hl;  // local key of reg 4g
                    const f32x4 kb4 = *(const MMU_LDS f32x4*)(kbs + kl);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float x = s[st][4 * g + j] * c + kb4[j];
                        if (diag && (kb * 64 + kl + j > qrow)) x = -INFINITY;
                        s[st][4 * g + j] = x;
                        mx = fmaxf(mx, x);
                    }
                }
        } else {
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[st][r]);
            mx *= c;  // c > 0: max commutes with the scale
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        FSTAMP(fb1);
        const float m_new = fmaxf(m, mx);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = fast_exp2(m - m_use);
        float rs = 0.f;
        if (masked) {
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = fast_exp2(s[st][r] - m_use);
                    s[st][r] = p;
                    rs += p;
                }
        } else {
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = fast_exp2(__builtin_fmaf(s[st][r], c, -m_use));
                    s[st][r] = p;
                    rs += p;
                }
        }
        lsum = lsum * alpha + rs;
        m = m_new;
#pragma unroll
        for (int i = 0; i < NDT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
        FSTAMP(fb2);
        // O^T[d][q] += V^T[d][key] P^T[key][q]
        {
            // V^T fragments: one (key half, 16-key step) combination = NDT fragments per batch, software-pipelined like
            // the K fragments (the 2-fragment batches with a barrier in between cost one LDS round trip per 2 MFMAs).
            bf16x8 vf[2][NDT];
            auto load_v = [&](int ci, bf16x8 (&dst)[NDT]) {
                const int st = ci >> 1, sk = ci & 1;
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt)
                    dst[dt] = tr_load(vt + (st * 32 + sk * 16) * (D * 2) + fo_v.tr1[dt],
                                      vt + (st * 32 + sk * 16) * (D * 2) + fo_v.tr2[dt]);
            };
            load_v(0, vf[0]);
#pragma unroll
            for (int ci = 0; ci < 4; ++ci) {
                if (ci + 1 < 4) load_v(ci + 1, vf[(ci + 1) & 1]);
                const bf16x8 pf = pack8(s[ci >> 1], ci & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt)
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[ci & 1][dt], pf, o[dt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#ifdef MMU_ATTN_STAMP
        {
            const unsigned long long fb3 = stamp_now();
            fs_qk += fb1 - fb0; fs_sm += fb2 - fb1; fs_pv += fb3 - fb2;
            fs_masked += masked ? 1 : 0;
            ++fs_blocks;
        }
#endif
    };
    FSTAMP(f_start);
    stage(0, 0);
    for (int kb = 0; kb < nblk; ++kb) {
        const int cur = kb & 1;
        FSTAMP(fl0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        FSTAMP(fl1);
        if (kb + 1 < nblk) stage(kb + 1, cur ^ 1);
        if (kb > 0) {
            fo_k.toggle_rows(2 * TB);
            fo_v.toggle_tr(2 * TB);
        }
#ifdef MMU_ATTN_STAMP
        {
            const unsigned long long fl2 = stamp_now();
            fs_wait += fl1 - fl0; fs_stage += fl2 - fl1;
        }
#endif
        const bool active = !CAUSAL || (kb * 64 <= q0 + 31);
        if (active) {
            // wave-uniform: does any element of this block need a mask?
            const bool masked = (CAUSAL && (kb * 64 + 63 > q0)) || (kb * 64 + 64 > S) ||
                                (a.kmask != nullptr && (MMU_NO_KFLAG || ((const MMU_LDS int*)kflag)[cur] == 0));
            block(masked, kb, cur);
        }
    }
#ifdef MMU_ATTN_STAMP
    {
        const unsigned long long f_end = stamp_now();
        if (threadIdx.x == 192 && blockIdx.x < 16384) {
            unsigned long long* w = g_fwd_stamps + 8 * (size_t)blockIdx.x;
            w[0] = fs_blocks; w[1] = fs_wait; w[2] = fs_stage; w[3] = fs_qk; w[4] = fs_sm; w[5] = fs_pv; w[6] = f_end - f_start;
            w[7] = fs_masked;
        }
    }
#endif
    const float ltot = lsum + __shfl_xor(lsum, 32, 64);
    const float inv = ltot > 0.f ? 1.f / ltot : 0.f;
    if constexpr (D == 128) {
        // O leaves through LDS like the gradients of the backward kernels (store_tile32x128_via_lds: whole 128-B lines
        // instead of 32 rows x 32 B per store instruction); the K / V buffers are free once every wave is out of the loop.
        __syncthreads();
        if (q0 < S) {
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= inv;
            u32x2 w[4][4];
            grad_words(o, hl, nullptr, nullptr, w);
            // a.dT (forward, round 4): also leave O^T [H * D][ldt] — the A operand of the o-projection's weight gradient —
            // read column-wise from the same staging image (mmu_attn_fwd_t; sequence lengths that are multiples of 8)
            store_tile32x128_via_lds<false>((MMU_LDS char*)smem + wave * 4096, w, lane, a.o + (rowbase + q0) * a.ldo + hh * D,
                                            a.ldo, a.dT ? a.dT + (size_t)(hh * D) * a.ldt + rowbase + q0 : nullptr, a.ldt,
                                            min(32, S - q0));
        }
        if (qvalid && hl == 0 && a.lse)
            a.lse[((size_t)b * a.H + hh) * S + qrow] = ltot > 0.f ? (m + __log2f(ltot)) * LN2 : INFINITY;
        return;
    }
    if (qvalid) {
        bf16* op = a.o + (rowbase + qrow) * a.ldo + hh * D;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 t;
#pragma unroll
                for (int j = 0; j < 4; ++j) t[j] = f2bf(o[dt][4 * g + j] * inv);
                *reinterpret_cast<bf16x4*>(op + dt * 32 + 8 * g + 4 * hl) = t;
            }
        if (hl == 0 && a.lse)
            a.lse[((size_t)b * a.H + hh) * S + qrow] = ltot > 0.f ? (m + __log2f(ltot)) * LN2 : INFINITY;
    }
}

// A/B builds only (-DMMU_ATTN_DELTA_PASS); the default build forms both rows in the dQ kernel (launched first).
// delta[b][h][s] = sum_d dO[b,s,h,d] * O[b,s,h,d]: D/8 lanes per (row, head), 16-B loads, shuffle reduce
// Also writes lse2[b][h][s] = lse * log2(e) behind delta (the workspace holds 2 * B*H*S floats): the backward kernels
// then form p = exp2(fma(s, c, -lse2)) with one instruction per element instead of a multiply and an fma.
template <int D>
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16* __restrict__ o, int ldo,
                                                         const bf16* __restrict__ dout, int lddo,
                                                         float* __restrict__ delta, const float* __restrict__ lse, int B,
                                                         int S, int H) {
    constexpr int LPG = D / 8;  // lanes per (row, head) group
    const size_t gid = ((size_t)blockIdx.x * 256 + threadIdx.x) / LPG;
    const int sub = threadIdx.x % LPG;
    const size_t total = (size_t)B * S * H;
    float acc = 0.f;
    size_t row = 0;
    int h = 0;
    if (gid < total) {
        h = (int)(gid % H);
        row = gid / H;
        const bf16x8 ov = *reinterpret_cast<const bf16x8*>(o + row * ldo + h * D + sub * 8);
        const bf16x8 dv = *reinterpret_cast<const bf16x8*>(dout + row * lddo + h * D + sub * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += bf2f(ov[j]) * bf2f(dv[j]);
    }
#pragma unroll
    for (int off = LPG / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (gid < total && sub == 0) {
        const int b = (int)(row / S), sq = (int)(row % S);
        const size_t i = ((size_t)b * H + h) * S + sq;
        delta[i] = acc;
        delta[total + i] = lse[i] * LOG2E;
    }
}

// ------------------------------------------------------------------------------------------------
// dQ. Same geometry as forward. dQ^T[d][q] = sum_key K^T[d][key] dS^T[key][q],
// dS^T = P^T * (dP^T - delta[q]) * scale, P^T = exp2(S^T c - lse2[q]), dP^T = V dO^T.
// LDS: 2 x (K tile (row + transposed reads) | V tile (row reads)) + key bias.
// ------------------------------------------------------------------------------------------------
template <int D, bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TB = 64 * D * 2;
    constexpr int NKS = D / 16, NDT = D / 32;
    FSTAMP(q_start);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hl = lane >> 5;
    int qb, hh, b;
    decode_wg((a.S + 127) / 128, a.H, a.B, true, qb, hh, b);
    const int S = a.S;
    const size_t rowbase = (size_t)b * S;
    const bf16* qp = a.q + rowbase * a.ld + hh * D;
    const bf16* kp = a.k + rowbase * a.ld + hh * D;
    const bf16* vp = a.v + rowbase * a.ld + hh * D;
    const bf16* dop = a.dout + rowbase * a.lddo + hh * D;
    float* kbias = reinterpret_cast<float*>(smem + 4 * TB);
    int* kflag = reinterpret_cast<int*>(smem + 4 * TB + 2 * 64 * 4);  // [2]: every key of the staged block is valid

    const int q0 = qb * 128 + wave * 32;
    const int qrow = q0 + (lane & 31);
    const bool qvalid = qrow < S;
    bf16x8 qf[NKS], dof[NKS];
    {
        const size_t rr = (size_t)(qvalid ? qrow : S - 1);
        const bf16* qsrc = qp + rr * a.ld + 8 * hl;
        const bf16* dsrc = dop + rr * a.lddo + 8 * hl;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            qf[ks] = *reinterpret_cast<const bf16x8*>(qsrc + 16 * ks);
            dof[ks] = *reinterpret_cast<const bf16x8*>(dsrc + 16 * ks);
        }
    }
    const size_t sidx = ((size_t)b * a.H + hh) * S + (qvalid ? qrow : S - 1);
    // delta / lse2 (see below): the O fragments are requested here, next to Q and dO, and consumed behind the first
    // staging call, so that the K / V request of block 0 does not wait for them
    bf16x8 of[NKS];
    float lse_row = 0.f;
    if (a.fuse_delta) {
        const bf16* osrc = a.o + (rowbase + (size_t)(qvalid ? qrow : S - 1)) * a.ldo + hh * D + 8 * hl;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) of[ks] = *reinterpret_cast<const bf16x8*>(osrc + 16 * ks);
        lse_row = a.lse[sidx];
    }
    FragOffsets<D, KIND_ROW> fo;
    fo.init(lane);
    f32x16 acc[NDT];
#pragma unroll
    for (int i = 0; i < NDT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const float c = a.scale * LOG2E;

    const int kv_end = CAUSAL ? min(S, qb * 128 + 128) : S;
    const int nblk = (kv_end + 63) / 64;

    // K and V tiles share one set of precomputed per-lane offsets (same row stride, same row-read swizzle)
    RowStager<D, KIND_ROW, 64> st_kv;
    st_kv.init(a.ld, wave, lane);
    auto load_mask = [&](int kb) -> int {  // one stage call ahead and in front of the LDS-DMA: see attn_fwd_kernel
        const int key = kb * 64 + (int)threadIdx.x;
        return (threadIdx.x < 64 && a.kmask != nullptr && key < S) ? a.kmask[rowbase + key] : 1;
    };
    int mk_pre = load_mask(0);
    auto stage = [&](int kb, int buf) {
        char* base = smem + buf * 2 * TB;
        const int mk = mk_pre;
        mk_pre = load_mask(kb + 1);
        st_kv.issue(base, kp, a.ld, kb * 64, S, wave);
        st_kv.issue(base + TB, vp, a.ld, kb * 64, S, wave);
        if (threadIdx.x < 64) {
            const int key = kb * 64 + threadIdx.x;
            const bool ok = key < S && mk != 0;
            kbias[buf * 64 + threadIdx.x] = ok ? 0.f : -INFINITY;
            const bool all_ok = __all(ok);  // see attn_fwd_kernel: a mask that masks nothing here takes the plain path
            if (threadIdx.x == 0) kflag[buf] = all_ok ? 1 : 0;
        }
    };

    stage(0, 0);
    float lse2, dlt;
    if (a.fuse_delta) {
        // delta = rowsum(dO o O) of this lane's row from the dO fragments already held (the two half-waves own the two
        // 8-column halves of every 16-column step), published with lse * log2(e) for the dK/dV kernel that runs after
        float part = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) part += bf2f(of[ks][j]) * bf2f(dof[ks][j]);
        dlt = part + __shfl_xor(part, 32, 64);
        const float l2 = lse_row * LOG2E;
        lse2 = qvalid ? l2 : INFINITY;
        if (qvalid && hl == 0) {
            float* w = const_cast<float*>(a.delta);
            w[sidx] = dlt;
            w[(size_t)a.B * a.H * S + sidx] = l2;
        }
    } else {
        lse2 = qvalid ? a.delta[(size_t)a.B * a.H * S + sidx] : INFINITY;  // lse * log2(e), from the delta kernel
        dlt = a.delta[sidx];
    }
#ifdef MMU_ATTN_STAMP
    unsigned long long qs_wait = 0, qs_stage = 0, qs_comp = 0;
#endif
    FSTAMP(q_loop);
    for (int kb = 0; kb < nblk; ++kb) {
        const int cur = kb & 1;
        FSTAMP(qb0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        FSTAMP(qb1);
        const bool keys_ok = ((const MMU_LDS int*)kflag)[cur] != 0;  // read before the next stage call rewrites the other slot
        if (kb + 1 < nblk) stage(kb + 1, cur ^ 1);
        if (kb > 0) {  // the fragment offsets follow the buffer (in-place XOR; tile bases are immediates below)
            fo.toggle_rows(2 * TB);
            fo.toggle_tr(2 * TB);
        }
        FSTAMP(qb2);
        const bool active = !CAUSAL || (kb * 64 <= q0 + 31);
        if (active) {
            const MMU_LDS char* kt = (const MMU_LDS char*)smem;
            const MMU_LDS char* vt = kt + TB;
            const MMU_LDS float* kbs = (const MMU_LDS float*)(smem + 4 * TB) + cur * 64;
            const bool diag = CAUSAL && (kb * 64 + 63 > q0);
            // wave-uniform: a block below the diagonal with 64 real, unmasked keys needs no bias, compare or select
            const bool plain = !diag && keys_ok && !(MMU_NO_KFLAG && a.kmask != nullptr);  // (keys_ok: sequence end + key mask)
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                f32x16 s, dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
                for (int h4 = 0; h4 < NKS; h4 += 4) {
                    // 4 K and 4 V fragments per batch (32 registers): all 8 + 8 at once pushed the kernel past 256
                    // VGPRs, and scratch reloads inside this loop queue behind the LDS-DMA on vmcnt
                    bf16x8 kf[4], vf[4];
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        kf[ks] = *(const MMU_LDS bf16x8*)(kt + st * 32 * (D * 2) + fo.row[h4 + ks]);
                        vf[ks] = *(const MMU_LDS bf16x8*)(vt + st * 32 * (D * 2) + fo.row[h4 + ks]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[h4 + ks], s, 0, 0, 0);
                        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[ks], dof[h4 + ks], dp, 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // dS^T without the 1/sqrt(D) factor: it multiplies the finished dQ accumulators once, after the sweep
                if (plain) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[r] = fast_exp2(__builtin_fmaf(s[r], c, -lse2)) * (dp[r] - dlt);
                } else {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int kl = st * 32 + 8 * g + 4 * hl;
                        const f32x4 kb4 = *(const MMU_LDS f32x4*)(kbs + kl);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float x = __builtin_fmaf(s[4 * g + j], c, kb4[j] - lse2);
                            if (diag && (kb * 64 + kl + j > qrow)) x = -INFINITY;
                            s[4 * g + j] = fast_exp2(x) * (dp[4 * g + j] - dlt);
                        }
                    }
                }
#pragma unroll
                for (int sk = 0; sk < 2; ++sk) {
                    const bf16x8 dsf = pack8(s, sk);
                    bf16x8 ktf[NDT];
#pragma unroll
                    for (int dt = 0; dt < NDT; ++dt)
                        ktf[dt] = tr_load(kt + (st * 32 + sk * 16) * (D * 2) + fo.tr1[dt],
                                          kt + (st * 32 + sk * 16) * (D * 2) + fo.tr2[dt]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int dt = 0; dt < NDT; ++dt)
                        acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf[dt], dsf, acc[dt], 0, 0, 0);
                }
            }
        }
#ifdef MMU_ATTN_STAMP
        {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long qb3 = stamp_now();
            qs_wait += qb1 - qb0; qs_stage += qb2 - qb1; qs_comp += qb3 - qb2;
        }
#endif
    }
    FSTAMP(q_loop_end);
    if constexpr (D == 128) {
        // The gradient tile leaves through LDS: whole 128-B lines row-major and, on request, the transposed copy for the q
        // weight gradient (store_tile32x128_via_lds).  The K / V tile buffers are free once every wave has left the key
        // loop.  a.dT is only passed for sequence lengths that are multiples of 8 (launch_bwd).
        __syncthreads();
        if (q0 < S) {
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[dt][r] *= a.scale;
            const bf16* cr = nullptr;
            const bf16* sr = nullptr;
            if (a.rope_pos) {
                const int pos = a.rope_pos[rowbase + (qvalid ? qrow : S - 1)];
                cr = a.rope_cos + (size_t)pos * 64;
                sr = a.rope_sin + (size_t)pos * 64;
            }
            u32x2 w[4][4];
            grad_words(acc, hl, cr, sr, w);
            store_tile32x128_via_lds<false>((MMU_LDS char*)smem + wave * 4096, w, lane,
                                            a.dq + (rowbase + q0) * a.ldg + hh * D, a.ldg,
                                            a.dT ? a.dT + (size_t)(hh * D) * a.ldt + rowbase + q0 : nullptr, a.ldt,
                                            min(32, S - q0));
        }
#ifdef MMU_ATTN_STAMP
        {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stores have left
            const unsigned long long q_end = stamp_now();
            if (threadIdx.x == 192 && blockIdx.x < 16384) {  // dQ kernel, last wave: second half of the forward's slots
                unsigned long long* w = g_fwd_stamps + 8 * (size_t)blockIdx.x;
                w[0] = nblk; w[1] = qs_wait; w[2] = qs_stage; w[3] = qs_comp; w[4] = q_loop - q_start;
                w[5] = q_end - q_loop_end; w[6] = q_end - q_start; w[7] = 1;
            }
        }
#endif
        return;
    }
    if (qvalid) {
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[dt][r] *= a.scale;
        bf16* op = a.dq + (rowbase + qrow) * a.ldg + hh * D;
        if constexpr (D == 128) {
            const bf16* cr = nullptr;
            const bf16* sr = nullptr;
            if (a.rope_pos) {
                const int pos = a.rope_pos[rowbase + qrow];
                cr = a.rope_cos + (size_t)pos * 64;
                sr = a.rope_sin + (size_t)pos * 64;
            }
            store_grad_row(op, acc, hl, cr, sr);
        } else {
            store_grad_row_plain<NDT>(op, acc, hl);  // head_dim 64 (CLIP tower): no rotary embedding
        }
    }
}

// ------------------------------------------------------------------------------------------------
// dK / dV. grid (ceil(S/128), H, B), 512 threads = 8 waves.  Wave w: key group kw = w&3 (keys kvb*128 + 32kw ..
// +31, "key on the lane") and query half qh = w>>2: every iteration the block stages TWO 32-row query tiles and
// waves 0-3 take the first, waves 4-7 the second, so each SIMD holds two waves that hide each other's LDS
// latency.  A wave keeps dK^T, dV^T partial accumulators (128 registers) for its keys; the two partials of a key
// group are summed through LDS once, after the sweep.  The block's K and V tiles (128 keys) stay in LDS for
// the whole sweep (B operands of S = Q K^T and dP = dO V^T); Q / dO tiles are double-buffered by LDS-DMA and
// read by rows (S, dP) and transposed (dK^T += Q^T dS, dV^T += dO^T P); lse / delta rows arrive by 4-byte
// LDS-DMA (no register round trip inside the loop).  Per-lane LDS offsets of every fragment are computed once.
// LDS: K 128xD | V 128xD | 2 x (Q 64xD | dO 64xD) | 2 x (64 lse | 64 delta)  = 129 KiB.
// ------------------------------------------------------------------------------------------------
template <int D, bool CAUSAL>
__global__ __launch_bounds__(512, 2) void attn_bwd_dkdv_kernel(AttnArgs a) {
    static_assert(D == 128 || D == 64, "backward is built for head_dim 128 (Llama) and 64 (CLIP tower)");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KVB = 128 * D * 2;  // K (or V) tile bytes
    constexpr int TB = 64 * D * 2;    // Q (or dO) pair-tile bytes (2 x 32 rows)
    constexpr int NKS = D / 16, NDT = D / 32;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kw = wave & 3, qh = wave >> 2;
    const int hl = lane >> 5;
    int kvb, hh, b;
    decode_wg((a.S + 127) / 128, a.H, a.B, false, kvb, hh, b);
    const int S = a.S;
    const size_t rowbase = (size_t)b * S;
    const bf16* qp = a.q + rowbase * a.ld + hh * D;
    const bf16* kp = a.k + rowbase * a.ld + hh * D;
    const bf16* vp = a.v + rowbase * a.ld + hh * D;
    const bf16* dop = a.dout + rowbase * a.lddo + hh * D;
    const float* lsep = a.delta + ((size_t)a.B * a.H + (size_t)b * a.H + hh) * S;  // lse * log2(e), left by the dQ kernel
    const float* dltp = a.delta + ((size_t)b * a.H + hh) * S;
    char* qd_base = smem + 2 * KVB;                                   // [2][Q | dO]
    float* rowc = reinterpret_cast<float*>(smem + 2 * KVB + 4 * TB);  // [2][lse 64 | delta 64]

    const int key0 = kvb * 128 + kw * 32;
    const int key = key0 + (lane & 31);
    const bool kvalid = key < S && (a.kmask == nullptr || a.kmask[rowbase + key] != 0);
    const bool all_valid = __all(kvalid);  // wave-uniform: tiles below the diagonal then need no select at all

    // Per-lane LDS byte offsets of this wave's Q-tile fragments inside buffer 0 (query half qh included); the dO tile
    // (+TB), the second k-step of a transposed read (+16 rows) and the K / V tiles are instruction immediates or one
    // wave-uniform delta away, and the double buffer is followed by an in-place XOR (see FragOffsets::toggle_rows).
    constexpr int QOFF = 2 * KVB;  // Q/dO double buffer: 64 KiB into the LDS, buffers 32 KiB apart
    int off_row[NKS];  // row fragment, k-step ks: row (lane&31), chunk 2ks+hl
    {
        const int r = lane & 31;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
            off_row[ks] = QOFF + qh * 32 * (D * 2) + r * (D * 2) + (((2 * ks + hl) ^ swz<D, KIND_ROW>(r)) << 4);
    }
    int off_tr1[NDT], off_tr2[NDT];  // transposed fragment for k-step 0 (k-step 1: + 16 rows = 16*D*2 bytes)
    {
        const int g = lane >> 4, gb = g & 1, h = g >> 1, i = lane & 15, q = i >> 2, pp = i & 3;
        const int r1 = 4 * h + q, r2 = r1 + 8;
        const int sub = (pp & 1) << 3;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
            const int ch = 4 * dt + 2 * gb + (pp >> 1);
            off_tr1[dt] = QOFF + qh * 32 * (D * 2) + r1 * (D * 2) + ((ch ^ swz<D, KIND_ROW>(r1)) << 4) + sub;
            off_tr2[dt] = QOFF + qh * 32 * (D * 2) + r2 * (D * 2) + ((ch ^ swz<D, KIND_ROW>(r2)) << 4) + sub;
        }
    }

    f32x16 dk[NDT], dv[NDT];
#pragma unroll
    for (int i = 0; i < NDT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[i][r] = 0.f; dv[i][r] = 0.f; }
    const float c = a.scale * LOG2E;

    // query-tile PAIRS: pair p covers rows 64p .. 64p+63
    const int pr0 = CAUSAL ? (kvb * 128) / 64 : 0;
    const int pr1 = (S + 63) / 64;

    auto stage = [&](int pr, int buf) {
        char* base = qd_base + buf * 2 * TB;
        stage_rows<D, KIND_ROW, 64, 8>(base, qp, a.ld, pr * 64, S, wave, lane);
        stage_rows<D, KIND_ROW, 64, 8>(base + TB, dop, a.lddo, pr * 64, S, wave, lane);
        if (wave < 2) {
            // wave 0: lse of the 64 rows, wave 1: delta, by 4-byte LDS-DMA (natural-log lse; scaled on use)
            const int qr = pr * 64 + lane;
            const float* src = wave == 0 ? (qr < S ? lsep + qr : reinterpret_cast<const float*>(mmu_inf16))
                                         : (qr < S ? dltp + qr : reinterpret_cast<const float*>(mmu_zero16));
            glds4(src, rowc + buf * 128 + wave * 64);
        }
    };

    STAMP(t_start);
#ifdef MMU_ATTN_STAMP
    unsigned long long t_wait = 0;
    int n_iter = 0;
#endif
    // block-resident K and V tiles (rows past S read the zero page)
    stage_rows<D, KIND_ROW, 128, 8>(smem, kp, a.ld, kvb * 128, S, wave, lane);
    stage_rows<D, KIND_ROW, 128, 8>(smem + KVB, vp, a.ld, kvb * 128, S, wave, lane);
    if (pr0 < pr1) stage(pr0, 0);

    const MMU_LDS char* lds0 = (const MMU_LDS char*)smem;

    for (int pr = pr0; pr < pr1; ++pr) {
        const int cur = (pr - pr0) & 1;
        STAMP(t_w0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#ifdef MMU_ATTN_STAMP
        if (pr == pr0) t_wait = 0;  // the first wait belongs to the prologue (reported separately)
        const unsigned long long t_w1 = stamp_now();
        if (pr > pr0) t_wait += t_w1 - t_w0;
        ++n_iter;
#endif
        if (pr == pr0) {
#ifdef MMU_ATTN_STAMP
            if (threadIdx.x == 0 && blockIdx.x < 16384) g_attn_stamps[4 * blockIdx.x + 0] = t_w1 - t_start;  // prologue
#endif
        }
        if (pr + 1 < pr1) stage(pr + 1, cur ^ 1);
        if (pr > pr0) {  // offsets follow the Q/dO buffer
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) off_row[ks] ^= 2 * TB;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) { off_tr1[dt] ^= 2 * TB; off_tr2[dt] ^= 2 * TB; }
        }
        const int q_first = pr * 64 + qh * 32;  // first query row of this wave's tile
        const bool active = q_first < S && (!CAUSAL || (q_first + 31 >= key0));
        if (active) {
            // K tile rows of this wave = the Q offsets shifted by a wave-uniform amount (one scalar per iteration)
            const int kdelta = kw * 32 * (D * 2) - (QOFF + qh * 32 * (D * 2) + cur * 2 * TB);
            const MMU_LDS float* rc = (const MMU_LDS float*)(smem + 2 * KVB + 4 * TB) + cur * 128 + qh * 32;
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
            // S[q][key] = Q K^T and dP[q][key] = dO V^T in batches of 4 k-steps: with the 128 dK^T / dV^T accumulators
            // resident, 8 + 8 fragments at once spilled (132 B of scratch, reloads queue on vmcnt behind the LDS-DMA)
#pragma unroll
            for (int h4 = 0; h4 < NKS; h4 += 4) {
                bf16x8 fa[4], fb[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    fa[ks] = *(const MMU_LDS bf16x8*)(lds0 + off_row[h4 + ks]);
                    fb[ks] = *(const MMU_LDS bf16x8*)(lds0 + (off_row[h4 + ks] + kdelta));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks], fb[ks], s, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int h4 = 0; h4 < NKS; h4 += 4) {
                bf16x8 fa[4], fb[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    fa[ks] = *(const MMU_LDS bf16x8*)(lds0 + off_row[h4 + ks] + TB);
                    fb[ks] = *(const MMU_LDS bf16x8*)(lds0 + (off_row[h4 + ks] + kdelta) + KVB);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks], fb[ks], dp, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // P and dS (without the 1/sqrt(D) factor: it multiplies the finished dK accumulators once, after the sweep)
            const bool diag = CAUSAL && (q_first < key0 + 31);
            if (all_valid && !diag) {  // wave-uniform: no select
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ql = 8 * g + 4 * hl;  // local q row of reg 4g
                    const f32x4 l4 = *(const MMU_LDS f32x4*)(rc + ql);
                    const f32x4 d4 = *(const MMU_LDS f32x4*)(rc + 64 + ql);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float pv = fast_exp2(__builtin_fmaf(s[4 * g + j], c, -l4[j]));
                        s[4 * g + j] = pv;
                        dp[4 * g + j] = pv * (dp[4 * g + j] - d4[j]);
                    }
                }
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ql = 8 * g + 4 * hl;
                    const f32x4 l4 = *(const MMU_LDS f32x4*)(rc + ql);
                    const f32x4 d4 = *(const MMU_LDS f32x4*)(rc + 64 + ql);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float x = __builtin_fmaf(s[4 * g + j], c, -l4[j]);
                        bool ok = kvalid;
                        if (diag && (key > q_first + ql + j)) ok = false;
                        const float pv = ok ? fast_exp2(x) : 0.f;
                        s[4 * g + j] = pv;
                        dp[4 * g + j] = pv * (dp[4 * g + j] - d4[j]);
                    }
                }
            }
            // dV^T[d][key] += dO^T[d][q] P[q][key] ; dK^T[d][key] += Q^T[d][q] dS[q][key]
#pragma unroll
            for (int sk = 0; sk < 2; ++sk) {
                const bf16x8 pf = pack8(s, sk);
                const bf16x8 dsf = pack8(dp, sk);
                bf16x8 fo[NDT], fq[NDT];
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
                    const bf16x4 o1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (MMU_LDS bf16x4*)(lds0 + off_tr1[dt] + TB + sk * 16 * D * 2));
                    const bf16x4 o2 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (MMU_LDS bf16x4*)(lds0 + off_tr2[dt] + TB + sk * 16 * D * 2));
                    fo[dt] = __builtin_shufflevector(o1, o2, 0, 1, 2, 3, 4, 5, 6, 7);
                    const bf16x4 q1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (MMU_LDS bf16x4*)(lds0 + off_tr1[dt] + sk * 16 * D * 2));
                    const bf16x4 q2 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (MMU_LDS bf16x4*)(lds0 + off_tr2[dt] + sk * 16 * D * 2));
                    fq[dt] = __builtin_shufflevector(q1, q2, 0, 1, 2, 3, 4, 5, 6, 7);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
                    dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fo[dt], pf, dv[dt], 0, 0, 0);
                    dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fq[dt], dsf, dk[dt], 0, 0, 0);
                }
            }
        }
    }
    STAMP(t_loop_end);
    // sum the two query-half partials of every key group through LDS: waves 4-7 publish, waves 0-3 add + store.
    // slab layout [kw][reg 0..2*NR-1][lane] fp32 -> lane-contiguous, conflict-free; 4 x (D/4) KiB over the dead tiles.
    constexpr int NR = NDT * 16;  // accumulator registers of dK^T (and of dV^T)
    __syncthreads();
    float* slab = reinterpret_cast<float*>(smem) + kw * (2 * NR * 64);
    if (qh == 1) {
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                slab[(dt * 16 + r) * 64 + lane] = dk[dt][r];
                slab[(NR + dt * 16 + r) * 64 + lane] = dv[dt][r];
            }
    }
    __syncthreads();
    if (qh == 0 && key < S) {
        bf16* dkp = a.dk + (rowbase + key) * a.ldg + hh * D;
        bf16* dvp = a.dv + (rowbase + key) * a.ldg + hh * D;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                dk[dt][r] = (dk[dt][r] + slab[(dt * 16 + r) * 64 + lane]) * a.scale;
                dv[dt][r] += slab[(NR + dt * 16 + r) * 64 + lane];
            }
        if constexpr (D == 128) {
            const bf16* cr = nullptr;
            const bf16* sr = nullptr;
            if (a.rope_pos) {
                const int pos = a.rope_pos[rowbase + key];
                cr = a.rope_cos + (size_t)pos * 64;
                sr = a.rope_sin + (size_t)pos * 64;
            }
            store_grad_row(dkp, dk, hl, cr, sr);
            store_grad_row(dvp, dv, hl, nullptr, nullptr);
        } else {
            store_grad_row_plain<NDT>(dkp, dk, hl);
            store_grad_row_plain<NDT>(dvp, dv, hl);
        }
    }
#ifdef MMU_ATTN_STAMP
    {
        const unsigned long long t_end = stamp_now();
        if (threadIdx.x == 0 && blockIdx.x < 16384) {
            g_attn_stamps[4 * blockIdx.x + 1] = t_loop_end - t_start;                      // start -> end of the sweep
            g_attn_stamps[4 * blockIdx.x + 2] = t_wait;                                    // loop-top vmcnt + barrier waits
            g_attn_stamps[4 * blockIdx.x + 3] = ((t_end - t_start) << 8) | (unsigned)n_iter;  // total cycles, iterations
        }
    }
#endif
}


template <int D, bool CAUSAL>
int launch_fwd(const AttnArgs& a, hipStream_t s) {
    constexpr int LDS = 4 * 64 * D * 2 + 2 * 64 * 4 + 16;
    static bool cfg = false;
    if (!cfg) {
        int e = set_lds(attn_fwd_kernel<D, CAUSAL>, LDS);
        if (e) return e;
        cfg = true;
    }
    hipLaunchKernelGGL((attn_fwd_kernel<D, CAUSAL>), dim3(((a.S + 127) / 128) * a.H * a.B), dim3(256), LDS, s, a);
    MMU_CHECK_LAUNCH();
    return 0;
}

#ifndef MMU_DKDV_DEFAULT
#define MMU_DKDV_DEFAULT 4
#endif
static int g_dkdv_variant = MMU_DKDV_DEFAULT;  // mmu_attn_set_dkdv_variant: 8 = eight-wave kernel, 4 = four-wave persistent kernel
// measurement hook (bench.py): an event the NEXT backward call records between its dQ and its dK/dV kernel, on the
// stream they are launched on; one-shot.  Lets the two kernels of one C call be timed separately, in the step.
static hipEvent_t g_bwd_mid_event = nullptr;

template <int D, bool CAUSAL>
int launch_bwd(const AttnArgs& a, hipStream_t s) {
    constexpr int LDS_DQ = 4 * 64 * D * 2 + 2 * 64 * 4 + 16;
    constexpr int LDS_KV = 2 * 128 * D * 2 + 4 * 64 * D * 2 + 2 * 128 * 4;
    static bool cfg = false;
    if (!cfg) {
        int e = set_lds(attn_bwd_dq_kernel<D, CAUSAL>, LDS_DQ);
        if (e) return e;
        e = set_lds(attn_bwd_dkdv_kernel<D, CAUSAL>, LDS_KV);
        if (e) return e;
        cfg = true;
    }
    // delta / lse2 rows: by the dQ kernel itself (launched first, so the dK/dV kernel finds them), or by a pass of their own
#ifdef MMU_ATTN_DELTA_PASS  // A/B builds only
    constexpr bool fuse = false;
#else
    constexpr bool fuse = true;
#endif
    if (!fuse) {
        const size_t nthr = (size_t)a.B * a.S * a.H * (D / 8);
        hipLaunchKernelGGL((attn_delta_kernel<D>), dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, s, a.o, a.ldo,
                           a.dout, a.lddo, const_cast<float*>(a.delta), a.lse, a.B, a.S, a.H);
        MMU_CHECK_LAUNCH();
    }
    // Transposed gradient copy (a.dT): the head_dim-128 dQ kernel and the four-wave dK/dV kernel write their parts
    // themselves when the sequence length is a multiple of 8 (the copy moves 16-B groups of 8 tokens); every other part
    // goes through a strided transpose of the finished row-major block.
    const int HD = a.H * D, M = a.B * a.S;
    bf16* const dT = a.dT;
    AttnArgs k = a;  // what the kernels see
    k.fuse_delta = fuse ? 1 : 0;
    const bool fused_t = dT && D == 128 && a.S % 8 == 0;
    if (!fused_t) k.dT = nullptr;
    auto transpose_part = [&](int part, const bf16* src) -> int {
        return mmu_internal_transpose_strided(src, a.ldg, dT + (size_t)part * HD * a.ldt, a.ldt, M, HD, s);
    };
    hipLaunchKernelGGL((attn_bwd_dq_kernel<D, CAUSAL>), dim3(((a.S + 127) / 128) * a.H * a.B), dim3(256), LDS_DQ, s, k);
    MMU_CHECK_LAUNCH();
    if (g_bwd_mid_event) {
        const hipError_t ee = hipEventRecord(g_bwd_mid_event, s);
        g_bwd_mid_event = nullptr;
        if (ee != hipSuccess) return (int)ee;
    }
    bool four_wave = false;
    if constexpr (D == 128) {
        // 4 waves x 512 registers, persistent, K / V in registers, 4-stage Q/dO ring (attention_dkdv4.hip); row strides
        // must fit its 24-bit offset arithmetic, else the eight-wave kernel runs.
        if (g_dkdv_variant == 4 && a.ld < (1 << 23) && a.lddo < (1 << 23)) {
            const int e4 = mmu_internal_launch_dkdv4(&k, CAUSAL ? 1 : 0, s);
            if (e4) return e4;
            four_wave = true;
        }
    }
    if (!four_wave) {
        hipLaunchKernelGGL((attn_bwd_dkdv_kernel<D, CAUSAL>), dim3(((a.S + 127) / 128) * a.H * a.B), dim3(512), LDS_KV, s, k);
        MMU_CHECK_LAUNCH();
    }
    if (dT) {
        int e = fused_t ? 0 : transpose_part(0, a.dq);
        if (!e && !(fused_t && four_wave)) e = transpose_part(1, a.dk);
        if (!e && !(fused_t && four_wave)) e = transpose_part(2, a.dv);
        if (e) return e;
    }
    return 0;
}

bool attn_args_ok(const AttnArgs& a, int D) {
    if (a.B <= 0 || a.S <= 0 || a.H <= 0) return false;
    if ((a.ld & 7) || (a.ldo & 3)) return false;
    if (((uintptr_t)a.q | (uintptr_t)a.k | (uintptr_t)a.v) & 15) return false;
    if ((uintptr_t)a.o & 7) return false;
    return D == 64 || D == 128;
}

}  // namespace

// q/k/v: bf16 rows of `ld` elements, head h at column h*head_dim of each pointer; sequences are consecutive
// blocks of S rows. o: [B*S][ldo]. lse: [B][H][S] fp32 (may be null for inference). kmask: [B][S] int32 or null.
extern "C" int mmu_attn_fwd(const void* q, const void* k, const void* v, int ld, void* o, int ldo, float* lse,
                            const int* kmask, int B, int S, int H, int head_dim, float scale, int causal,
                            void* stream) {
    AttnArgs a = {};
    a.q = (const bf16*)q; a.k = (const bf16*)k; a.v = (const bf16*)v; a.ld = ld;
    a.o = (bf16*)o; a.ldo = ldo; a.lse = lse; a.kmask = kmask;
    a.B = B; a.S = S; a.H = H; a.scale = scale;
    if (!attn_args_ok(a, head_dim)) return -1;
    hipStream_t s = (hipStream_t)stream;
    if (head_dim == 128) return causal ? launch_fwd<128, true>(a, s) : launch_fwd<128, false>(a, s);
    return causal ? launch_fwd<64, true>(a, s) : launch_fwd<64, false>(a, s);
}

// mmu_attn_fwd that also leaves O^T = [H * head_dim][ldt] (ldt >= B * S, a multiple of 8): the transposed activation the
// o-projection's weight gradient dW_o = dY^T O wants as its reduction-contiguous operand, written from the staging image the
// row-major store goes through anyway (head_dim 128; B * S and S multiples of 8, else -1: the caller transposes O itself).
extern "C" int mmu_attn_fwd_t(const void* q, const void* k, const void* v, int ld, void* o, int ldo, void* o_t, int ldt,
                              float* lse, const int* kmask, int B, int S, int H, int head_dim, float scale, int causal,
                              void* stream) {
    AttnArgs a = {};
    a.q = (const bf16*)q; a.k = (const bf16*)k; a.v = (const bf16*)v; a.ld = ld;
    a.o = (bf16*)o; a.ldo = ldo; a.lse = lse; a.kmask = kmask;
    a.B = B; a.S = S; a.H = H; a.scale = scale;
    a.dT = (bf16*)o_t; a.ldt = ldt;
    if (!attn_args_ok(a, head_dim) || head_dim != 128 || !o_t) return -1;
    if ((S & 7) || (ldt & 7) || (long long)ldt < (long long)B * S || ((uintptr_t)o_t & 15) || ((uintptr_t)o & 15) || (ldo & 7))
        return -1;
    hipStream_t s = (hipStream_t)stream;
    return causal ? launch_fwd<128, true>(a, s) : launch_fwd<128, false>(a, s);
}

// Backward (head_dim 128 or 64): needs q,k,v,o,lse of the forward plus dout; writes dq,dk,dv (row stride ldg,
// head h at column h*head_dim of each pointer) and uses delta [B][H][S] fp32 as workspace.
extern "C" int mmu_attn_bwd_rope(const void* q, const void* k, const void* v, int ld, const void* o, int ldo,
                                 const float* lse, const int* kmask, const void* dout, int lddo, float* delta,
                                 void* dq, void* dk, void* dv, int ldg, int B, int S, int H, int head_dim, float scale,
                                 int causal, const int* rope_pos, const void* rope_cos, const void* rope_sin,
                                 void* stream);

extern "C" int mmu_attn_bwd(const void* q, const void* k, const void* v, int ld, const void* o, int ldo,
                            const float* lse, const int* kmask, const void* dout, int lddo, float* delta, void* dq,
                            void* dk, void* dv, int ldg, int B, int S, int H, int head_dim, float scale, int causal,
                            void* stream) {
    return mmu_attn_bwd_rope(q, k, v, ld, o, ldo, lse, kmask, dout, lddo, delta, dq, dk, dv, ldg, B, S, H, head_dim,
                             scale, causal, nullptr, nullptr, nullptr, stream);
}

// mmu_attn_bwd with the inverse RoPE rotation (gradient w.r.t. the un-rotated q, k) fused into the dq / dk stores;
// rope_pos [B*S] int32, rope_cos / rope_sin [n_pos][64] bf16 (all three NULL: plain mmu_attn_bwd).
// dqkv_t (may be NULL): additionally leaves [dq | dk | dv]^T = [3 * H * head_dim][ldt] (ldt >= B * S, a multiple of 8)
// for the q|k|v weight gradient, i.e. what transposing the packed [B*S][3 * H * head_dim] gradient would give.
static int attn_bwd_impl(const void* q, const void* k, const void* v, int ld, const void* o, int ldo, const float* lse,
                         const int* kmask, const void* dout, int lddo, float* delta, void* dq, void* dk, void* dv, int ldg,
                         int B, int S, int H, int head_dim, float scale, int causal, const int* rope_pos,
                         const void* rope_cos, const void* rope_sin, void* dqkv_t, int ldt, void* stream) {
    AttnArgs a = {};
    a.rope_pos = rope_pos; a.rope_cos = (const bf16*)rope_cos; a.rope_sin = (const bf16*)rope_sin;
    if (rope_pos && (!rope_cos || !rope_sin)) return -1;
    a.q = (const bf16*)q; a.k = (const bf16*)k; a.v = (const bf16*)v; a.ld = ld;
    a.o = (bf16*)const_cast<void*>(o); a.ldo = ldo; a.lse = const_cast<float*>(lse); a.kmask = kmask;
    a.B = B; a.S = S; a.H = H; a.scale = scale;
    a.dout = (const bf16*)dout; a.lddo = lddo; a.delta = delta;
    a.dq = (bf16*)dq; a.dk = (bf16*)dk; a.dv = (bf16*)dv; a.ldg = ldg;
    a.dT = (bf16*)dqkv_t; a.ldt = ldt;
    if (!attn_args_ok(a, head_dim) || (lddo & 7) || (ldg & 3) || !lse || !delta) return -1;
    if (head_dim != 128 && rope_pos) return -1;  // the fused inverse RoPE is a head_dim-128 (Llama) epilogue
    if (((uintptr_t)dout & 15) || (((uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 7)) return -1;
    if (dqkv_t) {  // the transposed copy moves 16-B pieces of 8 tokens
        if ((long long)ldt < (long long)B * S || (ldt & 7) || ((B * S) & 7) || (ldg & 7) || ((uintptr_t)dqkv_t & 15) ||
            (((uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 15))
            return -1;
    }
    hipStream_t s = (hipStream_t)stream;
    if (head_dim == 64) return causal ? launch_bwd<64, true>(a, s) : launch_bwd<64, false>(a, s);
    return causal ? launch_bwd<128, true>(a, s) : launch_bwd<128, false>(a, s);
}
extern "C" int mmu_attn_bwd_rope(const void* q, const void* k, const void* v, int ld, const void* o, int ldo,
                                 const float* lse, const int* kmask, const void* dout, int lddo, float* delta,
                                 void* dq, void* dk, void* dv, int ldg, int B, int S, int H, int head_dim, float scale,
                                 int causal, const int* rope_pos, const void* rope_cos, const void* rope_sin,
                                 void* stream) {
    return attn_bwd_impl(q, k, v, ld, o, ldo, lse, kmask, dout, lddo, delta, dq, dk, dv, ldg, B, S, H, head_dim, scale,
                         causal, rope_pos, rope_cos, rope_sin, nullptr, 0, stream);
}
extern "C" int mmu_attn_bwd_rope_t(const void* q, const void* k, const void* v, int ld, const void* o, int ldo,
                                   const float* lse, const int* kmask, const void* dout, int lddo, float* delta,
                                   void* dq, void* dk, void* dv, int ldg, int B, int S, int H, int head_dim, float scale,
                                   int causal, const int* rope_pos, const void* rope_cos, const void* rope_sin,
                                   void* dqkv_t, int ldt, void* stream) {
    if (!dqkv_t) return -1;
    return attn_bwd_impl(q, k, v, ld, o, ldo, lse, kmask, dout, lddo, delta, dq, dk, dv, ldg, B, S, H, head_dim, scale,
                         causal, rope_pos, rope_cos, rope_sin, dqkv_t, ldt, stream);
}

#ifdef MMU_ATTN_STAMP
extern "C" int mmu_debug_read_fwd_stamps(unsigned long long* host_out, int n_blocks) {
    if (n_blocks < 1 || n_blocks > 16384) return -1;
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_fwd_stamps), (size_t)n_blocks * 64, 0, hipMemcpyDeviceToHost);
}
extern "C" int mmu_debug_read_attn_stamps(unsigned long long* host_out, int n_blocks) {
    if (n_blocks < 1 || n_blocks > 16384) return -1;
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_attn_stamps), (size_t)n_blocks * 32, 0, hipMemcpyDeviceToHost);
}
#endif

// dK/dV kernel of the head_dim-128 backward: 8 = eight waves, K / V tiles in LDS, two partial sums per key group;
// 4 (default: -0.6 % step time in a same-box A/B, profiles/r02_dkdv4_instep_ab.txt) = four waves, persistent, K / V
// fragments in registers and a 4-stage Q/dO ring.  Process-wide A/B / tuning switch.
// Bits 8 and up: workgroups of the persistent four-wave kernel (0 = one per CU, else a multiple of 8; the tests use a
// small grid so that every workgroup crosses item seams).
// One-shot: the next mmu_attn_bwd* call records `event` (a hipEvent_t created with timing enabled) on its stream between
// the dQ and the dK/dV kernel.  nullptr cancels.  Measurement only (bench.py times the two kernels separately in the step).
extern "C" int mmu_attn_set_bwd_mid_event(void* event) {
    g_bwd_mid_event = (hipEvent_t)event;
    return 0;
}

extern "C" int mmu_attn_set_dkdv_variant(int v) {
    const int kind = v & 255, grid = v >> 8;
    if (v < 0 || (kind != 4 && kind != 8)) return -1;
    if (mmu_internal_set_dkdv4_grid(grid)) return -1;
    g_dkdv_variant = kind;
    return 0;
}
