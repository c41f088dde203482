// Loss kernels over [N, V] bf16 logits for gfx950: HBM-streaming, one 256-thread block per row, online
// softmax statistics via wave shuffles, fp32 math on bf16 logits, softmax never materialised.
//
// Replaces mm/trainer_utils.py:16-21 logits2probs + F.kl_div call sites (mm/trainer.py:203-208,220-225,
// 237-242,250-255,271-276,294-299,435-440), the NPO ratio (mm/trainer.py:348-353), the entropy term
// (:195-196), get_batch_loss (mm/trainer_utils.py:73-81) and the shifted CE the HF model computes for
// `outputs.loss` (transformers loss_utils.py ForCausalLMLoss).  SURVEY.md §8a rows R3-R7, K14.
#include "common.h"

namespace {

struct OnlineLse {
    float m, s;
    __device__ __forceinline__ void init() { m = -INFINITY; s = 0.f; }
    __device__ __forceinline__ void add(float x) {
        if (x > m) {
            s = s * __expf(m - x) + 1.f;
            m = x;
        } else {
            s += __expf(x - m);
        }
    }
    __device__ __forceinline__ void merge(float m2, float s2) {
        const float mn = fmaxf(m, m2);
        if (mn == -INFINITY) return;
        s = s * __expf(m - mn) + s2 * __expf(m2 - mn);
        m = mn;
    }
};

__device__ __forceinline__ void wave_merge(OnlineLse& o) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float m2 = __shfl_xor(o.m, off, 64), s2 = __shfl_xor(o.s, off, 64);
        o.merge(m2, s2);
    }
}

// block-wide logsumexp for a row of V bf16 logits (V % 8 == 0). scratch: 8 floats.
__device__ __forceinline__ float block_lse(const bf16* __restrict__ row, int V, float* scratch) {
    OnlineLse o;
    o.init();
    const int nch = V >> 3;
    for (int ch = threadIdx.x; ch < nch; ch += 256) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(row + ch * 8);
        float mx = bf2f(v[0]);
#pragma unroll
        for (int j = 1; j < 8; ++j) mx = fmaxf(mx, bf2f(v[j]));
        float sm = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) sm += __expf(bf2f(v[j]) - mx);
        o.merge(mx, sm);
    }
    wave_merge(o);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) {
        scratch[w] = o.m;
        scratch[4 + w] = o.s;
    }
    __syncthreads();
    OnlineLse t;
    t.m = scratch[0];
    t.s = scratch[4];
#pragma unroll
    for (int i = 1; i < 4; ++i) t.merge(scratch[i], scratch[4 + i]);
    return t.m + __logf(t.s);
}

__device__ __forceinline__ long long shifted_target(const long long* __restrict__ labels, int row, int S, int shift) {
    if (!shift) return labels[row];
    const int s = row % S;
    return (s + 1 < S) ? labels[row + 1] : -100;
}

// Cross entropy forward: lse[row], loss_row[row] = lse - logit[target] (0 when target == ignore_index)
__global__ __launch_bounds__(256) void ce_fwd_kernel(const bf16* __restrict__ logits, int ldv,
                                                     const long long* __restrict__ labels, float* __restrict__ lse,
                                                     float* __restrict__ loss_row, int M, int V, int S, int shift) {
    __shared__ float scratch[8];
    const int row = blockIdx.x;
    const bf16* lr = logits + (size_t)row * ldv;
    const float l = block_lse(lr, V, scratch);
    if (threadIdx.x == 0) {
        lse[row] = l;
        const long long t = shifted_target(labels, row, S, shift);
        loss_row[row] = (t >= 0 && t < V) ? (l - bf2f(lr[t])) : 0.f;
    }
}

// Cross entropy backward: dlogits[row][v] = coef * (exp(logit - lse) - [v == target]);
// coef = seg_scale[seg] / (normalize ? seg_cnt[seg] : 1) for a valid target, else 0. seg = row / seg_rows.
// In-place (dlogits == logits) is allowed: every element is read before it is written by the same lane.
__global__ __launch_bounds__(256) void ce_bwd_kernel(const bf16* __restrict__ logits, int ldv,
                                                     const long long* __restrict__ labels,
                                                     const float* __restrict__ lse,
                                                     const float* __restrict__ seg_scale,
                                                     const float* __restrict__ seg_cnt, int seg_rows, int normalize,
                                                     bf16* __restrict__ dlogits, int ldd, int M, int V, int S,
                                                     int shift) {
    const int row = blockIdx.x;
    const long long t = shifted_target(labels, row, S, shift);
    const int seg = row / seg_rows;
    const bool valid = (t >= 0 && t < V);
    float coef = 0.f;
    if (valid) {
        coef = seg_scale[seg];
        if (normalize) coef /= fmaxf(seg_cnt[seg], 1.f);
    }
    const bf16* lr = logits + (size_t)row * ldv;
    bf16* dr = dlogits + (size_t)row * ldd;
    const float l = lse[row];
    const int nch = V >> 3;
    for (int ch = threadIdx.x; ch < nch; ch += 256) {
        bf16x8 o;
        if (coef == 0.f) {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = f2bf(0.f);
        } else {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(lr + ch * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float p = __expf(bf2f(v[j]) - l);
                if ((long long)(ch * 8 + j) == t) p -= 1.f;
                o[j] = f2bf(coef * p);
            }
        }
        *reinterpret_cast<bf16x8*>(dr + ch * 8) = o;
    }
}

// Segment reduce: out_sum[seg] = sum of vals over the segment's rows, out_cnt[seg] = number of rows with
// a valid (shifted) target (or all rows when labels == null). One block per segment; deterministic.
__global__ __launch_bounds__(256) void seg_reduce_kernel(const float* __restrict__ vals,
                                                         const long long* __restrict__ labels, int seg_rows, int M,
                                                         int V, int S, int shift, float* __restrict__ out_sum,
                                                         float* __restrict__ out_cnt) {
    __shared__ float scratch[4];
    const int seg = blockIdx.x;
    const int r0 = seg * seg_rows, r1 = min(M, r0 + seg_rows);
    float s = 0.f, c = 0.f;
    for (int r = r0 + threadIdx.x; r < r1; r += 256) {
        s += vals[r];
        if (labels) {
            const long long t = shifted_target(labels, r, S, shift);
            c += (t >= 0 && t < V) ? 1.f : 0.f;
        } else {
            c += 1.f;
        }
    }
    s = block_sum<4>(s, scratch);
    c = block_sum<4>(c, scratch);
    if (threadIdx.x == 0) {
        out_sum[seg] = s;
        if (out_cnt) out_cnt[seg] = c;
    }
}

// ------------------------------------------------------------------------------------------------
// Dual-logit row losses. stats[row] = {lse_s, lse_t, row_value, aux}
//   mode 0  KL(input=student, target=teacher): sum_v p_t (lp_t - lp_s)     (grad_diff_*_kl, kl, sku)
//   mode 1  KL(input=teacher, target=student): sum_v p_s (lp_s - lp_t)     (scrub, llmu)
//   mode 2  NPO: sum_v log(1 + (p_s / (p_t + 1e-8))^beta); aux = sum_v beta r^beta / (1 + r^beta)
//   mode 3  entropy term: sum_v p_s lp_s   (teacher unused)
//   mode 4  nlp NPO form (nlp/dataloader.py:396-401): sum_v log(1 + (lp_s - lp_t)^beta); aux = sum_v dterm/dr
//           (torch.pow semantics: NaN for a negative base with a non-integer exponent, as in the reference)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rowloss_fwd_kernel(int mode, const bf16* __restrict__ slog,
                                                          const bf16* __restrict__ tlog, int ldv, float beta,
                                                          float* __restrict__ stats, int M, int V) {
    __shared__ float scratch[8];
    const int row = blockIdx.x;
    const bf16* sr = slog + (size_t)row * ldv;
    const bf16* tr = tlog ? tlog + (size_t)row * ldv : nullptr;
    const float ls = block_lse(sr, V, scratch);
    const float lt = (mode == 3 || !tr) ? 0.f : block_lse(tr, V, scratch);
    float acc = 0.f, aux = 0.f;
    const int nch = V >> 3;
    for (int ch = threadIdx.x; ch < nch; ch += 256) {
        const bf16x8 sv = *reinterpret_cast<const bf16x8*>(sr + ch * 8);
        bf16x8 tv = sv;
        if (mode != 3) tv = *reinterpret_cast<const bf16x8*>(tr + ch * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float lps = bf2f(sv[j]) - ls;
            const float lpt = bf2f(tv[j]) - lt;
            if (mode == 0) {
                acc += __expf(lpt) * (lpt - lps);
            } else if (mode == 1) {
                acc += __expf(lps) * (lps - lpt);
            } else if (mode == 2) {
                const float rb = __expf(beta * (lps - __logf(__expf(lpt) + 1e-8f)));  // r^beta
                acc += log1pf(rb);
                aux += beta * rb / (1.f + rb);
            } else if (mode == 4) {
                const float r = lps - lpt;
                const float rb = powf(r, beta);
                acc += logf(1.f + rb);
                aux += beta * powf(r, beta - 1.f) / (1.f + rb);
            } else {
                acc += __expf(lps) * lps;
            }
        }
    }
    acc = block_sum<4>(acc, scratch);
    aux = block_sum<4>(aux, scratch);
    if (threadIdx.x == 0) {
        float* st = stats + (size_t)row * 4;
        st[0] = ls;
        st[1] = lt;
        st[2] = acc;
        st[3] = aux;
    }
}

// dlogits_s[row][v] = coef * d(row_value)/d(s_v), coef = *coef_ptr (device scalar, e.g. grad_out * weight / N)
__global__ __launch_bounds__(256) void rowloss_bwd_kernel(int mode, const bf16* __restrict__ slog,
                                                          const bf16* __restrict__ tlog, int ldv, float beta,
                                                          const float* __restrict__ stats,
                                                          const float* __restrict__ coef_ptr,
                                                          bf16* __restrict__ dlogits, int ldd, int M, int V) {
    const int row = blockIdx.x;
    const bf16* sr = slog + (size_t)row * ldv;
    const bf16* tr = tlog ? tlog + (size_t)row * ldv : nullptr;
    bf16* dr = dlogits + (size_t)row * ldd;
    const float* st = stats + (size_t)row * 4;
    const float ls = st[0], lt = st[1], val = st[2], aux = st[3];
    const float coef = *coef_ptr;
    const int nch = V >> 3;
    for (int ch = threadIdx.x; ch < nch; ch += 256) {
        const bf16x8 sv = *reinterpret_cast<const bf16x8*>(sr + ch * 8);
        bf16x8 tv = sv;
        if (mode != 3) tv = *reinterpret_cast<const bf16x8*>(tr + ch * 8);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float lps = bf2f(sv[j]) - ls;
            const float lpt = bf2f(tv[j]) - lt;
            const float ps = __expf(lps);
            float g;
            if (mode == 0) {
                g = ps - __expf(lpt);
            } else if (mode == 1) {
                g = ps * ((lps - lpt) - val);
            } else if (mode == 2) {
                const float rb = __expf(beta * (lps - __logf(__expf(lpt) + 1e-8f)));
                g = beta * rb / (1.f + rb) - ps * aux;
            } else if (mode == 4) {
                const float r = lps - lpt;
                g = beta * powf(r, beta - 1.f) / (1.f + powf(r, beta)) - ps * aux;
            } else {
                g = ps * (lps - val);
            }
            o[j] = f2bf(coef * g);
        }
        *reinterpret_cast<bf16x8*>(dr + ch * 8) = o;
    }
}

}  // namespace

extern "C" int mmu_ce_fwd(const void* logits, int ldv, const long long* labels, float* lse, float* loss_row, int M,
                          int V, int S, int shift, void* stream) {
    if (M <= 0 || (V & 7) || (ldv & 7) || S <= 0) return -1;
    hipLaunchKernelGGL(ce_fwd_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, (const bf16*)logits, ldv, labels,
                       lse, loss_row, M, V, S, shift);
    MMU_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmu_ce_bwd(const void* logits, int ldv, const long long* labels, const float* lse,
                          const float* seg_scale, const float* seg_cnt, int seg_rows, int normalize, void* dlogits,
                          int ldd, int M, int V, int S, int shift, void* stream) {
    if (M <= 0 || (V & 7) || (ldv & 7) || (ldd & 7) || S <= 0 || seg_rows <= 0) return -1;
    hipLaunchKernelGGL(ce_bwd_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, (const bf16*)logits, ldv, labels,
                       lse, seg_scale, seg_cnt, seg_rows, normalize, (bf16*)dlogits, ldd, M, V, S, shift);
    MMU_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmu_seg_reduce(const float* vals, const long long* labels, int seg_rows, int M, int V, int S, int shift,
                              float* out_sum, float* out_cnt, void* stream) {
    if (M <= 0 || seg_rows <= 0 || S <= 0) return -1;
    const int nseg = (M + seg_rows - 1) / seg_rows;
    hipLaunchKernelGGL(seg_reduce_kernel, dim3(nseg), dim3(256), 0, (hipStream_t)stream, vals, labels, seg_rows, M, V,
                       S, shift, out_sum, out_cnt);
    MMU_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmu_rowloss_fwd(int mode, const void* slog, const void* tlog, int ldv, float beta, float* stats, int M,
                               int V, void* stream) {
    if (M <= 0 || (V & 7) || (ldv & 7) || mode < 0 || mode > 4 || (mode != 3 && !tlog)) return -1;
    hipLaunchKernelGGL(rowloss_fwd_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, mode, (const bf16*)slog,
                       (const bf16*)tlog, ldv, beta, stats, M, V);
    MMU_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmu_rowloss_bwd(int mode, const void* slog, const void* tlog, int ldv, float beta, const float* stats,
                               const float* coef, void* dlogits, int ldd, int M, int V, void* stream) {
    if (M <= 0 || (V & 7) || (ldv & 7) || (ldd & 7) || mode < 0 || mode > 4 || (mode != 3 && !tlog)) return -1;
    hipLaunchKernelGGL(rowloss_bwd_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, mode, (const bf16*)slog,
                       (const bf16*)tlog, ldv, beta, stats, coef, (bf16*)dlogits, ldd, M, V);
    MMU_CHECK_LAUNCH();
    return 0;
}
