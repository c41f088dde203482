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

// Online log-sum-exp that also carries w = sum_i e^(x_i - m) y_i, so that sum_i softmax(x)_i y_i = w / s needs no
// second walk (and no second exp) over the row once the log-sum-exp is known.
struct OnlineLseW {
    float m, s, w;
    __device__ __forceinline__ void init() { m = -INFINITY; s = 0.f; w = 0.f; }
    __device__ __forceinline__ void merge(float m2, float s2, float w2) {
        const float mn = fmaxf(m, m2);
        if (mn == -INFINITY) return;
        const float a = __expf(m - mn), b = __expf(m2 - mn);
        s = s * a + s2 * b;
        w = w * a + w2 * b;
        m = mn;
    }
};
__device__ __forceinline__ void wave_merge(OnlineLseW& o) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float m2 = __shfl_xor(o.m, off, 64), s2 = __shfl_xor(o.s, off, 64), w2 = __shfl_xor(o.w, off, 64);
        o.merge(m2, s2, w2);
    }
}
// x: the logits the softmax is taken over, y: the per-element factor (8 of each)
__device__ __forceinline__ void lse_chunk_w(OnlineLseW& o, const bf16x8& x, const float (&y)[8]) {
    float mx = bf2f(x[0]);
#pragma unroll
    for (int j = 1; j < 8; ++j) mx = fmaxf(mx, bf2f(x[j]));
    float sm = 0.f, wm = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float e = __expf(bf2f(x[j]) - mx);
        sm += e;
        wm += e * y[j];
    }
    o.merge(mx, sm, wm);
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
// Per-element term of the row value (shared by the streaming and the register-resident kernels: same arithmetic).
__device__ __forceinline__ void rowloss_term(int mode, float s, float t, float ls, float lt, float beta, float& acc,
                                             float& aux) {
    const float lps = s - ls;
    const float lpt = t - lt;
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

__device__ __forceinline__ void lse_chunk(OnlineLse& o, const bf16x8& v) {
    float mx = bf2f(v[0]);
#pragma unroll
    for (int j = 1; j < 8; ++j) mx = fmaxf(mx, bf2f(v[j]));
    float sm = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) sm += __expf(bf2f(v[j]) - mx);
    o.merge(mx, sm);
}

__device__ __forceinline__ float block_lse_finish(OnlineLse o, float* scratch) {
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

// Streaming form (any V): the log-sum-exp pass and the value pass each walk the rows, i.e. student and teacher rows
// are read twice (L2 catches part of the second walk).
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
        for (int j = 0; j < 8; ++j) rowloss_term(mode, bf2f(sv[j]), bf2f(tv[j]), ls, lt, beta, acc, aux);
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

// Register-resident form, V <= 512 * 8 * 8 = 32768 (LLaVA-1.5: 32064): 512 threads per row, each loads its <= 8 chunks
// of the student row and of the teacher row ONCE (2 x 32 VGPRs, 16 loads of 16 B in flight per lane), then takes the
// two log-sum-exps and the row value from registers: every logit crosses HBM exactly once = the algorithmic
// 2 * M * V * 2 bytes.  <= 128 VGPRs -> two rows per CU, one loading while the other computes.  Same per-element
// arithmetic as the streaming kernel; the summation order over threads differs (512 vs 256 threads per row).
template <int MODE>
__global__ __launch_bounds__(512, 4) void rowloss_fwd_regs_kernel(const bf16* __restrict__ slog,
                                                                  const bf16* __restrict__ tlog, int ldv, float beta,
                                                                  float* __restrict__ stats, int M, int V) {
    constexpr int NC = 8;
    constexpr bool HAS_T = MODE != 3;
    constexpr int mode = MODE;  // compile-time: the per-element body stays small enough to unroll 8 x 8 times
    __shared__ float scratch[16];
    __shared__ float wscratch[40];
    const int row = blockIdx.x;
    const bf16* sr = slog + (size_t)row * ldv;
    const bf16* tr = HAS_T ? tlog + (size_t)row * ldv : nullptr;
    const int nch = V >> 3;
    bf16x8 sv[NC], tv[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int ch = threadIdx.x + i * 512;
        if (ch < nch) {
            sv[i] = *reinterpret_cast<const bf16x8*>(sr + ch * 8);
            if (HAS_T) tv[i] = *reinterpret_cast<const bf16x8*>(tr + ch * 8);
        }
    }
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    auto finish = [&](OnlineLse o) -> float {
        wave_merge(o);
        __syncthreads();
        if (l == 0) {
            scratch[w] = o.m;
            scratch[8 + w] = o.s;
        }
        __syncthreads();
        OnlineLse t;
        t.m = scratch[0];
        t.s = scratch[8];
#pragma unroll
        for (int i = 1; i < 8; ++i) t.merge(scratch[i], scratch[8 + i]);
        return t.m + __logf(t.s);
    };
    if constexpr (MODE == 0 || MODE == 1 || MODE == 3) {
        // The three expectation-type rows in ONE walk over the registers and one exp per logit: with x the logits of the
        // distribution the expectation is taken under (teacher for mode 0, student for modes 1 and 3) and
        // y = x - other (modes 0, 1) or y = x (mode 3),  row value = E_softmax(x)[y] - lse(x) + lse(other)  (mode 3:
        // E[x] - lse(x)); E[y] = w / s from the weighted online log-sum-exp.  (The streaming kernel keeps the three-walk
        // form p * (lp - lp'); the two agree to fp32 rounding of the row sums.)
        OnlineLseW ow;
        OnlineLse oo;
        ow.init();
        oo.init();
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            if (threadIdx.x + i * 512 < nch) {
                const bf16x8& x = (MODE == 0) ? tv[i] : sv[i];
                float y[8];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    y[j] = (MODE == 3) ? bf2f(sv[i][j]) : (MODE == 0 ? bf2f(tv[i][j]) - bf2f(sv[i][j]) : bf2f(sv[i][j]) - bf2f(tv[i][j]));
                lse_chunk_w(ow, x, y);
                if (MODE != 3) lse_chunk(oo, MODE == 0 ? sv[i] : tv[i]);
            }
        }
        wave_merge(ow);
        if (MODE != 3) wave_merge(oo);
        __syncthreads();
        if (l == 0) {
            wscratch[w] = ow.m;
            wscratch[8 + w] = ow.s;
            wscratch[16 + w] = ow.w;
            wscratch[24 + w] = oo.m;
            wscratch[32 + w] = oo.s;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            OnlineLseW tw;
            OnlineLse to;
            tw.m = wscratch[0]; tw.s = wscratch[8]; tw.w = wscratch[16];
            to.m = wscratch[24]; to.s = wscratch[32];
#pragma unroll
            for (int i = 1; i < 8; ++i) {
                tw.merge(wscratch[i], wscratch[8 + i], wscratch[16 + i]);
                to.merge(wscratch[24 + i], wscratch[32 + i]);
            }
            const float lx = tw.m + __logf(tw.s);
            const float lo = (MODE == 3) ? 0.f : to.m + __logf(to.s);
            float* st = stats + (size_t)row * 4;
            st[0] = (MODE == 0) ? lo : lx;   // lse of the student row
            st[1] = (MODE == 0) ? lx : lo;   // lse of the teacher row (0 when unused)
            st[2] = tw.w / tw.s - lx + lo;
            st[3] = 0.f;
        }
        return;
    }
    OnlineLse os, ot;
    os.init();
    ot.init();
#pragma unroll
    for (int i = 0; i < NC; ++i)
        if (threadIdx.x + i * 512 < nch) lse_chunk(os, sv[i]);
    const float ls = finish(os);

    float lt = 0.f;
    if (HAS_T) {
#pragma unroll
        for (int i = 0; i < NC; ++i)
            if (threadIdx.x + i * 512 < nch) lse_chunk(ot, tv[i]);
        lt = finish(ot);
    }
    float acc = 0.f, aux = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        if (threadIdx.x + i * 512 < nch) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                rowloss_term(mode, bf2f(sv[i][j]), HAS_T ? bf2f(tv[i][j]) : bf2f(sv[i][j]), ls, lt, beta, acc, aux);
        }
    }
    acc = block_sum<8>(acc, scratch);
    aux = block_sum<8>(aux, scratch);
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


// ------------------------------------------------------------------------------------------------
// RMU terms (mm/trainer.py:302-326, nlp/dataloader.py:339-367) on the captured decoder-layer activation
// [B, S, H] bf16 (fp32 math, deterministic two-stage reductions: per-block partials, then one block folds them).
//   mse:      sum_i (a_i - ref_i)^2, ref = b_i (teacher activation) or cvec[i % H] (control vector broadcast over
//             every token: the text trainer's form); backward da_i = coef * (a_i - ref_i)
//   seqmean:  mean[b, h] = (1/S) sum_s a[b, s, h]
//   pair:     sum_{i<B, j<Bc, h} (mean[i,h] - ctrl[j,h])^2 — the mm trainer's mse_loss of a [B,H] input against a
//             [B,1,H] target, which broadcasts to [B,B,H] (a quirk of the reference kept on purpose);
//             backward da[i,s,h] = coef * sum_j (mean[i,h] - ctrl[j,h]) / S
// ------------------------------------------------------------------------------------------------
constexpr int RED_MAX_PARTS = 2048;

__global__ __launch_bounds__(256) void fold_parts_kernel(const float* __restrict__ part, int nparts, float* __restrict__ out) {
    __shared__ float scratch[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) s += part[i];
    s = block_sum<4>(s, scratch);
    if (threadIdx.x == 0) out[0] = s;
}

__global__ __launch_bounds__(256) void mse_fwd_kernel(const bf16* __restrict__ a, const bf16* __restrict__ b,
                                                      const float* __restrict__ cvec, size_t n8, int H,
                                                      float* __restrict__ part) {
    __shared__ float scratch[4];
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const bf16x8 av = *reinterpret_cast<const bf16x8*>(a + i * 8);
        if (b) {
            const bf16x8 bv = *reinterpret_cast<const bf16x8*>(b + i * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = bf2f(av[j]) - bf2f(bv[j]);
                s += d * d;
            }
        } else {
            const int h0 = (int)((i * 8) % (size_t)H);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = bf2f(av[j]) - cvec[h0 + j];
                s += d * d;
            }
        }
    }
    s = block_sum<4>(s, scratch);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void mse_bwd_kernel(const bf16* __restrict__ a, const bf16* __restrict__ b,
                                                      const float* __restrict__ cvec, size_t n8, int H,
                                                      const float* __restrict__ coef_ptr, bf16* __restrict__ da) {
    const float coef = *coef_ptr;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const bf16x8 av = *reinterpret_cast<const bf16x8*>(a + i * 8);
        bf16x8 o;
        if (b) {
            const bf16x8 bv = *reinterpret_cast<const bf16x8*>(b + i * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = f2bf(coef * (bf2f(av[j]) - bf2f(bv[j])));
        } else {
            const int h0 = (int)((i * 8) % (size_t)H);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = f2bf(coef * (bf2f(av[j]) - cvec[h0 + j]));
        }
        *reinterpret_cast<bf16x8*>(da + i * 8) = o;
    }
}

// grid (H / 512, B, nsplit): a wave covers 512 features (8 per lane), the 4 waves take rows s0 + w, s0 + w + 4, ...
// of the block's 128-row slice; partial[split][b][h] holds the slice sums.
constexpr int SEQMEAN_ROWS = 128;
__global__ __launch_bounds__(256) void seqmean_part_kernel(const bf16* __restrict__ a, int S, int H,
                                                           float* __restrict__ partial, int B) {
    __shared__ float red[4][512];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int h0 = blockIdx.x * 512 + l * 8;
    const int b = blockIdx.y, split = blockIdx.z;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int s0 = split * SEQMEAN_ROWS, s1 = min(S, s0 + SEQMEAN_ROWS);
    if (h0 < H) {
        for (int s = s0 + w; s < s1; s += 4) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(a + ((size_t)b * S + s) * H + h0);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += bf2f(v[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[w][l * 8 + j] = acc[j];
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 256) {
        const int h = blockIdx.x * 512 + i;
        if (h < H) partial[((size_t)split * B + b) * H + h] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
    }
}
__global__ __launch_bounds__(256) void seqmean_fold_kernel(const float* __restrict__ partial, int nsplit, int BH, float inv_s,
                                                           float* __restrict__ mean) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= BH) return;
    float s = 0.f;
    for (int k = 0; k < nsplit; ++k) s += partial[(size_t)k * BH + i];
    mean[i] = s * inv_s;
}

__global__ __launch_bounds__(256) void rmu_pair_fwd_kernel(const float* __restrict__ mean, const float* __restrict__ ctrl,
                                                           int B, int Bc, int H, float* __restrict__ part) {
    __shared__ float scratch[4];
    float s = 0.f;
    const size_t total = (size_t)B * Bc * H;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int h = (int)(e % H);
        const int i = (int)((e / H) % B);
        const int j = (int)(e / ((size_t)H * B));
        const float d = mean[(size_t)i * H + h] - ctrl[(size_t)j * H + h];
        s += d * d;
    }
    s = block_sum<4>(s, scratch);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// grid (ceil(H/8/256), S, B): da[b, s, h] = coef / S * sum_j (mean[b,h] - ctrl[j,h])
__global__ __launch_bounds__(256) void rmu_pair_bwd_kernel(const float* __restrict__ mean, const float* __restrict__ ctrl,
                                                           int Bc, int S, int H, const float* __restrict__ coef_ptr,
                                                           bf16* __restrict__ da) {
    const int h0 = (blockIdx.x * 256 + threadIdx.x) * 8;
    if (h0 >= H) return;
    const int s = blockIdx.y, b = blockIdx.z;
    const float coef = *coef_ptr / (float)S;
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float m = mean[(size_t)b * H + h0 + j];
        float g = 0.f;
        for (int c = 0; c < Bc; ++c) g += m - ctrl[(size_t)c * H + h0 + j];
        o[j] = f2bf(coef * g);
    }
    *reinterpret_cast<bf16x8*>(da + ((size_t)b * S + s) * H + h0) = o;
}


// Per-sequence evaluation statistics (mm/eval.py:46-92): out[seq] = {sum_t CE (shifted, ignore -100) = get_batch_loss,
// #shifted targets != -100, #labels != -100 of the UNSHIFTED row — the count mm/eval.py:70-71 normalises by}.
__global__ __launch_bounds__(256) void seq_stats_kernel(const float* __restrict__ loss_row, const long long* __restrict__ labels,
                                                        int S, int V, float* __restrict__ out) {
    __shared__ float scratch[4];
    const int seq = blockIdx.x;
    float s = 0.f, cs = 0.f, cu = 0.f;
    for (int t = threadIdx.x; t < S; t += 256) {
        const int r = seq * S + t;
        s += loss_row[r];
        const long long tgt = shifted_target(labels, r, S, 1);
        cs += (tgt >= 0 && tgt < V) ? 1.f : 0.f;
        cu += labels[r] != -100 ? 1.f : 0.f;
    }
    s = block_sum<4>(s, scratch);
    cs = block_sum<4>(cs, scratch);
    cu = block_sum<4>(cu, scratch);
    if (threadIdx.x == 0) {
        out[seq * 3 + 0] = s;
        out[seq * 3 + 1] = cs;
        out[seq * 3 + 2] = cu;
    }
}

// logits2probs (mm/trainer_utils.py:16-21) as a materialising kernel, for callers that use the helper by name: out[row] =
// log_softmax(logits[row]) (log_ != 0) or softmax(logits[row]), fp32 like the reference's autocast path; lse[row] is kept
// for the backward.  One block per row: one walk for the log-sum-exp, one for the output (the second comes from L2).
__global__ __launch_bounds__(256) void logprobs_fwd_kernel(const bf16* __restrict__ logits, int ldv, float* __restrict__ out,
                                                           float* __restrict__ lse, int V, int log_) {
    __shared__ float scratch[8];
    const int row = blockIdx.x;
    const bf16* lr = logits + (size_t)row * ldv;
    const float l = block_lse(lr, V, scratch);
    if (threadIdx.x == 0) lse[row] = l;
    float* orow = out + (size_t)row * V;
    const int nch = V >> 3;
    for (int ch = threadIdx.x; ch < nch; ch += 256) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(lr + ch * 8);
        f32x4 a, b;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float x0 = bf2f(v[j]) - l, x1 = bf2f(v[4 + j]) - l;
            a[j] = log_ ? x0 : __expf(x0);
            b[j] = log_ ? x1 : __expf(x1);
        }
        *reinterpret_cast<f32x4*>(orow + ch * 8) = a;
        *reinterpret_cast<f32x4*>(orow + ch * 8 + 4) = b;
    }
}

// backward of the above for an upstream gradient g [M,V] fp32: log-softmax: dx = g - p * sum_v g; softmax: dx = p * (g -
// sum_v g p), p = exp(x - lse).  dlogits bf16.
__global__ __launch_bounds__(256) void logprobs_bwd_kernel(const bf16* __restrict__ logits, int ldv, const float* __restrict__ lse,
                                                           const float* __restrict__ g, bf16* __restrict__ dlogits, int ldd,
                                                           int V, int log_) {
    __shared__ float scratch[4];
    const int row = blockIdx.x;
    const bf16* lr = logits + (size_t)row * ldv;
    const float* gr = g + (size_t)row * V;
    const float l = lse[row];
    const int nch = V >> 3;
    float acc = 0.f;
    for (int ch = threadIdx.x; ch < nch; ch += 256) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(gr + ch * 8), b = *reinterpret_cast<const f32x4*>(gr + ch * 8 + 4);
        if (log_) {
            acc += (a[0] + a[1]) + (a[2] + a[3]) + (b[0] + b[1]) + (b[2] + b[3]);
        } else {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(lr + ch * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc += a[j] * __expf(bf2f(v[j]) - l) + b[j] * __expf(bf2f(v[4 + j]) - l);
        }
    }
    const float tot = block_sum<4>(acc, scratch);
    bf16* dr = dlogits + (size_t)row * ldd;
    for (int ch = threadIdx.x; ch < nch; ch += 256) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(lr + ch * 8);
        const f32x4 a = *reinterpret_cast<const f32x4*>(gr + ch * 8), b = *reinterpret_cast<const f32x4*>(gr + ch * 8 + 4);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float p = __expf(bf2f(v[j]) - l), gj = j < 4 ? a[j] : b[j - 4];
            o[j] = f2bf(log_ ? gj - p * tot : p * (gj - tot));
        }
        *reinterpret_cast<bf16x8*>(dr + ch * 8) = o;
    }
}
}  // namespace

static int g_rowloss_streaming = 0;

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
    // mode 4 (the text trainer's NPO form: powf per element) does not fit 128 registers without scratch: streaming kernel
    if (V <= 512 * 8 * 8 && !g_rowloss_streaming && mode != 4) {
#define MMU_ROWLOSS_REGS(MODE_)                                                                                   \
    hipLaunchKernelGGL(rowloss_fwd_regs_kernel<MODE_>, dim3(M), dim3(512), 0, (hipStream_t)stream, (const bf16*)slog, \
                       (const bf16*)tlog, ldv, beta, stats, M, V)
        switch (mode) {
            case 0: MMU_ROWLOSS_REGS(0); break;
            case 1: MMU_ROWLOSS_REGS(1); break;
            case 2: MMU_ROWLOSS_REGS(2); break;
            default: MMU_ROWLOSS_REGS(3); break;
        }
#undef MMU_ROWLOSS_REGS
    } else {
        hipLaunchKernelGGL(rowloss_fwd_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, mode, (const bf16*)slog,
                           (const bf16*)tlog, ldv, beta, stats, M, V);
    }
    MMU_CHECK_LAUNCH();
    return 0;
}

// A/B and test switch: 1 forces the streaming (two-walk) row-loss kernel for every V.
extern "C" int mmu_rowloss_set_streaming(int on) {
    g_rowloss_streaming = on ? 1 : 0;
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

// ---- RMU terms.  out / scratch buffers: 1 + mmu_red_scratch_floats() floats, result in slot 0.
extern "C" int mmu_red_scratch_floats(void) { return RED_MAX_PARTS; }

extern "C" int mmu_mse_fwd(const void* a, const void* b, const float* cvec, long long n, int H, float* out, void* stream) {
    if (n <= 0 || (n & 7) || H <= 0 || (H & 7) || (!b && !cvec) || !out) return -1;
    const size_t n8 = (size_t)n / 8;
    size_t grid = (n8 + 255) / 256;
    if (grid > RED_MAX_PARTS) grid = RED_MAX_PARTS;
    hipLaunchKernelGGL(mse_fwd_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, (const bf16*)a, (const bf16*)b,
                       cvec, n8, H, out + 1);
    MMU_CHECK_LAUNCH();
    hipLaunchKernelGGL(fold_parts_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, out + 1, (int)grid, out);
    MMU_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmu_mse_bwd(const void* a, const void* b, const float* cvec, long long n, int H, const float* coef, void* da,
                           void* stream) {
    if (n <= 0 || (n & 7) || H <= 0 || (H & 7) || (!b && !cvec) || !coef || !da) return -1;
    const size_t n8 = (size_t)n / 8;
    size_t grid = (n8 + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(mse_bwd_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, (const bf16*)a, (const bf16*)b,
                       cvec, n8, H, coef, (bf16*)da);
    MMU_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmu_seqmean_nsplit(int S) { return (S + SEQMEAN_ROWS - 1) / SEQMEAN_ROWS; }

// partial: mmu_seqmean_nsplit(S) * B * H floats of scratch; mean: [B, H] floats.
extern "C" int mmu_seqmean_fwd(const void* a, int B, int S, int H, float* partial, float* mean, void* stream) {
    if (B <= 0 || S <= 0 || H <= 0 || (H & 7) || !partial || !mean) return -1;
    const int ns = mmu_seqmean_nsplit(S);
    hipLaunchKernelGGL(seqmean_part_kernel, dim3((H + 511) / 512, B, ns), dim3(256), 0, (hipStream_t)stream, (const bf16*)a,
                       S, H, partial, B);
    MMU_CHECK_LAUNCH();
    hipLaunchKernelGGL(seqmean_fold_kernel, dim3((B * H + 255) / 256), dim3(256), 0, (hipStream_t)stream, partial, ns, B * H,
                       1.f / (float)S, mean);
    MMU_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmu_rmu_pair_fwd(const float* mean, const float* ctrl, int B, int Bc, int H, float* out, void* stream) {
    if (B <= 0 || Bc <= 0 || H <= 0 || !mean || !ctrl || !out) return -1;
    size_t grid = ((size_t)B * Bc * H + 255) / 256;
    if (grid > RED_MAX_PARTS) grid = RED_MAX_PARTS;
    hipLaunchKernelGGL(rmu_pair_fwd_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, mean, ctrl, B, Bc, H, out + 1);
    MMU_CHECK_LAUNCH();
    hipLaunchKernelGGL(fold_parts_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, out + 1, (int)grid, out);
    MMU_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmu_rmu_pair_bwd(const float* mean, const float* ctrl, int B, int Bc, int S, int H, const float* coef, void* da,
                                void* stream) {
    if (B <= 0 || Bc <= 0 || S <= 0 || H <= 0 || (H & 7) || !mean || !ctrl || !coef || !da || S > 65535 || B > 65535) return -1;
    hipLaunchKernelGGL(rmu_pair_bwd_kernel, dim3((H / 8 + 255) / 256, S, B), dim3(256), 0, (hipStream_t)stream, mean, ctrl, Bc,
                       S, H, coef, (bf16*)da);
    MMU_CHECK_LAUNCH();
    return 0;
}

// Evaluation pass: per-sequence CE sum + token counts for n_seq sequences of S rows in ONE call (row log-sum-exp and
// row loss land in the two scratch vectors of n_seq * S floats; out: [n_seq][3]).
extern "C" int mmu_seq_ce_stats(const void* logits, int ldv, const long long* labels, int n_seq, int S, int V,
                                float* lse_scratch, float* row_scratch, float* out, void* stream) {
    if (n_seq <= 0 || S <= 0 || (V & 7) || (ldv & 7) || !labels || !lse_scratch || !row_scratch || !out) return -1;
    hipLaunchKernelGGL(ce_fwd_kernel, dim3(n_seq * S), dim3(256), 0, (hipStream_t)stream, (const bf16*)logits, ldv, labels,
                       lse_scratch, row_scratch, n_seq * S, V, S, 1);
    MMU_CHECK_LAUNCH();
    hipLaunchKernelGGL(seq_stats_kernel, dim3(n_seq), dim3(256), 0, (hipStream_t)stream, row_scratch, labels, S, V, out);
    MMU_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmu_logprobs_fwd(const void* logits, int ldv, float* out, float* lse, int M, int V, int log_, void* stream) {
    if (M <= 0 || (V & 7) || (ldv & 7)) return -1;
    hipLaunchKernelGGL(logprobs_fwd_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, (const bf16*)logits, ldv, out, lse, V,
                       log_);
    MMU_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmu_logprobs_bwd(const void* logits, int ldv, const float* lse, const float* g, void* dlogits, int ldd, int M,
                                int V, int log_, void* stream) {
    if (M <= 0 || (V & 7) || (ldv & 7) || (ldd & 7)) return -1;
    hipLaunchKernelGGL(logprobs_bwd_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, (const bf16*)logits, ldv, lse, g,
                       (bf16*)dlogits, ldd, V, log_);
    MMU_CHECK_LAUNCH();
    return 0;
}
