// Optimiser-side HBM-bound kernels for gfx950 over flat bf16 parameter / gradient buffers:
// grad-norm^2, fused clip + AdamW (decoupled weight decay, fp32 moments) + optional grad zeroing.
// Replaces the step that follows compute_loss in the HF loop the reference inherits
// (clip_grad_norm_(1.0) -> optimizer.step() -> zero_grad; mm/forget.py:139-168 configures adamw / lr / wd).
// The reference's bitsandbytes 8-bit state layout is absent from /root/reference (parity unpinned);
// fp32-moment AdamW with torch.optim.AdamW update order is the defined behaviour here.
// Algorithmic bytes per parameter: read p2+g2+m4+v4, write p2+m4+v4(+g2) = 22 (24 with zeroing).
#include "common.h"

namespace {

// Sum of squares in two passes with a FIXED summation order (per-block partials, then one block folds them): every
// data-parallel rank derives the clip factor from the same all-reduced gradient and must get the same bits, or the
// replicas drift apart (a float atomicAdd over the blocks differs in the last ulp from run to run).
constexpr int SUMSQ_MAX_PARTS = 2048;
__global__ __launch_bounds__(256) void sumsq_kernel(const bf16* __restrict__ g, size_t n8, float* __restrict__ part) {
    __shared__ float scratch[4];
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(g + i * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float f = bf2f(v[j]);
            s += f * f;
        }
    }
    s = block_sum<4>(s, scratch);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void sumsq_fold_kernel(const float* __restrict__ part, int nparts, float* __restrict__ out) {
    __shared__ float scratch[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) s += part[i];
    s = block_sum<4>(s, scratch);
    if (threadIdx.x == 0) out[0] = s;
}

// L1 / L0 "norm" of d = bf16(p - t) (t = null: d = p): out[0] += sum |d|, out[1] += #(d != 0)  (fp64 accumulators).
// mm/trainer.py:451-481 builds torch.cat of every (param - teacher_param) and calls torch.norm(p=1 / p=0) on it: a
// 13.5 GB temporary at 7B; here it is one streaming pass over the two flat buffers.
__global__ __launch_bounds__(256) void l1l0_fwd_kernel(const bf16* __restrict__ p, const bf16* __restrict__ t, size_t n8,
                                                       double* __restrict__ out) {
    __shared__ float scratch[4];
    float s = 0.f, c = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(p + i * 8);
        bf16x8 b;
        if (t) b = *reinterpret_cast<const bf16x8*>(t + i * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float d = t ? round_bf16(bf2f(a[j]) - bf2f(b[j])) : bf2f(a[j]);
            s += fabsf(d);
            c += d != 0.f ? 1.f : 0.f;
        }
    }
    s = block_sum<4>(s, scratch);
    __syncthreads();
    c = block_sum<4>(c, scratch);
    if (threadIdx.x == 0) {
        atomicAdd(out, (double)s);
        atomicAdd(out + 1, (double)c);
    }
}

// g += (lambda1 * *gout) * sign(bf16(p - t))   (the L0 term has zero gradient, as torch.norm(p=0) does)
__global__ __launch_bounds__(256) void l1_bwd_kernel(const bf16* __restrict__ p, const bf16* __restrict__ t, size_t n8,
                                                     bf16* __restrict__ g, float lambda1, const float* __restrict__ gout) {
    const float coef = lambda1 * (gout ? *gout : 1.f);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(p + i * 8);
        bf16x8 b;
        if (t) b = *reinterpret_cast<const bf16x8*>(t + i * 8);
        bf16x8 gv = *reinterpret_cast<const bf16x8*>(g + i * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float d = t ? round_bf16(bf2f(a[j]) - bf2f(b[j])) : bf2f(a[j]);
            const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
            gv[j] = f2bf(bf2f(gv[j]) + coef * sg);
        }
        *reinterpret_cast<bf16x8*>(g + i * 8) = gv;
    }
}

// Element ranges [lo, hi) of p (multiples of 8, sorted, disjoint) that get NO weight decay: transformers.Trainer puts
// biases and norm weights in a weight_decay = 0 group (create_optimizer / get_decay_parameter_names), which is the loop
// the reference trains under (mm/forget.py:139-190).  The table is staged in LDS and searched once per 8-element group.
constexpr int MAX_NODECAY = 512;
__device__ __forceinline__ bool in_nodecay(const long long* __restrict__ nd, int n, long long e) {
    int lo = 0, hi = n;  // first range with start > e
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (nd[2 * mid] <= e) lo = mid + 1; else hi = mid;
    }
    return lo > 0 && e < nd[2 * lo - 1];
}

struct AdamArgs {
    const long long* nodecay;
    int n_nodecay;
    bf16* p;
    bf16* g;
    float* m;
    float* v;
    size_t n8;
    float lr, beta1, beta2, eps, wd, bc1, bc2_sqrt;
    const float* sumsq;  // device: total grad norm^2 (already summed over every bucket), or null
    float max_norm;      // <= 0: no clipping
    float grad_scale;    // static multiplier on g (e.g. 1/world for a sum all-reduce)
    int zero_grad;
};

__global__ __launch_bounds__(256) void adamw_kernel(AdamArgs a) {
    __shared__ long long nd[2 * MAX_NODECAY];
    for (int i = threadIdx.x; i < 2 * a.n_nodecay; i += 256) nd[i] = a.nodecay[i];
    if (a.n_nodecay) __syncthreads();
    float gs = a.grad_scale;
    if (a.sumsq && a.max_norm > 0.f) {
        const float tn = sqrtf(*a.sumsq) * a.grad_scale;
        gs *= fminf(1.f, a.max_norm / (tn + 1e-6f));
    }
    const float step = a.lr / a.bc1;
    const float decay_on = 1.f - a.lr * a.wd;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n8; i += (size_t)gridDim.x * blockDim.x) {
        const float decay = (a.n_nodecay && in_nodecay(nd, a.n_nodecay, (long long)(i * 8))) ? 1.f : decay_on;
        const bf16x8 pv = *reinterpret_cast<const bf16x8*>(a.p + i * 8);
        const bf16x8 gv = *reinterpret_cast<const bf16x8*>(a.g + i * 8);
        f32x4 m0 = *reinterpret_cast<const f32x4*>(a.m + i * 8), m1 = *reinterpret_cast<const f32x4*>(a.m + i * 8 + 4);
        f32x4 v0 = *reinterpret_cast<const f32x4*>(a.v + i * 8), v1 = *reinterpret_cast<const f32x4*>(a.v + i * 8 + 4);
        bf16x8 po;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float g = bf2f(gv[j]) * gs;
            float mm = (j < 4 ? m0[j & 3] : m1[j & 3]);
            float vv = (j < 4 ? v0[j & 3] : v1[j & 3]);
            mm = a.beta1 * mm + (1.f - a.beta1) * g;
            vv = a.beta2 * vv + (1.f - a.beta2) * g * g;
            float p = bf2f(pv[j]) * decay;
            p -= step * mm / (sqrtf(vv) / a.bc2_sqrt + a.eps);
            po[j] = f2bf(p);
            if (j < 4) { m0[j & 3] = mm; v0[j & 3] = vv; } else { m1[j & 3] = mm; v1[j & 3] = vv; }
        }
        *reinterpret_cast<bf16x8*>(a.p + i * 8) = po;
        *reinterpret_cast<f32x4*>(a.m + i * 8) = m0;
        *reinterpret_cast<f32x4*>(a.m + i * 8 + 4) = m1;
        *reinterpret_cast<f32x4*>(a.v + i * 8) = v0;
        *reinterpret_cast<f32x4*>(a.v + i * 8 + 4) = v1;
        if (a.zero_grad) {
            bf16x8 z;
#pragma unroll
            for (int j = 0; j < 8; ++j) z[j] = f2bf(0.f);
            *reinterpret_cast<bf16x8*>(a.g + i * 8) = z;
        }
    }
}

// AdamW with BLOCK-WISE 8-BIT moments (the reference trains with optim="adamw_bnb_8bit", mm/forget.py:156; bitsandbytes
// 0.43.3 is not in /root/reference, so this restates the published algorithm — Dettmers et al., "8-bit Optimizers via
// Block-wise Quantization" — and is "parity unpinned" against bnb itself): each block of 2048 parameters keeps its
// two moments as 8-bit codes of a 256-entry dynamic (tree) quantisation map — signed for m, unsigned for v — times one
// fp32 absmax per block.  One workgroup = one block: dequantise, update, take the block's new absmax, apply the step to
// the bf16 parameters, requantise to the nearest code.  10 B/param of HBM traffic instead of 24.
constexpr int Q8_BLOCK = 2048;
struct Adam8Args {
    const long long* nodecay;
    int n_nodecay;
    bf16* p;
    bf16* g;
    unsigned char* c1;
    unsigned char* c2;
    float* am1;
    float* am2;
    const float* map1;  // 256 ascending values in [-1, 1]
    const float* map2;  // 256 ascending values in [0, 1]
    size_t n;
    float lr, beta1, beta2, eps, wd, bc1, bc2_sqrt;
    const float* sumsq;
    float max_norm, grad_scale;
    int zero_grad;
};

__device__ __forceinline__ int q8_nearest(const float* __restrict__ map, float x) {
    // largest i with (map[i-1] + map[i]) / 2 <= x  == index of the nearest code (ties to the upper one)
    int lo = 0, hi = 255;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int mid = (lo + hi + 1) >> 1;
        const bool up = x >= 0.5f * (map[mid - 1] + map[mid]);
        lo = up ? mid : lo;
        hi = up ? hi : mid - 1;
    }
    return lo;
}

__global__ __launch_bounds__(256) void adamw8bit_kernel(Adam8Args a) {
    __shared__ float m1s[256], m2s[256];
    __shared__ float red[2][4];
    __shared__ long long nd[2 * MAX_NODECAY];
    for (int i = threadIdx.x; i < 2 * a.n_nodecay; i += 256) nd[i] = a.nodecay[i];
    m1s[threadIdx.x] = a.map1[threadIdx.x];
    m2s[threadIdx.x] = a.map2[threadIdx.x];
    float gs = a.grad_scale;
    if (a.sumsq && a.max_norm > 0.f) {
        const float tn = sqrtf(*a.sumsq) * a.grad_scale;
        gs *= fminf(1.f, a.max_norm / (tn + 1e-6f));
    }
    const float step = a.lr / a.bc1;
    const size_t blk = blockIdx.x;
    const size_t i0 = blk * Q8_BLOCK + (size_t)threadIdx.x * 8;
    const bool live = i0 < a.n;  // n % 8 == 0: a thread's 8 elements are all in or all out
    const float am1 = a.am1[blk], am2 = a.am2[blk];
    __syncthreads();
    float m[8], v[8];
    bf16x8 pv, gv;
    unsigned long long k1 = 0, k2 = 0;
    float mx1 = 0.f, mx2 = 0.f;
    if (live) {
        pv = *reinterpret_cast<const bf16x8*>(a.p + i0);
        gv = *reinterpret_cast<const bf16x8*>(a.g + i0);
        k1 = *reinterpret_cast<const unsigned long long*>(a.c1 + i0);
        k2 = *reinterpret_cast<const unsigned long long*>(a.c2 + i0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float g = bf2f(gv[j]) * gs;
            m[j] = a.beta1 * (m1s[(k1 >> (8 * j)) & 255] * am1) + (1.f - a.beta1) * g;
            v[j] = a.beta2 * (m2s[(k2 >> (8 * j)) & 255] * am2) + (1.f - a.beta2) * g * g;
            mx1 = fmaxf(mx1, fabsf(m[j]));
            mx2 = fmaxf(mx2, v[j]);
        }
    }
    mx1 = wave_max(mx1);
    mx2 = wave_max(mx2);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = mx1; red[1][threadIdx.x >> 6] = mx2; }
    __syncthreads();
    const float nm1 = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
    const float nm2 = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
    if (threadIdx.x == 0) { a.am1[blk] = nm1; a.am2[blk] = nm2; }
    if (!live) return;
    const float r1 = nm1 > 0.f ? 1.f / nm1 : 0.f, r2 = nm2 > 0.f ? 1.f / nm2 : 0.f;
    const float decay = (a.n_nodecay && in_nodecay(nd, a.n_nodecay, (long long)i0)) ? 1.f : 1.f - a.lr * a.wd;
    bf16x8 po;
    unsigned long long o1 = 0, o2 = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float p = bf2f(pv[j]) * decay;
        p -= step * m[j] / (sqrtf(v[j]) / a.bc2_sqrt + a.eps);
        po[j] = f2bf(p);
        o1 |= (unsigned long long)q8_nearest(m1s, m[j] * r1) << (8 * j);
        o2 |= (unsigned long long)q8_nearest(m2s, v[j] * r2) << (8 * j);
    }
    *reinterpret_cast<bf16x8*>(a.p + i0) = po;
    *reinterpret_cast<unsigned long long*>(a.c1 + i0) = o1;
    *reinterpret_cast<unsigned long long*>(a.c2 + i0) = o2;
    if (a.zero_grad) {
        bf16x8 z;
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = f2bf(0.f);
        *reinterpret_cast<bf16x8*>(a.g + i0) = z;
    }
}

__global__ void scale_kernel(bf16* __restrict__ x, size_t n8, float s) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        bf16x8 v = *reinterpret_cast<bf16x8*>(x + i * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = f2bf(bf2f(v[j]) * s);
        *reinterpret_cast<bf16x8*>(x + i * 8) = v;
    }
}

}  // namespace

// out_sumsq[0] <- sum(g^2), n % 8 == 0; deterministic (see sumsq_kernel).
extern "C" int mmu_sumsq_bf16(const void* g, long long n, float* out_sumsq, void* stream) {
    if (n <= 0 || (n & 7)) return -1;
    const size_t n8 = (size_t)n / 8;
    size_t grid = (n8 + 255) / 256;
    if (grid > 2048) grid = 2048;
    // out_sumsq[0] <- sum g^2; out_sumsq[1 .. 1 + mmu_sumsq_scratch_floats()) is scratch for the per-block partials
    hipLaunchKernelGGL(sumsq_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, (const bf16*)g, n8, out_sumsq + 1);
    MMU_CHECK_LAUNCH();
    hipLaunchKernelGGL(sumsq_fold_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, out_sumsq + 1, (int)grid, out_sumsq);
    MMU_CHECK_LAUNCH();
    return 0;
}
extern "C" int mmu_sumsq_scratch_floats(void) { return SUMSQ_MAX_PARTS; }

extern "C" int mmu_l1l0_fwd(const void* p, const void* t, long long n, double* out2, void* stream) {
    if (n <= 0 || (n & 7) || !out2) return -1;
    const size_t n8 = (size_t)n / 8;
    size_t grid = (n8 + 255) / 256;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(l1l0_fwd_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, (const bf16*)p, (const bf16*)t,
                       n8, out2);
    MMU_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmu_l1_bwd(const void* p, const void* t, long long n, void* g, float lambda1, const float* gout,
                          void* stream) {
    if (n <= 0 || (n & 7) || !g) return -1;
    const size_t n8 = (size_t)n / 8;
    size_t grid = (n8 + 255) / 256;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(l1_bwd_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, (const bf16*)p, (const bf16*)t,
                       n8, (bf16*)g, lambda1, gout);
    MMU_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmu_adamw_bf16(void* p, void* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                              float eps, float weight_decay, int step, const float* sumsq, float max_norm,
                              float grad_scale, int zero_grad, const long long* nodecay, int n_nodecay, void* stream) {
    if (n <= 0 || (n & 7) || step < 1 || n_nodecay < 0 || n_nodecay > MAX_NODECAY || (n_nodecay && !nodecay)) return -1;
    AdamArgs a;
    a.nodecay = nodecay; a.n_nodecay = n_nodecay;
    a.p = (bf16*)p; a.g = (bf16*)g; a.m = m; a.v = v; a.n8 = (size_t)n / 8;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = weight_decay;
    a.bc1 = 1.f - powf(beta1, (float)step);
    a.bc2_sqrt = sqrtf(1.f - powf(beta2, (float)step));
    a.sumsq = sumsq; a.max_norm = max_norm; a.grad_scale = grad_scale; a.zero_grad = zero_grad;
    size_t grid = (a.n8 + 255) / 256;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(adamw_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, a);
    MMU_CHECK_LAUNCH();
    return 0;
}

// codes1 / codes2: n bytes each; absmax1 / absmax2: mmu_adamw8bit_nblocks(n) floats each; map1 / map2: 256 floats
// (device).  A fresh state is codes = index of 0.0 in the map, absmax = 0.
extern "C" int mmu_adamw8bit_nblocks(long long n) { return (int)((n + Q8_BLOCK - 1) / Q8_BLOCK); }
extern "C" int mmu_adamw8bit_bf16(void* p, void* g, void* codes1, void* codes2, float* absmax1, float* absmax2,
                                  const float* map1, const float* map2, long long n, float lr, float beta1, float beta2,
                                  float eps, float weight_decay, int step, const float* sumsq, float max_norm,
                                  float grad_scale, int zero_grad, const long long* nodecay, int n_nodecay, void* stream) {
    if (n_nodecay < 0 || n_nodecay > MAX_NODECAY || (n_nodecay && !nodecay)) return -1;
    if (n <= 0 || (n & 7) || step < 1 || !codes1 || !codes2 || !absmax1 || !absmax2 || !map1 || !map2) return -1;
    if (((uintptr_t)codes1 | (uintptr_t)codes2) & 7) return -1;
    Adam8Args a;
    a.nodecay = nodecay; a.n_nodecay = n_nodecay;
    a.p = (bf16*)p; a.g = (bf16*)g; a.c1 = (unsigned char*)codes1; a.c2 = (unsigned char*)codes2;
    a.am1 = absmax1; a.am2 = absmax2; a.map1 = map1; a.map2 = map2; a.n = (size_t)n;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = weight_decay;
    a.bc1 = 1.f - powf(beta1, (float)step);
    a.bc2_sqrt = sqrtf(1.f - powf(beta2, (float)step));
    a.sumsq = sumsq; a.max_norm = max_norm; a.grad_scale = grad_scale; a.zero_grad = zero_grad;
    hipLaunchKernelGGL(adamw8bit_kernel, dim3(mmu_adamw8bit_nblocks(n)), dim3(256), 0, (hipStream_t)stream, a);
    MMU_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmu_scale_bf16(void* x, long long n, float s, void* stream) {
    if (n <= 0 || (n & 7)) return -1;
    const size_t n8 = (size_t)n / 8;
    size_t grid = (n8 + 255) / 256;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(scale_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, (bf16*)x, n8, s);
    MMU_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmu_abi_version(void) { return 4; }  // 4: round 4 added entry points (fused SwiGLU GEMMs, logprobs, LayerNorm parameter gradients, dropout, queue reset); no existing signature changed
