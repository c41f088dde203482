// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the forget/retain step.
// Wave = 64 lanes everywhere; no other target is supported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define MMU_LDS __attribute__((address_space(3)))
#define MMU_GLOBAL __attribute__((address_space(1)))

#define MMU_CHECK_LAUNCH()                      \
    do {                                        \
        hipError_t e__ = hipGetLastError();     \
        if (e__ != hipSuccess) return (int)e__; \
    } while (0)

// 16 bytes of zeros: out-of-range tile chunks are redirected here so that LDS-DMA staging
// zero-fills tails without a per-element branch in the main loop.
static __device__ __attribute__((aligned(16))) const uint32_t mmu_zero16[4] = {0u, 0u, 0u, 0u};

// bf16 <-> f32 through the bit pattern.  A plain `(float)(bf16)x` round trip is NOT used for intermediate
// roundings: clang keeps bf16 expressions in excess (f32) precision and elides the rounding, which changes
// results where the reference rounds every bf16 tensor op (measured on gfx950: RoPE off by 1 ulp).
// +inf words: LDS-DMA source for out-of-range log-sum-exp entries (exp2(x - inf) = 0)
static __device__ __attribute__((aligned(16))) const uint32_t mmu_inf16[4] = {0x7f800000u, 0x7f800000u, 0x7f800000u,
                                                                              0x7f800000u};

__device__ __forceinline__ float bf2f(bf16 x) {
    return __uint_as_float(((uint32_t)__builtin_bit_cast(unsigned short, x)) << 16);
}
__device__ __forceinline__ bf16 f2bf(float x) { return (bf16)x; }  // v_cvt_pk_bf16_f32: RNE, NaN-preserving
__device__ __forceinline__ float round_bf16(float x) { return bf2f((bf16)x); }

// async global -> LDS, 16 B per lane; LDS destination = wave-uniform base + lane*16.
// MMU_GLDS_AUX: cache-policy bits of the load (A/B builds only; sc0 = 1, nt = 2, sc1 = 16; measured on the GEMM,
// profiles/r03_glds_cache_policy_ab.txt: the default policy is the fastest)
#ifndef MMU_GLDS_AUX
#define MMU_GLDS_AUX 0
#endif
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const MMU_GLOBAL void*)gsrc, (MMU_LDS void*)lds_wave_base, 16, 0, MMU_GLDS_AUX);
}

// async global -> LDS, 4 B per lane; LDS destination = wave-uniform base + lane*4.
__device__ __forceinline__ void glds4(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const MMU_GLOBAL void*)gsrc, (MMU_LDS void*)lds_wave_base, 4, 0, 0);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide sum for blockDim.x == 64*NW; scratch must hold NW floats. All threads get the result.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* scratch) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) scratch[w] = v;
    __syncthreads();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) r += scratch[i];
    return r;
}
template <int NW>
__device__ __forceinline__ float block_max(float v, float* scratch) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) scratch[w] = v;
    __syncthreads();
    float r = scratch[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) r = fmaxf(r, scratch[i]);
    return r;
}
