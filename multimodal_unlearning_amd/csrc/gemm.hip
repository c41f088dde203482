// bf16 MFMA GEMM for gfx950: C[M,N] = epilogue(sum_k Aop[m,k] * Bop[n,k]), fp32 accumulate.
//
// Stands in for every nn.Linear forward / dgrad / wgrad that transformers' LLaVA executes under
// mm/trainer.py:180-433 `model(**inputs)` and the autograd backward behind mm/forget.py:190
// (SURVEY.md §8a rows K1,K3,K5,K8,K11,K12,K13).
//
// Operand storage (AT / BT template flags):
//   AT=false: A stored [M][K]  (reduction dim contiguous)   AT=true: A stored [K][M]
//   BT=false: B stored [N][K]  (torch Linear weight layout) BT=true: B stored [K][N]
//   forward  y = x W^T          : AT=0 BT=0  (A=x [M,K],  B=W [N,K])
//   dgrad    dx = dy W          : AT=0 BT=1  (A=dy [M,Nw], B=W [Nw(red),K(out)])
//   wgrad    dW = dy^T x        : AT=1 BT=1  (A=dy [Mtok(red),Nw], B=x [Mtok(red),K])
//
// Structure: 128x128x64 tile, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 MFMA
// 16x16x32 tiles. Global->LDS by LDS-DMA (global_load_lds_dwordx4, 16 B/lane), double-buffered,
// one barrier per K tile. LDS images are lane-linear (DMA requirement) with the XOR swizzle applied
// on the per-lane SOURCE address and on the fragment read (guide rule 21):
//   reduction-contiguous tile [128 rows][64 k]: 128-B rows, chunk' = chunk ^ (row & 7)   -> ds_read_b128
//   reduction-major tile      [64 k][128 cols]: 256-B rows, chunk' = chunk ^ X(k),
//        X(k) = ((k&3)<<1)|(((k>>3)&1)<<3)                                              -> ds_read_b64_tr_b16
// Tails (any M, N; K) are zero-filled by redirecting the per-lane source to a 16-B zero page.
// The MFMA is issued with operands swapped (weight/N fragment as A, token/M fragment as B) so each
// lane ends up with 4 consecutive n of one output row -> 8-byte bf16x4 stores.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * 64 * 2;  // 16 KiB per operand tile

struct GemmArgs {
    const bf16* A;
    const bf16* B;
    bf16* C;
    const bf16* bias;   // [N] or null
    const bf16* resid;  // [M][ldr] or null
    int M, N, K;
    int lda, ldb, ldc, ldr;
    int act;         // 0 none, 1 gelu(erf), 2 quick_gelu
    int accumulate;  // C = bf16(float(C) + acc) (grad accumulation)
    int tiles_m, tiles_n;
};

__device__ __forceinline__ int swzT(int k) { return ((k & 3) << 1) | (((k >> 3) & 1) << 3); }

// Stage one 128x64 operand tile. T=false: G is [rows][K] (ld = row stride); T=true: G is [K][rows].
template <bool T>
__device__ __forceinline__ void stage_tile(char* lds_tile, const bf16* __restrict__ G, int ld, int row0, int k0,
                                           int rows_dim, int Kd, int wave, int lane) {
    const bf16* zero = reinterpret_cast<const bf16*>(mmu_zero16);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = wave * 4 + i;
        const bf16* src;
        if (!T) {
            const int r = t * 8 + (lane >> 3);
            const int c = (lane & 7) ^ (r & 7);
            const int gr = row0 + r, gk = k0 + c * 8;
            src = (gr < rows_dim && gk < Kd) ? (G + (size_t)gr * ld + gk) : zero;
        } else {
            const int kr = t * 4 + (lane >> 4);
            const int c = (lane & 15) ^ swzT(kr);
            const int gk = k0 + kr, gm = row0 + c * 8;
            src = (gk < Kd && gm < rows_dim) ? (G + (size_t)gk * ld + gm) : zero;
        }
        glds16(src, lds_tile + t * 1024);
    }
}

// Fragment for 16 rows (or columns) starting at `base`, k-step ks (32 k each):
// lane l holds index base + (l&15), k = ks*32 + 8*(l>>4) + j, j = 0..7.
template <bool T>
__device__ __forceinline__ bf16x8 read_frag(const MMU_LDS char* tile, int base, int ks, int lane) {
    if (!T) {
        const int r = base + (lane & 15);
        const int cl = ks * 4 + (lane >> 4);
        return *(const MMU_LDS bf16x8*)(tile + r * 128 + ((cl ^ (r & 7)) << 4));
    } else {
        const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
        const int k = ks * 32 + 8 * g + q;
        const int chunk = (base >> 3) + (p >> 1);
        const MMU_LDS char* a0 = tile + k * 256 + ((chunk ^ swzT(k)) << 4) + ((p & 1) << 3);
        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((MMU_LDS bf16x4*)(a0));
        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((MMU_LDS bf16x4*)(a0 + 4 * 256));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
}

__device__ __forceinline__ float act_apply(float v, int act) {
    if (act == 1) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
    if (act == 2) return v / (1.0f + __expf(-1.702f * v));
    return v;
}

template <bool AT, bool BT, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware (bijective) remap, then grouped ordering so that co-resident tiles share panels in L2.
    int pid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = pid & 7, loc = pid >> 3;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    constexpr int GROUP_M = 8;
    const int per_group = GROUP_M * p.tiles_n;
    const int gid = pid / per_group;
    const int first_m = gid * GROUP_M;
    const int gsz = min(p.tiles_m - first_m, GROUP_M);
    const int tile_m = first_m + (pid % per_group) % gsz;
    const int tile_n = (pid % per_group) / gsz;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // LDS map: [buf0: A | B][buf1: A | B], 16 KiB each

    f32x4 acc[4][4];  // [nt][mt]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (p.K + BK - 1) / BK;
    stage_tile<AT>(smem, p.A, p.lda, m0, 0, p.M, p.K, wave, lane);
    stage_tile<BT>(smem + TILE_BYTES, p.B, p.ldb, n0, 0, p.N, p.K, wave, lane);

    for (int t = 0; t < nk; ++t) {
        const int cur = t & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // tile t landed for every wave; every wave is done reading buffer cur^1
        if (t + 1 < nk) {
            char* nxt = smem + (cur ^ 1) * 2 * TILE_BYTES;
            stage_tile<AT>(nxt, p.A, p.lda, m0, (t + 1) * BK, p.M, p.K, wave, lane);
            stage_tile<BT>(nxt + TILE_BYTES, p.B, p.ldb, n0, (t + 1) * BK, p.N, p.K, wave, lane);
        }
        const MMU_LDS char* tA = (const MMU_LDS char*)smem + cur * 2 * TILE_BYTES;
        const MMU_LDS char* tB = tA + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) af[mt] = read_frag<AT>(tA, wm * 64 + mt * 16, ks, lane);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) bfr[nt] = read_frag<BT>(tB, wn * 64 + nt * 16, ks, lane);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[nt], af[mt], acc[nt][mt], 0, 0, 0);
        }
    }

    // Epilogue: acc[nt][mt][j] = C[m = m0+wm*64+mt*16+(lane&15)][n = n0+wn*64+nt*16+4*(lane>>4)+j]
    // EPI 0: plain bf16 store; 1: bias / activation / residual; 2: accumulate into C (grad accumulation).
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = m0 + wm * 64 + mt * 16 + (lane & 15);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n = n0 + wn * 64 + nt * 16 + 4 * (lane >> 4);
            if (m >= p.M || n >= p.N) continue;  // N % 8 == 0 is enforced by the host wrapper
            const f32x4 v = acc[nt][mt];
            bf16* cp = p.C + (size_t)m * p.ldc + n;
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = v[j];
            if (EPI == 1) {
                if (p.bias) {
                    const bf16x4 b = *reinterpret_cast<const bf16x4*>(p.bias + n);
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] += bf2f(b[j]);
                }
                // HF rounds the Linear output to bf16 before the activation / residual add; keep that order.
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = round_bf16((o[j]));
                if (p.act) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = round_bf16((act_apply(o[j], p.act)));
                }
                if (p.resid) {
                    const bf16x4 r = *reinterpret_cast<const bf16x4*>(p.resid + (size_t)m * p.ldr + n);
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] += bf2f(r[j]);
                }
            } else if (EPI == 2) {
                const bf16x4 c = *reinterpret_cast<const bf16x4*>(cp);
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] += bf2f(c[j]);
            }
            bf16x4 out;
#pragma unroll
            for (int j = 0; j < 4; ++j) out[j] = f2bf(o[j]);
            *reinterpret_cast<bf16x4*>(cp) = out;
        }
    }
}

template <bool AT, bool BT, int EPI>
int launch_gemm_epi(const GemmArgs& a, hipStream_t stream) {
    static bool configured = false;
    auto kern = gemm_bf16_kernel<AT, BT, EPI>;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES);
        if (e != hipSuccess) return (int)e;
        configured = true;
    }
    const int grid = a.tiles_m * a.tiles_n;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 4 * TILE_BYTES, stream, a);
    MMU_CHECK_LAUNCH();
    return 0;
}

template <bool AT, bool BT>
int launch_gemm(const GemmArgs& a, hipStream_t stream) {
    if (a.accumulate) return launch_gemm_epi<AT, BT, 2>(a, stream);
    if (a.bias || a.resid || a.act) return launch_gemm_epi<AT, BT, 1>(a, stream);
    return launch_gemm_epi<AT, BT, 0>(a, stream);
}

}  // namespace

// C-ABI. layout: 0 = NT (A[M,K], B[N,K]); 1 = NN (A[M,K], B[K,N]); 2 = TN (A[K,M], B[K,N]).
// Returns 0, a hipError_t, or -1 for an argument the kernel does not support.
extern "C" int mmu_gemm_bf16(int layout, const void* A, const void* B, void* C, const void* bias, const void* resid,
                             int M, int N, int K, int lda, int ldb, int ldc, int ldr, int act, int accumulate,
                             void* stream) {
    if (M <= 0 || N <= 0 || K <= 0) return -1;
    if ((N & 7) || (lda & 7) || (ldb & 7) || (ldc & 3) || (resid && (ldr & 3))) return -1;
    if (layout == 0 && (K & 7)) return -1;              // both operands K-contiguous: 16-B chunks along K
    if (layout == 1 && (K & 7)) return -1;              // A K-contiguous
    if (layout == 2 && (M & 7)) return -1;              // A stored [K][M]: chunks along M
    if (accumulate && (bias || resid || act)) return -1;
    if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) return -1;
    GemmArgs a;
    a.A = (const bf16*)A; a.B = (const bf16*)B; a.C = (bf16*)C;
    a.bias = (const bf16*)bias; a.resid = (const bf16*)resid;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldr = ldr;
    a.act = act; a.accumulate = accumulate;
    a.tiles_m = (M + BM - 1) / BM; a.tiles_n = (N + BN - 1) / BN;
    hipStream_t s = (hipStream_t)stream;
    switch (layout) {
        case 0: return launch_gemm<false, false>(a, s);
        case 1: return launch_gemm<false, true>(a, s);
        case 2: return launch_gemm<true, true>(a, s);
        default: return -1;
    }
}
