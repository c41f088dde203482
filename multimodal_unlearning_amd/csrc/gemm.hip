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
#include <atomic>

#include "common.h"

struct RopeReq { const int* pos; const bf16* cos; const bf16* sin; int cols; };
static int g_force_tile = 0;  // mmu_gemm_set_tile
static int g_group_m = 0;     // mmu_gemm_set_group_m
static int g_xcd_map = 1;     // mmu_gemm_set_xcd_map: 1 (default) adjacent 32-tile blocks on the 8 XCDs per round, 0 one contiguous chunk per XCD
static int g_persist = 1;     // mmu_gemm_set_persistent: 0 one workgroup per tile; 1 persistent, tiles drawn from a device-side queue (default); 2 persistent, static walk (A/B)
static int* g_tileq = nullptr;   // mmu_gemm_set_tile_queue: TQ_SLOTS x 16 zeroed ints (8 per-XCD-class tile counters + 1 exit counter per slot)
static std::atomic<unsigned> g_tileq_seq{0};  // launches take the slots round-robin (also from several host threads), so kernels in flight on different streams never share one
static int g_num_cu = 256;    // MI355X; replaced by the device's own count on the first launch (cu_count())
static int g_xcd_skew = 0;       // mmu_gemm_set_xcd_skew: start delay per XCD index, in units of s_sleep(32) (~1 us); experiment
static int g_tail_split = 1;     // mmu_gemm_set_tail_split: 0 off, 1 heuristic, 2 whenever the remainder is <= half a round
static float* g_sk_ws = nullptr;  // mmu_gemm_set_workspace
static size_t g_sk_ws_bytes = 0;
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
    int xcd_map;  // 1: adjacent 32-tile blocks of the raster order on the 8 XCDs per round; 0: one contiguous chunk per XCD
    int group_m;  // tile rasterisation: tiles of GROUP_M consecutive tile rows are swept column by column
    // fused low-rank update (LoRA): acc += U[M][lora_k] * BL[N][lora_k]^T after the main K loop; lora_k in {0,32,64}
    const bf16* lora_u;
    const bf16* lora_b;
    int lora_k;
    // fused RoPE (256^2 kernel, plain epilogue only): columns n < rope_cols are rotated per 128-wide head with the
    // rotate-half convention, position rope_pos[m], tables [n_pos][64] bf16 (apply_rotary_pos_emb after the q/k Linear)
    const int* rope_pos;
    const bf16* rope_cos;
    const bf16* rope_sin;
    int rope_cols;
    // split-K (128^2 kernel, EPI 3): blockIdx.y takes K-tiles [y*kt_per_split, ...) and adds its fp32 partial tile into
    // c32 [M][N] with atomics (tiny outputs only: skinny LoRA wgrads)
    float* c32;
    int kt_per_split;
    // split-K tail (256^2 kernel): the last sk_tiles virtual tiles (ids >= sk_first), which would occupy a whole round
    // of the 256 CUs, are each cut into sk_split equal K ranges handled by sk_tiles * sk_split extra workgroups
    // (worker c: tile c / sk_split, range c % sk_split); partial accumulators go to sk_ws [worker][65536] fp32 and
    // sk_fold_kernel reduces them.  Workers of one range index run the same K-tiles at the same time, so the tiles of
    // a row / column still share their panels in L2 (a unit-balanced stream-K split loses that: measured 12 % slower).
    int sk_tiles, sk_first, sk_split;
    float* sk_ws;
    // persistent launch with a tile queue: tq[c] (c = blockIdx.x & 7, the XCD class of the workgroup) counts the tiles
    // of class c handed out so far (virtual tile ids c, c + 8, ...), tq[8] the workgroups that have left; the last one
    // to leave zeroes the slot for the next launch that uses it.  nullptr: static walk vt, vt + grid, ...
    int* tq;
    // fused SwiGLU epilogues (256^2 kernel only; epi_mode 4 / 5, see mmu_gemm_swiglu_fwd / mmu_gemm_swiglu_bwd):
    //  4: C = gate|up [M][2 gu_I] in its natural column order — the B rows of a 256-column tile are taken from TWO places,
    //     gate rows i0 .. i0+127 and up rows gu_I + i0 .. (64 + 64 per 128-column group), so a wave's b = 0 / 1 accumulator
    //     halves hold g and u of the SAME hidden unit —, C2 = act = bf16(silu(g)) * u [M][gu_I], CT = act^T [gu_I][ldct]
    //  5: the GEMM's output dact [M][gu_I] is never stored: with g, u read from `gu` the epilogue writes C = dgu [M][2 gu_I]
    //     and CT = dgu^T [2 gu_I][ldct]
    int xcd_skew;  // 256^2 kernel: workgroups of XCD x (blockIdx.x & 7) start x * xcd_skew * ~1 us late (0 = off)
    int epi_mode, gu_I;
    bf16* C2;
    int ldc2;
    bf16* CT;
    int ldct;
    const bf16* gu;
    int ldgu;
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

// Fragment of a [rows][kl] row-major bf16 matrix straight from global memory (low-rank epilogue operands):
// lane l holds row `row0 + (l&15)`, k = ks*32 + 8*(l>>4) + 0..7; rows past `rows` read as zeros.
__device__ __forceinline__ bf16x8 global_frag(const bf16* __restrict__ base, int kl, int row0, int rows, int ks, int lane) {
    const int r = row0 + (lane & 15);
    bf16x8 z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = f2bf(0.f);
    if (r >= rows) return z;
    return *reinterpret_cast<const bf16x8*>(base + (size_t)r * kl + ks * 32 + 8 * (lane >> 4));
}

__device__ __forceinline__ float act_apply(float v, int act) {
    if (act == 1) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
    if (act == 2) return v / (1.0f + __expf(-1.702f * v));
    return v;
}

// Store 4 consecutive n of row m.  EPI 0: plain bf16 store; 1: bias / activation / residual with the bf16
// rounding points of the HF modules; 2: accumulate into C (gradient accumulation).
template <int EPI>
__device__ __forceinline__ void epi_store(const GemmArgs& p, int m, int n, const f32x4 v) {
    if (m >= p.M || n >= p.N) return;  // N % 8 == 0 is enforced by the host wrapper
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
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = round_bf16(o[j]);
        if (p.act) {
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = round_bf16(act_apply(o[j], p.act));
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

// 8 consecutive columns of one row (16-B store; bias / residual / C read as 16-B loads): the 256^2 kernel's epilogue
// after the cross-lane swap.  Rounding points as in epi_store.
template <int EPI>
__device__ __forceinline__ void epi_store8(const GemmArgs& p, int m, int n, const f32x4 lo, const f32x4 hi) {
    if (m >= p.M || n >= p.N) return;  // N % 8 == 0: the chunk is all in or all out
    bf16* cp = p.C + (size_t)m * p.ldc + n;
    float o[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { o[j] = lo[j]; o[4 + j] = hi[j]; }
    if (EPI == 1) {
        if (p.bias) {
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(p.bias + n);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] += bf2f(b[j]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = round_bf16(o[j]);
        if (p.act) {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = round_bf16(act_apply(o[j], p.act));
        }
        if (p.resid) {
            const bf16x8 r = *reinterpret_cast<const bf16x8*>(p.resid + (size_t)m * p.ldr + n);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] += bf2f(r[j]);
        }
    } else if (EPI == 2) {
        const bf16x8 c = *reinterpret_cast<const bf16x8*>(cp);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] += bf2f(c[j]);
    }
    bf16x8 out;
#pragma unroll
    for (int j = 0; j < 8; ++j) out[j] = f2bf(o[j]);
    *reinterpret_cast<bf16x8*>(cp) = out;  // (nontemporal stores measured 0.4 % slower end to end)
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

    const int nk_all = (p.K + BK - 1) / BK;
    const int t_begin = (EPI == 3) ? blockIdx.y * p.kt_per_split : 0;
    const int nk = (EPI == 3) ? min(nk_all, t_begin + p.kt_per_split) : nk_all;
    if (t_begin >= nk) return;
    stage_tile<AT>(smem, p.A, p.lda, m0, t_begin * BK, p.M, p.K, wave, lane);
    stage_tile<BT>(smem + TILE_BYTES, p.B, p.ldb, n0, t_begin * BK, p.N, p.K, wave, lane);

    for (int t = t_begin; t < nk; ++t) {
        const int cur = (t - t_begin) & 1;
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

    if (p.lora_k) {  // fused low-rank update
        for (int ks = 0; ks < p.lora_k / 32; ++ks) {
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) af[mt] = global_frag(p.lora_u, p.lora_k, m0 + wm * 64 + mt * 16, p.M, ks, lane);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) bfr[nt] = global_frag(p.lora_b, p.lora_k, n0 + wn * 64 + nt * 16, p.N, ks, lane);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[nt], af[mt], acc[nt][mt], 0, 0, 0);
        }
    }

    // Epilogue: acc[nt][mt][j] = C[m = m0+wm*64+mt*16+(lane&15)][n = n0+wn*64+nt*16+4*(lane>>4)+j]
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = m0 + wm * 64 + mt * 16 + (lane & 15);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n = n0 + wn * 64 + nt * 16 + 4 * (lane >> 4);
            if (EPI == 3) {
                if (m < p.M && n < p.N) {
                    float* dst = p.c32 + (size_t)m * p.N + n;
#pragma unroll
                    for (int j = 0; j < 4; ++j) atomicAdd(dst + j, acc[nt][mt][j]);
                }
            } else {
                epi_store<(EPI == 3 ? 0 : EPI)>(p, m, n, acc[nt][mt]);
            }
        }
    }
}

// dst_bf16[i] += src32[i]; src32[i] = 0   (fold a split-K fp32 accumulator into the bf16 gradient)
__global__ void fold_f32_kernel(float* __restrict__ src, bf16* __restrict__ dst, const bf16* __restrict__ mask, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float v = src[i];
        src[i] = 0.f;
        if (mask) v *= bf2f(mask[i]);
        dst[i] = f2bf(bf2f(dst[i]) + v);
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


// =================================================================================================
// 256x256x64 tile, 512 threads = 8 waves (2 M x 4 N), wave output 128x64, 8 phases per 2 K-tiles.
//
// Each wave's 128x64 output is four quadrants (a,b) = (64-row block a) x (32-col block b); one phase computes
// one quadrant for one K-tile (16 MFMA 16x16x32).  The LDS holds, per K-tile buffer d (2 of them), four
// 16-KiB HALF-TILES defined by those blocks, not by contiguous rows:
//     A-h[a]: local row wm*64 + i  <->  tile row wm*128 + a*64 + i      (i < 64, both wm)
//     B-h[b]: local row wn*32 + j  <->  tile col (wn>>1)*128 + b*64 + (wn&1)*32 + j   (j < 32, all four wn)
// so a half-tile is read in exactly ONE phase (A-h0,B-h0: phase 1; B-h1: 2; A-h1: 3; none: 4) and can be
// restaged one phase later.  Every phase issues the LDS-DMA of one future half-tile (2 x 16-B per lane):
//     ph1 A-h1(t1)->d1 | ph2 B-h0(t0+2)->d0 | ph3 A-h0(t0+2)->d0 | ph4 B-h1(t0+2)->d0, vmcnt(6)
//     ph5 A-h1(t0+2)->d0 | ph6 B-h0(t1+2)->d1 | ph7 A-h0(t1+2)->d1 | ph8 B-h1(t1+2)->d1, vmcnt(6)
// The wm=1 waves run ONE BARRIER BEHIND the wm=0 waves (stagger), so each SIMD always has one wave in its MFMA
// cluster while its partner issues LDS reads / LDS-DMA.
// WAR (with the stagger): a half-tile is restaged >= 2 phases after the phase that read it, or 1 phase after
// when its reads were retired by the counted lgkmcnt(8) BEFORE that phase's first barrier (B-h0: read first in
// phases 1/5, restaged in phases 2/6).
// RAW: the counted vmcnt(6) of phase 4 (8) leaves the 3 youngest half-tiles in flight and retires all of
// K-tile t1 (t0+2); it sits before that phase's first barrier and the tile is first read one phase later.
// vmcnt is never 0 inside the loop.  K-tiles past the end (odd tile count, prefetch overrun) read the zero page.
// =================================================================================================
namespace g256 {

typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
constexpr int HB = 16384;  // bytes per half-tile
constexpr int LDS_BYTES = 8 * HB;
constexpr int TQ_SLOTS = 64;  // tile-queue slots of 16 ints (64 B) each

// local row/col of a half-tile -> row/col of the 256-wide tile: 64-element runs stay contiguous so that a K-major
// half-tile is fetched in whole 128-B lines.
__device__ __forceinline__ int a_map(int r) { return ((r >> 6) << 7) + (r & 63); }  // + a*64
__device__ __forceinline__ int b_map(int r) { return ((r >> 6) << 7) + (r & 63); }  // + b*64

// per-thread staging descriptors for one half-tile kind: 2 LDS-DMA instructions
struct StageDesc {
    const bf16* src[2];  // source at k0 = 0 (nullptr: row/col out of range -> zero page)
    int kofs[2];         // element offset along K of this lane's chunk (K-contiguous) or k row (K-major)
};

// gu_I > 0 (B operand of the fused gate|up + SwiGLU forward, K-contiguous weights only): half-tile `sub` = 0 holds GATE rows,
// `sub` = 1 the matching UP rows: local row r <-> weight row sub * gu_I + base0 / 2 + (r >> 6) * 64 + (r & 63).
template <bool T, bool IS_A>
__device__ __forceinline__ StageDesc make_desc(const bf16* G, int ld, int base0, int sub, int dim, int wave, int lane,
                                               int gu_I = 0) {
    StageDesc d;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int t = wave * 2 + i;
        if (!T) {
            const int r = t * 8 + (lane >> 3);
            const int c = (lane & 7) ^ (r & 7);
            const int g = (!IS_A && gu_I > 0) ? sub * gu_I + (base0 >> 1) + ((r >> 6) << 6) + (r & 63)
                                              : base0 + (IS_A ? a_map(r) : b_map(r)) + sub * 64;
            d.kofs[i] = c * 8;
            d.src[i] = (g < dim) ? (G + (size_t)g * ld + c * 8) : nullptr;
        } else {
            const int kr = t * 4 + (lane >> 4);
            const int c = (lane & 15) ^ swzT(kr);
            const int lc = c * 8;
            const int g = base0 + (IS_A ? a_map(lc) : b_map(lc)) + sub * 64;
            d.kofs[i] = kr;
            d.src[i] = (g < dim) ? (G + (size_t)kr * ld + g) : nullptr;
        }
    }
    return d;
}

template <bool T>
__device__ __forceinline__ void stage_half(char* dst, const StageDesc& d, int ld, int k0, int Kd, int wave) {
    const bf16* zero = reinterpret_cast<const bf16*>(mmu_zero16);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const bool ok = d.src[i] != nullptr && (k0 + d.kofs[i] < Kd);
        const bf16* s = ok ? (T ? d.src[i] + (size_t)k0 * ld : d.src[i] + k0) : zero;
        glds16(s, dst + (wave * 2 + i) * 1024);
    }
}

// Virtual tile id -> (m0, n0): bijective XCD-aware remap (virtual ids congruent mod 8 stay on one XCD, also in the
// persistent launch where block b walks ids b, b + grid, ...), then GROUP_M rasterisation.
__device__ __forceinline__ void decode_tile(const GemmArgs& p, int vt, int& m0, int& n0) {
    const int total = p.tiles_m * p.tiles_n;
    int pid = vt;
    if (p.xcd_map == 1) {
        // Round 3: in every round the 8 XCDs take ADJACENT 32-tile blocks of the raster order (4 tile rows x 64 tile
        // columns together), so they stream the same four A panels at the same time — one HBM read and Infinity-Cache
        // hits for the other seven — instead of eight far-apart chunks that share nothing (+0.3..0.8 % step time in
        // same-box A/Bs, profiles/r03_xcd_map_instep_ab.txt).  Ids past the last whole 256-tile super-block (the
        // split-K tail works on exactly those) keep the natural order; the map is a bijection on both parts.
        const int full = total & ~255, xcd = pid & 7, loc = pid >> 3;
        if (pid < full) pid = ((loc >> 5) << 8) + (xcd << 5) + (loc & 31);
    } else {
        const int q = total >> 3, r = total & 7, xcd = pid & 7, loc = pid >> 3;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int per_group = p.group_m * p.tiles_n;
    const int gid = pid / per_group;
    const int first_m = gid * p.group_m;
    const int gsz = min(p.tiles_m - first_m, p.group_m);
    m0 = (first_m + (pid % per_group) % gsz) * 256;
    n0 = ((pid % per_group) / gsz) * 256;
}

// Reduce the split-K partial accumulators of remainder tile r = blockIdx.x: C (+)= sum of its sk_split fp32 partials.  Workspace slot layout = the accumulator layout of the GEMM kernel:
// [fragment (a,mt,b,nt)][wave][lane][4 consecutive columns].
__global__ __launch_bounds__(256) void sk_fold_kernel(GemmArgs p, int epi) {
    const int r = blockIdx.x;
    const int nk = (p.K + 63) / 64;
    const int kc = (nk + p.sk_split - 1) / p.sk_split;
    int m0, n0;
    decode_tile(p, p.sk_first + r, m0, n0);
    // all partials of an element are requested before the first is added (at most 8 ranges: launch_256): the kernel is a
    // latency chain of one round trip, not sk_split of them (1 544 launches per step at the reference's micro-batch)
    auto partial_sum = [&](int idx) {
        f32x4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bool live = i < p.sk_split && i * kc < nk;  // an empty range: that worker wrote nothing
            const int slot = r * p.sk_split + (live ? i : 0);
            v[i] = *reinterpret_cast<const f32x4*>(p.sk_ws + (size_t)slot * 65536 + (size_t)idx * 4);
            if (!live) v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        f32x4 sum = v[0];
#pragma unroll
        for (int i = 1; i < 8; ++i)  // same order of additions as before: ((p0 + p1) + p2) + ...
#pragma unroll
            for (int j = 0; j < 4; ++j) sum[j] += v[i][j];
        return sum;
    };
    for (int idx = blockIdx.y * 256 + threadIdx.x; idx < 16384; idx += gridDim.y * 256) {
        const int frag = idx >> 9, wave = (idx >> 6) & 7, lane = idx & 63;
        const int nt = frag & 1, b = (frag >> 1) & 1, mt = (frag >> 2) & 3, a = frag >> 4;
        const int wm = wave >> 2, wn = wave & 3;
        const int m = m0 + wm * 128 + a * 64 + mt * 16 + (lane & 15);
        const int cih = (wn & 1) * 32 + nt * 16 + 4 * (lane >> 4);  // column inside the 128-wide head (< 64)
        const int n = n0 + (wn >> 1) * 128 + b * 64 + cih;
        if (p.rope_pos && n - b * 64 < p.rope_cols) {
            // fused-RoPE tile: the b = 0 thread of a (c, c + 64) pair folds both halves and rotates them with the
            // arithmetic of the GEMM kernel's own epilogue (bf16-rounded x, cos / sin tables, rounded products)
            if (b) continue;
            const f32x4 s0 = partial_sum(idx), s1 = partial_sum(idx + 1024);  // frag | 2
            f32x4 y0 = s0, y1 = s1;
            if (m < p.M) {
                const int pos = p.rope_pos[m];
                const bf16x4 c4 = *reinterpret_cast<const bf16x4*>(p.rope_cos + (size_t)pos * 64 + cih);
                const bf16x4 s4 = *reinterpret_cast<const bf16x4*>(p.rope_sin + (size_t)pos * 64 + cih);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float x1 = round_bf16(s0[j]), x2 = round_bf16(s1[j]);
                    const float cf = bf2f(c4[j]), sf = bf2f(s4[j]);
                    y0[j] = round_bf16(x1 * cf) + round_bf16(-x2 * sf);
                    y1[j] = round_bf16(x2 * cf) + round_bf16(x1 * sf);
                }
            }
            epi_store<0>(p, m, n, y0);
            epi_store<0>(p, m, n + 64, y1);
            continue;
        }
        const f32x4 sum = partial_sum(idx);
        // the GEMM kernel's own epilogue on the summed tile: 0 plain, 1 bias / activation / residual (HF rounding
        // points), 2 accumulate into C
        if (epi == 1) epi_store<1>(p, m, n, sum);
        else if (epi == 2) epi_store<2>(p, m, n, sum);
        else epi_store<0>(p, m, n, sum);
    }
}

#ifdef MMU_CLOCK_STAMP
__device__ unsigned long long g_clock_stamps[2 * 1024];
#endif

template <bool AT, bool BT, int EPI>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;
#ifdef MMU_CLOCK_STAMP  // diagnostic build only (tools/gemm_clock.py): shader clock held under this load
    const unsigned long long stamp_t0 = __builtin_amdgcn_s_memtime(), stamp_r0 = __builtin_amdgcn_s_memrealtime();
#endif

    // XCD start skew (experiment, mmu_gemm_set_xcd_skew): all workgroups run tiles of the same duration in lockstep, so their
    // epilogue stores hit HBM together and the matrix pipes idle meanwhile.  The 32 CUs of an XCD must stay in K-lockstep
    // (they share panels through their L2), but the 8 XCDs need not: XCD x starts x * skew late.
    if (p.xcd_skew) {
        const int nsl = (int)(blockIdx.x & 7) * p.xcd_skew;
        for (int i = 0; i < nsl; ++i) __builtin_amdgcn_s_sleep(32);
    }
    const int total_tiles = p.tiles_m * p.tiles_n;
    const int nk = (p.K + 63) / 64;
    // Work of this workgroup: whole tiles (vt, then vt + grid in the persistent launch), or — split-K worker — one K
    // range [k0, k1) (in K-tiles) of one remainder tile.
    const bool sk_worker = p.sk_tiles > 0 && (int)blockIdx.x >= p.sk_first;
    // Tile queue (persistent launch): a workgroup that the hardware starts late — its CU was held by another kernel, e.g.
    // RCCL's all-reduce overlapping the last backward sweep — takes fewer tiles instead of keeping a fixed share waiting
    // (static walk, 8 squatting workgroups: 2 446 -> 4 266 us on the qkv projection; profiles/r03_cu_contention_*.json).
    // Thread 0 draws the next tile while the current one computes; the id crosses the waves through 4 bytes of LDS
    // behind the tile buffers.
    int* const tq = p.tq;
    const int tq_class = blockIdx.x & 7;
    volatile int* const tq_lds = reinterpret_cast<volatile int*>(smem + LDS_BYTES);
    int tq_next = 0;
    int vt = blockIdx.x, k0 = 0, k1 = nk;
    if (tq) {
        if (threadIdx.x == 0) *tq_lds = tq_class + 8 * atomicAdd(tq + tq_class, 1);
        __syncthreads();
        vt = __builtin_amdgcn_readfirstlane(*tq_lds);
        if (vt >= total_tiles) {  // more workgroups than tiles of this class: nothing left
            if (threadIdx.x == 0 && atomicAdd(tq + 8, 1) == (int)gridDim.x - 1) {
#pragma unroll
                for (int i = 0; i < 9; ++i) tq[i] = 0;
            }
            return;
        }
        if (threadIdx.x == 0) tq_next = tq_class + 8 * atomicAdd(tq + tq_class, 1);
    }
    if (sk_worker) {
        const int c = blockIdx.x - p.sk_first;
        const int kc = (nk + p.sk_split - 1) / p.sk_split;
        vt = p.sk_first + c / p.sk_split;
        k0 = (c % p.sk_split) * kc;
        k1 = min(nk, k0 + kc);
        if (k0 >= k1) return;
    }

    // LDS map: buffer d in {0,1}: [A-h0 | A-h1 | B-h0 | B-h1], 16 KiB each
    auto buf = [&](int d, int op, int h) -> char* { return smem + ((d * 2 + op) * 2 + h) * HB; };

    int m0, n0;
    decode_tile(p, vt, m0, n0);
#ifdef MMU_EXP_SAMETILE
#define G256_SRC_M(m) 0
#define G256_SRC_N(n) 0
#else
#define G256_SRC_M(m) (m)
#define G256_SRC_N(n) (n)
#endif
    StageDesc dA0 = make_desc<AT, true>(p.A, p.lda, G256_SRC_M(m0), 0, p.M, wave, lane);
    StageDesc dA1 = make_desc<AT, true>(p.A, p.lda, G256_SRC_M(m0), 1, p.M, wave, lane);
    const int gu_rows = (EPI == 4) ? p.gu_I : 0;  // B rows of a tile come from the gate AND the up half of the weight
    StageDesc dB0 = make_desc<BT, false>(p.B, p.ldb, G256_SRC_N(n0), 0, p.N, wave, lane, gu_rows);
    StageDesc dB1 = make_desc<BT, false>(p.B, p.ldb, G256_SRC_N(n0), 1, p.N, wave, lane, gu_rows);

    f32x4 acc[2][2][2][4];  // [a][b][nt][mt]
    bf16x8 fa[2][4];     // [ks][mt]   current A block
    bf16x8 fb[2][2][2];  // [b][ks][nt]

    int kb = k0 * 64, klim = min(p.K, k1 * 64), niter = (k1 - k0 + 1) / 2;  // K window of the current segment

#define G256_BAR()                               \
    do {                                         \
        __builtin_amdgcn_sched_barrier(0);       \
        __builtin_amdgcn_s_barrier();            \
        __builtin_amdgcn_sched_barrier(0);       \
    } while (0)
#define G256_LGKM8()                                   \
    do {                                               \
        __builtin_amdgcn_sched_barrier(0);             \
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); \
    } while (0)
#define G256_STAGE_A(desc, d, h, kt) stage_half<AT>(buf(d, 0, h), desc, p.lda, kb + (kt) * 64, klim, wave)
#define G256_STAGE_B(desc, d, h, kt) stage_half<BT>(buf(d, 1, h), desc, p.ldb, kb + (kt) * 64, klim, wave)
#define G256_READ_A(d, a)                                                                                  \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)    \
        fa[ks][mt] = read_frag<AT>((const MMU_LDS char*)buf(d, 0, a), wm * 64 + mt * 16, ks, lane)
#define G256_READ_B(d, b)                                                                                  \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)    \
        fb[b][ks][nt] = read_frag<BT>((const MMU_LDS char*)buf(d, 1, b), wn * 32 + nt * 16, ks, lane)
#define G256_MMA(a, b)                                                                                      \
    do {                                                                                                    \
        __builtin_amdgcn_s_setprio(1);                                                                      \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)  \
            _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) acc[a][b][nt][mt] =                            \
                __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[b][ks][nt], fa[ks][mt], acc[a][b][nt][mt], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                      \
    } while (0)

    // Ceiling experiments (diagnostic builds only, tools/gemm_ceiling.py; results are WRONG on purpose): what the loop
    // would deliver without its LDS fragment reads (MMU_EXP_NOLDSREAD), without its LDS-DMA traffic (MMU_EXP_NODMA), or
    // with every tile's panels served by L2 (MMU_EXP_SAMETILE: all workgroups stream the panels of tile 0).
#ifdef MMU_EXP_NOLDSREAD
#define G256_LREAD_A(d, a) do { if (it == 0) { G256_READ_A(d, a); } } while (0)
#define G256_LREAD_B(d, b) do { if (it == 0) { G256_READ_B(d, b); } } while (0)
#else
#define G256_LREAD_A(d, a) G256_READ_A(d, a)
#define G256_LREAD_B(d, b) G256_READ_B(d, b)
#endif
#ifdef MMU_EXP_NODMA
#define G256_LSTAGE_A(desc, d, h, kt) do { if (it == 0) { G256_STAGE_A(desc, d, h, kt); } } while (0)
#define G256_LSTAGE_B(desc, d, h, kt) do { if (it == 0) { G256_STAGE_B(desc, d, h, kt); } } while (0)
#else
#define G256_LSTAGE_A(desc, d, h, kt) G256_STAGE_A(desc, d, h, kt)
#define G256_LSTAGE_B(desc, d, h, kt) G256_STAGE_B(desc, d, h, kt)
#endif
    // prologue: K-tile 0 complete in d0; B-h0, A-h0, B-h1 of K-tile 1 in flight towards d1
#define G256_PROLOGUE()              \
    do {                             \
        G256_STAGE_A(dA0, 0, 0, 0);  \
        G256_STAGE_B(dB0, 0, 0, 0);  \
        G256_STAGE_B(dB1, 0, 1, 0);  \
        G256_STAGE_A(dA1, 0, 1, 0);  \
        G256_STAGE_B(dB0, 1, 0, 1);  \
        G256_STAGE_A(dA0, 1, 0, 1);  \
        G256_STAGE_B(dB1, 1, 1, 1);  \
    } while (0)
    G256_PROLOGUE();

  for (;;) {  // persistent launch: this block's tiles vt, vt + grid, ... (one pass when grid == tiles)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[a][b][nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // vmcnt counts in issue order: with the epilogue stores of the previous tile issued after this prologue, this
    // also waits for all but the last 6 of them
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    G256_BAR();
    // Stagger: the wm=1 waves (the second wave of every SIMD) run one barrier behind the wm=0 waves, so on each
    // SIMD one wave is in its MFMA cluster while the other issues its LDS reads / LDS-DMA.
    if (wm == 1) G256_BAR();

    for (int it = 0; it < niter; ++it) {
        const int t0 = 2 * it, t1 = t0 + 1;
        // ---- phase 1: B reads first so lgkmcnt(8) retires them before the barrier (B-h0 is restaged in phase 2)
        G256_LREAD_B(0, 0);
        __builtin_amdgcn_sched_barrier(0);
        G256_LREAD_A(0, 0);
        G256_LSTAGE_A(dA1, 1, 1, t1);
        G256_LGKM8();
        G256_BAR();
        G256_MMA(0, 0);
        G256_BAR();
        // ---- phase 2
        G256_LREAD_B(0, 1);
        G256_LSTAGE_B(dB0, 0, 0, t0 + 2);
        G256_BAR();
        G256_MMA(0, 1);
        G256_BAR();
        // ---- phase 3
        G256_LREAD_A(0, 1);
        G256_LSTAGE_A(dA0, 0, 0, t0 + 2);
        G256_BAR();
        G256_MMA(1, 1);
        G256_BAR();
        // ---- phase 4
        G256_LSTAGE_B(dB1, 0, 1, t0 + 2);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        G256_BAR();
        G256_MMA(1, 0);
        G256_BAR();
        // ---- phase 5
        G256_LREAD_B(1, 0);
        __builtin_amdgcn_sched_barrier(0);
        G256_LREAD_A(1, 0);
        G256_LSTAGE_A(dA1, 0, 1, t0 + 2);
        G256_LGKM8();
        G256_BAR();
        G256_MMA(0, 0);
        G256_BAR();
        // ---- phase 6
        G256_LREAD_B(1, 1);
        G256_LSTAGE_B(dB0, 1, 0, t1 + 2);
        G256_BAR();
        G256_MMA(0, 1);
        G256_BAR();
        // ---- phase 7
        G256_LREAD_A(1, 1);
        G256_LSTAGE_A(dA0, 1, 0, t1 + 2);
        G256_BAR();
        G256_MMA(1, 1);
        G256_BAR();
        // ---- phase 8
        G256_LSTAGE_B(dB1, 1, 1, t1 + 2);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        G256_BAR();
        G256_MMA(1, 0);
        G256_BAR();
    }
    // Tile queue: the draw made when this tile started has long returned.  Thread 0 (a wm = 0 wave) publishes it BEFORE
    // the balance barrier: that barrier pairs with the wm = 1 waves' last in-loop barrier, so every wave passes one
    // barrier between this write and its read below and the hand-over costs no barrier of its own.
    if (tq && threadIdx.x == 0) {
        *tq_lds = tq_next;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (wm == 0) G256_BAR();  // balance the stagger barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the overrun prefetches before the LDS is released
    // the SwiGLU epilogues stage their transposed outputs in the tile buffers: EVERY wave's LDS-DMA must have landed first
    const bool lds_epi = (EPI == 4 || EPI == 5) && p.CT != nullptr;
    if (lds_epi) G256_BAR();

    if (p.lora_k && k0 == 0) {  // fused low-rank update: acc += U[M][kl] * BL[N][kl]^T (once per tile)
        for (int ks = 0; ks < p.lora_k / 32; ++ks) {
            bf16x8 lu[2][4], lb[2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    lu[a][mt] = global_frag(p.lora_u, p.lora_k, m0 + wm * 128 + a * 64 + mt * 16, p.M, ks, lane);
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    lb[b][nt] = global_frag(p.lora_b, p.lora_k, n0 + (wn >> 1) * 128 + b * 64 + (wn & 1) * 32 + nt * 16, p.N,
                                            ks, lane);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt)
                            acc[a][b][nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lb[b][nt], lu[a][mt],
                                                                                        acc[a][b][nt][mt], 0, 0, 0);
        }
    }

    // Next tile (persistent launch, the default: +1.6 % end to end over one workgroup per tile, same-box A/B on two
    // boxes): decode it now, stage its prologue after the epilogue.  (Measured on MI355X: requesting the prologue
    // BEFORE the epilogue stores, or prefetching it in the last iteration of the current tile, gains nothing — vmcnt
    // retires in issue order, so the first counted wait of the next tile also waits for the stores, and the extra
    // live state costs more than the hidden latency; a start skew between CUs costs more than the bursts it spreads.)
    const int cm0 = m0, cn0 = n0;
    const bool partial = sk_worker;  // a K range of a remainder tile: fp32 partial to the workspace
    const int slot = sk_worker ? (int)blockIdx.x - p.sk_first : 0;
    if (tq) {
        vt = __builtin_amdgcn_readfirstlane(*tq_lds);
        if (threadIdx.x == 0 && vt < total_tiles) tq_next = tq_class + 8 * atomicAdd(tq + tq_class, 1);
    } else {
        vt += gridDim.x;
    }
    const bool more = p.sk_tiles == 0 && vt < total_tiles;  // persistent launch only
    if (more) {
        decode_tile(p, vt, m0, n0);
        dA0 = make_desc<AT, true>(p.A, p.lda, G256_SRC_M(m0), 0, p.M, wave, lane);
        dA1 = make_desc<AT, true>(p.A, p.lda, G256_SRC_M(m0), 1, p.M, wave, lane);
        dB0 = make_desc<BT, false>(p.B, p.ldb, G256_SRC_N(n0), 0, p.N, wave, lane, gu_rows);
        dB1 = make_desc<BT, false>(p.B, p.ldb, G256_SRC_N(n0), 1, p.N, wave, lane, gu_rows);
    }

    // Epilogue.  The MFMA leaves each lane with 4 consecutive columns of one row per (nt) tile: stored as is, a wave
    // store covers 16 rows x 32 B and every 128-B line of C is written by four separate partial-line requests (the
    // stores were 8-9 % of a K=4096 GEMM).  v_permlane16_swap exchanges the odd 16-lane rows of the nt=0 tile with the
    // even rows of the nt=1 tile, after which a lane owns 8 consecutive columns (one 16-B store, 64-B runs per row):
    //   lane group g = lane>>4:  columns (g&1)*16 + (g>>1)*8 .. +7 of the wave's 32-column run.
    // b = 0 / 1 are swapped identically, so they stay the two halves (c, c + 64) of a RoPE pair.
    if (partial) {
        float* ws = p.sk_ws + (size_t)slot * 65536;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        const int frag = ((a * 4 + mt) * 2 + b) * 2 + nt;
                        *reinterpret_cast<f32x4*>(ws + ((size_t)(frag * 8 + wave) * 64 + lane) * 4) = acc[a][b][nt][mt];
                    }
    } else if constexpr (EPI == 4 || EPI == 5) {
        // ---- fused SwiGLU epilogues (see GemmArgs::epi_mode) ----
        const int g4 = lane >> 4;
        const int c8 = (g4 & 1) * 16 + (g4 >> 1) * 8;       // first of the lane's 8 columns inside the wave's 32-column run
        const int cih = (wn & 1) * 32 + c8;
        const int I = p.gu_I;
        // transposed outputs: a wave's [128 rows][32 columns] bf16 tile is written column-major into 8448 B of LDS (column
        // stride 264 B: the four 16-lane groups of a ds_write_b16 land 16 banks apart), read back as 16-B row pieces of
        // the TRANSPOSED matrix (two ds_read_b64) and stored as 4 x 256-B runs per instruction.
        MMU_LDS char* xw = (MMU_LDS char*)smem + wave * 8448;
        const int m_w = cm0 + wm * 128;                      // first row of the wave's tile
        auto stage_col8 = [&](int a, int mt, const bf16x8& v) {
            const int r = a * 64 + mt * 16 + (lane & 15);
#pragma unroll
            for (int j = 0; j < 8; ++j) *(MMU_LDS bf16*)(xw + (c8 + j) * 264 + r * 2) = v[j];
        };
        // t0 -> (first column of the wave's run, row m_w) of the transposed output; ncol: columns of the run that exist
        auto flush_t = [&](bf16* t0, int ncol) {
            const int ck = (lane & 15) * 8;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int col = it * 4 + g4;
                const u32x2_t lo = *(const MMU_LDS u32x2_t*)(xw + col * 264 + ck * 2);
                const u32x2_t hi = *(const MMU_LDS u32x2_t*)(xw + col * 264 + ck * 2 + 8);
                if (m_w + ck + 8 <= p.M && col < ncol) {
                    u32x4_t o = {lo[0], lo[1], hi[0], hi[1]};
                    *reinterpret_cast<u32x4_t*>(t0 + (size_t)col * p.ldct + ck) = o;
                }
            }
        };
        if constexpr (EPI == 4) {
            const int gi = (cn0 >> 1) + (wn >> 1) * 64 + cih;  // hidden unit = act column = gate column; up column gi + I
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int m = cm0 + wm * 128 + a * 64 + mt * 16 + (lane & 15);
                    bf16x8 g8, u8, a8;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const auto r0 = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[a][0][0][mt][j]),
                                                                         __float_as_uint(acc[a][0][1][mt][j]), false, false);
                        const auto r1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[a][1][0][mt][j]),
                                                                         __float_as_uint(acc[a][1][1][mt][j]), false, false);
                        g8[j] = f2bf(__uint_as_float(r0[0])); g8[4 + j] = f2bf(__uint_as_float(r0[1]));
                        u8[j] = f2bf(__uint_as_float(r1[0])); u8[4 + j] = f2bf(__uint_as_float(r1[1]));
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) {  // the arithmetic of swiglu_fwd_kernel on the bf16-rounded g, u
                        const float gf = bf2f(g8[j]);
                        a8[j] = f2bf(round_bf16(gf / (1.0f + __expf(-gf))) * bf2f(u8[j]));
                    }
                    if (m < p.M && gi < I) {
                        *reinterpret_cast<bf16x8*>(p.C + (size_t)m * p.ldc + gi) = g8;
                        *reinterpret_cast<bf16x8*>(p.C + (size_t)m * p.ldc + I + gi) = u8;
                        *reinterpret_cast<bf16x8*>(p.C2 + (size_t)m * p.ldc2 + gi) = a8;
                    }
                    if (lds_epi) stage_col8(a, mt, a8);
                }
            if (lds_epi) flush_t(p.CT + (size_t)(gi - c8) * p.ldct + m_w, I - (gi - c8));
        } else {
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int ib = cn0 + (wn >> 1) * 128 + b * 64 + cih;  // hidden unit (column of the never-stored dact)
                bf16x8 du_keep[2][4];
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        const int m = cm0 + wm * 128 + a * 64 + mt * 16 + (lane & 15);
                        float d8[8];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[a][b][0][mt][j]),
                                                                            __float_as_uint(acc[a][b][1][mt][j]), false, false);
                            d8[j] = round_bf16(__uint_as_float(r[0]));       // dact as the stored bf16 tensor would hold it
                            d8[4 + j] = round_bf16(__uint_as_float(r[1]));
                        }
                        bf16x8 dg8, du8;
#pragma unroll
                        for (int j = 0; j < 8; ++j) { dg8[j] = f2bf(0.f); du8[j] = f2bf(0.f); }
                        const bool ok = m < p.M && ib < I;
                        if (ok) {
                            const bf16x8 g8 = *reinterpret_cast<const bf16x8*>(p.gu + (size_t)m * p.ldgu + ib);
                            const bf16x8 u8 = *reinterpret_cast<const bf16x8*>(p.gu + (size_t)m * p.ldgu + I + ib);
#pragma unroll
                            for (int j = 0; j < 8; ++j) {  // the arithmetic of swiglu_bwd_kernel
                                const float gf = bf2f(g8[j]), uf = bf2f(u8[j]), df = d8[j];
                                const float sig = 1.0f / (1.0f + __expf(-gf));
                                du8[j] = f2bf(df * round_bf16(gf * sig));
                                dg8[j] = f2bf(df * uf * (sig * (1.0f + gf * (1.0f - sig))));
                            }
                            *reinterpret_cast<bf16x8*>(p.C + (size_t)m * p.ldc + ib) = dg8;
                            *reinterpret_cast<bf16x8*>(p.C + (size_t)m * p.ldc + I + ib) = du8;
                        }
                        if (lds_epi) stage_col8(a, mt, dg8);
                        du_keep[a][mt] = du8;
                    }
                if (lds_epi) {
                    flush_t(p.CT + (size_t)(ib - c8) * p.ldct + m_w, I - (ib - c8));
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt) stage_col8(a, mt, du_keep[a][mt]);
                    flush_t(p.CT + (size_t)(I + ib - c8) * p.ldct + m_w, I - (ib - c8));
                }
            }
        }
        if (lds_epi && more) G256_BAR();  // the next tile's prologue DMA of OTHER waves lands in this wave's staging area
    } else {
    const int g4 = lane >> 4;
    const int cih = (wn & 1) * 32 + (g4 & 1) * 16 + (g4 >> 1) * 8;  // column inside the 128-wide group (< 64)
    const int n = cn0 + (wn >> 1) * 128 + cih;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int m = cm0 + wm * 128 + a * 64 + mt * 16 + (lane & 15);
            f32x4 y[2][2];  // [b][first / second 4 columns]
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[a][b][0][mt][j]),
                                                                    __float_as_uint(acc[a][b][1][mt][j]), false, false);
                    y[b][0][j] = __uint_as_float(r[0]);
                    y[b][1][j] = __uint_as_float(r[1]);
                }
            if (EPI == 0 && p.rope_pos && n < p.rope_cols && m < p.M) {
                const int pos = p.rope_pos[m];
                const bf16x8 c8 = *reinterpret_cast<const bf16x8*>(p.rope_cos + (size_t)pos * 64 + cih);
                const bf16x8 s8 = *reinterpret_cast<const bf16x8*>(p.rope_sin + (size_t)pos * 64 + cih);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float x1 = round_bf16(y[0][h][j]), x2 = round_bf16(y[1][h][j]);
                        const float cf = bf2f(c8[h * 4 + j]), sf = bf2f(s8[h * 4 + j]);
                        y[0][h][j] = round_bf16(x1 * cf) + round_bf16(-x2 * sf);
                        y[1][h][j] = round_bf16(x2 * cf) + round_bf16(x1 * sf);
                    }
            }
            epi_store8<EPI>(p, m, n, y[0][0], y[0][1]);
            epi_store8<EPI>(p, m, n + 64, y[1][0], y[1][1]);
        }
    }
    if (!more) break;
    G256_PROLOGUE();
  }
    if (tq && threadIdx.x == 0 && atomicAdd(tq + 8, 1) == (int)gridDim.x - 1) {  // last workgroup out: every draw has been made
#pragma unroll
        for (int i = 0; i < 9; ++i) tq[i] = 0;
    }
#ifdef MMU_CLOCK_STAMP
    // the stamps leave through a buffer of their own that nothing else reads; no output depends on them
    if (threadIdx.x == 0 && blockIdx.x < 1024) {
        g_clock_stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - stamp_t0;          // shader cycles
        g_clock_stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - stamp_r0;  // 100 MHz ticks
    }
#endif
#undef G256_PROLOGUE
#undef G256_LREAD_A
#undef G256_LREAD_B
#undef G256_LSTAGE_A
#undef G256_LSTAGE_B
#undef G256_SRC_M
#undef G256_SRC_N
#undef G256_BAR
#undef G256_LGKM8
#undef G256_STAGE_A
#undef G256_STAGE_B
#undef G256_READ_A
#undef G256_READ_B
#undef G256_MMA
}

// Compute units of the current device, read once (the tail split and the persistent grid are sized by it; the
// split-K workspace contract stays 256 slots = mmu_gemm_workspace_bytes()).
static int cu_count() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0 && v <= 256)
            n = v;
        else
            n = 256;
        g_num_cu = n;
    }
    return n;
}

template <bool AT, bool BT, int EPI>
int launch_epi(GemmArgs a, hipStream_t stream) {
    static bool configured = false;
    cu_count();
    auto kern = gemm256_kernel<AT, BT, EPI>;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES + 16);
        if (e != hipSuccess) return (int)e;
        configured = true;
    }
    a.tiles_m = (a.M + 255) / 256;
    a.tiles_n = (a.N + 255) / 256;
    a.group_m = g_group_m > 0 ? g_group_m : 4;
    a.xcd_map = g_xcd_map;
    a.xcd_skew = g_xcd_skew;
    const int tiles = a.tiles_m * a.tiles_n;
    // Split-K tail: with T tiles on 256 CUs the last T % 256 tiles occupy a whole round; when they are at most half a
    // round, cut each into S = 256 / rem K ranges over rem * S extra workgroups (gate_up wgrad: 96 tiles x 2 ranges
    // = 5.5 rounds instead of 6).  The fold kernel applies the tile's epilogue (plain, bias / activation / residual
    // with the HF rounding points, accumulate, or the fused RoPE rotation of the q|k columns) to the summed partials.
    // With the reference's default micro-batch (2 pairs = 4352 tokens = 17 tile rows) the o / down projections are
    // 272 tiles = 1.06 rounds: their 16 remainder tiles become 128 K-range workers instead of a second round.
    const int rem = tiles % g_num_cu;
    const int nk = (a.K + 63) / 64;
    int split = rem ? g_num_cu / rem : 0;
    if (split > 8) split = 8;  // (16 ranges of 4 K-tiles measured 1.5 % slower at the reference's micro-batch)
    if (split > nk) split = nk;
    bool sk = g_tail_split && split >= 2 && !(g_tail_split == 3 && (EPI == 1 || a.rope_pos)) && g_sk_ws &&
              g_sk_ws_bytes >= (size_t)256 * 65536 * sizeof(float) && EPI < 4;  // (the fold kernel has no SwiGLU epilogue)
    if (sk && g_tail_split != 2) {
        const double saving_us = (1.0 - 1.0 / split) * nk * 1.45;  // measured us per K-tile of one tile
        const double cost_us = 15.0 + 0.1 * rem * split;           // partial write + fold read at ~5 TB/s
        sk = saving_us > 1.5 * cost_us;
    }
    if (sk) {
        a.sk_tiles = rem;
        a.sk_first = tiles - rem;
        a.sk_split = split;
        a.sk_ws = g_sk_ws;
        hipLaunchKernelGGL(kern, dim3(a.sk_first + rem * split), dim3(512), LDS_BYTES + 16, stream, a);
        MMU_CHECK_LAUNCH();
        hipLaunchKernelGGL(sk_fold_kernel, dim3(rem, 64), dim3(256), 0, stream, a, EPI);  // one element group per thread
        MMU_CHECK_LAUNCH();
        return 0;
    }
    // persistent: mode 1 draws tiles from a queue slot (needs mmu_gemm_set_tile_queue; without one it is one workgroup
    // per tile, which shares the GPU gracefully), mode 2 is the static walk
    const bool persist = tiles > g_num_cu && (g_persist == 2 || (g_persist == 1 && g_tileq));
    if (persist && g_persist == 1) a.tq = g_tileq + (size_t)(g_tileq_seq.fetch_add(1, std::memory_order_relaxed) % TQ_SLOTS) * 16;
    const int grid = persist ? g_num_cu : tiles;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS_BYTES + 16, stream, a);
    MMU_CHECK_LAUNCH();
    return 0;
}

template <bool AT, bool BT>
int launch(const GemmArgs& a, hipStream_t stream) {
    if constexpr (!AT) {
        if constexpr (!BT) {
            if (a.epi_mode == 4) return launch_epi<false, false, 4>(a, stream);
        }
        if (a.epi_mode == 5) return launch_epi<false, BT, 5>(a, stream);
    }
    if (a.epi_mode) return -1;
    if (a.accumulate) return launch_epi<AT, BT, 2>(a, stream);
    if (a.bias || a.resid || a.act) return launch_epi<AT, BT, 1>(a, stream);
    return launch_epi<AT, BT, 0>(a, stream);
}

}  // namespace g256

}  // namespace


// 256^2 tiles (8-phase kernel) when the grid fills the 256 CUs reasonably; the 128^2 kernel (2 blocks/CU)
// otherwise.  Wave quantisation: cost ~ rounds * work per CU per round / rate; a 256^2 tile is 4 units at a
// measured ~1.45x per-CU rate, the 128^2 kernel runs 2 one-unit tiles per CU per round.
static bool pick_256(int M, int N, int K) {
    if (g_force_tile == 256) return true;
    if (g_force_tile == 128) return false;
    if (K < 256) return false;
    const long t256 = (long)((M + 255) / 256) * ((N + 255) / 256);
    const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    const double c256 = (double)((t256 + 255) / 256) * (4.0 / 1.45);
    const double c128 = (double)((t128 + 511) / 512) * 2.0;
    return c256 <= c128;
}

// C-ABI. layout: 0 = NT (A[M,K], B[N,K]); 1 = NN (A[M,K], B[K,N]); 2 = TN (A[K,M], B[K,N]).
// Returns 0, a hipError_t, or -1 for an argument the kernel does not support.
extern "C" int mmu_gemm_bf16_lora(int layout, const void* A, const void* B, void* C, const void* bias,
                                  const void* resid, int M, int N, int K, int lda, int ldb, int ldc, int ldr, int act,
                                  int accumulate, const void* lora_u, const void* lora_b, int lora_k, void* stream);
extern "C" int mmu_gemm_bf16_rope(int layout, const void* A, const void* B, void* C, int M, int N, int K, int lda,
                                  int ldb, int ldc, const void* lora_u, const void* lora_b, int lora_k,
                                  const int* rope_pos, const void* rope_cos, const void* rope_sin, int rope_cols,
                                  void* stream);

extern "C" int mmu_gemm_bf16(int layout, const void* A, const void* B, void* C, const void* bias, const void* resid,
                             int M, int N, int K, int lda, int ldb, int ldc, int ldr, int act, int accumulate,
                             void* stream) {
    return mmu_gemm_bf16_lora(layout, A, B, C, bias, resid, M, N, K, lda, ldb, ldc, ldr, act, accumulate, nullptr,
                              nullptr, 0, stream);
}

// Same as mmu_gemm_bf16 plus a fused low-rank update acc += U[M][lora_k] * BL[N][lora_k]^T (both row-major,
// contiguous, 16-B aligned; lora_k in {32, 64}; pad unused columns with zeros).  LoRA: U = s * x A^T, BL = B.
static int gemm_dispatch(int layout, const void* A, const void* B, void* C, const void* bias, const void* resid, int M,
                         int N, int K, int lda, int ldb, int ldc, int ldr, int act, int accumulate, const void* lora_u,
                         const void* lora_b, int lora_k, const RopeReq& rope, void* stream);
extern "C" int mmu_gemm_bf16_lora(int layout, const void* A, const void* B, void* C, const void* bias,
                                  const void* resid, int M, int N, int K, int lda, int ldb, int ldc, int ldr, int act,
                                  int accumulate, const void* lora_u, const void* lora_b, int lora_k, void* stream) {
    return gemm_dispatch(layout, A, B, C, bias, resid, M, N, K, lda, ldb, ldc, ldr, act, accumulate, lora_u, lora_b, lora_k,
                         RopeReq{nullptr, nullptr, nullptr, 0}, stream);
}
static int gemm_dispatch(int layout, const void* A, const void* B, void* C, const void* bias, const void* resid, int M,
                         int N, int K, int lda, int ldb, int ldc, int ldr, int act, int accumulate, const void* lora_u,
                         const void* lora_b, int lora_k, const RopeReq& rope, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0) return -1;
    if (lora_k != 0 && (lora_k != 32 && lora_k != 64)) return -1;
    if (lora_k != 0 && (!lora_u || !lora_b || (((uintptr_t)lora_u | (uintptr_t)lora_b) & 15))) return -1;
    if ((N & 7) || (lda & 7) || (ldb & 7) || (ldc & 3) || (resid && (ldr & 3))) return -1;
    if (layout == 0 && (K & 7)) return -1;              // both operands K-contiguous: 16-B chunks along K
    if (layout == 1 && (K & 7)) return -1;              // A K-contiguous
    if (layout == 2 && (M & 7)) return -1;              // A stored [K][M]: chunks along M
    if (accumulate && (bias || resid || act)) return -1;
    if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) return -1;
    GemmArgs a = {};
    a.A = (const bf16*)A; a.B = (const bf16*)B; a.C = (bf16*)C;
    a.bias = (const bf16*)bias; a.resid = (const bf16*)resid;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldr = ldr;
    a.act = act; a.accumulate = accumulate;
    a.lora_u = (const bf16*)lora_u; a.lora_b = (const bf16*)lora_b; a.lora_k = lora_k;
    a.rope_pos = rope.pos; a.rope_cos = rope.cos; a.rope_sin = rope.sin; a.rope_cols = rope.cols;
    // the 256^2 kernel's epilogue moves 16-B chunks of C / residual / bias
    const bool use256 = pick_256(M, N, K) && !(ldc & 7) && !(resid && ((ldr & 7) || ((uintptr_t)resid & 15))) &&
                        !(bias && ((uintptr_t)bias & 15));
    if (a.rope_pos && !use256) return -2;  // fused RoPE exists in the 256^2 kernel only
    a.tiles_m = (M + BM - 1) / BM; a.tiles_n = (N + BN - 1) / BN;
    hipStream_t s = (hipStream_t)stream;
    if (use256) {
        switch (layout) {
            case 0: return g256::launch<false, false>(a, s);
            case 1: return g256::launch<false, true>(a, s);
            case 2: return g256::launch<true, true>(a, s);
            default: return -1;
        }
    }
    switch (layout) {
        case 0: return launch_gemm<false, false>(a, s);
        case 1: return launch_gemm<false, true>(a, s);
        case 2: return launch_gemm<true, true>(a, s);
        default: return -1;
    }
}

// Tile-config override for A/B tests: 0 = heuristic, 128 / 256 = force that kernel.
extern "C" int mmu_gemm_set_tile(int tile) {
    if (tile != 0 && tile != 128 && tile != 256) return -1;
    g_force_tile = tile;
    return 0;
}

// Rasterisation override for the 256^2 kernel (tuning): 0 = default.
extern "C" int mmu_gemm_set_group_m(int group_m) {
    if (group_m < 0 || group_m > 64) return -1;
    g_group_m = group_m;
    return 0;
}

// 1 if mmu_gemm_bf16 would run the 256x256 kernel for this shape (the only one with the fused RoPE epilogue).
extern "C" int mmu_gemm_uses_256(int M, int N, int K) { return pick_256(M, N, K) ? 1 : 0; }

// y = x W^T (+ LoRA) with RoPE applied to columns [0, rope_cols) in the epilogue (q|k part of the fused qkv
// projection, head_dim 128, N % 256 alignment of heads).  Returns -2 when the shape would not run on the 256^2
// kernel: the caller then issues mmu_gemm_bf16 + mmu_rope.
extern "C" int mmu_gemm_bf16_rope(int layout, const void* A, const void* B, void* C, int M, int N, int K, int lda,
                                  int ldb, int ldc, const void* lora_u, const void* lora_b, int lora_k,
                                  const int* rope_pos, const void* rope_cos, const void* rope_sin, int rope_cols,
                                  void* stream) {
    if (!rope_pos || !rope_cos || !rope_sin || (rope_cols & 127) || rope_cols > N) return -1;
    // the request travels in the kernel arguments (no process-global state: safe from any host thread / stream)
    const RopeReq rope = {rope_pos, (const bf16*)rope_cos, (const bf16*)rope_sin, rope_cols};
    return gemm_dispatch(layout, A, B, C, nullptr, nullptr, M, N, K, lda, ldb, ldc, 0, 0, 0, lora_u, lora_b, lora_k, rope,
                         stream);
}

// Split-K GEMM for SMALL outputs (skinny LoRA wgrads): grid.y = splits blocks per tile each reduce a K range and
// atomically add their fp32 partial into c32 [M][N] (must be zero on entry; left holding the result).
// mmu_fold_f32_into_bf16 then adds it into the bf16 gradient (optionally through a 0/1 mask) and clears it.
extern "C" int mmu_gemm_bf16_splitk(int layout, const void* A, const void* B, float* c32, int M, int N, int K, int lda,
                                    int ldb, int splits, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || splits < 1 || splits > 1024 || !c32) return -1;
    if ((N & 7) || (lda & 7) || (ldb & 7)) return -1;
    if (layout != 2 && (K & 7)) return -1;
    if (layout == 2 && (M & 7)) return -1;
    if (((uintptr_t)A | (uintptr_t)B) & 15) return -1;
    GemmArgs a = {};
    a.A = (const bf16*)A; a.B = (const bf16*)B; a.c32 = c32;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb;
    a.tiles_m = (M + BM - 1) / BM; a.tiles_n = (N + BN - 1) / BN;
    const int nk = (K + BK - 1) / BK;
    a.kt_per_split = (nk + splits - 1) / splits;
    const int ny = (nk + a.kt_per_split - 1) / a.kt_per_split;
    hipStream_t s = (hipStream_t)stream;
    auto go = [&](auto kern, bool& configured) -> int {
        if (!configured) {  // once per kernel instantiation, not per launch
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES);
            if (e != hipSuccess) return (int)e;
            configured = true;
        }
        hipLaunchKernelGGL(kern, dim3(a.tiles_m * a.tiles_n, ny), dim3(256), 4 * TILE_BYTES, s, a);
        hipError_t e2 = hipGetLastError();
        return (int)e2;
    };
    static bool cfg[3] = {false, false, false};
    switch (layout) {
        case 0: return go(gemm_bf16_kernel<false, false, 3>, cfg[0]);
        case 1: return go(gemm_bf16_kernel<false, true, 3>, cfg[1]);
        case 2: return go(gemm_bf16_kernel<true, true, 3>, cfg[2]);
        default: return -1;
    }
}

extern "C" int mmu_fold_f32_into_bf16(float* src32, void* dst, const void* mask, long long n, void* stream) {
    if (n <= 0) return -1;
    size_t grid = ((size_t)n + 255) / 256;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(fold_f32_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, src32, (bf16*)dst,
                       (const bf16*)mask, (size_t)n);
    MMU_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmu_gemm_set_xcd_map(int mode) {
    if (mode < 0 || mode > 1) return -1;
    g_xcd_map = mode;
    return 0;
}
extern "C" int mmu_gemm_set_xcd_skew(int units) {
    if (units < 0 || units > 64) return -1;
    g_xcd_skew = units;
    return 0;
}
extern "C" int mmu_gemm_set_persistent(int mode) {
    if (mode < 0 || mode > 2) return -1;
    g_persist = mode;
    return 0;
}
// Device memory for the tile queues of the persistent launch: `bytes` >= mmu_gemm_tile_queue_bytes(), 16-B aligned; nullptr
// unregisters (launches with more tiles than CUs then run one workgroup per tile).  The library never allocates.  The
// counters must be zero whenever no launch is in flight: this call zeroes them (memset on the NULL stream + a device
// synchronise: registration is a set-up step, not a hot path), the kernels leave them zero again, and
// mmu_gemm_reset_tile_queue() restores that state after a launch died between its draws and its exit (a counter left
// non-zero would make later launches skip tiles: first draws past the tile count return without computing).
extern "C" int mmu_gemm_set_tile_queue(void* counters, long long bytes) {
    if (counters && (bytes < (long long)(g256::TQ_SLOTS * 16 * sizeof(int)) || ((uintptr_t)counters & 15))) return -1;
    if (counters) {
        hipError_t e = hipMemset(counters, 0, g256::TQ_SLOTS * 16 * sizeof(int));
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) return (int)e;
    }
    g_tileq = (int*)counters;
    return 0;
}
extern "C" int mmu_gemm_reset_tile_queue(void) {
    if (!g_tileq) return 0;
    hipError_t e = hipDeviceSynchronize();  // nothing may be drawing while the counters are cleared
    if (e == hipSuccess) e = hipMemset(g_tileq, 0, g256::TQ_SLOTS * 16 * sizeof(int));
    if (e == hipSuccess) e = hipDeviceSynchronize();
    return (int)e;
}
extern "C" long long mmu_gemm_tile_queue_bytes(void) { return (long long)(g256::TQ_SLOTS * 16 * sizeof(int)); }

// Split-K tail of the 256^2 kernel.  mode: 0 off, 1 when the model of saved rounds vs fold traffic says it pays
// (default), 2 whenever the remainder tiles fill at most half a round (tests).  It needs a device workspace of at
// least 256 * 256 KiB = 64 MiB registered with mmu_gemm_set_workspace (the library never allocates).
extern "C" int mmu_gemm_set_tail_split(int mode) {
    if (mode < 0 || mode > 3) return -1;  // 3: heuristic without the bias / residual epilogue (round-1 behaviour, A/B)
    g_tail_split = mode;
    return 0;
}
extern "C" int mmu_gemm_set_workspace(void* ws, long long bytes) {
    if (bytes < 0 || ((uintptr_t)ws & 15)) return -1;
    g_sk_ws = (float*)ws;
    g_sk_ws_bytes = ws ? (size_t)bytes : 0;
    return 0;
}
extern "C" long long mmu_gemm_workspace_bytes(void) { return 256LL * 65536 * sizeof(float); }

#ifdef MMU_CLOCK_STAMP
// Diagnostic build only: copy the per-workgroup {shader cycles, 100 MHz ticks} of the LAST 256^2 launch to the host.
extern "C" int mmu_debug_read_clock_stamps(unsigned long long* host_out, int n_pairs) {
    if (n_pairs < 1 || n_pairs > 1024) return -1;
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g256::g_clock_stamps), (size_t)n_pairs * 16, 0, hipMemcpyDeviceToHost);
}
#endif

// ---- fused SwiGLU GEMMs (256^2 kernel; return -2 when the shape would not run on it: the caller then issues the plain GEMM
// and the elementwise kernel).  LlamaMLP: down(act_fn(gate(x)) * up(x)), modeling_llama.py:174-176. ----
//
// gate|up projection + SwiGLU in one launch: gu [M][2 I] = x W^T (W = [gate rows | up rows] [2 I][K], natural order, as
// every other consumer of gu expects it), act [M][I] = bf16(silu(g)) * u, and — act_t != NULL — act^T [I][ldt] (the
// down-projection's weight-gradient operand; needs M % 8 == 0).  Replaces mmu_gemm_bf16 + mmu_swiglu_fwd[_both] with
// bit-identical outputs (the epilogue rounds g, u to bf16 before the activation exactly as the two-kernel path reads them).
extern "C" int mmu_gemm_swiglu_fwd(const void* x, const void* w_gu, void* gu, void* act, void* act_t, int M, int I, int K,
                                   int ldx, int ldw, int ldgu, int ldact, int ldt, void* stream) {
    if (M <= 0 || I <= 0 || K <= 0 || !x || !w_gu || !gu || !act) return -1;
    if ((I & 127) || (K & 7) || (ldx & 7) || (ldw & 7) || (ldgu & 7) || (ldact & 7)) return -1;
    if (((uintptr_t)x | (uintptr_t)w_gu | (uintptr_t)gu | (uintptr_t)act | (uintptr_t)act_t) & 15) return -1;
    if (act_t && ((M & 7) || (ldt & 7) || ldt < M)) return -1;
    if (!pick_256(M, 2 * I, K)) return -2;
    GemmArgs a = {};
    a.A = (const bf16*)x; a.B = (const bf16*)w_gu; a.C = (bf16*)gu;
    a.M = M; a.N = 2 * I; a.K = K; a.lda = ldx; a.ldb = ldw; a.ldc = ldgu;
    a.epi_mode = 4; a.gu_I = I; a.C2 = (bf16*)act; a.ldc2 = ldact; a.CT = (bf16*)act_t; a.ldct = ldt;
    return g256::launch<false, false>(a, (hipStream_t)stream);
}

// down-projection dgrad + SwiGLU backward in one launch: dact = dh W_down (w_layout 0: w = W_down^T [I][K = d], the resident
// transposed copy; 1: w = W_down [K = d][I]) stays in registers; with g, u from gu [M][2 I] the epilogue writes dgu [M][2 I]
// and — dgu_t != NULL — dgu^T [2 I][ldt] (the gate|up weight-gradient operand; M % 8 == 0).  Replaces mmu_gemm_bf16 +
// mmu_swiglu_bwd[_t] with bit-identical outputs (dact is rounded to bf16 before use, as the stored tensor was).
extern "C" int mmu_gemm_swiglu_bwd(const void* dh, const void* w, int w_layout, const void* gu, void* dgu, void* dgu_t, int M,
                                   int I, int K, int lddh, int ldw, int ldgu, int lddgu, int ldt, void* stream) {
    if (M <= 0 || I <= 0 || K <= 0 || !dh || !w || !gu || !dgu || (w_layout != 0 && w_layout != 1)) return -1;
    if ((I & 7) || (K & 7) || (lddh & 7) || (ldw & 7) || (ldgu & 7) || (lddgu & 7)) return -1;
    if (((uintptr_t)dh | (uintptr_t)w | (uintptr_t)gu | (uintptr_t)dgu | (uintptr_t)dgu_t) & 15) return -1;
    if (dgu_t && ((M & 7) || (ldt & 7) || ldt < M)) return -1;
    if (!pick_256(M, I, K)) return -2;
    GemmArgs a = {};
    a.A = (const bf16*)dh; a.B = (const bf16*)w; a.C = (bf16*)dgu;
    a.M = M; a.N = I; a.K = K; a.lda = lddh; a.ldb = ldw; a.ldc = lddgu;
    a.epi_mode = 5; a.gu_I = I; a.CT = (bf16*)dgu_t; a.ldct = ldt; a.gu = (const bf16*)gu; a.ldgu = ldgu;
    return w_layout == 0 ? g256::launch<false, false>(a, (hipStream_t)stream) : g256::launch<false, true>(a, (hipStream_t)stream);
}
