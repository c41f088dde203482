// Shared device helpers of the attention kernels (attention.hip: forward, dQ, eight-wave dK/dV; attention_dkdv4.hip: the
// four-wave register-resident dK/dV kernel, compiled with its own flags).  Everything is internal to libmmu_hip.so.
#pragma once
#include <type_traits>

#include "common.h"

namespace {


#ifdef MMU_ATTN_STAMP  // diagnostic build only (tools/attn_stamps.py): where a dK/dV workgroup spends its cycles
__device__ unsigned long long g_attn_stamps[4 * 16384];
__device__ __forceinline__ unsigned long long stamp_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
#define STAMP(var) const unsigned long long var = stamp_now()
// forward kernel: per workgroup (its last wave: active on every block of a causal tile) 8 words
// [blocks, loop-top wait, stage issue, QK^T until the row max, softmax + O rescale, P V issue, total, masked blocks]
__device__ unsigned long long g_fwd_stamps[8 * 16384];
#define FSTAMP(var)                           \
    __builtin_amdgcn_sched_barrier(0);        \
    const unsigned long long var = stamp_now(); \
    __builtin_amdgcn_sched_barrier(0)
#else
#define FSTAMP(var)
#define STAMP(var)
#endif

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

enum { KIND_ROW = 0, KIND_TR = 1 };

template <int D, int KIND>
__device__ __forceinline__ int swz(int r) {
    if (D == 128) return ((r & 3) << 2) | ((r >> 2) & 3);
    // D == 64: 128-B rows (two rows per 256-B bank row)
    return KIND == KIND_ROW ? ((r >> 1) & 7) : (((r >> 1) & 1) << 2);
}

// Stage R rows x D bf16 of a row-major matrix (row stride ld elements) into a swizzled LDS tile.
// Rows >= row_limit read the zero page.
template <int D, int KIND, int R, int NW = 4>
__device__ __forceinline__ void stage_rows(char* tile, const bf16* __restrict__ g, int ld, int row0, int row_limit,
                                           int wave, int lane) {
    constexpr int ROWB = D * 2, RPI = 1024 / ROWB, CPR = ROWB / 16, NI = R / RPI, PW = NI / NW;
    static_assert(PW >= 1 && PW * NW == NI, "tile does not split evenly over the waves");
    const bf16* zero = reinterpret_cast<const bf16*>(mmu_zero16);
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int t = wave * PW + i;
        const int r = t * RPI + lane / CPR;
        const int c = (lane % CPR) ^ swz<D, KIND>(r);
        const bf16* src = (row0 + r < row_limit) ? (g + (size_t)(row0 + r) * ld + c * 8) : zero;
        glds16(src, tile + t * 1024);
    }
}

// The same staging with the per-lane part of the address (row * ld + swizzled chunk) computed ONCE: per tile only a
// wave-uniform base pointer moves, and full tiles skip the per-lane bounds select (the per-block address arithmetic of
// stage_rows was ~150 issue slots in front of every block's MFMAs).
template <int D, int KIND, int R, int NW = 4>
struct RowStager {
    static constexpr int ROWB = D * 2, RPI = 1024 / ROWB, CPR = ROWB / 16, NI = R / RPI, PW = NI / NW;
    int off[PW];  // elements from the tile's first row
    int row[PW];
    __device__ __forceinline__ void init(int ld, int wave, int lane) {
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int t = wave * PW + i;
            const int r = t * RPI + lane / CPR;
            const int c = (lane % CPR) ^ swz<D, KIND>(r);
            row[i] = r;
            off[i] = r * ld + c * 8;
        }
    }
    __device__ __forceinline__ void issue(char* tile, const bf16* __restrict__ g, int ld, int row0, int row_limit,
                                          int wave) const {
        const bf16* base = g + (size_t)row0 * ld;
        if (row0 + R <= row_limit) {
#pragma unroll
            for (int i = 0; i < PW; ++i) glds16(base + off[i], tile + (wave * PW + i) * 1024);
        } else {
            const bf16* zero = reinterpret_cast<const bf16*>(mmu_zero16);
#pragma unroll
            for (int i = 0; i < PW; ++i)
                glds16((row0 + row[i] < row_limit) ? base + off[i] : zero, tile + (wave * PW + i) * 1024);
        }
    }
};

// Row-read fragment: lane l gets tile[base + (l&31)][16*ks + 8*(l>>5) + 0..7]
template <int D, int KIND>
__device__ __forceinline__ bf16x8 row_frag(const MMU_LDS char* tile, int base, int ks, int lane) {
    const int r = base + (lane & 31);
    const int ch = 2 * ks + (lane >> 5);
    return *(const MMU_LDS bf16x8*)(tile + r * (D * 2) + ((ch ^ swz<D, KIND>(r)) << 4));
}

// Transposed fragment: lane l gets tile[rbase + 16*s + 8*(j>>2) + 4*(l>>5) + (j&3)][c0 + (l&31)], j = 0..7
// (the k order of an accumulator tile used as the other operand, guide §3).
template <int D, int KIND>
__device__ __forceinline__ bf16x8 tr_frag(const MMU_LDS char* tile, int rbase, int s, int c0, int lane) {
    const int g = lane >> 4, gb = g & 1, h = g >> 1, i = lane & 15, q = i >> 2, p = i & 3;
    const int r1 = rbase + 16 * s + 4 * h + q;
    const int r2 = r1 + 8;
    const int ch = (c0 >> 3) + 2 * gb + (p >> 1);
    const int sub = (p & 1) << 3;
    const MMU_LDS char* a1 = tile + r1 * (D * 2) + ((ch ^ swz<D, KIND>(r1)) << 4) + sub;
    const MMU_LDS char* a2 = tile + r2 * (D * 2) + ((ch ^ swz<D, KIND>(r2)) << 4) + sub;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((MMU_LDS bf16x4*)a1);
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((MMU_LDS bf16x4*)a2);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}


// Per-lane LDS byte offsets of the 32x32x16 fragments, relative to a tile address whose fragment starts at a row
// that is a multiple of 32 (row / transposed reads then only add compile-time constants).
template <int D, int KIND>
struct FragOffsets {
    int row[D / 16];               // row fragment of k-step ks: row (lane&31), 16-B chunk 2ks + (lane>>5)
    int tr1[D / 32], tr2[D / 32];  // transposed fragment, 32-column block dt, k-step 0 (k-step 1: + 16 rows)
    __device__ __forceinline__ void init(int lane) {
        const int hl = lane >> 5, r = lane & 31;
#pragma unroll
        for (int ks = 0; ks < D / 16; ++ks) row[ks] = r * (D * 2) + (((2 * ks + hl) ^ swz<D, KIND>(r)) << 4);
        const int g = lane >> 4, gb = g & 1, h = g >> 1, i = lane & 15, q = i >> 2, pp = i & 3;
        const int r1 = 4 * h + q, r2 = r1 + 8, sub = (pp & 1) << 3;
#pragma unroll
        for (int dt = 0; dt < D / 32; ++dt) {
            const int ch = 4 * dt + 2 * gb + (pp >> 1);
            tr1[dt] = r1 * (D * 2) + ((ch ^ swz<D, KIND>(r1)) << 4) + sub;
            tr2[dt] = r2 * (D * 2) + ((ch ^ swz<D, KIND>(r2)) << 4) + sub;
        }
    }
    // Double-buffered tiles sit 2^k bytes apart: the offsets follow the buffer by an in-place XOR (D/16 + D/16 v_xor per
    // block) instead of one v_add of the tile base in front of every LDS read (72 per block in the backward kernels:
    // a quarter of their VALU instructions); everything else about an address is an instruction immediate.
    __device__ __forceinline__ void toggle_rows(int bit) {
#pragma unroll
        for (int ks = 0; ks < D / 16; ++ks) row[ks] ^= bit;
    }
    __device__ __forceinline__ void toggle_tr(int bit) {
#pragma unroll
        for (int dt = 0; dt < D / 32; ++dt) { tr1[dt] ^= bit; tr2[dt] ^= bit; }
    }
};

__device__ __forceinline__ bf16x8 tr_load(const MMU_LDS char* p1, const MMU_LDS char* p2) {
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((MMU_LDS bf16x4*)p1);
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((MMU_LDS bf16x4*)p2);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

__device__ __forceinline__ bf16x8 pack8(const f32x16& v, int s) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = f2bf(v[8 * s + j]);
    return r;
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// Store one gradient row fragment held as acc[dt][4g+j] <-> d = dt*32 + 8g + 4*hl + j (D = 128), optionally applying
// the transpose rotation of RoPE to the (d, d+64) pairs = (dt, dt+2): same arithmetic as rope_kernel(inverse=1) on the
// bf16-rounded gradient.
__device__ __forceinline__ void store_grad_row(bf16* dst, const f32x16 (&acc)[4], int hl, const bf16* cos_row,
                                               const bf16* sin_row) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = dt * 32 + 8 * g + 4 * hl;  // < 64
            bf16x4 t, u;
            if (cos_row) {
                const bf16x4 c4 = *reinterpret_cast<const bf16x4*>(cos_row + d);
                const bf16x4 s4 = *reinterpret_cast<const bf16x4*>(sin_row + d);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a = round_bf16(acc[dt][4 * g + j]), b = round_bf16(acc[dt + 2][4 * g + j]);
                    const float cf = bf2f(c4[j]), sf = bf2f(s4[j]);
                    t[j] = f2bf(round_bf16(a * cf) + round_bf16(b * sf));
                    u[j] = f2bf(round_bf16(b * cf) + round_bf16(-a * sf));
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    t[j] = f2bf(acc[dt][4 * g + j]);
                    u[j] = f2bf(acc[dt + 2][4 * g + j]);
                }
            }
            *reinterpret_cast<bf16x4*>(dst + d) = t;
            *reinterpret_cast<bf16x4*>(dst + d + 64) = u;
        }
}


typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// 16-B global store under a per-lane predicate WITHOUT a branch and without the compiler seeing a store: EXEC is masked
// and restored inside the statement.  For the persistent dK/dV kernel, where a predicated C++ store (s_cbranch_execz around
// it) or any store the compiler still counts as outstanding brings `s_waitcnt vmcnt` into the step loop (see its header).
// The caller drains with a real s_waitcnt vmcnt(0) before the stored registers are reused.
__device__ __forceinline__ void store16_masked_asm(void* ptr, u32x4 val, int pred) {
    unsigned long long m, keep;
    asm volatile(
        "v_cmp_ne_u32_e64 %0, 0, %2\n\ts_and_saveexec_b64 %1, %0\n\tglobal_store_dwordx4 %3, %4, off\n\ts_mov_b64 exec, %1"
        : "=&s"(m), "=&s"(keep)
        : "v"(pred), "v"(ptr), "v"(val)
        : "memory", "scc");  // s_and_saveexec writes SCC
}

// A 32-row x 128-column bf16 tile held "row on the lane" — lane (row = lane & 31, hl = lane >> 5) holds w[dt][g] = the
// four columns 32 dt + 8 g + 4 hl .. + 3 of its row, the layout of the dQ^T / dK^T / dV^T accumulators — written out
// through a 4-KiB LDS area private to the wave, one 64-column half at a time:
//  * row-major: 8 ds_write_b64 in, 4 ds_read_b128 out with 8 lanes per row, so a store instruction writes 8 whole
//    128-B lines (storing straight from the registers touches 32 rows x 32 B per instruction: ~440 cycles of issue);
//  * transposed (t0 != nullptr): the same image read column-wise with ds_read_b64_tr_b16 (a 16-lane group takes
//    4 rows x 16 columns, every lane leaves with 4 consecutive rows of ONE column); two reads = 8 rows = 16 B of a row of
//    the transposed matrix, a store instruction writes 16 such rows x 64 B.
// Image: [row][16 granules of 8 B], granule ^= ((row >> 1) & 3) << 1.  g0 -> (first row, first column) of the row-major
// destination (row stride ldg), t0 -> (first column, first row) of the transposed one (row stride ldt).  Rows
// [0, rows_valid) are written (the transposed copy in whole groups of 8 rows: the caller only asks for it when the valid
// row count is a multiple of 8); ASM_ST: the stores are store16_masked_asm.  EXEC must be full on entry
// (ds_read_b64_tr_b16), and nothing but this wave may touch xw[0, 4096).
template <bool ASM_ST>
__device__ __forceinline__ void store_tile32x128_via_lds(MMU_LDS char* xw, const u32x2 (&w)[4][4], int lane, bf16* g0,
                                                         int ldg, bf16* t0, int ldt, int rows_valid) {
    const int kr = lane & 31, rr = lane >> 3, hl = lane >> 5;
    const int w0 = kr * 128 + ((hl ^ (((kr >> 1) & 3) << 1)) << 3);
    const int r0 = rr * 128 + (((lane & 7) ^ ((rr >> 1) & 3)) << 4);
    bf16* gp = g0 + (size_t)rr * ldg + (lane & 7) * 8;
    auto st = [&](bf16* dst, u32x4 v, bool ok) {
        if (ASM_ST) store16_masked_asm(dst, v, ok ? 1 : 0);
        else if (ok) *reinterpret_cast<u32x4*>(dst) = v;
    };
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int dtl = 0; dtl < 2; ++dtl)
#pragma unroll
            for (int g = 0; g < 4; ++g) *(MMU_LDS u32x2*)(xw + (w0 ^ (g << 4)) + dtl * 64) = w[2 * h + dtl][g];
        u32x4 rows[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) rows[i] = *(const MMU_LDS u32x4*)(xw + r0 + i * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) st(gp + (size_t)i * 8 * ldg + h * 64, rows[i], i * 8 + rr < rows_valid);
        if (t0) {
            const int g4 = lane >> 4, i16 = lane & 15;
            const int trow = 8 * g4 + (i16 >> 2);
            const int a0 = trow * 128, a1 = (trow + 4) * 128;
            const int sw0 = ((trow >> 1) & 3) << 1, sw1 = (((trow + 4) >> 1) & 3) << 1;
            bf16* tp = t0 + (size_t)(h * 64 + i16) * ldt + 8 * g4;
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                const int gi = 4 * cb + (i16 & 3);
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((MMU_LDS bf16x4*)(xw + a0 + ((gi ^ sw0) << 3)));
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((MMU_LDS bf16x4*)(xw + a1 + ((gi ^ sw1) << 3)));
                st(tp + (size_t)(16 * cb) * ldt, __builtin_bit_cast(u32x4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7)),
                   8 * g4 + 8 <= rows_valid);
            }
        }
    }
}

// The bf16 words of a gradient tile for store_tile32x128_via_lds: acc[dt][4 g + j] <-> column 32 dt + 8 g + 4 hl + j,
// optionally with the transpose rotation of RoPE on the (d, d + 64) pairs = (dt, dt + 2) (same arithmetic as
// rope_kernel(inverse=1) on the bf16-rounded gradient; cos_row / sin_row: the row's 64 table entries, or nullptr).
__device__ __forceinline__ void grad_words(const f32x16 (&acc)[4], int hl, const bf16* cos_row, const bf16* sin_row,
                                           u32x2 (&w)[4][4]) {
    if (cos_row) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = dt * 32 + 8 * g + 4 * hl;  // < 64
                const bf16x4 c4 = *reinterpret_cast<const bf16x4*>(cos_row + d);
                const bf16x4 s4 = *reinterpret_cast<const bf16x4*>(sin_row + d);
                bf16x4 t, u;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float x = round_bf16(acc[dt][4 * g + j]), y = round_bf16(acc[dt + 2][4 * g + j]);
                    const float cf = bf2f(c4[j]), sf = bf2f(s4[j]);
                    t[j] = f2bf(round_bf16(x * cf) + round_bf16(y * sf));
                    u[j] = f2bf(round_bf16(y * cf) + round_bf16(-x * sf));
                }
                w[dt][g] = __builtin_bit_cast(u32x2, t);
                w[dt + 2][g] = __builtin_bit_cast(u32x2, u);
            }
    } else {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 t;
#pragma unroll
                for (int j = 0; j < 4; ++j) t[j] = f2bf(acc[dt][4 * g + j]);
                w[dt][g] = __builtin_bit_cast(u32x2, t);
            }
    }
}

// Plain store of a gradient row fragment for any head_dim (acc[dt][4g+j] <-> d = dt*32 + 8g + 4*hl + j).
template <int NDT>
__device__ __forceinline__ void store_grad_row_plain(bf16* dst, const f32x16 (&acc)[NDT], int hl) {
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bf16x4 t;
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] = f2bf(acc[dt][4 * g + j]);
            *reinterpret_cast<bf16x4*>(dst + dt * 32 + 8 * g + 4 * hl) = t;
        }
}

struct AttnArgs {
    const bf16* q;
    const bf16* k;
    const bf16* v;
    int ld;  // row stride of q/k/v (elements)
    bf16* o;
    int ldo;
    float* lse;        // [B][H][S] natural-log lse of the scaled scores (+inf for rows with no valid key)
    const int* kmask;  // [B][S] 1 = attend, 0 = masked key; may be null
    int B, S, H;
    float scale;
    // backward
    const bf16* dout;
    int lddo;
    const float* delta;  // [B][H][S]
    bf16* dq;
    bf16* dk;
    bf16* dv;
    int ldg;  // row stride of dq/dk/dv
    // optional fused inverse RoPE on the dq / dk stores (head_dim 128): positions [B*S], tables [n_pos][64] bf16
    const int* rope_pos;
    const bf16* rope_cos;
    const bf16* rope_sin;
    // optional transposed copy of the gradients for the q|k|v weight gradient: dT [3][H * D][ldt], element
    // (part, h * D + d, token) with part 0 = dq, 1 = dk, 2 = dv and token = b * S + s — dqkv^T without a transpose pass
    bf16* dT;
    int ldt;
    int fuse_delta;  // backward: the dQ kernel forms delta / lse2 itself and writes them for the dK/dV kernel
};


// Workgroup -> (tile, head, batch) for a 1-D grid of nx * H * B workgroups.  The hardware deals consecutive workgroup
// ids round-robin over the 8 XCDs (each with its own L2), so with the natural order the nx tiles of one (batch, head)
// — which all stream the same K / V (or Q / dO) rows — land on 8 different L2s and every one of them fetches that
// head again.  Here the XCD index (id & 7) selects the (batch, head) pair inside a group of 8 pairs and the remaining
// bits walk that pair's tiles, so one head's tiles run back to back on ONE XCD and hit in its L2.  Heavy (late, causal)
// tiles first.  Falls back to the natural order when H * B is not a multiple of 8.
__device__ __forceinline__ void decode_wg_at(int w, int nx, int H, int B, bool heavy_is_last, int& tile, int& hh,
                                             int& b) {
    const int nbh = H * B;
    int bh, t;
    if ((nbh & 7) == 0) {
        const int xcd = w & 7, slot = w >> 3;
        bh = (slot / nx) * 8 + xcd;
        t = slot % nx;
    } else {
        bh = w / nx;
        t = w % nx;
    }
    tile = heavy_is_last ? nx - 1 - t : t;  // causal: late query tiles / early key tiles carry the most blocks
    hh = bh % H;
    b = bh / H;
}
__device__ __forceinline__ void decode_wg(int nx, int H, int B, bool heavy_is_last, int& tile, int& hh, int& b) {
    decode_wg_at(blockIdx.x, nx, H, B, heavy_is_last, tile, hh, b);
}

template <typename K>
int set_lds(K kern, int bytes) {
    hipError_t e =
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    return (int)e;
}

}  // namespace

// attention_dkdv4.hip: launches the four-wave dK/dV kernel (head_dim 128); returns 0 or a hipError_t
int mmu_internal_launch_dkdv4(const void* attn_args, int causal, hipStream_t stream);
// workgroups of that (persistent) kernel: 0 = one per CU, else a multiple of 8 (tests force several items per workgroup)
int mmu_internal_set_dkdv4_grid(int n);
// elementwise.hip: dst[C][R] (row stride ld_dst) = src[R][C] (row stride ld_src)^T
int mmu_internal_transpose_strided(const void* src, int ld_src, void* dst, int ld_dst, int R, int C, hipStream_t stream);
