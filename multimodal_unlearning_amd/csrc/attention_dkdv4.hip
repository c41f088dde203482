// Four-wave, register-resident, persistent dK/dV kernel of the head_dim-128 attention backward (see the comment above
// the kernel).  A translation unit of its own: it is the one kernel that owns the whole 512-entry register file.
#include "attention_common.h"

namespace {

// LDS-DMA issued from inline asm (guide §5.7): hipcc does not see these loads, so it inserts no `s_waitcnt vmcnt(0)` in
// front of later LDS reads (with the builtin form it drains the whole ring before every transposed read: the run-ahead
// would be lost).  Completion is counted by hand: `s_waitcnt vmcnt(N)` + s_barrier at the top of a step.
// M0 (the DMA's LDS base) is written and restored inside the statement.
// Per-lane address form (ragged stages: rows past the sequence are clamped / redirected per lane)
__device__ __forceinline__ void glds16_asm(const void* gsrc, unsigned lds_wave_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_wave_base)
                 : "memory");
}
__device__ __forceinline__ void glds4_asm(const void* gsrc, unsigned lds_wave_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_wave_base)
                 : "memory");
}
// Scalar base + 32-bit per-lane byte offset (the steady state: one VGPR per piece, no 64-bit address arithmetic).  M0
// is not saved here: nothing else in this translation unit reads M0 (LDS instructions do not on gfx9+, and there is no
// compiler-generated LDS-DMA, s_movrel or sendmsg in the kernel — check the .s for `m0` after a compiler change).
__device__ __forceinline__ void glds16_sv(const void* sbase, unsigned voff, unsigned lds_wave_base) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_wave_base)
                 : "memory");
}
// 4 B per lane from scalar base + 4 * lane; the lane offset is formed inside the statement (v_mbcnt ignores EXEC), so no
// VGPR has to stay live for it across the step loop (hipcc spilled exactly that one to scratch, and its reload
// brought an `s_waitcnt vmcnt(0)` into every step)
__device__ __forceinline__ void glds4_lane(const void* sbase, unsigned lds_wave_base) {
    unsigned t;
    asm volatile(
        "v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_lshlrev_b32 %0, 2, %0\n\t"
        "s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1"
        : "=&v"(t)
        : "s"(sbase), "s"(lds_wave_base)
        : "memory");
}

// s_waitcnt immediates (gfx9 layout: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] at [15:14])
constexpr int waitcnt_vm(int n) { return (n & 15) | ((n >> 4) << 14) | 0x0F70; }

// ------------------------------------------------------------------------------------------------
// dK / dV, round-2 structure (head_dim 128): 256 threads = 4 waves, ONE wave per SIMD with the whole 512-entry
// register file, one PERSISTENT workgroup per CU walking the (key block, head, batch) items  v = blockIdx.x,
// blockIdx.x + gridDim.x, ...  (gridDim.x is a multiple of 8, so the item -> XCD map of decode_wg_at — one head's key
// blocks on one XCD's L2 — is the same as for a one-item-per-workgroup launch, and because 5 k mod 9 walks all key
// blocks every workgroup gets the same mix of long and short causal sweeps).
// Wave w owns keys kvb*128 + 32w .. +31 for a whole item ("key on the lane"): its K and V fragments (the B operands
// of S = Q K^T and dP = dO V^T) are loaded once per item, straight from global memory, and stay in registers (64); its
// dK^T / dV^T accumulators (128, AGPRs) need no cross-wave reduction; no K / V tile lives in LDS.
// The LDS holds only a 4-stage ring of 64-row (Q | dO | lse2, delta) stages filled by LDS-DMA three STEPS ahead, and
// the ring runs across item seams: while an item's last steps and its stores run, the first stages of the next item
// are already landing, so there is no per-item prologue (measured on the one-item-per-workgroup version of this
// kernel: prologue 8.7 k + store tail 10.6 k cycles of a 69 k-cycle average workgroup, tools/attn_stamps.py --dkdv4).
// The nine DMA pieces a wave contributes to a stage are issued at the boundaries of the MFMA regions of a step, not
// in a burst at its top (branch-free: the same code for full and ragged stages).  The barrier that opens a step sits in
// front of the last MFMA region of the previous one, whose LDS reads are already the next step's first row fragments.
// Per 32-row query tile there is ONE body (both tiles of a step always run; masked tiles zero their P in a short block
// of their own) made of four regions of 8 MFMAs; every MFMA gap is its own scheduling region with its slice of LDS reads
// and VALU work pinned in it, see `tile` below.
// At the end of an item each half matrix of dK / dV goes through a 4-KiB LDS area private to the wave so that a store
// instruction writes 8 whole 128-B lines; the next item's K / V fragments are requested before those stores go out and
// everything is drained once per item (with a counted wait hipcc kept re-checking the loads inside the step loop).
// What the compiler must not be allowed to do here, each found the hard way (tools/attn_stamps.py, the .s):
//  * two instantiations of the tile body -> dK / dV in different registers per path, 128 v_accvgpr_mov per step;
//  * a branch between MFMA regions -> the VALU work of a region sinks into the next block, nothing overlaps it;
//  * per-row predicated C++ stores / `if (more)` around the next item's loads -> s_waitcnt vmcnt(15..0) in EVERY step
//    (the structurised control flow hides what is outstanding), i.e. a drain of the DMA ring per step;
//  * a value kept live for the DMA (lane * 4) -> spilled to scratch, reloaded behind vmcnt(0) in every step.
// Measured reason for the rebuild (tools/attn_stamps.py on the 8-wave kernel): 5 850 cycles per iteration for 2 x 1 024
// cycles of MFMA issue per SIMD, an epilogue worth 2.3 iterations (cross-wave partial sums through 128 KiB of LDS),
// one workgroup per CU meeting at a barrier with the next stage only one iteration ahead.
// ------------------------------------------------------------------------------------------------
#ifndef DKDV4_EXP  // timing experiments only (tools/attn_stamps.py): 1 no in-step DMA, 2 no softmax VALU, 4 no transposed
#define DKDV4_EXP 0  // LDS reads, 8 no stores, 16 no row LDS reads, 32 no K/V reload — results are garbage, addresses unchanged
#endif
constexpr int DKDV4_STAGES = 4;

template <bool CAUSAL>
__global__ __launch_bounds__(256, 1) void attn_bwd_dkdv4_kernel(AttnArgs a, int n_items) {
    constexpr int D = 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TB = 64 * D * 2;             // one 64-row tile (Q or dO): 16 KiB
    constexpr int STB = 2 * TB;                // Q | dO of one stage: 32 KiB
    constexpr int ROWC = DKDV4_STAGES * STB;   // row constants: [stage][lse2 64 | delta 64] floats
    constexpr int XLDS = ROWC + DKDV4_STAGES * 512;  // 4 x 4 KiB: each wave's transposition area for its stores
    constexpr int NKS = D / 16, NDT = D / 32, NST = DKDV4_STAGES;
    STAMP(t_start);
#ifdef MMU_ATTN_STAMP
    unsigned long long t_vm = 0, t_bar = 0, t_clean = 0, n_clean = 0, t_epi = 0, n_it = 0, n_steps = 0, t_e_math = 0, t_e_kv = 0,
                       t_e_st = 0, t_masked = 0;
#endif
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hl = lane >> 5;
    const int S = a.S, G = gridDim.x;
    const int nx = (S + 127) / 128;
    if ((int)blockIdx.x >= n_items) return;

    // per-lane fragment offsets inside a stage's Q tile (query tile 0; tile 1 = + 32 rows, dO = + TB: immediates)
    int off_row[NKS], off_tr1[NDT], off_tr2[NDT];
    {
        const int r = lane & 31;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) off_row[ks] = r * (D * 2) + (((2 * ks + hl) ^ swz<D, KIND_ROW>(r)) << 4);
        const int g = lane >> 4, gb = g & 1, h = g >> 1, i = lane & 15, q = i >> 2, pp = i & 3;
        const int r1 = 4 * h + q, r2 = r1 + 8, sub = (pp & 1) << 3;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
            const int ch = 4 * dt + 2 * gb + (pp >> 1);
            off_tr1[dt] = r1 * (D * 2) + ((ch ^ swz<D, KIND_ROW>(r1)) << 4) + sub;
            off_tr2[dt] = r2 * (D * 2) + ((ch ^ swz<D, KIND_ROW>(r2)) << 4) + sub;
        }
    }
    const unsigned lds_base = (unsigned)(uintptr_t)(MMU_LDS char*)smem;
    const float c = a.scale * LOG2E;

    // ---- the DMA cursor: the stage that is issued next (all wave-uniform).  Past the last stage of the last item it
    // stays where it is: the pieces then re-read that stage into a ring slot nobody reads, which keeps the issue and the
    // counted waits free of branches.
    int d_item = blockIdx.x, d_pr, d_pr1, issued = 0, gs_started = 0;
    const bf16* d_q;
    const bf16* d_do;
    const float* d_lse;
    const float* d_dl;
    // per stage (dma_stage): this wave's 16 rows of the Q / dO tile, its row constants, its LDS addresses
    const char* s_q;
    const char* s_do;
    const float* s_rc;
    unsigned s_lds, s_lds_rc;
    int s_lim, s_rows;
    const unsigned ldq2 = (unsigned)a.ld * 2u, lddo2 = (unsigned)a.lddo * 2u;  // row strides in bytes (< 2^24: launcher)
    auto dma_stage = [&]() {
        const int row0 = d_pr * 64;
        const int wrow = min(row0 + wave * 16, S - 1);  // this wave's first row, kept inside the sequence
        s_q = reinterpret_cast<const char*>(d_q + (size_t)wrow * a.ld);
        s_do = reinterpret_cast<const char*>(d_do + (size_t)wrow * a.lddo);
        s_lim = S - 1 - wrow;  // rows wrow + 0 .. s_lim exist (>= 15 on a full stage)
        s_rows = S - row0;     // row constants that exist (>= 64 on a full stage)
        s_rc = ((wave & 1) ? d_dl : d_lse) + row0;  // waves 0, 2: the 64 lse2 values; waves 1, 3: the 64 deltas (twice each:
        s_lds = lds_base + (issued % NST) * STB + wave * 4096;  // the same bytes to the same place, and no lane predicate)
        s_lds_rc = lds_base + ROWC + ((issued % NST) * 128 + (wave & 1) * 64) * 4;
    };
    auto dma_open = [&](int v) {
        int kvb, hh, b;
        decode_wg_at(v, nx, a.H, a.B, false, kvb, hh, b);
        const size_t rowbase = (size_t)b * S;
        d_q = a.q + rowbase * a.ld + hh * D;
        d_do = a.dout + rowbase * a.lddo + hh * D;
        d_lse = a.delta + ((size_t)a.B * a.H + (size_t)b * a.H + hh) * S;  // lse * log2(e)
        d_dl = a.delta + ((size_t)b * a.H + hh) * S;
        d_pr = CAUSAL ? kvb * 2 : 0;
        d_pr1 = (S + 63) / 64;
    };
    dma_open(d_item);
    dma_stage();
    // piece p (0..3 Q, 4..7 dO, 8 row constants) of the cursor's stage: straight-line code, the same for a full stage and
    // for the ragged last stage of a sequence that is not a multiple of 64 (a branch would cut the MFMA regions the
    // pieces sit between into basic blocks).  Rows past the sequence re-read row S-1 and take their lse2 from the +inf
    // page, so P = 0 exactly for them.
    const int rr4 = lane >> 4;
    const unsigned chunk0 = (unsigned)(((lane & 15) ^ (rr4 << 2)) << 4);
    auto dma_piece = [&](int p) {
        if (p < 8) {
            const int t = p & 3;
            const bool is_do = p >= 4;
            const unsigned row = (unsigned)min(4 * t + rr4, s_lim);
            const unsigned voff = __umul24(row, is_do ? lddo2 : ldq2) + (chunk0 ^ (unsigned)(t << 4));  // full-rate 24-bit multiply
            glds16_sv(is_do ? s_do : s_q, voff, s_lds + (is_do ? TB : 0) + t * 1024);
        } else {
            const float* src = lane < s_rows ? s_rc + lane : reinterpret_cast<const float*>((wave & 1) ? mmu_zero16 : mmu_inf16);
            glds4_asm(src, __builtin_amdgcn_readfirstlane(s_lds_rc));
        }
    };
    auto dma_advance = [&]() {
        ++issued;
        if (d_pr + 1 < d_pr1) ++d_pr;
        else if (d_item + G < n_items) { d_item += G; dma_open(d_item); }
        dma_stage();
    };
    auto dma_pieces = [&](int p0, int n) {
#pragma unroll
        for (int q = p0; q < p0 + n; ++q)
            if (!(DKDV4_EXP & 1) || gs_started == 0) dma_piece(q);
        if (p0 + n == 9) dma_advance();
    };
#pragma unroll
    for (int i = 0; i < NST - 1; ++i) dma_pieces(0, 9);

    // ---- item state of the compute side
    int kvb, hh, b;
    decode_wg_at(blockIdx.x, nx, a.H, a.B, false, kvb, hh, b);
    // K / V fragments of this wave's 32 keys: lane (key, hl) holds K[key][16 ks + 8 hl .. +7] (B operand map).  Keys past
    // the sequence re-read row S-1 (their P is masked to 0, the operands only have to be finite).
    bf16x8 kf[NKS], vf[NKS];
    bool kvalid;
    auto load_kv = [&](int kvb_, int hh_, int b_) {
        const size_t rowbase = (size_t)b_ * S;
        const int key = kvb_ * 128 + wave * 32 + (lane & 31);
        const size_t krow = (rowbase + (key < S ? key : S - 1)) * a.ld + hh_ * D + 8 * hl;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            kf[ks] = *reinterpret_cast<const bf16x8*>(a.k + krow + 16 * ks);
            vf[ks] = *reinterpret_cast<const bf16x8*>(a.v + krow + 16 * ks);
        }
        kvalid = key < S && (a.kmask == nullptr || a.kmask[rowbase + key] != 0);
    };
    load_kv(kvb, hh, b);
    // The modelled wait tells the compiler that these loads have landed, and the empty statement that "rewrites" the
    // fragments makes it settle every wait it still wants for them HERE: without it hipcc re-checks the loads with
    // s_waitcnt vmcnt(15) .. vmcnt(0) in front of the first MFMAs of every step and drains the DMA ring each time.
    auto settle_kv = [&]() {
        asm volatile("" : "+v"(kf[0]), "+v"(kf[1]), "+v"(kf[2]), "+v"(kf[3]), "+v"(kf[4]), "+v"(kf[5]), "+v"(kf[6]), "+v"(kf[7]),
                          "+v"(vf[0]), "+v"(vf[1]), "+v"(vf[2]), "+v"(vf[3]), "+v"(vf[4]), "+v"(vf[5]), "+v"(vf[6]), "+v"(vf[7]));
    };
    __builtin_amdgcn_s_waitcnt(waitcnt_vm(0));
    settle_kv();

    f32x16 dk[NDT], dv[NDT];
    gs_started = 1;
    const MMU_LDS char* lds0 = (const MMU_LDS char*)smem;
    int gs = 0;  // steps done by this workgroup = index of the stage the next step reads

    // The barrier that opens step gs+1 ("every wave's pieces of stage gs+1 are in LDS; slot gs%NST is free") sits in front
    // of the LAST MFMA region of step gs, whose LDS reads are then already the first row fragments of step gs+1: a step
    // starts on its first MFMA with nothing outstanding.  The first step of the workgroup is opened here.
    bf16x8 fq[NKS];
    f32x4 l4[4];
    asm volatile("s_waitcnt vmcnt(18)" ::: "memory");  // stage 0 has landed (stages 1, 2 may be outstanding)
    __builtin_amdgcn_s_barrier();
    {
        const MMU_LDS float* rc0 = (const MMU_LDS float*)(smem + ROWC);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) fq[ks] = *(const MMU_LDS bf16x8*)(lds0 + off_row[ks]);
#pragma unroll
        for (int g = 0; g < 4; ++g) l4[g] = *(const MMU_LDS f32x4*)(rc0 + 8 * g + 4 * hl);
    }

    for (int item = blockIdx.x; item < n_items; item += G) {
        const int key0 = kvb * 128 + wave * 32;
        const int key = key0 + (lane & 31);
        const bool all_valid = __all(kvalid);
#pragma unroll
        for (int i = 0; i < NDT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dk[i][r] = 0.f; dv[i][r] = 0.f; }
        const int pr0 = CAUSAL ? kvb * 2 : 0;
        const int pr1 = (S + 63) / 64;

        for (int pr = pr0; pr < pr1; ++pr, ++gs) {
            const int buf = gs % NST, nbuf = (gs + 1) % NST;
            STAMP(t_w2);
#ifdef MMU_ATTN_STAMP
            n_steps += 1;
#endif
            const MMU_LDS char* qt = lds0 + buf * STB;  // wave-uniform stage base (one scalar add per address below)
            const MMU_LDS float* rc = (const MMU_LDS float*)(smem + ROWC) + buf * 128;
            const MMU_LDS char* qt_next = lds0 + nbuf * STB;
            const MMU_LDS float* rc_next = (const MMU_LDS float*)(smem + ROWC) + nbuf * 128;
#define SB() __builtin_amdgcn_sched_barrier(0)
#define SG(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
// keeps the values computed in a region in that region (hipcc otherwise sinks a region's VALU work into the block of its
// first use, behind the next branch, where nothing overlaps it)
#define PIN8(v) asm volatile("" ::"v"(v))
            // One 32-row query tile against this wave's 32 keys: ONE body for every kind of tile (a second instantiation
            // made hipcc keep the dK / dV accumulators in different registers per path and copy all 128 at every merge).
            // Four scheduling regions of 8 MFMAs each; every region carries the LDS reads of a LATER region and the VALU
            // work of an EARLIER one between its MFMAs:
            //   R1  S = Q K^T chain          | transposed dO^T fragments of rows 0..15, dO row fragments
            //   R2  dP = dO V^T chain        | P = exp2(S c - lse2) | transposed dO^T, rows 16..31, delta
            //       (masked tiles only: P = 0 on the rows above the diagonal / for padded keys; a short block of its own)
            //   R3  dV^T += dO^T P           | bf16 P, dS = P (dP - delta) and its bf16 fragments | transposed Q^T fragments
            //   R4  dK^T += Q^T dS           | Q row fragments + lse2 of the NEXT tile (half 0 only)
            auto tile = [&](int half, int p0, int np, bf16x8(&fq)[NKS], f32x4(&l)[4]) {
                const MMU_LDS char* qtile = qt + half * 32 * (D * 2);
                bf16x8 to[NDT], tq[NDT], to2[NDT], tq2[NDT], fo[NKS];
                f32x4 d[4];
                f32x16 sv, dpv;
                const int q_first = pr * 64 + half * 32;
                const bool diag = CAUSAL && (q_first < key0 + 31);
                const bool masked = diag || !all_valid;  // wave-uniform
                // DMA pieces p0 .. p0+np-1 of the stage three steps ahead go out at the region boundaries.
                // Every MFMA gap is a scheduling region of its own (sched_barrier after each): what overlaps which MFMA
                // is written down here, not left to the scheduler's grouping heuristics.
                const int n1 = np > 0 ? 1 : 0, n2 = np > 1 ? 1 : 0, n3 = np > 2 ? 1 : 0, n4 = np > 3 ? np - 3 : 0;
                dma_pieces(p0, n1);
                // ---- R1
#pragma unroll
                for (int r = 0; r < 16; ++r) sv[r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    sv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fq[ks], kf[ks], sv, 0, 0, 0);
                    if (!(ks & 1)) {
                        const int dt = ks >> 1;
                        to[dt] = (DKDV4_EXP & 4) ? kf[dt] : tr_load(qtile + TB + off_tr1[dt], qtile + TB + off_tr2[dt]);
                    }
                    fo[ks] = (DKDV4_EXP & 16) ? kf[ks] : *(const MMU_LDS bf16x8*)(qtile + TB + off_row[ks]);
                    SB();
                }
                dma_pieces(p0 + n1, n2);
                // ---- R2
#pragma unroll
                for (int r = 0; r < 16; ++r) dpv[r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    dpv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fo[ks], vf[ks], dpv, 0, 0, 0);
                    if (!(DKDV4_EXP & 2)) {
#pragma unroll
                        for (int e = 2 * ks; e < 2 * ks + 2; ++e) sv[e] = fast_exp2(__builtin_fmaf(sv[e], c, -l[e >> 2][e & 3]));
                        asm volatile("" : "+v"(sv[2 * ks]), "+v"(sv[2 * ks + 1]));  // the slice stays in this gap (see PIN8)
                    }
                    if (ks < 4) to2[ks] = (DKDV4_EXP & 4) ? kf[4 + ks]
                                                          : tr_load(qtile + 16 * (D * 2) + TB + off_tr1[ks], qtile + 16 * (D * 2) + TB + off_tr2[ks]);
                    else d[ks - 4] = *(const MMU_LDS f32x4*)(rc + 64 + half * 32 + 8 * (ks - 4) + 4 * hl);
                    SB();
                }
                if (masked) {  // row 8g + 4hl + j of this tile is masked for this lane's key iff 8g + j < kd2
                    const int kd2 = !kvalid ? 1000 : (diag ? key - q_first - 4 * hl : -1000);
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int j = 0; j < 4; ++j) sv[4 * g + j] = (8 * g + j < kd2) ? 0.f : sv[4 * g + j];
                }
                dma_pieces(p0 + n1 + n2, n3);
                // ---- R3
                const bf16x8 pf = pack8(sv, 0);
                bf16x8 pf2, dsf, dsf2;
                SB();
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (i < 4) dv[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(to[i], pf, dv[i], 0, 0, 0);
                    else dv[i - 4] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(to2[i - 4], pf2, dv[i - 4], 0, 0, 0);
                    if (i == 0) { pf2 = pack8(sv, 1); asm volatile("" : "+v"(pf2)); }
                    if (i < 4) tq[i] = (DKDV4_EXP & 4) ? vf[i] : tr_load(qtile + off_tr1[i], qtile + off_tr2[i]);
                    else tq2[i - 4] = (DKDV4_EXP & 4) ? vf[i] : tr_load(qtile + 16 * (D * 2) + off_tr1[i - 4], qtile + 16 * (D * 2) + off_tr2[i - 4]);
                    if (!(DKDV4_EXP & 2)) {
#pragma unroll
                        for (int e = 2 * i; e < 2 * i + 2; ++e) dpv[e] = sv[e] * (dpv[e] - d[e >> 2][e & 3]);
                        asm volatile("" : "+v"(dpv[2 * i]), "+v"(dpv[2 * i + 1]));
                    }
                    if (i == 4) { dsf = pack8(dpv, 0); asm volatile("" : "+v"(dsf)); }
                    SB();
                }
                dma_pieces(p0 + n1 + n2 + n3, n4);
                // ---- R4 (tile 0: next = tile 1 of this stage; tile 1: next = tile 0 of the next stage, behind the
                // barrier that opens the next step: this step's stage is not read after this point)
                const MMU_LDS char* qn = half == 0 ? qt + 32 * (D * 2) : qt_next;
                const MMU_LDS float* rn = half == 0 ? rc + 32 : rc_next;
                if (half == 1) {
#ifdef MMU_ATTN_STAMP
                    const unsigned long long t_b0 = stamp_now();
#endif
                    asm volatile("s_waitcnt vmcnt(18)" ::: "memory");  // stage gs+1 has landed (gs+2, gs+3 may be outstanding)
#ifdef MMU_ATTN_STAMP
                    const unsigned long long t_b1 = stamp_now();
#endif
                    __builtin_amdgcn_s_barrier();
#ifdef MMU_ATTN_STAMP
                    const unsigned long long t_b2 = stamp_now();
                    t_vm += t_b1 - t_b0;
                    t_bar += t_b2 - t_b1;
#endif
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (i < 4) dk[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tq[i], dsf, dk[i], 0, 0, 0);
                    else dk[i - 4] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tq2[i - 4], dsf2, dk[i - 4], 0, 0, 0);
                    if (i == 0) { dsf2 = pack8(dpv, 1); asm volatile("" : "+v"(dsf2)); }
                    fq[i] = (DKDV4_EXP & 16) ? vf[i] : *(const MMU_LDS bf16x8*)(qn + off_row[i]);
                    if (i < 4) l[i] = *(const MMU_LDS f32x4*)(rn + 8 * i + 4 * hl);
                    SB();
                }
            };
            // Both tiles always run: a tile above the causal diagonal or past the sequence end contributes exact zeros (its
            // P is masked / its lse2 is +inf), and the waves meet at the next barrier anyway, so skipping it would only
            // bring branches (and with them a second kind of step) into the body.
#ifdef MMU_ATTN_STAMP
            const bool clean = all_valid && !(CAUSAL && pr * 64 < key0 + 31) && pr * 64 + 32 < S;
#endif
            tile(0, 0, 5, fq, l4);
            tile(1, 5, 4, fq, l4);
#ifdef MMU_ATTN_STAMP
            if (clean) { t_clean += stamp_now() - t_w2; n_clean += 1; }
            else t_masked += stamp_now() - t_w2;
#endif
#undef SB
#undef SG
#undef PIN8
        }

        // ---- the item's gradients: scale dK, transpose rotation of RoPE on the (d, d+64) pairs of dK (same arithmetic
        // as rope_kernel(inverse=1) on the bf16-rounded gradient), bf16; lane (key, hl) holds d = 32 dt + 8 g + 4 hl + j.
        STAMP(t_e0);
        const int out_key = key, out_hh = hh, out_b = b;  // the row index is re-formed below (a 64-bit VGPR pair less to keep)
        u32x2 wk[NDT][4], wv[NDT][4];  // bf16x4 words of dK / dV per (dt, g)
        {
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) dk[dt][r] *= a.scale;
            const bf16* cr = nullptr;
            const bf16* sr = nullptr;
            if (a.rope_pos) {
                const int pos = out_key < S ? a.rope_pos[(size_t)out_b * S + out_key] : 0;
                cr = a.rope_cos + (size_t)pos * 64;
                sr = a.rope_sin + (size_t)pos * 64;
            }
            grad_words(dk, hl, cr, sr, wk);
            grad_words(dv, hl, nullptr, nullptr, wv);
        }
        STAMP(t_e1);
        // the next item's K / V (and key mask) are requested BEFORE this item's stores go out; the wait in front of the
        // next item's first step then only has to leave the stores outstanding
        // (unconditionally: after the last item the current one is simply requested again — with the request, the wait
        // and the statement that settles it on ONE path the compiler's count of what is outstanding is exact; behind an
        // `if (more)` its structurised control flow lost track and every step re-checked the 16 loads)
        const bool more = item + G < n_items;
        decode_wg_at(more ? item + G : item, nx, a.H, a.B, false, kvb, hh, b);
        if (!((DKDV4_EXP & 32) && a.scale != 12345.f)) load_kv(kvb, hh, b);
        STAMP(t_e2);
        // Stores: a wave instruction that writes "one key per lane" touches 32 rows (32 B each) and costs ~440 cycles of
        // issue (measured: 7.1 k cycles for the 16 of an item); instead each half matrix (32 keys x 64 d) goes through a
        // 4-KiB LDS area private to the wave — 8 ds_write_b64 in, 4 ds_read_b128 out with 8 lanes per key — so a store
        // instruction writes 8 whole 128-B lines.  [key][16 granules of 8 B], granule ^= ((key >> 1) & 3) << 1.
        // One straight path of stores: with a per-row predicate in C++ (and its skip branches) hipcc could no longer tell
        // what is outstanding at the seam and re-checked the next item's K / V loads in every step of the loop.  Rows past
        // the sequence end (a ragged last key block) are masked inside the store statement itself (store16_masked_asm).
        const bool stores = key0 < S;  // wave-uniform
        auto store_all = [&]() {
            MMU_LDS char* xw = (MMU_LDS char*)smem + XLDS + wave * 4096;
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));  // keeps the address arithmetic below here (hoisted to the item's top it was spilled)
            const size_t row0 = (size_t)out_b * S + key0;
            const size_t g_off = row0 * a.ldg + out_hh * D;
            bf16* t_k = a.dT ? a.dT + ((size_t)a.H * D + out_hh * D) * a.ldt + row0 : nullptr;
            bf16* t_v = a.dT ? t_k + (size_t)a.H * D * a.ldt : nullptr;
            const int rows_valid = min(32, S - key0);
            store_tile32x128_via_lds<true>(xw, wk, lane_e, a.dk + g_off, a.ldg, t_k, a.ldt, rows_valid);
            store_tile32x128_via_lds<true>(xw, wv, lane_e, a.dv + g_off, a.ldg, t_v, a.ldt, rows_valid);
        };
        const bool do_stores = !((DKDV4_EXP & 8) && a.scale != 12345.f);  // (the experiment keeps the stores reachable: no DCE)
        if (stores && do_stores) store_all();
        STAMP(t_e3);
        // A full drain: the compiler only counts the K / V loads (the stores above are inside asm statements), so this is
        // the wait that lets it consider them landed; the stores went out some thousand cycles ago and cost ~280 cycles of
        // wait here (stamps).  Any weaker form left `s_waitcnt vmcnt(11) .. vmcnt(0)` in front of the loop's first writes to
        // the registers the loads target — in EVERY step, i.e. a drain of the DMA ring per step.
        __builtin_amdgcn_s_waitcnt(waitcnt_vm(0));
        settle_kv();
#ifdef MMU_ATTN_STAMP
        t_e_st += t_e3 - t_e2;
#endif
#ifdef MMU_ATTN_STAMP
        t_epi += stamp_now() - t_e0;
        t_e_math += t_e1 - t_e0;
        t_e_kv += t_e2 - t_e1;
        n_it += 1;
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the phantom stages behind the last step
#ifdef MMU_ATTN_STAMP
    {
        const unsigned long long t_end = stamp_now();
        if (threadIdx.x == 0 && blockIdx.x < 4096) {
            unsigned long long* o = g_attn_stamps + 16 * blockIdx.x;
            o[0] = n_it; o[1] = n_steps; o[2] = t_vm; o[3] = t_bar;
            o[4] = t_end - t_start; o[5] = t_epi; o[6] = t_clean; o[7] = n_clean;
            o[8] = t_e_math; o[9] = t_e_kv; o[10] = t_e_st; o[11] = t_masked;
        }
    }
#endif
}

int g_dkdv4_grid = 0;  // 0 = one workgroup per CU

}  // namespace

#ifdef MMU_ATTN_STAMP
extern "C" int mmu_debug_read_dkdv4_stamps(unsigned long long* host_out, int n_blocks) {
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_attn_stamps), (size_t)n_blocks * 128, 0, hipMemcpyDeviceToHost);
}
#endif

int mmu_internal_set_dkdv4_grid(int n) {
    if (n < 0 || (n & 7)) return 1;  // multiples of 8 keep the item -> XCD map
    g_dkdv4_grid = n;
    return 0;
}

int mmu_internal_launch_dkdv4(const void* attn_args, int causal, hipStream_t stream) {
    const AttnArgs& a = *reinterpret_cast<const AttnArgs*>(attn_args);
    constexpr int LDS4 = DKDV4_STAGES * (2 * 64 * 128 * 2) + DKDV4_STAGES * 128 * 4 + 4 * 4096;
    static bool cfg[2] = {false, false};
    static int n_cu = 0;
    if (!cfg[causal ? 1 : 0]) {
        int e = causal ? set_lds(attn_bwd_dkdv4_kernel<true>, LDS4) : set_lds(attn_bwd_dkdv4_kernel<false>, LDS4);
        if (e) return e;
        cfg[causal ? 1 : 0] = true;
    }
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 1;
        n_cu = prop.multiProcessorCount;
    }
    const int n_items = ((a.S + 127) / 128) * a.H * a.B;
    int g = g_dkdv4_grid ? g_dkdv4_grid : (n_cu & ~7);
    if (g > n_items) g = (n_items + 7) & ~7;
    const dim3 grid(g);
    if (causal) hipLaunchKernelGGL((attn_bwd_dkdv4_kernel<true>), grid, dim3(256), LDS4, stream, a, n_items);
    else hipLaunchKernelGGL((attn_bwd_dkdv4_kernel<false>), grid, dim3(256), LDS4, stream, a, n_items);
    MMU_CHECK_LAUNCH();
    return 0;
}
