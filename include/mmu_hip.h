/* C-ABI of libmmu_hip.so — the MI355X (gfx950) kernels behind the forget/retain step.
 *
 * The reference (somvy/multimodal_unlearning) is pure Python and has no native ABI; each entry point below
 * names the reference call (file:line under /root/reference, or the transformers op that call reaches)
 * whose arithmetic it replaces.  Conventions:
 *   - every pointer is a DEVICE pointer unless named host_*; bf16 tensors are `void*` to raw bfloat16 bits;
 *   - row-major, leading dimensions in ELEMENTS; `stream` is a hipStream_t (0 = default stream);
 *   - nothing allocates, synchronises or touches host memory: safe to capture into a hipGraph;
 *   - return 0 on success, a hipError_t (>0) from the runtime, or -1 for an unsupported argument.
 */
#ifndef MMU_HIP_H
#define MMU_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

/* ABI version, bumped on any signature change or added entry point (4 = round 4: additions only). */
int mmu_abi_version(void);

/* ---- GEMM (nn.Linear fwd / dgrad / wgrad reached from mm/trainer.py:180-433 `model(**inputs)` and the
 * backward of mm/forget.py:190 trainer.train(); SURVEY §8a K1,K3,K5,K8,K11,K12,K13) ----
 * C[M,N] = epi(sum_k Aop[m,k] Bop[n,k]), fp32 accumulate, bf16 out.
 * layout 0 (NT): A[M][K], B[N][K]   forward  y = x W^T
 * layout 1 (NN): A[M][K], B[K][N]   dgrad    dx = dy W
 * layout 2 (TN): A[K][M], B[K][N]   wgrad    dW = dy^T x
 * epi: v = bf16(acc + bias[n]); act (0 none, 1 gelu-erf, 2 quick-gelu) -> bf16; + resid[m][n] -> bf16
 *      (the rounding points of transformers' bf16 modules); accumulate=1: C = bf16(float(C) + acc). */
int mmu_gemm_bf16(int layout, const void* A, const void* B, void* C, const void* bias, const void* resid,
                  int M, int N, int K, int lda, int ldb, int ldc, int ldr, int act, int accumulate, void* stream);
/* mmu_gemm_bf16 plus a fused low-rank update acc += U[M][lora_k] * BL[N][lora_k]^T before the epilogue (peft LoRA,
 * mm/forget.py:126-137: y = W x + (alpha/r) B (A x); U = (alpha/r) x A^T, BL = B).  U, BL: contiguous row-major
 * bf16, lora_k in {32, 64}, unused columns zero (block structure of fused q|k|v and gate|up adapters). */
int mmu_gemm_bf16_lora(int layout, const void* A, const void* B, void* C, const void* bias, const void* resid,
                       int M, int N, int K, int lda, int ldb, int ldc, int ldr, int act, int accumulate,
                       const void* lora_u, const void* lora_b, int lora_k, void* stream);
/* 1 if mmu_gemm_bf16 would use the 256x256 kernel for this shape. */
int mmu_gemm_uses_256(int M, int N, int K);
/* y = x W^T (+LoRA) with apply_rotary_pos_emb (modeling_llama.py rotate-half, head_dim 128) applied to columns
 * [0, rope_cols) in the epilogue — the q|k part of the fused qkv projection.  rope_pos [M] int32, tables [n_pos][64]
 * bf16.  Returns -2 if the shape would not run on the 256x256 kernel (caller: mmu_gemm_bf16 + mmu_rope). */
int mmu_gemm_bf16_rope(int layout, const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                       const void* lora_u, const void* lora_b, int lora_k, const int* rope_pos, const void* rope_cos,
                       const void* rope_sin, int rope_cols, void* stream);
/* LlamaMLP down(act_fn(gate(x)) * up(x)) (modeling_llama.py:174-176) with the elementwise stage inside the GEMMs (256x256
 * kernel; -2 when the shape would not run on it: the caller then issues mmu_gemm_bf16 + mmu_swiglu_*).
 * mmu_gemm_swiglu_fwd: gu [M][2I] = x W^T (W = [gate rows | up rows] [2I][K], natural order), act [M][I] = bf16(silu(g)) * u
 *   and, act_t != NULL, act^T [I][ldt] (M % 8 == 0).  A 256-column tile takes its B rows from BOTH halves of W, so a wave
 *   holds g and u of the same hidden unit.  Bit-identical to mmu_gemm_bf16 + mmu_swiglu_fwd[_both].
 * mmu_gemm_swiglu_bwd: dact = dh W_down stays in registers (w_layout 0: w = W_down^T [I][K]; 1: w = W_down [K][I]); with g, u
 *   from gu the epilogue writes dgu [M][2I] and, dgu_t != NULL, dgu^T [2I][ldt].  Bit-identical to mmu_gemm_bf16 +
 *   mmu_swiglu_bwd[_t]. */
int mmu_gemm_swiglu_fwd(const void* x, const void* w_gu, void* gu, void* act, void* act_t, int M, int I, int K, int ldx,
                        int ldw, int ldgu, int ldact, int ldt, void* stream);
int mmu_gemm_swiglu_bwd(const void* dh, const void* w, int w_layout, const void* gu, void* dgu, void* dgu_t, int M, int I,
                        int K, int lddh, int ldw, int ldgu, int lddgu, int ldt, void* stream);
/* Split-K GEMM for small outputs (skinny LoRA adapter wgrads: [32,in] and [out,32] reduced over all tokens): `splits`
 * blocks per tile each reduce a K range and atomically add their fp32 partial into c32 [M][N] (zero on entry).
 * mmu_fold_f32_into_bf16: dst_bf16 += c32 (* mask); c32 = 0. */
int mmu_gemm_bf16_splitk(int layout, const void* A, const void* B, float* c32, int M, int N, int K, int lda, int ldb,
                         int splits, void* stream);
int mmu_fold_f32_into_bf16(float* src32, void* dst, const void* mask, long long n, void* stream);
/* Tile configuration override (A/B tests, tuning): 0 = heuristic, 128 = 128x128x64 4-wave kernel,
 * 256 = 256x256x64 8-wave 8-phase kernel.  Process-wide. */
int mmu_gemm_set_tile(int tile);
/* Tile rasterisation of the 256x256 kernel (tuning): GROUP_M tile rows swept column by column; 0 = default. */
int mmu_gemm_set_group_m(int group_m);
/* Which tiles the 8 XCDs work on at the same time (256x256 kernel): 1 (default) = adjacent 32-tile blocks of the raster
 * order in every round (shared A panels: one HBM read + Infinity-Cache hits), 0 = one contiguous chunk of the order per XCD
 * (rounds 1-2; A/B). */
int mmu_gemm_set_xcd_map(int mode);
/* 256x256 kernel, experiment switch: workgroups of XCD x (blockIdx.x & 7) start x * units * ~1 us late, so that the eight
 * XCDs' epilogue stores do not hit HBM at the same moment (each XCD's 32 CUs stay in lockstep: they share panels in L2).
 * 0 (default) = off. */
int mmu_gemm_set_xcd_skew(int units);
/* 256x256 kernel launch mode.  1 (default) = persistent: one workgroup per compute unit draws tiles from a device-side
 * queue, so a workgroup whose CU is held by another kernel (RCCL's all-reduce during the overlapped gradient exchange of
 * mm/forget.py:23-31's DDP counterpart) takes fewer tiles instead of delaying a fixed share; needs
 * mmu_gemm_set_tile_queue, without it mode 1 launches one workgroup per tile.  0 = one workgroup per tile.
 * 2 = persistent with a static tile walk (A/B tests: degrades badly when CUs are shared). */
int mmu_gemm_set_persistent(int mode);
/* Tile-queue counters of the persistent launch: device memory of >= mmu_gemm_tile_queue_bytes() (4 KiB), 16-B aligned;
 * the call zeroes it (memset + device synchronise: a set-up step) and the kernels leave it zero; NULL unregisters (launches
 * with more tiles than CUs then run one workgroup per tile).  The library never allocates. */
int mmu_gemm_set_tile_queue(void* counters, long long bytes);
/* Clears the registered tile-queue counters after a device synchronise: call it after a GEMM launch reported an error (or a
 * kernel fault was handled) so that a half-drawn queue slot cannot make later launches skip tiles.  No-op without a queue. */
int mmu_gemm_reset_tile_queue(void);
long long mmu_gemm_tile_queue_bytes(void);
/* Split-K tail of the 256x256 kernel: when the last (tiles % 256) tiles fill at most half a round of the 256 CUs, each
 * is cut into 256 / remainder K ranges on extra workgroups (fp32 partials in the workspace, reduced by a fold
 * kernel) instead of occupying a whole round.  mode 0 off, 1 heuristic (default), 2 whenever applicable.  The caller
 * registers a device workspace of >= mmu_gemm_workspace_bytes() (64 MiB); without one the tail split stays off. */
int mmu_gemm_set_tail_split(int mode);
int mmu_gemm_set_workspace(void* ws, long long bytes);
long long mmu_gemm_workspace_bytes(void);

/* ---- Attention (replaces the flash-attn wheel: mm/forget.py:86,98,106 attn_implementation=
 * "flash_attention_2"; SURVEY §8a K3 (CLIP, non-causal, head_dim 64) and K10 (Llama, causal, 128)) ----
 * q/k/v: rows of `ld` elements, head h at column h*head_dim of each pointer, sequence b = rows [b*S,(b+1)*S).
 * lse [B][H][S] fp32 (natural log of the scaled-score partition; may be NULL forward-only).
 * kmask [B][S] int32 (1 = key may be attended; NULL = all). */
int mmu_attn_fwd(const void* q, const void* k, const void* v, int ld, void* o, int ldo, float* lse,
                 const int* kmask, int B, int S, int H, int head_dim, float scale, int causal, void* stream);
/* mmu_attn_fwd that also leaves O^T [H * head_dim][ldt] (ldt >= B * S, multiple of 8; head_dim 128, S % 8 == 0): the
 * transposed activation of the o-projection's weight gradient (autograd's o_proj.weight.grad = dY^T O), written from the
 * staging image of the row-major store — no transpose pass over O in the backward. */
int mmu_attn_fwd_t(const void* q, const void* k, const void* v, int ld, void* o, int ldo, void* o_t, int ldt, float* lse,
                   const int* kmask, int B, int S, int H, int head_dim, float scale, int causal, void* stream);
/* head_dim 128 (Llama, causal) or 64 (CLIP tower, adapted by LoRA: non-causal); delta: fp32 workspace of 2 * B * H * S floats ([0]: rowsum(dO * O), [1]: lse * log2 e). */
int mmu_attn_bwd(const void* q, const void* k, const void* v, int ld, const void* o, int ldo, const float* lse,
                 const int* kmask, const void* dout, int lddo, float* delta, void* dq, void* dk, void* dv,
                 int ldg, int B, int S, int H, int head_dim, float scale, int causal, void* stream);

/* mmu_attn_bwd with the inverse RoPE rotation fused into the dq / dk stores (gradient w.r.t. the un-rotated q, k). */
int mmu_attn_bwd_rope(const void* q, const void* k, const void* v, int ld, const void* o, int ldo, const float* lse,
                      const int* kmask, const void* dout, int lddo, float* delta, void* dq, void* dk, void* dv, int ldg,
                      int B, int S, int H, int head_dim, float scale, int causal, const int* rope_pos,
                      const void* rope_cos, const void* rope_sin, void* stream);

/* mmu_attn_bwd_rope that additionally leaves [dq | dk | dv]^T = dqkv_t [3 * H * head_dim][ldt] (ldt >= B * S, a multiple of
 * 8; B * S a multiple of 8; dq / dk / dv 16-B aligned with ldg a multiple of 8): the A operand of the q|k|v weight
 * gradient (mm/trainer.py backward of q_proj / k_proj / v_proj) without a transpose pass.  For head_dim 128 and S a
 * multiple of 8 the dQ and the persistent dK/dV kernels write their parts themselves; otherwise (and for the eight-wave
 * dK/dV kernel) the part is a strided transpose of the finished block. */
int mmu_attn_bwd_rope_t(const void* q, const void* k, const void* v, int ld, const void* o, int ldo, const float* lse,
                        const int* kmask, const void* dout, int lddo, float* delta, void* dq, void* dk, void* dv, int ldg,
                        int B, int S, int H, int head_dim, float scale, int causal, const int* rope_pos,
                        const void* rope_cos, const void* rope_sin, void* dqkv_t, int ldt, void* stream);

/* dK/dV kernel of the head_dim-128 backward: low byte 8 = eight waves, K / V tiles in LDS; 4 = four waves (one per SIMD,
 * 512 registers), persistent, K / V fragments in registers, 4-stage Q/dO ring running across item seams.  Bits 8 and up:
 * workgroups of the four-wave kernel (0 = one per CU, else a multiple of 8).  Process-wide tuning / A-B switch. */
int mmu_attn_set_dkdv_variant(int variant);
/* Measurement hook, one-shot: the next mmu_attn_bwd* call records `event` (hipEvent_t, timing enabled) on its stream
 * between its dQ and its dK/dV kernel, so the two kernels of one call can be timed separately inside a step.  NULL cancels. */
int mmu_attn_set_bwd_mid_event(void* event);

/* ---- Norms (LlamaRMSNorm / CLIP LayerNorm; SURVEY §8a K2,K7) ---- */
int mmu_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int M, int d, float eps, void* stream);
int mmu_rmsnorm_bwd_nparts(int M); /* dw_part must hold nparts*d floats */
int mmu_rmsnorm_bwd(const void* dy, const void* x, const void* w, const float* rstd, const void* dres, void* dx,
                    float* dw_part, void* dw, int accumulate_dw, int M, int d, void* stream);
int mmu_layernorm_fwd(const void* x, const void* w, const void* b, void* y, int M, int d, float eps, void* stream);

/* ---- CLIP embeddings (CLIPVisionEmbeddings: patch conv as im2col+GEMM, cls/pos add; K1,K2) ---- */
/* Backward pieces of the CLIP tower, needed when its Linears carry LoRA adapters (peft targets every nn.Linear,
 * mm/forget.py:126-137): LayerNorm input gradient (+ residual gradient), quick-GELU forward / backward on the saved
 * pre-activation, and the head-dim padding copy [M][H][d_src] -> [M][H][d_dst] that lets head_dim-64 attention run
 * through the head_dim-128 backward kernels (zero padding leaves Q K^T and every gradient unchanged). */
int mmu_layernorm_bwd(const void* dy, const void* x, const void* w, const void* dres, void* dx, int M, int d, float eps,
                      void* stream);
/* LayerNorm parameter gradients for an UNfrozen CLIP tower (`freeze_vision_module: false`, mm/forget.py:117-118,
 * config/mm/forget.yaml:57): dgamma[j] (+)= sum_r dy[r][j] * xhat[r][j], dbeta[j] (+)= sum_r dy[r][j].  Workspace: stats
 * 2 * M floats, part 2 * mmu_colsum_nparts(M) * d floats. */
int mmu_layernorm_bwd_params(const void* dy, const void* x, float* stats, float* part, void* dgamma, void* dbeta,
                             int accumulate, int M, int d, float eps, void* stream);

/* peft's lora_dropout (mm/forget.py:128-136; nn.Dropout on the adapter's input): y[i] = x[i] / (1 - p) where a counter-based
 * hash of (seed, i) keeps the element, else 0.  The backward regenerates the mask from the seed (same call on the gradient);
 * in place allowed; n % 8 == 0.  The bit stream is the library's own, not torch's Philox stream (parity unpinned vs peft). */
int mmu_dropout_bf16(const void* x, void* y, long long n, float p, unsigned long long seed, void* stream);
int mmu_quick_gelu_fwd(const void* x, void* y, long long n, void* stream);
int mmu_quick_gelu_bwd(const void* x, const void* dy, void* dx, long long n, void* stream);
int mmu_head_pad(const void* src, int ld_src, int d_src, void* dst, int ld_dst, int d_dst, int M, int H, void* stream);
int mmu_im2col_patch(const float* pix, void* out, int B, int C, int H, int P, int Kpad, void* stream);
/* CLIP image rescale + normalise on the device (replaces the processor's numpy rescale/normalize reached from
 * mm/dataset.py:310-317 `processor(text=..., images=...)`): in_u8 [n_img][3][hw] uint8 (resized + centre-cropped on
 * the host) -> out fp32 (x/255 - mean[c]) / std[c].  mean3/std3: HOST pointers to 3 floats.  hw % 16 == 0. */
int mmu_clip_normalize_u8(const void* in_u8, float* out, long long n_img, int hw, const float* mean3, const float* std3,
                          void* stream);
int mmu_clip_embed(const void* patch, const void* cls, const void* pos, void* out, int B, int T, int d, void* stream);

/* ---- RoPE on the fused qkv buffer, rotate-half convention (apply_rotary_pos_emb; K9) ---- */
int mmu_rope_table(void* cos_t, void* sin_t, int n_pos, int head_dim, float theta, void* stream);
int mmu_rope(void* qkv, const int* pos_ids, const void* cos_t, const void* sin_t, int M, int ld, int n_q_heads,
             int n_k_heads, int head_dim, int k_off, int inverse, void* stream);

/* ---- SwiGLU / GELU / add (LlamaMLP, LlavaMultiModalProjector act; K5,K12) ---- */
int mmu_swiglu_fwd(const void* gate_up, void* h, int M, int I, void* stream);
int mmu_swiglu_bwd(const void* gate_up, const void* dh, void* dgate_up, int M, int I, void* stream);
int mmu_gelu_fwd(const void* x, void* y, long long n, void* stream);
int mmu_gelu_bwd(const void* x, const void* dy, void* dx, long long n, void* stream);
int mmu_add_bf16(const void* a, const void* b, void* out, long long n, void* stream);
/* dst[C][R] = src[R][C]^T: keeps W^T resident so dgrad runs on the NT GEMM path. */
int mmu_transpose_bf16(const void* src, void* dst, int R, int C, void* stream);
/* Fused producers of TRANSPOSED wgrad operands (the NT wgrad path reads X^T / dY^T): SwiGLU recompute -> act^T
 * [I][M]; RMSNorm recompute from the saved rstd -> y^T [d][M]; SwiGLU backward -> dgu [M][2I] and dgu^T [2I][M]. */
int mmu_swiglu_fwd_t(const void* gate_up, void* actT, int M, int I, void* stream);
/* Forward SwiGLU that writes act [M][I] and act^T [I][M] in one pass (act^T kept for the down-proj wgrad instead of
 * being recomputed in backward: +M*I*2 B of HBM per layer, one read of gate|up less). */
int mmu_swiglu_fwd_both(const void* gate_up, void* act, void* actT, int M, int I, void* stream);
int mmu_rmsnorm_apply_t(const void* x, const void* w, const float* rstd, void* yT, int M, int d, void* stream);
int mmu_swiglu_bwd_t(const void* gate_up, const void* dh, void* dgate_up, void* dgate_upT, int M, int I, void* stream);
int mmu_colsum_nparts(int M); /* part must hold nparts*d floats */
int mmu_colsum_bf16(const void* x, float* part, void* dst, int accumulate, int M, int d, void* stream);

/* ---- Token embedding + image-feature scatter (modeling_llava.py masked_scatter; K6) ---- */
int mmu_image_slots(const long long* ids, int M, long long image_token, int* slot, int* count, void* stream);
int mmu_embed_fwd(const long long* ids, const int* slot, const void* table, const void* img, void* out, int M,
                  int d, int vocab, void* stream);
int mmu_embed_bwd(const long long* ids, const int* slot, const void* dh, void* dimg, float* scratch, int* flag,
                  void* grad_table, int M, int d, int vocab, void* stream);

/* ---- Losses over [N,V] bf16 logits (mm/trainer_utils.py:16-21 logits2probs, :73-81 get_batch_loss;
 * F.kl_div call sites mm/trainer.py:203-208,220-225,237-242,250-255,271-276,294-299,435-440; NPO :348-353;
 * entropy :195-196; the model's own shifted CE = `outputs.loss`; SURVEY §8a R3-R7,K14) ----
 * labels are the UNSHIFTED [B][S] int64 labels; shift=1 applies the causal-LM shift (target of row (b,s) is
 * labels[b][s+1], last position ignored); ignore_index = -100. */
int mmu_ce_fwd(const void* logits, int ldv, const long long* labels, float* lse, float* loss_row, int M, int V,
               int S, int shift, void* stream);
int mmu_ce_bwd(const void* logits, int ldv, const long long* labels, const float* lse, const float* seg_scale,
               const float* seg_cnt, int seg_rows, int normalize, void* dlogits, int ldd, int M, int V, int S,
               int shift, void* stream);
int mmu_seg_reduce(const float* vals, const long long* labels, int seg_rows, int M, int V, int S, int shift,
                   float* out_sum, float* out_cnt, void* stream);
/* mode 0: KL(input=student,target=teacher) 1: KL(input=teacher,target=student) 2: NPO 3: entropy sum p log p
 * 4: nlp-trainer NPO form sum log(1 + (lp_s - lp_t)^beta) (nlp/dataloader.py:396-401).
 * stats [M][4] = {lse_s, lse_t, row value, aux}. */
int mmu_rowloss_fwd(int mode, const void* slog, const void* tlog, int ldv, float beta, float* stats, int M, int V,
                    void* stream);
int mmu_rowloss_bwd(int mode, const void* slog, const void* tlog, int ldv, float beta, const float* stats,
                    const float* coef, void* dlogits, int ldd, int M, int V, void* stream);

/* remove_image_tokens (mm/trainer_utils.py:48-70): dst[r][:] = src[rows[r]][:] for n_dst_rows rows of V bf16 (V % 8 == 0);
 * rows: device int64, flat source row indices computed on the device by the caller (no host loop, no sync). */
int mmu_gather_rows_bf16(const void* src, const long long* rows, void* dst, long long n_src_rows, int n_dst_rows, int V,
                         void* stream);
/* Supervised-row selection in front of the lm_head (K13/K14: HF's shifted CE ignores every position whose next label is
 * -100, mm/dataset.py:325-339 masks image + question tokens, so only the selected hidden-state rows reach the head):
 * mmu_gather_rows_bf16 with rows[r] < 0 writes a zero row (padding slot); mmu_scatter_rows_bf16 is the backward,
 * dst[rows[r]][:] = src[r][:] for rows[r] >= 0 (distinct rows; the caller zeroes dst first). */
int mmu_scatter_rows_bf16(const void* src, const long long* rows, void* dst, long long n_dst_rows, int n_src_rows, int V,
                          void* stream);

/* Evaluation truth-ratio pass (mm/eval.py:46-92 eval_perturbation_ratio: get_batch_loss + token counts) for n_seq
 * sequences of S rows in one call: out[seq] = {sum_t shifted CE, #shifted targets != -100, #labels != -100 (unshifted:
 * the normaliser of mm/eval.py:70-71)}.  lse_scratch / row_scratch: n_seq * S floats each. */
int mmu_seq_ce_stats(const void* logits, int ldv, const long long* labels, int n_seq, int S, int V, float* lse_scratch,
                     float* row_scratch, float* out, void* stream);

/* logits2probs (mm/trainer_utils.py:16-21) for callers that use the helper by name: out [M,V] fp32 = log_softmax(logits)
 * (log_ != 0) or softmax(logits) over V, lse [M] kept for the backward; mmu_logprobs_bwd turns an upstream gradient
 * g [M,V] fp32 into dlogits bf16.  (The loss branches of mm/trainer.py never call these: mmu_rowloss_* read the logits
 * once and materialise nothing.) */
int mmu_logprobs_fwd(const void* logits, int ldv, float* out, float* lse, int M, int V, int log_, void* stream);
int mmu_logprobs_bwd(const void* logits, int ldv, const float* lse, const float* g, void* dlogits, int ldd, int M, int V,
                     int log_, void* stream);

/* 1 forces the streaming row-loss kernel (student / teacher rows walked twice); default 0: rows of V <= 32768 logits are
 * held in registers and cross HBM once (same arithmetic, different summation order).  A/B and test switch. */
int mmu_rowloss_set_streaming(int on);

/* ---- RMU terms (mm/trainer.py:302-326: mse_loss(mean_seq(act_f), control) * beta + mse_loss(act_r, act_r_teacher);
 * nlp/dataloader.py:339-367: mse_loss(act_f, control broadcast)) on the captured decoder-layer activation
 * [B, S, H] bf16.  Deterministic two-stage reductions; `out` buffers hold 1 + mmu_red_scratch_floats() floats, the
 * result in slot 0.
 *   mmu_mse_fwd       out[0] = sum_i (a_i - ref_i)^2, ref = b_i (bf16 [n]) or, when b == NULL, cvec[i % H] (fp32 [H])
 *   mmu_mse_bwd       da_i = *coef * (a_i - ref_i)            (the caller folds 2 / n and the upstream gradient into coef)
 *   mmu_seqmean_fwd   mean[b,h] = (1/S) sum_s a[b,s,h];  partial = mmu_seqmean_nsplit(S) * B * H floats of scratch
 *   mmu_rmu_pair_fwd  out[0] = sum_{i<B, j<Bc, h} (mean[i,h] - ctrl[j,h])^2   ([B,H] vs [Bc,1,H] broadcasts to [Bc,B,H])
 *   mmu_rmu_pair_bwd  da[i,s,h] = *coef / S * sum_j (mean[i,h] - ctrl[j,h])   (bf16 [B,S,H]) */
int mmu_red_scratch_floats(void);
int mmu_mse_fwd(const void* a, const void* b, const float* cvec, long long n, int H, float* out, void* stream);
int mmu_mse_bwd(const void* a, const void* b, const float* cvec, long long n, int H, const float* coef, void* da,
                void* stream);
int mmu_seqmean_nsplit(int S);
int mmu_seqmean_fwd(const void* a, int B, int S, int H, float* partial, float* mean, void* stream);
int mmu_rmu_pair_fwd(const float* mean, const float* ctrl, int B, int Bc, int H, float* out, void* stream);
int mmu_rmu_pair_bwd(const float* mean, const float* ctrl, int B, int Bc, int S, int H, const float* coef, void* da,
                     void* stream);

/* ---- L1 / L0 regulariser of compute_loss (mm/trainer.py:451-481: torch.norm(cat(param - teacher_param), p=1 | 0)) over
 * the flat trainable buffer: d = bf16(p - t) (t = NULL: d = p); out2[0] += sum|d|, out2[1] += #(d != 0) in fp64.
 * Backward: g += lambda1 * *gout * sign(d) (gout: device scalar = upstream gradient, NULL = 1; L0 has no gradient). */
int mmu_l1l0_fwd(const void* p, const void* t, long long n, double* out2, void* stream);
int mmu_l1_bwd(const void* p, const void* t, long long n, void* g, float lambda1, const float* gout, void* stream);

/* ---- Optimiser step after compute_loss (HF loop: clip_grad_norm_ -> AdamW -> zero_grad; mm/forget.py:139-168) ---- */
/* out_sumsq[0] <- sum g^2, deterministic (fixed summation order: every DP rank must derive the same clip factor from
 * the same all-reduced gradient); out_sumsq needs 1 + mmu_sumsq_scratch_floats() floats (partials after slot 0). */
int mmu_sumsq_bf16(const void* g, long long n, float* out_sumsq, void* stream);
int mmu_sumsq_scratch_floats(void);
/* nodecay: device array of n_nodecay (<= 512) sorted, disjoint element ranges [lo, hi) of p (multiples of 8) that get no
 * weight decay — transformers.Trainer.create_optimizer decays neither biases nor norm weights — or NULL / 0. */
int mmu_adamw_bf16(void* p, void* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                   float weight_decay, int step, const float* sumsq, float max_norm, float grad_scale, int zero_grad,
                   const long long* nodecay, int n_nodecay, void* stream);
/* AdamW with block-wise 8-bit moments (the reference's optim="adamw_bnb_8bit", mm/forget.py:156; bitsandbytes is not in
 * /root/reference: published algorithm restated, parity unpinned): blocks of 2048 parameters, 8-bit codes of a
 * 256-entry dynamic quantisation map (map1 signed for m, map2 unsigned for v; device pointers) times an fp32 absmax
 * per block.  codes*: n bytes each; absmax*: mmu_adamw8bit_nblocks(n) floats each.  Fresh state: codes = index of
 * 0.0 in the map, absmax = 0.  Same clip / grad_scale / zero_grad semantics as mmu_adamw_bf16. */
int mmu_adamw8bit_nblocks(long long n);
int mmu_adamw8bit_bf16(void* p, void* g, void* codes1, void* codes2, float* absmax1, float* absmax2, const float* map1,
                       const float* map2, long long n, float lr, float beta1, float beta2, float eps, float weight_decay,
                       int step, const float* sumsq, float max_norm, float grad_scale, int zero_grad,
                       const long long* nodecay, int n_nodecay, void* stream);
int mmu_scale_bf16(void* x, long long n, float s, void* stream);

#ifdef __cplusplus
}
#endif
#endif
