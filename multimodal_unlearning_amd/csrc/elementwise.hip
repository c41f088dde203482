// HBM-bound kernels of the LLaVA forward/backward for gfx950: RMSNorm, LayerNorm, RoPE, SwiGLU, GELU,
// token-embedding gather / image-feature scatter (+ backward), CLIP patch im2col and embedding assembly.
// These stand in for the elementwise ATen ops under transformers' LlavaForConditionalGeneration that
// mm/trainer.py:180-433 `model(**inputs)` and its backward execute (SURVEY.md §8a rows K2,K4,K6,K7,K9,K12).
// All loads/stores are 16-byte vectors (8 bf16 per lane); reductions use wave shuffles.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// RMSNorm (modeling_llama.py:53-67 in transformers): fp32 internals, y = w * bf16(x * rstd)
// One wave per row; the row is held in registers (MAXCH chunks of 8 per lane -> d <= 512*MAXCH).
// ------------------------------------------------------------------------------------------------
template <int MAXCH>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w,
                                                          bf16* __restrict__ y, float* __restrict__ rstd_out, int M,
                                                          int d, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nch = d >> 3;
    const bf16* xr = x + (size_t)row * d;
    float v[MAXCH][8];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < MAXCH; ++c) {
        const int ch = c * 64 + lane;
        if (ch < nch) {
            const bf16x8 t = *reinterpret_cast<const bf16x8*>(xr + ch * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[c][j] = bf2f(t[j]);
                ss += v[c][j] * v[c][j];
            }
        }
    }
    ss = wave_sum(ss);
    const float rstd = rsqrtf(ss / (float)d + eps);
    if (lane == 0 && rstd_out) rstd_out[row] = rstd;
    bf16* yr = y + (size_t)row * d;
#pragma unroll
    for (int c = 0; c < MAXCH; ++c) {
        const int ch = c * 64 + lane;
        if (ch < nch) {
            const bf16x8 wv = *reinterpret_cast<const bf16x8*>(w + ch * 8);
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = f2bf(bf2f(wv[j]) * round_bf16((v[c][j] * rstd)));
            *reinterpret_cast<bf16x8*>(yr + ch * 8) = o;
        }
    }
}

// Backward: g = dy*w ; xh = x*rstd ; dx = rstd*(g - xh*mean(g*xh)) (+ dres) ; dw_partial[blk] += dy*bf16(xh).
// One 256-thread BLOCK per row (grid-stride over rows): a thread owns NCH fixed 8-column chunks, so the row fits in
// ~70 registers (8 waves/SIMD hide the HBM latency), the row dot product is one block reduction, and the weight-
// gradient partial sums stay in the owning thread's registers for the whole sweep: one fp32 partial row per block
// goes to `dw_part` [gridDim.x][d] with no LDS reduction.
template <int NCH>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x,
                                                          const bf16* __restrict__ w, const float* __restrict__ rstd,
                                                          const bf16* __restrict__ dres, bf16* __restrict__ dx,
                                                          float* __restrict__ dw_part, int M, int d) {
    __shared__ float scratch[4];
    const int nch = d >> 3;
    float dwacc[NCH][8], wv[NCH][8];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int ch = c * 256 + threadIdx.x;
#pragma unroll
        for (int j = 0; j < 8; ++j) { dwacc[c][j] = 0.f; wv[c][j] = 0.f; }
        if (ch < nch) {
            const bf16x8 t = *reinterpret_cast<const bf16x8*>(w + ch * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) wv[c][j] = bf2f(t[j]);
        }
    }
    for (int row = blockIdx.x; row < M; row += gridDim.x) {
        const float rs = rstd[row];
        const size_t off = (size_t)row * d;
        float g[NCH][8], xh[NCH][8];
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int ch = c * 256 + threadIdx.x;
            if (ch < nch) {
                const bf16x8 tdy = *reinterpret_cast<const bf16x8*>(dy + off + ch * 8);
                const bf16x8 tx = *reinterpret_cast<const bf16x8*>(x + off + ch * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float dyv = bf2f(tdy[j]);
                    xh[c][j] = bf2f(tx[j]) * rs;
                    g[c][j] = dyv * wv[c][j];
                    dot += g[c][j] * xh[c][j];
                    dwacc[c][j] += dyv * round_bf16(xh[c][j]);
                }
            }
        }
        dot = block_sum<4>(dot, scratch) / (float)d;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int ch = c * 256 + threadIdx.x;
            if (ch < nch) {
                bf16x8 o;
                if (dres) {
                    const bf16x8 r = *reinterpret_cast<const bf16x8*>(dres + off + ch * 8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = f2bf(rs * (g[c][j] - xh[c][j] * dot) + bf2f(r[j]));
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = f2bf(rs * (g[c][j] - xh[c][j] * dot));
                }
                *reinterpret_cast<bf16x8*>(dx + off + ch * 8) = o;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int ch = c * 256 + threadIdx.x;
        if (ch < nch) {
            float* dst = dw_part + (size_t)blockIdx.x * d + ch * 8;
            *reinterpret_cast<f32x4*>(dst) = f32x4{dwacc[c][0], dwacc[c][1], dwacc[c][2], dwacc[c][3]};
            *reinterpret_cast<f32x4*>(dst + 4) = f32x4{dwacc[c][4], dwacc[c][5], dwacc[c][6], dwacc[c][7]};
        }
    }
}

// dst[j] (+)= sum_p part[p][j]  (column reduce of the per-block partials; bf16 destination).
// Block = 32 columns x 8 row groups (128-B coalesced segments), LDS reduce of the 8 groups: d/32 blocks with
// nparts/8 independent loads per thread (the one-thread-per-column version was a 1024-deep serial chain: 246 us).
__global__ __launch_bounds__(256) void colsum_partials_kernel(const float* __restrict__ part, int nparts, int d,
                                                              bf16* __restrict__ dst, int accumulate) {
    __shared__ float red[8][32];
    const int c = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int j = blockIdx.x * 32 + c;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (j < d) {
        int p = grp;
        for (; p + 24 < nparts; p += 32) {
            s0 += part[(size_t)p * d + j];
            s1 += part[(size_t)(p + 8) * d + j];
            s2 += part[(size_t)(p + 16) * d + j];
            s3 += part[(size_t)(p + 24) * d + j];
        }
        for (; p < nparts; p += 8) s0 += part[(size_t)p * d + j];
    }
    red[grp][c] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (grp == 0 && j < d) {
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) s += red[g][c];
        if (accumulate) s += bf2f(dst[j]);
        dst[j] = f2bf(s);
    }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm forward (CLIP, frozen): fp32 statistics, one rounding at the output. One wave per row.
// Optional fused "CLIP embeddings" producer: row = (cls | patch) + position embedding.
// ------------------------------------------------------------------------------------------------
template <int MAXCH>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w,
                                                            const bf16* __restrict__ b, bf16* __restrict__ y, int M,
                                                            int d, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nch = d >> 3;
    const bf16* xr = x + (size_t)row * d;
    float v[MAXCH][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < MAXCH; ++c) {
        const int ch = c * 64 + lane;
        if (ch < nch) {
            const bf16x8 t = *reinterpret_cast<const bf16x8*>(xr + ch * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[c][j] = bf2f(t[j]);
                s += v[c][j];
            }
        }
    }
    const float mean = wave_sum(s) / (float)d;
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < MAXCH; ++c) {
        const int ch = c * 64 + lane;
        if (ch < nch) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float t = v[c][j] - mean;
                ss += t * t;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(ss) / (float)d + eps);
    bf16* yr = y + (size_t)row * d;
#pragma unroll
    for (int c = 0; c < MAXCH; ++c) {
        const int ch = c * 64 + lane;
        if (ch < nch) {
            const bf16x8 wv = *reinterpret_cast<const bf16x8*>(w + ch * 8);
            const bf16x8 bv = *reinterpret_cast<const bf16x8*>(b + ch * 8);
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = f2bf((v[c][j] - mean) * rstd * bf2f(wv[j]) + bf2f(bv[j]));
            *reinterpret_cast<bf16x8*>(yr + ch * 8) = o;
        }
    }
}

// CLIP embeddings (modeling_clip.py CLIPVisionEmbeddings): out[b,0]=cls+pos[0]; out[b,1+p]=patch[b,p]+pos[1+p]
__global__ void clip_embed_kernel(const bf16* __restrict__ patch, const bf16* __restrict__ cls,
                                  const bf16* __restrict__ pos, bf16* __restrict__ out, int B, int T, int d) {
    const int nch = d >> 3;
    const size_t total = (size_t)B * T * nch;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % nch);
        const size_t row = i / nch;
        const int t = (int)(row % T);
        const int b = (int)(row / T);
        const bf16x8 pv = *reinterpret_cast<const bf16x8*>(pos + (size_t)t * d + ch * 8);
        bf16x8 src;
        if (t == 0)
            src = *reinterpret_cast<const bf16x8*>(cls + ch * 8);
        else
            src = *reinterpret_cast<const bf16x8*>(patch + ((size_t)b * (T - 1) + (t - 1)) * d + ch * 8);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = f2bf(bf2f(src[j]) + bf2f(pv[j]));
        *reinterpret_cast<bf16x8*>(out + row * d + ch * 8) = o;
    }
}

// im2col for the stride=P, kernel=P patch conv: out[b*np + py*g + px][c*P*P + iy*P + ix] (zero-padded to Kpad)
__global__ void im2col_patch_kernel(const float* __restrict__ pix, bf16* __restrict__ out, int B, int C, int H,
                                    int P, int Kpad) {
    const int g = H / P;
    const int K = C * P * P;
    const size_t total = (size_t)B * g * g * Kpad;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % Kpad);
        const size_t row = i / Kpad;
        float v = 0.f;
        if (k < K) {
            const int ix = k % P, iy = (k / P) % P, c = k / (P * P);
            const int px = (int)(row % g), py = (int)((row / g) % g), b = (int)(row / ((size_t)g * g));
            v = pix[(((size_t)b * C + c) * H + (py * P + iy)) * H + (px * P + ix)];
        }
        out[i] = f2bf(v);
    }
}

// CLIP pixel normalisation on the device: uint8 [n][3][hw] -> fp32 (x/255 - mean[c]) / std[c], with the rounding
// points of transformers' numpy path (double product rounded to fp32, then fp32 subtract and true divide), so the
// host ships 1 byte per pixel over PCIe instead of 4.  16 pixels per thread (16-B load, 4 x 16-B stores).
__global__ __launch_bounds__(256) void clip_normalize_u8_kernel(const unsigned char* __restrict__ in,
                                                                float* __restrict__ out, size_t n16, int hw16,
                                                                float m0, float m1, float m2, float s0, float s1,
                                                                float s2) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)((i / hw16) % 3);
        const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2);
        const float sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
        const uint4 raw = *reinterpret_cast<const uint4*>(in + i * 16);
        const unsigned int wds[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float r = (float)((double)((wds[q] >> (8 * j)) & 0xffu) * (1.0 / 255.0));
                o[j] = (r - mean) / sd;
            }
            *reinterpret_cast<f32x4*>(out + i * 16 + q * 4) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// RoPE on the fused qkv buffer [M][ld] (q heads at col 0, k heads at col k_off), rotate-half convention
// (modeling_llama.py rotate_half / apply_rotary_pos_emb): bf16 products, bf16 sum, cos/sin tables in bf16.
// inverse=1 applies the transpose rotation (backward).
// ------------------------------------------------------------------------------------------------
__global__ void rope_kernel(bf16* __restrict__ qkv, const int* __restrict__ pos_ids, const bf16* __restrict__ cos_t,
                            const bf16* __restrict__ sin_t, int M, int ld, int n_heads_total, int head_dim,
                            int k_off, int n_q_heads, int inverse) {
    const int half = head_dim >> 1;
    const int cph = half >> 3;  // 16-B chunks per half head
    const size_t total = (size_t)M * n_heads_total * cph;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % cph);
        const int h = (int)((i / cph) % n_heads_total);
        const size_t row = i / ((size_t)cph * n_heads_total);
        const int pos = pos_ids[row];
        const int col = (h < n_q_heads ? h * head_dim : k_off + (h - n_q_heads) * head_dim) + ch * 8;
        bf16* p1 = qkv + row * ld + col;
        bf16* p2 = p1 + half;
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(p1);
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(p2);
        const bf16x8 c = *reinterpret_cast<const bf16x8*>(cos_t + (size_t)pos * half + ch * 8);
        const bf16x8 s = *reinterpret_cast<const bf16x8*>(sin_t + (size_t)pos * half + ch * 8);
        bf16x8 o1, o2;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float af = bf2f(a[j]), bfv = bf2f(b[j]), cf = bf2f(c[j]);
            const float sf = inverse ? -bf2f(s[j]) : bf2f(s[j]);
            o1[j] = f2bf(round_bf16((af * cf)) + round_bf16((-bfv * sf)));
            o2[j] = f2bf(round_bf16((bfv * cf)) + round_bf16((af * sf)));
        }
        *reinterpret_cast<bf16x8*>(p1) = o1;
        *reinterpret_cast<bf16x8*>(p2) = o2;
    }
}

// cos/sin tables [n_pos][head_dim/2] in bf16 (LlamaRotaryEmbedding: fp32 angle, then cast)
__global__ void rope_table_kernel(bf16* __restrict__ cos_t, bf16* __restrict__ sin_t, int n_pos, int half,
                                  float theta) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pos * half) return;
    const int p = i / half, j = i % half;
    const float inv_freq = 1.0f / powf(theta, (float)(2 * j) / (float)(2 * half));
    const float ang = (float)p * inv_freq;
    cos_t[i] = f2bf(cosf(ang));
    sin_t[i] = f2bf(sinf(ang));
}

// ------------------------------------------------------------------------------------------------
// SwiGLU: gu [M][2I] (gate | up)  ->  h = bf16(silu(gate)) * up   (modeling_llama.py LlamaMLP)
// ------------------------------------------------------------------------------------------------
__global__ void swiglu_fwd_kernel(const bf16* __restrict__ gu, bf16* __restrict__ h, int M, int I) {
    const int nch = I >> 3;
    const size_t total = (size_t)M * nch;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % nch);
        const size_t row = i / nch;
        const bf16x8 g = *reinterpret_cast<const bf16x8*>(gu + row * 2 * I + ch * 8);
        const bf16x8 u = *reinterpret_cast<const bf16x8*>(gu + row * 2 * I + I + ch * 8);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float gf = bf2f(g[j]);
            const float act = gf / (1.0f + __expf(-gf));
            o[j] = f2bf(round_bf16((act)) * bf2f(u[j]));
        }
        *reinterpret_cast<bf16x8*>(h + row * I + ch * 8) = o;
    }
}

__global__ void swiglu_bwd_kernel(const bf16* __restrict__ gu, const bf16* __restrict__ dh, bf16* __restrict__ dgu,
                                  int M, int I) {
    const int nch = I >> 3;
    const size_t total = (size_t)M * nch;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % nch);
        const size_t row = i / nch;
        const bf16x8 g = *reinterpret_cast<const bf16x8*>(gu + row * 2 * I + ch * 8);
        const bf16x8 u = *reinterpret_cast<const bf16x8*>(gu + row * 2 * I + I + ch * 8);
        const bf16x8 d = *reinterpret_cast<const bf16x8*>(dh + row * I + ch * 8);
        bf16x8 og, ou;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float gf = bf2f(g[j]), uf = bf2f(u[j]), df = bf2f(d[j]);
            const float sig = 1.0f / (1.0f + __expf(-gf));
            const float act = gf * sig;
            ou[j] = f2bf(df * round_bf16((act)));
            og[j] = f2bf(df * uf * (sig * (1.0f + gf * (1.0f - sig))));
        }
        *reinterpret_cast<bf16x8*>(dgu + row * 2 * I + ch * 8) = og;
        *reinterpret_cast<bf16x8*>(dgu + row * 2 * I + I + ch * 8) = ou;
    }
}

// LayerNorm backward, input gradient only (the CLIP tower's LayerNorm weights stay frozen; its Linears may carry LoRA
// adapters: mm/forget.py:126-137 targets every nn.Linear).  One wave per row, statistics recomputed from x:
//   xh = (x - mean) rstd, g = dy w, dx = rstd (g - mean(g) - xh mean(g xh)) (+ dres).
template <int MAXCH>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x,
                                                            const bf16* __restrict__ w, const bf16* __restrict__ dres,
                                                            bf16* __restrict__ dx, int M, int d, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nch = d >> 3;
    const size_t off = (size_t)row * d;
    float v[MAXCH][8], g[MAXCH][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < MAXCH; ++c) {
        const int ch = c * 64 + lane;
#pragma unroll
        for (int j = 0; j < 8; ++j) { v[c][j] = 0.f; g[c][j] = 0.f; }
        if (ch < nch) {
            const bf16x8 t = *reinterpret_cast<const bf16x8*>(x + off + ch * 8);
            const bf16x8 gy = *reinterpret_cast<const bf16x8*>(dy + off + ch * 8);
            const bf16x8 wv = *reinterpret_cast<const bf16x8*>(w + ch * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[c][j] = bf2f(t[j]);
                g[c][j] = bf2f(gy[j]) * bf2f(wv[j]);
                s += v[c][j];
            }
        }
    }
    const float mean = wave_sum(s) / (float)d;
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < MAXCH; ++c) {
        const int ch = c * 64 + lane;
        if (ch < nch) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[c][j] -= mean;
                ss += v[c][j] * v[c][j];
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(ss) / (float)d + eps);
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int c = 0; c < MAXCH; ++c) {
        const int ch = c * 64 + lane;
        if (ch < nch) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[c][j] *= rstd;  // xh
                sg += g[c][j];
                sgx += g[c][j] * v[c][j];
            }
        }
    }
    const float mg = wave_sum(sg) / (float)d, mgx = wave_sum(sgx) / (float)d;
#pragma unroll
    for (int c = 0; c < MAXCH; ++c) {
        const int ch = c * 64 + lane;
        if (ch < nch) {
            bf16x8 o;
            if (dres) {
                const bf16x8 r = *reinterpret_cast<const bf16x8*>(dres + off + ch * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = f2bf(rstd * (g[c][j] - mg - v[c][j] * mgx) + bf2f(r[j]));
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = f2bf(rstd * (g[c][j] - mg - v[c][j] * mgx));
            }
            *reinterpret_cast<bf16x8*>(dx + off + ch * 8) = o;
        }
    }
}

// quick-GELU x * sigmoid(1.702 x) (CLIP MLP act) forward / backward on the bf16 pre-activation
__global__ void quick_gelu_fwd_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + i * 8);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float f = bf2f(v[j]);
            o[j] = f2bf(f / (1.0f + __expf(-1.702f * f)));
        }
        *reinterpret_cast<bf16x8*>(y + i * 8) = o;
    }
}
__global__ void quick_gelu_bwd_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy, bf16* __restrict__ dx,
                                      size_t n8) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + i * 8);
        const bf16x8 g = *reinterpret_cast<const bf16x8*>(dy + i * 8);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float f = bf2f(v[j]);
            const float sg = 1.0f / (1.0f + __expf(-1.702f * f));
            o[j] = f2bf(bf2f(g[j]) * sg * (1.0f + 1.702f * f * (1.0f - sg)));
        }
        *reinterpret_cast<bf16x8*>(dx + i * 8) = o;
    }
}

// Head-dim padding copies: dst [M][H][dd] <- src [M][H][ds] (row strides ld*, elements), zero-filling dd > ds or
// dropping the tail when dd < ds.  Used to run head_dim-64 attention (CLIP) through the head_dim-128 backward kernels.
__global__ void head_pad_kernel(const bf16* __restrict__ src, int lds_, int ds, bf16* __restrict__ dst, int ldd, int dd,
                                int M, int H) {
    const int cd = dd >> 3, cs = ds >> 3;
    const size_t total = (size_t)M * H * cd;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cd);
        const int h = (int)((i / cd) % H);
        const size_t m = i / ((size_t)cd * H);
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = f2bf(0.f);
        if (c < cs) v = *reinterpret_cast<const bf16x8*>(src + m * lds_ + h * ds + c * 8);
        *reinterpret_cast<bf16x8*>(dst + m * ldd + h * dd + c * 8) = v;
    }
}

// GELU (erf) forward / backward (LlavaMultiModalProjector act)
__global__ void gelu_fwd_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + i * 8);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float f = bf2f(v[j]);
            o[j] = f2bf(0.5f * f * (1.0f + erff(f * 0.70710678118654752f)));
        }
        *reinterpret_cast<bf16x8*>(y + i * 8) = o;
    }
}
__global__ void gelu_bwd_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy, bf16* __restrict__ dx,
                                size_t n8) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + i * 8);
        const bf16x8 g = *reinterpret_cast<const bf16x8*>(dy + i * 8);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float f = bf2f(v[j]);
            const float cdf = 0.5f * (1.0f + erff(f * 0.70710678118654752f));
            const float pdf = 0.3989422804014327f * __expf(-0.5f * f * f);
            o[j] = f2bf(bf2f(g[j]) * (cdf + f * pdf));
        }
        *reinterpret_cast<bf16x8*>(dx + i * 8) = o;
    }
}

// out = a + b (bf16, 8-wide)
__global__ void add_kernel(const bf16* __restrict__ a, const bf16* __restrict__ b, bf16* __restrict__ o, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const bf16x8 x = *reinterpret_cast<const bf16x8*>(a + i * 8);
        const bf16x8 y = *reinterpret_cast<const bf16x8*>(b + i * 8);
        bf16x8 r;
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = f2bf(bf2f(x[j]) + bf2f(y[j]));
        *reinterpret_cast<bf16x8*>(o + i * 8) = r;
    }
}

// LayerNorm parameter gradients (unfrozen CLIP tower, mm/forget.py:117-118 `freeze_vision_module: false`):
//   dgamma[j] = sum_r dy[r][j] * xhat[r][j],  dbeta[j] = sum_r dy[r][j],  xhat = (x - mean_r) * rstd_r.
// Stage 1: one wave per row leaves {mean, rstd}; stage 2: one thread per column walks a block of rows (coalesced across the
// threads of a row) and leaves fp32 partials [2][nparts][d]; colsum_partials_kernel folds them into the bf16 gradients.
__global__ __launch_bounds__(256) void layernorm_stats_kernel(const bf16* __restrict__ x, float* __restrict__ stats, int M,
                                                              int d, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nch = d >> 3;
    const bf16* xr = x + (size_t)row * d;
    float s = 0.f;
    for (int ch = lane; ch < nch; ch += 64) {
        const bf16x8 t = *reinterpret_cast<const bf16x8*>(xr + ch * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += bf2f(t[j]);
    }
    const float mean = wave_sum(s) / (float)d;
    float ss = 0.f;
    for (int ch = lane; ch < nch; ch += 64) {
        const bf16x8 t = *reinterpret_cast<const bf16x8*>(xr + ch * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float c = bf2f(t[j]) - mean;
            ss += c * c;
        }
    }
    const float rstd = rsqrtf(wave_sum(ss) / (float)d + eps);
    if (lane == 0) {
        stats[2 * (size_t)row] = mean;
        stats[2 * (size_t)row + 1] = rstd;
    }
}

__global__ void layernorm_dwdb_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x, const float* __restrict__ stats,
                                      float* __restrict__ part, int M, int d, int rows_per_block, int nparts) {
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = min(M, r0 + rows_per_block);
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= d) return;
    float sw = 0.f, sb = 0.f;
    for (int r = r0; r < r1; ++r) {
        const float g = bf2f(dy[(size_t)r * d + j]);
        sw += g * (bf2f(x[(size_t)r * d + j]) - stats[2 * (size_t)r]) * stats[2 * (size_t)r + 1];
        sb += g;
    }
    part[(size_t)blockIdx.y * d + j] = sw;
    part[((size_t)nparts + blockIdx.y) * d + j] = sb;
}

// Dropout (peft's `lora_dropout`, mm/forget.py:128-136: nn.Dropout on the ADAPTER's input, y = W x + (alpha/r) B A drop(x)):
// y[i] = keep(i) ? x[i] / (1 - p) : 0 with keep(i) a counter-based hash of (seed, i), so the backward pass regenerates the
// same mask from the seed instead of storing it (the same kernel masks the gradient).  In place (y == x) is allowed.
// The bit stream is this library's own (a 64-bit mix of seed and element index), not torch's Philox stream.
__device__ __forceinline__ unsigned dropout_bits(unsigned long long seed, unsigned long long i) {
    unsigned long long z = seed + i * 0x9E3779B97F4A7C15ull;   // splitmix64 of the (seed, index) pair
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (unsigned)(z >> 40);  // 24 random bits
}
__global__ void dropout_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, size_t nch, unsigned thresh, float scale,
                               unsigned long long seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nch; i += (size_t)gridDim.x * blockDim.x) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + i * 8);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = f2bf(dropout_bits(seed, i * 8 + j) >= thresh ? bf2f(v[j]) * scale : 0.f);
        *reinterpret_cast<bf16x8*>(y + i * 8) = o;
    }
}

// column-sum of a bf16 matrix [M][d] into fp32 partials (bias gradients): block handles a row range.
__global__ void colsum_bf16_kernel(const bf16* __restrict__ x, float* __restrict__ part, int M, int d,
                                   int rows_per_block) {
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = min(M, r0 + rows_per_block);
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= d) return;
    float s = 0.f;
    for (int r = r0; r < r1; ++r) s += bf2f(x[(size_t)r * d + j]);
    part[(size_t)blockIdx.y * d + j] = s;
}

// ------------------------------------------------------------------------------------------------
// Token embedding gather + image-feature scatter (modeling_llava.py:233-248 masked_scatter):
// slot[r] = index of the image feature row for position r, or -1 for a text token.
// ------------------------------------------------------------------------------------------------
__global__ void image_slots_kernel(const long long* __restrict__ ids, int M, long long image_token,
                                   int* __restrict__ slot, int* __restrict__ count) {
    // single block; sequential chunked scan (M is a few thousand to ~1e5)
    __shared__ int base;
    __shared__ int wsum[16];
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int start = 0; start < M; start += blockDim.x) {
        const int r = start + threadIdx.x;
        const int is_img = (r < M && ids[r] == image_token) ? 1 : 0;
        const unsigned long long bal = __ballot(is_img);
        const int prefix = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(bal);
        __syncthreads();
        int woff = 0;
        for (int i = 0; i < wave; ++i) woff += wsum[i];
        int tot = 0;
        for (int i = 0; i < nw; ++i) tot += wsum[i];
        if (r < M) slot[r] = is_img ? (base + woff + prefix) : -1;
        __syncthreads();
        if (threadIdx.x == 0) base += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *count = base;
}

__global__ void embed_fwd_kernel(const long long* __restrict__ ids, const int* __restrict__ slot,
                                 const bf16* __restrict__ table, const bf16* __restrict__ img, bf16* __restrict__ out,
                                 int M, int d, int vocab) {
    const int nch = d >> 3;
    const size_t total = (size_t)M * nch;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % nch);
        const size_t row = i / nch;
        const int s = slot ? slot[row] : -1;
        bf16x8 v;
        if (s >= 0) {
            v = *reinterpret_cast<const bf16x8*>(img + (size_t)s * d + ch * 8);
        } else {
            long long id = ids[row];
            id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
            v = *reinterpret_cast<const bf16x8*>(table + (size_t)id * d + ch * 8);
        }
        *reinterpret_cast<bf16x8*>(out + row * d + ch * 8) = v;
    }
}

// Backward: image rows -> dimg[slot] = dh[row]; text rows -> fp32 atomicAdd into scratch[id] and flag[id]=1.
__global__ void embed_bwd_kernel(const long long* __restrict__ ids, const int* __restrict__ slot,
                                 const bf16* __restrict__ dh, bf16* __restrict__ dimg, float* __restrict__ scratch,
                                 int* __restrict__ flag, int M, int d, int vocab) {
    const int nch = d >> 3;
    const size_t total = (size_t)M * nch;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % nch);
        const size_t row = i / nch;
        const int s = slot ? slot[row] : -1;
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(dh + row * d + ch * 8);
        if (s >= 0) {
            if (dimg) *reinterpret_cast<bf16x8*>(dimg + (size_t)s * d + ch * 8) = v;
        } else if (scratch) {
            long long id = ids[row];
            id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
            float* dst = scratch + (size_t)id * d + ch * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) atomicAdd(dst + j, bf2f(v[j]));
            if (ch == 0) flag[id] = 1;
        }
    }
}

// grad[v] += scratch[v] for flagged rows; scratch and flag are returned to zero.
__global__ void embed_flush_kernel(float* __restrict__ scratch, int* __restrict__ flag, bf16* __restrict__ grad,
                                   int vocab, int d) {
    const int nch = d >> 3;
    for (int v = blockIdx.x; v < vocab; v += gridDim.x) {
        if (flag[v] == 0) continue;
        for (int ch = threadIdx.x; ch < nch; ch += blockDim.x) {
            float* sp = scratch + (size_t)v * d + ch * 8;
            bf16* gp = grad + (size_t)v * d + ch * 8;
            const bf16x8 g = *reinterpret_cast<const bf16x8*>(gp);
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                o[j] = f2bf(bf2f(g[j]) + sp[j]);
                sp[j] = 0.f;
            }
            *reinterpret_cast<bf16x8*>(gp) = o;
        }
        __syncthreads();
        if (threadIdx.x == 0) flag[v] = 0;
    }
}


// Tiled bf16 transpose: dst[c][r] = src[r][c]; 64x64 tiles through LDS, 16-B global accesses both ways.
// A thread loads 8 columns of two ADJACENT source rows, packs the row pair of every column into one dword and writes
// the transposed tile [64 columns][row pairs] with ds_write_b32; the read side is then one 16-B LDS read (8 source
// rows of one column) per 16-B global store.  The tile is 64 columns x TR source rows: a taller tile makes the
// output runs longer (TR * 2 bytes per output row and block; the large transposes are bound by the strided write).
template <int TR>
__global__ __launch_bounds__(256) void transpose_kernel(const bf16* __restrict__ src, bf16* __restrict__ dst, int R,
                                                        int C, int lds_, int ldd) {  // row strides of src / dst
    constexpr int NP = TR / 2;                                                  // row pairs per tile
    __shared__ __attribute__((aligned(16))) unsigned int tileT[64][NP + 4];    // [source column][row pair]
    const int r0 = blockIdx.y * TR, c0 = blockIdx.x * 64;
    const int tc = (threadIdx.x & 7) * 8;
    // The 8 threads of a row pair write columns 8 apart: with a row stride of NP + 4 dwords that is the same bank for all
    // of them (8 x (NP + 4) = 0 mod 32: 8-way conflicts on every ds_write_b32, which bounded these kernels at ~2.5 TB/s).
    // XOR-ing the 4-dword chunk index of the row pair with (column >> 3) & 7 spreads them over all 32 banks; the read
    // side takes whole chunks, so it only XORs its chunk index the same way.
    const int sw = (threadIdx.x & 7) << 2;
#pragma unroll
    for (int g = 0; g < NP / 32; ++g) {
        const int tp = (threadIdx.x >> 3) + g * 32;  // row pair
        const int r = r0 + 2 * tp, c = c0 + tc;
        bf16x8 a, b;
#pragma unroll
        for (int j = 0; j < 8; ++j) { a[j] = f2bf(0.f); b[j] = f2bf(0.f); }
        if (c < C) {
            if (r < R) a = *reinterpret_cast<const bf16x8*>(src + (size_t)r * lds_ + c);
            if (r + 1 < R) b = *reinterpret_cast<const bf16x8*>(src + (size_t)(r + 1) * lds_ + c);
        }
        const uint4 au = __builtin_bit_cast(uint4, a), bu = __builtin_bit_cast(uint4, b);
        const unsigned int aw[4] = {au.x, au.y, au.z, au.w}, bw[4] = {bu.x, bu.y, bu.z, bu.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {  // dword k holds columns 2k (low half) and 2k+1 (high half)
            tileT[tc + 2 * k][tp ^ sw] = (aw[k] & 0xffffu) | (bw[k] << 16);
            tileT[tc + 2 * k + 1][tp ^ sw] = (aw[k] >> 16) | (bw[k] & 0xffff0000u);
        }
    }
    __syncthreads();
    constexpr int QPR = TR / 8;          // 16-B chunks per output row
    constexpr int RPP = 256 / QPR;       // output rows per pass
#pragma unroll
    for (int p = 0; p < 64 / RPP; ++p) {
        const int cl = threadIdx.x / QPR + p * RPP;  // output row = source column
        const int q = threadIdx.x % QPR;             // 8 source rows = 4 row pairs = 16 B
        const int c = c0 + cl, r = r0 + q * 8;
        if (c < C && r < R) {
            const uint4 v = *reinterpret_cast<const uint4*>(&tileT[cl][(q ^ ((cl >> 3) & 7)) * 4]);
            *reinterpret_cast<uint4*>(dst + (size_t)c * ldd + r) = v;
        }
    }
}


// ------------------------------------------------------------------------------------------------
// Fused "elementwise producer -> transposed output" kernels: a 64-column tile is computed in registers, staged in
// LDS and written transposed ([cols][rows], 16-B stores), so the wgrad operands X^T / dY^T come out of their
// producers directly instead of a separate transpose pass (saves a full write + read of the tensor).
// MODE 0: act^T  = (bf16(silu(g)) * u)^T              from gate|up [M][2I]            (SwiGLU recompute in backward)
// MODE 1: y^T    = (w * bf16(x * rstd))^T             from x [M][d], w [d], rstd [M]   (RMSNorm recompute in backward)
// MODE 2: dgu and dgu^T from gate|up and dh           (SwiGLU backward: both layouts are consumed)
// MODE 3: act and act^T from gate|up                  (SwiGLU forward that also leaves the down-proj wgrad operand)
// R rows (tokens), C columns of the OUTPUT of the elementwise op; R % 8 == 0, C % 8 == 0.
// ------------------------------------------------------------------------------------------------
// Tile: 64 columns x TR source rows staged exactly like transpose_kernel (a thread computes 8 columns of two adjacent
// rows, packs each column's row pair into one dword, [column][row pair] image, one 16-B LDS read per 16-B store), so the
// transposed output leaves in runs of TR * 2 bytes per output row and workgroup — the 64 x 64 version of this kernel
// wrote 128-B runs and ran at 2.1-2.5 TB/s in the step (profiles/r02_rocprofv3_kernel_stats.csv).
#ifndef MMU_FUSED_T_ROWS
#define MMU_FUSED_T_ROWS 128
#endif
template <int MODE, int TR>
__global__ __launch_bounds__(256) void fused_t_kernel(const bf16* __restrict__ in0, const bf16* __restrict__ in1,
                                                      const float* __restrict__ rstd, bf16* __restrict__ out_n,
                                                      bf16* __restrict__ out_t, int R, int C) {
    constexpr int NP = TR / 2;            // row pairs per tile
    constexpr int NT = MODE == 2 ? 2 : 1;  // transposed outputs
    __shared__ __attribute__((aligned(16))) unsigned int tileT[NT][64][NP + 4];  // [output][source column][row pair]
    const int r0 = blockIdx.y * TR, c0 = blockIdx.x * 64;
    const int tc = (threadIdx.x & 7) * 8;
    const int c = c0 + tc;
    const int sw = (threadIdx.x & 7) << 2;  // bank swizzle of the staged image, see transpose_kernel
    bf16x8 w8;
    if (MODE == 1 && c < C) w8 = *reinterpret_cast<const bf16x8*>(in1 + c);
#pragma unroll
    for (int g = 0; g < NP / 32; ++g) {
        const int tp = (threadIdx.x >> 3) + g * 32;  // row pair
        bf16x8 o0[2], o1[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = r0 + 2 * tp + h;
#pragma unroll
            for (int j = 0; j < 8; ++j) { o0[h][j] = f2bf(0.f); o1[h][j] = f2bf(0.f); }
            if (r < R && c < C) {
                if (MODE == 0 || MODE == 3) {
                    const bf16x8 gg = *reinterpret_cast<const bf16x8*>(in0 + (size_t)r * 2 * C + c);
                    const bf16x8 u = *reinterpret_cast<const bf16x8*>(in0 + (size_t)r * 2 * C + C + c);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float gf = bf2f(gg[j]);
                        o0[h][j] = f2bf(round_bf16(gf / (1.0f + __expf(-gf))) * bf2f(u[j]));
                    }
                    if (MODE == 3) *reinterpret_cast<bf16x8*>(out_n + (size_t)r * C + c) = o0[h];
                } else if (MODE == 1) {
                    const bf16x8 x = *reinterpret_cast<const bf16x8*>(in0 + (size_t)r * C + c);
                    const float rs = rstd[r];
#pragma unroll
                    for (int j = 0; j < 8; ++j) o0[h][j] = f2bf(bf2f(w8[j]) * round_bf16(bf2f(x[j]) * rs));
                } else {
                    const bf16x8 gg = *reinterpret_cast<const bf16x8*>(in0 + (size_t)r * 2 * C + c);
                    const bf16x8 u = *reinterpret_cast<const bf16x8*>(in0 + (size_t)r * 2 * C + C + c);
                    const bf16x8 d = *reinterpret_cast<const bf16x8*>(in1 + (size_t)r * C + c);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float gf = bf2f(gg[j]), uf = bf2f(u[j]), df = bf2f(d[j]);
                        const float sig = 1.0f / (1.0f + __expf(-gf));
                        o1[h][j] = f2bf(df * round_bf16(gf * sig));                      // d up
                        o0[h][j] = f2bf(df * uf * (sig * (1.0f + gf * (1.0f - sig))));   // d gate
                    }
                    *reinterpret_cast<bf16x8*>(out_n + (size_t)r * 2 * C + c) = o0[h];
                    *reinterpret_cast<bf16x8*>(out_n + (size_t)r * 2 * C + C + c) = o1[h];
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const uint4 au = __builtin_bit_cast(uint4, t ? o1[0] : o0[0]), bu = __builtin_bit_cast(uint4, t ? o1[1] : o0[1]);
            const unsigned int aw[4] = {au.x, au.y, au.z, au.w}, bw[4] = {bu.x, bu.y, bu.z, bu.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {  // dword k holds columns 2k (low half) and 2k+1 (high half)
                tileT[t][tc + 2 * k][tp ^ sw] = (aw[k] & 0xffffu) | (bw[k] << 16);
                tileT[t][tc + 2 * k + 1][tp ^ sw] = (aw[k] >> 16) | (bw[k] & 0xffff0000u);
            }
        }
    }
    __syncthreads();
    constexpr int QPR = TR / 8;     // 16-B chunks per output row
    constexpr int RPP = 256 / QPR;  // output rows per pass
#pragma unroll
    for (int p = 0; p < 64 / RPP; ++p) {
        const int cl = threadIdx.x / QPR + p * RPP;  // output row = source column
        const int q = threadIdx.x % QPR;             // 8 source rows = 4 row pairs = 16 B
        const int cc = c0 + cl, r = r0 + q * 8;
        if (cc < C && r < R) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
                *reinterpret_cast<uint4*>(out_t + (size_t)(t * C + cc) * R + r) =
                    *reinterpret_cast<const uint4*>(&tileT[t][cl][(q ^ ((cl >> 3) & 7)) * 4]);
        }
    }
}
template <int MODE>
void launch_fused_t(const void* in0, const void* in1, const float* rstd, void* out_n, void* out_t, int R, int C, hipStream_t s) {
    constexpr int TR = (MODE == 2 && MMU_FUSED_T_ROWS > 128) ? 128 : MMU_FUSED_T_ROWS;  // two images: 64 KiB of static LDS at most
    hipLaunchKernelGGL((fused_t_kernel<MODE, TR>), dim3((C + 63) / 64, (R + TR - 1) / TR), dim3(256), 0, s, (const bf16*)in0,
                       (const bf16*)in1, rstd, (bf16*)out_n, (bf16*)out_t, R, C);
}

inline int grid_for(size_t work_items, int block) {
    size_t g = (work_items + block - 1) / block;
    if (g > 2048) g = 2048;  // 256 CUs x 8 blocks, grid-stride the rest
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

#define MMU_MAXCH_DISPATCH(d, CALL)          \
    do {                                     \
        if ((d) <= 1024) { CALL(2); }        \
        else if ((d) <= 4096) { CALL(8); }   \
        else if ((d) <= 8192) { CALL(16); }  \
        else return -1;                      \
    } while (0)

extern "C" int mmu_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int M, int d, float eps,
                               void* stream) {
    if (M <= 0 || (d & 7)) return -1;
    hipStream_t s = (hipStream_t)stream;
#define CALL(N)                                                                                              \
    hipLaunchKernelGGL(rmsnorm_fwd_kernel<N>, dim3((M + 3) / 4), dim3(256), 0, s, (const bf16*)x, (const bf16*)w, \
                       (bf16*)y, rstd, M, d, eps)
    MMU_MAXCH_DISPATCH(d, CALL);
#undef CALL
    MMU_CHECK_LAUNCH();
    return 0;
}

// dw_part must hold mmu_rmsnorm_bwd_nparts(M) * d floats.
extern "C" int mmu_rmsnorm_bwd_nparts(int M) {
    return M > 1024 ? 1024 : (M < 1 ? 1 : M);
}

extern "C" int mmu_rmsnorm_bwd(const void* dy, const void* x, const void* w, const float* rstd, const void* dres,
                               void* dx, float* dw_part, void* dw, int accumulate_dw, int M, int d, void* stream) {
    if (M <= 0 || (d & 7) || d > 8192) return -1;
    hipStream_t s = (hipStream_t)stream;
    const int grid = mmu_rmsnorm_bwd_nparts(M);
#define CALL(N)                                                                                              \
    hipLaunchKernelGGL(rmsnorm_bwd_kernel<N>, dim3(grid), dim3(256), 0, s, (const bf16*)dy, (const bf16*)x,  \
                       (const bf16*)w, rstd, (const bf16*)dres, (bf16*)dx, dw_part, M, d)
    if (d <= 2048) { CALL(1); } else if (d <= 4096) { CALL(2); } else if (d <= 6144) { CALL(3); } else { CALL(4); }
#undef CALL
    MMU_CHECK_LAUNCH();
    if (dw) {
        hipLaunchKernelGGL(colsum_partials_kernel, dim3((d + 31) / 32), dim3(256), 0, s, dw_part, grid, d, (bf16*)dw,
                           accumulate_dw);
        MMU_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int mmu_layernorm_fwd(const void* x, const void* w, const void* b, void* y, int M, int d, float eps,
                                 void* stream) {
    if (M <= 0 || (d & 7)) return -1;
    hipStream_t s = (hipStream_t)stream;
#define CALL(N)                                                                                                \
    hipLaunchKernelGGL(layernorm_fwd_kernel<N>, dim3((M + 3) / 4), dim3(256), 0, s, (const bf16*)x, (const bf16*)w, \
                       (const bf16*)b, (bf16*)y, M, d, eps)
    MMU_MAXCH_DISPATCH(d, CALL);
#undef CALL
    MMU_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmu_layernorm_bwd(const void* dy, const void* x, const void* w, const void* dres, void* dx, int M, int d,
                                 float eps, void* stream) {
    if (M <= 0 || (d & 7)) return -1;
    hipStream_t s = (hipStream_t)stream;
#define CALL(N)                                                                                                 \
    hipLaunchKernelGGL(layernorm_bwd_kernel<N>, dim3((M + 3) / 4), dim3(256), 0, s, (const bf16*)dy, (const bf16*)x, \
                       (const bf16*)w, (const bf16*)dres, (bf16*)dx, M, d, eps)
    if (d <= 1024) { CALL(2); } else if (d <= 2048) { CALL(4); } else { return -1; }
#undef CALL
    MMU_CHECK_LAUNCH();
    return 0;
}

// y = dropout(x) with drop probability p in [0, 1): element i is kept iff hash24(seed, i) >= p * 2^24 and scaled by 1/(1-p).
extern "C" int mmu_dropout_bf16(const void* x, void* y, long long n, float p, unsigned long long seed, void* stream) {
    if (n <= 0 || (n & 7) || !(p >= 0.f) || !(p < 1.f)) return -1;
    const unsigned thresh = (unsigned)(p * 16777216.0f);
    hipLaunchKernelGGL(dropout_kernel, dim3(grid_for((size_t)n / 8, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)x,
                       (bf16*)y, (size_t)n / 8, thresh, 1.0f / (1.0f - p), seed);
    MMU_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmu_colsum_nparts(int M);
// dgamma / dbeta of a LayerNorm (see layernorm_stats_kernel): stats = 2 * M floats, part = 2 * mmu_colsum_nparts(M) * d floats
// of workspace; accumulate != 0 adds into the bf16 gradients (gradient accumulation), else overwrites them.
extern "C" int mmu_layernorm_bwd_params(const void* dy, const void* x, float* stats, float* part, void* dgamma, void* dbeta,
                                        int accumulate, int M, int d, float eps, void* stream) {
    if (M <= 0 || d <= 0 || (d & 7) || !stats || !part || !dgamma || !dbeta) return -1;
    hipStream_t s = (hipStream_t)stream;
    const int nparts = mmu_colsum_nparts(M);
    hipLaunchKernelGGL(layernorm_stats_kernel, dim3((M + 3) / 4), dim3(256), 0, s, (const bf16*)x, stats, M, d, eps);
    MMU_CHECK_LAUNCH();
    hipLaunchKernelGGL(layernorm_dwdb_kernel, dim3((d + 255) / 256, nparts), dim3(256), 0, s, (const bf16*)dy, (const bf16*)x,
                       stats, part, M, d, 64, nparts);
    MMU_CHECK_LAUNCH();
    hipLaunchKernelGGL(colsum_partials_kernel, dim3((d + 31) / 32), dim3(256), 0, s, part, nparts, d, (bf16*)dgamma, accumulate);
    MMU_CHECK_LAUNCH();
    hipLaunchKernelGGL(colsum_partials_kernel, dim3((d + 31) / 32), dim3(256), 0, s, part + (size_t)nparts * d, nparts, d,
                       (bf16*)dbeta, accumulate);
    MMU_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmu_quick_gelu_fwd(const void* x, void* y, long long n, void* stream) {
    if (n <= 0 || (n & 7)) return -1;
    hipLaunchKernelGGL(quick_gelu_fwd_kernel, dim3(grid_for((size_t)n / 8, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16*)x, (bf16*)y, (size_t)n / 8);
    MMU_CHECK_LAUNCH();
    return 0;
}
extern "C" int mmu_quick_gelu_bwd(const void* x, const void* dy, void* dx, long long n, void* stream) {
    if (n <= 0 || (n & 7)) return -1;
    hipLaunchKernelGGL(quick_gelu_bwd_kernel, dim3(grid_for((size_t)n / 8, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16*)x, (const bf16*)dy, (bf16*)dx, (size_t)n / 8);
    MMU_CHECK_LAUNCH();
    return 0;
}
extern "C" int mmu_head_pad(const void* src, int ld_src, int d_src, void* dst, int ld_dst, int d_dst, int M, int H,
                            void* stream) {
    if (M <= 0 || H <= 0 || (d_src & 7) || (d_dst & 7) || (ld_src & 7) || (ld_dst & 7)) return -1;
    if (((uintptr_t)src | (uintptr_t)dst) & 15) return -1;
    const size_t work = (size_t)M * H * (d_dst >> 3);
    hipLaunchKernelGGL(head_pad_kernel, dim3(grid_for(work, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)src,
                       ld_src, d_src, (bf16*)dst, ld_dst, d_dst, M, H);
    MMU_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmu_clip_embed(const void* patch, const void* cls, const void* pos, void* out, int B, int T, int d,
                              void* stream) {
    if (B <= 0 || T <= 1 || (d & 7)) return -1;
    const size_t work = (size_t)B * T * (d >> 3);
    hipLaunchKernelGGL(clip_embed_kernel, dim3(grid_for(work, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16*)patch, (const bf16*)cls, (const bf16*)pos, (bf16*)out, B, T, d);
    MMU_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmu_im2col_patch(const float* pix, void* out, int B, int C, int H, int P, int Kpad, void* stream) {
    if (B <= 0 || H % P || Kpad < C * P * P || (Kpad & 7)) return -1;
    const size_t work = (size_t)B * (H / P) * (H / P) * Kpad;
    hipLaunchKernelGGL(im2col_patch_kernel, dim3(grid_for(work, 256)), dim3(256), 0, (hipStream_t)stream, pix,
                       (bf16*)out, B, C, H, P, Kpad);
    MMU_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmu_clip_normalize_u8(const void* in_u8, float* out, long long n_img, int hw, const float* mean3,
                                     const float* std3, void* stream) {
    // mean3/std3 are HOST pointers (three floats each, passed by value to the kernel)
    if (n_img <= 0 || hw <= 0 || (hw & 15) || !mean3 || !std3) return -1;
    if (((uintptr_t)in_u8 | (uintptr_t)out) & 15) return -1;
    const size_t n16 = (size_t)n_img * 3 * (hw >> 4);
    hipLaunchKernelGGL(clip_normalize_u8_kernel, dim3(grid_for(n16, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned char*)in_u8, out, n16, hw >> 4, mean3[0], mean3[1], mean3[2], std3[0], std3[1],
                       std3[2]);
    MMU_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmu_rope(void* qkv, const int* pos_ids, const void* cos_t, const void* sin_t, int M, int ld,
                        int n_q_heads, int n_k_heads, int head_dim, int k_off, int inverse, void* stream) {
    if (M <= 0 || (head_dim & 15) || (ld & 7) || (k_off & 7)) return -1;
    const int nh = n_q_heads + n_k_heads;
    const size_t work = (size_t)M * nh * (head_dim >> 4);
    hipLaunchKernelGGL(rope_kernel, dim3(grid_for(work, 256)), dim3(256), 0, (hipStream_t)stream, (bf16*)qkv, pos_ids,
                       (const bf16*)cos_t, (const bf16*)sin_t, M, ld, nh, head_dim, k_off, n_q_heads, inverse);
    MMU_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmu_rope_table(void* cos_t, void* sin_t, int n_pos, int head_dim, float theta, void* stream) {
    if (n_pos <= 0 || (head_dim & 15)) return -1;
    const int half = head_dim / 2;
    const int n = n_pos * half;
    hipLaunchKernelGGL(rope_table_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (bf16*)cos_t,
                       (bf16*)sin_t, n_pos, half, theta);
    MMU_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmu_swiglu_fwd(const void* gu, void* h, int M, int I, void* stream) {
    if (M <= 0 || (I & 7)) return -1;
    const size_t work = (size_t)M * (I >> 3);
    hipLaunchKernelGGL(swiglu_fwd_kernel, dim3(grid_for(work, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16*)gu, (bf16*)h, M, I);
    MMU_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmu_swiglu_bwd(const void* gu, const void* dh, void* dgu, int M, int I, void* stream) {
    if (M <= 0 || (I & 7)) return -1;
    const size_t work = (size_t)M * (I >> 3);
    hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(grid_for(work, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16*)gu, (const bf16*)dh, (bf16*)dgu, M, I);
    MMU_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmu_gelu_fwd(const void* x, void* y, long long n, void* stream) {
    if (n <= 0 || (n & 7)) return -1;
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3(grid_for((size_t)n / 8, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16*)x, (bf16*)y, (size_t)n / 8);
    MMU_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmu_gelu_bwd(const void* x, const void* dy, void* dx, long long n, void* stream) {
    if (n <= 0 || (n & 7)) return -1;
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3(grid_for((size_t)n / 8, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16*)x, (const bf16*)dy, (bf16*)dx, (size_t)n / 8);
    MMU_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmu_add_bf16(const void* a, const void* b, void* out, long long n, void* stream) {
    if (n <= 0 || (n & 7)) return -1;
    hipLaunchKernelGGL(add_kernel, dim3(grid_for((size_t)n / 8, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16*)a, (const bf16*)b, (bf16*)out, (size_t)n / 8);
    MMU_CHECK_LAUNCH();
    return 0;
}

// Bias gradient: dst[j] (+)= sum_r x[r][j]. part must hold mmu_colsum_nparts(M) * d floats.
extern "C" int mmu_colsum_nparts(int M) {
    int g = (M + 63) / 64;
    return g < 1 ? 1 : g;
}
extern "C" int mmu_colsum_bf16(const void* x, float* part, void* dst, int accumulate, int M, int d, void* stream) {
    if (M <= 0 || d <= 0) return -1;
    hipStream_t s = (hipStream_t)stream;
    const int nparts = mmu_colsum_nparts(M);
    hipLaunchKernelGGL(colsum_bf16_kernel, dim3((d + 255) / 256, nparts), dim3(256), 0, s, (const bf16*)x, part, M, d,
                       64);
    MMU_CHECK_LAUNCH();
    hipLaunchKernelGGL(colsum_partials_kernel, dim3((d + 31) / 32), dim3(256), 0, s, part, nparts, d, (bf16*)dst,
                       accumulate);
    MMU_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmu_image_slots(const long long* ids, int M, long long image_token, int* slot, int* count,
                               void* stream) {
    if (M <= 0) return -1;
    hipLaunchKernelGGL(image_slots_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, ids, M, image_token, slot,
                       count);
    MMU_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmu_embed_fwd(const long long* ids, const int* slot, const void* table, const void* img, void* out,
                             int M, int d, int vocab, void* stream) {
    if (M <= 0 || (d & 7)) return -1;
    const size_t work = (size_t)M * (d >> 3);
    hipLaunchKernelGGL(embed_fwd_kernel, dim3(grid_for(work, 256)), dim3(256), 0, (hipStream_t)stream, ids, slot,
                       (const bf16*)table, (const bf16*)img, (bf16*)out, M, d, vocab);
    MMU_CHECK_LAUNCH();
    return 0;
}

// scratch: [vocab][d] fp32, flag: [vocab] int32 — both must be zero on entry and are zero again on return
// (when grad_table is given). dimg / grad_table may be null (frozen).
extern "C" int mmu_embed_bwd(const long long* ids, const int* slot, const void* dh, void* dimg, float* scratch,
                             int* flag, void* grad_table, int M, int d, int vocab, void* stream) {
    if (M <= 0 || (d & 7)) return -1;
    hipStream_t s = (hipStream_t)stream;
    const size_t work = (size_t)M * (d >> 3);
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(grid_for(work, 256)), dim3(256), 0, s, ids, slot, (const bf16*)dh,
                       (bf16*)dimg, grad_table ? scratch : nullptr, flag, M, d, vocab);
    MMU_CHECK_LAUNCH();
    if (grad_table) {
        hipLaunchKernelGGL(embed_flush_kernel, dim3(2048), dim3(256), 0, s, scratch, flag, (bf16*)grad_table, vocab, d);
        MMU_CHECK_LAUNCH();
    }
    return 0;
}

// dst[C][R] = src[R][C]^T (R, C multiples of 8).  Used to keep W^T resident for the NT dgrad path.
// Strided form: src is [R][C] inside rows of ld_src elements, dst [C][R] inside rows of ld_dst (both multiples of 8,
// 16-B aligned bases): a column block of a wider matrix transposed into a row block of another.
int mmu_internal_transpose_strided(const void* src, int ld_src, void* dst, int ld_dst, int R, int C, hipStream_t stream) {
    if (R <= 0 || C <= 0 || (R & 7) || (C & 7) || (ld_src & 7) || (ld_dst & 7) || ld_src < C || ld_dst < R) return -1;
    if ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) return -1;
    if (R >= 4096) {  // long output rows: 256-row tiles (512-B output runs)
        hipLaunchKernelGGL(transpose_kernel<256>, dim3((C + 63) / 64, (R + 255) / 256), dim3(256), 0, stream,
                           (const bf16*)src, (bf16*)dst, R, C, ld_src, ld_dst);
    } else {
        hipLaunchKernelGGL(transpose_kernel<64>, dim3((C + 63) / 64, (R + 63) / 64), dim3(256), 0, stream,
                           (const bf16*)src, (bf16*)dst, R, C, ld_src, ld_dst);
    }
    MMU_CHECK_LAUNCH();
    return 0;
}
extern "C" int mmu_transpose_bf16(const void* src, void* dst, int R, int C, void* stream) {
    return mmu_internal_transpose_strided(src, C, dst, R, R, C, (hipStream_t)stream);
}

// act^T [I][M] = swiglu(gate|up [M][2I])^T
extern "C" int mmu_swiglu_fwd_t(const void* gu, void* actT, int M, int I, void* stream) {
    if (M <= 0 || (I & 7) || (M & 7)) return -1;
    launch_fused_t<0>(gu, nullptr, nullptr, nullptr, actT, M, I, (hipStream_t)stream);
    MMU_CHECK_LAUNCH();
    return 0;
}
// act [M][I] and act^T [I][M] in one pass over gate|up (forward; act^T is kept for the down-proj wgrad)
extern "C" int mmu_swiglu_fwd_both(const void* gu, void* act, void* actT, int M, int I, void* stream) {
    if (M <= 0 || (I & 7) || (M & 7)) return -1;
    launch_fused_t<3>(gu, nullptr, nullptr, act, actT, M, I, (hipStream_t)stream);
    MMU_CHECK_LAUNCH();
    return 0;
}
// y^T [d][M] = (w * bf16(x * rstd))^T  (RMSNorm output recomputed from the saved rstd)
extern "C" int mmu_rmsnorm_apply_t(const void* x, const void* w, const float* rstd, void* yT, int M, int d, void* stream) {
    if (M <= 0 || (d & 7) || (M & 7)) return -1;
    launch_fused_t<1>(x, w, rstd, nullptr, yT, M, d, (hipStream_t)stream);
    MMU_CHECK_LAUNCH();
    return 0;
}
// dgu [M][2I] and dgu^T [2I][M] from gate|up and dh
extern "C" int mmu_swiglu_bwd_t(const void* gu, const void* dh, void* dgu, void* dguT, int M, int I, void* stream) {
    if (M <= 0 || (I & 7) || (M & 7)) return -1;
    launch_fused_t<2>(gu, dh, nullptr, dgu, dguT, M, I, (hipStream_t)stream);
    MMU_CHECK_LAUNCH();
    return 0;
}

// remove_image_tokens (mm/trainer_utils.py:48-70) without the reference's host loop: the caller computes, on the
// device, the source row of every kept logits row (first `start` rows and last T - start rows of each sequence); this
// kernel copies the selected rows.  dst[r][:] = src[rows[r]][:], one workgroup per destination row, 16-B moves.
namespace {
__global__ __launch_bounds__(256) void gather_rows_kernel(const bf16* __restrict__ src, const long long* __restrict__ rows,
                                                          bf16* __restrict__ dst, int V8, long long n_src_rows) {
    const long long r = rows[blockIdx.x];
    if (r >= n_src_rows) return;  // never true for indices built by the callers; keeps a bad index in bounds
    bf16* d = dst + (size_t)blockIdx.x * V8 * 8;
    if (r < 0) {  // padding slot of a fixed-capacity row list (supervised-row selection): a zero row
        bf16x8 z;
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = f2bf(0.f);
        for (int c = threadIdx.x; c < V8; c += 256) *reinterpret_cast<bf16x8*>(d + c * 8) = z;
        return;
    }
    const bf16* s = src + (size_t)r * V8 * 8;
    for (int c = threadIdx.x; c < V8; c += 256) *reinterpret_cast<bf16x8*>(d + c * 8) = *reinterpret_cast<const bf16x8*>(s + c * 8);
}
// dst[rows[r]][:] = src[r][:] for rows[r] >= 0 (the inverse of the gather for a list of DISTINCT rows: the gradient of the
// selected hidden-state rows back into the [tokens, d] gradient, whose other rows the caller zeroed)
__global__ __launch_bounds__(256) void scatter_rows_kernel(const bf16* __restrict__ src, const long long* __restrict__ rows,
                                                           bf16* __restrict__ dst, int V8, long long n_dst_rows) {
    const long long r = rows[blockIdx.x];
    if (r < 0 || r >= n_dst_rows) return;
    const bf16* s = src + (size_t)blockIdx.x * V8 * 8;
    bf16* d = dst + (size_t)r * V8 * 8;
    for (int c = threadIdx.x; c < V8; c += 256) *reinterpret_cast<bf16x8*>(d + c * 8) = *reinterpret_cast<const bf16x8*>(s + c * 8);
}
}  // namespace
extern "C" int mmu_scatter_rows_bf16(const void* src, const long long* rows, void* dst, long long n_dst_rows, int n_src_rows,
                                     int V, void* stream) {
    if (n_src_rows <= 0 || n_dst_rows <= 0 || V <= 0 || (V & 7) || !rows) return -1;
    if (((uintptr_t)src | (uintptr_t)dst) & 15) return -1;
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(n_src_rows), dim3(256), 0, (hipStream_t)stream, (const bf16*)src, rows,
                       (bf16*)dst, V >> 3, n_dst_rows);
    MMU_CHECK_LAUNCH();
    return 0;
}
extern "C" int mmu_gather_rows_bf16(const void* src, const long long* rows, void* dst, long long n_src_rows, int n_dst_rows,
                                    int V, void* stream) {
    if (n_src_rows <= 0 || n_dst_rows <= 0 || V <= 0 || (V & 7) || !rows) return -1;
    if (((uintptr_t)src | (uintptr_t)dst) & 15) return -1;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(n_dst_rows), dim3(256), 0, (hipStream_t)stream, (const bf16*)src, rows,
                       (bf16*)dst, V >> 3, n_src_rows);
    MMU_CHECK_LAUNCH();
    return 0;
}
