// Kernels of the Transformer recogniser (TrOCR: ViT encoder + autoregressive decoder; reference
// app/ml/models/text_recognizer.py:39-69, which runs transformers' VisionEncoderDecoderModel.generate).  Every dense layer is
// the implicit-GEMM kernel of conv_igemm.hip used as a 1x1 convolution (vtd_api.cpp: build_linear); this file holds what is
// not a GEMM:
//   trocr_resample_kernel      TrOCRProcessor: BGR->RGB, Pillow bilinear (antialiased, 8-bit, two passes) to S x S, /255,
//                              (x-0.5)/0.5, written patch-major so the patch embedding is a plain GEMM
//   trocr_tokens_kernel        [CLS] + patch embeddings + learned positions -> fp32 residual stream
//   trocr_ln_kernel            LayerNorm over a row, optionally fused with the residual add (pre-LN ViT / post-LN decoder);
//                              the residual stream stays fp32, only GEMM inputs are fp16
//   trocr_attention_kernel     encoder self-attention, flash style on v_mfma_f32_16x16x32_f16: S^T = K Q^T so a lane owns one
//                              query column; the probabilities go straight from the accumulator registers into the second
//                              MFMA as its K-permuted operand (no LDS round trip, no shuffles), V is staged transposed
//   trocr_decode_attn_kernel   one query row against a KV cache / the encoder keys (decoder self- and cross-attention)
//   trocr_embed_kernel         token + position (offset 2) embedding gather
//   trocr_argmax_kernel        greedy step: arg-max (lowest index on ties), <eos>/<pad> bookkeeping
#include <cmath>
#include "vtd_common.h"

struct TrocrCrop {
    int frame, x0, y0, w, h;   // frame[y0:y0+h, x0:x0+w]
    int xtab, ytab;            // offsets into the coefficient pool: bounds[S][2] then kk[S][ks]
    int ksx, ksy;
};

namespace {

__device__ __forceinline__ uint8_t clip8_22(int v) {
    v >>= 22;
    return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
}

// One workgroup = R output rows of one crop.  Horizontal pass of the input rows those R rows need -> LDS (u8), vertical pass
// from LDS, normalise, store patch-major: out[crop][(oy/P)*(S/P) + ox/P][c*P*P + (oy%P)*P + ox%P], c in RGB order.
__global__ __launch_bounds__(256) void trocr_resample_kernel(const uint8_t* __restrict__ frames, int H, int W, const TrocrCrop* __restrict__ crops,
                                                             const int* __restrict__ pool, half_t* __restrict__ out, int S, int P, int R,
                                                             int lds_rows) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint8_t* tmp = (uint8_t*)smem;  // [lds_rows][S][3]
    const TrocrCrop c = crops[blockIdx.y];
    const int oy0 = blockIdx.x * R;
    const int* xb = pool + c.xtab;
    const int* xk = xb + 2 * S;
    const int* yb = pool + c.ytab;
    const int* yk = yb + 2 * S;
    const int oy1 = min(oy0 + R, S) - 1;
    const int row0 = yb[2 * oy0];
    const int row1 = yb[2 * oy1] + yb[2 * oy1 + 1];  // exclusive
    const uint8_t* src = frames + ((int64_t)c.frame * H + c.y0) * W * 3 + (int64_t)c.x0 * 3;
    for (int r = row0; r < row1 && r - row0 < lds_rows; ++r) {
        const uint8_t* row = src + (int64_t)r * W * 3;
        for (int ox = threadIdx.x; ox < S; ox += blockDim.x) {
            const int xmin = xb[2 * ox], n = xb[2 * ox + 1];
            const int* k = xk + ox * c.ksx;
            int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
            for (int x = 0; x < n; ++x) {
                const uint8_t* px = row + (x + xmin) * 3;
                const int kv = k[x];
                s0 += px[0] * kv; s1 += px[1] * kv; s2 += px[2] * kv;
            }
            uint8_t* t = tmp + ((r - row0) * S + ox) * 3;
            t[0] = clip8_22(s0); t[1] = clip8_22(s1); t[2] = clip8_22(s2);
        }
    }
    __syncthreads();
    const int pw = S / P;
    for (int i = threadIdx.x; i < R * S; i += blockDim.x) {
        const int oy = oy0 + i / S, ox = i % S;
        if (oy >= S) break;
        const int ymin = yb[2 * oy], n = yb[2 * oy + 1];
        const int* k = yk + oy * c.ksy;
        int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
        for (int y = 0; y < n; ++y) {
            const uint8_t* t = tmp + ((ymin - row0 + y) * S + ox) * 3;
            const int kv = k[y];
            s0 += t[0] * kv; s1 += t[1] * kv; s2 += t[2] * kv;
        }
        const uint8_t bgr[3] = {clip8_22(s0), clip8_22(s1), clip8_22(s2)};
        half_t* o = out + ((int64_t)blockIdx.y * pw * pw + (oy / P) * pw + ox / P) * (3 * P * P) + (oy % P) * P + ox % P;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {  // ch indexes RGB; the frame is BGR
            float v = (float)bgr[2 - ch] / 255.0f;
            v = (v - 0.5f) / 0.5f;
            o[ch * P * P] = (half_t)v;
        }
    }
}

// pixel_values [n,3,S,S] float32 (the tensor the reference hands to generate()) -> the same patch-major fp16 layout
__global__ void trocr_pixels_kernel(const float* __restrict__ x, half_t* __restrict__ out, int S, int P, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int ox = i % S, oy = (i / S) % S, ch = (i / ((int64_t)S * S)) % 3;
    const int64_t b = i / ((int64_t)3 * S * S);
    const int pw = S / P;
    out[(b * pw * pw + (oy / P) * pw + ox / P) * (3 * P * P) + ch * P * P + (oy % P) * P + ox % P] = (half_t)x[i];
}

// stream[b][t][c] = (t == 0 ? cls[c] : patch[b][t-1][c]) + pos[t][c]
__global__ void trocr_tokens_kernel(const float* __restrict__ patch, const float* __restrict__ cls, const float* __restrict__ pos,
                                    float* __restrict__ stream, int T, int C, int ld_patch, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = i % C, t = (i / C) % T;
    const int64_t b = i / ((int64_t)C * T);
    const float v = t == 0 ? cls[c] : patch[(b * (T - 1) + t - 1) * ld_patch + c];
    stream[i] = v + pos[t * C + c];
}

// LayerNorm of one row per wave.  mode 0: out = LN(x).  mode 1 (pre-LN): x += y; out = LN(x).  mode 2 (post-LN): x = LN(x + y)
// (y may be null); out = x.  x is the fp32 residual stream, out16 the fp16 GEMM input, both [rows][C]; y has row stride ldy.
template <int MAXV>
__global__ __launch_bounds__(256) void trocr_ln_kernel(float* __restrict__ x, const float* __restrict__ y, int ldy, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, half_t* __restrict__ out16, float* __restrict__ out32,
                                                       int rows, int C, float eps, int mode) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float v[MAXV];
    const int nv = C / 64;
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        if (i < nv) {
            const int c = i * 64 + lane;
            float t = x[(int64_t)row * C + c];
            if (mode != 0 && y) t += y[(int64_t)row * ldy + c];
            v[i] = t;
            sum += t;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (i < nv) sq += (v[i] - mean) * (v[i] - mean);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    const float rstd = 1.0f / sqrtf(sq / (float)C + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        if (i < nv) {
            const int c = i * 64 + lane;
            const float n = (v[i] - mean) * rstd * gamma[c] + beta[c];
            if (mode == 1) x[(int64_t)row * C + c] = v[i];
            if (mode == 2) x[(int64_t)row * C + c] = n;
            if (out16) out16[(int64_t)row * C + c] = (half_t)n;
            if (out32) out32[(int64_t)row * C + c] = n;
        }
    }
}

// The same with 16-byte accesses (C % 256 == 0, 16-byte aligned rows): a lane owns float4 number i*64 + lane of the row, 3-4 loads
// per operand instead of 12-16.  At the decoder's 272 rows the kernel is three dependent memory round trips whatever the row length;
// fewer instructions in each is what is left to take.
template <int MAXV4>
__global__ __launch_bounds__(256) void trocr_ln4_kernel(float* __restrict__ x, const float* __restrict__ y, int ldy, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, half_t* __restrict__ out16, float* __restrict__ out32,
                                                        int rows, int C, float eps, int mode) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    floatx4 v[MAXV4];
    const int nv = C / 256;
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV4; ++i) {
        if (i < nv) {
            const int c = (i * 64 + lane) * 4;
            floatx4 t = *(const floatx4*)(x + (int64_t)row * C + c);
            if (mode != 0 && y) t += *(const floatx4*)(y + (int64_t)row * ldy + c);
            v[i] = t;
            sum += (t[0] + t[1]) + (t[2] + t[3]);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV4; ++i)
        if (i < nv) {
            const floatx4 d = v[i] - mean;
            sq += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    const float rstd = 1.0f / sqrtf(sq / (float)C + eps);
#pragma unroll
    for (int i = 0; i < MAXV4; ++i) {
        if (i < nv) {
            const int c = (i * 64 + lane) * 4;
            const floatx4 g = *(const floatx4*)(gamma + c), b = *(const floatx4*)(beta + c);
            const floatx4 n = (v[i] - mean) * rstd * g + b;
            if (mode == 1) *(floatx4*)(x + (int64_t)row * C + c) = v[i];
            if (mode == 2) *(floatx4*)(x + (int64_t)row * C + c) = n;
            if (out16) *(half4*)(out16 + (int64_t)row * C + c) = half4{(half_t)n[0], (half_t)n[1], (half_t)n[2], (half_t)n[3]};
            if (out32) *(floatx4*)(out32 + (int64_t)row * C + c) = n;
        }
    }
}

// Encoder self-attention.  qkv: [B][T][3*C] fp16 (q | k | v, head h at columns h*64).  Workgroup = 64 queries of one (b, head),
// wave = 16 queries; keys in blocks of 32 staged in LDS (K row-major, V transposed).
constexpr int ATT_KPAD = 72;   // halfs per K row in LDS (64 + 8: conflict-free 16-byte fragment reads)
constexpr int ATT_VPAD = 40;   // halfs per V^T row (32 keys + 8)
__global__ __launch_bounds__(256) void trocr_attention_kernel(const half_t* __restrict__ qkv, half_t* __restrict__ out, int T, int C, float scale) {
    __shared__ __attribute__((aligned(16))) half_t ks[32 * ATT_KPAD];
    __shared__ __attribute__((aligned(16))) half_t vt[64 * ATT_VPAD];
    const int b = blockIdx.z, head = blockIdx.y, q0 = blockIdx.x * 64;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, fr = lane & 15, fq = lane >> 4;
    const int64_t ld = 3 * (int64_t)C;
    const half_t* base = qkv + (int64_t)b * T * ld + head * 64;
    // Q fragments (second MFMA operand: row = query fr, K chunk fq), pre-scaled
    const int q = q0 + w * 16 + fr;
    half8 qf[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        half8 t = {0, 0, 0, 0, 0, 0, 0, 0};
        if (q < T) t = *(const half8*)(base + (int64_t)q * ld + kk * 32 + fq * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = (half_t)((float)t[e] * scale);
        qf[kk] = t;
    }
    floatx4 acc_o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc_o[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    const int krow = tid >> 3, kcol = (tid & 7) * 8;  // staging: thread -> (key row 0..31, 8 halfs)
    for (int k0 = 0; k0 < T; k0 += 32) {
        __syncthreads();
        {
            const int key = k0 + krow;
            half8 kv = {0, 0, 0, 0, 0, 0, 0, 0}, vv = kv;
            if (key < T) {
                kv = *(const half8*)(base + (int64_t)key * ld + C + kcol);
                vv = *(const half8*)(base + (int64_t)key * ld + 2 * C + kcol);
            }
            *(half8*)(ks + krow * ATT_KPAD + kcol) = kv;
#pragma unroll
            for (int e = 0; e < 8; ++e) vt[(kcol + e) * ATT_VPAD + krow] = vv[e];
        }
        __syncthreads();
        // S^T[key][q]: two 16-key sub-blocks, K dim 64 = 2 x 32
        floatx4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const half8 ka = *(const half8*)(ks + fr * ATT_KPAD + kk * 32 + fq * 8);
            const half8 kb = *(const half8*)(ks + (16 + fr) * ATT_KPAD + kk * 32 + fq * 8);
            s0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ka, qf[kk], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(kb, qf[kk], s1, 0, 0, 0);
        }
        // lane holds keys k0 + fq*4 + e (s0) and k0 + 16 + fq*4 + e (s1) of query fr
        float sv[8];
        float bm = -INFINITY;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sv[e] = (k0 + fq * 4 + e < T) ? s0[e] : -INFINITY;
            sv[4 + e] = (k0 + 16 + fq * 4 + e < T) ? s1[e] : -INFINITY;
            bm = fmaxf(bm, fmaxf(sv[e], sv[4 + e]));
        }
        bm = fmaxf(bm, __shfl_xor(bm, 16));
        bm = fmaxf(bm, __shfl_xor(bm, 32));
        const float m_new = fmaxf(m_run, bm);
        const float corr = expf(m_run - m_new);   // first block: exp(-inf) = 0
        half8 pf;
        float ps = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float p = expf(sv[e] - m_new);
            ps += p;
            pf[e] = (half_t)p;
        }
        l_run = l_run * corr + ps;
        m_run = m_new;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc_o[i] *= corr;
            // V^T fragment: row d = i*16 + fr, K entries = the same key permutation as pf
            const half4 va = *(const half4*)(vt + (i * 16 + fr) * ATT_VPAD + fq * 4);
            const half4 vb = *(const half4*)(vt + (i * 16 + fr) * ATT_VPAD + 16 + fq * 4);
            const half8 vf = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
            acc_o[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf, acc_o[i], 0, 0, 0);
        }
    }
    l_run += __shfl_xor(l_run, 16);
    l_run += __shfl_xor(l_run, 32);
    if (q < T) {
        const float inv = 1.0f / l_run;
        half_t* o = out + ((int64_t)b * T + q) * C + head * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            half4 hv;
#pragma unroll
            for (int e = 0; e < 4; ++e) hv[e] = (half_t)(acc_o[i][e] * inv);
            *(half4*)(o + i * 16 + fq * 4) = hv;
        }
    }
}

// One query row per (crop, head) against L keys: q [B][ldq] fp16 (already scaled), K / V rows of crop b at
// k[b*bsk + key*ldk + head*64 ...].  One wave per (b, head).  A K / V row of a head is one 128-byte line: eight lanes read it
// (16 bytes each) and a wave-instruction covers eight keys in full lines -- one lane per key row instead would touch 64 lines
// for 16 bytes each, eight times over (measured: 43 % of the whole TrOCR pipeline in the first version of this kernel).
__global__ __launch_bounds__(64) void trocr_decode_attn_kernel(const half_t* __restrict__ q, int ldq, const half_t* __restrict__ k,
                                                               const half_t* __restrict__ v, int64_t bsk, int ldk, int L,
                                                               half_t* __restrict__ out, int ldo) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* p = (float*)smem;  // [L rounded up to 8]
    const int b = blockIdx.y, head = blockIdx.x, lane = threadIdx.x;
    const int sub = lane >> 3, seg = lane & 7;   // key within a group of eight, 16-byte segment of the row
    const half8 qv = *(const half8*)(q + (int64_t)b * ldq + head * 64 + seg * 8);
    float qf[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) qf[e] = (float)qv[e];
    const half_t* kb = k + (int64_t)b * bsk + head * 64 + seg * 8;
    const half_t* vb = v + (int64_t)b * bsk + head * 64 + seg * 8;
    float mx = -INFINITY;
    for (int k0 = 0; k0 < L; k0 += 8) {
        const int key = k0 + sub;
        float s = 0.f;
        if (key < L) {
            const half8 kv = *(const half8*)(kb + (int64_t)key * ldk);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += qf[e] * (float)kv[e];
        }
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        s = key < L ? s : -INFINITY;
        if (seg == 0) p[key] = s;
        mx = fmaxf(mx, s);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    __syncthreads();
    const int lpad = (L + 7) & ~7;
    float sum = 0.f;
    for (int key = lane; key < lpad; key += 64) {
        const float e = key < L ? expf(p[key] - mx) : 0.f;
        p[key] = e;
        sum += e;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    __syncthreads();
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < L; k0 += 8) {
        const int key = k0 + sub;
        if (key < L) {
            const float pk = p[key];
            const half8 vv = *(const half8*)(vb + (int64_t)key * ldk);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += pk * (float)vv[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        acc[e] += __shfl_xor(acc[e], 8);
        acc[e] += __shfl_xor(acc[e], 16);
        acc[e] += __shfl_xor(acc[e], 32);
    }
    if (sub == 0) {
        half8 hv;
#pragma unroll
        for (int e = 0; e < 8; ++e) hv[e] = (half_t)(acc[e] / sum);
        *(half8*)(out + (int64_t)b * ldo + head * 64 + seg * 8) = hv;
    }
}

// x[b][:] = embed[token[b]][:] + pos[position + 2][:]   (TrOCRLearnedPositionalEmbedding: offset 2)
__global__ void trocr_embed_kernel(const int32_t* __restrict__ ids, int ld_ids, int col, const float* __restrict__ embed,
                                   const float* __restrict__ pos, float* __restrict__ x, int D, int position, int vocab) {
    const int b = blockIdx.x;
    int tok = ids[(int64_t)b * ld_ids + col];
    tok = tok < 0 ? 0 : tok >= vocab ? vocab - 1 : tok;
    for (int c = threadIdx.x; c < D; c += blockDim.x)
        x[(int64_t)b * D + c] = embed[(int64_t)tok * D + c] + pos[(int64_t)(position + 2) * D + c];
}

// Greedy step for row b: arg-max over V logits (lowest index wins ties, as torch.argmax), then GenerationMixin's bookkeeping:
// a finished row emits <pad>; <eos> finishes the row.  forced != null: teacher forcing (the forced token is emitted instead).
__global__ __launch_bounds__(256) void trocr_argmax_kernel(const float* __restrict__ logits, int64_t ld, int V, int32_t* __restrict__ ids, int ld_ids,
                                                           int col, int32_t* __restrict__ done, const int32_t* __restrict__ forced, int ld_forced,
                                                           int forced_len, int eos, int pad) {
    __shared__ float sm[4];
    __shared__ int si[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* row = logits + (int64_t)b * ld;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = tid; i < V; i += 256) {
        const float v = row[i];
        if (v > best || (v == best && i < bi)) { best = v; bi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o);
        const int oi = __shfl_xor(bi, o);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if ((tid & 63) == 0) { sm[tid >> 6] = best; si[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (sm[w] > best || (sm[w] == best && si[w] < bi)) { best = sm[w]; bi = si[w]; }
        int tok = bi;
        if (forced) tok = col < forced_len ? forced[(int64_t)b * ld_forced + col] : pad;
        else if (done[b]) tok = pad;
        ids[(int64_t)b * ld_ids + col] = tok;
        if (!forced && tok == eos) done[b] = 1;
    }
}

__global__ void trocr_count_done_kernel(const int32_t* __restrict__ done, int n, int32_t* __restrict__ out) {
    int c = 0;
    for (int i = threadIdx.x; i < n; i += 64) c += done[i] ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if (threadIdx.x == 0) *out = c;
}

}  // namespace

int vtd_launch_trocr_resample(const uint8_t* frames, int H, int W, const TrocrCrop* crops_dev, const int* pool, half_t* out, int ncrops, int S,
                              int P, int rows_per_block, int lds_rows, hipStream_t s) {
    if (ncrops <= 0 || S % P || rows_per_block <= 0) return -2401;
    const size_t lds = (size_t)lds_rows * S * 3;
    if (lds > 150 * 1024) return -2402;
    static bool attr = false;
    if (!attr) {
        VTD_HIP_CHECK(hipFuncSetAttribute((const void*)trocr_resample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    hipLaunchKernelGGL(trocr_resample_kernel, dim3((S + rows_per_block - 1) / rows_per_block, ncrops), dim3(256), lds, s, frames, H, W, crops_dev,
                       pool, out, S, P, rows_per_block, lds_rows);
    return -(int)hipGetLastError();
}

int vtd_launch_trocr_pixels(const float* x, half_t* out, int n, int S, int P, hipStream_t s) {
    const int64_t total = (int64_t)n * 3 * S * S;
    hipLaunchKernelGGL(trocr_pixels_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, out, S, P, total);
    return -(int)hipGetLastError();
}

int vtd_launch_trocr_tokens(const float* patch, int ld_patch, const float* cls, const float* pos, float* stream, int n, int T, int C, hipStream_t s) {
    const int64_t total = (int64_t)n * T * C;
    hipLaunchKernelGGL(trocr_tokens_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, patch, cls, pos, stream, T, C, ld_patch, total);
    return -(int)hipGetLastError();
}

int vtd_launch_trocr_ln(float* x, const float* y, int ldy, const float* gamma, const float* beta, half_t* out16, float* out32, int rows, int C,
                        float eps, int mode, hipStream_t s) {
    if (rows <= 0 || (C & 63) || C > 2048) return -2403;
    const dim3 grid((rows + 3) / 4), block(256);
    const bool al = !((uintptr_t)x & 15) && !((uintptr_t)y & 15) && !(ldy & 3) && !((uintptr_t)gamma & 15) && !((uintptr_t)beta & 15) &&
                    !((uintptr_t)out16 & 7) && !((uintptr_t)out32 & 15);
    if (!(C & 255) && al) {
        if (C <= 1024) hipLaunchKernelGGL(trocr_ln4_kernel<4>, grid, block, 0, s, x, y, ldy, gamma, beta, out16, out32, rows, C, eps, mode);
        else hipLaunchKernelGGL(trocr_ln4_kernel<8>, grid, block, 0, s, x, y, ldy, gamma, beta, out16, out32, rows, C, eps, mode);
        return -(int)hipGetLastError();
    }
    if (C <= 1024) hipLaunchKernelGGL(trocr_ln_kernel<16>, grid, block, 0, s, x, y, ldy, gamma, beta, out16, out32, rows, C, eps, mode);
    else hipLaunchKernelGGL(trocr_ln_kernel<32>, grid, block, 0, s, x, y, ldy, gamma, beta, out16, out32, rows, C, eps, mode);
    return -(int)hipGetLastError();
}

int vtd_launch_trocr_attention(const half_t* qkv, half_t* out, int n, int T, int C, int heads, hipStream_t s) {
    if (heads * 64 != C) return -2404;
    hipLaunchKernelGGL(trocr_attention_kernel, dim3((T + 63) / 64, heads, n), dim3(256), 0, s, qkv, out, T, C, 0.125f);
    return -(int)hipGetLastError();
}

int vtd_launch_trocr_decode_attn(const half_t* q, int ldq, const half_t* k, const half_t* v, int64_t batch_stride, int ldk, int L, half_t* out,
                                 int ldo, int n, int heads, hipStream_t s) {
    if (L <= 0 || L > 16384) return -2405;
    hipLaunchKernelGGL(trocr_decode_attn_kernel, dim3(heads, n), dim3(64), (size_t)((L + 7) & ~7) * 4, s, q, ldq, k, v, batch_stride, ldk, L, out, ldo);
    return -(int)hipGetLastError();
}

int vtd_launch_trocr_embed(const int32_t* ids, int ld_ids, int col, const float* embed, const float* pos, float* x, int n, int D, int position,
                           int vocab, hipStream_t s) {
    hipLaunchKernelGGL(trocr_embed_kernel, dim3(n), dim3(256), 0, s, ids, ld_ids, col, embed, pos, x, D, position, vocab);
    return -(int)hipGetLastError();
}

int vtd_launch_trocr_argmax(const float* logits, int64_t ld, int V, int32_t* ids, int ld_ids, int col, int32_t* done, const int32_t* forced,
                            int ld_forced, int forced_len, int eos, int pad, int n, hipStream_t s) {
    hipLaunchKernelGGL(trocr_argmax_kernel, dim3(n), dim3(256), 0, s, logits, ld, V, ids, ld_ids, col, done, forced, ld_forced, forced_len, eos, pad);
    return -(int)hipGetLastError();
}

int vtd_launch_trocr_count_done(const int32_t* done, int n, int32_t* out, hipStream_t s) {
    hipLaunchKernelGGL(trocr_count_done_kernel, dim3(1), dim3(64), 0, s, done, n, out);
    return -(int)hipGetLastError();
}
