// Kernels of the Transformer recogniser (TrOCR: ViT encoder + autoregressive decoder; reference
// app/ml/models/text_recognizer.py:39-69, which runs transformers' VisionEncoderDecoderModel.generate).  Every dense layer is
// the implicit-GEMM kernel of conv_igemm.hip used as a 1x1 convolution (vtd_api.cpp: build_linear); this file holds what is
// not a GEMM:
//   trocr_resample_kernel      TrOCRProcessor: BGR->RGB, Pillow bilinear (antialiased, 8-bit, two passes) to S x S, /255,
//                              (x-0.5)/0.5, written patch-major so the patch embedding is a plain GEMM
//   trocr_tokens_kernel        [CLS] + patch embeddings + learned positions -> fp32 residual stream
//   trocr_ln_kernel            LayerNorm over a row, optionally fused with the residual add (pre-LN ViT / post-LN decoder);
//                              the residual stream stays fp32, only GEMM inputs are fp16
// (encoder self-attention: trocr_attention.hip; the decoder chain: trocr_decode.hip; the encoder pass's dense layers at whole-batch
// token counts: dense_gemm.hip)
#include <cmath>
#include <cstdlib>
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
                                                       int rows, int C, float eps, int mode, int64_t prows) {
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
            if (out16) out16[prows ? (((int64_t)(c >> 5) * prows + row) << 5) + (c & 31) : (int64_t)row * C + c] = (half_t)n;
            if (out32) out32[(int64_t)row * C + c] = n;
        }
    }
}

// The same with 16-byte accesses (C % 256 == 0, 16-byte aligned rows): a lane owns float4 number i*64 + lane of the row, 3-4 loads
// per operand instead of 12-16.  At the decoder's 272 rows the kernel is three dependent memory round trips whatever the row length;
// fewer instructions in each is what is left to take.
template <int MAXV4>
__global__ __launch_bounds__(1024) void trocr_ln4_kernel(float* __restrict__ x, const float* __restrict__ y, int ldy, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, half_t* __restrict__ out16, float* __restrict__ out32,
                                                        int rows, int C, float eps, int mode, int64_t prows) {
    // one wave per row; 4 waves per workgroup, 16 when the fp16 output is K-panel-major (16 consecutive rows = 1 KB contiguous per panel)
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
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
            // (prows: the fp16 GEMM input in K-panel-major order, [C / 32][prows][32] -- dense_gemm.hip reads 1 KB contiguous per LDS-DMA piece)
            if (out16) *(half4*)(out16 + (prows ? (((int64_t)(c >> 5) * prows + row) << 5) + (c & 31) : (int64_t)row * C + c)) = half4{(half_t)n[0], (half_t)n[1], (half_t)n[2], (half_t)n[3]};
            if (out32) *(floatx4*)(out32 + (int64_t)row * C + c) = n;
        }
    }
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
                        float eps, int mode, int64_t out16_prows, hipStream_t s) {
    if (rows <= 0 || (C & 63) || C > 2048) return -2403;
    const dim3 grid((rows + 3) / 4), block(256);
    const bool al = !((uintptr_t)x & 15) && !((uintptr_t)y & 15) && !(ldy & 3) && !((uintptr_t)gamma & 15) && !((uintptr_t)beta & 15) &&
                    !((uintptr_t)out16 & 7) && !((uintptr_t)out32 & 15);
    if (!(C & 255) && al) {
        static const int wide = [] { const char* e = std::getenv("VTD_LN_PANEL_WAVES"); return e ? atoi(e) : 16; }();
        const int waves = out16_prows ? (wide == 4 || wide == 8 || wide == 16 ? wide : 16) : 4;
        const dim3 g4((rows + waves - 1) / waves), b4(64 * waves);
        if (C <= 1024) hipLaunchKernelGGL(trocr_ln4_kernel<4>, g4, b4, 0, s, x, y, ldy, gamma, beta, out16, out32, rows, C, eps, mode, out16_prows);
        else hipLaunchKernelGGL(trocr_ln4_kernel<8>, g4, b4, 0, s, x, y, ldy, gamma, beta, out16, out32, rows, C, eps, mode, out16_prows);
        return -(int)hipGetLastError();
    }
    if (C <= 1024) hipLaunchKernelGGL(trocr_ln_kernel<16>, grid, block, 0, s, x, y, ldy, gamma, beta, out16, out32, rows, C, eps, mode, out16_prows);
    else hipLaunchKernelGGL(trocr_ln_kernel<32>, grid, block, 0, s, x, y, ldy, gamma, beta, out16, out32, rows, C, eps, mode, out16_prows);
    return -(int)hipGetLastError();
}

