// CRNN recogniser stages for gfx950 that are not plain implicit-GEMM convolutions:
//   K6  crop + cv2.resize(img,(128,32)) + /255            (app/ml/inference/pipeliine.py:121, text_recognizer.py:118-119)
//   K7a conv1 3x3 (3->64) + BN + ReLU + maxpool 2x2 fused  (text_recognizer.py:17)
//   K8  bidirectional LSTM recurrence: one 16-wave workgroup per 16 crops x direction, state in registers/LDS
//                                                          (text_recognizer.py:26,34; nn.LSTM gate order i,f,g,o)
//   K9  softmax + argmax + the reference's greedy CTC decode (text_recognizer.py:126,142-167)
#include "../../include/vtd.h"
#include "vtd_common.h"

namespace {

// ------------------------------------------------------------------------------------------------ K6
// One workgroup per crop.  8-bit bilinear exactly as OpenCV's fixed-point path: 11-bit coefficients,
// horizontal pass in int, vertical combine ((b0*(h0>>4))>>16 + (b1*(h1>>4))>>16 + 2)>>2; an exact 2x2
// decimation (256x64 source) takes the area-average route like cv::resize does.
struct CropParams {
    const uint8_t* frames;  // [n, H, W, 3] BGR
    const int32_t* boxes;   // [ncrops][5] frame, x1, y1, x2, y2  (crop = frame[y1:y2, x1:x2])
    uint8_t* out;           // [ncrops, 32, 128, 3]
    int H, W, ncrops;
};

__device__ __forceinline__ void linear_coef(int ssize, int dsize, int d, int& ofs, int& c0, int& c1) {
    const double scale = 1.0 / ((double)dsize / (double)ssize);
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
    ofs = s;
    c0 = __float2int_rn((1.f - f) * 2048.f);
    c1 = __float2int_rn(f * 2048.f);
}

__global__ __launch_bounds__(256) void crop_resize_kernel(const CropParams p) {
    const int crop = blockIdx.x;
    const int32_t* b = p.boxes + crop * 5;
    const int x1 = b[1], y1 = b[2], sw = b[3] - b[1], sh = b[4] - b[2];
    uint8_t* dst = p.out + (int64_t)crop * 32 * 128 * 3;
    if (sw <= 0 || sh <= 0 || b[0] < 0 || x1 < 0 || y1 < 0 || b[3] > p.W || b[4] > p.H) {  // never read outside the frame
        for (int i = threadIdx.x; i < 32 * 128 * 3; i += 256) dst[i] = 0;
        return;
    }
    const uint8_t* src = p.frames + ((int64_t)b[0] * p.H + y1) * p.W * 3 + (int64_t)x1 * 3;
    const int64_t stride = (int64_t)p.W * 3;
    const bool area = (sw == 256 && sh == 64);
    for (int i = threadIdx.x; i < 32 * 128; i += 256) {
        const int dy = i >> 7, dx = i & 127;
        if (area) {
            const uint8_t* r0 = src + (int64_t)(2 * dy) * stride + (2 * dx) * 3;
            const uint8_t* r1 = r0 + stride;
#pragma unroll
            for (int c = 0; c < 3; ++c) dst[i * 3 + c] = (uint8_t)((r0[c] + r0[3 + c] + r1[c] + r1[3 + c] + 2) >> 2);
            continue;
        }
        int sx0, a0, a1, sy0, b0, b1;
        linear_coef(sw, 128, dx, sx0, a0, a1);
        linear_coef(sh, 32, dy, sy0, b0, b1);
        const int sx1 = min(sx0 + 1, sw - 1), sy1 = min(sy0 + 1, sh - 1);
        const uint8_t* r0 = src + (int64_t)sy0 * stride;
        const uint8_t* r1 = src + (int64_t)sy1 * stride;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int h0 = r0[sx0 * 3 + c] * a0 + r0[sx1 * 3 + c] * a1;
            const int h1 = r1[sx0 * 3 + c] * a0 + r1[sx1 * 3 + c] * a1;
            int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
            dst[i * 3 + c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
}

// ------------------------------------------------------------------------------------------------ K7a
// conv1 3x3 (3 -> 64, K = 27 padded to 32) + BN + ReLU + 2x2 max-pool in one launch.  One workgroup = one pooled output
// row of one crop = 2 conv rows x 128 px = 256 GEMM rows.  The im2col tile [256][32] fp16 is built in LDS with the rows
// ordered so that the four members of a pool window sit at the same lane position of a wave's four M-fragments: the
// 2x2 max is then an element-wise max of four accumulator registers -- no shuffles, no second pass.
struct Conv1Params {
    const uint8_t* in_u8;   // [n,32,128,3] uint8 (scaled by 1/255 here) or NULL
    const float* in_f32;    // [n,3,32,128] float32 (reference-format tensor) or NULL
    const half_t* w;        // [64][32] fp16 BN-folded, k = (r*3+s)*3 + c, k >= 27 zero
    const float* bias;      // [64]
    half_t* out;            // [n, 16+2, 64+2, 64] ring 1
    int n;
};

__global__ __launch_bounds__(256) void crnn_conv1_pool_kernel(const Conv1Params p) {
    __shared__ __attribute__((aligned(16))) half_t atile[256 * 32];  // swizzled: 16-byte chunk c of row r at c ^ ((r>>2)&3)
    __shared__ half_t tile[4][130][4];                                // input rows 2*py-1 .. 2*py+2, columns -1..128, zero padded
    const int crop = blockIdx.y, py = blockIdx.x;
    for (int i = threadIdx.x; i < 4 * 130 * 3; i += 256) {
        const int c = i % 3, xx = (i / 3) % 130, r = i / (3 * 130);
        const int y = 2 * py - 1 + r, x = xx - 1;
        float v = 0.f;
        if (y >= 0 && y < 32 && x >= 0 && x < 128) {
            if (p.in_u8) v = (float)p.in_u8[(((int64_t)crop * 32 + y) * 128 + x) * 3 + c] / 255.0f;
            else v = p.in_f32[(((int64_t)crop * 3 + c) * 32 + y) * 128 + x];
        }
        tile[r][xx][c] = (half_t)v;
    }
    __syncthreads();
    {   // thread t builds GEMM row t: wave w = t>>6 owns pooled pixels 16w..16w+15; row = w*64 + member*16 + window
        const int t = threadIdx.x, w = t >> 6, member = (t >> 4) & 3, win = t & 15;
        const int oy = member >> 1, ox = 2 * (w * 16 + win) + (member & 1);
        half_t vals[32];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int c = 0; c < 3; ++c) vals[(r * 3 + s) * 3 + c] = tile[oy + r][ox + s][c];
#pragma unroll
        for (int k = 27; k < 32; ++k) vals[k] = (half_t)0.f;
#pragma unroll
        for (int ck = 0; ck < 4; ++ck) {
            half8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = vals[ck * 8 + e];
            *(half8*)(atile + t * 32 + ((ck ^ ((t >> 2) & 3)) * 8)) = v;
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    floatx4 acc[4][4];  // [channel tile][pool member]
    half8 wf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) wf[i] = *(const half8*)(p.w + (i * 16 + fr) * 32 + fq * 8);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = wv * 64 + j * 16 + fr;
        const half8 af = *(const half8*)(atile + row * 32 + ((fq ^ ((row >> 2) & 3)) * 8));
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], af, floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    }
    // lane: channels i*16 + fq*4 .. +4 of pooled pixel 16wv + fr; max over the four members, then bias + ReLU (monotone)
    half_t* o = p.out + (((int64_t)crop * 18 + py + 1) * 66 + wv * 16 + fr + 1) * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 bv = *(const float4*)(p.bias + i * 16 + fq * 4);
        const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
        half4 hv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float m = fmaxf(fmaxf(acc[i][0][e], acc[i][1][e]), fmaxf(acc[i][2][e], acc[i][3][e])) + bb[e];
            hv[e] = (half_t)(m > 0.f ? m : 0.f);
        }
        *(half4*)(o + i * 16 + fq * 4) = hv;
    }
}

// ------------------------------------------------------------------------------------------------ K8
// LSTM recurrence for one layer, both directions.  grid = (ceil(D/16), 2 directions), 1024 threads = 16 waves.
// Wave w owns hidden units [16w, 16w+16) for all four gates, so the gate maths for a unit never crosses lanes.
// Per step: h_{t-1} (16 crops x 256, fp16, LDS) is the MFMA B operand, W_hh rows are the A operand, the
// accumulators start from the hoisted input GEMM x_t W_ih^T + b (fp32), the wave applies the gate
// non-linearities to its own 16 units x 16 crops in registers (cell state never leaves registers), and
// writes h_t back to LDS (next step's operand) and to HBM (next layer's input).  One barrier per step.
struct LstmParams {
    const half_t* xs;     // [D*T, 2048] fp16: columns dir*1024 + gate*256 + unit
    const half_t* whh;    // [2][1024][256] fp16 (dir, gate*256+unit, k)
    half_t* hout;         // [D, T, 512] fp16 (fwd | rev)
    int D, T;
};

// gate non-linearities on the transcendental unit: v_exp_f32 + v_rcp_f32 (each ~1 ulp); an IEEE divide would cost
// ~10 VALU instructions per gate and the recurrence is latency-bound
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 2.f * __builtin_amdgcn_rcpf(1.f + __expf(-2.f * x)) - 1.f; }

__global__ __launch_bounds__(1024) void lstm_recurrence_kernel(const LstmParams p) {
    // W_hh of one direction is 512 KB -- exactly one CU's whole register file -- so it cannot be fully resident
    // next to the state.  Split per wave (32 fragments of 1 KB: 4 gates x 8 K-chunks):
    //   K-chunks 0-3 -> registers (64 VGPRs), 4-5 -> LDS (8 KB per wave, lane-linear), 6-7 -> re-streamed from L2
    // each step (128 KB per workgroup instead of 512 KB), issued first so the other 24 MFMAs cover their latency.
    constexpr int HROW = 264;  // 256 + 8 halves of padding: rows land on different LDS banks
    extern __shared__ __attribute__((aligned(16))) char lsm[];
    half_t (*hbuf)[16][HROW] = (half_t (*)[16][HROW])lsm;          // [2][16][HROW]
    char* wlds = lsm + 2 * 16 * HROW * sizeof(half_t);              // [16 waves][8 frags][64 lanes][16 B]
    const int dir = blockIdx.y;
    const int crop0 = blockIdx.x * 16;
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;          // hidden units [16wv, 16wv+16)
    const int fr = lane & 15, fq = lane >> 4;

    // fragment (gate g, chunk kc): A operand, row = unit (lane&15), k = 32*kc + 8*(lane>>4) .. +8
    const half_t* wlane = p.whh + (int64_t)dir * 1024 * 256 + (int64_t)(wv * 16 + fr) * 256 + fq * 8;
    half8 wreg[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) wreg[g][kc] = *(const half8*)(wlane + (int64_t)g * 256 * 256 + kc * 32);
    char* wl = wlds + (wv * 8) * 1024 + lane * 16;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int kc = 4; kc < 6; ++kc)
            *(half8*)(wl + (g * 2 + (kc - 4)) * 1024) = *(const half8*)(wlane + (int64_t)g * 256 * 256 + kc * 32);

    for (int i = threadIdx.x; i < 2 * 16 * HROW; i += 1024) (&hbuf[0][0][0])[i] = (half_t)0.f;
    float cst[4] = {0.f, 0.f, 0.f, 0.f};  // cell state of units 16wv + 4fq + j, crop crop0 + fr
    __syncthreads();

    const int crop = min(crop0 + fr, p.D - 1);
    const bool live = crop0 + fr < p.D;
    // The streamed W_hh fragments are issued at the top of a step and consumed by its last 8 MFMAs (24 MFMAs of cover);
    // the input projection of step s+1 does not depend on h either and is prefetched right after step s's gate maths, so
    // its HBM latency hides behind the barrier and the next step's MFMAs.  (Prefetching both across the barrier does not
    // fit the 128-VGPR budget of a 16-wave workgroup.)
    const half_t* xbase = p.xs + (int64_t)crop * p.T * 2048 + dir * 1024 + wv * 16 + fq * 4;
    half4 xin[4];
    {
        const int t0 = dir ? p.T - 1 : 0;
#pragma unroll
        for (int g = 0; g < 4; ++g) xin[g] = *(const half4*)(xbase + (int64_t)t0 * 2048 + g * 256);
    }
    for (int step = 0; step < p.T; ++step) {
        const int t = dir ? p.T - 1 - step : step;
        const int cur = step & 1;
        half8 ws[4][2];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int kc = 6; kc < 8; ++kc) ws[g][kc - 6] = *(const half8*)(wlane + (int64_t)g * 256 * 256 + kc * 32);
        floatx4 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            const half8 hf = *(const half8*)(&hbuf[cur][fr][kc * 32 + fq * 8]);
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[g][kc], hf, acc[g], 0, 0, 0);
        }
#pragma unroll
        for (int kc = 4; kc < 6; ++kc) {
            const half8 hf = *(const half8*)(&hbuf[cur][fr][kc * 32 + fq * 8]);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const half8 wf = *(const half8*)(wl + (g * 2 + (kc - 4)) * 1024);
                acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, hf, acc[g], 0, 0, 0);
            }
        }
#pragma unroll
        for (int kc = 6; kc < 8; ++kc) {
            const half8 hf = *(const half8*)(&hbuf[cur][fr][kc * 32 + fq * 8]);
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ws[g][kc - 6], hf, acc[g], 0, 0, 0);
        }
        const int tn = dir ? max(t - 1, 0) : min(t + 1, p.T - 1);  // next step's row (clamped: the last prefetch is unused)
        half4 hv;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float ig = sigmoidf_(acc[0][j] + (float)xin[0][j]), fg = sigmoidf_(acc[1][j] + (float)xin[1][j]),
                        gg = tanhf_(acc[2][j] + (float)xin[2][j]), og = sigmoidf_(acc[3][j] + (float)xin[3][j]);
            cst[j] = fg * cst[j] + ig * gg;
            hv[j] = (half_t)(og * tanhf_(cst[j]));
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) xin[g] = *(const half4*)(xbase + (int64_t)tn * 2048 + g * 256);
        *(half4*)(&hbuf[cur ^ 1][fr][wv * 16 + fq * 4]) = hv;
        if (live) *(half4*)(p.hout + ((int64_t)crop * p.T + t) * 512 + dir * 256 + wv * 16 + fq * 4) = hv;
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ K9
// One wave per sequence.  Row maxima / arg-maxima / softmax denominators by wave reduction, then lane 0
// replays the reference's decode loop (blank does not reset prev, '<unk>' emits nothing but resets it,
// confidence row = output length - 1).
struct DecodeParams {
    const float* logits;   // [n, T, ld] (first V columns valid)
    const int32_t* id2char; // [V] code point or -1 (blank / unk / unmapped)
    int32_t* out;          // [n][2 + T]: length, confidence bits (float), then `length` code points
    int n, T, V, ld, blank;
    int apply_softmax;     // 1: rows are logits (softmax(dim=2) fused, text_recognizer.py:126); 0: rows are probabilities
};

__global__ __launch_bounds__(64) void ctc_greedy_kernel(const DecodeParams p) {
    __shared__ int s_idx[128];
    __shared__ float s_pmax[128];
    const int seq = blockIdx.x, lane = threadIdx.x;
    for (int t = 0; t < p.T; ++t) {
        const float* row = p.logits + ((int64_t)seq * p.T + t) * p.ld;
        float best = -INFINITY;
        int bi = 0x7fffffff;
        for (int v = lane; v < p.V; v += 64) {
            const float x = row[v];
            if (x > best || (x == best && v < bi)) { best = x; bi = v; }
        }
        for (int d = 32; d >= 1; d >>= 1) {
            const float ob = __shfl_xor(best, d);
            const int oi = __shfl_xor(bi, d);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        float pmax = best;
        if (p.apply_softmax) {
            float sum = 0.f;
            for (int v = lane; v < p.V; v += 64) sum += expf(row[v] - best);
            for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d);
            pmax = 1.0f / sum;
        }
        if (lane == 0) { s_idx[t] = bi; s_pmax[t] = pmax; }
    }
    __syncthreads();
    if (lane == 0) {
        int32_t* o = p.out + (int64_t)seq * (2 + p.T);
        int len = 0, prev = -1;
        double csum = 0.0;
        for (int t = 0; t < p.T; ++t) {
            const int k = s_idx[t];
            if (k == p.blank || k == prev) continue;
            const int ch = (k >= 0 && k < p.V) ? p.id2char[k] : -1;
            if (ch >= 0) {
                o[2 + len] = ch;
                ++len;
                csum += (double)s_pmax[len - 1];
            }
            prev = k;
        }
        o[0] = len;
        const float conf = len ? (float)(csum / (double)len) : 0.f;
        o[1] = __float_as_int(conf);
    }
}

// dense copy of the first V columns of a [rows, ld] fp32 matrix
__global__ void compact_rows_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t rows, int V, int ld) {
    const int64_t total = rows * V;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / V;
        out[i] = in[r * ld + (i - r * V)];
    }
}

}  // namespace

int vtd_launch_crop_resize(const uint8_t* frames, int H, int W, const int32_t* boxes, int ncrops, uint8_t* out, hipStream_t s) {
    CropParams p{frames, boxes, out, H, W, ncrops};
    hipLaunchKernelGGL(crop_resize_kernel, dim3(ncrops), dim3(256), 0, s, p);
    return -(int)hipGetLastError();
}

int vtd_launch_crnn_conv1(const uint8_t* in_u8, const float* in_f32, const half_t* w, const float* bias, half_t* out, int n, hipStream_t s) {
    Conv1Params p{in_u8, in_f32, w, bias, out, n};
    hipLaunchKernelGGL(crnn_conv1_pool_kernel, dim3(16, n), dim3(256), 0, s, p);
    return -(int)hipGetLastError();
}

int vtd_launch_lstm(const half_t* xs, const half_t* whh, half_t* hout, int D, int T, hipStream_t s) {
    LstmParams p{xs, whh, hout, D, T};
    constexpr int lds = 2 * 16 * 264 * 2 + 16 * 8 * 1024;  // h double buffer + LDS-resident W_hh slice
    static bool attr_done = false;
    if (!attr_done) {
        VTD_HIP_CHECK(hipFuncSetAttribute((const void*)lstm_recurrence_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_done = true;
    }
    hipLaunchKernelGGL(lstm_recurrence_kernel, dim3((D + 15) / 16, 2), dim3(1024), lds, s, p);
    return -(int)hipGetLastError();
}

int vtd_launch_ctc_greedy(const float* logits, int n, int T, int V, int ld, const int32_t* id2char, int blank, int apply_softmax,
                          int32_t* out, hipStream_t s) {
    if (T > 128 || T <= 0 || V <= 0) return -1040;
    DecodeParams p{logits, id2char, out, n, T, V, ld, blank, apply_softmax};
    hipLaunchKernelGGL(ctc_greedy_kernel, dim3(n), dim3(64), 0, s, p);
    return -(int)hipGetLastError();
}

int vtd_launch_compact_rows(const float* in, float* out, int64_t rows, int V, int ld, hipStream_t s) {
    const int64_t total = rows * V;
    hipLaunchKernelGGL(compact_rows_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 4096)), dim3(256), 0, s, in, out, rows, V, ld);
    return -(int)hipGetLastError();
}
