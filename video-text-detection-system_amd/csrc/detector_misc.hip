// HBM-bound detector stages for gfx950: fused frame preprocess (K1), NCHW->internal conversion,
// stem max-pool, and the final ConvTranspose(64->1)+sigmoid of the DB head.
#include "vtd_common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// K1: BGR uint8 HWC frame -> RGB, Pillow antialiased-bilinear resize to 640x640 (horizontal pass to a
// uint8 intermediate, then vertical -- bit-exact integer arithmetic), /255, (x-mean)/std, fp16 NHWC4
// into the ring-3 network input.  Replaces cv2.cvtColor + ToPILImage/Resize/ToTensor/Normalize
// (app/ml/models/text_detector.py:99-104,119-124).
//
// One workgroup = TY output rows of one frame.  Phase 1 resamples the input rows those output rows
// touch horizontally into LDS (coalesced row reads, each input row is read by at most two workgroups);
// phase 2 runs the vertical taps out of LDS and writes 8-byte pixels.
// ------------------------------------------------------------------------------------------------
constexpr int PRE_TY = 16;
constexpr int PRE_OUT = 640;
constexpr int PRE_CHUNK = 4;  // raw input rows staged per pass
#ifndef VTD_PRE_NT
#define VTD_PRE_NT 640
#endif
constexpr int PRE_NT = VTD_PRE_NT;   // threads of the fast path's workgroup (an instrumented build may pass -DVTD_PRE_NT=256 for the A/B)
constexpr int PRE_COLS = (PRE_OUT + PRE_NT - 1) / PRE_NT;   // output columns per thread

struct PreParams {
    const uint8_t* frames;  // [n, H, W, 3]
    half_t* out;            // [n, 646, 646, 4]
    const int* xb;          // [640][2] xmin, count
    const int* xk;          // [640][ksx]
    const int* yb;          // [640][2]
    const int* yk;          // [640][ksy]
    int H, W, ksx, ksy;
    int max_rows;           // LDS rows reserved per workgroup
    unsigned quads_magic;   // fast path: ceil(2^32 / (W / 4))
};

__device__ __forceinline__ uint8_t clip8_22(int v) {
    v >>= 22;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

__global__ __launch_bounds__(256) void preprocess_kernel(const PreParams p) {
    // LDS: [max_rows][640*3] horizontally resampled rows, then a staging area for PRE_CHUNK raw input rows
    extern __shared__ __attribute__((aligned(16))) uint8_t rows[];
    const int oy0 = blockIdx.x * PRE_TY;
    const int img = blockIdx.y;
    const int y_first = p.yb[2 * oy0];
    const int oy_last = min(oy0 + PRE_TY, PRE_OUT) - 1;
    const int y_end = p.yb[2 * oy_last] + p.yb[2 * oy_last + 1];
    const int nrows = y_end - y_first;
    const int row_bytes = p.W * 3;
    const int row_pad = (row_bytes + 15) & ~15;
    uint8_t* raw = rows + ((p.max_rows * PRE_OUT * 3 + 15) & ~15);
    const uint8_t* src = p.frames + (int64_t)img * p.H * row_bytes;

    // phase 1: raw rows -> LDS with the widest loads the row pitch allows (coalesced), then the horizontal taps run
    // out of LDS; every input row is fetched from HBM once per workgroup
    for (int r0 = 0; r0 < nrows; r0 += PRE_CHUNK) {
        const int nr = min(PRE_CHUNK, nrows - r0);
        if ((row_bytes & 15) == 0) {
            const int vec_per_row = row_bytes >> 4;
            for (int i = threadIdx.x; i < nr * vec_per_row; i += 256) {
                const int rr = i / vec_per_row, v = i - rr * vec_per_row;
                *(uint4*)(raw + rr * row_pad + v * 16) = *(const uint4*)(src + (int64_t)(y_first + r0 + rr) * row_bytes + v * 16);
            }
        } else {
            for (int i = threadIdx.x; i < nr * row_bytes; i += 256) {
                const int rr = i / row_bytes, v = i - rr * row_bytes;
                raw[rr * row_pad + v] = src[(int64_t)(y_first + r0 + rr) * row_bytes + v];
            }
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < nr * PRE_OUT; idx += 256) {
            const int rr = idx / PRE_OUT, ox = idx - rr * PRE_OUT;
            const uint8_t* row = raw + rr * row_pad;
            const int xmin = p.xb[2 * ox], cnt = p.xb[2 * ox + 1];
            const int* k = p.xk + ox * p.ksx;
            int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
            for (int t = 0; t < cnt; ++t) {
                const int kv = k[t];
                const uint8_t* px = row + (xmin + t) * 3;
                s0 += px[0] * kv;
                s1 += px[1] * kv;
                s2 += px[2] * kv;
            }
            uint8_t* d = rows + ((r0 + rr) * PRE_OUT + ox) * 3;
            d[0] = clip8_22(s0);
            d[1] = clip8_22(s1);
            d[2] = clip8_22(s2);
        }
        __syncthreads();
    }

    // phase 2: vertical taps + normalise; input channel order is BGR, output RGB0
    const float mean[3] = {0.485f, 0.456f, 0.406f};
    const float stdv[3] = {0.229f, 0.224f, 0.225f};
    for (int idx = threadIdx.x; idx < PRE_TY * PRE_OUT; idx += 256) {
        const int ty = idx / PRE_OUT, ox = idx - ty * PRE_OUT;
        const int oy = oy0 + ty;
        if (oy >= PRE_OUT) break;
        const int ymin = p.yb[2 * oy] - y_first, cnt = p.yb[2 * oy + 1];
        const int* k = p.yk + oy * p.ksy;
        int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
        for (int t = 0; t < cnt; ++t) {
            const int kv = k[t];
            const uint8_t* px = rows + ((ymin + t) * PRE_OUT + ox) * 3;
            s0 += px[0] * kv;
            s1 += px[1] * kv;
            s2 += px[2] * kv;
        }
        const float b = (float)clip8_22(s0), g = (float)clip8_22(s1), r = (float)clip8_22(s2);
        half4 o;
        o[0] = (half_t)((r / 255.0f - mean[0]) / stdv[0]);
        o[1] = (half_t)((g / 255.0f - mean[1]) / stdv[1]);
        o[2] = (half_t)((b / 255.0f - mean[2]) / stdv[2]);
        o[3] = (half_t)0.f;
        *(half4*)(p.out + (((int64_t)img * 646 + oy + 3) * 646 + ox + 3) * 4) = o;
    }
}

// Fast path of K1 for the common tap counts (720p: 5 x 5 taps, 1080p: 7 x 5).  Same integer arithmetic, same results; what
// changes is how the bytes move:
//   * raw BGR rows are widened to one dword per pixel (B | G<<8 | R<<16) while they are staged in LDS, and the horizontally
//     resampled rows are kept in that form too: a tap is ONE aligned LDS read for all three channels instead of three
//     byte reads (the old kernel was bound by LDS / TA instruction issue: 15 byte reads + 7 table loads per output pixel);
//   * a thread owns fixed output columns (t, t+256, t+512) for the whole horizontal pass, so its tap tables live in
//     registers; taps past a window's end carry zero weights (Pillow pads its tables the same way);
//   * the vertical pass produces 4 neighbouring pixels per thread: one 16-byte LDS read per tap, one 32-byte store.
// PRE_NT threads = one output column per thread (640): ten waves per workgroup, two workgroups per CU.  With 256 threads (three columns per
// thread, the third on half the lanes) the waves sat parked at the chunk barriers for 63 % of their cycles (r04_pmc_sq_summary.json) with
// only eight of them on a CU to cover for one another.
template <int KSX, int KSY>
__global__ __launch_bounds__(PRE_NT) void preprocess_fast_kernel(const PreParams p) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    // [max_rows][640] dwords (resampled rows) | PRE_CHUNK x [W + KSX] dwords (raw rows, widened) | 16 x (2 + KSY) ints (row taps)
    uint32_t* const rows = (uint32_t*)lds;
    const int raw_pitch = p.W + KSX;
    uint32_t* const raw = rows + p.max_rows * PRE_OUT;
    int* const ytab = (int*)(raw + PRE_CHUNK * raw_pitch);
    half_t* const lut = (half_t*)(ytab + PRE_TY * (2 + KSY));  // [3][256]: normalised value of every 8-bit level, per output channel
    const int oy0 = blockIdx.x * PRE_TY;
    const int img = blockIdx.y;
    const int y_first = p.yb[2 * oy0];
    const int oy_last = min(oy0 + PRE_TY, PRE_OUT) - 1;
    const int y_end = p.yb[2 * oy_last] + p.yb[2 * oy_last + 1];
    const int nrows = y_end - y_first;
    const int row_bytes = p.W * 3;
    const uint8_t* src = p.frames + (int64_t)img * p.H * row_bytes;
    const int tid = threadIdx.x;

    // this thread's output columns and their horizontal taps
    int xmin[PRE_COLS], kx[PRE_COLS][KSX];
#pragma unroll
    for (int s = 0; s < PRE_COLS; ++s) {
        const int ox = min(tid + PRE_NT * s, PRE_OUT - 1);
        xmin[s] = p.xb[2 * ox];
#pragma unroll
        for (int t = 0; t < KSX; ++t) kx[s][t] = p.xk[ox * KSX + t];
    }
    for (int i = tid; i < PRE_TY * (2 + KSY); i += PRE_NT) {
        const int ty = i / (2 + KSY), e = i - ty * (2 + KSY);
        const int oy = min(oy0 + ty, PRE_OUT - 1);
        ytab[i] = e < 2 ? p.yb[2 * oy + e] : p.yk[oy * KSY + (e - 2)];
    }
    // zero the KSX padding pixels behind every raw row once (read with zero weights only, but they must be defined)
    for (int i = tid; i < PRE_CHUNK * KSX; i += PRE_NT) raw[(i / KSX) * raw_pitch + p.W + (i % KSX)] = 0u;
    {   // The resampled pixel is an 8-bit level, so (v / 255 - mean) / std has 256 possible values per channel: evaluate the
        // reference's expression (two IEEE divisions, ~25 instructions) once per level instead of once per pixel.
        const float mean[3] = {0.485f, 0.456f, 0.406f};
        const float stdv[3] = {0.229f, 0.224f, 0.225f};
        if (tid < 256) {
#pragma unroll
            for (int c = 0; c < 3; ++c) lut[c * 256 + tid] = (half_t)(((float)tid / 255.0f - mean[c]) / stdv[c]);
        }
    }

    const int quads = p.W >> 2;  // 4 pixels = 12 bytes = 3 dwords (W % 4 == 0 on this path)
    // raw rows travel HBM -> registers -> LDS, two chunks ahead: the registers of chunks c+1 and c+2 are in flight while chunk c is
    // resampled (one chunk ahead left the waves waiting for memory: a chunk's taps take about 1 us, a load 2-3 us)
    constexpr int MAXIT = (PRE_CHUNK * 512 + PRE_NT - 1) / PRE_NT;  // PRE_CHUNK * (W / 4) / PRE_NT <= MAXIT  <=>  W <= 2048 (checked by the launcher)
    uint32_t stg0[MAXIT][3], stg1[MAXIT][3];
    auto fetch = [&](int r0, uint32_t (&stg)[MAXIT][3]) {
        const int nr = min(PRE_CHUNK, nrows - r0);
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int i = tid + it * PRE_NT;
            if (i < nr * quads) {
                const int rr = (int)__umulhi((unsigned)i, p.quads_magic), q = i - rr * quads;  // i / quads (exact: i * quads < 2^32)
                const uint32_t* g = (const uint32_t*)(src + (int64_t)(y_first + r0 + rr) * row_bytes) + q * 3;
                stg[it][0] = g[0]; stg[it][1] = g[1]; stg[it][2] = g[2];
            }
        }
    };
    auto chunk = [&](int r0, uint32_t (&stg)[MAXIT][3]) {   // stage chunk r0 from its registers, refill them with chunk r0 + 2, resample
        const int nr = min(PRE_CHUNK, nrows - r0);
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int i = tid + it * PRE_NT;
            if (i < nr * quads) {
                const int rr = (int)__umulhi((unsigned)i, p.quads_magic), q = i - rr * quads;
                const uint32_t d0 = stg[it][0], d1 = stg[it][1], d2 = stg[it][2];
                uint4 px;
                px.x = d0 & 0xffffffu;
                px.y = ((d0 >> 24) | (d1 << 8)) & 0xffffffu;
                px.z = ((d1 >> 16) | (d2 << 16)) & 0xffffffu;
                px.w = d2 >> 8;
                *(uint4*)(raw + rr * raw_pitch + q * 4) = px;
            }
        }
        __syncthreads();
        if (r0 + 2 * PRE_CHUNK < nrows) fetch(r0 + 2 * PRE_CHUNK, stg);
        for (int rr = 0; rr < nr; ++rr) {
            const uint32_t* row = raw + rr * raw_pitch;
#pragma unroll
            for (int s = 0; s < PRE_COLS; ++s) {
                const int ox = tid + PRE_NT * s;
                if (ox < PRE_OUT) {
                    // 8-bit level x 22-bit coefficient (bilinear: never negative): the 24-bit multiply-add runs at full rate, a
                    // 32-bit integer multiply at a quarter of it (this loop was 135 v_mul_lo_u32 in the kernel's ISA)
                    uint32_t s0 = 1u << 21, s1 = 1u << 21, s2 = 1u << 21;
#pragma unroll
                    for (int t = 0; t < KSX; ++t) {
                        const uint32_t px = row[xmin[s] + t];
                        const uint32_t kt = (uint32_t)kx[s][t];
                        s0 = __umul24(px & 0xffu, kt) + s0;
                        s1 = __umul24((px >> 8) & 0xffu, kt) + s1;
                        s2 = __umul24(px >> 16, kt) + s2;
                    }
                    rows[(r0 + rr) * PRE_OUT + ox] = (uint32_t)clip8_22((int)s0) | ((uint32_t)clip8_22((int)s1) << 8) | ((uint32_t)clip8_22((int)s2) << 16);
                }
            }
        }
        __syncthreads();
    };
    fetch(0, stg0);
    if (PRE_CHUNK < nrows) fetch(PRE_CHUNK, stg1);
    for (int r0 = 0; r0 < nrows; r0 += 2 * PRE_CHUNK) {
        chunk(r0, stg0);
        if (r0 + PRE_CHUNK < nrows) chunk(r0 + PRE_CHUNK, stg1);
    }

    // vertical taps + normalise, 4 pixels per thread; input channel order is BGR, output RGB0
    for (int idx = tid; idx < PRE_TY * (PRE_OUT / 4); idx += PRE_NT) {
        const int ty = idx / (PRE_OUT / 4), g = idx - ty * (PRE_OUT / 4);
        const int oy = oy0 + ty;
        if (oy >= PRE_OUT) break;
        const int* yt = ytab + ty * (2 + KSY);
        const int ymin = yt[0] - y_first;
        uint32_t acc[4][3];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[q][c] = 1u << 21;
#pragma unroll
        for (int t = 0; t < KSY; ++t) {
            const uint32_t kv = (uint32_t)yt[2 + t];
            // rows past the window carry kv == 0; clamp the row index so the read stays inside the buffer
            const int r = min(ymin + t, nrows - 1);
            const uint4 px = *(const uint4*)(rows + r * PRE_OUT + g * 4);
            const uint32_t pv[4] = {px.x, px.y, px.z, px.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc[q][0] = __umul24(pv[q] & 0xffu, kv) + acc[q][0];
                acc[q][1] = __umul24((pv[q] >> 8) & 0xffu, kv) + acc[q][1];
                acc[q][2] = __umul24(pv[q] >> 16, kv) + acc[q][2];
            }
        }
        half8 o01, o23;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const half_t h0 = lut[clip8_22((int)acc[q][2])];        // R
            const half_t h1 = lut[256 + clip8_22((int)acc[q][1])];  // G
            const half_t h2 = lut[512 + clip8_22((int)acc[q][0])];  // B
            if (q < 2) { o01[q * 4] = h0; o01[q * 4 + 1] = h1; o01[q * 4 + 2] = h2; o01[q * 4 + 3] = (half_t)0.f; }
            else { o23[(q - 2) * 4] = h0; o23[(q - 2) * 4 + 1] = h1; o23[(q - 2) * 4 + 2] = h2; o23[(q - 2) * 4 + 3] = (half_t)0.f; }
        }
        // pixel (oy+3, 4g+3) of the ring-3 row: 8-byte aligned only, so two 8-byte-aligned 16-byte halves are not guaranteed
        half_t* dst = p.out + (((int64_t)img * 646 + oy + 3) * 646 + g * 4 + 3) * 4;
        *(half4*)(dst) = half4{o01[0], o01[1], o01[2], o01[3]};
        *(half4*)(dst + 4) = half4{o01[4], o01[5], o01[6], o01[7]};
        *(half4*)(dst + 8) = half4{o23[0], o23[1], o23[2], o23[3]};
        *(half4*)(dst + 12) = half4{o23[4], o23[5], o23[6], o23[7]};
    }
}

// Reference-format network input ([n,3,640,640] float32, already normalised) -> ring-3 NHWC4 fp16.
__global__ void nchw_to_input_kernel(const float* __restrict__ x, half_t* __restrict__ out, int n) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)n * 640 * 640;
    if (idx >= total) return;
    const int img = (int)(idx / (640 * 640));
    const int rem = (int)(idx - (int64_t)img * 640 * 640);
    const int y = rem / 640, xx = rem - y * 640;
    const float* base = x + (int64_t)img * 3 * 640 * 640 + rem;
    half4 o = {(half_t)base[0], (half_t)base[640 * 640], (half_t)base[2 * 640 * 640], (half_t)0.f};
    *(half4*)(out + (((int64_t)img * 646 + y + 3) * 646 + xx + 3) * 4) = o;
}

// Generic max-pool over ring-padded NHWC fp16, 8 channels (16 bytes) per thread.  Inputs are post-ReLU
// (>= 0) so the zero ring is equivalent to -inf padding (every window holds a real pixel).
struct PoolParams {
    const half_t* in;
    half_t* out;
    int n, c, ho, wo;
    int in_hp, in_wp, in_y0, in_x0;  // in_y0 = ring_in - pad
    int out_hp, out_wp, out_ring;
    int kh, kw, sh, sw;
};

__global__ void maxpool_kernel(const PoolParams p) {
    const int cg = p.c >> 3;
    const int64_t total = (int64_t)p.n * p.ho * p.wo * cg;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int g = (int)(idx % cg);
        int64_t t = idx / cg;
        const int ox = (int)(t % p.wo);
        t /= p.wo;
        const int oy = (int)(t % p.ho);
        const int img = (int)(t / p.ho);
        half8 m;
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = (half_t)(-65504.f);
        for (int r = 0; r < p.kh; ++r)
            for (int s = 0; s < p.kw; ++s) {
                const half8 v = *(const half8*)(p.in + (((int64_t)img * p.in_hp + oy * p.sh + p.in_y0 + r) * p.in_wp +
                                                         ox * p.sw + p.in_x0 + s) * p.c + g * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) m[e] = v[e] > m[e] ? v[e] : m[e];
            }
        *(half8*)(p.out + (((int64_t)img * p.out_hp + oy + p.out_ring) * p.out_wp + ox + p.out_ring) * p.c + g * 8) = m;
    }
}

}  // namespace

int vtd_launch_preprocess(const uint8_t* frames, int n, int H, int W, half_t* out, const int* xb, const int* xk, int ksx,
                          const int* yb, const int* yk, int ksy, int max_rows, hipStream_t stream) {
    PreParams p{frames, out, xb, xk, yb, yk, H, W, ksx, ksy, max_rows, 0u};
    if ((W & 3) == 0 && W <= 2048 && ksy == 5 && (ksx == 5 || ksx == 7)) {  // 720p / 1080p class sizes: the widened-pixel fast path
        const int lds_fast = (max_rows * PRE_OUT + PRE_CHUNK * (W + ksx) + PRE_TY * (2 + ksy)) * 4 + 3 * 256 * 2;
        p.quads_magic = (unsigned)(((1ull << 32) + (W / 4) - 1) / (W / 4));
        if (lds_fast <= 160 * 1024) {
            static bool attr5 = false, attr7 = false;
            if (ksx == 5) {
                if (!attr5) {
                    VTD_HIP_CHECK(hipFuncSetAttribute((const void*)preprocess_fast_kernel<5, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                    attr5 = true;
                }
                hipLaunchKernelGGL((preprocess_fast_kernel<5, 5>), dim3(PRE_OUT / PRE_TY, n), dim3(PRE_NT), lds_fast, stream, p);
            } else {
                if (!attr7) {
                    VTD_HIP_CHECK(hipFuncSetAttribute((const void*)preprocess_fast_kernel<7, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                    attr7 = true;
                }
                hipLaunchKernelGGL((preprocess_fast_kernel<7, 5>), dim3(PRE_OUT / PRE_TY, n), dim3(PRE_NT), lds_fast, stream, p);
            }
            return -(int)hipGetLastError();
        }
    }
    const int lds = ((max_rows * PRE_OUT * 3 + 15) & ~15) + PRE_CHUNK * ((W * 3 + 15) & ~15);
    if (lds > 160 * 1024) return -1010;
    static bool attr_done = false;
    if (!attr_done) {
        VTD_HIP_CHECK(hipFuncSetAttribute((const void*)preprocess_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
    }
    hipLaunchKernelGGL(preprocess_kernel, dim3(PRE_OUT / PRE_TY, n), dim3(256), lds, stream, p);
    return -(int)hipGetLastError();
}

int vtd_launch_nchw_to_input(const float* x, half_t* out, int n, hipStream_t stream) {
    const int64_t total = (int64_t)n * 640 * 640;
    hipLaunchKernelGGL(nchw_to_input_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, x, out, n);
    return -(int)hipGetLastError();
}

int vtd_launch_maxpool(const TensorDesc& in, const TensorDesc& out, int n, int kh, int kw, int sh, int sw, int pad_h, int pad_w,
                       hipStream_t stream) {
    if (in.c != out.c || (in.c & 7)) return -1020;
    if (in.ring < pad_h || in.ring < pad_w) return -1021;
    if ((out.h - 1) * sh - pad_h + kh > in.h + (in.hp - in.h - in.ring) || (out.w - 1) * sw - pad_w + kw > in.w + (in.wp - in.w - in.ring))
        return -1022;
    PoolParams p{in.ptr, out.ptr, n, in.c, out.h, out.w, in.hp, in.wp, in.ring - pad_h, in.ring - pad_w,
                 out.hp, out.wp, out.ring, kh, kw, sh, sw};
    const int64_t total = (int64_t)n * out.h * out.w * (in.c >> 3);
    const int blocks = (int)((total + 255) / 256 < 256 * 16 ? (total + 255) / 256 : 256 * 16);
    hipLaunchKernelGGL(maxpool_kernel, dim3(blocks), dim3(256), 0, stream, p);
    return -(int)hipGetLastError();
}
