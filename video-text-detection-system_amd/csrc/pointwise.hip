// FPN lateral connection on the 128-channel trunk map in one streaming launch: 1x1 convolution 128 -> 256 (+ bias) plus the
// nearest-2x upsampled coarser pyramid level (app/ml/models/text_detector.py: FeaturePyramidNetwork inner_blocks + top-down add;
// SURVEY B.3).  As an implicit GEMM this layer has K = 128: two K-steps between a prologue and an epilogue, 183 MB moved for
// 10 GFLOP -- the generic kernel ran it at 2.6 TB/s with the pixel tile re-read by each of its four channel tiles.  Here
//
//   * persistent workgroups of 4 waves; wave b owns output channels 64b..64b+63 and keeps their 64 x 128 weights in registers
//     (16 fragments) for the whole launch;
//   * pixels arrive by LDS-DMA, 64 at a time (16 KB, one 256-byte line per pixel, 16-byte chunks XOR-swizzled by pixel & 15 so
//     the 16 rows of a ds_read_b128 lane group fall on 16 different bank quads), through a 4-tile ring: every input line is
//     fetched once per launch and up to three tiles per workgroup are in flight;
//   * the MFMA row -> channel assignment is permuted so that a lane ends up with channels 32k + 8fq .. + 7 (k = 0, 1) of its
//     four pixels: bias, residual and fp16 convert run on registers and the tile leaves as 16-byte stores, 64 contiguous
//     bytes per pixel and instruction -- no LDS round trip for the output.
//
// Arithmetic order is that of conv_igemm.hip (K ascending in steps of 32 into a zero accumulator; then + bias; then + residual; one
// fp16 rounding), so the two kernels give bit-identical maps and kernel choice cannot change results.
#include "vtd_common.h"

namespace {

struct PointwiseParams {
    const half_t* in;     // [n][in_hp][in_wp][128] fp16, ring in_ring
    const half_t* wgt;    // [256][128] fp16 (BN / bias folded as for conv_igemm)
    const float* bias;    // [256]
    const half_t* res;    // [n][res_hp][res_wp][256] or null: added at (oy >> 1, ox >> 1)
    half_t* out;          // [n][out_hp][out_wp][256], ring out_ring
    int n, h, w, in_hp, in_wp, in_ring, out_hp, out_wp, out_ring, res_hp, res_wp, res_ring;
    int tiles;            // n*h*w / 64
    uint64_t magic_w, magic_hw;  // ceil(2^40 / w), ceil(2^40 / (h*w))
};

__device__ __forceinline__ int pw_div(int m, uint64_t magic) { return (int)(((uint64_t)(uint32_t)m * magic) >> 40); }
template <int N>
__device__ __forceinline__ void pw_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

constexpr int PW_NST = 4, PW_TILE_BYTES = 64 * 256;
constexpr int PW_LDS = PW_NST * PW_TILE_BYTES + 256 * 4;

template <bool RES>
__global__ __launch_bounds__(256, 2) void pointwise128_kernel(const PointwiseParams p) {
    extern __shared__ __attribute__((aligned(16))) char pw_smem[];
    float* const bias_lds = (float*)(pw_smem + PW_NST * PW_TILE_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int blk = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int hw = p.h * p.w;

    // weights: fragment (i, s) of this wave = MFMA rows r = 0..15 of block i, K = 32s + 8fq ..; row r of block i stands for channel
    // 64 blk + 32 (i >> 1) + 8 (r >> 2) + 4 (i & 1) + (r & 3), so accumulator acc[i][j][e] of lane (fr, fq) is channel
    // 64 blk + 32 (i >> 1) + 8 fq + 4 (i & 1) + e of pixel 16 j + fr
    half8 wf[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ch = 64 * blk + 32 * (i >> 1) + 8 * (fr >> 2) + 4 * (i & 1) + (fr & 3);
#pragma unroll
        for (int s = 0; s < 4; ++s) wf[i][s] = *(const half8*)(p.wgt + ch * 128 + 32 * s + 8 * fq);
    }
    bias_lds[tid] = p.bias[tid];
    __syncthreads();
    // have the weights' wait HERE: left to the compiler it lands in front of the loop's first MFMA as s_waitcnt vmcnt(0), which
    // there also sits out every LDS-DMA and residual read in flight, once per tile
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int s = 0; s < 4; ++s) asm volatile("" ::"v"(wf[i][s]));

    auto tile_origin = [&](int tile, int& img, int& oy0, int& ox0) {
        const int m0 = __builtin_amdgcn_readfirstlane(tile) * 64;
        img = pw_div(m0, p.magic_hw);
        const int rem = m0 - img * hw;
        oy0 = pw_div(rem, p.magic_w);
        ox0 = rem - oy0 * p.w;
    };
    // wave blk brings pieces blk, blk+4, blk+8, blk+12 of a tile (4 pixels x 256 bytes each)
    const int d_px = lane >> 4;
    auto issue = [&](int tile, int stage) {
        int img, oy0, ox0;
        tile_origin(tile, img, oy0, ox0);
        const half_t* base = p.in + (((int64_t)img * p.in_hp + oy0 + p.in_ring) * p.in_wp + p.in_ring) * 128;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int piece = blk + 4 * k;
            const int px = piece * 4 + d_px;
            int ox = ox0 + px, row = 0;
            if (ox >= p.w) { ox -= p.w; row = p.in_wp; }
            const half_t* src = base + (row + ox) * 128 + (((lane & 15) ^ (px & 15)) << 3);
            __builtin_amdgcn_global_load_lds((const VTD_AS1 void*)src, (VTD_AS3 void*)(pw_smem + stage * PW_TILE_BYTES + piece * 1024), 16, 0, 0);
        }
    };

    const int first = blockIdx.x, stride = gridDim.x;
    if (first >= p.tiles) return;
    const int nt = (p.tiles - first + stride - 1) / stride;
#pragma unroll
    for (int a = 0; a < PW_NST - 1; ++a)
        if (a < nt) issue(first + a * stride, a);

    int st = 0;
    for (int k = 0; k < nt; ++k) {
        const int tile = first + k * stride;
        int img, oy0, ox0;
        tile_origin(tile, img, oy0, ox0);
        // this lane's four pixels: output address and (RES) the coarser level's pixel
        int64_t o_off[4];
        half8 rv[4][2];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int ox = ox0 + j * 16 + fr, oy = oy0;
            if (ox >= p.w) { ox -= p.w; oy += 1; }
            o_off[j] = (((int64_t)img * p.out_hp + oy + p.out_ring) * p.out_wp + ox + p.out_ring) * 256 + 64 * blk + 8 * fq;
            if constexpr (RES) {
                const half_t* r = p.res + (((int64_t)img * p.res_hp + (oy >> 1) + p.res_ring) * p.res_wp + (ox >> 1) + p.res_ring) * 256 + 64 * blk + 8 * fq;
                rv[j][0] = *(const half8*)r;
                rv[j][1] = *(const half8*)(r + 32);
            }
        }
        // Tile k has landed once nothing older than the younger tiles' loads (and this tile's residual reads) is outstanding.  The
        // previous tile's stores are younger than tile k's loads as well: they can only lengthen this wait.
        if (k + PW_NST - 1 <= nt) pw_wait_vmcnt<4 * (PW_NST - 2) + (RES ? 8 : 0)>(); else pw_wait_vmcnt<RES ? 8 : 0>();
        __builtin_amdgcn_s_barrier();

        const char* sb = pw_smem + st * PW_TILE_BYTES;
        floatx4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
        // fragment (j, s): pixel 16j + fr (row & 15 = fr), logical chunk 4s + fq
        half8 af[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) af[0][j] = *(const half8*)(sb + (j * 16 + fr) * 256 + ((fq ^ fr) << 4));
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s < 3) {
#pragma unroll
                for (int j = 0; j < 4; ++j) af[(s + 1) & 1][j] = *(const half8*)(sb + (j * 16 + fr) * 256 + (((4 * (s + 1) + fq) ^ fr) << 4));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i][s], af[s & 1][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }

        // epilogue on registers: (acc + bias) + residual -> fp16 -> two 16-byte stores per pixel
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const floatx4 b0 = *(const floatx4*)(bias_lds + 64 * blk + 32 * kk + 8 * fq), b1 = *(const floatx4*)(bias_lds + 64 * blk + 32 * kk + 8 * fq + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                half8 hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v0 = acc[2 * kk][j][e] + b0[e], v1 = acc[2 * kk + 1][j][e] + b1[e];
                    if constexpr (RES) {
                        v0 += (float)rv[j][kk][e];
                        v1 += (float)rv[j][kk][4 + e];
                    }
                    hv[e] = (half_t)v0;
                    hv[4 + e] = (half_t)v1;
                }
                *(half8*)(p.out + o_off[j] + 32 * kk) = hv;
            }
        }
        // the stage tile k-1 was read from is free since this iteration's barrier: tile k + PW_NST - 1 goes there.  (Issued here and not
        // at the top: the compiler guards the residual registers with s_waitcnt vmcnt(0), which would also sit out these loads.)
        if (k + PW_NST - 1 < nt) issue(tile + (PW_NST - 1) * stride, st == 0 ? PW_NST - 1 : st - 1);
        st = st + 1 == PW_NST ? 0 : st + 1;
    }
    pw_wait_vmcnt<0>();  // no LDS-DMA may still be in flight when the workgroup's LDS is handed on
}

}  // namespace

bool vtd_pointwise128_supported(const ConvParams& c) {
    // 1x1, stride 1, 128 -> 256 channels, NHWC output, optional nearest-2x upsampled residual; rows of at least one tile
    const int plain = c.flags & ~EPI_RESIDUAL;
    return !c.plist && !c.in2 && !c.pool_pw && c.K == 128 && c.in_c == 128 && c.cout == 256 && c.out_c == 256 && c.cin_steps == 2 && c.kw == 1 && c.stride == 1 && plain == 0 &&
           (!(c.flags & EPI_RESIDUAL) || (c.res && c.res_shift == 1)) && c.wo >= 64 && (c.ho * c.wo) % 64 == 0 && c.M % (c.ho * c.wo) == 0 &&
           (uint64_t)c.M * (uint64_t)(c.ho * c.wo) < (1ull << 40);
}

int vtd_launch_pointwise128(const ConvParams& c, hipStream_t stream) {
    if (!vtd_pointwise128_supported(c)) return -2401;
    PointwiseParams p;
    p.in = c.in; p.wgt = c.wgt; p.bias = c.bias; p.res = (c.flags & EPI_RESIDUAL) ? c.res : nullptr; p.out = (half_t*)c.out;
    p.n = c.M / (c.ho * c.wo); p.h = c.ho; p.w = c.wo;
    p.in_hp = c.in_hp; p.in_wp = c.in_wp; p.in_ring = c.in_y0;  // pad 0: in_y0 = ring
    if (c.in_y0 != c.in_x0) return -2402;
    p.out_hp = c.out_hp; p.out_wp = c.out_wp; p.out_ring = c.out_ring;
    p.res_hp = c.res_hp; p.res_wp = c.res_wp; p.res_ring = c.res_ring;
    p.tiles = c.M / 64;
    p.magic_w = ((1ull << 40) + c.wo - 1) / c.wo;
    p.magic_hw = ((1ull << 40) + (uint64_t)c.ho * c.wo - 1) / ((uint64_t)c.ho * c.wo);
    const int grid = p.tiles < 512 ? p.tiles : 512;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)pointwise128_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, PW_LDS);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)pointwise128_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, PW_LDS);
        if (e != hipSuccess) return -(int)e;
        attr_set = true;
    }
    if (p.res) hipLaunchKernelGGL(pointwise128_kernel<true>, dim3(grid), dim3(256), PW_LDS, stream, p);
    else hipLaunchKernelGGL(pointwise128_kernel<false>, dim3(grid), dim3(256), PW_LDS, stream, p);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}
