// DBNet stem in one launch: conv 7x7/s2/p3 (3 -> 64) + BatchNorm + ReLU + max-pool 3x3/s2/p1.
// Replaces backbone[0..3] of the reference (app/ml/models/text_detector.py:25-33, torchvision resnet conv1/bn1/relu/maxpool).
//
// Why its own kernel: as an implicit GEMM the stem is the one layer with a tiny K (147) and the largest M, so the
// generic kernel spends its time re-fetching the same input pixels for 49 taps and writing a 420 MB map that the
// max-pool immediately reads back.  Here
//   * a workgroup owns a 7x8 block of POOLED pixels = a 15x17 patch of conv outputs (255 of the 256 GEMM rows of its
//     tile; 14 % halo recompute) and stages the 35x40-pixel NHWC4 input patch (11 KB) in LDS once;
//   * K is walked as 7 kernel rows x (8 taps x 4 ch): the 8 halfs a lane feeds to v_mfma_f32_16x16x32_f16 are 2
//     neighbouring input pixels = one aligned 16-byte LDS read, no im2col buffer is ever built;
//   * all 64x224 folded weights live in registers (28 fragments / lane) for the whole persistent loop;
//   * the conv tile goes to LDS as fp16 after bias+ReLU (out-of-image positions forced to 0: post-ReLU values are
//     >= 0, so a zero is as good as -inf for the pool's padding), the 3x3/s2 max runs from LDS and only the pooled
//     160x160x64 map is written to HBM (105 MB instead of 420 + 105 MB).
// The next tile's patch is fetched into registers before the MFMA loop and parked in the other LDS buffer after it.
#include "vtd_common.h"

namespace {

constexpr int SP_PT_ROWS = 7, SP_PT_COLS = 8;         // pooled pixels per tile
constexpr int SP_CT_ROWS = 15, SP_CT_COLS = 17;       // conv outputs per tile
constexpr int SP_PATCH_ROWS = 37, SP_PATCH_COLS = 40; // input pixels staged (35 used + 2 rows touched by the idle 256th row)
constexpr int SP_PATCH_BYTES = SP_PATCH_ROWS * SP_PATCH_COLS * 8;
constexpr int SP_CT_PITCH = 144;                      // bytes per conv pixel in LDS (64 ch fp16 + 16 pad)
constexpr int SP_WREG_ROWS = 5;                       // kernel rows whose weights stay in registers
constexpr int SP_UNITS = 35 * 20;                     // 16-byte units of a patch that are actually loaded

struct StemPoolParams {
    const half_t* in;    // [n][in_hp][in_wp][4] fp16, ring 3
    const half_t* w;     // [7 ky][4 cout tiles][64 lanes][8] fp16: lane (fr,fq) of tile i holds cout 16i+fr, taps 2fq,2fq+1, 4 ch
    const float* bias;   // [64]
    half_t* out;         // [n][out_hp][out_wp][64], ring out_ring
    int n, in_hp, in_wp, conv_h, conv_w, pool_h, pool_w, out_hp, out_wp, out_ring;
    int tiles_x, tiles_y, total_tiles;
};

__device__ __forceinline__ void sp_tile_coords(const StemPoolParams& p, int tile, int& img, int& py0, int& px0) {
    const int per_img = p.tiles_x * p.tiles_y;
    img = tile / per_img;
    const int r = tile - img * per_img;
    const int ty = r / p.tiles_x;
    py0 = ty * SP_PT_ROWS;
    px0 = (r - ty * p.tiles_x) * SP_PT_COLS;
}

// unit u of the patch of tile (img, py0, px0): 2 input pixels, zero outside the padded image
__device__ __forceinline__ uint4 sp_fetch_unit(const StemPoolParams& p, int img, int py0, int px0, int u) {
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (u < SP_UNITS) {
        const int prow = u / 20, c2 = u - prow * 20;
        const int iy = 4 * py0 - 2 + prow, ix = 4 * px0 - 2 + 2 * c2;
        if (iy >= 0 && iy < p.in_hp && ix >= 0 && ix + 1 < p.in_wp)
            v = *(const uint4*)(p.in + (((int64_t)img * p.in_hp + iy) * p.in_wp + ix) * 4);
    }
    return v;
}

__global__ __launch_bounds__(256, 2) void stem_pool_kernel(const StemPoolParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char patch[2][SP_PATCH_BYTES];
    __shared__ __attribute__((aligned(16))) unsigned char ctile[256 * SP_CT_PITCH];
    __shared__ __attribute__((aligned(16))) unsigned char wlds[(7 - SP_WREG_ROWS) * 4 * 64 * 16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;

    // folded weights: resident in registers for every tile this workgroup processes
    // (kernel rows 0..SP_WREG_ROWS-1 in VGPRs, the rest in LDS, so that the 256-VGPR budget of 2 waves/SIMD holds)
    half8 wreg[SP_WREG_ROWS][4];
#pragma unroll
    for (int ky = 0; ky < SP_WREG_ROWS; ++ky)
#pragma unroll
        for (int i = 0; i < 4; ++i) wreg[ky][i] = *(const half8*)(p.w + ((ky * 4 + i) * 64 + lane) * 8);
    for (int u = tid; u < (7 - SP_WREG_ROWS) * 4 * 64; u += 256)
        *(half8*)(wlds + u * 16) = *(const half8*)(p.w + (SP_WREG_ROWS * 4 * 64 + u) * 8);
    // this lane's four GEMM rows (conv pixels of the patch) -> LDS byte offset of tap (ky=0, kx=2fq)
    int a_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int pix = wv * 64 + j * 16 + fr;
        const int c_row = pix / SP_CT_COLS, c_col = pix - c_row * SP_CT_COLS;
        a_off[j] = ((2 * c_row) * SP_PATCH_COLS + 2 * c_col + 2 * fq) * 8;
    }

    // rows 35, 36 of both patch buffers are only read by the idle 256th GEMM row: keep them defined
    for (int i = tid; i < 2 * 2 * SP_PATCH_COLS * 2; i += 256) {
        const int b = i / (2 * SP_PATCH_COLS * 2), r = i - b * (2 * SP_PATCH_COLS * 2);
        *(uint32_t*)(patch[b] + 35 * SP_PATCH_COLS * 8 + r * 4) = 0u;
    }

    int tile = blockIdx.x;
    if (tile >= p.total_tiles) return;
    int img, py0, px0;
    sp_tile_coords(p, tile, img, py0, px0);
    {
        uint4 v[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) v[k] = sp_fetch_unit(p, img, py0, px0, tid + k * 256);
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (tid + k * 256 < SP_UNITS) *(uint4*)(patch[0] + (tid + k * 256) * 16) = v[k];
    }
    __syncthreads();

    int cur = 0;
    for (; tile < p.total_tiles; tile += gridDim.x) {
        // prefetch the next tile's patch into registers (lands while the matrix cores work)
        const int ntile = tile + gridDim.x;
        const bool has_next = ntile < p.total_tiles;
        int nimg = 0, npy0 = 0, npx0 = 0;
        uint4 nv[3];
        if (has_next) {
            sp_tile_coords(p, ntile, nimg, npy0, npx0);
#pragma unroll
            for (int k = 0; k < 3; ++k) nv[k] = sp_fetch_unit(p, nimg, npy0, npx0, tid + k * 256);
        }

        floatx4 acc[4][4];  // [cout tile][pixel fragment]
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
        const unsigned char* pb = patch[cur];
        // software pipeline by hand: the fragments of kernel row ky+1 are read while row ky multiplies; the scheduling
        // barriers stop the compiler from hoisting all 28 LDS reads (which would blow the register budget)
        half8 af[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) af[0][j] = *(const half8*)(pb + a_off[j]);
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) {
            if (ky < 6) {
#pragma unroll
                for (int j = 0; j < 4; ++j) af[(ky + 1) & 1][j] = *(const half8*)(pb + a_off[j] + (ky + 1) * (SP_PATCH_COLS * 8));
            }
            if (ky < SP_WREG_ROWS) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[ky < SP_WREG_ROWS ? ky : 0][i], af[ky & 1][j], acc[i][j], 0, 0, 0);
            } else {
                half8 wf[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) wf[i] = *(const half8*)(wlds + (((ky - SP_WREG_ROWS) * 4 + i) * 64 + lane) * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], af[ky & 1][j], acc[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }

        if (has_next) {
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (tid + k * 256 < SP_UNITS) *(uint4*)(patch[cur ^ 1] + (tid + k * 256) * 16) = nv[k];
        }

        // bias + ReLU, zero outside the conv map, fp16 -> LDS conv tile (lane: 4 consecutive channels of one pixel)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pix = wv * 64 + j * 16 + fr;
            const int c_row = pix / SP_CT_COLS, c_col = pix - c_row * SP_CT_COLS;
            const int cy = 2 * py0 - 1 + c_row, cx = 2 * px0 - 1 + c_col;
            const bool valid = cy >= 0 && cy < p.conv_h && cx >= 0 && cx < p.conv_w;
            unsigned char* dst = ctile + (wv * 64 + j * 16 + fr) * SP_CT_PITCH + fq * 8;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const floatx4 b4 = *(const floatx4*)(p.bias + i * 16 + fq * 4);
                half4 hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = acc[i][j][e] + b4[e];
                    hv[e] = (half_t)((valid && v > 0.f) ? v : 0.f);
                }
                *(half4*)(dst + i * 32) = hv;
            }
        }
        __syncthreads();

        // 3x3/s2 max over the conv tile: item = (pooled pixel, 8-channel group)
        for (int item = tid; item < SP_PT_ROWS * SP_PT_COLS * 8; item += 256) {
            const int q = item >> 3, cg = item & 7;
            const int qy = q >> 3, qx = q & 7;
            const int py = py0 + qy, px = px0 + qx;
            if (py < p.pool_h && px < p.pool_w) {
                const unsigned char* src = ctile + ((2 * qy) * SP_CT_COLS + 2 * qx) * SP_CT_PITCH + cg * 16;
                half8 m = *(const half8*)src;
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int s = 0; s < 3; ++s) {
                        if (r == 0 && s == 0) continue;
                        const half8 v = *(const half8*)(src + (r * SP_CT_COLS + s) * SP_CT_PITCH);
#pragma unroll
                        for (int e = 0; e < 8; ++e) m[e] = v[e] > m[e] ? v[e] : m[e];
                    }
                *(half8*)(p.out + (((int64_t)img * p.out_hp + py + p.out_ring) * p.out_wp + px + p.out_ring) * 64 + cg * 8) = m;
            }
        }
        __syncthreads();
        cur ^= 1;
        img = nimg; py0 = npy0; px0 = npx0;
    }
}

}  // namespace

// Fragment-ordered weight image for stem_pool_kernel from the folded [64][7][7][3] weights (see StemPoolParams::w).
void vtd_stem_pool_pack_weights(const float* w_folded /* [64][3][7][7] already scaled by BN */, half_t* packed /* 7*4*64*8 */) {
    for (int ky = 0; ky < 7; ++ky)
        for (int i = 0; i < 4; ++i)
            for (int lane = 0; lane < 64; ++lane) {
                const int fr = lane & 15, fq = lane >> 4, co = i * 16 + fr;
                for (int e = 0; e < 8; ++e) {
                    const int kx = 2 * fq + (e >> 2), c = e & 3;
                    float v = 0.f;
                    if (kx < 7 && c < 3) v = w_folded[((co * 3 + c) * 7 + ky) * 7 + kx];
                    packed[((ky * 4 + i) * 64 + lane) * 8 + e] = (half_t)v;
                }
            }
}

int vtd_launch_stem_pool(const TensorDesc& in, const TensorDesc& out, const half_t* w_packed, const float* bias, int n,
                         hipStream_t stream) {
    if (in.c != 4 || in.ring != 3 || out.c != 64 || (in.h & 3) || (in.w & 3) || out.h != in.h / 4 || out.w != in.w / 4 ||
        in.hp != in.h + 6 || in.wp != in.w + 6 || (in.wp & 1) || n <= 0 || n > in.n || n > out.n)
        return -2001;
    if (out.w % SP_PT_COLS) return -2002;  // column tiles must be whole (the row direction may end in a partial tile)
    StemPoolParams p;
    p.in = in.ptr; p.w = w_packed; p.bias = bias; p.out = out.ptr;
    p.n = n; p.in_hp = in.hp; p.in_wp = in.wp; p.conv_h = in.h / 2; p.conv_w = in.w / 2; p.pool_h = out.h; p.pool_w = out.w;
    p.out_hp = out.hp; p.out_wp = out.wp; p.out_ring = out.ring;
    p.tiles_x = out.w / SP_PT_COLS;
    p.tiles_y = (out.h + SP_PT_ROWS - 1) / SP_PT_ROWS;
    p.total_tiles = n * p.tiles_x * p.tiles_y;
    const int grid = p.total_tiles < 512 ? p.total_tiles : 512;  // 2 resident workgroups per CU, persistent over tiles
    hipLaunchKernelGGL(stem_pool_kernel, dim3(grid), dim3(256), 0, stream, p);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

