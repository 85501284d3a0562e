// DB-head tail in one launch: ConvTranspose2d(64->64,k2,s2) + BatchNorm + ReLU + ConvTranspose2d(64->1,k2,s2) + sigmoid
// (app/ml/models/text_detector.py:59-75, layers 3..7 of probability_head / threshold_head) on the 160x160x64 head map,
// writing the 640x640 fp32 map.  Neither the 320x320x64 intermediate nor an fp32 staging tile exists anywhere:
//
//   * persistent workgroups of 4 waves; wave b owns sub-position b = (ky,kx) of the first transposed conv for the 64
//     pixels of the tile: a 64(ch) x 64(px) x 64(K) GEMM whose weights (8 fragments) stay in registers for the whole
//     launch, while the pixels (one 128-byte line each) arrive by LDS-DMA through a 5-tile ring, four tiles ahead of the
//     maths, so the launch is bound by HBM bandwidth and not by its latency;
//   * the BN shift starts the accumulators, ReLU is applied to them after the fp16 repack to fp16 and fed -- as they sit in the
//     lanes -- into a second MFMA against the (K-permuted, row-replicated) 64->4 weights of the last transposed conv, so
//     the 64-long dot products run on the matrix pipe as well and every lane group ends up with all four logits;
//   * lane group q applies the sigmoid to logit q; the 4x256 fp32 output rows of the tile are assembled in LDS (4 KB,
//     double buffered, one barrier per tile) and leave as 16-byte row-contiguous stores.
#include "vtd_common.h"

namespace {

struct HeadTailParams {
    const half_t* in;     // [n][hp][wp][64] fp16 (post BN+ReLU head map), ring `ring`
    const half_t* w1;     // [4 blk][4 i][2 s][64 lanes][8] fp16: ConvT1 (BN folded) fragment order
    const float* bias1;   // [4 blk][64] fp32 (BN shift)
    const half_t* w2;     // [2 s][64 lanes][8] fp16: ConvT2 weights, K permuted to the accumulator layout, rows o = row & 3
    float* out;           // [n][4h][4w] fp32
    float b2;
    int n, h, w, hp, wp, ring;
    int tiles;            // n*h*w / 64
    uint64_t magic_w;     // ceil(2^40 / w)
    uint64_t magic_hw;    // ceil(2^40 / (h*w))
};

__device__ __forceinline__ int ht_div(int m, uint64_t magic) { return (int)(((uint64_t)(uint32_t)m * magic) >> 40); }

template <int N>
__device__ __forceinline__ void ht_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Pixel ring: HT_NST tiles of 64 pixels x 128 bytes, filled by LDS-DMA HT_NST-1 tiles ahead of the maths.  Round 1 read the pixel
// fragments straight into registers one tile ahead: 8 KB in flight per workgroup, and the launch ran at the memory LATENCY
// (1.7 TB/s of reads, PMC) with each of the four waves fetching the same lines.  Now every line is fetched once per workgroup,
// HT_NST-1 tiles are in flight per workgroup (three workgroups per CU: 96 KB), and the waves read their fragments from LDS (16-byte chunks of a pixel row
// XOR-swizzled by (pixel >> 1) & 7 on the way in, so the 16 pixels of a ds_read_b128 lane group hit distinct banks).
constexpr int HT_NST = 5, HT_WG_PER_CU = 3, HT_TILE_BYTES = 64 * 128;
constexpr int HT_LDS = HT_NST * HT_TILE_BYTES + 2 * 4 * 256 * 4;

__global__ __launch_bounds__(256, HT_WG_PER_CU) void head_tail_kernel(const HeadTailParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float (*const otile)[4][256] = (float (*)[4][256])(smem + HT_NST * HT_TILE_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int blk = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4, lrow = lane >> 3;
    const int hw = p.h * p.w;

    half8 w1f[4][2], w2f[2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int s = 0; s < 2; ++s) w1f[i][s] = *(const half8*)(p.w1 + ((((blk * 4 + i) * 2 + s) * 64) + lane) * 8);
#pragma unroll
    for (int s = 0; s < 2; ++s) w2f[s] = *(const half8*)(p.w2 + (s * 64 + lane) * 8);
    floatx4 b1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) b1[i] = *(const floatx4*)(p.bias1 + blk * 64 + i * 16 + fq * 4);

    // tile -> (image, row, column) of its first pixel on the scalar unit; a lane's pixel is at most one row wrap away (w >= 64)
    auto tile_origin = [&](int tile, int& img, int& oy0, int& ox0) {
        const int m0 = __builtin_amdgcn_readfirstlane(tile) * 64;
        img = ht_div(m0, p.magic_hw);
        const int rem = m0 - img * hw;
        oy0 = ht_div(rem, p.magic_w);
        ox0 = rem - oy0 * p.w;
    };
    // wave `blk` brings pieces blk and blk + 4 of the tile (8 pixels x 128 bytes each): two LDS-DMA instructions per wave and tile
    auto issue = [&](int tile, int stage) {
        int img, oy0, ox0;
        tile_origin(tile, img, oy0, ox0);
        const half_t* base = p.in + (((int64_t)img * p.hp + oy0 + p.ring) * p.wp + p.ring) * 64;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int piece = blk + 4 * i;
            const int r = piece * 8 + lrow;
            int ox = ox0 + r, row = 0;
            if (ox >= p.w) { ox -= p.w; row = p.wp; }
            const half_t* src = base + (row + ox) * 64 + (((lane & 7) ^ ((r >> 1) & 7)) << 3);
            __builtin_amdgcn_global_load_lds((const VTD_AS1 void*)src, (VTD_AS3 void*)(smem + stage * HT_TILE_BYTES + piece * 1024), 16, 0, 0);
        }
    };
    // thread = (output row r of the 4, tile pixel px): 16 bytes, contiguous over px within an image row
    auto store_out = [&](int tile, int ob) {
        const int r = tid >> 6, px = tid & 63;
        int img, oy, ox;
        tile_origin(tile, img, oy, ox);
        ox += px;
        if (ox >= p.w) { ox -= p.w; oy += 1; }
        const floatx4 v = *(const floatx4*)&otile[ob][r][4 * px];
        *(floatx4*)(p.out + ((int64_t)img * 4 * p.h + 4 * oy + r) * (4 * p.w) + 4 * ox) = v;
    };

    const int first = blockIdx.x, stride = gridDim.x;
    if (first >= p.tiles) return;
    const int nt = (p.tiles - first + stride - 1) / stride;  // tiles of this workgroup
#pragma unroll
    for (int a = 0; a < HT_NST - 1; ++a)
        if (a < nt) issue(first + a * stride, a);

    // fragment af[j][s] = channels 32s + 8fq.. of tile pixel 16j + fr: row (16j + fr) * 128, chunk (4s + fq) ^ ((fr >> 1) & 7)
    const int key = (fr >> 1) & 7;
    const int foff0 = fr * 128 + ((fq ^ key) << 4), foff1 = fr * 128 + (((4 + fq) ^ key) << 4);
    const int ky = blk >> 1, kx = blk & 1;
    int ob = 0, st = 0;
    for (int k = 0; k < nt; ++k) {
        const int tile = first + k * stride;
        // Loads land in order: with HT_NST-2 younger tiles behind it, tile k has landed when at most 2 (HT_NST-2) of this wave's loads
        // are outstanding.  (Output stores share the counter and retire out of order with the loads: every store that could
        // still be outstanding here is YOUNGER than tile k's loads, so it can only make this wait longer, never shorter than needed.)
        if (k + HT_NST - 1 <= nt) ht_wait_vmcnt<2 * (HT_NST - 2)>(); else ht_wait_vmcnt<0>();
        // this wave's otile writes of tile k-1 must have reached LDS before anyone reads them behind the barrier: a raw s_barrier
        // orders nothing by itself (found by the repeatability soak in tests/test_gpu_tuning.py: rare stale 16-byte pieces under load)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // tile k is in LDS for every wave; everyone is done with tile k-1's fragments and its otile
        if (k > 0) store_out(tile - stride, ob ^ 1);
        if (k + HT_NST - 1 < nt) issue(tile + (HT_NST - 1) * stride, st == 0 ? HT_NST - 1 : st - 1);

        const char* sb = smem + st * HT_TILE_BYTES;
        half8 cur[4][2];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            cur[j][0] = *(const half8*)(sb + j * 2048 + foff0);
            cur[j][1] = *(const half8*)(sb + j * 2048 + foff1);
        }
        floatx4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = b1[i];  // the BN shift rides in the accumulator
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1f[i][s], cur[j][s], acc[i][j], 0, 0, 0);

        // lane now holds channels 16i + 4fq + e of pixel 16j + fr.  ReLU, fp16, and straight into the second MFMA:
        // K-step s of that product takes element t of the lane as channel 16(2s + (t>>2)) + 4fq + (t&3).
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            floatx4 lg = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                half8 a;
#pragma unroll
                for (int t = 0; t < 8; ++t) a[t] = (half_t)acc[2 * s + (t >> 2)][j][t & 3];
                a = __builtin_elementwise_max(a, half8{0, 0, 0, 0, 0, 0, 0, 0});  // ReLU, packed
                lg = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2f[s], a, lg, 0, 0, 0);
            }
            // rows 4fq + r of the result all carry logit r: lane group fq finishes logit fq
            const float l = (fq == 0 ? lg[0] : fq == 1 ? lg[1] : fq == 2 ? lg[2] : lg[3]) + p.b2;
            const float pr = 1.f / (1.f + __expf(-l));
            otile[ob][2 * ky + (fq >> 1)][4 * (j * 16 + fr) + 2 * kx + (fq & 1)] = pr;
        }
        ob ^= 1;
        st = st + 1 == HT_NST ? 0 : st + 1;
    }
    __syncthreads();
    store_out(first + (nt - 1) * stride, ob ^ 1);
}

}  // namespace

// w1_gemm: [256 rows = blk*64 + cout][64 K] fp16 (build_convt layout, BN scale folded) -> fragment order
void vtd_head_tail_pack_w1(const half_t* w1_gemm, half_t* packed /* 4*4*2*64*8 */) {
    for (int blk = 0; blk < 4; ++blk)
        for (int i = 0; i < 4; ++i)
            for (int s = 0; s < 2; ++s)
                for (int lane = 0; lane < 64; ++lane) {
                    const int fr = lane & 15, fq = lane >> 4;
                    for (int t = 0; t < 8; ++t)
                        packed[((((blk * 4 + i) * 2 + s) * 64) + lane) * 8 + t] = w1_gemm[(size_t)(blk * 64 + i * 16 + fr) * 64 + s * 32 + fq * 8 + t];
                }
}

// w2: [4 outputs o = ky*2+kx][64 ch] fp32 -> [2 s][64 lanes][8] fp16, row (lane & 15) carries output (row & 3)
void vtd_head_tail_pack_w2(const float* w2, half_t* packed /* 2*64*8 */) {
    for (int s = 0; s < 2; ++s)
        for (int lane = 0; lane < 64; ++lane) {
            const int row = lane & 15, fq = lane >> 4;
            for (int t = 0; t < 8; ++t) {
                const int ch = 16 * (2 * s + (t >> 2)) + 4 * fq + (t & 3);
                packed[(s * 64 + lane) * 8 + t] = (half_t)w2[(row & 3) * 64 + ch];
            }
        }
}

int vtd_launch_head_tail(const TensorDesc& in, const half_t* w1, const float* bias1, const half_t* w2, float b2, float* out, int n,
                         hipStream_t stream) {
    if (in.c != 64 || in.w < 64 || n <= 0 || n > in.n || !w1 || !bias1 || !w2 || !out) return -2101;
    const int64_t M = (int64_t)n * in.h * in.w;
    if ((in.h * in.w) % 64 || (uint64_t)M * (uint64_t)(in.h * in.w) >= (1ull << 40)) return -2102;  // tiles never straddle images; magic division exact
    HeadTailParams p;
    p.in = in.ptr; p.w1 = w1; p.bias1 = bias1; p.w2 = w2; p.out = out; p.b2 = b2;
    p.n = n; p.h = in.h; p.w = in.w; p.hp = in.hp; p.wp = in.wp; p.ring = in.ring;
    p.tiles = (int)(M / 64);
    p.magic_w = ((1ull << 40) + in.w - 1) / in.w;
    p.magic_hw = ((1ull << 40) + (uint64_t)in.h * in.w - 1) / ((uint64_t)in.h * in.w);
    const int grid = p.tiles < 256 * HT_WG_PER_CU ? p.tiles : 256 * HT_WG_PER_CU;
    static bool attr_set = false;  // engines are built under a lock; the attribute is per function, not per stream
    if (!attr_set) {
        hipError_t ea = hipFuncSetAttribute((const void*)head_tail_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, HT_LDS);
        if (ea != hipSuccess) return -(int)ea;
        attr_set = true;
    }
    hipLaunchKernelGGL(head_tail_kernel, dim3(grid), dim3(256), HT_LDS, stream, p);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}
