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
//   * all 64x224 folded weights live in registers (28 fragments / lane) for the whole persistent loop; the BN shift starts
//     the accumulators;
//   * the conv tile goes to LDS as fp16 after bias+ReLU (out-of-image positions forced to 0: post-ReLU values are
//     >= 0, so a zero is as good as -inf for the pool's padding), the 3x3/s2 max runs from LDS and only the pooled
//     160x160x64 map is written to HBM (105 MB instead of 420 + 105 MB).
// Patches arrive by LDS-DMA through a 3-deep ring, two tiles ahead of the maths (round 1 fetched one tile ahead into registers
// and consumed it right after the MFMA loop: the launch ran at the load latency), and each XCD sweeps its own eighth of the tiles.
#include <cstdlib>
#include "vtd_common.h"

namespace {

constexpr int SP_PT_ROWS = 7, SP_PT_COLS = 8;         // pooled pixels per tile
constexpr int SP_CT_COLS = 17;                        // conv outputs per tile: 15 rows x 17 columns
constexpr int SP_PATCH_COLS = 40;                     // input pixels per staged patch row (35 rows used)
constexpr int SP_PIECES = 12;                         // 1 KB LDS-DMA pieces per patch: 35 rows x 20 units of 16 bytes = 700 units
constexpr int SP_PATCH_BYTES = SP_PIECES * 1024;      // (the 256th, idle GEMM row reads rows 35, 36: still inside)
constexpr int SP_NST = 3;                             // patch ring depth
constexpr int SP_CT_PITCH = 144;                      // bytes per conv pixel in LDS (64 ch fp16 + 16 pad)
constexpr int SP_LDS = SP_NST * SP_PATCH_BYTES + 256 * SP_CT_PITCH + 256;

struct StemPoolParams {
    const half_t* in;    // [n][in_hp][in_wp][4] fp16, ring 3
    const half_t* w;     // [7 ky][4 cout tiles][64 lanes][8] fp16: lane (fr,fq) of tile i holds cout sp_chan(i, fr), taps 2fq,2fq+1, 4 ch
    const float* bias;   // [64]
    half_t* out;         // [n][out_hp][out_wp][64], ring out_ring
    int n, in_hp, in_wp, conv_h, conv_w, pool_h, pool_w, out_hp, out_wp, out_ring;
    int tiles_x, tiles_y, total_tiles;
};

// Post-ReLU fp16 values are >= +0, so their bit patterns order like int16: the pool (and the ReLU itself: every negative half, -0
// included, is a negative int16) run on v_pk_max_i16.  The float form costs twice the instructions: llvm's maxnum quiets each
// operand first (v_pk_max_f16 x, x, x), 68 instead of 32 per pooled item.
typedef short short4v __attribute__((ext_vector_type(4)));
typedef short short8v __attribute__((ext_vector_type(8)));
__device__ __forceinline__ half8 sp_max(half8 a, half8 b) {
    return __builtin_bit_cast(half8, __builtin_elementwise_max(__builtin_bit_cast(short8v, a), __builtin_bit_cast(short8v, b)));
}

// MFMA row fr of cout tile i carries this output channel: permuted so that a lane's accumulators of a tile PAIR are 8 consecutive
// channels -- acc[2k][.][e] and acc[2k+1][.][e] of lane group fq = channels 32k + 8fq + e and 32k + 8fq + 4 + e -- and the conv tile
// goes to LDS as one 16-byte chunk per (pixel, pair) instead of two 8-byte halves.  ds_write_b64 is served in groups of 16 lanes:
// 16 pixels at the 144-byte pitch put two lanes on every bank (SQ_LDS_BANK_CONFLICT: 40 % of this kernel's LDS cycles, round-3
// counters); ds_write_b128's groups of 8 lanes cover 8 pixels x 16 bytes = all 32 banks once.
__host__ __device__ constexpr int sp_chan(int tile, int row) { return 32 * (tile >> 1) + 8 * (row >> 2) + 4 * (tile & 1) + (row & 3); }

template <int N>
__device__ __forceinline__ void sp_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// workgroup barrier that orders LDS only (__syncthreads would also sit out every global load and store in flight)
__device__ __forceinline__ void sp_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

__device__ __forceinline__ void sp_tile_coords(const StemPoolParams& p, int tile, int& img, int& py0, int& px0) {
    const int per_img = p.tiles_x * p.tiles_y;
    img = tile / per_img;
    const int r = tile - img * per_img;
    const int ty = r / p.tiles_x;
    py0 = ty * SP_PT_ROWS;
    px0 = (r - ty * p.tiles_x) * SP_PT_COLS;
}

// DBG: timing-only variants for tools/stem_experiment.sh (compile-time, so the product instantiation DBG = 0 is untouched;
// results of the others are garbage): 1 no pool, 2 neither conv tile nor pool, 3 one of seven kernel rows, 4 no loads, 5 no stores,
// 6 MFMA phase at raised priority, 7 = 6 + the CU's second workgroup starts late, 8 late start only (6..8 compute the real result)
template <int DBG>
__global__ __launch_bounds__(256, 2) void stem_pool_kernel(const StemPoolParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sp_smem[];
    unsigned char* const ring = sp_smem;                                 // [SP_NST][SP_PATCH_BYTES]
    unsigned char* const ctile = sp_smem + SP_NST * SP_PATCH_BYTES;      // [256][SP_CT_PITCH]
    float* const bias_lds = (float*)(ctile + 256 * SP_CT_PITCH);         // [64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;

    // Tile order.  Consecutive workgroup ids sit on different XCDs, each with its own L2: with tile = blockIdx + k * grid the
    // neighbours that share a patch halo were always fetched through different L2s (PMC: 221 MB read for a 105 MB input).  Each
    // XCD now owns a contiguous eighth of the tiles and its workgroups sweep it side by side, so halos are L2 hits.
    int first, stride, last;
    if ((gridDim.x & 7) == 0) {
        const int xcd = blockIdx.x & 7, per = gridDim.x >> 3;
        const int lo = (int)(((int64_t)p.total_tiles * xcd) >> 3);
        last = (int)(((int64_t)p.total_tiles * (xcd + 1)) >> 3);
        first = lo + (blockIdx.x >> 3);
        stride = per;
    } else {
        first = blockIdx.x; stride = gridDim.x; last = p.total_tiles;
    }
    if (first >= last) return;
    const int nt = (last - first + stride - 1) / stride;

    // folded weights: all 28 fragments stay in registers for every tile this workgroup processes
    half8 wreg[7][4];
#pragma unroll
    for (int ky = 0; ky < 7; ++ky)
#pragma unroll
        for (int i = 0; i < 4; ++i) wreg[ky][i] = *(const half8*)(p.w + ((ky * 4 + i) * 64 + lane) * 8);
    if (tid < 64) bias_lds[tid] = p.bias[tid];
    // this lane's four GEMM rows (conv pixels of the patch) -> LDS byte offset of tap (ky=0, kx=2fq)
    int a_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int pix = wv * 64 + j * 16 + fr;
        const int c_row = pix / SP_CT_COLS, c_col = pix - c_row * SP_CT_COLS;
        a_off[j] = ((2 * c_row) * SP_PATCH_COLS + 2 * c_col + 2 * fq) * 8;
    }
    // The patch arrives by LDS-DMA, SP_NST-1 tiles ahead: wave wv brings pieces wv, wv+4, wv+8 (64 units of 2 pixels each, laid
    // out [patch row][20 units]).  Units past the 700th repeat the last row.  Coordinates outside the padded image are CLAMPED,
    // not zero-filled: such pixels only ever feed conv outputs outside the conv map, which the epilogue replaces by 0, or the
    // zero weights of the eighth tap -- and whatever is read instead is a finite fp16 of the same tensor.
    int u_row[3], u_col[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int u = (wv + 4 * k) * 64 + lane;
        u = u < 700 ? u : 699;
        u_row[k] = u / 20;
        u_col[k] = 2 * (u - u_row[k] * 20);
    }
    auto issue = [&](int tile, int stage) {
        if constexpr (DBG == 4) return;
        int img, py0, px0;
        sp_tile_coords(p, __builtin_amdgcn_readfirstlane(tile), img, py0, px0);
        const half_t* base = p.in + (int64_t)img * p.in_hp * p.in_wp * 4;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            int iy = 4 * py0 - 2 + u_row[k], ix = 4 * px0 - 2 + u_col[k];
            iy = iy < 0 ? 0 : iy > p.in_hp - 1 ? p.in_hp - 1 : iy;
            ix = ix < 0 ? 0 : ix > p.in_wp - 2 ? p.in_wp - 2 : ix;
            __builtin_amdgcn_global_load_lds((const VTD_AS1 void*)(base + (iy * p.in_wp + ix) * 4),
                                             (VTD_AS3 void*)(ring + stage * SP_PATCH_BYTES + (wv + 4 * k) * 1024), 16, 0, 0);
        }
    };

    if constexpr (DBG == 7 || DBG == 8) {  // the second workgroup of a CU (by dispatch order: a guess, for speed only) starts half a tile late
        if (((blockIdx.x >> 3) & 32) != 0) { __builtin_amdgcn_s_sleep(60); }
    }
    issue(first, 0);
    if (nt > 1) { issue(first + stride, 1); sp_wait_vmcnt<3>(); } else sp_wait_vmcnt<0>();
    sp_lds_barrier();

    int st = 0;
    for (int k = 0; k < nt; ++k) {
        const int tile = first + k * stride;
        int img, py0, px0;
        sp_tile_coords(p, __builtin_amdgcn_readfirstlane(tile), img, py0, px0);
        // the stage tile k-1 was read from is free (barrier at the end of the last iteration): tile k+2 goes there
        if (k + 2 < nt) issue(tile + 2 * stride, st == 0 ? SP_NST - 1 : st - 1);

        floatx4 acc[4][4];  // [cout tile][pixel fragment]; the BN shift rides in the accumulator
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = *(const floatx4*)(bias_lds + sp_chan(i, fq * 4));
        const unsigned char* pb = ring + st * SP_PATCH_BYTES;
        // software pipeline by hand: the fragments of kernel row ky+1 are read while row ky multiplies; the scheduling
        // barriers stop the compiler from hoisting all 28 LDS reads (which would blow the register budget)
        half8 af[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) af[0][j] = *(const half8*)(pb + a_off[j]);
        if constexpr (DBG == 6 || DBG == 7) __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) {
            if (ky < 6) {
#pragma unroll
                for (int j = 0; j < 4; ++j) af[(ky + 1) & 1][j] = *(const half8*)(pb + a_off[j] + (ky + 1) * (SP_PATCH_COLS * 8));
            }
            if (DBG == 3 && ky > 0) { for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(af[ky & 1][j])); continue; }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[ky][i], af[ky & 1][j], acc[i][j], 0, 0, 0);
            // pin the order inside the block: the four reads of the next kernel row FIRST, then the 16 MFMAs (left alone, hipcc put
            // the reads behind 12 of them and waited for them 4 MFMAs later: ~100 exposed cycles per kernel row)
            if (ky < 6) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
            __builtin_amdgcn_sched_barrier(0);
        }

        if constexpr (DBG == 6 || DBG == 7) __builtin_amdgcn_s_setprio(0);
        // ReLU (packed, after the fp16 convert), zero outside the conv map, -> LDS conv tile (lane: 4 consecutive channels of one pixel)
        if constexpr (DBG == 2) {  // neither the conv tile nor the pool: keep the accumulators alive
            float t = 0.f;
            for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
            if (t == 123.456f) p.out[0] = (half_t)t;
            if (k + 2 < nt) sp_wait_vmcnt<3>(); else sp_wait_vmcnt<0>();
            sp_lds_barrier();
            st = st + 1 == SP_NST ? 0 : st + 1;
            continue;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pix = wv * 64 + j * 16 + fr;
            const int c_row = pix / SP_CT_COLS, c_col = pix - c_row * SP_CT_COLS;
            const int cy = 2 * py0 - 1 + c_row, cx = 2 * px0 - 1 + c_col;
            const bool valid = cy >= 0 && cy < p.conv_h && cx >= 0 && cx < p.conv_w;
            unsigned char* dst = ctile + pix * SP_CT_PITCH + fq * 16;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                typedef float float2v __attribute__((ext_vector_type(2)));
                typedef _Float16 half2v __attribute__((ext_vector_type(2)));
                const floatx4 a = acc[2 * k][j], b = acc[2 * k + 1][j];
                const half2v h0 = __builtin_convertvector((float2v{a[0], a[1]}), half2v);  // v_cvt_pk_f16_f32
                const half2v h1 = __builtin_convertvector((float2v{a[2], a[3]}), half2v);
                const half2v h2 = __builtin_convertvector((float2v{b[0], b[1]}), half2v);
                const half2v h3 = __builtin_convertvector((float2v{b[2], b[3]}), half2v);
                half8 hv = half8{h0[0], h0[1], h1[0], h1[1], h2[0], h2[1], h3[0], h3[1]};
                hv = __builtin_bit_cast(half8, __builtin_elementwise_max(__builtin_bit_cast(short8v, hv), short8v{0, 0, 0, 0, 0, 0, 0, 0}));
                if (!valid) hv = half8{0, 0, 0, 0, 0, 0, 0, 0};
                *(half8*)(dst + k * 64) = hv;   // channels 32k + 8fq .. + 7: chunk 4k + fq of the pixel
            }
        }
        sp_lds_barrier();

        // 3x3/s2 max over the conv tile: item = (pooled pixel, 8-channel group)
        // A wave's 64 items are one pooled row: 8 pixels x 8 channel groups; waves 0..2 own two rows (7 rows in all).  Which lane takes
        // which matters: ds_read_b128 is served in four fixed 16-lane groups and with cg = lane & 7 each group hit its banks 2.75
        // times (11 LDS cycles per read, SQ_LDS_BANK_CONFLICT); with this assignment the 16 chunks of a group fall on 16 different
        // bank quads.  All 9 window reads of an item leave before the first max: left to itself hipcc waited after every one to three
        // reads, six dependent LDS round trips per item (~2.2 k of the ~9 k cycles of a tile).
        if constexpr (DBG != 1) {
            const int cg = (lane >> 2) & 7;
            const int qx = ((lane >> 5) & 1) | ((lane & 3) << 1);
            const bool two = wv < 3;  // wave-uniform: pooled rows wv and wv + 4
            const unsigned char* src0 = ctile + ((2 * wv) * SP_CT_COLS + 2 * qx) * SP_CT_PITCH + cg * 16;
            const int px = px0 + qx;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                if (it == 1 && !two) break;
                const unsigned char* src = src0 + it * (8 * SP_CT_COLS * SP_CT_PITCH);
                half8 v[9];
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) v[r * 3 + c] = *(const half8*)(src + (r * SP_CT_COLS + c) * SP_CT_PITCH);
                __builtin_amdgcn_sched_group_barrier(0x100, 9, 0);  // all nine reads in flight before the first max
                half8 m = sp_max(v[0], v[1]);
#pragma unroll
                for (int q = 2; q < 9; ++q) m = sp_max(m, v[q]);
                const int py = py0 + wv + 4 * it;
                if (DBG == 5) asm volatile("" ::"v"(m));
                else if (py < p.pool_h && px < p.pool_w)
                    *(half8*)(p.out + (((int64_t)img * p.out_hp + py + p.out_ring) * p.out_wp + px + p.out_ring) * 64 + cg * 8) = m;
            }
        }
        // Tile k+1 must be in LDS for everyone after the barrier below.  Loads land in order, so it has landed once at most the 3
        // loads of tile k+2 are outstanding; the pooled stores just issued share the counter and retire out of order with loads,
        // but they are YOUNGER than tile k+1's loads and can only lengthen this wait.
        if (k + 2 < nt) sp_wait_vmcnt<3>(); else sp_wait_vmcnt<0>();
        sp_lds_barrier();
        st = st + 1 == SP_NST ? 0 : st + 1;
    }
}

}  // namespace

// Fragment-ordered weight image for stem_pool_kernel from the folded [64][7][7][3] weights (see StemPoolParams::w).
void vtd_stem_pool_pack_weights(const float* w_folded /* [64][3][7][7] already scaled by BN */, half_t* packed /* 7*4*64*8 */) {
    for (int ky = 0; ky < 7; ++ky)
        for (int i = 0; i < 4; ++i)
            for (int lane = 0; lane < 64; ++lane) {
                const int fr = lane & 15, fq = lane >> 4, co = sp_chan(i, fr);
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
    void (*kern)(const StemPoolParams) = stem_pool_kernel<0>;
#ifdef VTD_STEM_EXPERIMENT
    if (const char* e = getenv("VTD_STEM_DEBUG")) {
        switch (atoi(e)) {
            case 1: kern = stem_pool_kernel<1>; break;
            case 2: kern = stem_pool_kernel<2>; break;
            case 3: kern = stem_pool_kernel<3>; break;
            case 4: kern = stem_pool_kernel<4>; break;
            case 5: kern = stem_pool_kernel<5>; break;
            case 6: kern = stem_pool_kernel<6>; break;
            case 7: kern = stem_pool_kernel<7>; break;
            case 8: kern = stem_pool_kernel<8>; break;
            default: break;
        }
    }
    {
        hipError_t ea = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SP_LDS);
        if (ea != hipSuccess) return -(int)ea;
    }
#else
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t ea = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SP_LDS);
        if (ea != hipSuccess) return -(int)ea;
        attr_set = true;
    }
#endif
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), SP_LDS, stream, p);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

