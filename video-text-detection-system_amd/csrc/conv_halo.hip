// 3x3 / stride 1 / pad 1 convolution (+ folded BN, residual, ReLU) with HALO-TILE staging of the input.
// Serves the stride-1 3x3 layers of the ResNet stages (reference: torchvision BasicBlock / Bottleneck conv3x3 under
// app/ml/models/text_detector.py:25-33) and of the CRNN stack (text_recognizer.py:15-30).
//
// Why a second convolution kernel.  conv_igemm.hip gathers an im2col row per tap: every input pixel travels global -> LDS
// nine times.  Timing experiments show that this gather -- LDS-DMA issue + the per-CU TA path, not HBM and
// not LDS reads -- is what bounds the layers with few output channels.  Here a workgroup owns a TH x TW block of output
// pixels (256 GEMM rows) and stages the (TH+2) x (TW+2) input halo of one 64-channel chunk ONCE; the nine taps are nine
// K-steps that read the same LDS image at shifted rows.  Per K-step only the weights (BN x 128 bytes) still stream in.
//   * LDS image of the halo: one 128-byte row per pixel, 16-byte chunks XOR-swizzled by (row & 6) on the source side of
//     the LDS-DMA: with that pattern a ds_read_b128 of 16 CONSECUTIVE halo rows starting at ANY row is conflict-free in all
//     four hardware lane groups (brute-forced over the linear swizzles; the ((row >> 1) & 7) pattern of the aligned
//     tiles in conv_igemm.hip is 2-way to 4-way for starts that are not multiples of 4, i.e. for most tap shifts);
//   * weights: 3-stage ring filled by LDS-DMA, counted vmcnt + one s_barrier per K-step (as conv_igemm.hip);
//     the halo of the next channel chunk is fetched piecewise during the current chunk's K-steps;
//   * waves: 4 x (BN/64), each a 64 x 64 sub-tile (16 accumulator fragments), v_mfma_f32_16x16x32_f16, weights as the first
//     operand so a lane owns 4 consecutive channels; epilogue through an fp32 LDS tile -> 16-byte NHWC stores.
#include <cstdio>
#include <cstdlib>
#include "vtd_common.h"

namespace {

struct HaloParams {
    const half_t* in;     // [n][in_hp][in_wp][cin] fp16, ring in_ring >= 1 (zero)
    const half_t* wgt;    // [cout][9*cin] fp16, K = (r*3+s)*cin + c  (conv_igemm's packing)
    const float* bias;    // [cout]
    const half_t* res;    // optional residual, [n][res_hp][res_wp][cout]
    half_t* out;          // [n][out_hp][out_wp][cout]
    int n, h, w, cin, cout;
    int in_hp, in_wp, in_ring, out_hp, out_wp, out_ring, res_hp, res_wp, res_ring;
    int tiles_x, tiles_y, tiles_n, relu;
    unsigned long long* stamps;  // debug (VTD_HALO_STAMPS=1): 4 s_memtime stamps per workgroup
    int dbg;                     // instrumented build only (-DVTD_CONV_EXPERIMENT, VTD_C64_DEBUG): 1 = every pixel block of a workgroup reads and
                                 // writes ONE block's addresses (operands out of L2, results wrong): the layer-1 kernel without its HBM traffic
};

template <int N>
__device__ __forceinline__ void hl_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int BN, int TW>
__global__ __launch_bounds__((BN / 64) * 256) void conv_halo_kernel(const HaloParams p) {
    constexpr int WN = BN / 64, NW = 4 * WN, NT = NW * 64, TH = 256 / TW;
    constexpr int HWD = TW + 2, HROWS = (TH + 2) * HWD, HPIECES = (HROWS + 7) / 8, HBYTES = HPIECES * 1024;
    constexpr int HPW = (HPIECES + NW - 1) / NW;  // halo pieces per wave
    constexpr int HPS = (HPW + 7) / 8;            // of which issued per K-step while the previous chunk computes (taps 0..7)
    constexpr int BSTAGE = BN * 128, BPW = BN / 8 / NW;
    constexpr int EPI_ROW = BN * 4 + 16;
    static_assert(BPW == 2, "two weight pieces per wave and K-step");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int nchunks = p.cin >> 6;
    const int hb_stride = nchunks > 1 ? HBYTES : 0;
    char* const bring = smem + (nchunks > 1 ? 2 : 1) * HBYTES;

    // ---- XCD-aware tile assignment (bijective for any grid size); the N tiles of one pixel block are neighbours
    const int nblk = gridDim.x, b = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = b & 7;
    int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b >> 3);
    const int tn = tile % p.tiles_n;
    tile /= p.tiles_n;
    const int tx = tile % p.tiles_x;
    tile /= p.tiles_x;
    const int ty = tile % p.tiles_y;
    const int img = tile / p.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane >> 3;
    const int K = 9 * p.cin;

    // ---- halo loader: this lane's source element offset for each of the wave's pieces (chunk 0)
    int hoff[HPW];
#pragma unroll
    for (int k = 0; k < HPW; ++k) {
        const int piece = w + k * NW;
        int row = piece * 8 + lrow;
        row = row < HROWS ? row : HROWS - 1;
        const int hy = row / HWD, hx = row - hy * HWD;
        int gy = y0 - 1 + hy + p.in_ring, gx = x0 - 1 + hx + p.in_ring;
        gy = gy < p.in_hp ? gy : p.in_hp - 1;  // partial tiles: stay inside the allocation (those outputs are masked)
        gx = gx < p.in_wp ? gx : p.in_wp - 1;
        const int c_log = (lane & 7) ^ (row & 6);
        hoff[k] = ((img * p.in_hp + gy) * p.in_wp + gx) * p.cin + c_log * 8;
    }
    auto issue_halo = [&](int k, int chunk, int buf) {
        const int piece = w + k * NW;
        if (piece < HPIECES)
            __builtin_amdgcn_global_load_lds((const VTD_AS1 void*)(p.in + hoff[k] + chunk * 64),
                                             (VTD_AS3 void*)(smem + buf * hb_stride + piece * 1024), 16, 0, 0);
    };
    // ---- weight loader: rows n0 + (i*NW + w)*8 + lrow of the [cout][K] matrix, 128 bytes per K-step
    const half_t* bsrc[BPW];
#pragma unroll
    for (int i = 0; i < BPW; ++i) {
        const int row = (i * NW + w) * 8 + lrow;
        const int c_log = (lane & 7) ^ ((row >> 1) & 7);
        bsrc[i] = p.wgt + (int64_t)(n0 + row) * K + c_log * 8;
    }
    auto issue_b = [&](int step, int stage) {
        const int chunk = step / 9, tap = step - chunk * 9;
        const int koff = tap * p.cin + chunk * 64;
#pragma unroll
        for (int i = 0; i < BPW; ++i)
            __builtin_amdgcn_global_load_lds((const VTD_AS1 void*)(bsrc[i] + koff),
                                             (VTD_AS3 void*)(bring + stage * BSTAGE + (i * NW + w) * 1024), 16, 0, 0);
    };

    // ---- compute state: wave (wm, wn) owns rows wm*64.. (4 tile rows when TW = 16, 2 when TW = 32) x channels wn*64..
    const int wm = w / WN, wn = w - wm * WN;
    const int fr = lane & 15, fq = lane >> 4;
    int hbase[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = wm * 64 + j * 16 + fr;
        hbase[j] = (m / TW) * HWD + (m % TW);
    }
    const int b_lane_off = (wn * 64 + fr) * 128;
    const int bswz = (fr >> 1) & 7;

    floatx4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int nsteps = nchunks * 9;
    unsigned long long t_start = 0, t_first = 0, t_loop = 0;
    if (p.stamps) t_start = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int k = 0; k < HPW; ++k) issue_halo(k, 0, 0);
    issue_b(0, 0);
    issue_b(1, 1);

    // K loop: channel chunks outside, the nine taps unrolled inside (tap, ring stage = tap % 3 and the halo piece indices are
    // compile-time constants)
    int s = 0;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const char* hb = smem + (chunk & 1) * hb_stride;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap, ++s) {
            constexpr int kStageOf[9] = {0, 1, 2, 0, 1, 2, 0, 1, 2};
            const int stage = kStageOf[tap];
            if (s + 1 < nsteps) hl_wait_vmcnt<BPW>(); else hl_wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();  // weights of step s (and any halo piece issued before them) landed; stage s+2 is free
            if (p.stamps && s == 0) t_first = __builtin_amdgcn_s_memtime();
            if (tap < 8 && chunk + 1 < nchunks) {
#pragma unroll
                for (int e = 0; e < HPS; ++e)
                    if (tap * HPS + e < HPW) issue_halo(tap * HPS + e, chunk + 1, (chunk + 1) & 1);
            }
            if (s + 2 < nsteps) issue_b(s + 2, kStageOf[(tap + 2) % 9]);
            const char* sb = bring + stage * BSTAGE;
            const int tapoff = (tap / 3) * HWD + (tap % 3);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                half8 af[4], bf[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int hrow = hbase[j] + tapoff;
                    af[j] = *(const half8*)(hb + hrow * 128 + (((fq + 4 * kk) ^ (hrow & 6)) << 4));
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) bf[i] = *(const half8*)(sb + b_lane_off + i * 2048 + (((fq + 4 * kk) ^ bswz) << 4));
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[i], af[j], acc[i][j], 0, 0, 0);
            }
        }
    }

    // ---- epilogue phase 1: accumulators -> fp32 tile in LDS (lane: 4 consecutive channels of pixel fr)
    __syncthreads();
    if (p.stamps) t_loop = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *(floatx4*)(smem + (wm * 64 + j * 16 + fr) * EPI_ROW + (wn * 64 + i * 16 + fq * 4) * 4) = acc[i][j];
    __syncthreads();

    // ---- phase 2: thread = 8 consecutive channels of one pixel: bias, residual, ReLU, one 16-byte store
    constexpr int CPR = BN / 8, RPI = NT / CPR;
    const int cc = tid % CPR, r0 = tid / CPR;
    const int ch = n0 + cc * 8;
    const floatx4 bias0 = *(const floatx4*)(p.bias + ch), bias1 = *(const floatx4*)(p.bias + ch + 4);
#pragma unroll 4
    for (int it = 0; it < 256 / RPI; ++it) {
        const int m = it * RPI + r0;
        const int y = y0 + m / TW, x = x0 + m % TW;
        if (y >= p.h || x >= p.w) continue;
        floatx4 v0 = *(const floatx4*)(smem + m * EPI_ROW + cc * 32);
        floatx4 v1 = *(const floatx4*)(smem + m * EPI_ROW + cc * 32 + 16);
        v0 += bias0;
        v1 += bias1;
        if (p.res) {
            const half8 rv = *(const half8*)(p.res + (((int64_t)img * p.res_hp + y + p.res_ring) * p.res_wp + x + p.res_ring) * p.cout + ch);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v0[e] += (float)rv[e]; v1[e] += (float)rv[4 + e]; }
        }
        half8 hv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = v0[e], c = v1[e];
            hv[e] = (half_t)((p.relu && a < 0.f) ? 0.f : a);
            hv[4 + e] = (half_t)((p.relu && c < 0.f) ? 0.f : c);
        }
        *(half8*)(p.out + (((int64_t)img * p.out_hp + y + p.out_ring) * p.out_wp + x + p.out_ring) * p.cout + ch) = hv;
    }
    if (p.stamps && tid == 0) {
        unsigned long long* o = p.stamps + (int64_t)blockIdx.x * 4;
        o[0] = t_start; o[1] = t_first; o[2] = t_loop; o[3] = __builtin_amdgcn_s_memtime();
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Second generation of the halo-tile kernel for any Cin (multiple of 64) and 64 output channels per workgroup, built like
// head_entry_halo256_kernel (head_entry_halo.hip): 256-pixel block, four waves of 64 pixels x 64 channels, ONE halo buffer
// (the fetch of the next channel chunk's halo is exposed; the second workgroup on the CU covers it), a 4-stage weight ring
// whose counted wait leaves only the newest stage in flight, so that the fragment reads of half K-step h+1 -- including the
// first half of the NEXT K-step -- are issued under the MFMAs of half-step h.  41 KB halo + 32 KB ring = 74 KB: two
// workgroups per CU.  Layers with more output channels run one workgroup per 64 of them on the same pixel block (the halo is
// then fetched once per 64 output channels, out of L2).
template <int TW>
__global__ __launch_bounds__(256, 2) void conv_halo64_kernel(const HaloParams p) {
    constexpr int TH = 256 / TW, HWD = TW + 2, HROWS = (TH + 2) * HWD, HPIECES = (HROWS + 7) / 8, HBYTES = HPIECES * 1024;
    constexpr int HPW = (HPIECES + 3) / 4, BSTAGE = 64 * 128, EPI_ROW = 64 * 4 + 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const hb = smem;
    char* const bring = smem + HBYTES;

    const int nblk = gridDim.x, b = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = b & 7;
    int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b >> 3);
    const int tn = tile % p.tiles_n;
    tile /= p.tiles_n;
    const int tx = tile % p.tiles_x;
    tile /= p.tiles_x;
    const int ty = tile % p.tiles_y;
    const int img = tile / p.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW, n0 = tn * 64;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane >> 3, fr = lane & 15, fq = lane >> 4;
    const int K = 9 * p.cin, nchunks = p.cin >> 6, nsteps = nchunks * 9;

    // (source offsets are recomputed per chunk: once every nine K-steps, cheaper than 11 live registers under a 256-VGPR cap)
    auto issue_halo = [&](int chunk) {
#pragma unroll
        for (int k = 0; k < HPW; ++k) {
            if (w + 4 * k >= HPIECES) break;
            int row = (w + 4 * k) * 8 + lrow;
            row = row < HROWS ? row : HROWS - 1;
            const int hy = row / HWD, hx = row - hy * HWD;
            int gy = y0 - 1 + hy + p.in_ring, gx = x0 - 1 + hx + p.in_ring;
            gy = gy < p.in_hp ? gy : p.in_hp - 1;
            gx = gx < p.in_wp ? gx : p.in_wp - 1;
            const int off = ((img * p.in_hp + gy) * p.in_wp + gx) * p.cin + ((lane & 7) ^ (row & 6)) * 8;
            __builtin_amdgcn_global_load_lds((const VTD_AS1 void*)(p.in + off + chunk * 64), (VTD_AS3 void*)(hb + (w + 4 * k) * 1024), 16, 0, 0);
        }
    };
    const half_t* bsrc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (i * 4 + w) * 8 + lrow;
        bsrc[i] = p.wgt + (int64_t)(n0 + row) * K + ((lane & 7) ^ ((row >> 1) & 7)) * 8;
    }
    auto issue_b = [&](int step, int stage) {
        const int chunk = step / 9, tap = step - chunk * 9;
        const int koff = tap * p.cin + chunk * 64;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const VTD_AS1 void*)(bsrc[i] + koff), (VTD_AS3 void*)(bring + stage * BSTAGE + (i * 4 + w) * 1024), 16, 0, 0);
    };

    int hbase[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = w * 64 + j * 16 + fr;
        hbase[j] = (m / TW) * HWD + (m % TW);
    }
    const int b_lane_off = fr * 128;
    const int bswz = (fr >> 1) & 7;
    auto load_frags = [&](int tapoff, int stage, int kk, half8 (&af)[4], half8 (&bf)[4]) {
        const char* sb = bring + stage * BSTAGE;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int hrow = hbase[j] + tapoff;
            af[j] = *(const half8*)(hb + hrow * 128 + (((fq + 4 * kk) ^ (hrow & 6)) << 4));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) bf[i] = *(const half8*)(sb + b_lane_off + i * 2048 + (((fq + 4 * kk) ^ bswz) << 4));
    };
    floatx4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    issue_halo(0);
    issue_b(0, 0);
    issue_b(1, 1);
    issue_b(2, 2);
    half8 fa[2][4], fb[2][4];
    int s = 0, stage = 0;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
#pragma unroll 1  // (unrolled, the 72 fragment addresses of a chunk are all kept live and spill)
        for (int tap = 0; tap < 9; ++tap, ++s) {
            // tap 0 follows a fresh halo (newer than every weight load): full wait; otherwise only the newest ring stage may be
            // in flight, i.e. the weights of steps s and s+1 have landed for every wave once the barrier is passed
            if (tap == 0 || s + 2 >= nsteps) hl_wait_vmcnt<0>(); else hl_wait_vmcnt<2>();
            __builtin_amdgcn_s_barrier();
            if (s + 3 < nsteps) issue_b(s + 3, (stage + 3) & 3);
            const int tapoff = (tap / 3) * HWD + (tap % 3);
            if (tap == 0) load_frags(tapoff, stage, 0, fa[0], fb[0]);  // nothing could be prefetched across the halo switch
            load_frags(tapoff, stage, 1, fa[1], fb[1]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[0][i], fa[0][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (tap < 8) load_frags(((tap + 1) / 3) * HWD + ((tap + 1) % 3), (stage + 1) & 3, 0, fa[0], fb[0]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[1][i], fa[1][j], acc[i][j], 0, 0, 0);
            if (tap < 8) {
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (tap == 8 && chunk + 1 < nchunks) {
                __builtin_amdgcn_s_barrier();  // every wave is done with this chunk's halo before the next one replaces it
                issue_halo(chunk + 1);
            }
            stage = (stage + 1) & 3;
        }
    }

    // ---- epilogue: accumulators -> fp32 LDS tile -> bias, residual, ReLU, 16-byte NHWC stores
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *(floatx4*)(smem + (w * 64 + j * 16 + fr) * EPI_ROW + (i * 16 + fq * 4) * 4) = acc[i][j];
    __syncthreads();
    const int cc = tid & 7, r0 = tid >> 3;
    const int ch = n0 + cc * 8;
    const floatx4 bias0 = *(const floatx4*)(p.bias + ch), bias1 = *(const floatx4*)(p.bias + ch + 4);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        half8 rv[4];
        if (p.res) {  // four residual rows of this thread first, then the arithmetic: one exposed latency per four rows
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = (half * 4 + q) * 32 + r0;
                int y = y0 + m / TW, x = x0 + m % TW;
                y = y < p.h ? y : p.h - 1;
                x = x < p.w ? x : p.w - 1;
                rv[q] = *(const half8*)(p.res + (((int64_t)img * p.res_hp + y + p.res_ring) * p.res_wp + x + p.res_ring) * p.cout + ch);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = (half * 4 + q) * 32 + r0;
            const int y = y0 + m / TW, x = x0 + m % TW;
            if (y >= p.h || x >= p.w) continue;
            floatx4 v0 = *(const floatx4*)(smem + m * EPI_ROW + cc * 32) + bias0;
            floatx4 v1 = *(const floatx4*)(smem + m * EPI_ROW + cc * 32 + 16) + bias1;
            if (p.res) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { v0[e] += (float)rv[q][e]; v1[e] += (float)rv[q][4 + e]; }
            }
            half8 hv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                hv[e] = (half_t)((p.relu && v0[e] < 0.f) ? 0.f : v0[e]);
                hv[4 + e] = (half_t)((p.relu && v1[e] < 0.f) ? 0.f : v1[e]);
            }
            *(half8*)(p.out + (((int64_t)img * p.out_hp + y + p.out_ring) * p.out_wp + x + p.out_ring) * p.cout + ch) = hv;
        }
    }
}

template <int TW>
int halo64_launch(const HaloParams& p, hipStream_t stream) {
    constexpr int TH = 256 / TW, HROWS = (TH + 2) * (TW + 2), HBYTES = (HROWS + 7) / 8 * 1024;
    constexpr int lds = HBYTES + 4 * 64 * 128;
    static_assert(lds >= 256 * (64 * 4 + 16) && 2 * lds <= 160 * 1024, "epilogue tile fits; two workgroups per CU");
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)conv_halo64_kernel<TW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return -(int)e;
        attr_done = true;
    }
    hipLaunchKernelGGL((conv_halo64_kernel<TW>), dim3(p.n * p.tiles_y * p.tiles_x * p.tiles_n), dim3(256), lds, stream, p);
    return -(int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------------
// 64 -> 64 channels (ResNet layer1): PERSISTENT workgroups with the whole 3x3x64x64 weight set resident in LDS.
// Measured on the kernel above (VTD_HALO_STAMPS): an LDS-DMA takes ~2 us to land, so a 3-stage weight ring leaves the matrix
// pipe waiting on memory 80 % of the time when a K-step is only 0.2 us of MFMA work.  For Cin = Cout = 64 the weights are
// 72 KB: they are loaded ONCE per workgroup, the halo of the next pixel block is fetched a whole block (~2 us of MFMAs)
// ahead into the other halo buffer, and the K loop has no barrier and no wait in it at all.  LDS: 72 KB weights + 2 x 41 KB
// halo = 154 KB, one 4-wave workgroup per CU.  The weight rows are permuted in LDS so that a lane's 16 accumulators of a pixel
// are 16 consecutive channels: residual and output move as 16-byte pieces straight from / to HBM, no staging tile.
template <int TW, bool RELU, bool RES>
__global__ __launch_bounds__(256, 1) void conv3x3_c64_persistent_kernel(const HaloParams p) {
    constexpr int TH = 256 / TW, HWD = TW + 2, HROWS = (TH + 2) * HWD, HPIECES = (HROWS + 7) / 8, HBYTES = HPIECES * 1024;
    constexpr int HPW = (HPIECES + 3) / 4;  // halo pieces per wave (every wave issues exactly HPW loads: uniform vmcnt)
    constexpr int WBYTES = 9 * 64 * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const wl = smem;                       // [9 taps][64 cout][128 B], 16-byte chunks swizzled by (cout >> 1) & 7
    char* const hbuf = smem + WBYTES;            // 2 halo buffers (+ 1 KB landing pad for the padding loads)

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane >> 3, fr = lane & 15, fq = lane >> 4;
    const int tiles_per_img = p.tiles_x * p.tiles_y, total = p.n * tiles_per_img;

    // ---- weights: 72 one-KB pieces, 18 per wave, once
#pragma unroll
    for (int k = 0; k < 18; ++k) {
        const int piece = w + 4 * k, tap = piece >> 3, row = (piece & 7) * 8 + lrow;
        const int c_log = (lane & 7) ^ ((row >> 1) & 7);
        // LDS row = MFMA fragment i (row >> 4), fragment row rho (row & 15); it carries output channel 16*(rho>>2) + 4*i + (rho&3),
        // so that the lane group q = rho>>2 ends up owning the 16 CONSECUTIVE channels 16q .. 16q+15 of its pixels
        const int cout = 16 * ((row & 15) >> 2) + 4 * (row >> 4) + (row & 3);
        __builtin_amdgcn_global_load_lds((const VTD_AS1 void*)(p.wgt + cout * 576 + tap * 64 + c_log * 8),
                                         (VTD_AS3 void*)(wl + piece * 1024), 16, 0, 0);
    }
    floatx4 bias4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bias4[i] = *(const floatx4*)(p.bias + fq * 16 + i * 4);
    // consume the bias here: otherwise its first use (accumulator init) sits inside the block loop and the compiler parks an
    // s_waitcnt vmcnt(0) there, which would wait for the NEXT block's halo in every iteration
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(bias4[i]));

    auto tile_coords = [&](int t, int& img, int& y0, int& x0) {
        img = t / tiles_per_img;
        const int r = t - img * tiles_per_img, ty = r / p.tiles_x;
        y0 = ty * TH;
        x0 = (r - ty * p.tiles_x) * TW;
    };
    // Per-lane source offset of each halo piece relative to the block's first halo pixel, and the LDS destination: both are
    // block independent (the launcher guarantees whole blocks: h % TH == 0, w % TW == 0), so a block costs one add per piece.
    int hrel[HPW];
    int hdst[HPW];
#pragma unroll
    for (int k = 0; k < HPW; ++k) {
        const int piece = w + 4 * k;
        int row = piece * 8 + lrow;
        row = row < HROWS ? row : HROWS - 1;
        const int hy = row / HWD, hx = row - hy * HWD;
        hrel[k] = (hy * p.in_wp + hx) * 64 + ((lane & 7) ^ (row & 6)) * 8;
        hdst[k] = piece < HPIECES ? WBYTES + piece * 1024 : WBYTES + 2 * HBYTES;
    }
    auto issue_halo = [&](int img, int y0, int x0, int buf) {
        const half_t* base = p.in + ((int64_t)(img * p.in_hp + y0 - 1 + p.in_ring) * p.in_wp + x0 - 1 + p.in_ring) * 64;
#pragma unroll
        for (int k = 0; k < HPW; ++k)
            __builtin_amdgcn_global_load_lds((const VTD_AS1 void*)(base + hrel[k]),
                                             (VTD_AS3 void*)(smem + hdst[k] + (w + 4 * k < HPIECES ? buf * HBYTES : 0)), 16, 0, 0);
    };
    // likewise for the residual / output pixels of this lane's four fragments
    int orel[4], rrel[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = w * 64 + j * 16 + fr;
        orel[j] = ((m / TW) * p.out_wp + (m % TW)) * 64 + fq * 16;
        rrel[j] = ((m / TW) * p.res_wp + (m % TW)) * 64 + fq * 16;
    }

    int hbase[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = w * 64 + j * 16 + fr;
        hbase[j] = (m / TW) * HWD + (m % TW);
    }
    const int bswz = (fr >> 1) & 7;

    auto load_res = [&](int img, int y0, int x0, half8 (&rv)[4][2]) {
        const half_t* rbase = p.res + ((int64_t)(img * p.res_hp + y0 + p.res_ring) * p.res_wp + x0 + p.res_ring) * 64;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            rv[j][0] = *(const half8*)(rbase + rrel[j]);
            rv[j][1] = *(const half8*)(rbase + rrel[j] + 8);
        }
    };

    // Block schedule: at the top of block b the residual(b) loads and the halo(b+1) LDS-DMA go out, then a counted wait for
    // halo(b) (issued a whole block earlier), then the barrier-free K loop, then the epilogue.  (A schedule that moves the
    // counted wait in front of the stores -- on gfx9 stores share vmcnt with loads but retire out of order with them, so the
    // wait below also sits out the previous block's store acks -- was tried and lost: the compiler answers the extra live
    // loads with s_waitcnt vmcnt(0) at the loop head.)
    int tile = blockIdx.x;
    if (tile >= total) return;  // (grid <= total: never taken)
    int img, y0, x0;
    tile_coords(tile, img, y0, x0);
    issue_halo(img, y0, x0, 0);
    int cur = 0;
    unsigned long long acc_wait = 0, acc_loop = 0, acc_epi = 0, t_a = 0, t_b = 0, t_c = 0;
    for (; tile < total; tile += gridDim.x, cur ^= 1) {
        if (p.stamps) t_a = __builtin_amdgcn_s_memtime();
        const int ntile = tile + gridDim.x;
        int nimg = img, ny0 = y0, nx0 = x0;
        half8 rv[4][2];
        if (RES) load_res(img, y0, x0, rv);
        if (ntile < total) tile_coords(ntile, nimg, ny0, nx0);
        issue_halo(nimg, ny0, nx0, cur ^ 1);  // past the last block: a harmless reload (keeps the load count uniform)
        // everything older than the HPW (+8) loads just issued has landed: weights and this block's halo
        if (RES) hl_wait_vmcnt<HPW + 8>(); else hl_wait_vmcnt<HPW>();
        __builtin_amdgcn_s_barrier();
        if (p.stamps) t_b = __builtin_amdgcn_s_memtime();

        floatx4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = bias4[i];
        const char* hb = hbuf + cur * HBYTES;
        // 18 half K-steps (tap, 32 channels), software pipelined by hand: one wave per SIMD has nobody to hide its LDS latency
        // behind, so the 8 fragment reads of half-step h+1 are issued ahead of the 16 MFMAs of half-step h (pinned below).
        auto load_frags = [&](int hs, half8 (&af)[4], half8 (&bf)[4]) {
            const int tap = hs >> 1, kk = hs & 1;
            const int tapoff = (tap / 3) * HWD + (tap % 3);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int hrow = hbase[j] + tapoff;
                af[j] = *(const half8*)(hb + hrow * 128 + (((fq + 4 * kk) ^ (hrow & 6)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) bf[i] = *(const half8*)(wl + tap * 8192 + (i * 16 + fr) * 128 + (((fq + 4 * kk) ^ bswz) << 4));
        };
        half8 fa[2][4], fb[2][4];
        load_frags(0, fa[0], fb[0]);
#pragma unroll
        for (int hs = 0; hs < 18; ++hs) {
            if (hs + 1 < 18) load_frags(hs + 1, fa[(hs + 1) & 1], fb[(hs + 1) & 1]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[hs & 1][i], fa[hs & 1][j], acc[i][j], 0, 0, 0);
            if (hs + 1 < 18) {
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);   // the 8 fragment reads of half-step h+1 go out first ...
                __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);  // ... and are in flight under the 16 MFMAs of half-step h
            }
            __builtin_amdgcn_sched_barrier(0);  // keep each half-step's prefetch inside its own half-step
        }

        // ---- epilogue: (+ residual) -> ReLU -> fp16; a lane owns channels 16fq..16fq+15 of its four pixels: 16-byte stores
        if (p.stamps) {
            asm volatile("s_nop 0" ::"v"(acc[3][3][3]));
            t_c = __builtin_amdgcn_s_memtime();
        }
        half_t* obase = p.out + ((int64_t)(img * p.out_hp + y0 + p.out_ring) * p.out_wp + x0 + p.out_ring) * 64;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                half8 hv;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float v = acc[2 * h + (k >> 2)][j][k & 3];
                    if (RES) v += (float)rv[j][h][k];
                    hv[k] = (half_t)(RELU ? fmaxf(v, 0.f) : v);
                }
                *(half8*)(obase + orel[j] + h * 8) = hv;
            }
        __builtin_amdgcn_s_barrier();  // every wave is done with this halo buffer before the block after next lands in it
        img = nimg; y0 = ny0; x0 = nx0;
        if (p.stamps) {
            const unsigned long long t_d = __builtin_amdgcn_s_memtime();
            acc_wait += t_b - t_a; acc_loop += t_c - t_b; acc_epi += t_d - t_c;
        }
    }
    if (p.stamps && tid == 0) {
        unsigned long long* o = p.stamps + (int64_t)blockIdx.x * 4;
        o[0] = 0; o[1] = acc_wait; o[2] = acc_wait + acc_loop; o[3] = acc_wait + acc_loop + acc_epi;
    }
    hl_wait_vmcnt<0>();  // no LDS-DMA may still be in flight when the workgroup's LDS is handed on
}

// ------------------------------------------------------------------------------------------------------------------------
// The same convolution with TWO wave groups per workgroup taking turns (8 waves, one workgroup per CU, weights shared).
// Stamps of the kernel above (VTD_HALO_STAMPS, per 16 x 16 pixel block): 12.5 k cycles of which the barrier-free K loop is 6.7 k
// (4.6 k of MFMA issue); the other 5.8 k -- LDS-DMA issue for the next halo (one wave per SIMD: nobody covers the ~150 cycles a
// wave is held per instruction), the residual round trip, the epilogue -- leave the matrix pipe idle.  Here group A (waves 0-3)
// multiplies its block while group B (waves 4-7, the second wave on every SIMD) finishes the previous block of its own, fetches
// its next halo and waits for it; one workgroup barrier, then they swap.  A group's halo buffer is single (it is refilled in the
// group's own service phase, after its K loop), so LDS is the same 154 KB; accumulators and residual stay in registers across
// the barrier.  No counted waits anywhere: a service phase ends with vmcnt(0), a compute phase issues nothing but the residual
// reads.  Arithmetic per output is identical to the kernel above (same taps, same order): bit-identical maps.
template <int TW, bool RELU, bool RES>
__global__ __launch_bounds__(512, 1) void conv3x3_c64_duo_kernel(const HaloParams p) {
    constexpr int TH = 256 / TW, HWD = TW + 2, HROWS = (TH + 2) * HWD, HPIECES = (HROWS + 7) / 8, HBYTES = HPIECES * 1024;
    constexpr int HPW = (HPIECES + 3) / 4;  // halo pieces per wave of a group
    constexpr int WBYTES = 9 * 64 * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const wl = smem;                       // [9 taps][64 cout][128 B], 16-byte chunks swizzled by (cout >> 1) & 7
    char* const hbuf = smem + WBYTES;            // one halo buffer per group (+ 1 KB landing pad for the padding loads)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wv >> 2, w = wv & 3;           // group, wave within the group
    const int lrow = lane >> 3, fr = lane & 15, fq = lane >> 4;
    const int tiles_per_img = p.tiles_x * p.tiles_y, total = p.n * tiles_per_img;

    // ---- weights: 72 one-KB pieces, 9 per wave, once (row permutation as in the kernel above)
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int piece = wv + 8 * k, tap = piece >> 3, row = (piece & 7) * 8 + lrow;
        const int c_log = (lane & 7) ^ ((row >> 1) & 7);
        const int cout = 16 * ((row & 15) >> 2) + 4 * (row >> 4) + (row & 3);
        __builtin_amdgcn_global_load_lds((const VTD_AS1 void*)(p.wgt + cout * 576 + tap * 64 + c_log * 8),
                                         (VTD_AS3 void*)(wl + piece * 1024), 16, 0, 0);
    }
    // (register budget: two waves per SIMD = 256 registers; the bias lives in LDS, per-block addresses are recomputed per block)
    float* const bias_lds = (float*)(smem + WBYTES + 2 * HBYTES + 1024);
    if (tid < 64) bias_lds[tid] = p.bias[tid];

    auto tile_coords = [&](int t, int& img, int& y0, int& x0) {
        img = t / tiles_per_img;
        const int r = t - img * tiles_per_img, ty = r / p.tiles_x;
        y0 = ty * TH;
        x0 = (r - ty * p.tiles_x) * TW;
    };
    auto issue_halo = [&](int t) {
        int img, y0, x0;
        tile_coords(t, img, y0, x0);
        const half_t* base = p.in + ((int64_t)(img * p.in_hp + y0 - 1 + p.in_ring) * p.in_wp + x0 - 1 + p.in_ring) * 64;
#pragma unroll
        for (int k = 0; k < HPW; ++k) {
            const int piece = w + 4 * k;
            int row = piece * 8 + lrow;
            row = row < HROWS ? row : HROWS - 1;
            const int hy = row / HWD, hx = row - hy * HWD;
            const int hrel = (hy * p.in_wp + hx) * 64 + ((lane & 7) ^ (hx & 6)) * 8;  // chunks keyed by the halo COLUMN: see load_frags
            const int hdst = piece < HPIECES ? WBYTES + g * HBYTES + piece * 1024 : WBYTES + 2 * HBYTES;
            __builtin_amdgcn_global_load_lds((const VTD_AS1 void*)(base + hrel), (VTD_AS3 void*)(smem + hdst), 16, 0, 0);
        }
    };
    // pixel m = w * 64 + j * 16 + fr of the block: row m / TW, column m % TW (TW is 16 or 32: shifts)
    auto px_row = [&](int j) { return (w * 64 + j * 16 + fr) / TW; };
    auto px_col = [&](int j) { return (w * 64 + j * 16 + fr) % TW; };
    // A lane's four pixels share their column modulo 16 (TW = 16: rows w*4 + j; TW = 32: rows w*2 + (j >> 1), columns fr + 16 (j & 1)), and
    // the halo's 16-byte chunks are XOR-ed with (halo column & 6): one address per half K-step, four reads at compile-time offsets
    // (the kernel above keys by the halo ROW: six address instructions per read; here that cost the registers two waves per SIMD lack)
    const int a_lane_off = (px_row(0) * HWD + px_col(0)) * 128;
    const int b_lane_off = fr * 128;
    const int bswz = (fr >> 1) & 7;
    const char* const hb = hbuf + g * HBYTES;

    // this workgroup's blocks q_j = blockIdx + j * grid; group g takes j = g, g + 2, ...
    const int nq = (total - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int nmine = (nq - g + 1) >> 1;
    const int phases = 2 * ((nq + 1) >> 1) > 2 * (nq >> 1) + 1 ? 2 * ((nq + 1) >> 1) : 2 * (nq >> 1) + 1;
#ifdef VTD_CONV_EXPERIMENT   // timing only: what the kernel costs when nothing it touches has to come from (or go to) HBM
    auto block_of = [&](int i) { return p.dbg == 1 ? (int)blockIdx.x : (int)blockIdx.x + (2 * i + g) * (int)gridDim.x; };
#else
    auto block_of = [&](int i) { return (int)blockIdx.x + (2 * i + g) * (int)gridDim.x; };
#endif

    if (nmine > 0) issue_halo(block_of(0));
    hl_wait_vmcnt<0>();               // weights (every wave's share) and the first halo
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the bias is in LDS
    __builtin_amdgcn_s_barrier();

    floatx4 acc[4][4];
    half8 rv[4][2];
    for (int phase = 0; phase < phases; ++phase) {
        if ((phase & 1) == g) {
            // ---- compute phase: K loop of my block i (barrier-free: weights resident, halo complete)
            const int i = (phase - g) >> 1;
            if (i < nmine) {
                if (RES) {
                    int img, y0, x0;
                    tile_coords(block_of(i), img, y0, x0);
                    const half_t* rbase = p.res + ((int64_t)(img * p.res_hp + y0 + p.res_ring) * p.res_wp + x0 + p.res_ring) * 64;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int rrel = (px_row(j) * p.res_wp + px_col(j)) * 64 + fq * 16;
                        rv[j][0] = *(const half8*)(rbase + rrel);
                        rv[j][1] = *(const half8*)(rbase + rrel + 8);
                    }
                }
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[a][j] = *(const floatx4*)(bias_lds + fq * 16 + a * 4);
                auto load_frags = [&](int hs, half8 (&af)[4], half8 (&bf)[4]) {
                    const int tap = hs >> 1, kk = hs & 1;
                    const int tapoff = (tap / 3) * HWD + (tap % 3), dx = tap % 3;
                    const char* pa = hb + a_lane_off + tapoff * 128 + (((fq + 4 * kk) ^ ((fr + dx) & 6)) << 4);
                    const char* pb = wl + tap * 8192 + b_lane_off + (((fq + 4 * kk) ^ bswz) << 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) af[j] = *(const half8*)(pa + (TW == 16 ? j * HWD : (j >> 1) * HWD + (j & 1) * 16) * 128);
#pragma unroll
                    for (int a = 0; a < 4; ++a) bf[a] = *(const half8*)(pb + a * 2048);
                };
                half8 fa[2][4], fb[2][4];
                load_frags(0, fa[0], fb[0]);
#pragma unroll
                for (int hs = 0; hs < 18; ++hs) {
                    if (hs + 1 < 18) load_frags(hs + 1, fa[(hs + 1) & 1], fb[(hs + 1) & 1]);
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[a][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[hs & 1][a], fa[hs & 1][j], acc[a][j], 0, 0, 0);
                    if (hs + 1 < 18) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    // Ask for the NEXT half-step's fragments here, 16 MFMAs after their reads left: the s_waitcnt lands where only those
                    // eight reads are outstanding.  Left to itself hipcc waited for them in front of the MFMAs of the half-step after
                    // -- as lgkmcnt(0), behind the eight reads issued just before: every second half-step sat out a full LDS round trip
                    if (hs + 1 < 18) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(fa[(hs + 1) & 1][j]));
#pragma unroll
                        for (int a = 0; a < 4; ++a) asm volatile("" ::"v"(fb[(hs + 1) & 1][a]));
                    }
                }
            }
        } else {
            // ---- service phase: finish block i, fetch the halo of block i + 1, wait for it (the partner group is multiplying)
            const int i = (phase - g - 1) >> 1;
            if (phase - g - 1 >= 0 && i < nmine) {
                // The next halo leaves FIRST (this group's halo buffer is free: its K loop ended before the barrier just passed), so its
                // round trip runs under the epilogue's converts and stores instead of behind them.  (Round 4, tools/gpu_l1_bound.sh: with
                // every operand served from L2 the four layer-1 launches still took 56 / 66 us against 59 / 78 -- the phases were bound by
                // this service phase's serial store -> fetch -> wait chain, not by HBM bytes.)
                if (i + 1 < nmine) issue_halo(block_of(i + 1));
                int img, y0, x0;
                tile_coords(block_of(i), img, y0, x0);
                half_t* obase = p.out + ((int64_t)(img * p.out_hp + y0 + p.out_ring) * p.out_wp + x0 + p.out_ring) * 64;
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        half8 hv;
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            float v = acc[2 * h + (k >> 2)][j][k & 3];
                            if (RES) v += (float)rv[j][h][k];
                            hv[k] = (half_t)(RELU ? fmaxf(v, 0.f) : v);
                        }
                        *(half8*)(obase + (px_row(j) * p.out_wp + px_col(j)) * 64 + fq * 16 + h * 8) = hv;
                    }
            }
            hl_wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();
    }
    hl_wait_vmcnt<0>();
}

template <int TW, bool RELU, bool RES>
int c64_launch(const HaloParams& p, hipStream_t stream) {
    constexpr int TH = 256 / TW, HROWS = (TH + 2) * (TW + 2), HBYTES = (HROWS + 7) / 8 * 1024;
    constexpr int lds = 9 * 64 * 128 + 2 * HBYTES + 1024 + 256;  // (+256: the duo kernel's bias)
    static_assert(lds <= 160 * 1024, "LDS budget");
    if (p.h % TH || p.w % TW) return -2205;  // whole pixel blocks only (block-independent addressing)
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)conv3x3_c64_persistent_kernel<TW, RELU, RES>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return -(int)e;
        attr_done = true;
    }
    const int total = p.n * p.tiles_y * p.tiles_x;
    const int grid = total < 256 ? total : 256;  // one workgroup per CU
    static const bool want_stamps = [] { const char* e = getenv("VTD_HALO_STAMPS"); return e && e[0] == '1'; }();
    if (want_stamps) {  // debug: where a workgroup's time goes, summed over its pixel blocks (synchronises!)
        HaloParams q = p;
        unsigned long long* dev = nullptr;
        if (hipMalloc(&dev, (size_t)grid * 32) != hipSuccess) return -2204;
        q.stamps = dev;
        hipLaunchKernelGGL((conv3x3_c64_persistent_kernel<TW, RELU, RES>), dim3(grid), dim3(256), lds, stream, q);
        (void)hipStreamSynchronize(stream);
        unsigned long long* h = (unsigned long long*)malloc((size_t)grid * 32);
        (void)hipMemcpy(h, dev, (size_t)grid * 32, hipMemcpyDeviceToHost);
        double a = 0, b = 0, c = 0;
        for (int i = 0; i < grid; ++i) { a += (double)h[4 * i + 1]; b += (double)(h[4 * i + 2] - h[4 * i + 1]); c += (double)(h[4 * i + 3] - h[4 * i + 2]); }
        const double blocks = (double)total / grid;
        fprintf(stderr, "[c64 stamps] res=%d blocks/WG %.1f: per block ticks: issue+wait %.1f  K-loop %.1f  epilogue %.1f\n", (int)RES, blocks,
                a / grid / blocks, b / grid / blocks, c / grid / blocks);
        free(h);
        (void)hipFree(dev);
        return 0;
    }
    const char* duo_env = getenv("VTD_C64_DUO");  // tests: 0 = the one-group kernel (read per launch so a test can flip it)
    const bool duo = !(duo_env && duo_env[0] == '0');
#ifdef VTD_CONV_EXPERIMENT
    if (duo) {
        HaloParams q = p;
        const char* d = getenv("VTD_C64_DEBUG");
        q.dbg = d ? atoi(d) : 0;
        hipError_t e = hipFuncSetAttribute((const void*)conv3x3_c64_duo_kernel<TW, RELU, RES>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return -(int)e;
        hipLaunchKernelGGL((conv3x3_c64_duo_kernel<TW, RELU, RES>), dim3(grid), dim3(512), lds, stream, q);
        return -(int)hipGetLastError();
    }
#endif
    if (duo) {
        static bool attr_duo = false;
        if (!attr_duo) {
            hipError_t e = hipFuncSetAttribute((const void*)conv3x3_c64_duo_kernel<TW, RELU, RES>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e != hipSuccess) return -(int)e;
            attr_duo = true;
        }
        hipLaunchKernelGGL((conv3x3_c64_duo_kernel<TW, RELU, RES>), dim3(grid), dim3(512), lds, stream, p);
        return -(int)hipGetLastError();
    }
    hipLaunchKernelGGL((conv3x3_c64_persistent_kernel<TW, RELU, RES>), dim3(grid), dim3(256), lds, stream, p);
    return -(int)hipGetLastError();
}

template <int TW>
int c64_dispatch(const HaloParams& p, hipStream_t stream) {
    if (p.relu) return p.res ? c64_launch<TW, true, true>(p, stream) : c64_launch<TW, true, false>(p, stream);
    return p.res ? c64_launch<TW, false, true>(p, stream) : c64_launch<TW, false, false>(p, stream);
}

template <int BN, int TW>
int halo_launch(const HaloParams& p, hipStream_t stream) {
    constexpr int TH = 256 / TW, HROWS = (TH + 2) * (TW + 2), HBYTES = (HROWS + 7) / 8 * 1024;
    const int nchunks = p.cin / 64;
    const int main_bytes = (nchunks > 1 ? 2 : 1) * HBYTES + 3 * BN * 128;
    const int epi_bytes = 256 * (BN * 4 + 16);
    const int lds = main_bytes > epi_bytes ? main_bytes : epi_bytes;
    if (lds > 160 * 1024) return -2203;
    static int attr_lds = 0;
    if (lds > attr_lds) {
        hipError_t e = hipFuncSetAttribute((const void*)conv_halo_kernel<BN, TW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return -(int)e;
        attr_lds = lds;
    }
    const int grid = p.n * p.tiles_y * p.tiles_x * p.tiles_n;
    static const bool want_stamps = [] { const char* e = getenv("VTD_HALO_STAMPS"); return e && e[0] == '1'; }();
    if (want_stamps) {  // debug: phase timing of every workgroup, printed to stderr (synchronises!)
        HaloParams q = p;
        unsigned long long* dev = nullptr;
        if (hipMalloc(&dev, (size_t)grid * 32) != hipSuccess) return -2204;
        q.stamps = dev;
        hipLaunchKernelGGL((conv_halo_kernel<BN, TW>), dim3(grid), dim3((BN / 64) * 256), lds, stream, q);
        (void)hipStreamSynchronize(stream);
        unsigned long long* h = (unsigned long long*)malloc((size_t)grid * 32);
        (void)hipMemcpy(h, dev, (size_t)grid * 32, hipMemcpyDeviceToHost);
        double pro = 0, loop = 0, epi = 0;
        unsigned long long t0 = ~0ull, t1 = 0;
        for (int i = 0; i < grid; ++i) {
            pro += (double)(h[4 * i + 1] - h[4 * i]); loop += (double)(h[4 * i + 2] - h[4 * i + 1]); epi += (double)(h[4 * i + 3] - h[4 * i + 2]);
            t0 = h[4 * i] < t0 ? h[4 * i] : t0; t1 = h[4 * i + 3] > t1 ? h[4 * i + 3] : t1;
        }
        fprintf(stderr, "[halo stamps] BN=%d TW=%d cin=%d cout=%d grid=%d: per-WG ticks (100 MHz) prologue %.1f  K-loop %.1f  epilogue %.1f; kernel span %llu ticks\n",
                BN, TW, p.cin, p.cout, grid, pro / grid, loop / grid, epi / grid, t1 - t0);
        free(h);
        (void)hipFree(dev);
        return 0;
    }
    hipLaunchKernelGGL((conv_halo_kernel<BN, TW>), dim3(grid), dim3((BN / 64) * 256), lds, stream, p);
    return -(int)hipGetLastError();
}

}  // namespace

// Shape test: can (and should) this convolution run on the halo kernel?  *tw_out = 16 or 32 picks the pixel-block shape.
bool vtd_conv_halo_supported(const ConvParams& c, int* bn_out, int* tw_out) {
    if (c.plist || c.in2 || c.pool_pw || c.stride != 1 || c.kw != 3 || c.K != 9 * c.in_c || (c.in_c & 63)) return false;  // 3x3, stride 1, whole chunks
    if (c.cin_steps * 64 != c.in_c || c.s_step != c.in_c || c.r_step != c.in_wp * c.in_c || c.k_hi_step != 32) return false;
    if ((c.cout & 63) || c.cout != c.cout_pad || c.out_c != c.cout) return false;
    if (c.flags & ~(EPI_RELU | EPI_RESIDUAL)) return false;
    if ((c.flags & EPI_RESIDUAL) && (c.res_shift || !c.res)) return false;
    if (c.in_y0 < 0 || c.in_x0 != c.in_y0 || c.ho <= 0 || c.wo <= 0 || c.M % (c.ho * c.wo)) return false;
    // pixel-block shape: 16x16, or 8x32 for short wide maps; at least 85 % of the block grid must be real pixels
    const int h = c.ho, w = c.wo;
    auto eff = [&](int th, int tw) { return (double)h * w / ((double)((h + th - 1) / th * th) * ((w + tw - 1) / tw * tw)); };
    const double e16 = eff(16, 16), e32 = eff(8, 32);
    const int tw = e16 >= e32 ? 16 : 32;
    if ((tw == 16 ? e16 : e32) < 0.85) return false;
    *tw_out = tw;
    *bn_out = (c.cout % 128 == 0) ? 128 : 64;
    return true;
}

// The persistent resident-weight variant (bn = 1 in vtd_launch_conv_halo): 64 -> 64 channels, whole pixel blocks.
bool vtd_conv_halo_c64_supported(const ConvParams& c, int tw) {
    return c.in_c == 64 && c.cout == 64 && c.wo % tw == 0 && c.ho % (256 / tw) == 0;
}

int vtd_launch_conv_halo(const ConvParams& c, int bn, int tw, hipStream_t stream) {
    HaloParams p;
    p.in = c.in; p.wgt = c.wgt; p.bias = c.bias; p.res = (c.flags & EPI_RESIDUAL) ? c.res : nullptr; p.out = (half_t*)c.out;
    p.h = c.ho; p.w = c.wo; p.n = c.M / (c.ho * c.wo); p.cin = c.in_c; p.cout = c.cout;
    p.in_hp = c.in_hp; p.in_wp = c.in_wp; p.in_ring = c.in_y0 + 1;  // in_y0 = ring - pad
    p.out_hp = c.out_hp; p.out_wp = c.out_wp; p.out_ring = c.out_ring;
    p.res_hp = c.res_hp; p.res_wp = c.res_wp; p.res_ring = c.res_ring;
    p.relu = (c.flags & EPI_RELU) ? 1 : 0;
    p.stamps = nullptr;
    p.dbg = 0;
    if (c.in_y0 != c.in_x0 || c.in_y0 < 0 || c.K != 9 * c.in_c || c.out_c != c.cout || p.n <= 0 || (c.in_c & 63) || (c.cout & 63) ||
        (tw != 16 && tw != 32) || c.stride != 1 || c.M != p.n * c.ho * c.wo)
        return -2201;  // shapes are validated here: a mismatch must never reach a kernel
    if ((int64_t)p.n * c.in_hp * c.in_wp * c.in_c >= (1ll << 31)) return -2202;  // 32-bit element offsets in the loader
    const int th = 256 / tw;
    p.tiles_x = (p.w + tw - 1) / tw; p.tiles_y = (p.h + th - 1) / th; p.tiles_n = bn > 1 ? p.cout / bn : 1;
    if (bn == 2) {  // second-generation kernel: 64 output channels per workgroup, hand-pipelined (conv_halo64_kernel)
        p.tiles_n = p.cout / 64;
        return tw == 16 ? halo64_launch<16>(p, stream) : halo64_launch<32>(p, stream);
    }
    if (bn == 1) {  // persistent resident-weight variant (cin = cout = 64)
        if (p.cin != 64 || p.cout != 64) return -2201;
        p.tiles_n = 1;
        return tw == 16 ? c64_dispatch<16>(p, stream) : c64_dispatch<32>(p, stream);
    }
    if (bn == 64 && tw == 16) return halo_launch<64, 16>(p, stream);
    if (bn == 64 && tw == 32) return halo_launch<64, 32>(p, stream);
    if (bn == 128 && tw == 16) return halo_launch<128, 16>(p, stream);
    if (bn == 128 && tw == 32) return halo_launch<128, 32>(p, stream);
    return -2201;
}
