// Composed DB-head entry (FPN lateral(C2) + top-down add + P2 smooth 3x3 + head conv 3x3 + BN + ReLU, see vtd_api.cpp
// compose_head_entry; reference layers app/ml/models/text_detector.py:36-66) for the four INTERIOR parity classes, with the
// A operand staged as HALO PLANES instead of gathered per tap.
//
// conv_igemm.hip's classed mode gathers one im2col row per output pixel and tap: 61 K-steps x 16 one-KB LDS-DMA pieces of
// A per 128-pixel tile.  DESIGN.md section 6: that gather (not HBM, not the MFMAs) is what bounds the kernel.  Here a tile is an
// 8 x 16 block of same-parity output pixels, i.e. of L3 pixels (y >> 1, x >> 1), and
//   * the 5x5 window on C2 splits by the parity of the tap's source pixel into four "planes" of C2 (every second row /
//     column): within a plane the 25 taps are 9 + 6 + 6 + 4 taps of a 3x3 neighbourhood on a 10 x 18 pixel halo;
//   * the parity-grouped 3x3 window on L3 is a 3x3 neighbourhood on a 10 x 18 halo of L3 per 64-channel chunk.
// So the K loop is 4*nch1 + 4 groups; a group stages ONE halo (23 pieces) and then runs its taps as K-steps that read that
// LDS image at shifted rows (16-byte chunks XOR-swizzled by (row & 6): conflict-free for any shift, see conv_halo.hip); only the weights (8 pieces per
// K-step) still stream.  184 halo pieces per tile replace 976 gathered ones.  The original weight matrix is used as is: a
// per-class step table maps each K-step to its column offset in it.
// LDS: 24 KB halo + a RING x 8 KB weight ring (RING-1 K-steps of weights in flight per workgroup) -> 2-4 workgroups per CU,
// which hide the (exposed) halo fetch at each group switch.
#include <cstdio>
#include <cstdlib>
#include "vtd_common.h"

namespace {

constexpr int HE_HW = 18, HE_ROWS = 180, HE_PIECES = 23, HE_HALO_BYTES = 24 * 1024;
constexpr int HE_BSTAGE = 64 * 128;
constexpr int HE_EPI_ROW = 64 * 4 + 16;

struct HeadHaloParams {
    const half_t* c2;      // [n][c2_hp][c2_wp][c2_c], ring c2_ring >= 2
    const half_t* l3;      // [n][l3_hp][l3_wp][256], ring l3_ring >= 1
    const half_t* wgt;     // [16 classes][64][K] fp16 (compose_head_entry's layout)
    const float* bias_tab; // [25][64]
    half_t* out;           // [n][out_hp][out_wp][64]
    const int* steps;      // [4 interior classes][nsteps][2]: {k offset (elements), tapoff | first << 8 | src << 9 | chunk << 12}
    int n, h, w, K, nsteps;
    int c2_hp, c2_wp, c2_c, c2_ring, l3_hp, l3_wp, l3_ring, out_hp, out_wp, out_ring;
    int blocks_y, blocks_x;
    unsigned long long* stamps;  // debug (VTD_HALO_STAMPS=1): per workgroup {wait+barrier, compute, group switches, total} cycles
};

template <int N>
__device__ __forceinline__ void he_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int RING, int NW>
__global__ __launch_bounds__(NW * 64) void head_entry_halo_kernel(const HeadHaloParams p) {
    constexpr int NT = NW * 64, FN = 8 / NW;  // NW = 4: waves 2 x 2, 64 px x 32 ch each; NW = 2: 64 px x 64 ch each (fewer LDS reads per MFMA)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const hb = smem;
    char* const bring = smem + HE_HALO_BYTES;
    int* const stab = (int*)(bring + RING * HE_BSTAGE);  // this class's step table, staged once (a scalar load per K-step sat on
                                                      // the critical path: barrier -> s_load -> weights issue -> MFMAs)

    // ---- tile: (image, block row, block column, interior class); the four classes of a block are neighbours (shared L2 lines)
    const int nblk = gridDim.x, b = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = b & 7;
    int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b >> 3);
    const int cls4 = tile & 3;
    tile >>= 2;
    const int bx = tile % p.blocks_x;
    tile /= p.blocks_x;
    const int by = tile % p.blocks_y;
    const int img = tile / p.blocks_y;
    const int py = cls4 >> 1, px = cls4 & 1;            // row / column parity of the class's pixels
    // L3-index range of the class: even pixels 2..h-2 -> 1..h/2-1, odd pixels 1..h-3 -> 0..h/2-2
    const int ly_min = py ? 0 : 1, lx_min = px ? 0 : 1;
    const int ly_cnt = p.h / 2 - 1, lx_cnt = p.w / 2 - 1;
    const int ly0 = ly_min + by * 8, lx0 = lx_min + bx * 16;
    const int yk = py ? 2 : 1, xk = px ? 2 : 1;          // compose_head_entry's kind indices -> weight class yk*4 + xk
    const half_t* wcls = p.wgt + (int64_t)(yk * 4 + xk) * 64 * p.K;
    const int* steps = p.steps + (int64_t)cls4 * p.nsteps * 2;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane >> 3, fr = lane & 15, fq = lane >> 4;

    // ---- halo loader: group (src, chunk) -> 23 pieces of 8 pixels x 128 bytes; wave w takes pieces w, w+4, ...
    auto issue_halo = [&](int src, int chunk) {
#pragma unroll
        for (int k = 0; k < 24 / NW; ++k) {
            const int piece = w + NW * k;
            if (piece >= HE_PIECES) break;
            int row = piece * 8 + lrow;
            row = row < HE_ROWS ? row : HE_ROWS - 1;
            const int i = row / HE_HW, j = row - i * HE_HW;
            const int c_log = (lane & 7) ^ (row & 6);
            const half_t* g;
            if (src < 4) {  // C2 plane (a, b): pixel (2(ly0-1+i)+a, 2(lx0-1+j)+b)
                int yy = 2 * (ly0 - 1 + i) + (src >> 1) + p.c2_ring, xx = 2 * (lx0 - 1 + j) + (src & 1) + p.c2_ring;
                yy = yy < p.c2_hp ? yy : p.c2_hp - 1;  // partial blocks: stay inside the allocation (those rows are masked)
                xx = xx < p.c2_wp ? xx : p.c2_wp - 1;
                g = p.c2 + ((int64_t)(img * p.c2_hp + yy) * p.c2_wp + xx) * p.c2_c + chunk * 64 + c_log * 8;
            } else {        // L3: pixel (ly0-1+i, lx0-1+j)
                int yy = ly0 - 1 + i + p.l3_ring, xx = lx0 - 1 + j + p.l3_ring;
                yy = yy < p.l3_hp ? yy : p.l3_hp - 1;
                xx = xx < p.l3_wp ? xx : p.l3_wp - 1;
                g = p.l3 + ((int64_t)(img * p.l3_hp + yy) * p.l3_wp + xx) * 256 + chunk * 64 + c_log * 8;
            }
            __builtin_amdgcn_global_load_lds((const VTD_AS1 void*)g, (VTD_AS3 void*)(hb + piece * 1024), 16, 0, 0);
        }
    };
    // ---- weight loader: rows (i*4 + w)*8 + lrow of the class's [64][K] matrix, 128 bytes per K-step
    const half_t* bsrc[8 / NW];
#pragma unroll
    for (int i = 0; i < 8 / NW; ++i) {
        const int row = (i * NW + w) * 8 + lrow;
        bsrc[i] = wcls + (int64_t)row * p.K + ((lane & 7) ^ ((row >> 1) & 7)) * 8;
    }
    auto issue_b = [&](int koff, int stage) {
#pragma unroll
        for (int i = 0; i < 8 / NW; ++i)
            __builtin_amdgcn_global_load_lds((const VTD_AS1 void*)(bsrc[i] + koff),
                                             (VTD_AS3 void*)(bring + stage * HE_BSTAGE + (i * NW + w) * 1024), 16, 0, 0);
    };

    // ---- compute state: waves 2 x 2, wave tile 64 pixels x 32 channels
    const int wm = NW == 4 ? w >> 1 : w, wn = NW == 4 ? w & 1 : 0;
    int hbase[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = wm * 64 + j * 16 + fr;
        hbase[j] = (m >> 4) * HE_HW + (m & 15);  // halo row of tile pixel (r, c) at window offset (0, 0); taps add (u+1)*18 + (v+1)
    }
    const int b_lane_off = (wn * FN * 16 + fr) * 128;
    const int bswz = (fr >> 1) & 7;
    floatx4 acc[FN][4];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    for (int i = tid; i < 2 * p.nsteps; i += NT) stab[i] = steps[i];
    __syncthreads();
    // Weights run RING-1 K-steps ahead of their use, independent of the halo groups (an LDS-DMA needs ~2 us to land; with 64
    // output channels a K-step is only 0.13 us of MFMA work, so the bytes in flight per CU are what sets the speed).
    auto koff_of = [&](int s) { return __builtin_amdgcn_readfirstlane(stab[2 * s]); };
    int d_t = __builtin_amdgcn_readfirstlane(stab[1]);
    issue_halo((d_t >> 9) & 7, (d_t >> 12) & 15);
#pragma unroll
    for (int a = 0; a < RING - 1; ++a)
        if (a < p.nsteps) issue_b(koff_of(a), a);
    int n_t = __builtin_amdgcn_readfirstlane(p.nsteps > 1 ? stab[3] : 0);  // descriptor of step s+1, one step ahead of its use
    bool fresh_halo = true;
    int stage = 0;
    unsigned long long t0 = 0, t1 = 0, t2 = 0, a_wait = 0, a_comp = 0, a_sw = 0, t_begin = 0;
    if (p.stamps) t_begin = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < p.nsteps; ++s) {
        if (p.stamps) t0 = __builtin_amdgcn_s_memtime();
        // step s's weights are older than the RING-2 weight loads issued after them; a freshly issued halo is the newest load
        if (fresh_halo || s + RING - 2 >= p.nsteps) he_wait_vmcnt<0>(); else he_wait_vmcnt<(8 / NW) * (RING - 2)>();
        __builtin_amdgcn_s_barrier();  // landed for every wave; everyone left step s-1 (its ring stage may be refilled)
        if (p.stamps) t1 = __builtin_amdgcn_s_memtime();
        const int tapoff = d_t & 0xff;
        if (s + RING - 1 < p.nsteps) {
            const int st = stage + RING - 1 >= RING ? stage - 1 : stage + RING - 1;
            issue_b(koff_of(s + RING - 1), st);
        }
        const int s2 = s + 2 < p.nsteps ? s + 2 : s;
        const int nn_t = __builtin_amdgcn_readfirstlane(stab[2 * s2 + 1]);
        const char* sb = bring + stage * HE_BSTAGE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            half8 af[4], bf[FN];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int hrow = hbase[j] + tapoff;
                af[j] = *(const half8*)(hb + hrow * 128 + (((fq + 4 * kk) ^ (hrow & 6)) << 4));
            }
#pragma unroll
            for (int i = 0; i < FN; ++i) bf[i] = *(const half8*)(sb + b_lane_off + i * 2048 + (((fq + 4 * kk) ^ bswz) << 4));
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[i], af[j], acc[i][j], 0, 0, 0);
        }
        fresh_halo = false;
        if (p.stamps) {
            asm volatile("s_nop 0" ::"v"(acc[0][0][0]));
            t2 = __builtin_amdgcn_s_memtime();
            a_wait += t1 - t0;
            a_comp += t2 - t1;
        }
        if (s + 1 < p.nsteps && ((n_t >> 8) & 1)) {
            __builtin_amdgcn_s_barrier();  // group switch: every wave is done with the halo before it is replaced
            issue_halo((n_t >> 9) & 7, (n_t >> 12) & 15);
            fresh_halo = true;
            if (p.stamps) a_sw += __builtin_amdgcn_s_memtime() - t2;
        }
        d_t = n_t;
        n_t = nn_t;
        stage = stage + 1 == RING ? 0 : stage + 1;
    }

    if (p.stamps && tid == 0) {
        unsigned long long* o = p.stamps + (int64_t)blockIdx.x * 4;
        o[0] = a_wait; o[1] = a_comp; o[2] = a_sw; o[3] = __builtin_amdgcn_s_memtime() - t_begin;
    }
    // ---- epilogue: accumulators -> fp32 LDS tile -> position-dependent bias, ReLU, 16-byte NHWC stores
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < FN; ++i)
            *(floatx4*)(smem + (wm * 64 + j * 16 + fr) * HE_EPI_ROW + (wn * FN * 16 + i * 16 + fq * 4) * 4) = acc[i][j];
    __syncthreads();
    const int cc = tid & 7, r0 = tid >> 3;
#pragma unroll
    for (int it = 0; it < 1024 / NT; ++it) {
        const int m = it * (NT / 8) + r0;
        const int r = m >> 4, c = m & 15;
        if (ly0 + r >= ly_min + ly_cnt || lx0 + c >= lx_min + lx_cnt) continue;
        const int oy = 2 * (ly0 + r) + py, ox = 2 * (lx0 + c) + px;
        const int yc = oy == 1 ? 1 : oy == p.h - 2 ? 3 : 2;  // interior classes never touch rows / columns 0 and h-1
        const int xc = ox == 1 ? 1 : ox == p.w - 2 ? 3 : 2;
        const float* bt = p.bias_tab + (yc * 5 + xc) * 64 + cc * 8;
        const floatx4 b0 = *(const floatx4*)bt, b1 = *(const floatx4*)(bt + 4);
        const floatx4 v0 = *(const floatx4*)(smem + m * HE_EPI_ROW + cc * 32) + b0;
        const floatx4 v1 = *(const floatx4*)(smem + m * HE_EPI_ROW + cc * 32 + 16) + b1;
        half8 hv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            hv[e] = (half_t)fmaxf(v0[e], 0.f);
            hv[4 + e] = (half_t)fmaxf(v1[e], 0.f);
        }
        *(half8*)(p.out + (((int64_t)img * p.out_hp + oy + p.out_ring) * p.out_wp + ox + p.out_ring) * 64 + cc * 8) = hv;
    }
}


// ------------------------------------------------------------------------------------------------------------------------
// Same computation, bigger register tiles.  tools/mfma_peak.hip: the part is power bound, a wave issues an MFMA at best
// every 17 cycles and what a kernel can change is the LDS traffic per MFMA.  Above, a wave owns 64 pixels x 32 channels
// (6 fragment reads per 8 MFMAs) and four waves per SIMD hide each other's LDS latency.  Here a tile is a 16 x 16 block of
// same-parity pixels (256 GEMM rows, halo 18 x 18), four waves of 64 pixels x 64 channels each (8 reads per 16 MFMAs: a third
// less LDS traffic, a third less halo per pixel), two workgroups per CU, and the latency hiding is done by hand as in
// conv3x3_c64_persistent_kernel: the 8 fragment reads of half K-step h+1 are issued under the 16 MFMAs of half-step h.
// That needs the NEXT K-step's weights in LDS one step early, hence a 4-stage weight ring with a counted wait that leaves
// only the newest stage in flight.  Across a group switch (new halo) there is nothing to prefetch: the first half-step of a
// group reads its fragments in the open.
constexpr int HB_HW = 18, HB_ROWS = 324, HB_PIECES = 41, HB_HALO_BYTES = 41 * 1024, HB_RING = 4;

template <bool STAMPS>
__global__ __launch_bounds__(256, 2) void head_entry_halo256_kernel(const HeadHaloParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const hb = smem;
    char* const bring = smem + HB_HALO_BYTES;
    int* const stab = (int*)(bring + HB_RING * HE_BSTAGE);

    const int nblk = gridDim.x, b = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = b & 7;
    int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b >> 3);
    const int cls4 = tile & 3;
    tile >>= 2;
    const int bx = tile % p.blocks_x;
    tile /= p.blocks_x;
    const int by = tile % p.blocks_y;
    const int img = tile / p.blocks_y;
    const int py = cls4 >> 1, px = cls4 & 1;
    const int ly_min = py ? 0 : 1, lx_min = px ? 0 : 1;
    const int ly_cnt = p.h / 2 - 1, lx_cnt = p.w / 2 - 1;
    const int ly0 = ly_min + by * 16, lx0 = lx_min + bx * 16;
    const half_t* wcls = p.wgt + (int64_t)((py ? 2 : 1) * 4 + (px ? 2 : 1)) * 64 * p.K;
    const int* steps = p.steps + (int64_t)cls4 * p.nsteps * 2;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane >> 3, fr = lane & 15, fq = lane >> 4;

    auto issue_halo = [&](int src, int chunk) {
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const int piece = w + 4 * k;
            if (piece >= HB_PIECES) break;
            int row = piece * 8 + lrow;
            row = row < HB_ROWS ? row : HB_ROWS - 1;
            const int i = row / HB_HW, j = row - i * HB_HW;
            const int c_log = (lane & 7) ^ (j & 6);   // keyed by the halo COLUMN: see load_frags
            const half_t* g;
            if (src < 4) {
                int yy = 2 * (ly0 - 1 + i) + (src >> 1) + p.c2_ring, xx = 2 * (lx0 - 1 + j) + (src & 1) + p.c2_ring;
                yy = yy < p.c2_hp ? yy : p.c2_hp - 1;
                xx = xx < p.c2_wp ? xx : p.c2_wp - 1;
                g = p.c2 + ((int64_t)(img * p.c2_hp + yy) * p.c2_wp + xx) * p.c2_c + chunk * 64 + c_log * 8;
            } else {
                int yy = ly0 - 1 + i + p.l3_ring, xx = lx0 - 1 + j + p.l3_ring;
                yy = yy < p.l3_hp ? yy : p.l3_hp - 1;
                xx = xx < p.l3_wp ? xx : p.l3_wp - 1;
                g = p.l3 + ((int64_t)(img * p.l3_hp + yy) * p.l3_wp + xx) * 256 + chunk * 64 + c_log * 8;
            }
            __builtin_amdgcn_global_load_lds((const VTD_AS1 void*)g, (VTD_AS3 void*)(hb + piece * 1024), 16, 0, 0);
        }
    };
    const half_t* bsrc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (i * 4 + w) * 8 + lrow;
        bsrc[i] = wcls + (int64_t)row * p.K + ((lane & 7) ^ ((row >> 1) & 7)) * 8;
    }
    auto issue_b = [&](int koff, int stage) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const VTD_AS1 void*)(bsrc[i] + koff),
                                             (VTD_AS3 void*)(bring + stage * HE_BSTAGE + (i * 4 + w) * 1024), 16, 0, 0);
    };

    int hbase[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = w * 64 + j * 16 + fr;
        hbase[j] = (m >> 4) * HB_HW + (m & 15);
    }
    const int b_lane_off = fr * 128;
    const int bswz = (fr >> 1) & 7;
    floatx4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    for (int i = tid; i < 2 * p.nsteps; i += 256) stab[i] = steps[i];
    __syncthreads();
    auto koff_of = [&](int s) { return __builtin_amdgcn_readfirstlane(stab[2 * s]); };
    // fragments of (tap offset, ring stage, half kk).  PMC (SQ_INSTS_VALU / SQ_INSTS_MFMA) put this kernel at 3.7 vector instructions
    // per MFMA -- 8 issue cycles for the MFMA + 15 for VALU against the 16 cycles the MFMA occupies the pipe: the waves were
    // issue-bound, and most of it was address arithmetic of these reads (row = base + tap, row * 128, (row & 6) ^ chunk, shift,
    // add: six instructions per read).  Now: the 16-byte chunks of a halo row are XOR-swizzled by the row's halo COLUMN x (rows of
    // one lane group are consecutive columns, so this is as conflict-free as the row key), x = fr + (tap column) is the same for
    // the lane's four pixels, and those are 18 rows = 2304 bytes apart: ONE address per half step, four reads at immediate offsets.
    const int a_lane_off = ((w * 4) * HB_HW + fr) * 128;
    auto load_frags = [&](int tapoff, int stage, int kk, half8 (&af)[4], half8 (&bf)[4]) {
        const char* sb = bring + stage * HE_BSTAGE;
        const int dx = (tapoff >> 16) & 3;   // the tap's halo column offset, wave-uniform (bits 16-17 of the step descriptor)
        const char* pa = hb + a_lane_off + (tapoff & 0xff) * 128 + (((fq + 4 * kk) ^ ((fr + dx) & 6)) << 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) af[j] = *(const half8*)(pa + j * (HB_HW * 128));
#pragma unroll
        for (int i = 0; i < 4; ++i) bf[i] = *(const half8*)(sb + b_lane_off + i * 2048 + (((fq + 4 * kk) ^ bswz) << 4));
    };

    // Step descriptors come out of LDS too.  They are read at the END of a step, behind that step's fragment prefetch, and
    // consumed after the wait at the top of the next step (which is due anyway for the prefetched fragments): read anywhere
    // else they put an s_waitcnt lgkmcnt(0) in front of the MFMAs and the fragment prefetch is lost.
    auto raw_k = [&](int s) { return stab[2 * (s < p.nsteps ? s : p.nsteps - 1)]; };
    auto raw_t = [&](int s) { return stab[2 * (s < p.nsteps ? s : p.nsteps - 1) + 1]; };
    int d_t = __builtin_amdgcn_readfirstlane(raw_t(0)), n_t = __builtin_amdgcn_readfirstlane(raw_t(1));
    issue_halo((d_t >> 9) & 7, (d_t >> 12) & 15);
#pragma unroll
    for (int a = 0; a < HB_RING - 1; ++a)
        if (a < p.nsteps) issue_b(koff_of(a), a);
    int v_k3 = raw_k(HB_RING - 1), v_t2 = raw_t(2);  // koff(s+3), desc(s+2) for s = 0 (vector copies, made uniform after the wait)
    half8 fa[2][4], fb[2][4];
    bool have_frags = false;  // fragments of (s, half 0) already prefetched during step s-1
    bool fresh_halo = true;
    int stage = 0;
    unsigned long long t0 = 0, t1 = 0, t2 = 0, a_wait = 0, a_comp = 0, a_sw = 0, t_begin = 0;
    if constexpr (STAMPS) t_begin = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < p.nsteps; ++s) {
        if constexpr (STAMPS) t0 = __builtin_amdgcn_s_memtime();
        // all waves: weights of steps s and s+1 landed (only the newest ring stage may still be in flight); a fresh halo is newer
        // than every weight load, so it needs the full wait
        if (fresh_halo || s + 2 >= p.nsteps) he_wait_vmcnt<0>(); else he_wait_vmcnt<2>();
        __builtin_amdgcn_s_barrier();
        if constexpr (STAMPS) t1 = __builtin_amdgcn_s_memtime();
        const int k3 = __builtin_amdgcn_readfirstlane(v_k3), nn_t = __builtin_amdgcn_readfirstlane(v_t2);
        if (s + HB_RING - 1 < p.nsteps) issue_b(k3, (stage + HB_RING - 1) & 3);
        const int tapoff = d_t;
        const bool next_same_group = s + 1 < p.nsteps && !((n_t >> 8) & 1);
        // Where the waits for fragment reads go is decided HERE, by asking for the registers (an empty asm): hipcc's wait-count pass
        // does not count LDS reads past a batch of eight -- it put s_waitcnt lgkmcnt(0) in front of the MFMAs of half 0, behind the
        // eight reads of half 1 issued just above: every K-step sat out a full LDS round trip in the open and the prefetch was lost
        // (ISA of the round-2 kernel).  Each batch is now waited for 16 MFMAs after it left and before anything younger is issued.
        if (!have_frags) {  // first step of a halo group: nothing was prefetched, these reads are in the open
            load_frags(tapoff, stage, 0, fa[0], fb[0]);
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(fa[0][j]));
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(fb[0][i]));
        }
        // half 0: prefetch half 1 of this step
        load_frags(tapoff, stage, 1, fa[1], fb[1]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[0][i], fa[0][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(fa[1][j]));
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(fb[1][i]));
        // half 1: prefetch half 0 of the next step (same halo group only), then the descriptors of the steps after it
        if (next_same_group) load_frags(n_t, (stage + 1) & 3, 0, fa[0], fb[0]);
        v_k3 = raw_k(s + HB_RING);
        v_t2 = raw_t(s + 3);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[1][i], fa[1][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        have_frags = next_same_group;
        fresh_halo = false;
        if constexpr (STAMPS) {
            asm volatile("s_nop 0" ::"v"(acc[0][0][0]));
            t2 = __builtin_amdgcn_s_memtime();
            a_wait += t1 - t0;
            a_comp += t2 - t1;
        }
        if (s + 1 < p.nsteps && !next_same_group) {
            __builtin_amdgcn_s_barrier();  // group switch: every wave is done with the halo before it is replaced
            issue_halo((n_t >> 9) & 7, (n_t >> 12) & 15);
            fresh_halo = true;
            if constexpr (STAMPS) a_sw += __builtin_amdgcn_s_memtime() - t2;
        }
        d_t = n_t;
        n_t = nn_t;
        stage = (stage + 1) & 3;
    }

    if (STAMPS && tid == 0) {
        unsigned long long* o = p.stamps + (int64_t)blockIdx.x * 4;
        o[0] = a_wait; o[1] = a_comp; o[2] = a_sw; o[3] = __builtin_amdgcn_s_memtime() - t_begin;
    }
    // ---- epilogue: accumulators -> fp32 LDS tile -> position-dependent bias, ReLU, 16-byte NHWC stores
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *(floatx4*)(smem + (w * 64 + j * 16 + fr) * HE_EPI_ROW + (i * 16 + fq * 4) * 4) = acc[i][j];
    __syncthreads();
    const int cc = tid & 7, r0 = tid >> 3;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int m = it * 32 + r0;
        const int r = m >> 4, c = m & 15;
        if (ly0 + r >= ly_min + ly_cnt || lx0 + c >= lx_min + lx_cnt) continue;
        const int oy = 2 * (ly0 + r) + py, ox = 2 * (lx0 + c) + px;
        const int yc = oy == 1 ? 1 : oy == p.h - 2 ? 3 : 2;
        const int xc = ox == 1 ? 1 : ox == p.w - 2 ? 3 : 2;
        const float* bt = p.bias_tab + (yc * 5 + xc) * 64 + cc * 8;
        const floatx4 b0 = *(const floatx4*)bt, b1 = *(const floatx4*)(bt + 4);
        const floatx4 v0 = *(const floatx4*)(smem + m * HE_EPI_ROW + cc * 32) + b0;
        const floatx4 v1 = *(const floatx4*)(smem + m * HE_EPI_ROW + cc * 32 + 16) + b1;
        half8 hv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            hv[e] = (half_t)fmaxf(v0[e], 0.f);
            hv[4 + e] = (half_t)fmaxf(v1[e], 0.f);
        }
        *(half8*)(p.out + (((int64_t)img * p.out_hp + oy + p.out_ring) * p.out_wp + ox + p.out_ring) * 64 + cc * 8) = hv;
    }
}

}  // namespace

// Step table of one interior class (py, px): entries {k offset into the [64][K] class matrix, tapoff | first<<8 | src<<9 | chunk<<12 | tap column<<16}.
// nch1 = C2 channels / 64.  Returns the number of steps (25*nch1 + 36).
// (extern "C": tests/test_abi.py replays the schedules on the host and looks the helper up by name)
extern "C" int vtd_head_entry_halo_steps(int py, int px, int nch1, int* out /* [(25*nch1+36)*2] */) {
    int s = 0;
    const int c2ch = nch1 * 64;
    for (int ch = 0; ch < nch1; ++ch)
        for (int plane = 0; plane < 4; ++plane) {
            bool first = true;
            for (int dy = 0; dy < 5; ++dy)
                for (int dx = 0; dx < 5; ++dx) {
                    const int ty = py + dy - 2, tx = px + dx - 2;
                    const int uy = ty >= 0 ? ty / 2 : -((-ty + 1) / 2), ux = tx >= 0 ? tx / 2 : -((-tx + 1) / 2);  // floor
                    const int a = ty - 2 * uy, b = tx - 2 * ux;
                    if (a * 2 + b != plane) continue;
                    out[2 * s] = (dy * 5 + dx) * c2ch + ch * 64;
                    out[2 * s + 1] = ((uy + 1) * 18 + (ux + 1)) | ((first ? 1 : 0) << 8) | (plane << 9) | (ch << 12) | ((ux + 1) << 16);
                    first = false;
                    ++s;
                }
        }
    for (int ch = 0; ch < 4; ++ch)
        for (int ij = 0; ij < 9; ++ij) {
            out[2 * s] = 25 * c2ch + ij * 256 + ch * 64;
            out[2 * s + 1] = ((ij / 3) * 18 + (ij % 3)) | ((ij == 0 ? 1 : 0) << 8) | (4 << 9) | (ch << 12) | ((ij % 3) << 16);
            ++s;
        }
    return s;
}

int vtd_launch_head_entry_halo(const ConvParams& c, const int* steps_dev, int nsteps, int big_tiles, hipStream_t stream) {
    if (!c.plist || !c.in2 || !c.bias_tab || c.cout != 64 || c.in2_c != 256 || (c.in_c & 63) || c.in_y0 < 0 || (c.img_h & 1) || (c.img_w & 1) ||
        c.tiles_per_img <= 0 || nsteps != 25 * (c.in_c / 64) + 36 || c.K != 25 * c.in_c + 9 * 256)
        return -2301;
    HeadHaloParams p;
    p.c2 = c.in; p.l3 = c.in2; p.wgt = c.wgt; p.bias_tab = c.bias_tab; p.out = (half_t*)c.out; p.steps = steps_dev;
    p.n = c.M / (c.tiles_per_img * 128); p.h = c.img_h; p.w = c.img_w; p.K = c.K; p.nsteps = nsteps;
    p.c2_hp = c.in_hp; p.c2_wp = c.in_wp; p.c2_c = c.in_c; p.c2_ring = c.in_y0 + 2;  // in_y0 = ring - 2
    p.l3_hp = c.in2_hp; p.l3_wp = c.in2_wp; p.l3_ring = c.in2_ring;
    p.out_hp = c.out_hp; p.out_wp = c.out_wp; p.out_ring = c.out_ring;
    const int cnt_y = c.img_h / 2 - 1, cnt_x = c.img_w / 2 - 1;
    p.blocks_y = (cnt_y + 7) / 8; p.blocks_x = (cnt_x + 15) / 16;
    if (p.n <= 0 || p.c2_ring < 2 || p.l3_ring < 1) return -2302;
    if (big_tiles) {  // 16 x 16 pixel blocks, 64 x 64 register tiles, hand-pipelined fragment reads
        p.blocks_y = (cnt_y + 15) / 16; p.blocks_x = (cnt_x + 15) / 16;
        p.stamps = nullptr;
        const int lds256 = HB_HALO_BYTES + HB_RING * HE_BSTAGE + 2 * nsteps * 4;
        if (lds256 < 256 * HE_EPI_ROW) return -2303;
        static bool attr256 = false;
        if (!attr256) {
            hipError_t e = hipFuncSetAttribute((const void*)head_entry_halo256_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess) e = hipFuncSetAttribute((const void*)head_entry_halo256_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return -(int)e;
            attr256 = true;
        }
        const int grid256 = p.n * p.blocks_y * p.blocks_x * 4;
        static const bool stamps256 = [] { const char* e = getenv("VTD_HALO_STAMPS"); return e && e[0] == '1'; }();
        if (stamps256) {  // debug: where a workgroup's K loop goes (synchronises!)
            unsigned long long* dev = nullptr;
            if (hipMalloc(&dev, (size_t)grid256 * 32) != hipSuccess) return -2304;
            p.stamps = dev;
            hipLaunchKernelGGL(head_entry_halo256_kernel<true>, dim3(grid256), dim3(256), lds256, stream, p);
            (void)hipStreamSynchronize(stream);
            unsigned long long* h = (unsigned long long*)malloc((size_t)grid256 * 32);
            (void)hipMemcpy(h, dev, (size_t)grid256 * 32, hipMemcpyDeviceToHost);
            double a = 0, b = 0, cc = 0, d = 0;
            for (int i = 0; i < grid256; ++i) { a += (double)h[4 * i]; b += (double)h[4 * i + 1]; cc += (double)h[4 * i + 2]; d += (double)h[4 * i + 3]; }
            fprintf(stderr, "[head_entry_halo256 stamps] grid %d: per workgroup cycles: wait+barrier %.0f  compute %.0f  group switch %.0f  K loop total %.0f\n",
                    grid256, a / grid256, b / grid256, cc / grid256, d / grid256);
            free(h);
            (void)hipFree(dev);
            return 0;
        }
        hipLaunchKernelGGL(head_entry_halo256_kernel<false>, dim3(grid256), dim3(256), lds256, stream, p);
        return -(int)hipGetLastError();
    }
    static const int ring = [] { const char* e = getenv("VTD_HEAD_HALO_RING"); const int r = e ? atoi(e) : 2; return (r == 2 || r == 3) ? r : 2; }();
    static const int nw = [] { const char* e = getenv("VTD_HEAD_HALO_WAVES"); return (e && atoi(e) == 2) ? 2 : 4; }();
    const int lds = HE_HALO_BYTES + ring * HE_BSTAGE + 2 * nsteps * 4;  // halo + weight ring + step table (>= the 34 KB epilogue tile)
    const int grid = p.n * p.blocks_y * p.blocks_x * 4;
    p.stamps = nullptr;
    static const bool want_stamps = [] { const char* e = getenv("VTD_HALO_STAMPS"); return e && e[0] == '1'; }();
    static bool attr_done = false;  // (ring, nw) are fixed for the life of the process: one instantiation is ever launched
    auto go = [&](auto kernel, int threads) {
        if (!attr_done) {
            hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return -(int)e;
            attr_done = true;
        }
        if (want_stamps) {  // debug: where a workgroup's K loop goes (synchronises!)
            unsigned long long* dev = nullptr;
            if (hipMalloc(&dev, (size_t)grid * 32) != hipSuccess) return -2304;
            p.stamps = dev;
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads), lds, stream, p);
            (void)hipStreamSynchronize(stream);
            unsigned long long* h = (unsigned long long*)malloc((size_t)grid * 32);
            (void)hipMemcpy(h, dev, (size_t)grid * 32, hipMemcpyDeviceToHost);
            double a = 0, b = 0, c = 0, d = 0;
            for (int i = 0; i < grid; ++i) { a += (double)h[4 * i]; b += (double)h[4 * i + 1]; c += (double)h[4 * i + 2]; d += (double)h[4 * i + 3]; }
            fprintf(stderr, "[head_entry_halo stamps] ring %d waves %d grid %d: per workgroup cycles: wait+barrier %.0f  compute %.0f  group switch %.0f  K loop total %.0f\n",
                    ring, nw, grid, a / grid, b / grid, c / grid, d / grid);
            free(h);
            (void)hipFree(dev);
            return 0;
        }
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads), lds, stream, p);
        return -(int)hipGetLastError();
    };
    if (nw == 4) return ring == 2 ? go(head_entry_halo_kernel<2, 4>, 256) : go(head_entry_halo_kernel<3, 4>, 256);
    return ring == 2 ? go(head_entry_halo_kernel<2, 2>, 128) : go(head_entry_halo_kernel<3, 2>, 128);
}
