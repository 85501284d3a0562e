// Composed DB-head entry, third generation ("pair" kernel): what head_entry_halo.hip's stamps say is left on the table --
//   * two LDS-DMA issues per wave and K-step (60-185 cycles each) for the weights: here ONE workgroup of 8 waves computes TWO
//     16x16 pixel blocks of the same parity class against ONE weight ring, so a wave issues one weight piece per K-step and the
//     weight bytes per MFMA halve (the global->LDS path, not HBM and not the MFMAs, bounds the previous kernels);
//   * a group switch (new halo) that stalls the workgroup for the fetch latency eight times per tile: here a halo covers 32
//     channels instead of 64 (20.25 KB), each tile keeps THREE of them in LDS and the halo of group g+2 is fetched while group g
//     computes -- no wave ever waits for a halo it has just requested;
//   * the K loop walks HALF steps (one tap x 32 channels) that are paired into K-steps of 64 across group boundaries; C2 and L3
//     groups alternate so that every short group (4 or 6 taps) follows a 9-tap one and prefetch distance never drops below 6 steps.
// One barrier per K-step (32 MFMAs per wave); fragment reads of the next half step are always in flight under the MFMAs of
// the current one (there is no "first half-step of a group in the open" any more).
// LDS: 2 tiles x 3 halo slots x 21 KB + 4-stage weight ring x 8 KB + step table = 160 KB -> one workgroup per CU, two waves per SIMD.
// Same maths, same weight matrix and bias table as conv_igemm's classed mode (vtd_api.cpp compose_head_entry; reference layers
// app/ml/models/text_detector.py:36-66); border classes stay on the gathered 128-row tiles.
#include <cstdio>
#include <cstdlib>
#include "vtd_common.h"

namespace {

constexpr int HP_HW = 18, HP_ROWS = 324, HP_PIECES = 21, HP_SLOT = HP_PIECES * 1024, HP_RING = 4, HP_STAGE = 64 * 128;
constexpr int HP_EPI_ROW = 64 * 4 + 16;
constexpr int HP_TILE_HALO = 3 * HP_SLOT;

struct PairParams {
    const half_t* c2;
    const half_t* l3;
    const half_t* wgt;      // [16 classes][64][K]
    const float* bias_tab;  // [25][64]
    half_t* out;
    const int* half_steps;  // [4 classes][nh][2]: {k offset, tapoff | slot << 8 | src << 10 | chunk32 << 13}
    const int* plan;        // [4 classes][nsteps][2]: halo prefetch of this K-step: {src | chunk32 << 3 | slot << 8, first piece | count << 8}
    int n, h, w, K, nh, nsteps;
    int c2_hp, c2_wp, c2_c, c2_ring, l3_hp, l3_wp, l3_ring, out_hp, out_wp, out_ring;
    int blocks_y, blocks_x, pairs;
};

template <int N>
__device__ __forceinline__ void hp_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ void hp_wait_dyn(int n) {  // wave-uniform count of loads that may stay in flight
    switch (n) {
        case 0: hp_wait_vmcnt<0>(); break;
        case 1: hp_wait_vmcnt<1>(); break;
        case 2: hp_wait_vmcnt<2>(); break;
        default: hp_wait_vmcnt<3>(); break;
    }
}

__global__ __launch_bounds__(512) void head_entry_pair_kernel(const PairParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tsel = w >> 2, wq = w & 3;                 // tile of the pair, wave within the tile (pixel rows 4*wq .. 4*wq+3)
    char* const hb = smem + tsel * HP_TILE_HALO;         // this tile's three halo slots
    char* const bring = smem + 2 * HP_TILE_HALO;
    int* const stab = (int*)(bring + HP_RING * HP_STAGE);
    int* const ptab = stab + 2 * p.nh;

    // ---- workgroup -> (image, class, pair of blocks); XCD-aware deal, the four classes of a pair are neighbours
    const int nblk = gridDim.x, b = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = b & 7;
    int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b >> 3);
    const int cls4 = t & 3;
    t >>= 2;
    const int pair = t % p.pairs;
    const int img = t / p.pairs;
    const int nblocks = p.blocks_y * p.blocks_x;
    int blk = 2 * pair + tsel;
    const bool tile_valid = blk < nblocks;
    blk = tile_valid ? blk : nblocks - 1;               // the idle half of an odd last pair recomputes the last block and stores nothing
    const int by = blk / p.blocks_x, bx = blk - by * p.blocks_x;
    const int py = cls4 >> 1, px = cls4 & 1;
    const int ly_min = py ? 0 : 1, lx_min = px ? 0 : 1;
    const int ly_cnt = p.h / 2 - 1, lx_cnt = p.w / 2 - 1;
    const int ly0 = ly_min + by * 16, lx0 = lx_min + bx * 16;
    const half_t* wcls = p.wgt + (int64_t)((py ? 2 : 1) * 4 + (px ? 2 : 1)) * 64 * p.K;
    const int fr = lane & 15, fq = lane >> 4;

    // ---- halo piece loader (own tile): piece = 16 rows x 64 B; LDS position (row, c) holds logical chunk c ^ ((row >> 1) & 2)
    const int hl_row = lane >> 2, hl_c = lane & 3;
    auto issue_halo_piece = [&](int src, int chunk, int slot, int piece) {
        int row = piece * 16 + hl_row;
        row = row < HP_ROWS ? row : HP_ROWS - 1;
        const int i = row / HP_HW, j = row - i * HP_HW;
        const int c_log = hl_c ^ ((row >> 1) & 2);
        const half_t* g;
        if (src < 4) {
            int yy = 2 * (ly0 - 1 + i) + (src >> 1) + p.c2_ring, xx = 2 * (lx0 - 1 + j) + (src & 1) + p.c2_ring;
            yy = yy < p.c2_hp ? yy : p.c2_hp - 1;
            xx = xx < p.c2_wp ? xx : p.c2_wp - 1;
            g = p.c2 + ((int64_t)(img * p.c2_hp + yy) * p.c2_wp + xx) * p.c2_c + chunk * 32 + c_log * 8;
        } else {
            int yy = ly0 - 1 + i + p.l3_ring, xx = lx0 - 1 + j + p.l3_ring;
            yy = yy < p.l3_hp ? yy : p.l3_hp - 1;
            xx = xx < p.l3_wp ? xx : p.l3_wp - 1;
            g = p.l3 + ((int64_t)(img * p.l3_hp + yy) * p.l3_wp + xx) * 256 + chunk * 32 + c_log * 8;
        }
        __builtin_amdgcn_global_load_lds((const VTD_AS1 void*)g, (VTD_AS3 void*)(hb + slot * HP_SLOT + piece * 1024), 16, 0, 0);
    };
    // ---- weight loader: wave w fills rows 8w .. 8w+7 of a stage ([64 ch][64 K] = half step a | half step b, chunks XOR-swizzled)
    const int wrow = w * 8 + (lane >> 3);
    const int wc_log = (lane & 7) ^ ((wrow >> 1) & 7);
    const half_t* wsrc = wcls + (int64_t)wrow * p.K + (wc_log & 3) * 8;
    const bool w_second = wc_log >= 4;
    auto issue_b = [&](int koff_a, int koff_b, int stage) {
        __builtin_amdgcn_global_load_lds((const VTD_AS1 void*)(wsrc + (w_second ? koff_b : koff_a)),
                                         (VTD_AS3 void*)(bring + stage * HP_STAGE + w * 1024), 16, 0, 0);
    };

    int hbase[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) hbase[j] = (wq * 4 + j) * HP_HW + fr;
    const int b_lane_off = fr * 128;
    const int bswz = (fr >> 1) & 7;
    floatx4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    {
        const int* hs = p.half_steps + (int64_t)cls4 * p.nh * 2;
        const int* pl = p.plan + (int64_t)cls4 * p.nsteps * 2;
        for (int i = tid; i < 2 * p.nh; i += 512) stab[i] = hs[i];
        for (int i = tid; i < 2 * p.nsteps; i += 512) ptab[i] = pl[i];
    }
    __syncthreads();
    auto koff_of = [&](int h) { return __builtin_amdgcn_readfirstlane(stab[2 * (h < p.nh ? h : p.nh - 1)]); };
    auto desc_of = [&](int h) { return __builtin_amdgcn_readfirstlane(stab[2 * (h < p.nh ? h : p.nh - 1) + 1]); };

    // ---- prologue: the first two groups' halos (their plan entries are the two leading pseudo-steps -1 / -2 encoded in ptab[0..3] of
    // the plan's own prologue slots, see vtd_head_entry_pair_tables) and the weights of K-steps 0..2
    {
        const int g0 = __builtin_amdgcn_readfirstlane(stab[1]);
        // group of half step 0
        for (int pc = wq; pc < HP_PIECES; pc += 4) issue_halo_piece((g0 >> 10) & 7, (g0 >> 13) & 7, (g0 >> 8) & 3, pc);
        // second group: first half step whose slot differs from group 0's
        int h1 = 1;
        while (h1 < p.nh && ((__builtin_amdgcn_readfirstlane(stab[2 * h1 + 1]) >> 8) & 3) == ((g0 >> 8) & 3)) ++h1;
        const int g1 = desc_of(h1);
        for (int pc = wq; pc < HP_PIECES; pc += 4) issue_halo_piece((g1 >> 10) & 7, (g1 >> 13) & 7, (g1 >> 8) & 3, pc);
#pragma unroll
        for (int a = 0; a < HP_RING - 1; ++a)
            if (a < p.nsteps) issue_b(koff_of(2 * a), koff_of(2 * a + 1), a);
    }

    auto load_a = [&](int desc, half8 (&af)[4]) {
        const char* hs = hb + ((desc >> 8) & 3) * HP_SLOT;
        const int tapoff = desc & 0xff;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int hrow = hbase[j] + tapoff;
            af[j] = *(const half8*)(hs + hrow * 64 + ((fq ^ ((hrow >> 1) & 2)) << 4));
        }
    };
    auto load_b = [&](int stage, int kk, half8 (&bf)[4]) {
        const char* sb = bring + stage * HP_STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) bf[i] = *(const half8*)(sb + b_lane_off + i * 2048 + (((fq + 4 * kk) ^ bswz) << 4));
    };

    half8 fa[2][4], fb[2][4];
    int issued_prev = 0;   // loads this wave issued in the previous K-step: the only ones allowed to be in flight at a step's barrier
    int d0 = desc_of(0), d1 = desc_of(1);
    for (int s = 0; s < p.nsteps; ++s) {
        // everything issued before the previous step has landed: weights of steps s and s+1, halos of every group that starts before
        // step s+2.  First step: the whole prologue.
        if (s == 0) hp_wait_vmcnt<0>(); else hp_wait_dyn(issued_prev);
        __builtin_amdgcn_s_barrier();
        const int stage = s & 3;
        int issued = 0;
        if (s + HP_RING - 1 < p.nsteps) {
            issue_b(koff_of(2 * (s + 3)), koff_of(2 * (s + 3) + 1), (s + 3) & 3);
            issued = 1;
        }
        {   // this step's share of the halo prefetch plan: pieces first + wq, first + wq + 4 (< first + count)
            const int pg = __builtin_amdgcn_readfirstlane(ptab[2 * s]), pr = __builtin_amdgcn_readfirstlane(ptab[2 * s + 1]);
            const int first = pr & 0xff, count = pr >> 8;
            if (wq < count) {
                issue_halo_piece(pg & 7, (pg >> 3) & 7, (pg >> 8) & 3, first + wq);
                ++issued;
            }
            if (wq + 4 < count) {
                issue_halo_piece(pg & 7, (pg >> 3) & 7, (pg >> 8) & 3, first + wq + 4);
                ++issued;
            }
        }
        issued_prev = issued;
        const int nd0 = desc_of(2 * s + 2), nd1 = desc_of(2 * s + 3);
        if (s == 0) {
            load_a(d0, fa[0]);
            load_b(stage, 0, fb[0]);
        }
        // half 0 multiplies while half 1's fragments are read
        load_a(d1, fa[1]);
        load_b(stage, 1, fb[1]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[0][i], fa[0][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
        __builtin_amdgcn_sched_barrier(0);
        // half 1 multiplies while the next step's half 0 fragments are read (its weights and halo landed before this step's barrier)
        if (s + 1 < p.nsteps) {
            load_a(nd0, fa[0]);
            load_b((s + 1) & 3, 0, fb[0]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[1][i], fa[1][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        d0 = nd0;
        d1 = nd1;
    }

    // ---- epilogue: accumulators -> fp32 LDS tile (one per tile of the pair) -> position-dependent bias, ReLU, 16-byte NHWC stores
    hp_wait_vmcnt<0>();
    __syncthreads();
    char* const et = smem + tsel * 256 * HP_EPI_ROW;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *(floatx4*)(et + (wq * 64 + j * 16 + fr) * HP_EPI_ROW + (i * 16 + fq * 4) * 4) = acc[i][j];
    __syncthreads();
    if (!tile_valid) return;
    const int lt = tid & 255, cc = lt & 7, r0 = lt >> 3;
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
        const floatx4 v0 = *(const floatx4*)(et + m * HP_EPI_ROW + cc * 32) + b0;
        const floatx4 v1 = *(const floatx4*)(et + m * HP_EPI_ROW + cc * 32 + 16) + b1;
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

// Tables of one interior class (py, px) for a C2 map of c2ch channels:
//   half_steps [nh][2]  {k offset into the class's [64][K] matrix, tapoff | slot << 8 | src << 10 | chunk32 << 13}
//   plan [nsteps][2]    halo pieces to request during K-step s: {src | chunk32 << 3 | slot << 8, first piece | count << 8}
// Groups (one source plane x 32 channels) alternate L3 / C2 while both remain; group g lives in halo slot g % 3 and is requested
// from the first K-step that begins after group g-3's last half step (its slot's previous tenant), 8 pieces per step.
// Returns nh (= 25 * c2ch / 32 + 72), or a negative value when the plan cannot be met.
int vtd_head_entry_pair_tables(int py, int px, int c2ch, int* half_steps, int* plan) {
    struct Group { int src, chunk, ntaps; int taps[9][2]; /* tapoff, koff */ int first_h, last_h; };
    Group groups[64];
    int ng = 0;
    const int nc2 = c2ch / 32;
    // C2 groups: chunk-major, planes inside
    Group c2g[32];
    int n_c2 = 0;
    for (int ch = 0; ch < nc2; ++ch)
        for (int plane = 0; plane < 4; ++plane) {
            Group g;
            g.src = plane; g.chunk = ch; g.ntaps = 0;
            for (int dy = 0; dy < 5; ++dy)
                for (int dx = 0; dx < 5; ++dx) {
                    const int ty = py + dy - 2, tx = px + dx - 2;
                    const int uy = ty >= 0 ? ty / 2 : -((-ty + 1) / 2), ux = tx >= 0 ? tx / 2 : -((-tx + 1) / 2);
                    if ((ty - 2 * uy) * 2 + (tx - 2 * ux) != plane) continue;
                    g.taps[g.ntaps][0] = (uy + 1) * 18 + (ux + 1);
                    g.taps[g.ntaps][1] = (dy * 5 + dx) * c2ch + ch * 32;
                    ++g.ntaps;
                }
            c2g[n_c2++] = g;
        }
    Group l3g[8];
    for (int ch = 0; ch < 8; ++ch) {
        Group g;
        g.src = 4; g.chunk = ch; g.ntaps = 9;
        for (int ij = 0; ij < 9; ++ij) {
            g.taps[ij][0] = (ij / 3) * 18 + (ij % 3);
            g.taps[ij][1] = 25 * c2ch + ij * 256 + ch * 32;
        }
        l3g[ch] = g;
    }
    for (int a = 0, b = 0; a < 8 || b < n_c2;) {
        if (a < 8) groups[ng++] = l3g[a++];
        if (b < n_c2) groups[ng++] = c2g[b++];
        if (a >= 8) while (b < n_c2) groups[ng++] = c2g[b++];
    }
    int h = 0;
    for (int g = 0; g < ng; ++g) {
        groups[g].first_h = h;
        for (int k = 0; k < groups[g].ntaps; ++k, ++h) {
            half_steps[2 * h] = groups[g].taps[k][1];
            half_steps[2 * h + 1] = groups[g].taps[k][0] | ((g % 3) << 8) | (groups[g].src << 10) | (groups[g].chunk << 13);
        }
        groups[g].last_h = h - 1;
    }
    const int nh = h;
    if (nh & 1) return -1;
    const int nsteps = nh / 2;
    for (int s = 0; s < nsteps; ++s) plan[2 * s] = plan[2 * s + 1] = 0;
    // groups 0 and 1 are fetched by the prologue; group g >= 2 from the first step after group g-3's tenant... slot g % 3 was used by
    // group g-3, free once the step holding that group's last half step has completed
    int next_free_step = 0;
    for (int g = 2; g < ng; ++g) {
        int start = g >= 3 ? groups[g - 3].last_h / 2 + 1 : 0;
        if (start < next_free_step) start = next_free_step;     // one group's pieces per step slot
        const int need_by = groups[g].first_h / 2;              // step that first reads this halo (maybe as its half 1, maybe prefetched
                                                                // one step earlier): everything must be issued by need_by - 3
        int piece = 0, s = start;
        while (piece < 21) {
            if (s >= nsteps || plan[2 * s + 1]) return -2;
            const int cnt = 21 - piece < 8 ? 21 - piece : 8;
            plan[2 * s] = groups[g].src | (groups[g].chunk << 3) | ((g % 3) << 8);
            plan[2 * s + 1] = piece | (cnt << 8);
            piece += cnt;
            ++s;
        }
        if (s - 1 > need_by - 3) return -3;
        next_free_step = s;
    }
    return nh;
}

int vtd_launch_head_entry_pair(const ConvParams& c, const int* half_steps_dev, const int* plan_dev, int nh, hipStream_t stream) {
    if (!c.plist || !c.in2 || !c.bias_tab || c.cout != 64 || c.in2_c != 256 || (c.in_c & 31) || c.in_y0 < 0 || (c.img_h & 1) || (c.img_w & 1) ||
        c.tiles_per_img <= 0 || nh != 25 * (c.in_c / 32) + 72 || c.K != 25 * c.in_c + 9 * 256)
        return -2501;
    PairParams p;
    p.c2 = c.in; p.l3 = c.in2; p.wgt = c.wgt; p.bias_tab = c.bias_tab; p.out = (half_t*)c.out; p.half_steps = half_steps_dev; p.plan = plan_dev;
    p.n = c.M / (c.tiles_per_img * 128); p.h = c.img_h; p.w = c.img_w; p.K = c.K; p.nh = nh; p.nsteps = nh / 2;
    p.c2_hp = c.in_hp; p.c2_wp = c.in_wp; p.c2_c = c.in_c; p.c2_ring = c.in_y0 + 2;
    p.l3_hp = c.in2_hp; p.l3_wp = c.in2_wp; p.l3_ring = c.in2_ring;
    p.out_hp = c.out_hp; p.out_wp = c.out_wp; p.out_ring = c.out_ring;
    const int cnt_y = c.img_h / 2 - 1, cnt_x = c.img_w / 2 - 1;
    p.blocks_y = (cnt_y + 15) / 16; p.blocks_x = (cnt_x + 15) / 16;
    p.pairs = (p.blocks_y * p.blocks_x + 1) / 2;
    if (p.n <= 0 || p.c2_ring < 2 || p.l3_ring < 1) return -2502;
    const int lds = 2 * HP_TILE_HALO + HP_RING * HP_STAGE + (2 * nh + 2 * p.nsteps) * 4;
    if (lds > 160 * 1024 || lds < 2 * 256 * HP_EPI_ROW) return -2503;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)head_entry_pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return -(int)e;
        attr = true;
    }
    hipLaunchKernelGGL(head_entry_pair_kernel, dim3(p.n * p.pairs * 4), dim3(512), lds, stream, p);
    return -(int)hipGetLastError();
}
