// Composed DB-head entry convolution, interior parity classes, fourth generation: head_entry_halo256_kernel with the halo fetch
// taken off the critical path.  (Reference graph: FPN lateral(C2) + top-down add + P2 smooth 3x3 + head conv 3x3 + BN + ReLU,
// app/ml/models/text_detector.py:31-75, composed on the host into one convolution -- vtd_api.cpp: compose_head_entry.)
//
// head_entry_halo256 walks K in eight groups (four source-parity planes of C2, four 64-channel chunks of L3), each on its own
// 18 x 18 x 64-channel halo.  The halo buffer cannot be refilled before every wave is done with it, so every group switch was a
// barrier, 41 LDS-DMA pieces and a full round trip to L2 / HBM with nothing to compute: its stamps charge ~18 k of a tile's 112 k
// K-loop cycles to the eight switches.  A second halo buffer does not fit beside the weight ring at two workgroups per CU.
//
// Here a halo is two HALF halos of 32 channels, H0 and H1 (324 rows of 64 bytes each), and a group is walked as
//     all taps on channels 0-31 (H0), then all taps on channels 32-63 (H1).
// While the H1 half of group g is being multiplied, H0 is free and receives the first half of group g+1; while that is being
// multiplied, H1 receives the second.  Every half-halo fetch is issued at least two K-steps (~3.6 k cycles) before its first use,
// behind the step barrier that retires its predecessor, and is waited for with the counted vmcnt that the weight ring needs
// anyway.  A K-step (barrier to barrier, 32 MFMAs per wave, one 8 KB weight stage) is a pair of consecutive half-steps, each with
// its own tap, half-halo and 32 weight columns; the schedule is a host-built table (vtd_head_entry_half_schedule, which also
// replays it to check that no half-halo is read before it can have landed or refilled while a step still reads it).
//
// 64-byte halo rows change the banking: a ds_read_b128 lane group now covers four 256-byte bank rows, and the 16-byte chunk
// of a row is XOR-ed with 2 * ((column >> 2) & 1) -- found by exhaustive search, conflict-free for the three tap columns and
// every row alignment.  As in head_entry_halo256 the key depends on the halo COLUMN only, which is the same for the four pixels
// a lane owns: one address per half-step, four reads at immediate offsets.
#include <cstdlib>
#include <vector>
#include "vtd_common.h"

namespace {

constexpr int HH_HW = 18, HH_ROWS = 324, HH_PIECES = 21, HH_HALF_BYTES = HH_PIECES * 1024, HH_RING = 4;
constexpr int HH_BSTAGE = 64 * 128;
constexpr int HH_EPI_ROW = 64 * 4 + 16;
constexpr int HH_DMA = 6;  // half-halo LDS-DMA instructions per wave (21 pieces over 4 waves; the spare ones repeat piece 20)

struct HeadHalfParams {
    const half_t* c2;      // [n][c2_hp][c2_wp][c2_c], ring c2_ring >= 2
    const half_t* l3;      // [n][l3_hp][l3_wp][256], ring l3_ring >= 1
    const half_t* wgt;     // [16 classes][64][K] fp16 (compose_head_entry's layout)
    const float* bias_tab; // [25][64]
    half_t* out;           // [n][out_hp][out_wp][64]
    const int* sched;      // [4 interior classes][nsteps][4]: see vtd_head_entry_half_schedule
    int n, h, w, K, nsteps;
    int c2_hp, c2_wp, c2_c, c2_ring, l3_hp, l3_wp, l3_ring, out_hp, out_wp, out_ring;
    int blocks_y, blocks_x;
    unsigned long long* stamps;  // debug (VTD_HALO_STAMPS=1): per workgroup {wait+barrier, compute, -, K loop total} cycles
};

template <int N>
__device__ __forceinline__ void hh_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <bool STAMPS>
__global__ __launch_bounds__(256, 2) void head_entry_half_kernel(const HeadHalfParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const hbuf = smem;                                  // [2][HH_HALF_BYTES]
    char* const bring = smem + 2 * HH_HALF_BYTES;             // [HH_RING][HH_BSTAGE]
    int* const stab = (int*)(bring + HH_RING * HH_BSTAGE);   // [nsteps][4]

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
    const int* sched = p.sched + (int64_t)cls4 * p.nsteps * 4;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane >> 3, fr = lane & 15, fq = lane >> 4;

    // this lane's six half-halo rows (tile independent): halo row i, column x, and the logical 16-byte chunk it fetches
    int hrow_i[HH_DMA], hrow_x[HH_DMA], hchunk[HH_DMA], hpiece[HH_DMA];
#pragma unroll
    for (int k = 0; k < HH_DMA; ++k) {
        int piece = w + 4 * k;
        piece = piece < HH_PIECES ? piece : HH_PIECES - 1;
        int row = piece * 16 + (lane >> 2);
        row = row < HH_ROWS ? row : HH_ROWS - 1;
        const int i = row / HH_HW, x = row - i * HH_HW;
        hrow_i[k] = i; hrow_x[k] = x; hpiece[k] = piece;
        hchunk[k] = (lane & 3) ^ (((x >> 2) & 1) << 1);
    }
    // one half halo (channels 32 half .. 32 half + 31 of 64-channel chunk `chunk` of source `src`) into H[half]
    auto issue_half = [&](int src, int chunk, int half) {
#pragma unroll
        for (int k = 0; k < HH_DMA; ++k) {
            const half_t* g;
            if (src < 4) {
                int yy = 2 * (ly0 - 1 + hrow_i[k]) + (src >> 1) + p.c2_ring, xx = 2 * (lx0 - 1 + hrow_x[k]) + (src & 1) + p.c2_ring;
                yy = yy < p.c2_hp ? yy : p.c2_hp - 1;
                xx = xx < p.c2_wp ? xx : p.c2_wp - 1;
                g = p.c2 + ((int64_t)(img * p.c2_hp + yy) * p.c2_wp + xx) * p.c2_c + chunk * 64 + half * 32 + hchunk[k] * 8;
            } else {
                int yy = ly0 - 1 + hrow_i[k] + p.l3_ring, xx = lx0 - 1 + hrow_x[k] + p.l3_ring;
                yy = yy < p.l3_hp ? yy : p.l3_hp - 1;
                xx = xx < p.l3_wp ? xx : p.l3_wp - 1;
                g = p.l3 + ((int64_t)(img * p.l3_hp + yy) * p.l3_wp + xx) * 256 + chunk * 64 + half * 32 + hchunk[k] * 8;
            }
            __builtin_amdgcn_global_load_lds((const VTD_AS1 void*)g, (VTD_AS3 void*)(hbuf + half * HH_HALF_BYTES + hpiece[k] * 1024), 16, 0, 0);
        }
    };
    // weight stage of one K-step: 64 rows x (32 columns at k0 | 32 columns at k1), chunks XOR-swizzled by (row >> 1) & 7
    const half_t* bsrc[2];
    int bsel[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (i * 4 + w) * 8 + lrow;
        const int c_log = (lane & 7) ^ ((row >> 1) & 7);
        bsrc[i] = wcls + (int64_t)row * p.K + (c_log & 3) * 8;
        bsel[i] = c_log >> 2;
    }
    auto issue_b = [&](int kk, int stage) {
        const int k0 = kk & 0xffff, k1 = (kk >> 16) & 0xffff;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const VTD_AS1 void*)(bsrc[i] + (bsel[i] ? k1 : k0)),
                                             (VTD_AS3 void*)(bring + stage * HH_BSTAGE + (i * 4 + w) * 1024), 16, 0, 0);
    };

    const int b_lane_off = fr * 128;
    const int bswz = (fr >> 1) & 7;
    floatx4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    for (int i = tid; i < 4 * p.nsteps; i += 256) stab[i] = sched[i];
    __syncthreads();
    // a step's four descriptor words: d0, d1 = half-steps (tap row offset | tap column << 8 | half halo << 10 | fresh << 11),
    // kk = weight column offsets k0 | k1 << 16, aux = half-halo fetch to issue at the top of the step (valid << 15 | src | chunk << 3 | half << 7)
    auto word = [&](int s, int k) { return stab[4 * (s < p.nsteps ? s : p.nsteps - 1) + k]; };
    auto uni = [&](int v) { return __builtin_amdgcn_readfirstlane(v); };

    const int a_lane_off = ((w * 4) * HH_HW + fr) * 64;
    auto load_frags = [&](int d, int stage, int kk, half8 (&af)[4], half8 (&bf)[4]) {
        const int dx = (d >> 8) & 3;
        const char* pa = hbuf + ((d >> 10) & 1) * HH_HALF_BYTES + a_lane_off + (d & 0xff) * 64 + ((fq ^ ((((fr + dx) >> 2) & 1) << 1)) << 4);
        const char* sb = bring + stage * HH_BSTAGE;
#pragma unroll
        for (int j = 0; j < 4; ++j) af[j] = *(const half8*)(pa + j * (HH_HW * 64));
#pragma unroll
        for (int i = 0; i < 4; ++i) bf[i] = *(const half8*)(sb + b_lane_off + i * 2048 + (((fq + 4 * kk) ^ bswz) << 4));
    };

    // prologue: both halves of the first group, three weight stages
    {
        const int a0 = uni(word(0, 3));  // step 0 carries the first group's source in its aux word (valid bit clear: not a refill)
        issue_half(a0 & 7, (a0 >> 3) & 15, 0);
        issue_half(a0 & 7, (a0 >> 3) & 15, 1);
    }
#pragma unroll
    for (int a = 0; a < HH_RING - 1; ++a)
        if (a < p.nsteps) issue_b(uni(word(a, 2)), a);

    int d0 = uni(word(0, 0)), d1 = uni(word(0, 1));
    int n0 = uni(word(1, 0)), n1 = uni(word(1, 1));
    int aux = 0;                      // step 0 issues nothing
    int v_kk = word(HH_RING - 1, 2);  // kk(s+3), aux(s+1), d0(s+2), d1(s+2) for s = 0: vector copies, made uniform after the wait
    int v_aux = word(1, 3), v_n0 = word(2, 0), v_n1 = word(2, 1);
    half8 fa[2][4], fb[2][4];
    bool have_frags = false, issued_prev = false;
    int stage = 0;
    unsigned long long t0 = 0, t1 = 0, a_wait = 0, a_comp = 0, t_begin = 0;
    if constexpr (STAMPS) t_begin = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < p.nsteps; ++s) {
        if constexpr (STAMPS) t0 = __builtin_amdgcn_s_memtime();
        // Loads land in order.  Allowed in flight: the newest weight stage (2 loads) and the half halo issued at the top of the
        // previous step (6); everything older -- the weights of steps s and s+1, every half halo issued two or more steps ago --
        // has landed.  The schedule never needs a half halo sooner than two steps after its issue.
        if (s + 2 >= p.nsteps) hh_wait_vmcnt<0>(); else if (issued_prev) hh_wait_vmcnt<2 + HH_DMA>(); else hh_wait_vmcnt<2>();
        __builtin_amdgcn_s_barrier();
        if constexpr (STAMPS) t1 = __builtin_amdgcn_s_memtime();
        const int kk3 = uni(v_kk), naux = uni(v_aux), nn0 = uni(v_n0), nn1 = uni(v_n1);
        if (s + HH_RING - 1 < p.nsteps) issue_b(kk3, (stage + HH_RING - 1) & 3);
        issued_prev = (aux >> 15) & 1;
        if (issued_prev) issue_half(aux & 7, (aux >> 3) & 15, (aux >> 7) & 1);  // every wave is past the step that last read this half
        if (!have_frags) load_frags(d0, stage, 0, fa[0], fb[0]);
        load_frags(d1, stage, 1, fa[1], fb[1]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[0][i], fa[0][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
        __builtin_amdgcn_sched_barrier(0);
        // half 1: prefetch the first half-step of the next step unless it opens a freshly fetched half halo (that one is only
        // guaranteed behind the next barrier), then the descriptors of the steps after it
        const bool pre = s + 1 < p.nsteps && !((n0 >> 11) & 1);
        if (pre) load_frags(n0, (stage + 1) & 3, 0, fa[0], fb[0]);
        v_kk = word(s + HH_RING, 2);
        v_aux = word(s + 2, 3);
        v_n0 = word(s + 3, 0);
        v_n1 = word(s + 3, 1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[1][i], fa[1][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        have_frags = pre;
        if constexpr (STAMPS) {
            asm volatile("s_nop 0" ::"v"(acc[0][0][0]));
            a_wait += t1 - t0;
            a_comp += __builtin_amdgcn_s_memtime() - t1;
        }
        d0 = n0; d1 = n1;
        n0 = nn0; n1 = nn1;
        aux = naux;
        stage = (stage + 1) & 3;
    }

    if (STAMPS && tid == 0) {
        unsigned long long* o = p.stamps + (int64_t)blockIdx.x * 4;
        o[0] = a_wait; o[1] = a_comp; o[2] = 0; o[3] = __builtin_amdgcn_s_memtime() - t_begin;
    }
    // ---- epilogue: accumulators -> fp32 LDS tile -> position-dependent bias, ReLU, 16-byte NHWC stores
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *(floatx4*)(smem + (w * 64 + j * 16 + fr) * HH_EPI_ROW + (i * 16 + fq * 4) * 4) = acc[i][j];
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
        const floatx4 v0 = *(const floatx4*)(smem + m * HH_EPI_ROW + cc * 32) + b0;
        const floatx4 v1 = *(const floatx4*)(smem + m * HH_EPI_ROW + cc * 32 + 16) + b1;
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

// Half-step schedule of one interior class from its head_entry_halo256 step table (vtd_head_entry_halo_steps: per K-step
// {k offset, tap row offset | first << 8 | src << 9 | chunk << 12 | tap column << 16}, groups opened by `first`).
// out[s] = {d0, d1, k0 | k1 << 16, aux}; returns 0, or a negative code if the replay finds a hazard (never for the shapes the
// launcher admits; the check is what the parity tests lean on for the asynchronous part).
extern "C" int vtd_head_entry_half_schedule(const int* steps, int nsteps, int* out /* [nsteps][4] */) {
    struct Half { int koff, tapoff, dx, group, half; };
    std::vector<Half> hs;
    std::vector<int> gsrc, gchunk, gstart, gcount;
    for (int s = 0; s < nsteps;) {
        int e = s + 1;
        while (e < nsteps && !((steps[2 * e + 1] >> 8) & 1)) ++e;
        const int g = (int)gsrc.size();
        gsrc.push_back((steps[2 * s + 1] >> 9) & 7);
        gchunk.push_back((steps[2 * s + 1] >> 12) & 15);
        gstart.push_back((int)hs.size());
        gcount.push_back(e - s);
        for (int half = 0; half < 2; ++half)
            for (int t = s; t < e; ++t)
                hs.push_back({steps[2 * t] + 32 * half, steps[2 * t + 1] & 0xff, (steps[2 * t + 1] >> 16) & 3, g, half});
        s = e;
    }
    if ((int)hs.size() != 2 * nsteps || gsrc.empty()) return -2501;
    const int ng = (int)gsrc.size();
    // issue step of half halo (g, half), g >= 1: the step after the one that holds the last half-step on (g-1, half)
    std::vector<int> aux(nsteps, 0), fresh(2 * nsteps, 0);
    aux[0] = gsrc[0] | (gchunk[0] << 3);  // prologue source, valid bit clear
    for (int g = 1; g < ng; ++g)
        for (int half = 0; half < 2; ++half) {
            const int last_prev = gstart[g - 1] + half * gcount[g - 1] + gcount[g - 1] - 1;
            const int first_use = gstart[g] + half * gcount[g];
            const int s_issue = (last_prev >> 1) + 1, s_use = first_use >> 1;
            if (s_issue < 1 || s_issue >= nsteps || s_use - s_issue < 2 || (aux[s_issue] >> 15)) return -2502;
            aux[s_issue] = (1 << 15) | gsrc[g] | (gchunk[g] << 3) | (half << 7);
            // the first use must not be prefetched across the barrier that publishes the fetch
            if ((first_use & 1) == 0) fresh[first_use] = 1;
        }
    for (int s = 0; s < nsteps; ++s) {
        for (int k = 0; k < 2; ++k) {
            const Half& h = hs[2 * s + k];
            if (h.koff < 0 || h.koff > 0xffff || h.tapoff > 0xff) return -2503;
            out[4 * s + k] = h.tapoff | (h.dx << 8) | (h.half << 10) | (fresh[2 * s + k] << 11);
        }
        out[4 * s + 2] = hs[2 * s].koff | (hs[2 * s + 1].koff << 16);
        out[4 * s + 3] = aux[s];
    }
    // replay: content[half] = group held (as of the barrier at the top of a step), landed_at[half] = first step that may read it
    int content[2] = {0, 0}, ready[2] = {0, 0};
    for (int s = 0; s < nsteps; ++s) {
        if ((aux[s] >> 15) & 1) {
            const int half = (aux[s] >> 7) & 1;
            // no half-step of step s or later may still want the old content; steps < s are retired by the barrier
            for (int k = 2 * s; k < 2 * nsteps; ++k)
                if (hs[k].half == half && hs[k].group == content[half]) return -2504;
            content[half] += 1;
            ready[half] = s + 2;
        }
        for (int k = 0; k < 2; ++k) {
            const Half& h = hs[2 * s + k];
            if (content[h.half] != h.group || s < ready[h.half]) return -2505;
            // a first half-step is read one step early (prefetch) unless flagged fresh
            if (k == 0 && !fresh[2 * s] && s > 0 && s - 1 < ready[h.half] && ready[h.half] > 0) return -2506;
        }
    }
    return 0;
}

int vtd_launch_head_entry_half(const ConvParams& c, const int* sched_dev, int nsteps, hipStream_t stream) {
    if (!c.plist || !c.in2 || !c.bias_tab || c.cout != 64 || c.in2_c != 256 || (c.in_c & 63) || c.in_y0 < 0 || (c.img_h & 1) || (c.img_w & 1) ||
        c.tiles_per_img <= 0 || nsteps != 25 * (c.in_c / 64) + 36 || c.K != 25 * c.in_c + 9 * 256 || c.K > 0xffff || !sched_dev)
        return -2511;
    HeadHalfParams p;
    p.c2 = c.in; p.l3 = c.in2; p.wgt = c.wgt; p.bias_tab = c.bias_tab; p.out = (half_t*)c.out; p.sched = sched_dev;
    p.n = c.M / (c.tiles_per_img * 128); p.h = c.img_h; p.w = c.img_w; p.K = c.K; p.nsteps = nsteps;
    p.c2_hp = c.in_hp; p.c2_wp = c.in_wp; p.c2_c = c.in_c; p.c2_ring = c.in_y0 + 2;  // in_y0 = ring - 2
    p.l3_hp = c.in2_hp; p.l3_wp = c.in2_wp; p.l3_ring = c.in2_ring;
    p.out_hp = c.out_hp; p.out_wp = c.out_wp; p.out_ring = c.out_ring;
    const int cnt_y = c.img_h / 2 - 1, cnt_x = c.img_w / 2 - 1;
    p.blocks_y = (cnt_y + 15) / 16; p.blocks_x = (cnt_x + 15) / 16;
    if (p.n <= 0 || p.c2_ring < 2 || p.l3_ring < 1) return -2512;
    p.stamps = nullptr;
    const int lds = 2 * HH_HALF_BYTES + HH_RING * HH_BSTAGE + 4 * nsteps * 4;
    if (lds < 256 * HH_EPI_ROW || 2 * lds > 160 * 1024) return -2513;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)head_entry_half_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)head_entry_half_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return -(int)e;
        attr = true;
    }
    const int grid = p.n * p.blocks_y * p.blocks_x * 4;
    static const bool stamps = [] { const char* e = getenv("VTD_HALO_STAMPS"); return e && e[0] == '1'; }();
    if (stamps) {  // debug: where a workgroup's K loop goes (synchronises!)
        unsigned long long* dev = nullptr;
        if (hipMalloc(&dev, (size_t)grid * 32) != hipSuccess) return -2514;
        p.stamps = dev;
        hipLaunchKernelGGL(head_entry_half_kernel<true>, dim3(grid), dim3(256), lds, stream, p);
        (void)hipStreamSynchronize(stream);
        std::vector<unsigned long long> h((size_t)grid * 4);
        (void)hipMemcpy(h.data(), dev, (size_t)grid * 32, hipMemcpyDeviceToHost);
        double a = 0, b = 0, d = 0;
        for (int i = 0; i < grid; ++i) { a += (double)h[4 * i]; b += (double)h[4 * i + 1]; d += (double)h[4 * i + 3]; }
        fprintf(stderr, "[head_entry_half stamps] grid %d: per workgroup cycles: wait+barrier %.0f  compute %.0f  K loop total %.0f\n", grid, a / grid, b / grid, d / grid);
        (void)hipFree(dev);
        return 0;
    }
    hipLaunchKernelGGL(head_entry_half_kernel<false>, dim3(grid), dim3(256), lds, stream, p);
    return -(int)hipGetLastError();
}
