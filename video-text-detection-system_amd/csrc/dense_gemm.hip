// Dense GEMM for the Transformer recogniser's encoder pass:  out[M, N] = A[M, K] * W[N, K]^T + bias  (optionally exact GELU),
// A / W fp16 with K contiguous, fp32 accumulation, fp16 or fp32 output.  M is the token count of a whole crop batch (272 crops x 577
// tokens = 157 k rows), N = 768 ... 3072: thousands of 256 x 256 tiles, which is the regime where the implicit-GEMM kernel (conv_igemm:
// 256 x 128 tiles, one barrier per K-step, the two waves of a SIMD in lockstep behind it) leaves half of the matrix pipe idle.
//
// Structure (cdna_hip_programming.md, "The 256^2 8-phase template", re-derived for BK = 32 stages):
//   * 256 x 256 tile, 8 waves = 2 (M) x 4 (N), a wave owns 128 x 64 = 8 x 4 accumulator fragments (128 VGPRs);
//   * operands arrive by LDS-DMA (global_load_lds_dwordx4) in K stages of 32 through a ring of four 32 KB stages, three stages ahead
//     of the maths behind COUNTED s_waitcnt vmcnt -- never 0 inside the loop;
//   * 64-byte LDS rows, 16-byte chunks XOR-swizzled by dgm_key(row) on the SOURCE side (the DMA writes linearly) and on the read
//     side, so the 16 lanes of a ds_read_b128 group hit 16 different bank quads;
//   * a stage is two barrier intervals per wave: R (12 fragment reads of the stage + the wait that lets the NEXT stage's DMA land) and
//     MM (issue the DMA three stages ahead, 32 MFMAs).  The two wave groups (wm = 0 / 1: the two waves of every SIMD) run ONE interval
//     apart -- group 1 passes an extra barrier at the start, group 0 one at the end -- so while one wave of a SIMD multiplies, the other
//     reads its fragments, instead of both stalling on LDS and then both queueing for the pipe;
//   * hazards (checked in the comments at the barriers): a stage is read one interval AFTER the wait + barrier that published it, and
//     restaged only after a barrier that every reader reached with its fragments consumed by issued MFMAs;
//   * epilogue straight from the accumulators: weight rows are permuted on the source side so that a lane owns 8 consecutive output
//     channels -> bias, GELU, one 16-byte store (fp16) or two (fp32) per row and channel group.
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include "vtd_common.h"

namespace {

constexpr int DGM_BM = 256, DGM_BN = 256, DGM_BK = 32, DGM_NST = 4;
constexpr int DGM_STAGE = (DGM_BM + DGM_BN) * DGM_BK * 2;   // 32 KB
constexpr int DGM_LDS = DGM_NST * DGM_STAGE;                // 128 KB

struct DenseGemmParams {
    const half_t* A;     // [M][lda]
    const half_t* W;     // [w_rows][K] (w_rows >= N, padded with zeros)
    const float* bias;   // [>= N rounded up to 8]
    void* out;           // fp16 / fp32 [M][ldc]
    int M, N, K, lda, ldc, w_rows, tiles_n, flags;   // flags: EPI_OUT_F16 | EPI_OUT_F32 | EPI_GELU
    int tiles_m, gn, blocks_n;                       // tile order: super-blocks of DGM_GM x gn tiles per XCD (gn = 0: row-major runs)
    int nvb;                                         // virtual block ids to walk (tiles incl. the padding of the last super-block row)
    // Operand addressing, in elements: element (row, k) of A lives at row * a_rs + (k / 32) * a_ks + k % 32.  Row-major: a_rs = lda, a_ks = 32.
    // K-PANEL-MAJOR ([K / 32][rows][32]: a_rs = 32, a_ks = 32 x the producer's row count): the 16 rows x 64 bytes of an LDS-DMA piece are then
    // 1 KB of contiguous memory -- eight whole cache lines -- where the row-major layout makes them sixteen half lines (the BK = 32 stage is
    // 64 bytes of a row); measured on the encoder's shapes with a timing-only variant first: +7...10 % (tools/dense_gemm_bench.py).
    int a_rs, b_rs;
    int64_t a_ks, b_ks;
    int64_t out_prows;   // fp16 output in the same panel layout for the next GEMM ([N / 32][out_prows][32]); 0: row-major [M][ldc]
};
constexpr int DGM_GM = 8;

// Swizzle key of a 64-byte LDS row.  ds_read_b128 is served in four groups of 16 lanes that are NOT consecutive lanes
// (MI355X_MICROARCH.md, LDS: {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32): a group holds fragment rows 0-3 and 12-15 of
// one K chunk fq and rows 4-11 of chunk fq ^ 1.  Four rows share a 256-byte bank row, so rows j, j + 4, j + 8, j + 12 (row >> 2 =
// t = 0 .. 3) must land on four different chunks: {fq ^ k(0), (fq ^ 1) ^ k(1), (fq ^ 1) ^ k(2), fq ^ k(3)} distinct <=> k = (0, 3, 2, 1)
// = -t mod 4.  The first version used k(t) = t -- right for groups of consecutive lanes, a 2-way conflict on every read for the real
// ones (SQ_LDS_BANK_CONFLICT = half of SQ_LDS_IDX_ACTIVE, tools/gpu_pmc_dgm.sh).
__device__ __forceinline__ int dgm_key(int row) { return (0 - (row >> 2)) & 3; }

template <int N>
__device__ __forceinline__ void dgm_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Exact-erf GELU, 0.5 x (1 + erf(x / sqrt 2)), with erf from Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 + one ulp of the reciprocal, branch-free: one v_rcp_f32,
// one v_exp_f32, five fused multiply-adds) -- three orders of magnitude below the fp16 rounding of the value it produces; libdevice's erff
// expands to a two-branch polynomial whose eight interleaved copies per row do not fit beside 128 accumulators.
__device__ __forceinline__ float dgm_gelu(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);   // v_rcp_f32 (1 ulp): __frcp_rn is the ten-instruction IEEE division
    const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
    const float erf_abs = 1.0f - poly * __expf(-z * z);
    return 0.5f * x * (1.0f + copysignf(erf_abs, x));
}

__device__ __forceinline__ void dgm_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

struct DgmTile {
    int a[2], b[2];   // element offsets of this lane's chunk slots from p.A / p.W (both operands stay below 2^31 elements: launcher)
    int m0, n0;
};

// virtual block id -> tile (false: padding of the last super-block row / past the end)
__device__ __forceinline__ bool dgm_locate(const DenseGemmParams& p, int vb, int& tm, int& tn) {
    if (p.gn > 0) {
        // Super-block order: virtual block ids that are congruent mod 8 are worked by one XCD (one L2): XCD x takes super-blocks x, x + 8,
        // ... and its CUs work through a super-block of DGM_GM x gn tiles together: DGM_GM A panels + gn W panels (~0.4 MB each at
        // K = 768) serve 8 gn tiles out of that L2, instead of one fresh W panel per tile when N K exceeds the L2.
        const int xcd = vb & 7, local = vb >> 3, per = DGM_GM * p.gn;
        const int blk = (local / per) * 8 + xcd, idx = local - (local / per) * per;
        const int bm = blk / p.blocks_n, bn = blk - bm * p.blocks_n;
        tm = bm * DGM_GM + idx / p.gn;
        tn = bn * p.gn + idx % p.gn;
        return tm < p.tiles_m;
    }
    const int total = p.tiles_m * p.tiles_n;   // row-major runs: XCD x owns the x-th eighth of the tiles
    const int q = total >> 3, r8 = total & 7, xcd = vb & 7, local = vb >> 3;
    const int cnt = q + (xcd < r8 ? 1 : 0);
    if (local >= cnt) return false;
    const int tile = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + local;
    tm = tile / p.tiles_n;
    tn = tile - tm * p.tiles_n;
    return true;
}

template <bool GELU, int VAR = 0>   // Product: <false, 16> and <true, 1>; everything else only with -DVTD_DGM_EXPERIMENT (launcher below).
                                    // VAR (tools/dense_gemm_bench.py): 1 = wave groups in phase, 2 = no s_setprio, 4 = contiguous fetch (timing only),
                                    // 8 = all four LDS-DMA instructions of a stage in the R interval, 16 = the A pieces in R and the B pieces in MM,
                                    // 32 = no LDS-DMA after the prologue, 64 = fragments read once (both timing only, wrong results)
__global__ __launch_bounds__(512) void dense_gemm_kernel(const DenseGemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char dgm_smem[];
    const int tid = threadIdx.x, lane = tid & 63, fr = lane & 15, fq = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = w >> 2, wn = w & 3;
    const int G = gridDim.x;   // a multiple of 8: a workgroup's virtual block ids b, b + G, ... stay on its XCD

    // ---- loader: wave w brings pieces w and w + 8 (16 rows x 64 bytes each) of the A tile and of the B tile of every stage.
    // Lane l of a piece lands at LDS row 16 piece + (l >> 2), physical chunk l & 3, and therefore fetches logical chunk
    // (l & 3) ^ dgm_key(row) of that row.
    auto make_tile = [&](int tm, int tn, DgmTile& t) {
        t.m0 = tm * DGM_BM;
        t.n0 = tn * DGM_BN;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (w + 8 * i) * 16 + (lane >> 2);
            const int chunk = (lane & 3) ^ dgm_key(row);
            int m = t.m0 + row;
            m = m < p.M ? m : p.M - 1;                       // rows past the end re-read the last row (never written)
            t.a[i] = m * p.a_rs + chunk * 8;
            // B row `row` of the tile feeds MFMA row (row & 15) of fragment (row >> 4) & 3 of wave column row >> 6; it is fed with the
            // weights of the channel that makes a lane's accumulators 8 consecutive channels (see the epilogue)
            const int x = row & 63, blk = x >> 4, rr = x & 15;
            int n = t.n0 + (row - x) + 32 * (blk >> 1) + 8 * (rr >> 2) + 4 * (blk & 1) + (rr & 3);
            n = n < p.w_rows ? n : p.w_rows - 1;
            t.b[i] = n * p.b_rs + chunk * 8;
        }
    };
    int g = 0;                    // global stage counter of this workgroup: stage s of its t-th tile is g = t S + s, ring slot g & 3
    auto issue = [&](const DgmTile& t, int stage, int slot, int part = 3) {   // K offset 32 stage of tile t into ring slot `slot` (part: 1 = A pieces, 2 = B pieces)
        if constexpr ((VAR & 32) != 0) { if (g >= 1) return; }   // TIMING ONLY (wrong results): no operand traffic after the prologue
        char* base = dgm_smem + slot * DGM_STAGE;
        const int k = stage * DGM_BK;
        const half_t* const Ak = p.A + (int64_t)stage * p.a_ks;   // (wave-uniform: a scalar base; the lane adds its 32-bit offset)
        const half_t* const Wk = p.W + (int64_t)stage * p.b_ks;
        (void)k;
        if constexpr ((VAR & 4) != 0) {
            // TIMING EXPERIMENT ONLY (wrong results): every piece reads 1 KB of CONTIGUOUS memory (eight full 128-byte lines) instead of
            // sixteen 64-byte half lines -- same instruction count, same bytes: does the half-line gather cost anything?
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const half_t* a = p.A + (int64_t)t.m0 * p.lda + ((int64_t)(stage * 16 + w + 8 * i) * 512 + lane * 8);
                __builtin_amdgcn_global_load_lds((const VTD_AS1 void*)a, (VTD_AS3 void*)(base + (w + 8 * i) * 1024), 16, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int n0c = t.n0 < p.w_rows - 256 ? t.n0 : 0;
                const half_t* bsrc_ = p.W + (int64_t)n0c * p.K + ((int64_t)(stage * 16 + w + 8 * i) * 512 + lane * 8);
                __builtin_amdgcn_global_load_lds((const VTD_AS1 void*)bsrc_, (VTD_AS3 void*)(base + DGM_BM * 64 + (w + 8 * i) * 1024), 16, 0, 0);
            }
            return;
        }
        if (part & 1) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
                __builtin_amdgcn_global_load_lds((const VTD_AS1 void*)(Ak + (unsigned)t.a[i]), (VTD_AS3 void*)(base + (w + 8 * i) * 1024), 16, 0, 0);
        }
        if (part & 2) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
                __builtin_amdgcn_global_load_lds((const VTD_AS1 void*)(Wk + (unsigned)t.b[i]), (VTD_AS3 void*)(base + DGM_BM * 64 + (w + 8 * i) * 1024), 16, 0, 0);
        }
    };
    // Where a stage's four LDS-DMA instructions are issued: PART_R of them in the R interval (behind the fragment reads, while the other
    // wave of the SIMD multiplies), the rest at the top of the MM interval (in front of this wave's own 32 MFMAs, where nothing covers them)
    constexpr int PART_R = (VAR & 8) ? 3 : (VAR & 16) ? 1 : 0, PART_MM = 3 & ~PART_R;
    constexpr int N_R = (PART_R & 1 ? 2 : 0) + (PART_R & 2 ? 2 : 0);

    // ---- fragment read offsets (bytes inside a stage): A rows wm * 128 + 16 i + fr, B rows wn * 64 + 16 j + fr, logical chunk fq
    int a_off[8], b_off[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = wm * 128 + i * 16 + fr;
        a_off[i] = row * 64 + ((fq ^ dgm_key(row)) << 4);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = wn * 64 + j * 16 + fr;
        b_off[j] = DGM_BM * 64 + row * 64 + ((fq ^ dgm_key(row)) << 4);
    }

    floatx4 acc[8][4];
    // epilogue from registers: lane (fr, fq) holds row m0 + wm * 128 + 16 i + fr, channels n0 + wn * 64 + 32 jp + 8 fq + 0..7
    // (acc[i][2 jp][0..3] then acc[i][2 jp + 1][0..3], by the row permutation of the loader)
    auto epilogue = [&](int m0, int n0) {
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
            const int n = n0 + wn * 64 + jp * 32 + fq * 8;
            if (n >= p.N) continue;
            const floatx4 b0 = *(const floatx4*)(p.bias + n), b1 = *(const floatx4*)(p.bias + n + 4);
            // output element (m, n) at m * o_rs + coff: row-major o_rs = ldc, coff = n; panel-major o_rs = 32, coff = (n / 32) * 32 out_prows + n % 32
            const int64_t o_rs = p.out_prows ? 32 : p.ldc;
            const int64_t coff = p.out_prows ? ((int64_t)(n >> 5) * p.out_prows << 5) + (n & 31) : n;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int m = m0 + wm * 128 + i * 16 + fr;
                if (m >= p.M) continue;
                floatx4 v0 = acc[i][2 * jp] + b0, v1 = acc[i][2 * jp + 1] + b1;
                if constexpr (GELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v0[e] = dgm_gelu(v0[e]); v1[e] = dgm_gelu(v1[e]); }
                }
                if (p.flags & EPI_OUT_F16) {
                    half8 h;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { h[e] = (half_t)v0[e]; h[4 + e] = (half_t)v1[e]; }
                    *(half8*)((half_t*)p.out + (int64_t)m * o_rs + coff) = h;
                } else {
                    float* o = (float*)p.out + (int64_t)m * o_rs + coff;
                    *(floatx4*)o = v0;
                    *(floatx4*)(o + 4) = v1;
                }
                if constexpr (GELU) __builtin_amdgcn_sched_barrier(0);   // one row's GELU at a time (register budget)
            }
        }
    };

    // ---- this workgroup's tiles: virtual blocks b, b + G, ...  (persistent: the operand ring runs on ACROSS tiles, so a tile's prologue
    // latency and its epilogue hide under the neighbouring tiles' stages)
    int vb = blockIdx.x, tm = 0, tn = 0;
    while (vb < p.nvb && !dgm_locate(p, vb, tm, tn)) vb += G;
    if (vb >= p.nvb) return;   // uniform over the workgroup, before any barrier
    DgmTile cur, nxt;
    make_tile(tm, tn, cur);
    nxt = cur;

    const int S = p.K / DGM_BK;   // >= 4 (launcher)
    issue(cur, 0, 0);
    issue(cur, 1, 1);
    issue(cur, 2, 2);
    dgm_wait_vmcnt<8>();          // this wave's pieces of stage 0 have landed (stages 1, 2 stay in flight)
    dgm_barrier();                // hw barrier 0: stage 0 is published
    if ((VAR & 1) == 0 && wm == 1) dgm_barrier();   // group 1 runs one interval behind group 0 from here on

    half8 af[8], bf[4];
    bool have_prev = false;
    int prev_m0 = 0, prev_n0 = 0;
    for (;;) {
        int vb2 = vb + G, tm2 = 0, tn2 = 0;
        while (vb2 < p.nvb && !dgm_locate(p, vb2, tm2, tn2)) vb2 += G;
        const bool has_next = vb2 < p.nvb;
        if (has_next) make_tile(tm2, tn2, nxt);
        for (int s = 0; s < S; ++s, ++g) {
            // ---- interval R: fragments of stage g.  (Published: every wave waited for its pieces of this stage at the end of its
            // previous R interval -- or in the prologue -- and has passed a barrier since.)
            const char* st = dgm_smem + (g & (DGM_NST - 1)) * DGM_STAGE;
            // Stage g + 1 must have landed before the barrier that ends this interval; stage g + 2 (issued in the previous MM) may stay
            // in flight.  Loads return in order, so "at most 4 outstanding" means stage g + 1 is in.
            const bool more3 = s + 3 < S || has_next;   // stage g + 3 exists
            if (s == 0 && have_prev) {
                // first stage of a new tile: the previous tile's accumulators leave from here, while the other wave of this SIMD
                // multiplies.  The wait comes BEFORE the stores (they share the counter and retire in any order: behind them the same
                // wait would also sit out their acknowledgements), the fragment reads after them (their registers are free until then).
                dgm_wait_vmcnt<4>();
                epilogue(prev_m0, prev_n0);
                __builtin_amdgcn_sched_barrier(0);
                if ((VAR & 64) == 0 || g == 0) {   // (64: TIMING ONLY, fragments are read once)
#pragma unroll
                for (int j = 0; j < 4; ++j) bf[j] = *(const half8*)(st + b_off[j]);
#pragma unroll
                for (int i = 0; i < 8; ++i) af[i] = *(const half8*)(st + a_off[i]);
                }
                if constexpr (PART_R != 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (s + 3 < S) issue(cur, s + 3, (g + 3) & (DGM_NST - 1), PART_R);
                    else if (has_next) issue(nxt, s + 3 - S, (g + 3) & (DGM_NST - 1), PART_R);
                }
            } else {
                if ((VAR & 64) == 0 || g == 0) {   // (64: TIMING ONLY, fragments are read once)
#pragma unroll
                for (int j = 0; j < 4; ++j) bf[j] = *(const half8*)(st + b_off[j]);
#pragma unroll
                for (int i = 0; i < 8; ++i) af[i] = *(const half8*)(st + a_off[i]);
                }
                if constexpr (PART_R != 0) {
                    // Ring slot (g + 3) & 3 = (g - 1) & 3 is restaged one interval EARLIER than in the MM placement: its last readers
                    // were this group's R interval of stage g - 1 (two intervals ago) and the other group's, which ended at the barrier
                    // just passed -- behind an s_waitcnt lgkmcnt(0) (below), so those reads are complete, not merely issued.
                    __builtin_amdgcn_sched_barrier(0);
                    if (s + 3 < S) issue(cur, s + 3, (g + 3) & (DGM_NST - 1), PART_R);
                    else if (has_next) issue(nxt, s + 3 - S, (g + 3) & (DGM_NST - 1), PART_R);
                    // stage g + 1 has landed when only stage g + 2 and what was just issued of stage g + 3 are outstanding
                    if (more3) dgm_wait_vmcnt<4 + N_R>(); else if (s + 2 < S) dgm_wait_vmcnt<4>(); else dgm_wait_vmcnt<0>();
                } else {
                    if (s + 2 < S || has_next) dgm_wait_vmcnt<4>(); else dgm_wait_vmcnt<0>();
                }
            }
            if constexpr (PART_R != 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's fragment reads are COMPLETE at the barrier
            dgm_barrier();
            // ---- interval MM: restage ring slot (g + 3) & 3 = (g - 1) & 3.  Its last readers were the R intervals of stage g - 1: this
            // group's ended two barriers ago, the other group's one barrier ago at the latest, and every wave reached that barrier only
            // after its MFMAs of the MM interval of stage g - 1 -- which consume those fragments -- had been issued.
            if constexpr (PART_MM != 0) {
                if (s + 3 < S) issue(cur, s + 3, (g + 3) & (DGM_NST - 1), PART_MM);
                else if (has_next) issue(nxt, s + 3 - S, (g + 3) & (DGM_NST - 1), PART_MM);
            }
            if ((VAR & 2) == 0) __builtin_amdgcn_s_setprio(1);
            if (s == 0) {   // a tile's first stage starts its accumulators
                const floatx4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], zero, 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
            }
            if ((VAR & 2) == 0) __builtin_amdgcn_s_setprio(0);
            dgm_barrier();
        }
        have_prev = true;
        prev_m0 = cur.m0;
        prev_n0 = cur.n0;
        if (!has_next) break;
        cur = nxt;
        vb = vb2;
    }
    if ((VAR & 1) == 0 && wm == 0) dgm_barrier();   // group 0 pays back group 1's extra barrier (every wave executes the same number of barriers)
    epilogue(prev_m0, prev_n0);
}

}  // namespace

// Applicable when the shape is a plain dense layer the kernel's vector accesses fit: K a multiple of 32 with at least 4 stages, N a
// multiple of 8, 16-byte aligned rows.  Worth it when there are enough 256 x 256 tiles to fill the chip a few times over.
bool vtd_dense_gemm_supported(int64_t M, int N, int K, int lda, int ldc, int flags) {
    if (M < 256 || N < 256 || (N & 7) || (K & 31) || K < 128 || (lda & 7) || (ldc & 7)) return false;
    if (!(flags & (EPI_OUT_F16 | EPI_OUT_F32)) || (flags & ~(EPI_OUT_F16 | EPI_OUT_F32 | EPI_GELU))) return false;
    const int64_t tiles = ((M + DGM_BM - 1) / DGM_BM) * ((N + DGM_BN - 1) / DGM_BN);
    return tiles >= 512;
}

// a_prows / out_prows: row count of the K-panel-major A operand / fp16 output ([K / 32][prows][32]), 0 for row-major; w_panel: the weights are
// [K / 32][w_rows][32].
int vtd_launch_dense_gemm_ex(const half_t* A, int lda, int64_t a_prows, const half_t* W, int w_rows, int w_panel, const float* bias, void* out, int ldc,
                             int64_t out_prows, int64_t M, int N, int K, int flags, hipStream_t stream) {
    if (M < 256 || N < 256 || (N & 7) || (K & 31) || K < 128 || (lda & 7) || (ldc & 7) || M > 0x7fffffff ||
        !(flags & (EPI_OUT_F16 | EPI_OUT_F32)) || (flags & ~(EPI_OUT_F16 | EPI_OUT_F32 | EPI_GELU)) || (int64_t)w_rows * K >= 0x7fffffff)
        return -2601;
    if ((a_prows && (a_prows < M || a_prows * 32 >= 0x7fffffff)) || (out_prows && (out_prows < M || !(flags & EPI_OUT_F16) || (N & 31)))) return -2601;
    if (!a_prows && M * (int64_t)lda >= 0x7fffffff) {
        // the kernel addresses a row-major A with 32-bit element offsets: a taller operand (a recogniser pass of > ~1200 crops at the
        // 3072-wide fc2 input) runs as row blocks, each a launch of its own on the same stream -- every output row is computed exactly as
        // before.  (A panel-major A has 32 elements per row and panel: no such bound.)
        const int64_t rows = ((int64_t)0x7ffffffe / lda) / DGM_BM * DGM_BM;
        if (rows < DGM_BM) return -2601;
        const size_t esz = (flags & EPI_OUT_F16) ? sizeof(half_t) : sizeof(float);
        auto out_at = [&](int64_t m0) { return out_prows ? (void*)((half_t*)out + m0 * 32) : (void*)((char*)out + (size_t)m0 * ldc * esz); };
        for (int64_t m0 = 0; m0 < M; m0 += rows) {
            const int64_t mb = std::min(rows, M - m0);
            int rc;
            if (mb >= 256) rc = vtd_launch_dense_gemm_ex(A + m0 * lda, lda, 0, W, w_rows, w_panel, bias, out_at(m0), ldc, out_prows, mb, N, K, flags, stream);
            else rc = vtd_launch_dense_gemm_ex(A + (M - 256) * lda, lda, 0, W, w_rows, w_panel, bias, out_at(M - 256), ldc, out_prows, 256, N, K, flags, stream);  // a short tail: the last 256 rows again (same values)
            if (rc) return rc;
        }
        return 0;
    }
    DenseGemmParams p{A, W, bias, out, (int)M, N, K, lda, ldc, w_rows, (N + DGM_BN - 1) / DGM_BN, flags, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    p.a_rs = a_prows ? 32 : lda; p.a_ks = a_prows ? a_prows * 32 : 32;
    p.b_rs = w_panel ? 32 : K;   p.b_ks = w_panel ? (int64_t)w_rows * 32 : 32;
    p.out_prows = out_prows;
    p.tiles_m = (int)((M + DGM_BM - 1) / DGM_BM);
    // measurement switches are read ONCE per process (never per launch); 0: row-major runs per XCD (A/B measurements)
    static const int order = [] { const char* e = std::getenv("VTD_DGM_ORDER"); return e ? std::atoi(e) : 1; }();
    if (order) {
        for (int g = 4; g >= 1; --g)
            if (p.tiles_n % g == 0) { p.gn = g; break; }
        p.blocks_n = p.tiles_n / p.gn;
    }
    int64_t nvb = ((int64_t)p.tiles_m * p.tiles_n + 7) / 8 * 8;
    if (p.gn) {   // whole super-blocks, a multiple of 8 of them (one run per XCD)
        const int64_t blocks = (int64_t)((p.tiles_m + DGM_GM - 1) / DGM_GM) * p.blocks_n;
        nvb = (blocks + 7) / 8 * 8 * DGM_GM * p.gn;
    }
    p.nvb = (int)nvb;
    int grid = 256;   // one persistent workgroup per CU; fewer when there are fewer tiles (always a multiple of 8)
    static const int grid_env = [] { const char* e = std::getenv("VTD_DGM_GRID"); return e ? std::max(8, std::atoi(e) / 8 * 8) : 0; }();
    if (grid_env) grid = grid_env;
    else {
        // a stream confined to part of the chip (vtd_stream_create_masked: the encoder pass beside a decode): one workgroup per CU it may
        // use -- a 256-workgroup grid on 176 CUs would run as one full round and a 45 % round behind it
        uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipExtStreamGetCUMask(stream, 8, mask) == hipSuccess) {
            int cus = 0;
            for (uint32_t w : mask) cus += __builtin_popcount(w);
            if (cus >= 8 && cus < 256) grid = cus / 8 * 8;
        } else {
            (void)hipGetLastError();
        }
    }
    if (nvb < grid) grid = (int)nvb;
    // Product placement of the LDS-DMA instructions: A pieces in the R interval, B pieces in the MM interval (template VAR = 16; bitwise
    // identical to the all-in-MM placement VAR = 0, +3...5 % on the encoder's shapes).  The GELU GEMM runs its two wave groups IN PHASE
    // (VAR = 1).  Staggered, group 0's epilogue (128 values x ~17 VALU per lane: ~8.7 k cycles) fills one barrier interval and group 1's the
    // next -- the other group has one 512-cycle MM to do and then waits, so a tile pays the epilogue twice; in phase both epilogues share
    // one interval.  That is worth more than the stagger is at K = 768: 1.22 -> 1.14 ms (bitwise identical).
    // The product library holds exactly these two instantiations (tests/test_abi.py checks the symbol table).  The loop-structure A/B
    // variants and the knock-out variants that produce WRONG results exist only in an instrumented build (-DVTD_DGM_EXPERIMENT,
    // tools/dense_gemm_bench.py), where VTD_DGM_VARIANT selects them.
    static std::once_flag attr_once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(attr_once, [] {
        auto set = [](const void* f) {
            const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, DGM_LDS);
            if (e != hipSuccess) attr_err = e;
        };
        set((const void*)dense_gemm_kernel<false, 16>);
        set((const void*)dense_gemm_kernel<true, 1>);
#ifdef VTD_DGM_EXPERIMENT
        set((const void*)dense_gemm_kernel<true, 16>);
        set((const void*)dense_gemm_kernel<false, 0>);
        set((const void*)dense_gemm_kernel<true, 0>);
        set((const void*)dense_gemm_kernel<false, 1>);
        set((const void*)dense_gemm_kernel<false, 2>);
        set((const void*)dense_gemm_kernel<false, 4>);
        set((const void*)dense_gemm_kernel<false, 8>);
        set((const void*)dense_gemm_kernel<false, 32>);
        set((const void*)dense_gemm_kernel<false, 64>);
        set((const void*)dense_gemm_kernel<false, 96>);
#endif
    });
    VTD_HIP_CHECK(attr_err);
    const dim3 gd((unsigned)grid), bd(512);
#ifdef VTD_DGM_EXPERIMENT
    int var = 16;   // (instrumented build only: read per launch so tools/dense_gemm_bench.py can walk the variants in one process)
    if (const char* e = std::getenv("VTD_DGM_VARIANT")) var = std::atoi(e);
    if (flags & EPI_GELU) {
        if (var == 116) hipLaunchKernelGGL((dense_gemm_kernel<true, 16>), gd, bd, DGM_LDS, stream, p);
        else if (var == 100) hipLaunchKernelGGL((dense_gemm_kernel<true, 0>), gd, bd, DGM_LDS, stream, p);
        else hipLaunchKernelGGL((dense_gemm_kernel<true, 1>), gd, bd, DGM_LDS, stream, p);
    } else {
        switch (var) {
            case 0: hipLaunchKernelGGL((dense_gemm_kernel<false, 0>), gd, bd, DGM_LDS, stream, p); break;
            case 1: hipLaunchKernelGGL((dense_gemm_kernel<false, 1>), gd, bd, DGM_LDS, stream, p); break;
            case 2: hipLaunchKernelGGL((dense_gemm_kernel<false, 2>), gd, bd, DGM_LDS, stream, p); break;
            case 4: hipLaunchKernelGGL((dense_gemm_kernel<false, 4>), gd, bd, DGM_LDS, stream, p); break;
            case 8: hipLaunchKernelGGL((dense_gemm_kernel<false, 8>), gd, bd, DGM_LDS, stream, p); break;
            case 32: hipLaunchKernelGGL((dense_gemm_kernel<false, 32>), gd, bd, DGM_LDS, stream, p); break;
            case 64: hipLaunchKernelGGL((dense_gemm_kernel<false, 64>), gd, bd, DGM_LDS, stream, p); break;
            case 96: hipLaunchKernelGGL((dense_gemm_kernel<false, 96>), gd, bd, DGM_LDS, stream, p); break;
            default: hipLaunchKernelGGL((dense_gemm_kernel<false, 16>), gd, bd, DGM_LDS, stream, p); break;
        }
    }
#else
    if (flags & EPI_GELU) hipLaunchKernelGGL((dense_gemm_kernel<true, 1>), gd, bd, DGM_LDS, stream, p);
    else hipLaunchKernelGGL((dense_gemm_kernel<false, 16>), gd, bd, DGM_LDS, stream, p);
#endif
    return -(int)hipGetLastError();
}

int vtd_launch_dense_gemm(const half_t* A, int lda, const half_t* W, int w_rows, const float* bias, void* out, int ldc, int64_t M, int N, int K,
                          int flags, hipStream_t stream) {
    return vtd_launch_dense_gemm_ex(A, lda, 0, W, w_rows, 0, bias, out, ldc, 0, M, N, K, flags, stream);
}
