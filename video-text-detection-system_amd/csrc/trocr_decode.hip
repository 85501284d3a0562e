// Autoregressive decoder of the Transformer recogniser (TrOCR; text_recognizer.py:55-60 -> transformers TrOCRForCausalLM + greedy
// generate) as a chain of small gfx950 kernels over a COMPACT list of live rows.
//
// Shape of the problem: one query row per crop and step (M = a few hundred rows at most, shrinking as rows emit </s>), 12 layers,
// ~11 dependent launches per layer.  Nothing here is MFMA- or HBM-bound except the cross-attention read of the encoder keys / values;
// what decides the step time is the fixed cost of every launch in the chain, so the kernels are built for latency:
//
//  * dec_gemm: out[M, N] = A[M, K] * W[N, K]^T for M <= a few hundred.  A workgroup owns a (16 MF) x (16 NF) output tile; its four
//    waves split the K range four ways (no barrier, no LDS in the K loop: every wave streams its own fragments straight into
//    registers, four 32-deep sub-steps per batch of loads) and meet once, through LDS, for the final sum.  Optional split-K over
//    workgroups writes fp32 partial slabs that the NEXT kernel of the chain (the LayerNorm) sums -- the launch-boundary combine of
//    cdna_hip_programming.md section 5 ("Projection GEMM at M = 256", item 2): no atomics, fixed summation order, bitwise repeatable.
//  * dec_ln: residual + bias + split-K slabs -> LayerNorm -> fp32 stream + fp16 GEMM input, one wave per row.
//  * dec_attn<SELF | CROSS>: four waves per (row, head), eight lanes per 128-byte key / value row, three rounds of 32 keys in flight;
//    the self-attention workgroup also files the step's new key / value into the crop's cache rows.
//  * dec_argmax + dec_advance: greedy token, GenerationMixin's </s> / <pad> bookkeeping, and the COMPACTION of the row list: rows that
//    emitted </s> leave `active[]`, so every later kernel runs on the live rows only (per-row buffers are indexed by list position j,
//    caches / ids / encoder keys by crop = active[j]).  The live count goes to device memory (exact, read by every kernel) and to
//    pinned host memory (read by the host two steps late as an upper bound for the launch geometry and as the stop test).
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <cstdlib>
#include "vtd_common.h"

namespace {

constexpr int DG_OUT_F16 = 1, DG_GELU = 2, DG_PARTIAL = 4;

struct DecGemmParams {
    const half_t* A;      // [rows][lda] fp16
    const half_t* W;      // [N rounded up to the tile][K] fp16 (build_linear's layout)
    const float* bias;    // [N rounded up to the tile] (ignored for DG_PARTIAL)
    void* out;            // fp16 / fp32 [rows][ldc]; DG_PARTIAL: fp32 slabs, slab z at out + z * slab_stride
    int64_t slab_stride;
    const int* n_rows;    // device: exact live row count (<= M)
    int lda, ldc, M, N, K, ksplit, flags;
};

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

template <int MF, int NF, int RF>
__device__ __forceinline__ void finish_fragment(const DecGemmParams& p, const float* red, int f, int pass, int lane, int fr, int fq, int m0, int n0,
                                                int rows) {
        floatx4 v = *(const floatx4*)(red + ((0 * RF + f - pass) * 64 + lane) * 4);
#pragma unroll
        for (int q = 1; q < 4; ++q) v += *(const floatx4*)(red + ((q * RF + f - pass) * 64 + lane) * 4);   // fixed order: wave 0 + 1 + 2 + 3
        const int i = f / NF, j = f - i * NF;
        const int m = m0 + i * 16 + fr, n = n0 + j * 16 + fq * 4;   // lane: row fr, four consecutive columns
        if (m >= rows || n >= p.N) return;
        if (p.flags & DG_PARTIAL) {
            float* o = (float*)p.out + (int64_t)blockIdx.z * p.slab_stride + (int64_t)m * p.ldc + n;
            if (n + 3 < p.N) *(floatx4*)o = v;
            else
                for (int e = 0; e < 4 && n + e < p.N; ++e) o[e] = v[e];
            return;
        }
        v += *(const floatx4*)(p.bias + n);   // the bias vector is padded with zeros to the tile width (build_linear)
        if (p.flags & DG_GELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
        }
        if (p.flags & DG_OUT_F16) {
            half_t* o = (half_t*)p.out + (int64_t)m * p.ldc + n;
            if (n + 3 < p.N) *(half4*)o = half4{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
            else
                for (int e = 0; e < 4 && n + e < p.N; ++e) o[e] = (half_t)v[e];
        } else {
            float* o = (float*)p.out + (int64_t)m * p.ldc + n;
            if (n + 3 < p.N && !(((uintptr_t)o) & 15)) *(floatx4*)o = v;
            else
                for (int e = 0; e < 4 && n + e < p.N; ++e) o[e] = v[e];
        }
}

template <int MF, int NF>
__global__ __launch_bounds__(256) void dec_gemm_kernel(DecGemmParams p) {
    __shared__ __attribute__((aligned(16))) float red[4 * (MF * NF < 8 ? MF * NF : 8) * 256];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, fr = lane & 15, fq = lane >> 4;
    // The live row count is only needed to mask the stores: it is fetched here and consumed after the K loop, so its round trip is
    // not at the head of every launch's dependent chain.  (Row tiles past the live rows -- at most the two steps' worth the host's
    // bound lags behind -- read stale rows of valid buffers and write nothing.)
    const int rows = min(*p.n_rows, p.M);
    const int m0 = blockIdx.y * (16 * MF), n0 = blockIdx.x * (16 * NF);
    // the workgroup's K range in 32-deep sub-steps, dealt to the four waves as evenly as they divide (K = 192: 1 + 2 + 1 + 2)
    const int kper = p.K / p.ksplit, nsub = kper >> 5;     // host guarantees kper % 32 == 0
    const int sub0 = (nsub * w) >> 2, klen = (((nsub * (w + 1)) >> 2) - sub0) << 5;
    const int kbeg = blockIdx.z * kper + (sub0 << 5);
    // rows past the live count are read (the buffers hold max_crops rows, rounded up to the tile) and never written
    const half_t* ap = p.A + (int64_t)(m0 + fr) * p.lda + kbeg + fq * 8;
    const half_t* wp = p.W + (int64_t)(n0 + fr) * p.K + kbeg + fq * 8;
    floatx4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    // 32-deep sub-steps whose loads are issued together (<= 24 x 16 bytes per lane in flight): a 16-row tile walks K = 1024 / 4 waves in ONE
    // round trip, a 64-row tile in two
    constexpr int U = 24 / (MF + NF) >= 8 ? 8 : 24 / (MF + NF) >= 6 ? 6 : 24 / (MF + NF) >= 4 ? 4 : 2;
    for (int k = 0; k < klen; k += 32 * U) {
        half8 a[U][MF], b[U][NF];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kk = (k + 32 * u < klen) ? k + 32 * u : k;   // a short tail re-reads its first sub-step (result unused)
#pragma unroll
            for (int j = 0; j < NF; ++j) b[u][j] = *(const half8*)(wp + (int64_t)j * 16 * p.K + kk);
#pragma unroll
            for (int i = 0; i < MF; ++i) a[u][i] = *(const half8*)(ap + (int64_t)i * 16 * p.lda + kk);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (k + 32 * u < klen) {
#pragma unroll
                for (int i = 0; i < MF; ++i)
#pragma unroll
                    for (int j = 0; j < NF; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[u][j], a[u][i], acc[i][j], 0, 0, 0);
            }
    }
    // four K quarters -> one tile: the waves park their accumulators (eight fragments per pass: 32 KB of LDS), then wave w finishes
    // fragments w, w + 4, ... of the pass
    constexpr int RF = MF * NF < 8 ? MF * NF : 8;
#pragma unroll
    for (int pass = 0; pass < MF * NF; pass += RF) {
        if (pass) __syncthreads();
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                const int f = i * NF + j;
                if (f >= pass && f < pass + RF) *(floatx4*)(red + ((w * RF + f - pass) * 64 + lane) * 4) = acc[i][j];
            }
        __syncthreads();
        for (int f = pass + w; f < pass + RF; f += 4) finish_fragment<MF, NF, RF>(p, red, f, pass, lane, fr, fq, m0, n0, rows);
    }
}

// ---- the same GEMM for a TALL live list (thousands of rows: the first steps of a pass of many tickets).  dec_gemm_kernel streams its
// operands from L2 straight into registers -- right for a few hundred rows, 140 TFLOP/s at 3264.  This one stages them through LDS:
// 128 x 128 outputs per workgroup on 16 waves = 4 K-QUARTERS x (2 x 2 waves of 64 x 64).  The arithmetic is dec_gemm_kernel's, to the
// bit: a wave group accumulates exactly the K quarter that wave `kq` of the small kernel accumulates (same 32-deep sub-steps, same order,
// same MFMA operand roles), and the four quarters are summed 0 + 1 + 2 + 3 by the same finish_fragment -- so which of the two kernels ran,
// a function of the host's lagged row bound, never shows in a result.
constexpr int DT_BM = 128, DT_BN = 128;
constexpr int DT_STAGE = 4 * (DT_BM + DT_BN) * 64;   // bytes per K-step of 32: four quarters x (A tile + W tile), 64-byte rows
constexpr int DT_LDS = 2 * DT_STAGE;                  // 128 KB, double buffered; the quarter sums reuse it

__device__ __forceinline__ int dt_key(int row) { return (0 - (row >> 2)) & 3; }   // chunk swizzle of a 64-byte row (as dense_gemm.hip: conflict-free
                                                                                  // for the hardware's ds_read_b128 lane groups)

__global__ __launch_bounds__(1024) void dec_gemm_tall_kernel(DecGemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char dt_smem[];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, fr = lane & 15, fq = lane >> 4;
    const int kq = w >> 2, wm = (w >> 1) & 1, wn = w & 1;
    const int rows = min(*p.n_rows, p.M);
    const int m0 = blockIdx.y * DT_BM, n0 = blockIdx.x * DT_BN;
    const int kper = p.K / p.ksplit, nsub = kper >> 5;
    // quarter q walks sub-steps [sub0(q), sub0(q + 1)) of the workgroup's K range, as wave q of dec_gemm_kernel does
    const int n_pad = (p.N + 63) / 64 * 64;
    int steps_max = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) steps_max = max(steps_max, ((nsub * (q + 1)) >> 2) - ((nsub * q) >> 2));
    const int my_steps = ((nsub * (kq + 1)) >> 2) - ((nsub * kq) >> 2);

    // loader: 4096 16-byte chunks per stage, four per thread: chunk id c = tid + 1024 j -> quarter c >> 10, then 512 A chunks (row, k-chunk) and
    // 512 W chunks.  Rows past the bound re-read the last row (never stored); W rows past the padded N likewise.
    const half_t* src[4];
    int dst[4], lsteps[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = tid + 1024 * j, q = c >> 10, r = c & 1023, isw = r >> 9, row = (r & 511) >> 2, ch = r & 3;
        const int sub0 = (nsub * q) >> 2;
        lsteps[j] = ((nsub * (q + 1)) >> 2) - sub0;
        const int kbeg = blockIdx.z * kper + (sub0 << 5) + ch * 8;
        if (isw) {
            int n = n0 + row;
            n = n < n_pad ? n : n_pad - 1;
            src[j] = p.W + (int64_t)n * p.K + kbeg;
        } else {
            int m = m0 + row;
            m = m < p.M ? m : p.M - 1;
            src[j] = p.A + (int64_t)m * p.lda + kbeg;
        }
        dst[j] = q * ((DT_BM + DT_BN) * 64) + (isw ? DT_BM * 64 : 0) + row * 64 + ((ch ^ dt_key(row)) << 4);
    }
    half8 stg[4];
    auto fetch = [&](int s) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (s < lsteps[j]) stg[j] = *(const half8*)(src[j] + s * 32);
    };
    auto stage = [&](int s, int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (s < lsteps[j]) *(half8*)(dt_smem + buf * DT_STAGE + dst[j]) = stg[j];
    };
    floatx4 acc[4][4];   // [m fragment][n fragment] of this wave's 64 x 64
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    const char* const qbase = dt_smem + kq * ((DT_BM + DT_BN) * 64);
    int a_off[4], b_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ra = wm * 64 + i * 16 + fr, rb = wn * 64 + i * 16 + fr;
        a_off[i] = ra * 64 + ((fq ^ dt_key(ra)) << 4);
        b_off[i] = DT_BM * 64 + rb * 64 + ((fq ^ dt_key(rb)) << 4);
    }
    fetch(0);
    stage(0, 0);
    __syncthreads();
    for (int s = 0; s < steps_max; ++s) {
        if (s + 1 < steps_max) fetch(s + 1);             // in flight under this step's maths
        if (s < my_steps) {
            const char* st = qbase + (s & 1) * DT_STAGE;
            half8 af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { af[i] = *(const half8*)(st + a_off[i]); bf[i] = *(const half8*)(st + b_off[i]); }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
        }
        if (s + 1 < steps_max) stage(s + 1, (s + 1) & 1);   // the other buffer: last read in step s - 1, behind that step's barrier
        __syncthreads();
    }
    // quarter sums through LDS (the staging buffers are free behind the last barrier): four fragments per pass; the four waves of a spatial
    // position park theirs, then wave kq finishes fragment pass + kq with dec_gemm_kernel's own epilogue
    float* const red = (float*)dt_smem + (wm * 2 + wn) * (4 * 4 * 256);   // [quarter][4 fragments][64 lanes][4]
#pragma unroll
    for (int pass = 0; pass < 16; pass += 4) {
        if (pass) __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int f = i * 4 + j;
                if (f >= pass && f < pass + 4) *(floatx4*)(red + ((kq * 4 + f - pass) * 64 + lane) * 4) = acc[i][j];
            }
        __syncthreads();
        finish_fragment<4, 4, 4>(p, red, pass + kq, pass, lane, fr, fq, m0 + wm * 64, n0 + wn * 64, rows);
    }
}

// ---- LayerNorm of the decoder (post-LN): t = x + bias + sum of split-K slabs (residual mode) or embed[token] + position (embedding
// mode); x <- LN(t), x16 <- fp16(LN(t)).  One wave per row, float4 accesses (D % 64 == 0, D <= 2048).
struct DecLnParams {
    float* x;              // [rows][D] residual stream (in / out)
    half_t* x16;           // [rows][D]
    const float* slabs;    // residual mode: [nsplit][slab_stride] fp32 partial sums, row stride D
    int64_t slab_stride;
    int nsplit;
    const float* bias;     // [D]
    const float *gamma, *beta;
    const int* n_rows;
    int M, D;
    float eps;
    // embedding mode (embed != null)
    const float *embed, *pos;     // [vocab][D], [positions + 2][D]
    const int32_t* ids;           // [crops][ld_ids]
    const int32_t* active;        // [rows] -> crop
    int ld_ids, col, position, vocab;
};

// NS = number of split-K slabs, a template parameter: with a run-time trip count the slab loop is not unrolled and its loads go out
// one after the other -- eight dependent memory round trips in the LayerNorm behind fc2 (measured: 12.5 us per launch, the largest
// single item of a decode).
template <int MAXV4, int NS>
__global__ __launch_bounds__(256) void dec_ln_kernel(DecLnParams p) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= p.M) return;
    const bool live = row < *p.n_rows;   // consumed at the stores only (see dec_gemm): not at the head of the dependent chain
    floatx4 v[MAXV4], gam[MAXV4], bet[MAXV4];
    // gamma / beta are fetched with the data, not after the two reductions (one memory round trip less in every launch of the chain)
#pragma unroll
    for (int i = 0; i < MAXV4; ++i) {
        const int c = (i * 64 + lane) * 4;
        gam[i] = bet[i] = floatx4{0.f, 0.f, 0.f, 0.f};
        if (c < p.D) { gam[i] = *(const floatx4*)(p.gamma + c); bet[i] = *(const floatx4*)(p.beta + c); }
    }
    float sum = 0.f;
    const float* erow = nullptr;
    if (p.embed) {
        int tok = p.ids[(int64_t)p.active[row] * p.ld_ids + p.col];
        tok = tok < 0 ? 0 : tok >= p.vocab ? p.vocab - 1 : tok;
        erow = p.embed + (int64_t)tok * p.D;
    }
#pragma unroll
    for (int i = 0; i < MAXV4; ++i) {
        const int c = (i * 64 + lane) * 4;
        floatx4 t = {0.f, 0.f, 0.f, 0.f};
        if (c < p.D) {   // D % 4 == 0: a lane's four columns are in or out together
            if (p.embed) {
                t = *(const floatx4*)(erow + c) + *(const floatx4*)(p.pos + (int64_t)(p.position + 2) * p.D + c);
            } else {
                t = *(const floatx4*)(p.x + (int64_t)row * p.D + c) + *(const floatx4*)(p.bias + c);
                floatx4 sl[NS > 0 ? NS : 1];
#pragma unroll
                for (int s = 0; s < NS; ++s) sl[s] = *(const floatx4*)(p.slabs + (int64_t)s * p.slab_stride + (int64_t)row * p.D + c);
#pragma unroll
                for (int s = 0; s < NS; ++s) t += sl[s];   // slab order 0, 1, ...: the same sum as before
            }
        }
        v[i] = t;
        sum += (t[0] + t[1]) + (t[2] + t[3]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum / (float)p.D;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV4; ++i)
        if ((i * 64 + lane) * 4 < p.D) {
            const floatx4 d = v[i] - mean;
            sq += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    const float rstd = 1.0f / sqrtf(sq / (float)p.D + p.eps);
#pragma unroll
    for (int i = 0; i < MAXV4; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < p.D) {
            const floatx4 n = (v[i] - mean) * rstd * gam[i] + bet[i];
            if (live) {
                *(floatx4*)(p.x + (int64_t)row * p.D + c) = n;
                *(half4*)(p.x16 + (int64_t)row * p.D + c) = half4{(half_t)n[0], (half_t)n[1], (half_t)n[2], (half_t)n[3]};
            }
        }
    }
}

// ---- attention of one query row per (list row j, head) against L keys.  K / V rows of a head are one 128-byte line: eight lanes
// read it (16 bytes each), a wave-instruction covers eight keys in full lines.  SELF: q | k | v of the step come from the fused
// projection's [rows][3D] buffer; the wave files k / v into the crop's cache row `step` and attends to cache rows 0..step-1 plus the
// new one.  CROSS: keys / values of the crop's encoder tokens ([crops][T][D], written once per crop by the encoder pass).
struct DecAttnParams {
    const half_t* q;       // SELF: [rows][3D] (q at col 0, k at D, v at 2D), pre-scaled q; CROSS: [rows][D]
    int ldq;
    half_t *kc, *vc;       // SELF: caches [crops][Lmax][D]; CROSS: encoder keys / values [crops][T][D]
    int64_t crop_stride;   // elements between crops
    int L;                 // SELF: step + 1 keys (the last one is the new one); CROSS: T
    int D;
    half_t* out;           // [rows][D]
    const int32_t* active;
    const int* n_rows;
    int M;
};

// Workgroup = NW waves per (row, head) (4 for self-, 16 for cross-attention).  The first version gave a (row, head) to ONE wave that walked its keys eight at a time:
// 73 dependent load -> use rounds per pass for the 577 encoder tokens, i.e. ~100 us per launch however few rows were live (measured:
// 110 us at 64 live rows, 0.17 of the HBM rate).  Now the workgroup covers 8 NW keys per round (eight lanes per 128-byte row), three
// rounds are in flight per thread -- the score pass and the value pass are each two dependent batches of loads for 577 tokens at
// NW = 16 -- and the partial sums meet through LDS in a fixed order (bitwise repeatable).
template <bool SELF, int NW, int U>
__global__ __launch_bounds__(64 * NW) void dec_attn_kernel(DecAttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KR = 8 * NW;             // keys per round (eight lanes per 128-byte row)
    float* pr = (float*)smem;              // [L rounded up to KR] scores, then probabilities
    float* red = pr + ((p.L + KR - 1) / KR) * KR;  // [NW waves][64] partial outputs; [0 .. 2 NW) block reductions
    const int j = blockIdx.y, head = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (j >= min(*p.n_rows, p.M)) return;
    const int crop = p.active[j];
    const int sub = tid >> 3, seg = tid & 7;   // key within a round, 16-byte segment of the row
    const half_t* qrow = p.q + (int64_t)j * p.ldq + head * 64 + seg * 8;
    const half8 qv = *(const half8*)qrow;
    float qf[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) qf[e] = (float)qv[e];
    half_t* kb = p.kc + (int64_t)crop * p.crop_stride + head * 64 + seg * 8;
    half_t* vb = p.vc + (int64_t)crop * p.crop_stride + head * 64 + seg * 8;
    const int L = p.L, D = p.D;
    half8 knew = {0, 0, 0, 0, 0, 0, 0, 0}, vnew = knew;
    if (SELF) {   // this step's key / value: into the cache for the later steps, from registers for this one
        knew = *(const half8*)(qrow + D);
        vnew = *(const half8*)(qrow + 2 * D);
        if (sub == 0) {
            *(half8*)(kb + (int64_t)(L - 1) * D) = knew;
            *(half8*)(vb + (int64_t)(L - 1) * D) = vnew;
        }
    }
    // U = rounds in flight per thread.  A pass is ceil(L / (8 NW U)) DEPENDENT batches of loads: with U = 3 the 577 encoder tokens were
    // seven batches per pass (the seventh for one key), fourteen memory round trips per launch however few rows were live; U = 10
    // makes it two per pass (40 registers of loads in flight per thread).  A thread's keys and their order do not depend on U:
    // bit-identical outputs.  The self-attention (L <= 50) is one batch at U = 2.
    // ---- scores
    float mx = -INFINITY;
    for (int k0 = 0; k0 < L; k0 += KR * U) {
        half8 kv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int key = k0 + KR * u + sub;
            kv[u] = knew;
            if (key < L && !(SELF && key == L - 1)) kv[u] = *(const half8*)(kb + (int64_t)key * D);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int key = k0 + KR * u + sub;
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) s += qf[e] * (float)kv[u][e];
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 2);
            s += __shfl_xor(s, 4);
            if (key < L) {
                if (seg == 0) pr[key] = s;
                mx = fmaxf(mx, s);
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0) red[w] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int q = 1; q < NW; ++q) mx = fmaxf(mx, red[q]);
    // ---- softmax numerators and their sum (fixed order: thread strides, wave shuffles, waves 0 .. NW-1)
    float sum = 0.f;
    for (int key = tid; key < L; key += 64 * NW) {
        const float e = expf(pr[key] - mx);
        pr[key] = e;
        sum += e;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane == 0) red[NW + w] = sum;
    __syncthreads();
    sum = red[NW];
#pragma unroll
    for (int q = 1; q < NW; ++q) sum += red[NW + q];
    // ---- weighted values
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < L; k0 += KR * U) {
        half8 vv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int key = k0 + KR * u + sub;
            vv[u] = vnew;
            if (key < L && !(SELF && key == L - 1)) vv[u] = *(const half8*)(vb + (int64_t)key * D);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int key = k0 + KR * u + sub;
            if (key < L) {
                const float pk = pr[key];
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += pk * (float)vv[u][e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {   // the eight key lanes of a wave that share a segment
        acc[e] += __shfl_xor(acc[e], 8);
        acc[e] += __shfl_xor(acc[e], 16);
        acc[e] += __shfl_xor(acc[e], 32);
    }
    __syncthreads();   // the block reductions in red[] have been read by everyone
    if (lane < 8) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[w * 64 + lane * 8 + e] = acc[e];
    }
    __syncthreads();
    if (tid < 8) {
        half8 hv;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = tid * 8 + e;
            float o = red[c];
#pragma unroll
            for (int q = 1; q < NW; ++q) o += red[q * 64 + c];   // waves in order
            hv[e] = (half_t)(o / sum);
        }
        *(half8*)(p.out + (int64_t)j * D + head * 64 + tid * 8) = hv;
    }
}

// ---- greedy step: arg-max over V logits per live row (lowest index wins ties, as torch.argmax), 16-byte loads.
__global__ __launch_bounds__(256) void dec_argmax_kernel(const float* __restrict__ logits, int64_t ld, int V, int32_t* __restrict__ best_tok,
                                                         const int* __restrict__ n_rows, int M) {
    __shared__ float sm[4];
    __shared__ int si[4];
    const int j = blockIdx.x, tid = threadIdx.x;
    if (j >= min(*n_rows, M)) return;
    const float* row = logits + (int64_t)j * ld;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    const int v4 = (!(((uintptr_t)row) & 15)) ? (V >> 2) : 0;
    for (int i = tid; i < v4; i += 256) {
        const floatx4 t = *(const floatx4*)(row + 4 * i);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (t[e] > best) { best = t[e]; bi = 4 * i + e; }   // a thread walks its indices in increasing order: strict > keeps the lowest
    }
    for (int i = 4 * v4 + tid; i < V; i += 256) {
        const float t = row[i];
        if (t > best) { best = t; bi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o);
        const int oi = __shfl_xor(bi, o);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if ((tid & 63) == 0) { sm[tid >> 6] = best; si[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (sm[w] > best || (sm[w] == best && si[w] < bi)) { best = sm[w]; bi = si[w]; }
        best_tok[j] = bi;
    }
}

// ---- GenerationMixin's bookkeeping for the step and the compaction of the row list (one workgroup).
// For list row j (crop c = active[j]): token = forced token | <pad> if the row is finished | the arg-max; ids[c][col] = token;
// </s> finishes the row.  compact != 0: finished rows leave the list (order kept); the new count goes to *n_rows and to the pinned
// host word host_rows[slot] (the host reads it two steps late: an upper bound for its launch geometry, and zero = stop).
struct DecAdvanceParams {
    const int32_t* best_tok;   // [rows]
    int32_t* ids;              // [crops][ld_ids]
    int32_t* done;             // [crops]
    int32_t* active;           // [rows]
    int* n_rows;
    volatile int* host_rows;
    const int32_t* forced;     // [crops][ld_forced] or null
    int ld_ids, col, ld_forced, forced_len, eos, pad, compact;
};

__global__ __launch_bounds__(512) void dec_advance_kernel(DecAdvanceParams p) {
    __shared__ int wsum[8];
    __shared__ int base_s;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int rows = *p.n_rows;
    if (tid == 0) base_s = 0;
    __syncthreads();
    int live_total = 0;
    for (int j0 = 0; j0 < rows; j0 += 512) {   // rows <= 512 is one pass; larger lists walk in chunks (in-place writes never overtake reads)
        const int j = j0 + tid;
        int crop = -1, keep = 0;
        if (j < rows) {
            crop = p.active[j];
            int tok = p.best_tok[j];
            const int was_done = p.done[crop];
            if (p.forced) tok = p.col < p.forced_len ? p.forced[(int64_t)crop * p.ld_forced + p.col] : p.pad;
            else if (was_done) tok = p.pad;
            p.ids[(int64_t)crop * p.ld_ids + p.col] = tok;
            int now_done = was_done;
            if (!p.forced && tok == p.eos) { p.done[crop] = 1; now_done = 1; }
            keep = !now_done;
        }
        // stable compaction: exclusive prefix count of `keep` over the chunk
        const unsigned long long mask = __ballot(keep);
        const int before = __popcll(mask & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[w] = __popcll(mask);
        __syncthreads();
        int off = base_s;
        for (int q = 0; q < w; ++q) off += wsum[q];
        int chunk_total = 0;
        for (int q = 0; q < 8; ++q) chunk_total += wsum[q];
        __syncthreads();   // everyone has read its own active[j] (above) and the counts: in-place writes may start
        if (p.compact && keep) p.active[off + before] = crop;
        if (tid == 0) base_s = off + chunk_total;   // tid 0 is wave 0: off == base_s here
        live_total += chunk_total;
        __syncthreads();
    }
    if (tid == 0) {
        if (p.compact) *p.n_rows = live_total;
        *p.host_rows = live_total;
        __threadfence_system();
    }
}

// ids = [start, pad, pad, ...]; nothing finished; the list holds every row
__global__ void dec_init_kernel(int32_t* ids, int ld_ids, int32_t* done, int32_t* active, int* n_rows, int n, int start, int pad) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b == 0) *n_rows = n;
    if (b >= n) return;
    done[b] = 0;
    active[b] = b;
    for (int c = 0; c < ld_ids; ++c) ids[(int64_t)b * ld_ids + c] = c == 0 ? start : pad;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------- launchers
// flags: 1 = fp16 output, 2 = exact GELU, 4 = split-K partial slabs (fp32, no bias)
int vtd_launch_dec_gemm(const half_t* A, int lda, const half_t* W, const float* bias, void* out, int ldc, int64_t slab_stride, int M,
                        const int* n_rows_dev, int N, int K, int ksplit, int flags, int wide, hipStream_t s) {
    if (M <= 0 || N <= 0 || K <= 0 || ksplit <= 0 || (K % (ksplit * 32)) || !n_rows_dev) return -2501;
    if (ksplit > 1 && !(flags & DG_PARTIAL)) return -2502;
    DecGemmParams p{A, W, bias, out, slab_stride, n_rows_dev, lda, ldc, M, N, K, ksplit, flags};
    static const int tall = [] { const char* e = std::getenv("VTD_DEC_GEMM_TALL"); return e ? atoi(e) : 256; }();
    int lds_from = 768;   // (read per launch: tests flip it inside one process; a getenv is nothing beside a launch)
    if (const char* e = std::getenv("VTD_DEC_GEMM_LDS")) lds_from = atoi(e);
    if (M >= lds_from && !(lda & 7) && !(K & 7)) {   // a tall live list: operands staged through LDS, same arithmetic (dec_gemm_tall_kernel)
        static std::once_flag once;
        static hipError_t attr = hipSuccess;
        std::call_once(once, [] { attr = hipFuncSetAttribute((const void*)dec_gemm_tall_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DT_LDS); });
        if (attr != hipSuccess) return -(int)attr;
        const dim3 grid((N + DT_BN - 1) / DT_BN, (M + DT_BM - 1) / DT_BM, ksplit);
        hipLaunchKernelGGL(dec_gemm_tall_kernel, grid, dim3(1024), DT_LDS, s, p);
    } else if (wide || M >= tall) {   // 64 x 64 tiles: the vocabulary projection (N ~ 50k), where the A tile is re-read by every column tile, and
        // any projection of a tall live list (half the operand re-reads of the 64 x 32 tile; same K order per output, same result)
        const dim3 grid((N + 63) / 64, (M + 63) / 64, ksplit);
        hipLaunchKernelGGL((dec_gemm_kernel<4, 4>), grid, dim3(256), 0, s, p);
    } else if (M <= 16) {   // the tail of a decode (a handful of live rows): 16-row tiles, all of a wave's K range in one batch of loads.
        // Tile height never changes a result: every output's K order (wave quarters, sub-steps in order) is the same in all variants.
        const dim3 grid((N + 31) / 32, 1, ksplit);
        hipLaunchKernelGGL((dec_gemm_kernel<1, 2>), grid, dim3(256), 0, s, p);
    } else if (M <= 32) {
        const dim3 grid((N + 31) / 32, 1, ksplit);
        hipLaunchKernelGGL((dec_gemm_kernel<2, 2>), grid, dim3(256), 0, s, p);
    } else {
        const dim3 grid((N + 31) / 32, (M + 63) / 64, ksplit);
        hipLaunchKernelGGL((dec_gemm_kernel<4, 2>), grid, dim3(256), 0, s, p);
    }
    return -(int)hipGetLastError();
}

int vtd_launch_dec_ln(float* x, half_t* x16, const float* slabs, int64_t slab_stride, int nsplit, const float* bias, const float* gamma,
                      const float* beta, const int* n_rows_dev, int M, int D, float eps, const float* embed, const float* pos, const int32_t* ids,
                      const int32_t* active, int ld_ids, int col, int position, int vocab, hipStream_t s) {
    if (M <= 0 || (D & 63) || D > 2048 || !n_rows_dev) return -2503;
    DecLnParams p{x, x16, slabs, slab_stride, nsplit, bias, gamma, beta, n_rows_dev, M, D, eps, embed, pos, ids, active, ld_ids, col, position, vocab};
    const dim3 grid((M + 3) / 4);
#define VTD_DEC_LN(MAXV4)                                                                              \
    switch (embed ? 0 : nsplit) {                                                                      \
        case 0: hipLaunchKernelGGL((dec_ln_kernel<MAXV4, 0>), grid, dim3(256), 0, s, p); break;        \
        case 1: hipLaunchKernelGGL((dec_ln_kernel<MAXV4, 1>), grid, dim3(256), 0, s, p); break;        \
        case 2: hipLaunchKernelGGL((dec_ln_kernel<MAXV4, 2>), grid, dim3(256), 0, s, p); break;        \
        case 4: hipLaunchKernelGGL((dec_ln_kernel<MAXV4, 4>), grid, dim3(256), 0, s, p); break;        \
        case 8: hipLaunchKernelGGL((dec_ln_kernel<MAXV4, 8>), grid, dim3(256), 0, s, p); break;        \
        default: return -2505;                                                                         \
    }
    if (D <= 1024) { VTD_DEC_LN(4) } else { VTD_DEC_LN(8) }
#undef VTD_DEC_LN
    return -(int)hipGetLastError();
}

int vtd_launch_dec_attn(int self, const half_t* q, int ldq, half_t* kc, half_t* vc, int64_t crop_stride, int L, int D, half_t* out,
                        const int32_t* active, const int* n_rows_dev, int M, int heads, hipStream_t s) {
    if (L <= 0 || L > 16384 || heads * 64 != D || M <= 0) return -2504;
    DecAttnParams p{q, ldq, kc, vc, crop_stride, L, D, out, active, n_rows_dev, M};
    // four waves per (row, head) for both (16 waves for the cross-attention, measured: 95 us against 45 us per launch at 64 live rows --
    // two 1024-thread workgroups per CU leave too few (row, head) units in flight)
    const size_t lds = (size_t)((L + 31) / 32 * 32 + 4 * 64) * 4;
    if (self && L <= 64) hipLaunchKernelGGL((dec_attn_kernel<true, 4, 2>), dim3(heads, M), dim3(256), lds, s, p);
    else if (self) hipLaunchKernelGGL((dec_attn_kernel<true, 4, 4>), dim3(heads, M), dim3(256), lds, s, p);
    else hipLaunchKernelGGL((dec_attn_kernel<false, 4, 10>), dim3(heads, M), dim3(256), lds, s, p);
    return -(int)hipGetLastError();
}

int vtd_launch_dec_argmax(const float* logits, int64_t ld, int V, int32_t* best_tok, const int* n_rows_dev, int M, hipStream_t s) {
    hipLaunchKernelGGL(dec_argmax_kernel, dim3(M), dim3(256), 0, s, logits, ld, V, best_tok, n_rows_dev, M);
    return -(int)hipGetLastError();
}

int vtd_launch_dec_advance(const int32_t* best_tok, int32_t* ids, int ld_ids, int col, int32_t* done, int32_t* active, int* n_rows_dev,
                           int* host_rows, const int32_t* forced, int ld_forced, int forced_len, int eos, int pad, int compact, hipStream_t s) {
    DecAdvanceParams p{best_tok, ids, done, active, n_rows_dev, host_rows, forced, ld_ids, col, ld_forced, forced_len, eos, pad, compact};
    hipLaunchKernelGGL(dec_advance_kernel, dim3(1), dim3(512), 0, s, p);
    return -(int)hipGetLastError();
}

int vtd_launch_dec_init(int32_t* ids, int ld_ids, int32_t* done, int32_t* active, int* n_rows_dev, int n, int start, int pad, hipStream_t s) {
    hipLaunchKernelGGL(dec_init_kernel, dim3((n + 255) / 256), dim3(256), 0, s, ids, ld_ids, done, active, n_rows_dev, n, start, pad);
    return -(int)hipGetLastError();
}
