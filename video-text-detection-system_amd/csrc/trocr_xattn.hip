// Decoder cross-attention of the Transformer recogniser on the RAW encoder states (TrOCR; text_recognizer.py:55-60 -> transformers
// TrOCRAttention with encoder_hidden_states).
//
// The reference projects the encoder states E [T = 577 tokens][C = 768] of a crop to keys and values per decoder layer, K = E Wk^T + bk,
// V = E Wv^T + bv ([T][D = 1024] each), and every decode step reads both for every live row: 2.36 MB per row and layer, the one
// HBM-bound launch of the decode (5.0 TB/s, at its roofline).  Both projections are linear, so they commute with the attention:
//
//     scores_h[t] = q_h . K_h[t]        =  (q_h Wk_h) . E[t]  +  q_h . bk_h          (the last term is the same for every t: softmax drops it)
//     ctx_h       = sum_t P_h[t] V_h[t] =  (sum_t P_h[t] E[t]) Wv_h^T  +  bv_h       (sum_t P_h[t] = 1)
//
// i.e. attend over E itself with a composed query q'_h = q_h Wk_h ([C] per head) and project the attended state afterwards.  The
// two projections fold into the neighbouring dense layers on the host -- W'q = blockdiag(Wk_h)^T Wq ([H C][D]) replaces q_proj, W'o = Wo
// blockdiag(Wv_h) ([D][H C]) replaces out_proj, bias'o = bo + Wo bv -- so the launch count per step does not change, and the kernel below reads
// E once per row and layer: 0.89 MB instead of 2.36 MB.  The encoder pass no longer computes the 24 key / value projections (18 % of its
// GEMM FLOPs) nor writes their 28 MB per crop.
//
// Kernel: one 768-thread workgroup per live row, all H <= 16 heads at once, both products on the matrix pipe.  The token axis is cut into
// three segments (13 + 13 + 11 chunks of 16 tokens at T = 577), one per group of four waves; each group streams its segment through its own
// pair of LDS buffers (LDS-DMA, double buffered) with its own online softmax, and the three partial results are merged at the end in the
// fixed order 0, 1, 2 -- a row's arithmetic does not depend on how many rows the launch has.  Within a group, wave w owns the channels
// [w C/4, (w + 1) C/4):
//   scores  S^T[16 tokens x 16 heads] = E_chunk[16 x C] Q'^T -- the wave's quarter of the K = C contraction (v_mfma_f32_16x16x32_f16, E by
//           rows with ds_read_b128, Q' in registers for the whole row); the four partial tiles meet in LDS and every wave adds them in the
//           same order, so all four hold the same scores and run the same online softmax on their accumulators;
//   ctx'    [16 heads x C/4] += P[16 heads x 16 tokens] E_chunk[16 tokens x C/4] (v_mfma_f32_16x16x16_f16).  S^T leaves the matrix pipe with
//           the head on the lane and the tokens 4 (lane >> 4) + i in the registers, which IS the A-operand layout of that instruction: the
//           probabilities go from the accumulators to fp16 and straight back in.  The B operand wants four tokens of one channel per lane --
//           the same LDS image read with ds_read_b64_tr_b16 (no second, transposed copy).
// The 16-byte chunks of a token row are XOR-ed with f(token) = 0 2 4 .. 14 9 11 13 15 1 3 5 7: within a ds_read_b128 lane group the
// sixteen (token, k-chunk) reads fall on sixteen chunk columns, and within a 32-lane half the eight tokens of the transposed read fall on
// eight chunk pairs -- both kinds of read are conflict-free on rows whose pitch is a multiple of 256 bytes.
// Why twelve waves on one row rather than three rows of four: a chunk is one serial chain (scores -> exchange -> softmax -> products, about
// 1.2 us for a lone group), so a row alone on a CU took 46 us whatever the launch's height; three segments side by side cut that floor to a
// third with the same bytes in flight per CU (3 x 24 KB) and the same LDS (6 x 24 KB) as three independent rows.
#include <cmath>
#include <cstdlib>
#include <mutex>
#include "vtd_common.h"

namespace {

struct XAttnParams {
    const half_t* qp;       // [rows][H * C] composed queries (scaled), row j of the live list
    const half_t* enc;      // [crops][T][C] fp16 final encoder states of the slot
    half_t* cp;             // [rows][H * C] out: softmax-weighted encoder state per head
    const int32_t* active;  // live row -> crop
    const int* n_rows;      // device: exact live row count
    int M, H, T, C;
};

constexpr int XA_TC = 16;   // tokens per chunk
constexpr int XA_SEG = 3;   // token segments = wave groups per row

template <int N>
__device__ __forceinline__ void xa_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ int xa_f(int tok) { return (((tok << 1) ^ (tok & 8)) & 15) | (tok >> 3); }
// physical byte offset of logical 16-byte chunk `lc` of token row `tok` of a chunk (row pitch 2 C bytes)
__device__ __forceinline__ int xa_off(int tok, int lc, int row_bytes) { return tok * row_bytes + ((lc ^ xa_f(tok)) << 4); }

// ds_read_b64_tr_b16 in inline assembly: through the builtin the compiler takes the read for a possible alias of the LDS-DMA in flight and drains
// it (s_waitcnt vmcnt(0)) in front of the read.  The caller waits (xa_wait_lds) before the first use.
__device__ __forceinline__ half4 xa_tr_read(uint32_t lds_addr) {
    half4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(lds_addr));
    return v;
}
template <int N>
__device__ __forceinline__ void xa_wait_lds() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void xa_after_wait(half4& v) { asm volatile("" : "+v"(v)); }   // orders the value's uses behind the wait

// max / sum over the four lanes fr, fr + 16, fr + 32, fr + 48 (result in all four) on the VALU: v_permlane16_swap / v_permlane32_swap
__device__ __forceinline__ float xa_max4(float x) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float xa_sum4(float x) {   // (lane fr + 16 q holds part q): ((p0 + p1) + (p2 + p3)) in every lane
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

template <int KS>   // C = 32 KS
__global__ __launch_bounds__(256 * XA_SEG, 1) void dec_xattn_kernel(const XAttnParams p) {
    constexpr int C = 32 * KS, ROW = 2 * C, CHUNK = XA_TC * ROW, PIECES = CHUNK / 1024;
    constexpr int KW = KS / 4;      // 32-channel k-steps of the scores per wave
    constexpr int NT = C / 64;      // 16-channel tiles of ctx' per wave
    constexpr int PPW = PIECES / 4; // LDS-DMA pieces (loads) per wave and chunk
    static_assert(C % 128 == 0 && PIECES % 4 == 0, "whole swizzle groups, whole LDS-DMA pieces, whole k-steps per wave");
    extern __shared__ __attribute__((aligned(16))) char xsm[];   // [XA_SEG][2][XA_TC][C] fp16 | [XA_SEG][4 waves][64 lanes] partial score tiles
    const int j = blockIdx.x;
    if (j >= min(*p.n_rows, p.M)) return;                      // whole workgroups only: the transposed reads need every lane
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), g = wv >> 2, w = wv & 3;
    const int fr = lane & 15, fq = lane >> 4;
    char* const ebuf = xsm + g * 2 * CHUNK;
    floatx4* const part = (floatx4*)(xsm + XA_SEG * 2 * CHUNK) + g * 256;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(VTD_AS3 char*)ebuf;
    const half_t* const E = p.enc + (int64_t)p.active[j] * p.T * C;
    const int nchunks = (p.T + XA_TC - 1) / XA_TC, per = (nchunks + XA_SEG - 1) / XA_SEG;
    const int c_beg = g * per, nseg = max(0, min(per, nchunks - c_beg));   // this group's chunks: c_beg .. c_beg + nseg - 1

    // One 16-row image (a chunk of E, or the row's Q') -> LDS: wave w brings the 1 KB pieces w, w + 4, ...; the swizzle is on the source address.
    auto issue_rows = [&](char* dst, auto row_ptr) {
#pragma unroll
        for (int k = 0; k < PPW; ++k) {
            const int q = w + 4 * k;
            const int b = q * 1024 + lane * 16;
            const int tok = b / ROW, pc = (b - tok * ROW) >> 4;
            const int lc = pc ^ xa_f(tok);
            __builtin_amdgcn_global_load_lds((const VTD_AS1 void*)(row_ptr(tok) + lc * 8), (VTD_AS3 void*)(dst + q * 1024), 16, 0, 0);
        }
    };
    auto issue = [&](int c) {   // chunk c_beg + c -> buffer c & 1
        issue_rows(ebuf + (c & 1) * CHUNK, [&](int tok) {
            int t = (c_beg + c) * XA_TC + tok;
            t = t < p.T ? t : p.T - 1;                         // rows past the end repeat the last token (their scores are masked)
            return E + (int64_t)t * C;
        });
    };
    // Q' [16 heads][C] travels the same way, through the buffer the second chunk will use: with every global read of the kernel an LDS-DMA the
    // compiler has no register-destination load to wait for inside the loop (it would drain the DMA with vmcnt(0) at the first use).
    {
        const half_t* qrow = p.qp + (int64_t)j * p.H * C;
        issue_rows(ebuf + CHUNK, [&](int head) { return qrow + (int64_t)(head < p.H ? head : p.H - 1) * C; });   // (heads past H: a copy, never stored)
    }
    if (nseg > 0) {
        issue(0);
        xa_wait_vmcnt<PPW>();       // Q' has landed once only the chunk issued after it is outstanding
    } else {
        xa_wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    half8 qf[KW];                   // Q' fragments as the B operand: lane (head fr, channels 32 (w KW + kk) + 8 fq ..)
#pragma unroll
    for (int kk = 0; kk < KW; ++kk) qf[kk] = *(const half8*)(ebuf + CHUNK + xa_off(fr, (w * KW + kk) * 4 + fq, ROW));
    // (the buffer is restaged after the first chunk's barrier B1, which every wave reaches with these reads complete)

    // transposed read of tile nt: lane 4 q + r of group fq supplies token 4 fq + q, channels w C/4 + 16 nt + 4 r ..
    const int tr_tok = 4 * fq + (fr >> 2);
    const int tr_row = tr_tok * ROW + 8 * (fr & 1), tr_lc = w * (C / 32) + ((fr >> 1) & 1), tr_f = xa_f(tr_tok);
    const int sc_row = fr * ROW, sc_lc = w * (C / 32) + fq, sc_f = xa_f(fr);

    floatx4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = floatx4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;   // of head fr, in the exp2 domain (the four lanes of a head hold the same values)

    for (int c = 0; c < per; ++c) {            // every group walks `per` iterations (the barriers are the workgroup's), computing on its own chunks only
        xa_wait_vmcnt<0>();                    // this wave's pieces of chunk c have landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // B1: chunk c is in LDS for the group; the group is done with chunk c - 1 (its buffer, its partials)
        const bool have = c < nseg;
        if (c + 1 < nseg) issue(c + 1);
        const char* eb = ebuf + (c & 1) * CHUNK;
        floatx4 s = {0.f, 0.f, 0.f, 0.f};
        if (have) {
#pragma unroll
            for (int kk = 0; kk < KW; ++kk) {
                const half8 ef = *(const half8*)(eb + sc_row + (((sc_lc + 4 * kk) ^ sc_f) << 4));   // lane (token fr, k-chunk fq)
                s = __builtin_amdgcn_mfma_f32_16x16x32_f16(ef, qf[kk], s, 0, 0, 0);                  // D[token 4 fq + i][head fr]
            }
            part[w * 64 + lane] = s;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // B2: the four partial tiles are visible
        if (!have) continue;                   // (wave-uniform)
        s = part[lane];
        s += part[64 + lane];
        s += part[128 + lane];
        s += part[192 + lane];
        // online softmax of head fr over the chunk's 16 tokens: 4 registers x the four lanes fr, fr + 16, fr + 32, fr + 48
        const int t0 = (c_beg + c) * XA_TC + 4 * fq;
        float cm = -INFINITY;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            s[i] = t0 + i < p.T ? s[i] * 1.44269504088896340736f : -INFINITY;   // exp(x) = exp2(x log2 e)
            cm = fmaxf(cm, s[i]);
        }
        cm = xa_max4(cm);                                  // finite: a chunk's first token always exists
        const float mn = fmaxf(m_run, cm);
        const float alpha = __builtin_amdgcn_exp2f(m_run - mn);   // exp2(-inf) = 0 on the first chunk
        float cs = 0.f;
        half4 pf;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float pe = __builtin_amdgcn_exp2f(s[i] - mn);    // masked tokens: exp2(-inf) = 0
            cs += pe;
            pf[i] = (half_t)pe;
        }
        cs = xa_sum4(cs);
        l_run = l_run * alpha + cs;
        m_run = mn;
        if (__any(alpha != 1.0f)) {            // the running maximum moved for some head: rescale (rare after the first chunks)
            float ar[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) ar[i] = __shfl(alpha, 4 * fq + i);   // accumulator rows are heads 4 fq + i
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[nt][i] *= ar[i];
        }
        {
            constexpr int G = NT / 2;   // two groups: the second group's reads fly under the first group's products
            const uint32_t tb = lds0 + (c & 1) * CHUNK + tr_row;
            half4 tv[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) tv[nt] = xa_tr_read(tb + (((tr_lc + 2 * nt) ^ tr_f) << 4));
            xa_wait_lds<G>();
#pragma unroll
            for (int nt = 0; nt < G; ++nt) xa_after_wait(tv[nt]);
#pragma unroll
            for (int nt = 0; nt < G; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x16f16(pf, tv[nt], acc[nt], 0, 0, 0);   // D[head 4 fq + i][channel fr]
            xa_wait_lds<0>();
#pragma unroll
            for (int nt = G; nt < NT; ++nt) xa_after_wait(tv[nt]);
#pragma unroll
            for (int nt = G; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x16f16(pf, tv[nt], acc[nt], 0, 0, 0);
        }
    }
    // ---- merge the three segments in the order 0, 1, 2 (group 0 does it), normalise, store the row's [H][C] fp16 image in 16-byte pieces
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();              // everyone is done with the last chunk: the buffers are free
    floatx4* const xacc = (floatx4*)xsm;                           // [2 groups][4 waves][NT][64 lanes] accumulators of groups 1, 2 (2 x 2 CHUNK bytes)
    half_t* const obuf = (half_t*)(xsm + 4 * CHUNK);               // [16][C]
    float* const stats = (float*)(xsm + XA_SEG * 2 * CHUNK);       // [XA_SEG][16 heads][m, l]
    if (fq == 0 && w == 0) { stats[(g * 16 + fr) * 2] = m_run; stats[(g * 16 + fr) * 2 + 1] = l_run; }
    if (g > 0) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) xacc[(((g - 1) * 4 + w) * NT + nt) * 64 + lane] = acc[nt];
    }
    __syncthreads();
    if (g == 0) {
        float fa[4], fb[4], fc[4];             // factors of the three partial results for heads 4 fq + i, with 1 / l folded in
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int h = 4 * fq + i;
            float m = stats[h * 2], l = stats[h * 2 + 1], a0 = 1.f, a1 = 0.f, a2 = 0.f;
            {   // + segment 1
                const float m1 = stats[(16 + h) * 2], l1 = stats[(16 + h) * 2 + 1];
                const float mn = fmaxf(m, m1), ea = __builtin_amdgcn_exp2f(m - mn), eb = __builtin_amdgcn_exp2f(m1 - mn);
                a0 *= ea; a1 = eb; l = l * ea + l1 * eb; m = mn;
            }
            {   // + segment 2
                const float m2 = stats[(32 + h) * 2], l2 = stats[(32 + h) * 2 + 1];
                const float mn = fmaxf(m, m2), ea = __builtin_amdgcn_exp2f(m - mn), eb = __builtin_amdgcn_exp2f(m2 - mn);
                a0 *= ea; a1 *= ea; a2 = eb; l = l * ea + l2 * eb;
            }
            const float li = 1.0f / l;
            fa[i] = a0 * li; fb[i] = a1 * li; fc[i] = a2 * li;
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const floatx4 x1 = xacc[((0 * 4 + w) * NT + nt) * 64 + lane], x2 = xacc[((1 * 4 + w) * NT + nt) * 64 + lane];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                obuf[(4 * fq + i) * C + w * (C / 4) + 16 * nt + fr] = (half_t)((acc[nt][i] * fa[i] + x1[i] * fb[i]) + x2[i] * fc[i]);
        }
    }
    __syncthreads();
    uint4* const orow = (uint4*)(p.cp + (int64_t)j * p.H * C);
    for (int idx = tid; idx < p.H * C / 8; idx += 256 * XA_SEG) orow[idx] = ((const uint4*)obuf)[idx];
}

// ---- composed query: qp[r][h][e] = sum_d q[r][h 64 + d] WkT[h][e][d]  (q is scaled already; q . bk is constant over the tokens and drops out
// of the softmax).  One wave per (16 live rows, head, 128 channels): the 16 rows are the MFMA's columns, 16 channels its rows, K = 64.  Row
// 4 g + i of fragment u is channel 32 g + 4 u + i of the slice, so that a lane ends up with 32 consecutive channels of its row (64-byte stores).
struct XqParams {
    const half_t* q;      // [rows][ldq]
    const half_t* wkt;    // [H][C][64] fp16: Wk[h 64 + d][e] stored [h][e][d]
    half_t* qp;           // [rows][H * C]
    const int* n_rows;
    int ldq, M, H, C;
};

__global__ __launch_bounds__(64) void dec_xq_kernel(const XqParams p) {
    const int rows = min(*p.n_rows, p.M), r0 = blockIdx.x * 16, h = blockIdx.y, e0 = blockIdx.z * 128;
    if (r0 >= rows) return;
    const int lane = threadIdx.x, fr = lane & 15, fq = lane >> 4;
    const int r = r0 + fr < rows ? r0 + fr : rows - 1;
    const half_t* qrow = p.q + (int64_t)r * p.ldq + h * 64 + fq * 8;
    const half8 x0 = *(const half8*)qrow, x1 = *(const half8*)(qrow + 32);
    const half_t* wb = p.wkt + ((int64_t)h * p.C + e0 + 32 * (fr >> 2) + (fr & 3)) * 64 + fq * 8;
    half8 y0[8], y1[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        y0[u] = *(const half8*)(wb + u * 4 * 64);
        y1[u] = *(const half8*)(wb + u * 4 * 64 + 32);
    }
    half_t o[32];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        floatx4 acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(y0[u], x0, floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(y1[u], x1, acc, 0, 0, 0);   // D[channel e0 + 32 fq + 4 u + i][row fr]
#pragma unroll
        for (int i = 0; i < 4; ++i) o[4 * u + i] = (half_t)acc[i];
    }
    if (r0 + fr >= rows) return;
    half8* orow = (half8*)(p.qp + (int64_t)(r0 + fr) * p.H * p.C + (int64_t)h * p.C + e0 + 32 * fq);
#pragma unroll
    for (int v = 0; v < 4; ++v) orow[v] = half8{o[8 * v], o[8 * v + 1], o[8 * v + 2], o[8 * v + 3], o[8 * v + 4], o[8 * v + 5], o[8 * v + 6], o[8 * v + 7]};
}

// ---- value projection of the attended state: out[r][h 64 + d] = sum_e cp[r][h][e] Wv[h 64 + d][e] + bv[h 64 + d].  One workgroup per (16 live
// rows, head), eight waves: wave (jd, half) forms the 16 outputs d = 16 jd .. over one half of K = C; the halves meet in LDS.
struct XvParams {
    const half_t* cp;     // [rows][H * C]
    const half_t* wv;     // [D][C] fp16 (the checkpoint's v_proj.weight)
    const float* bv;      // [D]
    half_t* out;          // [rows][D]
    const int* n_rows;
    int M, H, C, D;
};

template <int KH>   // 32-deep K-steps per half: C = 64 KH
__global__ __launch_bounds__(512) void dec_xv_kernel(const XvParams p) {
    __shared__ floatx4 red[4 * 64];
    const int rows = min(*p.n_rows, p.M), r0 = blockIdx.x * 16, h = blockIdx.y;
    if (r0 >= rows) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, jd = wv & 3, kh = wv >> 2, fr = lane & 15, fq = lane >> 4;
    const int r = r0 + fr < rows ? r0 + fr : rows - 1;
    constexpr int C = 64 * KH, U = KH < 12 ? KH : KH % 12 == 0 ? 12 : 8;   // K-steps in flight together (2 x 16 bytes per lane each): the whole half at C = 768
    static_assert(KH % U == 0, "whole batches");
    const half_t* xrow = p.cp + (int64_t)r * p.H * C + (int64_t)h * C + kh * KH * 32 + fq * 8;
    const half_t* wb = p.wv + ((int64_t)h * 64 + jd * 16 + fr) * C + kh * KH * 32 + fq * 8;
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k0 = 0; k0 < KH; k0 += U) {
        half8 x[U], y[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            x[u] = *(const half8*)(xrow + (k0 + u) * 32);
            y[u] = *(const half8*)(wb + (k0 + u) * 32);
        }
        __builtin_amdgcn_sched_barrier(0);   // the batch's loads are all in flight before the first product waits for one
#pragma unroll
        for (int u = 0; u < U; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(y[u], x[u], acc, 0, 0, 0);   // D[d = 16 jd + 4 fq + i][row fr]
    }
    if (kh) red[jd * 64 + lane] = acc;
    __syncthreads();
    if (kh || r0 + fr >= rows) return;
    acc += red[jd * 64 + lane];    // lower half + upper half, always in this order
    const floatx4 b = *(const floatx4*)(p.bv + h * 64 + jd * 16 + 4 * fq);
    *(half4*)(p.out + (int64_t)(r0 + fr) * p.D + h * 64 + jd * 16 + 4 * fq) =
        half4{(half_t)(acc[0] + b[0]), (half_t)(acc[1] + b[1]), (half_t)(acc[2] + b[2]), (half_t)(acc[3] + b[3])};
}

}  // namespace

// (an encoder wider than 768 channels -- the `large` checkpoints -- keeps the key / value form: three groups' buffers would not fit one CU)
bool vtd_dec_xattn_supported(int H, int T, int C) { return H >= 1 && H <= 16 && T >= 1 && (C == 768 || C == 128 || C == 256 || C == 512); }

namespace {
template <int KS>
int xa_launch(const XAttnParams& p, hipStream_t s) {
    constexpr size_t lds = (size_t)XA_SEG * 2 * XA_TC * 2 * (32 * KS) + XA_SEG * 4 * 64 * sizeof(floatx4);
    static_assert(lds <= 160 * 1024, "the three groups' buffers fit the CU");
    static std::once_flag once;
    static hipError_t attr = hipSuccess;
    std::call_once(once, [] { attr = hipFuncSetAttribute((const void*)dec_xattn_kernel<KS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
    if (attr != hipSuccess) return -(int)attr;
    hipLaunchKernelGGL((dec_xattn_kernel<KS>), dim3(p.M), dim3(256 * XA_SEG), lds, s, p);
    return -(int)hipGetLastError();
}
}  // namespace

// M is the host's (lagged) upper bound of the live rows: it sizes the grid only.
int vtd_launch_dec_xattn(const half_t* qp, const half_t* enc, half_t* cp, const int32_t* active, const int* n_rows_dev, int M, int H, int T, int C,
                         hipStream_t s) {
    if (!qp || !enc || !cp || !active || !n_rows_dev || M <= 0 || !vtd_dec_xattn_supported(H, T, C)) return -2801;
    XAttnParams p{qp, enc, cp, active, n_rows_dev, M, H, T, C};
    switch (C) {
        case 768: return xa_launch<24>(p, s);
        case 128: return xa_launch<4>(p, s);
        case 256: return xa_launch<8>(p, s);
        default: return xa_launch<16>(p, s);
    }
}

int vtd_launch_dec_xq(const half_t* q, int ldq, const half_t* wkt, half_t* qp, const int* n_rows_dev, int M, int H, int C, hipStream_t s) {
    if (!q || !wkt || !qp || !n_rows_dev || M <= 0 || H <= 0 || (C & 127)) return -2802;
    XqParams p{q, wkt, qp, n_rows_dev, ldq, M, H, C};
    hipLaunchKernelGGL(dec_xq_kernel, dim3((M + 15) / 16, H, C / 128), dim3(64), 0, s, p);
    return -(int)hipGetLastError();
}

int vtd_launch_dec_xv(const half_t* cp, const half_t* wv, const float* bv, half_t* out, const int* n_rows_dev, int M, int H, int C, int D,
                      hipStream_t s) {
    if (!cp || !wv || !bv || !out || !n_rows_dev || M <= 0 || H * 64 != D || !vtd_dec_xattn_supported(H, 1, C)) return -2803;
    XvParams p{cp, wv, bv, out, n_rows_dev, M, H, C, D};
    const dim3 grid((M + 15) / 16, H);
    switch (C) {
        case 768: hipLaunchKernelGGL(dec_xv_kernel<12>, grid, dim3(512), 0, s, p); break;
        case 128: hipLaunchKernelGGL(dec_xv_kernel<2>, grid, dim3(512), 0, s, p); break;
        case 256: hipLaunchKernelGGL(dec_xv_kernel<4>, grid, dim3(512), 0, s, p); break;
        default: hipLaunchKernelGGL(dec_xv_kernel<8>, grid, dim3(512), 0, s, p); break;
    }
    return -(int)hipGetLastError();
}
