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
// Kernel: one 256-thread workgroup per live row, all H <= 16 heads at once.  E streams through LDS in chunks of 16 tokens (LDS-DMA,
// double buffered, 16-byte chunks XOR-swizzled by the token so that both the MFMA's row reads and the channel-parallel reads below are
// conflict-free).  Wave 3 forms the chunk's scores S[16 heads x 16 tokens] = Q'[16 x C] E_chunk^T on the matrix pipe (Q' in registers for
// the whole row) and runs the online softmax on the accumulators; waves 0-2 own four channels per lane and accumulate
// ctx'[h][c] += P[h][t] E[t][c] in fp32 for the 16 heads from the same LDS image (no transposed operand: the token axis is walked, not
// contracted in a matrix instruction).  Three workgroups per CU, so one's score phase runs under the others' accumulation.
#include <cmath>
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

template <int N>
__device__ __forceinline__ void xa_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// physical byte offset of logical 16-byte chunk `lc` of token row `tok` (row pitch 2 C bytes, chunks XOR-ed by the token inside groups of 16)
__device__ __forceinline__ int xa_off(int tok, int lc, int row_bytes) { return tok * row_bytes + (((lc & ~15) | ((lc & 15) ^ (tok & 15))) << 4); }

template <int KS>   // C = 32 KS
__global__ __launch_bounds__(256, 3) void dec_xattn_kernel(const XAttnParams p) {   // three workgroups per CU (<= 168 VGPRs)
    constexpr int C = 32 * KS, ROW = 2 * C, CHUNK = XA_TC * ROW, PIECES = CHUNK / 1024;
    static_assert(C % 128 == 0 && CHUNK % 1024 == 0, "whole swizzle groups, whole LDS-DMA pieces");
    extern __shared__ __attribute__((aligned(16))) char xsm[];
    char* const ebuf = xsm;                                  // [2][XA_TC][C] fp16
    float* const pbuf = (float*)(xsm + 2 * CHUNK);           // [XA_TC][16] probabilities of the chunk (un-normalised)
    float* const abuf = pbuf + XA_TC * 16;                   // [16] rescale factors of the chunk, then 1 / l at the end
    const int j = blockIdx.x;
    if (j >= min(*p.n_rows, p.M)) return;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const half_t* const E = p.enc + (int64_t)p.active[j] * p.T * C;
    const int nchunks = (p.T + XA_TC - 1) / XA_TC;

    auto issue = [&](int c) {   // chunk c -> buffer c & 1: wave w brings pieces w, w + 4, ...
        char* dst = ebuf + (c & 1) * CHUNK;
#pragma unroll
        for (int k = 0; k < (PIECES + 3) / 4; ++k) {
            const int q = w + 4 * k;
            if (q >= PIECES) break;
            const int b = q * 1024 + lane * 16;
            const int tok = b / ROW, pc = (b - tok * ROW) >> 4;
            const int lc = (pc & ~15) | ((pc & 15) ^ (tok & 15));
            int t = c * XA_TC + tok;
            t = t < p.T ? t : p.T - 1;                       // rows past the end repeat the last token (their scores are masked)
            __builtin_amdgcn_global_load_lds((const VTD_AS1 void*)(E + (int64_t)t * C + lc * 8), (VTD_AS3 void*)(dst + q * 1024), 16, 0, 0);
        }
    };

    // Two roles, two loops (so that the score wave's 96 registers of Q' and the accumulating waves' 64 accumulators never share a live
    // range): both run the same barrier sequence -- B1 at the top of a chunk, B2 in its middle, two more at the end.
    const auto chunk_top = [&](int c) {
        xa_wait_vmcnt<0>();                    // this wave's pieces of chunk c have landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // B1: chunk c is in LDS for everyone; everyone is done with chunk c - 1 (its buffer, pbuf, abuf)
        if (c + 1 < nchunks) issue(c + 1);
    };
    const auto lds_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    issue(0);
    if (w == 3) {
        // ---- score wave: S[16 heads x 16 tokens] = Q' E_chunk^T on the matrix pipe, online softmax on the accumulators
        half8 qf[KS];                 // Q' fragments, lane (head fr, k-chunk fq); heads past H are zero rows
        const half_t* qrow = p.qp + (int64_t)j * p.H * C + (int64_t)fr * C + fq * 8;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            qf[kk] = half8{0, 0, 0, 0, 0, 0, 0, 0};
            if (fr < p.H) qf[kk] = *(const half8*)(qrow + kk * 32);
        }
        float m_run[4], l_run[4];     // running max / sum of heads 4 fq + e (replicated over the 16 lanes of a row)
#pragma unroll
        for (int e = 0; e < 4; ++e) { m_run[e] = -INFINITY; l_run[e] = 0.f; }
        for (int c = 0; c < nchunks; ++c) {
            chunk_top(c);
            const char* eb = ebuf + (c & 1) * CHUNK;
            floatx4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                const half8 ef = *(const half8*)(eb + xa_off(fr, kk * 4 + fq, ROW));   // lane (token fr, k-chunk fq)
                s = __builtin_amdgcn_mfma_f32_16x16x32_f16(qf[kk], ef, s, 0, 0, 0);    // D[head 4 fq + e][token fr]
                if ((kk & 3) == 3) __builtin_amdgcn_sched_barrier(0);                  // four reads in flight at a time: Q' already holds 4 KS registers
            }
            const bool valid = c * XA_TC + fr < p.T;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float sv = valid ? s[e] : -INFINITY;
                float cm = sv;
                cm = fmaxf(cm, __shfl_xor(cm, 1));
                cm = fmaxf(cm, __shfl_xor(cm, 2));
                cm = fmaxf(cm, __shfl_xor(cm, 4));
                cm = fmaxf(cm, __shfl_xor(cm, 8));
                const float mn = fmaxf(m_run[e], cm);           // finite: a chunk's first token always exists
                const float alpha = expf(m_run[e] - mn);        // exp(-inf) = 0 on the first chunk
                const float pe = valid ? expf(sv - mn) : 0.f;
                float cs = pe;
                cs += __shfl_xor(cs, 1);
                cs += __shfl_xor(cs, 2);
                cs += __shfl_xor(cs, 4);
                cs += __shfl_xor(cs, 8);
                l_run[e] = l_run[e] * alpha + cs;
                m_run[e] = mn;
                pbuf[fr * 16 + 4 * fq + e] = pe;
                if (fr == 0) abuf[4 * fq + e] = alpha;
            }
            lds_barrier();                     // B2: the chunk's probabilities and rescale factors are visible
        }
        lds_barrier();                         // everyone is done reading abuf
        if (fr == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) abuf[4 * fq + e] = 1.0f / l_run[e];
        }
        lds_barrier();
        return;
    }
    // ---- accumulating waves: ctx'[h][c] += P[h][t] E[t][c] for 16 heads x my four channels, fp32
    const int ch0 = tid * 4;
    const bool pv_lane = ch0 < C;
    const int lc = (pv_lane ? ch0 : 0) >> 3, sub = (ch0 & 4) * 2;
    float acc[16][4];
#pragma unroll
    for (int h = 0; h < 16; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[h][i] = 0.f;
    for (int c = 0; c < nchunks; ++c) {
        chunk_top(c);
        const char* eb = ebuf + (c & 1) * CHUNK;
        lds_barrier();                         // B2
        if (!pv_lane) continue;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const floatx4 a = *(const floatx4*)(abuf + 4 * q4);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[4 * q4 + e][i] *= a[e];
        }
#pragma unroll 2
        for (int t = 0; t < XA_TC; ++t) {
            const half4 ev = *(const half4*)(eb + xa_off(t, lc, ROW) + sub);
            const float e0 = (float)ev[0], e1 = (float)ev[1], e2 = (float)ev[2], e3 = (float)ev[3];
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const floatx4 v = *(const floatx4*)(pbuf + t * 16 + 4 * q4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[4 * q4 + e][0] += v[e] * e0;
                    acc[4 * q4 + e][1] += v[e] * e1;
                    acc[4 * q4 + e][2] += v[e] * e2;
                    acc[4 * q4 + e][3] += v[e] * e3;
                }
            }
        }
    }
    lds_barrier();
    lds_barrier();                             // 1 / l of every head is in abuf
    if (pv_lane) {
        half_t* orow = p.cp + (int64_t)j * p.H * C + ch0;
#pragma unroll
        for (int h = 0; h < 16; ++h) {
            const float li = abuf[h];
            half4 hv;
            hv[0] = (half_t)(acc[h][0] * li); hv[1] = (half_t)(acc[h][1] * li); hv[2] = (half_t)(acc[h][2] * li); hv[3] = (half_t)(acc[h][3] * li);
            if (h < p.H) *(half4*)(orow + (int64_t)h * C) = hv;
        }
    }
}

// ---- composed query: qp[r][h][e] = sum_d q[r][h 64 + d] WkT[h][e][d]  (q is scaled already; q . bk is constant over the tokens and drops out
// of the softmax).  One wave per (16 live rows, head): the 16 rows are the MFMA's columns, a 16-channel slice of Wk_h^T its rows, K = 64.
struct XqParams {
    const half_t* q;      // [rows][ldq]
    const half_t* wkt;    // [H][C][64] fp16: Wk[h 64 + d][e] stored [h][e][d]
    half_t* qp;           // [rows][H * C]
    const int* n_rows;
    int ldq, M, H, C;
};

__global__ __launch_bounds__(64) void dec_xq_kernel(const XqParams p) {
    const int rows = min(*p.n_rows, p.M), r0 = blockIdx.x * 16, h = blockIdx.y;
    if (r0 >= rows) return;
    const int lane = threadIdx.x, fr = lane & 15, fq = lane >> 4;
    const int r = r0 + fr < rows ? r0 + fr : rows - 1;
    const half_t* qrow = p.q + (int64_t)r * p.ldq + h * 64 + fq * 8;
    const half8 x0 = *(const half8*)qrow, x1 = *(const half8*)(qrow + 32);
    const half_t* wb = p.wkt + ((int64_t)h * p.C + fr) * 64 + fq * 8;
    half_t* orow = p.qp + (int64_t)(r0 + fr) * p.H * p.C + (int64_t)h * p.C + 4 * fq;
    constexpr int U = 8;   // channel fragments whose weights are in flight together
    for (int e0 = 0; e0 < p.C; e0 += 16 * U) {
        half8 y0[U], y1[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + 16 * u < p.C ? e0 + 16 * u : e0;
            y0[u] = *(const half8*)(wb + (int64_t)e * 64);
            y1[u] = *(const half8*)(wb + (int64_t)e * 64 + 32);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (e0 + 16 * u >= p.C) break;
            floatx4 acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(y0[u], x0, floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(y1[u], x1, acc, 0, 0, 0);   // D[channel e0 + 16 u + 4 fq + i][row fr]
            if (r0 + fr < rows) *(half4*)(orow + e0 + 16 * u) = half4{(half_t)acc[0], (half_t)acc[1], (half_t)acc[2], (half_t)acc[3]};
        }
    }
}

// ---- value projection of the attended state: out[r][h 64 + d] = sum_e cp[r][h][e] Wv[h 64 + d][e] + bv[h 64 + d]
struct XvParams {
    const half_t* cp;     // [rows][H * C]
    const half_t* wv;     // [D][C] fp16 (the checkpoint's v_proj.weight)
    const float* bv;      // [D]
    half_t* out;          // [rows][D]
    const int* n_rows;
    int M, H, C, D;
};

__global__ __launch_bounds__(64) void dec_xv_kernel(const XvParams p) {
    const int rows = min(*p.n_rows, p.M), r0 = blockIdx.x * 16, h = blockIdx.y;
    if (r0 >= rows) return;
    const int lane = threadIdx.x, fr = lane & 15, fq = lane >> 4;
    const int r = r0 + fr < rows ? r0 + fr : rows - 1;
    const half_t* xrow = p.cp + (int64_t)r * p.H * p.C + (int64_t)h * p.C + fq * 8;
    const half_t* wb = p.wv + ((int64_t)h * 64 + fr) * p.C + fq * 8;
    floatx4 acc[4];
#pragma unroll
    for (int jd = 0; jd < 4; ++jd) acc[jd] = floatx4{0.f, 0.f, 0.f, 0.f};
    constexpr int U = 4;   // 32-deep K-steps in flight together (5 x 16 bytes per lane each)
    const int ks = p.C >> 5;
    for (int k0 = 0; k0 < ks; k0 += U) {
        half8 x[U], y[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kk = k0 + u < ks ? k0 + u : k0;
            x[u] = *(const half8*)(xrow + kk * 32);
#pragma unroll
            for (int jd = 0; jd < 4; ++jd) y[u][jd] = *(const half8*)(wb + (int64_t)jd * 16 * p.C + kk * 32);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (k0 + u >= ks) break;
#pragma unroll
            for (int jd = 0; jd < 4; ++jd) acc[jd] = __builtin_amdgcn_mfma_f32_16x16x32_f16(y[u][jd], x[u], acc[jd], 0, 0, 0);   // D[d = 16 jd + 4 fq + i][row fr]
        }
    }
    if (r0 + fr >= rows) return;
    half_t* orow = p.out + (int64_t)(r0 + fr) * p.D + h * 64 + 4 * fq;
#pragma unroll
    for (int jd = 0; jd < 4; ++jd) {
        const floatx4 b = *(const floatx4*)(p.bv + h * 64 + jd * 16 + 4 * fq);
        *(half4*)(orow + jd * 16) = half4{(half_t)(acc[jd][0] + b[0]), (half_t)(acc[jd][1] + b[1]), (half_t)(acc[jd][2] + b[2]), (half_t)(acc[jd][3] + b[3])};
    }
}

}  // namespace

bool vtd_dec_xattn_supported(int H, int T, int C) { return H >= 1 && H <= 16 && T >= 1 && (C == 768 || C == 128 || C == 256 || C == 512 || C == 1024); }

int vtd_launch_dec_xattn(const half_t* qp, const half_t* enc, half_t* cp, const int32_t* active, const int* n_rows_dev, int M, int H, int T, int C,
                         hipStream_t s) {
    if (!qp || !enc || !cp || !active || !n_rows_dev || M <= 0 || !vtd_dec_xattn_supported(H, T, C)) return -2801;
    XAttnParams p{qp, enc, cp, active, n_rows_dev, M, H, T, C};
    const size_t lds = (size_t)2 * XA_TC * 2 * C + (XA_TC * 16 + 16) * sizeof(float);
    switch (C) {
        case 768: hipLaunchKernelGGL((dec_xattn_kernel<24>), dim3(M), dim3(256), lds, s, p); break;
        case 128: hipLaunchKernelGGL((dec_xattn_kernel<4>), dim3(M), dim3(256), lds, s, p); break;
        case 256: hipLaunchKernelGGL((dec_xattn_kernel<8>), dim3(M), dim3(256), lds, s, p); break;
        case 512: hipLaunchKernelGGL((dec_xattn_kernel<16>), dim3(M), dim3(256), lds, s, p); break;
        default: hipLaunchKernelGGL((dec_xattn_kernel<32>), dim3(M), dim3(256), lds, s, p); break;
    }
    return -(int)hipGetLastError();
}

int vtd_launch_dec_xq(const half_t* q, int ldq, const half_t* wkt, half_t* qp, const int* n_rows_dev, int M, int H, int C, hipStream_t s) {
    if (!q || !wkt || !qp || !n_rows_dev || M <= 0 || H <= 0 || (C & 15)) return -2802;
    XqParams p{q, wkt, qp, n_rows_dev, ldq, M, H, C};
    hipLaunchKernelGGL(dec_xq_kernel, dim3((M + 15) / 16, H), dim3(64), 0, s, p);
    return -(int)hipGetLastError();
}

int vtd_launch_dec_xv(const half_t* cp, const half_t* wv, const float* bv, half_t* out, const int* n_rows_dev, int M, int H, int C, int D,
                      hipStream_t s) {
    if (!cp || !wv || !bv || !out || !n_rows_dev || M <= 0 || H * 64 != D || (C & 31)) return -2803;
    XvParams p{cp, wv, bv, out, n_rows_dev, M, H, C, D};
    hipLaunchKernelGGL(dec_xv_kernel, dim3((M + 15) / 16, H), dim3(64), 0, s, p);
    return -(int)hipGetLastError();
}
