// Encoder self-attention of the Transformer recogniser (ViT: 577 tokens, heads of 64; reference text_recognizer.py:55-60 ->
// transformers ViTSelfAttention: softmax(Q K^T / 8) V), flash style on v_mfma_f32_16x16x32_f16.
//
//   S^T = K Q^T    so a lane owns one query column: the online softmax runs on the accumulator registers (two shfl_xor per reduction)
//                  and the probabilities feed the second MFMA directly as its K-permuted operand -- no LDS round trip for P;
//   O^T = V^T P^T  needs V with the KEYS contiguous in a lane's operand.  The first version transposed V while staging it (eight 2-byte LDS
//                  writes per thread and key block: 1.3 ms per layer at 272 crops); the second had a pre-pass write V^T per (crop, head) to
//                  HBM once per layer (105 us per layer, 1 MB per crop) and read V^T tiles.  Now V is staged exactly as K is -- [64 keys][64]
//                  rows straight out of the qkv tensor -- and the operand is formed by gfx950's transposed LDS read: ds_read_b64_tr_b16 hands a
//                  lane four KEYS of one channel, two of them make the 8-deep fragment in the key order the probabilities come out in.  No
//                  pre-pass, no V^T buffer, the same products in the same order (bit-identical results).
//   tile           128 queries per workgroup (4 waves x 2 query fragments) against key blocks of 64: the K / V^T tile traffic and the
//                  barriers per query are a quarter of the first version's; the next block's tiles are fetched into registers while the
//                  current block multiplies (one barrier pair per 64 keys).
#include <cmath>
#include "vtd_common.h"

namespace {

constexpr int AT_KB = 64;      // keys per block
constexpr int AT_PADK = 80;    // halfs per LDS row of the K tile.  160 bytes: its ds_read_b128 fragment reads (row = fr, chunk = fq + 4 kk) hit 16
                               // different bank quads in each of the hardware's lane groups ({0-3, 12-15, 20-27}, ...).  The first version used
                               // 72 (144 bytes), conflict-free only for groups of consecutive lanes: SQ_LDS_BANK_CONFLICT was 40 % of the kernel's
                               // LDS cycles (tools/gpu_pmc_trocr.sh).  The staging stores stay conflict-free at this pitch because odd rows
                               // write the second 16 bytes of their 32-byte piece first (`sw` below).
                               // The V tile has the same pitch: its transposed reads (32-lane halves: 8 keys x 32 bytes) then fall on eight
                               // different 32-byte bank groups (r x 160 mod 256 = 0, 160, 64, 224, 128, 32, 192, 96).
typedef __fp16 at_fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

__global__ __launch_bounds__(256) void trocr_attention_kernel(const half_t* __restrict__ qkv, half_t* __restrict__ out, int T, int C, float scale,
                                                              int heads, int units, int qtiles, int64_t prows) {
    __shared__ __attribute__((aligned(16))) half_t ks[2][AT_KB * AT_PADK];   // K tile, key-major
    __shared__ __attribute__((aligned(16))) half_t vs[2][AT_KB * AT_PADK];   // V tile, key-major as well (read transposed)
    // Workgroup -> (crop, head, query tile).  The query tiles of one (crop, head) read the same K and V rows (148 KB): they must run on
    // ONE XCD to find them in its L2.  Consecutive workgroup ids go to different XCDs, so XCD x = id % 8 takes the units x, x + 8, ...
    // and walks each unit's `qtiles` tiles back to back (PMC before: 3.0 GB read per launch at 287 crops, every tile fetching K / V
    // from HBM; algorithmic 1.0 GB).
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int unit = (idx / qtiles) * 8 + xcd, qt = idx - (idx / qtiles) * qtiles;
    if (unit >= units) return;   // padding of the last group of eight units (uniform over the workgroup, before any barrier)
    const int b = unit / heads, head = unit - b * heads, q0 = qt * 128;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, fr = lane & 15, fq = lane >> 4;
    const int64_t ld = 3 * (int64_t)C;
    const half_t* base = qkv + (int64_t)b * T * ld + head * 64;
    // Q fragments (second MFMA operand: column = query fr, K chunk fq), pre-scaled; wave w owns queries q0 + 32 w + 16 g + fr
    half8 qf[2][2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int q = q0 + w * 32 + g * 16 + fr;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            half8 t = {0, 0, 0, 0, 0, 0, 0, 0};
            if (q < T) t = *(const half8*)(base + (int64_t)q * ld + kk * 32 + fq * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = (half_t)((float)t[e] * scale);
            qf[g][kk] = t;
        }
    }
    floatx4 acc_o[2][4];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc_o[g][i] = floatx4{0.f, 0.f, 0.f, 0.f};
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
    // staging: thread -> (row 0..63, 32-byte half of... ) two 16-byte pieces per tile and thread
    const int srow = tid >> 2, scol = (tid & 3) * 16;   // row of the tile, first of 16 halfs
    const int sw = (srow & 1) * 8;                        // K tile: odd rows handle the second half of their piece first (store bank conflicts)
    auto fetch = [&](int k0, half8* kreg, half8* vreg) {
        int key = k0 + srow;
        key = key < T ? key : T - 1;   // rows past the end: any valid row (their scores are masked)
        const half_t* ksrc = base + (int64_t)key * ld + C + scol;
        kreg[0] = *(const half8*)(ksrc + sw);
        kreg[1] = *(const half8*)(ksrc + (8 - sw));
        const half_t* vsrc = ksrc + C;   // the same key's value row (rows past the end: their probabilities are exactly 0, the row is finite)
        vreg[0] = *(const half8*)(vsrc + sw);
        vreg[1] = *(const half8*)(vsrc + (8 - sw));
    };
    auto stage = [&](int buf, const half8* kreg, const half8* vreg) {
        *(half8*)(&ks[buf][srow * AT_PADK + scol + sw]) = kreg[0];
        *(half8*)(&ks[buf][srow * AT_PADK + scol + (8 - sw)]) = kreg[1];
        *(half8*)(&vs[buf][srow * AT_PADK + scol + sw]) = vreg[0];
        *(half8*)(&vs[buf][srow * AT_PADK + scol + (8 - sw)]) = vreg[1];
    };
    half8 kreg[2], vreg[2];
    fetch(0, kreg, vreg);
    stage(0, kreg, vreg);
    __syncthreads();
    const int nblk = (T + AT_KB - 1) / AT_KB;
    for (int blk = 0; blk < nblk; ++blk) {
        const int k0 = blk * AT_KB, buf = blk & 1;
        if (blk + 1 < nblk) fetch(k0 + AT_KB, kreg, vreg);   // in flight under this block's maths
        // S^T[key][q] for the block: 4 key fragments x 2 query fragments, K dim 64 = 2 x 32
        floatx4 s[2][4];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int kf = 0; kf < 4; ++kf) s[g][kf] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int kf = 0; kf < 4; ++kf) {
                const half8 ka = *(const half8*)(&ks[buf][(kf * 16 + fr) * AT_PADK + kk * 32 + fq * 8]);
#pragma unroll
                for (int g = 0; g < 2; ++g) s[g][kf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ka, qf[g][kk], s[g][kf], 0, 0, 0);
            }
        // online softmax: lane holds keys k0 + 16 kf + 4 fq + e of query fr (per query fragment g).  The softmax pass is what bounds
        // this kernel (32 scores per lane and block against 32 MFMAs per wave): exponentials run as ONE v_exp_f32 each on
        // fma(s, log2 e, -m log2 e) instead of libdevice's expf (~10 instructions), and only the last key block pays for the
        // out-of-range mask.
        constexpr float LOG2E = 1.4426950408889634f;
        const bool partial = k0 + AT_KB > T;
        half8 pf[2][2];   // [g][key half]: probabilities in the key order (kf = 2 h: 4 fq + e, kf = 2 h + 1: 16 + 4 fq + e) of the PV operand
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            if (partial) {
#pragma unroll
                for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (k0 + kf * 16 + fq * 4 + e >= T) s[g][kf][e] = -INFINITY;
            }
            float bm = -INFINITY;
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                for (int e = 0; e < 4; ++e) bm = fmaxf(bm, s[g][kf][e]);
            bm = fmaxf(bm, __shfl_xor(bm, 16));
            bm = fmaxf(bm, __shfl_xor(bm, 32));
            const float m_new = fmaxf(m_run[g], bm);          // finite from the first block on (every block holds a real key)
            const float mneg = -m_new * LOG2E;
            const float corr = __builtin_amdgcn_exp2f((m_run[g] - m_new) * LOG2E);   // first block: exp2(-inf) = 0
            float ps = 0.f;
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pv = __builtin_amdgcn_exp2f(fmaf(s[g][kf][e], LOG2E, mneg));
                    ps += pv;
                    pf[g][kf >> 1][(kf & 1) * 4 + e] = (half_t)pv;
                }
            l_run[g] = l_run[g] * corr + ps;
            m_run[g] = m_new;
#pragma unroll
            for (int i = 0; i < 4; ++i) acc_o[g][i] *= corr;
        }
        // O^T += V^T P^T: d fragments i, key halves h (32 keys each, in the permuted order of pf)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // lane 4 q + r of a 16-lane group hands in the address of key row 32 h + 4 fq + q (+ 16), channels 16 i + 4 r ..; lane fr gets
                // channel 16 i + fr of those four keys
                const half_t* vrow = &vs[buf][(h * 32 + fq * 4 + (fr >> 2)) * AT_PADK + i * 16 + (fr & 3) * 4];
                const half4 va = __builtin_bit_cast(half4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((VTD_AS3 at_fp16x4*)vrow));
                const half4 vb = __builtin_bit_cast(half4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((VTD_AS3 at_fp16x4*)(vrow + 16 * AT_PADK)));
                const half8 vf = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
#pragma unroll
                for (int g = 0; g < 2; ++g) acc_o[g][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[g][h], acc_o[g][i], 0, 0, 0);
            }
        if (blk + 1 < nblk) {
            stage(buf ^ 1, kreg, vreg);   // the other buffer: last read in block blk - 1, behind the barrier below of that iteration
            __syncthreads();
        }
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        float l = l_run[g];
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        const int q = q0 + w * 32 + g * 16 + fr;
        if (q < T) {
            const float inv = 1.0f / l;
            const int64_t row = (int64_t)b * T + q;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                half4 hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) hv[e] = (half_t)(acc_o[g][i][e] * inv);
                const int ch = head * 64 + i * 16 + fq * 4;   // (prows: K-panel-major for the output projection's GEMM, [C / 32][prows][32])
                *(half4*)(out + (prows ? (((int64_t)(ch >> 5) * prows + row) << 5) + (ch & 31) : row * C + ch)) = hv;
            }
        }
    }
}

}  // namespace

int vtd_launch_trocr_attention(const half_t* qkv, half_t* out, int n, int T, int C, int heads, int64_t out_prows, hipStream_t s) {
    if (heads * 64 != C || n <= 0 || T <= 0) return -2404;
    const int qtiles = (T + 127) / 128, units = heads * n;
    const int64_t grid = (int64_t)((units + 7) / 8) * 8 * qtiles;
    hipLaunchKernelGGL(trocr_attention_kernel, dim3((unsigned)grid), dim3(256), 0, s, qkv, out, T, C, 0.125f, heads, units, qtiles, out_prows);
    return -(int)hipGetLastError();
}
