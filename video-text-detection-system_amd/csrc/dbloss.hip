// DB training loss, forward (first slice): the scalar the reference's training / validation step computes,
//     total = BCELoss(probability, probability_map) + BCELoss(threshold, threshold_map) + DiceLoss(probability, probability_map)
// (app/ml/training/trainer.py:48-56 training_step, :66-71 validation_step; DiceLoss :130-142; nn.BCELoss = mean over all elements of
// -(t * max(log p, -100) + (1 - t) * max(log1p(-p), -100)), torch's clamp), as ONE pass over the four fp32 maps.
//
// HBM-bound: every element of the two outputs and the two targets is read exactly once (16 bytes per pixel position, 16-byte loads:
// four positions per lane and load), nothing is written but 5 partial sums per workgroup.  The five sums (both BCE numerators, sum p t,
// sum p, sum t) are formed per element in fp32 exactly as torch forms them and ACCUMULATED in fp64 in a fixed order: lane-strided
// partials, a fixed shuffle tree per wave, waves in index order per workgroup, workgroups in a fixed order in the finishing kernel -- no
// atomics: the same bits on every run (the grid is a function of the element count only).
#include "vtd_common.h"

namespace {

constexpr int DBL_THREADS = 256;
constexpr int DBL_MAX_BLOCKS = 2048;

struct DbLossParams {
    const float* prob;      // [n] network probability map
    const float* thresh;    // [n] network threshold map (may be null: its BCE term is 0)
    const float* prob_t;    // [n] targets['probability_map']
    const float* thresh_t;  // [n] targets['threshold_map'] (null iff thresh is)
    double* partial;        // [blocks][5]
    int64_t n;
};

__device__ __forceinline__ float bce_term(float p, float t) {
    // torch/aten binary_cross_entropy (CPU and CUDA kernels alike): (t - 1) * max(log1p(-p), -100) - t * max(log(p), -100)
    const float l1 = fmaxf(log1pf(-p), -100.0f), l0 = fmaxf(logf(p), -100.0f);
    return (t - 1.0f) * l1 - t * l0;
}

__global__ __launch_bounds__(DBL_THREADS) void dbloss_partial_kernel(const DbLossParams p) {
    double s_bp = 0.0, s_bt = 0.0, s_pt = 0.0, s_p = 0.0, s_t = 0.0;
    const int64_t n4 = p.n >> 2;
    const int64_t stride = (int64_t)gridDim.x * DBL_THREADS;
    for (int64_t i = (int64_t)blockIdx.x * DBL_THREADS + threadIdx.x; i < n4; i += stride) {
        const floatx4 a = *(const floatx4*)(p.prob + 4 * i), b = *(const floatx4*)(p.prob_t + 4 * i);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            s_bp += (double)bce_term(a[e], b[e]);
            s_pt += (double)(a[e] * b[e]);   // the product is an fp32 value in the reference too (pred * target, then .sum())
            s_p += (double)a[e];
            s_t += (double)b[e];
        }
        if (p.thresh) {
            const floatx4 c = *(const floatx4*)(p.thresh + 4 * i), d = *(const floatx4*)(p.thresh_t + 4 * i);
#pragma unroll
            for (int e = 0; e < 4; ++e) s_bt += (double)bce_term(c[e], d[e]);
        }
    }
    // tail (n not a multiple of 4): the last workgroup's first lanes
    if (blockIdx.x == gridDim.x - 1) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        if (threadIdx.x < (p.n & 3)) {
            const float a = p.prob[i], b = p.prob_t[i];
            s_bp += (double)bce_term(a, b);
            s_pt += (double)(a * b);
            s_p += (double)a;
            s_t += (double)b;
            if (p.thresh) s_bt += (double)bce_term(p.thresh[i], p.thresh_t[i]);
        }
    }
    double v[5] = {s_bp, s_bt, s_pt, s_p, s_t};
    __shared__ double wsum[DBL_THREADS / 64][5];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v[k] += __shfl_down(v[k], off, 64);
        if (lane == 0) wsum[w][k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        double t = 0.0;
        for (int i = 0; i < DBL_THREADS / 64; ++i) t += wsum[i][threadIdx.x];
        p.partial[(int64_t)blockIdx.x * 5 + threadIdx.x] = t;
    }
}

// out[0..3] = prob BCE, threshold BCE, dice loss, total (float32, as the reference's tensors); sums[0..4] = the five sums (float64)
__global__ __launch_bounds__(64) void dbloss_finish_kernel(const double* partial, int blocks, int64_t n, float smooth, int has_thresh, float* out, double* sums) {
    // one wave: lane l sums workgroups l, l + 64, ... in order, then the fixed shuffle tree
    double s[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < blocks; b += 64)
#pragma unroll
        for (int k = 0; k < 5; ++k) s[k] += partial[(int64_t)b * 5 + k];
#pragma unroll
    for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s[k] += __shfl_down(s[k], off, 64);
    if (threadIdx.x != 0) return;
    // the reference's scalars are float32 tensors: mean-reduced BCE, and DiceLoss in float32 arithmetic on the three float32 sums
    const float prob_bce = (float)(s[0] / (double)n), thresh_bce = has_thresh ? (float)(s[1] / (double)n) : 0.0f;
    const float inter = (float)s[2], sp = (float)s[3], st = (float)s[4];
    const float dice = (2.0f * inter + smooth) / (sp + st + smooth);
    const float dice_loss = 1.0f - dice;
    out[0] = prob_bce;
    out[1] = thresh_bce;
    out[2] = dice_loss;
    out[3] = prob_bce + thresh_bce + dice_loss;   // trainer.py:56, same association
    if (sums)
        for (int k = 0; k < 5; ++k) sums[k] = s[k];
}

}  // namespace

int vtd_dbloss_ws_bytes() { return DBL_MAX_BLOCKS * 5 * (int)sizeof(double); }

int vtd_launch_dbloss(const float* prob, const float* thresh, const float* prob_t, const float* thresh_t, int64_t n, float smooth, double* workspace,
                      float* out4, double* sums5, hipStream_t stream) {
    if (!prob || !prob_t || !workspace || !out4 || n <= 0 || (thresh == nullptr) != (thresh_t == nullptr)) return -2701;
    if (((uintptr_t)prob | (uintptr_t)prob_t | (uintptr_t)thresh | (uintptr_t)thresh_t) & 15) return -2702;   // 16-byte loads
    // grid: a function of n only (fixed summation order for a given shape): enough 256-thread workgroups to put ~8 loads per lane in
    // flight on every CU for large maps, one workgroup per 4096 positions for small ones
    int64_t blocks = (n / 4 + DBL_THREADS * 4 - 1) / (DBL_THREADS * 4);
    if (blocks < 1) blocks = 1;
    if (blocks > DBL_MAX_BLOCKS) blocks = DBL_MAX_BLOCKS;
    DbLossParams p{prob, thresh, prob_t, thresh_t, workspace, n};
    hipLaunchKernelGGL(dbloss_partial_kernel, dim3((unsigned)blocks), dim3(DBL_THREADS), 0, stream, p);
    hipLaunchKernelGGL(dbloss_finish_kernel, dim3(1), dim3(64), 0, stream, (const double*)workspace, (int)blocks, n, smooth, thresh ? 1 : 0, out4, sums5);
    return -(int)hipGetLastError();
}
