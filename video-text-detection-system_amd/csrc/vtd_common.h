// Internal shared definitions for the gfx950 HIP kernels (not part of the C ABI; see include/vtd.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 half_t;
typedef half_t half8 __attribute__((ext_vector_type(8)));
typedef half_t half4 __attribute__((ext_vector_type(4)));
typedef half_t half2v __attribute__((ext_vector_type(2)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

#define VTD_AS1 __attribute__((address_space(1)))
#define VTD_AS3 __attribute__((address_space(3)))

// Activation tensor in HBM: NHWC fp16 with a zero ring of `ring` pixels on every side of every image
// (ring_b/ring_r may be larger on the network input).  Rings are zeroed once at allocation and never
// written, so 3x3/7x7 taps need no bounds checks and `buffer/global_load ... lds` can fetch them blind.
struct TensorDesc {
    half_t* ptr;
    int n, h, w, c;   // logical extents
    int ring;         // top/left ring (pixels)
    int hp, wp;       // padded extents (h + ring + ring_bottom, w + ring + ring_right)
};

static inline __host__ __device__ int64_t tensor_elems(const TensorDesc& t) {
    return (int64_t)t.n * t.hp * t.wp * t.c;
}

// Epilogue modes of the implicit-GEMM convolution
enum : int {
    EPI_RELU = 1,          // max(x, 0)
    EPI_RESIDUAL = 2,      // += res[n, oy>>res_shift, ox>>res_shift, ch]   (res_shift=1: fused nearest-2x upsample)
    EPI_PIXEL_SHUFFLE = 4, // ConvTranspose2d(k=2,s=2) as GEMM: channel block q=(ky*2+kx) -> pixel (2oy+ky, 2ox+kx)
    EPI_OUT_F32 = 8,       // plain [M, ldc] float32 row-major output (LSTM gate pre-activations, logits)
    EPI_OUT_F16 = 32,      // plain [M, ldc] fp16 row-major output (LSTM gate pre-activations)
    EPI_GELU = 64,         // exact (erf) GELU after the bias: the Transformer recogniser's MLPs
    EPI_HEAD_FINAL = 16,   // DB head tail fused: this GEMM is ConvT(64->64,k2,s2)+BN+ReLU; the epilogue applies
                           // ConvT(64->1,k2,s2)+sigmoid and writes the 4x4 fp32 probabilities of each input pixel
};

struct ConvParams {
    const half_t* in;      // input tensor base (ring-padded NHWC)
    const half_t* wgt;     // packed weights [Cout_pad][K] fp16, K contiguous, K in tap-major order
    // K walk, all wave-uniform so it runs on the scalar unit: a tap is `cin_steps` K-steps of 64 channels; after a
    // tap the gather moves `s_step` elements (next kernel column), after `kw` taps `r_step` (next kernel row)
    int cin_steps, kw, s_step, r_step;
    int k_hi_step;         // element offset between chunks 0-3 and 4-7 of a K-step (32 = contiguous; stem: one input row)
    const float* bias;     // [Cout_pad] fp32 (conv bias and BatchNorm folded)
    const half_t* res;     // optional residual tensor (ring-padded NHWC, C = cout)
    void* out;             // output tensor base
    int M;                 // n * ho * wo
    int K;                 // multiple of 64
    int cout;              // valid output channels
    int cout_pad;          // multiple of BN
    int ho, wo;            // output spatial extents (GEMM rows decompose to n, oy, ox)
    int in_hp, in_wp, in_c; // padded input extents and channel stride (elements per pixel)
    int in_y0, in_x0;      // ring_in - pad
    int stride;
    int out_hp, out_wp, out_c, out_ring;
    int res_hp, res_wp, res_ring, res_shift;
    int ps_cout;           // pixel-shuffle: channels per (ky,kx) block
    int flags;
    int dbg;               // timing experiments only (VTD_CONV_DEBUG): 1 = A gather pinned to tap 0, 2 = no A loads, 3 = no loads, 4 = no loop barrier, 5 = no epilogue, 6 = no fragment reads after the first
    int ldc;               // EPI_OUT_F32 row stride
    // EPI_OUT_F16 in column segments: columns [i * seg_cols, (i + 1) * seg_cols) go to seg_out[i] with row stride seg_ldc[i] (0: one
    // destination, `out` / `ldc`).  seg_cols is a multiple of every tile width, so a tile has one destination.  The TrOCR decoder's
    // q | k | v projections run as ONE GEMM that writes q, the key-cache row and the value-cache row.
    int seg_cols;
    void* seg_out[3];
    int seg_ldc[3];
    int epi_direct;        // set by vtd_launch_conv: epilogue straight from the accumulators (conv_igemm.hip)
    // max-pool 2 x pool_pw (stride = window) behind the ReLU, fused into the register epilogue: pool_pw = 1 or 2 (0: none); `out` is then
    // the POOLED tensor.  pool_log2 / pool_wq / pool_hqwq are derived by vtd_launch_conv (window size log2, pooled width, pooled pixels per image)
    int pool_pw, pool_log2, pool_wq, pool_hqwq;
    uint64_t magic_wo, magic_howo;  // ceil(2^40 / wo), ceil(2^40 / (ho * wo)) for it (pooled: of the pooled map)
    int magic_ok;          // set by vtd_launch_conv: the two multiplications are exact for every row of this launch (loader + epilogue)
    // ---- classed dual-source mode (fused FPN-top + head entry, see vtd_api.cpp: compose_head_entry)
    const uint32_t* plist;   // per-image pixel list, tile-aligned: y | x << 16, 0xffffffff = padding row
    const int* tile_combo;   // per tile in execution order: weight class | pixel-list chunk << 8
    const uint32_t* plist_b; // the same lists cut into 256-row tiles (configurations 10 / 11)
    const int* tile_combo_b;
    int tiles_per_img_b;
    int tiles_per_img;
    const half_t* in2;       // second source (L3): gathered at (y>>1, x>>1) - 1
    int in2_hp, in2_wp, in2_c, in2_ring;
    // ---- second K segment of a plain conv (in2 set, no pixel list; conv_igemm.hip DUAL): K-steps >= seg1_steps gather the 1x1 window of
    // in2 at (oy * in2_mul >> in2_shr, ox * in2_mul >> in2_shr) + in2_ring, cin_steps2 K-steps of 64 channels (kw2 = 1)
    int in2_mul, in2_shr;
    int seg1_steps;          // K-steps served by `in` (5x5 window); the rest walk `in2` (3x3 window)
    int cin_steps2, kw2, s_step2, r_step2;
    const float* bias_tab;   // [25][cout] border-class bias (y class * 5 + x class)
    int img_h, img_w;        // extents used for the border classes
    const float* head_w;   // EPI_HEAD_FINAL: [4][64] final ConvT weights (ky2*2+kx2 major), head_b its bias
    float head_b;
    float* prob_out;       // EPI_HEAD_FINAL: [n, 4*ho, 4*wo] float32
};

#define VTD_HIP_CHECK(expr)                                   \
    do {                                                      \
        hipError_t _e = (expr);                               \
        if (_e != hipSuccess) return -(int)_e;                \
    } while (0)
