// Implicit-GEMM convolution for gfx950 (MI355X): fp16 operands, fp32 MFMA accumulation.
//
// Replaces the ATen/oneDNN convolutions the reference reaches through nn.Conv2d / nn.ConvTranspose2d /
// nn.Linear in DBNet and CRNN (app/ml/models/text_detector.py:12-86, text_recognizer.py:12-37).
//
//   GEMM view      D[M x N] = A[M x K] * W^T,  M = n*ho*wo output pixels, N = Cout, K = kh*kw*Cin
//   activations    ring-padded NHWC fp16 (TensorDesc): every tap of every output pixel is in bounds, so
//                  the A tile is a pure gather of 16-byte chunks: address = pixel_base[m] + ktab[k/8]
//   weights        [Cout][K] fp16, K contiguous in ktab order, BatchNorm folded on the host
//   staging        global_load_lds_dwordx4 straight into LDS (no VGPR round trip), two LDS stages,
//                  BK = 64 (128-byte rows); rows are XOR-swizzled on the SOURCE side
//                  (chunk ^= (row>>1)&7) so every ds_read_b128 lane group hits 16 distinct 16-B slots
//   math           v_mfma_f32_16x16x32_f16, weights as the A operand and pixels as the B operand so one
//                  lane ends up with 4 consecutive channels of one pixel -> 8-byte packed stores
//   epilogue       +bias, optional residual (optionally nearest-2x upsampled: FPN top-down add),
//                  ReLU, fp16 NHWC store; or ConvTranspose(k2,s2) pixel shuffle; or fp32 row-major
//   scheduling     one 256-thread workgroup per BM x BN tile; the linear block id is re-dealt so that
//                  the 8 XCDs each own a contiguous run of tiles (neighbouring tiles share halo rows
//                  and the weight panel in that XCD's L2)
#include "vtd_common.h"

namespace {

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvParams p, const int tiles_n) {
    static_assert(WM * WN == 4, "four waves per workgroup");
    constexpr int A_BYTES = BM * 128;
    constexpr int B_BYTES = BN * 128;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int A_INST = BM / 32;  // global_load_lds instructions per wave per K-step for the A tile
    constexpr int B_INST = BN / 32;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int FM = TM / 16, FN = TN / 16;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    // ---- XCD-aware tile assignment (bijective for any grid size)
    const int nblk = gridDim.x, b = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = b & 7;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- loader state: each lane owns one 16-byte chunk slot of A_INST + B_INST rows
    const int lrow = lane >> 3;
    const int c_log = (lane & 7) ^ (((w & 1) << 2) | (lane >> 4));  // logical K-chunk this lane fetches
    const int howo = p.ho * p.wo;
    const half_t* aptr[A_INST];
#pragma unroll
    for (int i = 0; i < A_INST; ++i) {
        int m = m0 + (i * 4 + w) * 8 + lrow;
        m = m < p.M ? m : p.M - 1;
        const int img = m / howo;
        const int rem = m - img * howo;
        const int oy = rem / p.wo, ox = rem - oy * p.wo;
        aptr[i] = p.in + ((int64_t)(img * p.in_hp + oy * p.stride + p.in_y0) * p.in_wp + ox * p.stride + p.in_x0) * p.in_c;
    }
    const half_t* bptr[B_INST];
#pragma unroll
    for (int i = 0; i < B_INST; ++i) bptr[i] = p.wgt + (int64_t)(n0 + (i * 4 + w) * 8 + lrow) * p.K + c_log * 8;
    const int* ktab = p.ktab + c_log;

    auto stage = [&](int ks, int buf) {
        const int koff = ktab[ks * 8];
        char* abase = smem + buf * STAGE;
        char* bbase = abase + A_BYTES;
#pragma unroll
        for (int i = 0; i < A_INST; ++i)
            __builtin_amdgcn_global_load_lds((const VTD_AS1 void*)(aptr[i] + koff),
                                             (VTD_AS3 void*)(abase + (i * 4 + w) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < B_INST; ++i)
            __builtin_amdgcn_global_load_lds((const VTD_AS1 void*)(bptr[i] + ks * 64),
                                             (VTD_AS3 void*)(bbase + (i * 4 + w) * 1024), 16, 0, 0);
    };

    // ---- compute state
    const int wm = w / WN, wn = w - wm * WN;
    const int frow = lane & 15;
    const int swz = (lane >> 1) & 7;  // (row>>1)&7 for every fragment row this lane reads
    const int a_lane_off = (wm * TM + frow) * 128;
    const int b_lane_off = A_BYTES + (wn * TN + frow) * 128;

    floatx4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K >> 6;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int ks = 0; ks < nk; ++ks) {
        const int cur = ks & 1;
        if (ks + 1 < nk) stage(ks + 1, cur ^ 1);
        const char* sb = smem + cur * STAGE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int phys = (((lane >> 4) + 4 * kk) ^ swz) * 16;
            half8 af[FM], bf[FN];
#pragma unroll
            for (int j = 0; j < FM; ++j) af[j] = *(const half8*)(sb + a_lane_off + j * 2048 + phys);
#pragma unroll
            for (int i = 0; i < FN; ++i) bf[i] = *(const half8*)(sb + b_lane_off + i * 2048 + phys);
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int j = 0; j < FM; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[i], af[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- epilogue: lane holds channels ch..ch+3 (ch = .. + (lane>>4)*4) of pixel (lane&15) per fragment
    const int chq = (lane >> 4) * 4;
#pragma unroll
    for (int j = 0; j < FM; ++j) {
        const int m = m0 + wm * TM + j * 16 + frow;
        if (m >= p.M) continue;
        const int img = m / howo;
        const int rem = m - img * howo;
        const int oy = rem / p.wo, ox = rem - oy * p.wo;
#pragma unroll
        for (int i = 0; i < FN; ++i) {
            const int ch = n0 + wn * TN + i * 16 + chq;
            if (ch >= p.cout) continue;
            const float4 bv = *(const float4*)(p.bias + ch);
            float v[4] = {acc[i][j][0] + bv.x, acc[i][j][1] + bv.y, acc[i][j][2] + bv.z, acc[i][j][3] + bv.w};
            if (p.flags & EPI_RESIDUAL) {
                const int64_t ro = ((int64_t)(img * p.res_hp + (oy >> p.res_shift) + p.res_ring) * p.res_wp +
                                    (ox >> p.res_shift) + p.res_ring) * p.cout + ch;
                const half4 rv = *(const half4*)(p.res + ro);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
            }
            if (p.flags & EPI_RELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
            }
            if (p.flags & EPI_OUT_F32) {
                float* o = (float*)p.out + (int64_t)m * p.ldc + ch;
                if (ch + 3 < p.cout) {
                    *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    for (int e = 0; e < 4 && ch + e < p.cout; ++e) o[e] = v[e];
                }
            } else {
                int py = oy, px = ox, oc = ch;
                if (p.flags & EPI_PIXEL_SHUFFLE) {
                    const int blk = ch / p.ps_cout;
                    oc = ch - blk * p.ps_cout;
                    py = 2 * oy + (blk >> 1);
                    px = 2 * ox + (blk & 1);
                }
                const int64_t oo = ((int64_t)(img * p.out_hp + py + p.out_ring) * p.out_wp + px + p.out_ring) * p.out_c + oc;
                half4 hv = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                *(half4*)((half_t*)p.out + oo) = hv;
            }
        }
    }
}

template <int BM, int BN, int WM, int WN>
int launch_cfg(const ConvParams& p, hipStream_t stream) {
    const int tiles_m = (p.M + BM - 1) / BM;
    const int tiles_n = p.cout_pad / BN;
    constexpr int lds = 2 * (BM + BN) * 128;
    static bool attr_done = false;
    if (!attr_done) {
        VTD_HIP_CHECK(hipFuncSetAttribute((const void*)conv_igemm_kernel<BM, BN, WM, WN>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_done = true;
    }
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN>), dim3(tiles_m * tiles_n), dim3(256), lds, stream, p, tiles_n);
    return -(int)hipGetLastError();
}

}  // namespace

// Host entry used by the network graphs in vtd_api.cpp.  Shapes are validated here: a mismatch must
// never reach the kernel (an out-of-bounds gather can take the whole node down).
int vtd_launch_conv(const ConvParams& p, hipStream_t stream) {
    if (p.M <= 0 || p.K <= 0 || (p.K & 63) || p.cout <= 0 || (p.cout_pad & 63) || p.cout > p.cout_pad) return -1001;
    if ((p.cout & 3) && !(p.flags & EPI_OUT_F32)) return -1002;
    if ((p.flags & EPI_OUT_F32) && (p.ldc & 3)) return -1003;
    if (p.cout_pad % 128 == 0) return launch_cfg<128, 128, 2, 2>(p, stream);
    return launch_cfg<256, 64, 4, 1>(p, stream);
}
