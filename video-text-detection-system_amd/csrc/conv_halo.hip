// 3x3 / stride 1 / pad 1 convolution (+ folded BN, residual, ReLU) with HALO-TILE staging of the input.
// Serves the stride-1 3x3 layers of the ResNet stages (reference: torchvision BasicBlock / Bottleneck conv3x3 under
// app/ml/models/text_detector.py:25-33) and of the CRNN stack (text_recognizer.py:15-30).
//
// Why a second convolution kernel.  conv_igemm.hip gathers an im2col row per tap: every input pixel travels global -> LDS
// nine times.  Timing experiments show that this gather -- LDS-DMA issue + the per-CU TA path, not HBM and
// not LDS reads -- is what bounds the layers with few output channels.  Here a workgroup owns a TH x TW block of output
// pixels (256 GEMM rows) and stages the (TH+2) x (TW+2) input halo of one 64-channel chunk ONCE; the nine taps are nine
// K-steps that read the same LDS image at shifted rows.  Per K-step only the weights (BN x 128 bytes) still stream in.
//   * LDS image of the halo: one 128-byte row per pixel, 16-byte chunks XOR-swizzled by ((row >> 1) & 7) on the source
//     side of the LDS-DMA, so a 16-lane ds_read_b128 group reading 16 consecutive pixels is conflict-free for even
//     tap shifts and 2-way for odd ones;
//   * weights: 3-stage ring filled by LDS-DMA, counted vmcnt + one s_barrier per K-step (as conv_igemm.hip);
//     the halo of the next channel chunk is fetched piecewise during the current chunk's K-steps;
//   * waves: 4 x (BN/64), each a 64 x 64 sub-tile (16 accumulator fragments), v_mfma_f32_16x16x32_f16, weights as the first
//     operand so a lane owns 4 consecutive channels; epilogue through an fp32 LDS tile -> 16-byte NHWC stores.
#include <cstdio>
#include <cstdlib>
#include "vtd_common.h"

namespace {

struct HaloParams {
    const half_t* in;     // [n][in_hp][in_wp][cin] fp16, ring in_ring >= 1 (zero)
    const half_t* wgt;    // [cout][9*cin] fp16, K = (r*3+s)*cin + c  (conv_igemm's packing)
    const float* bias;    // [cout]
    const half_t* res;    // optional residual, [n][res_hp][res_wp][cout]
    half_t* out;          // [n][out_hp][out_wp][cout]
    int n, h, w, cin, cout;
    int in_hp, in_wp, in_ring, out_hp, out_wp, out_ring, res_hp, res_wp, res_ring;
    int tiles_x, tiles_y, tiles_n, relu;
    unsigned long long* stamps;  // debug (VTD_HALO_STAMPS=1): 4 s_memtime stamps per workgroup
};

template <int N>
__device__ __forceinline__ void hl_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int BN, int TW>
__global__ __launch_bounds__((BN / 64) * 256) void conv_halo_kernel(const HaloParams p) {
    constexpr int WN = BN / 64, NW = 4 * WN, NT = NW * 64, TH = 256 / TW;
    constexpr int HWD = TW + 2, HROWS = (TH + 2) * HWD, HPIECES = (HROWS + 7) / 8, HBYTES = HPIECES * 1024;
    constexpr int HPW = (HPIECES + NW - 1) / NW;  // halo pieces per wave
    constexpr int HPS = (HPW + 7) / 8;            // of which issued per K-step while the previous chunk computes (taps 0..7)
    constexpr int BSTAGE = BN * 128, BPW = BN / 8 / NW;
    constexpr int EPI_ROW = BN * 4 + 16;
    static_assert(BPW == 2, "two weight pieces per wave and K-step");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int nchunks = p.cin >> 6;
    const int hb_stride = nchunks > 1 ? HBYTES : 0;
    char* const bring = smem + (nchunks > 1 ? 2 : 1) * HBYTES;

    // ---- XCD-aware tile assignment (bijective for any grid size); the N tiles of one pixel block are neighbours
    const int nblk = gridDim.x, b = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = b & 7;
    int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b >> 3);
    const int tn = tile % p.tiles_n;
    tile /= p.tiles_n;
    const int tx = tile % p.tiles_x;
    tile /= p.tiles_x;
    const int ty = tile % p.tiles_y;
    const int img = tile / p.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane >> 3;
    const int K = 9 * p.cin;

    // ---- halo loader: this lane's source element offset for each of the wave's pieces (chunk 0)
    int hoff[HPW];
#pragma unroll
    for (int k = 0; k < HPW; ++k) {
        const int piece = w + k * NW;
        int row = piece * 8 + lrow;
        row = row < HROWS ? row : HROWS - 1;
        const int hy = row / HWD, hx = row - hy * HWD;
        int gy = y0 - 1 + hy + p.in_ring, gx = x0 - 1 + hx + p.in_ring;
        gy = gy < p.in_hp ? gy : p.in_hp - 1;  // partial tiles: stay inside the allocation (those outputs are masked)
        gx = gx < p.in_wp ? gx : p.in_wp - 1;
        const int c_log = (lane & 7) ^ ((row >> 1) & 7);
        hoff[k] = ((img * p.in_hp + gy) * p.in_wp + gx) * p.cin + c_log * 8;
    }
    auto issue_halo = [&](int k, int chunk, int buf) {
        const int piece = w + k * NW;
        if (piece < HPIECES)
            __builtin_amdgcn_global_load_lds((const VTD_AS1 void*)(p.in + hoff[k] + chunk * 64),
                                             (VTD_AS3 void*)(smem + buf * hb_stride + piece * 1024), 16, 0, 0);
    };
    // ---- weight loader: rows n0 + (i*NW + w)*8 + lrow of the [cout][K] matrix, 128 bytes per K-step
    const half_t* bsrc[BPW];
#pragma unroll
    for (int i = 0; i < BPW; ++i) {
        const int row = (i * NW + w) * 8 + lrow;
        const int c_log = (lane & 7) ^ ((row >> 1) & 7);
        bsrc[i] = p.wgt + (int64_t)(n0 + row) * K + c_log * 8;
    }
    auto issue_b = [&](int step, int stage) {
        const int chunk = step / 9, tap = step - chunk * 9;
        const int koff = tap * p.cin + chunk * 64;
#pragma unroll
        for (int i = 0; i < BPW; ++i)
            __builtin_amdgcn_global_load_lds((const VTD_AS1 void*)(bsrc[i] + koff),
                                             (VTD_AS3 void*)(bring + stage * BSTAGE + (i * NW + w) * 1024), 16, 0, 0);
    };

    // ---- compute state: wave (wm, wn) owns rows wm*64.. (4 tile rows when TW = 16, 2 when TW = 32) x channels wn*64..
    const int wm = w / WN, wn = w - wm * WN;
    const int fr = lane & 15, fq = lane >> 4;
    int hbase[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = wm * 64 + j * 16 + fr;
        hbase[j] = (m / TW) * HWD + (m % TW);
    }
    const int b_lane_off = (wn * 64 + fr) * 128;
    const int bswz = (fr >> 1) & 7;

    floatx4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int nsteps = nchunks * 9;
    unsigned long long t_start = 0, t_first = 0, t_loop = 0;
    if (p.stamps) t_start = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int k = 0; k < HPW; ++k) issue_halo(k, 0, 0);
    issue_b(0, 0);
    issue_b(1, 1);

    // K loop: channel chunks outside, the nine taps unrolled inside (tap, ring stage = tap % 3 and the halo piece indices are
    // compile-time constants)
    int s = 0;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const char* hb = smem + (chunk & 1) * hb_stride;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap, ++s) {
            constexpr int kStageOf[9] = {0, 1, 2, 0, 1, 2, 0, 1, 2};
            const int stage = kStageOf[tap];
            if (s + 1 < nsteps) hl_wait_vmcnt<BPW>(); else hl_wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();  // weights of step s (and any halo piece issued before them) landed; stage s+2 is free
            if (p.stamps && s == 0) t_first = __builtin_amdgcn_s_memtime();
            if (tap < 8 && chunk + 1 < nchunks) {
#pragma unroll
                for (int e = 0; e < HPS; ++e)
                    if (tap * HPS + e < HPW) issue_halo(tap * HPS + e, chunk + 1, (chunk + 1) & 1);
            }
            if (s + 2 < nsteps) issue_b(s + 2, kStageOf[(tap + 2) % 9]);
            const char* sb = bring + stage * BSTAGE;
            const int tapoff = (tap / 3) * HWD + (tap % 3);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                half8 af[4], bf[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int hrow = hbase[j] + tapoff;
                    af[j] = *(const half8*)(hb + hrow * 128 + (((fq + 4 * kk) ^ ((hrow >> 1) & 7)) << 4));
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) bf[i] = *(const half8*)(sb + b_lane_off + i * 2048 + (((fq + 4 * kk) ^ bswz) << 4));
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[i], af[j], acc[i][j], 0, 0, 0);
            }
        }
    }

    // ---- epilogue phase 1: accumulators -> fp32 tile in LDS (lane: 4 consecutive channels of pixel fr)
    __syncthreads();
    if (p.stamps) t_loop = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *(floatx4*)(smem + (wm * 64 + j * 16 + fr) * EPI_ROW + (wn * 64 + i * 16 + fq * 4) * 4) = acc[i][j];
    __syncthreads();

    // ---- phase 2: thread = 8 consecutive channels of one pixel: bias, residual, ReLU, one 16-byte store
    constexpr int CPR = BN / 8, RPI = NT / CPR;
    const int cc = tid % CPR, r0 = tid / CPR;
    const int ch = n0 + cc * 8;
    const floatx4 bias0 = *(const floatx4*)(p.bias + ch), bias1 = *(const floatx4*)(p.bias + ch + 4);
#pragma unroll 4
    for (int it = 0; it < 256 / RPI; ++it) {
        const int m = it * RPI + r0;
        const int y = y0 + m / TW, x = x0 + m % TW;
        if (y >= p.h || x >= p.w) continue;
        floatx4 v0 = *(const floatx4*)(smem + m * EPI_ROW + cc * 32);
        floatx4 v1 = *(const floatx4*)(smem + m * EPI_ROW + cc * 32 + 16);
        v0 += bias0;
        v1 += bias1;
        if (p.res) {
            const half8 rv = *(const half8*)(p.res + (((int64_t)img * p.res_hp + y + p.res_ring) * p.res_wp + x + p.res_ring) * p.cout + ch);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v0[e] += (float)rv[e]; v1[e] += (float)rv[4 + e]; }
        }
        half8 hv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = v0[e], c = v1[e];
            hv[e] = (half_t)((p.relu && a < 0.f) ? 0.f : a);
            hv[4 + e] = (half_t)((p.relu && c < 0.f) ? 0.f : c);
        }
        *(half8*)(p.out + (((int64_t)img * p.out_hp + y + p.out_ring) * p.out_wp + x + p.out_ring) * p.cout + ch) = hv;
    }
    if (p.stamps && tid == 0) {
        unsigned long long* o = p.stamps + (int64_t)blockIdx.x * 4;
        o[0] = t_start; o[1] = t_first; o[2] = t_loop; o[3] = __builtin_amdgcn_s_memtime();
    }
}

template <int BN, int TW>
int halo_launch(const HaloParams& p, hipStream_t stream) {
    constexpr int TH = 256 / TW, HROWS = (TH + 2) * (TW + 2), HBYTES = (HROWS + 7) / 8 * 1024;
    const int nchunks = p.cin / 64;
    const int main_bytes = (nchunks > 1 ? 2 : 1) * HBYTES + 3 * BN * 128;
    const int epi_bytes = 256 * (BN * 4 + 16);
    const int lds = main_bytes > epi_bytes ? main_bytes : epi_bytes;
    if (lds > 160 * 1024) return -2203;
    static int attr_lds = 0;
    if (lds > attr_lds) {
        hipError_t e = hipFuncSetAttribute((const void*)conv_halo_kernel<BN, TW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return -(int)e;
        attr_lds = lds;
    }
    const int grid = p.n * p.tiles_y * p.tiles_x * p.tiles_n;
    static const bool want_stamps = [] { const char* e = getenv("VTD_HALO_STAMPS"); return e && e[0] == '1'; }();
    if (want_stamps) {  // debug: phase timing of every workgroup, printed to stderr (synchronises!)
        HaloParams q = p;
        unsigned long long* dev = nullptr;
        if (hipMalloc(&dev, (size_t)grid * 32) != hipSuccess) return -2204;
        q.stamps = dev;
        hipLaunchKernelGGL((conv_halo_kernel<BN, TW>), dim3(grid), dim3((BN / 64) * 256), lds, stream, q);
        (void)hipStreamSynchronize(stream);
        unsigned long long* h = (unsigned long long*)malloc((size_t)grid * 32);
        (void)hipMemcpy(h, dev, (size_t)grid * 32, hipMemcpyDeviceToHost);
        double pro = 0, loop = 0, epi = 0;
        unsigned long long t0 = ~0ull, t1 = 0;
        for (int i = 0; i < grid; ++i) {
            pro += (double)(h[4 * i + 1] - h[4 * i]); loop += (double)(h[4 * i + 2] - h[4 * i + 1]); epi += (double)(h[4 * i + 3] - h[4 * i + 2]);
            t0 = h[4 * i] < t0 ? h[4 * i] : t0; t1 = h[4 * i + 3] > t1 ? h[4 * i + 3] : t1;
        }
        fprintf(stderr, "[halo stamps] BN=%d TW=%d cin=%d cout=%d grid=%d: per-WG ticks (100 MHz) prologue %.1f  K-loop %.1f  epilogue %.1f; kernel span %llu ticks\n",
                BN, TW, p.cin, p.cout, grid, pro / grid, loop / grid, epi / grid, t1 - t0);
        free(h);
        (void)hipFree(dev);
        return 0;
    }
    hipLaunchKernelGGL((conv_halo_kernel<BN, TW>), dim3(grid), dim3((BN / 64) * 256), lds, stream, p);
    return -(int)hipGetLastError();
}

}  // namespace

// Shape test: can (and should) this convolution run on the halo kernel?  *tw_out = 16 or 32 picks the pixel-block shape.
bool vtd_conv_halo_supported(const ConvParams& c, int* bn_out, int* tw_out) {
    if (c.plist || c.stride != 1 || c.kw != 3 || c.K != 9 * c.in_c || (c.in_c & 63)) return false;  // 3x3, stride 1, whole chunks
    if (c.cin_steps * 64 != c.in_c || c.s_step != c.in_c || c.r_step != c.in_wp * c.in_c || c.k_hi_step != 32) return false;
    if ((c.cout & 63) || c.cout != c.cout_pad || c.out_c != c.cout) return false;
    if (c.flags & ~(EPI_RELU | EPI_RESIDUAL)) return false;
    if ((c.flags & EPI_RESIDUAL) && (c.res_shift || !c.res)) return false;
    if (c.in_y0 < 0 || c.in_x0 != c.in_y0 || c.ho <= 0 || c.wo <= 0 || c.M % (c.ho * c.wo)) return false;
    // pixel-block shape: 16x16, or 8x32 for short wide maps; at least 85 % of the block grid must be real pixels
    const int h = c.ho, w = c.wo;
    auto eff = [&](int th, int tw) { return (double)h * w / ((double)((h + th - 1) / th * th) * ((w + tw - 1) / tw * tw)); };
    const double e16 = eff(16, 16), e32 = eff(8, 32);
    const int tw = e16 >= e32 ? 16 : 32;
    if ((tw == 16 ? e16 : e32) < 0.85) return false;
    *tw_out = tw;
    *bn_out = (c.cout % 128 == 0) ? 128 : 64;
    return true;
}

int vtd_launch_conv_halo(const ConvParams& c, int bn, int tw, hipStream_t stream) {
    HaloParams p;
    p.in = c.in; p.wgt = c.wgt; p.bias = c.bias; p.res = (c.flags & EPI_RESIDUAL) ? c.res : nullptr; p.out = (half_t*)c.out;
    p.h = c.ho; p.w = c.wo; p.n = c.M / (c.ho * c.wo); p.cin = c.in_c; p.cout = c.cout;
    p.in_hp = c.in_hp; p.in_wp = c.in_wp; p.in_ring = c.in_y0 + 1;  // in_y0 = ring - pad
    p.out_hp = c.out_hp; p.out_wp = c.out_wp; p.out_ring = c.out_ring;
    p.res_hp = c.res_hp; p.res_wp = c.res_wp; p.res_ring = c.res_ring;
    p.relu = (c.flags & EPI_RELU) ? 1 : 0;
    p.stamps = nullptr;
    if (c.in_y0 != c.in_x0 || c.K != 9 * c.in_c || c.out_c != c.cout || p.n <= 0) return -2201;
    if ((int64_t)p.n * c.in_hp * c.in_wp * c.in_c >= (1ll << 31)) return -2202;  // 32-bit element offsets in the loader
    const int th = 256 / tw;
    p.tiles_x = (p.w + tw - 1) / tw; p.tiles_y = (p.h + th - 1) / th; p.tiles_n = p.cout / bn;
    if (bn == 64 && tw == 16) return halo_launch<64, 16>(p, stream);
    if (bn == 64 && tw == 32) return halo_launch<64, 32>(p, stream);
    if (bn == 128 && tw == 16) return halo_launch<128, 16>(p, stream);
    if (bn == 128 && tw == 32) return halo_launch<128, 32>(p, stream);
    return -2201;
}
