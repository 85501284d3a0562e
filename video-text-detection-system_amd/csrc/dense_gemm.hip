// Dense GEMM for the Transformer recogniser's encoder pass:  out[M, N] = A[M, K] * W[N, K]^T + bias  (optionally exact GELU),
// A / W fp16 with K contiguous, fp32 accumulation, fp16 or fp32 output.  M is the token count of a whole crop batch (272 crops x 577
// tokens = 157 k rows), N = 768 ... 3072: thousands of 256 x 256 tiles, which is the regime where the implicit-GEMM kernel (conv_igemm:
// 256 x 128 tiles, one barrier per K-step, the two waves of a SIMD in lockstep behind it) leaves half of the matrix pipe idle.
//
// Structure (cdna_hip_programming.md, "The 256^2 8-phase template", re-derived for BK = 32 stages):
//   * 256 x 256 tile, 8 waves = 2 (M) x 4 (N), a wave owns 128 x 64 = 8 x 4 accumulator fragments (128 VGPRs);
//   * operands arrive by LDS-DMA (global_load_lds_dwordx4) in K stages of 32 through a ring of four 32 KB stages, three stages ahead
//     of the maths behind COUNTED s_waitcnt vmcnt -- never 0 inside the loop;
//   * 64-byte LDS rows, 16-byte chunks XOR-swizzled by (row >> 2) & 3 on the SOURCE side (the DMA writes linearly) and on the read
//     side, so the 16 lanes of a ds_read_b128 group hit 16 different bank quads;
//   * a stage is two barrier intervals per wave: R (12 fragment reads of the stage + the wait that lets the NEXT stage's DMA land) and
//     MM (issue the DMA three stages ahead, 32 MFMAs).  The two wave groups (wm = 0 / 1: the two waves of every SIMD) run ONE interval
//     apart -- group 1 passes an extra barrier at the start, group 0 one at the end -- so while one wave of a SIMD multiplies, the other
//     reads its fragments, instead of both stalling on LDS and then both queueing for the pipe;
//   * hazards (checked in the comments at the barriers): a stage is read one interval AFTER the wait + barrier that published it, and
//     restaged only after a barrier that every reader reached with its fragments consumed by issued MFMAs;
//   * epilogue straight from the accumulators: weight rows are permuted on the source side so that a lane owns 8 consecutive output
//     channels -> bias, GELU, one 16-byte store (fp16) or two (fp32) per row and channel group.
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
};

template <int N>
__device__ __forceinline__ void dgm_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ void dgm_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

__global__ __launch_bounds__(512) void dense_gemm_kernel(const DenseGemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char dgm_smem[];
    // ---- XCD-aware tile order (bijective for any grid): an XCD sweeps a contiguous run of tiles, N fastest, so the 256-row A panel
    // of a tile row is fetched from HBM once per XCD
    const int nblk = gridDim.x, b = blockIdx.x;
    const int q = nblk >> 3, r8 = nblk & 7, xcd = b & 7;
    const int tile = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (b >> 3);
    const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
    const int m0 = tm * DGM_BM, n0 = tn * DGM_BN;
    const int tid = threadIdx.x, lane = tid & 63, fr = lane & 15, fq = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = w >> 2, wn = w & 3;

    // ---- loader: wave w brings pieces w and w + 8 (16 rows x 64 bytes each) of the A tile and of the B tile of every stage.
    // Lane l of a piece lands at LDS row 16 piece + (l >> 2), physical chunk l & 3, and therefore fetches logical chunk
    // (l & 3) ^ ((row >> 2) & 3) of that row.
    const half_t* asrc[2];
    const half_t* bsrc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (w + 8 * i) * 16 + (lane >> 2);
        const int chunk = (lane & 3) ^ ((row >> 2) & 3);
        int m = m0 + row;
        m = m < p.M ? m : p.M - 1;                       // rows past the end re-read the last row (never written)
        asrc[i] = p.A + (int64_t)m * p.lda + chunk * 8;
        // B row `row` of the tile feeds MFMA row (row & 15) of fragment (row >> 4) & 3 of wave column row >> 6; it is fed with the
        // weights of the channel that makes a lane's accumulators 8 consecutive channels (see the epilogue)
        const int x = row & 63, blk = x >> 4, rr = x & 15;
        int n = n0 + (row - x) + 32 * (blk >> 1) + 8 * (rr >> 2) + 4 * (blk & 1) + (rr & 3);
        n = n < p.w_rows ? n : p.w_rows - 1;
        bsrc[i] = p.W + (int64_t)n * p.K + chunk * 8;
    }
    auto issue = [&](int stage) {   // stage index s: K offset 32 s, ring slot s & 3
        char* base = dgm_smem + (stage & (DGM_NST - 1)) * DGM_STAGE;
        const int k = stage * DGM_BK;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const VTD_AS1 void*)(asrc[i] + k), (VTD_AS3 void*)(base + (w + 8 * i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const VTD_AS1 void*)(bsrc[i] + k), (VTD_AS3 void*)(base + DGM_BM * 64 + (w + 8 * i) * 1024), 16, 0, 0);
    };

    // ---- fragment read offsets (bytes inside a stage): A rows wm * 128 + 16 i + fr, B rows wn * 64 + 16 j + fr, logical chunk fq
    int a_off[8], b_off[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = wm * 128 + i * 16 + fr;
        a_off[i] = row * 64 + ((fq ^ ((row >> 2) & 3)) << 4);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = wn * 64 + j * 16 + fr;
        b_off[j] = DGM_BM * 64 + row * 64 + ((fq ^ ((row >> 2) & 3)) << 4);
    }

    floatx4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int S = p.K / DGM_BK;   // >= 4 (launcher)
    issue(0);
    issue(1);
    issue(2);
    dgm_wait_vmcnt<8>();          // this wave's pieces of stage 0 have landed (stages 1, 2 stay in flight)
    dgm_barrier();                // hw barrier 0: stage 0 is published
    if (wm == 1) dgm_barrier();   // group 1 runs one interval behind group 0 from here on

    half8 af[8], bf[4];
    for (int s = 0; s < S; ++s) {
        // ---- interval R_s: fragments of stage s.  (Published: every wave waited for its stage-s pieces at the end of its R_{s-1}
        // -- or in the prologue -- and has passed a barrier since; this wave is at least one barrier past the last of those waits.)
        const char* st = dgm_smem + (s & (DGM_NST - 1)) * DGM_STAGE;
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[j] = *(const half8*)(st + b_off[j]);
#pragma unroll
        for (int i = 0; i < 8; ++i) af[i] = *(const half8*)(st + a_off[i]);
        // stage s + 1 must have landed before the barrier that ends this interval; stage s + 2 (issued in MM_{s-1}) may stay in flight
        if (s + 2 < S) dgm_wait_vmcnt<4>(); else dgm_wait_vmcnt<0>();
        dgm_barrier();
        // ---- interval MM_s: restage ring slot (s + 3) & 3 = (s - 1) & 3.  Its last readers were the R_{s-1} intervals: this group's
        // ended two barriers ago, the other group's one barrier ago at the latest, and every wave reached that barrier only after its
        // MFMAs of MM_{s-1} -- which consume those fragments -- had been issued, i.e. after the reads had returned.
        if (s + 3 < S) issue(s + 3);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        dgm_barrier();
    }
    if (wm == 0) dgm_barrier();   // group 0 pays back group 1's extra barrier (every wave executes 2 S + 2 barriers)

    // ---- epilogue from registers: lane (fr, fq) holds row m0 + wm * 128 + 16 i + fr, channels n0 + wn * 64 + 32 jp + 8 fq + 0..7
    // (acc[i][2 jp][0..3] then acc[i][2 jp + 1][0..3], by the row permutation of the loader)
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
        const int n = n0 + wn * 64 + jp * 32 + fq * 8;
        if (n >= p.N) continue;
        const floatx4 b0 = *(const floatx4*)(p.bias + n), b1 = *(const floatx4*)(p.bias + n + 4);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = m0 + wm * 128 + i * 16 + fr;
            if (m >= p.M) continue;
            floatx4 v0 = acc[i][2 * jp] + b0, v1 = acc[i][2 * jp + 1] + b1;
            if (p.flags & EPI_GELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v0[e] = 0.5f * v0[e] * (1.0f + erff(v0[e] * 0.70710678118654752f));
                    v1[e] = 0.5f * v1[e] * (1.0f + erff(v1[e] * 0.70710678118654752f));
                }
            }
            if (p.flags & EPI_OUT_F16) {
                half8 h;
#pragma unroll
                for (int e = 0; e < 4; ++e) { h[e] = (half_t)v0[e]; h[4 + e] = (half_t)v1[e]; }
                *(half8*)((half_t*)p.out + (int64_t)m * p.ldc + n) = h;
            } else {
                float* o = (float*)p.out + (int64_t)m * p.ldc + n;
                *(floatx4*)o = v0;
                *(floatx4*)(o + 4) = v1;
            }
        }
    }
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

int vtd_launch_dense_gemm(const half_t* A, int lda, const half_t* W, int w_rows, const float* bias, void* out, int ldc, int64_t M, int N, int K,
                          int flags, hipStream_t stream) {
    if (M < 256 || N < 256 || (N & 7) || (K & 31) || K < 128 || (lda & 7) || (ldc & 7) || M > 0x7fffffff ||
        !(flags & (EPI_OUT_F16 | EPI_OUT_F32)) || (flags & ~(EPI_OUT_F16 | EPI_OUT_F32 | EPI_GELU)))
        return -2601;
    DenseGemmParams p{A, W, bias, out, (int)M, N, K, lda, ldc, w_rows, (N + DGM_BN - 1) / DGM_BN, flags};
    static bool attr = false;
    if (!attr) {
        VTD_HIP_CHECK(hipFuncSetAttribute((const void*)dense_gemm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DGM_LDS));
        attr = true;
    }
    const int64_t tiles = ((M + DGM_BM - 1) / DGM_BM) * p.tiles_n;
    hipLaunchKernelGGL(dense_gemm_kernel, dim3((unsigned)tiles), dim3(512), DGM_LDS, stream, p);
    return -(int)hipGetLastError();
}
