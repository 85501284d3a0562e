// Implicit-GEMM convolution for gfx950 (MI355X): fp16 operands, fp32 MFMA accumulation.
//
// Replaces the ATen/oneDNN convolutions the reference reaches through nn.Conv2d / nn.ConvTranspose2d /
// nn.Linear in DBNet and CRNN (app/ml/models/text_detector.py:12-86, text_recognizer.py:12-37).
//
//   GEMM view      D[M x N] = A[M x K] * W^T,  M = n*ho*wo output pixels, N = Cout, K = kh*kw*Cin
//   activations    ring-padded NHWC fp16 (TensorDesc): every tap of every output pixel is in bounds, so
//                  the A tile is a pure gather of 16-byte chunks: pixel_base[m] + tap offset + chunk offset.
//                  The tap walk (channel step, kernel column, kernel row) is wave-uniform scalar arithmetic, so the
//                  only vector-memory traffic inside the main loop is the LDS-DMA itself
//   weights        [Cout][K] fp16, K contiguous in tap-major order, BatchNorm folded on the host
//   staging        global_load_lds_dwordx4 straight into LDS (no VGPR round trip), STAGES LDS buffers of
//                  BK = 64 (128-byte rows); two K-steps stay in flight across the (raw) barrier behind a counted
//                  s_waitcnt vmcnt; rows are XOR-swizzled on the SOURCE side (chunk ^= (row>>1)&7) so every
//                  ds_read_b128 lane group hits 16 distinct 16-byte slots
//   math           v_mfma_f32_16x16x32_f16; every wave owns a 64x64 sub-tile (4x4 fragments, 64 accumulators)
//   epilogue       accumulators -> fp32 tile in LDS -> each thread finishes 8 consecutive channels of one pixel:
//                  +bias, optional residual (optionally nearest-2x upsampled = FPN top-down add), ReLU,
//                  one 16-byte store (full 128/256-byte lines per pixel); or ConvTranspose(k2,s2) pixel shuffle;
//                  or fp32 row-major output
//   scheduling     one workgroup per BM x BN tile; the linear block id is re-dealt so that the 8 XCDs each own a
//                  contiguous run of tiles (neighbouring tiles share halo rows and the weight panel in that L2)
#include <cstdlib>
#include "vtd_common.h"

namespace {

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt immediate");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int V>
struct IntC { static constexpr int value = V; };

// scheduling pins for one half K-step: NLOAD LDS-DMA instructions spread evenly between NM MFMAs
template <int NM, int NLOAD, int K = 0, int PREV = 0>
__device__ __forceinline__ void pin_loads_between_mfmas() {
    if constexpr (K < NLOAD) {
        constexpr int AT = (K + 1) * NM / (NLOAD + 1);
        __builtin_amdgcn_sched_group_barrier(0x008, AT - PREV, 0);  // MFMA
        __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);          // VMEM (the LDS-DMA)
        pin_loads_between_mfmas<NM, NLOAD, K + 1, AT>();
    } else {
        __builtin_amdgcn_sched_group_barrier(0x008, NM - PREV, 0);
    }
}

// DUAL (plain tiles only): the K range has a second segment that gathers a 1x1 window of a SECOND tensor at (oy * in2_mul >> in2_shr,
// ox * in2_mul >> in2_shr) -- a residual branch's 1x1 projection (ResNet downsample: stride 2 -> in2_mul = 2) or a coarser pyramid
// level (nearest-2x up-sampling: in2_shr = 1) folded into this GEMM as extra K-steps: out = W1 * im2col(in) + W2 * in2, one launch, one
// accumulator set, the projected tensor never written.  The walk switches sources at K-step seg1_steps by recomputing the rows' base
// pointers in place (once per tile: no second pointer set stays live in the loop).
template <int BM, int BN, int WM, int WN, int STAGES, bool CLASSED = false, bool DUAL = false>
__global__ __launch_bounds__(WM* WN * 64) void conv_igemm_kernel(const ConvParams p, const int tiles_n) {
    static_assert(!(CLASSED && DUAL), "the classed op has its own second source");
    constexpr int NW = WM * WN;
    constexpr int NT = NW * 64;
    // Tile heights that are not a multiple of 8 * NW or of 16 * WM (208, 272: chosen so that a layer's tile count fills whole rounds
    // of the 256 CUs, see vtd_launch_conv) round the loader up to whole LDS-DMA instructions per wave (every wave issues the same
    // number: the counted waits depend on it; the spare ones fetch the tile's last 8 rows again, same bytes to the same LDS rows)
    // and give the LAST wave row fewer fragments (UNEVEN).
    constexpr int A_INST = (BM + 8 * NW - 1) / (8 * NW);  // global_load_lds instructions per wave per K-step for the A tile
    constexpr int A_PIECES = BM / 8;
    constexpr int A_BYTES = BM * 128;
    constexpr int B_BYTES = BN * 128;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int B_INST = BN / (8 * NW);
    constexpr int LOADS = A_INST + B_INST;
    constexpr int FM = (BM / 16 + WM - 1) / WM, TM = FM * 16, TN = BN / WN;  // fragments per wave row (the last one may have fewer)
    constexpr int FN = TN / 16;
    constexpr bool UNEVEN = (BM / 16) % WM != 0;
    static_assert(BM % 16 == 0 && (BM / 16 - (WM - 1) * FM) >= 1, "tile height");
    constexpr bool HEAD_TILE = BM == 64 && BN == 256;       // configuration 7: the fused DB-head tail needs all 256 columns of a pixel
    constexpr bool DIRECT_ONLY = UNEVEN || BM * (BN * 4 + 16) > 160 * 1024;  // no room (or no code) for the fp32 staging tile
    static_assert(!DIRECT_ONLY || (!CLASSED && !HEAD_TILE), "these tile shapes finish from registers only");
    static_assert(A_INST >= 1 && B_INST >= 1 && (NW == 4 || NW == 8 || NW == 16), "tile/wave shape");
    static_assert(STAGES >= 2 && STAGES <= 6, "pipeline depth");
    constexpr int EPI_ROW = BN * 4 + 16;  // fp32 tile row stride (bytes), padded by one 16-byte slot

    extern __shared__ __attribute__((aligned(16))) char smem[];

    // ---- XCD-aware tile assignment (bijective for any grid size)
    const int nblk = gridDim.x, b = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = b & 7;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Register epilogue (plain NHWC output, bias / residual / ReLU only): decided by the launcher, uniform over the grid
    const bool direct = !CLASSED && !(BM == 64 && BN == 256) && p.epi_direct != 0;

    // ---- loader state: each lane owns one 16-byte chunk slot of A_INST + B_INST rows
    const int lrow = lane >> 3;
    const int c_log = (lane & 7) ^ (((w & 1) << 2) | (lane >> 4));  // logical K-chunk this lane fetches
    const int c_off = (c_log >> 2) * p.k_hi_step + (c_log & 3) * 8; // its element offset inside a K-step
    const int howo = p.ho * p.wo;
    // GEMM row m -> output pixel.  Plain: raster order over (image, oy, ox).  Pooled (pool_log2 = log2 of the window size: a 2 x pool_pw
    // max-pool follows the conv's ReLU and is taken in the epilogue): WINDOW-major -- the rows of one pooling window are consecutive, so
    // they sit in neighbouring lanes of one accumulator fragment whatever the tile shape: m = window index * window size + (dy * pool_pw + dx)
    // (the two divisions are multiplications by ceil(2^40 / d) whenever the launcher found them exact for this launch -- magic_ok: an
    // integer division is ~35 vector instructions on this ISA, and a short-K tile (9 K-steps: the layer-2 entry, CRNN conv2, the 1x1
    // laterals) has only 72 MFMAs per wave to hide its A_INST x 2 of them behind: SQ counters had conv_igemm<128,64,2,2,2> at 6.9 vector
    // instructions per MFMA)
    auto div_rows = [&](int v, int d_howo, int& q, int& r) {
        q = p.magic_ok ? (int)(((uint64_t)(uint32_t)v * p.magic_howo) >> 40) : v / d_howo;
        r = v - q * d_howo;
    };
    auto div_cols = [&](int v, int d_wo, int& q, int& r) {
        q = p.magic_ok ? (int)(((uint64_t)(uint32_t)v * p.magic_wo) >> 40) : v / d_wo;
        r = v - q * d_wo;
    };
    auto row_pixel = [&](int m, int& img, int& oy, int& ox) {
        if (p.pool_log2 == 0) {
            int rem;
            div_rows(m, howo, img, rem);
            div_cols(rem, p.wo, oy, ox);
        } else {
            const int widx = m >> p.pool_log2, sub = m & ((1 << p.pool_log2) - 1);
            int rem, qy, qx;
            div_rows(widx, p.pool_hqwq, img, rem);
            div_cols(rem, p.pool_wq, qy, qx);
            oy = 2 * qy + (p.pool_pw == 2 ? sub >> 1 : sub);
            ox = p.pool_pw == 2 ? 2 * qx + (sub & 1) : qx;
        }
    };
    const half_t* aptr[A_INST];
    const half_t* aptr2[CLASSED ? A_INST : 1];
    // classed mode: a tile is BM entries of the per-image pixel list and carries its own weight class
    const int cl_img = CLASSED ? tm / p.tiles_per_img : 0;
    const int cl_entry = CLASSED ? p.tile_combo[tm - cl_img * p.tiles_per_img] : 0;  // weight class | pixel-list chunk << 8
    const int cl_lt = cl_entry >> 8;
#pragma unroll
    for (int i = 0; i < A_INST; ++i) {
        if constexpr (CLASSED) {
            const uint32_t pk = p.plist[cl_lt * BM + (i * NW + w) * 8 + lrow];
            const int oy = pk == 0xffffffffu ? 0 : (int)(pk & 0xffff), ox = pk == 0xffffffffu ? 0 : (int)(pk >> 16);
            aptr[i] = p.in + ((int64_t)(cl_img * p.in_hp + oy + p.in_y0) * p.in_wp + ox + p.in_x0) * p.in_c + c_off;
            aptr2[i] = p.in2 + ((int64_t)(cl_img * p.in2_hp + (oy >> 1) + p.in2_ring - 1) * p.in2_wp + (ox >> 1) + p.in2_ring - 1) * p.in2_c + c_off;
        } else {
            int piece = i * NW + w;
            piece = piece < A_PIECES ? piece : A_PIECES - 1;  // spare instruction of a tile height that is not a multiple of 8 * NW
            // (its swizzle key must be the one of the LDS rows it lands in: piece parity, not wave parity)
            const int cl_i = (lane & 7) ^ (((piece & 1) << 2) | (lane >> 4));
            const int c_off_i = (cl_i >> 2) * p.k_hi_step + (cl_i & 3) * 8;
            int m = m0 + piece * 8 + lrow;
            m = m < p.M ? m : p.M - 1;
            int img, oy, ox;
            row_pixel(m, img, oy, ox);
            aptr[i] = p.in + ((int64_t)(img * p.in_hp + oy * p.stride + p.in_y0) * p.in_wp + ox * p.stride + p.in_x0) * p.in_c + c_off_i;
        }
    }
    const half_t* wbase = p.wgt;
    if constexpr (CLASSED) wbase += (int64_t)(cl_entry & 0xff) * p.cout_pad * p.K;
    const half_t* bptr[B_INST];
#pragma unroll
    for (int i = 0; i < B_INST; ++i) {
        int brow = (i * NW + w) * 8 + lrow;  // B-tile row in LDS = MFMA row (brow & 15) of fragment block (brow >> 4)
        if (direct) {
            // register epilogue: LDS row `brow` is fed with the weights of another channel of the same wave sub-tile, chosen so
            // that the accumulators of a lane -- rows 4 fq .. 4 fq + 3 of every 16-row block -- are CONSECUTIVE channels:
            // acc[i][.][e] of lane group fq = channel 32 (i >> 1) + 8 fq + 4 (i & 1) + e of the sub-tile
            const int x = brow % TN, blk = x >> 4, r = x & 15;
            brow = brow - x + 32 * (blk >> 1) + 8 * (r >> 2) + 4 * (blk & 1) + (r & 3);
        }
        bptr[i] = wbase + (int64_t)(n0 + brow) * p.K + c_log * 8;
    }

    // K walk state (scalar): kb = element offset of the current K-step from the tap-(0,0) pixel
    int kb = 0, t_c = 0, t_s = 0;
    int w_cin = p.cin_steps, w_kw = p.kw, w_s = p.s_step, w_r = p.r_step;
    bool second = false;  // classed mode: walking the second source
    // One LDS-DMA instruction (1 KB) of K-step `ks` into buffer `buf`: j < A_INST feeds the A tile, the rest the B tile.
    auto issue_load = [&](int j, int ks, int buf) {
        char* abase = smem + buf * STAGE;
#ifdef VTD_CONV_EXPERIMENT  // timing-only variants (tools/conv_experiment.sh): results are garbage, durations are the point
        if (p.dbg >= 3) return;
        if (p.dbg == 2 && j < A_INST) return;
#endif
        if (j < A_INST) {
#ifdef VTD_CONV_EXPERIMENT
            const half_t* src = (second ? aptr2[CLASSED ? j : 0] : aptr[j]) + (p.dbg == 1 ? 0 : kb);
#else
            const half_t* src = (second ? aptr2[CLASSED ? j : 0] : aptr[j]) + kb;
#endif
            const int piece = j * NW + w < A_PIECES ? j * NW + w : A_PIECES - 1;
            __builtin_amdgcn_global_load_lds((const VTD_AS1 void*)src, (VTD_AS3 void*)(abase + piece * 1024), 16, 0, 0);
        } else {
            const int i = j - A_INST;
            __builtin_amdgcn_global_load_lds((const VTD_AS1 void*)(bptr[i] + ks * 64), (VTD_AS3 void*)(abase + A_BYTES + (i * NW + w) * 1024), 16, 0, 0);
        }
    };
    auto begin_step = [&](int ks) {  // before the first load of K-step ks
        if (CLASSED && ks == p.seg1_steps) {  // switch the walk to the second source's 3x3 window
            kb = 0; t_c = 0; t_s = 0; second = true;
            w_cin = p.cin_steps2; w_kw = p.kw2; w_s = p.s_step2; w_r = p.r_step2;
        }
        if constexpr (DUAL) {
            if (ks == p.seg1_steps) {  // every load of the first segment has been issued: re-aim the row pointers at the second tensor
                kb = 0; t_c = 0; t_s = 0;
                w_cin = p.cin_steps2; w_kw = p.kw2; w_s = p.s_step2; w_r = p.r_step2;
#pragma unroll
                for (int i = 0; i < A_INST; ++i) {
                    int piece = i * NW + w;
                    piece = piece < A_PIECES ? piece : A_PIECES - 1;
                    const int cl_i = (lane & 7) ^ (((piece & 1) << 2) | (lane >> 4));
                    const int c_off_i = (cl_i >> 2) * p.k_hi_step + (cl_i & 3) * 8;
                    int m = m0 + piece * 8 + lrow;
                    m = m < p.M ? m : p.M - 1;
                    int img, oy, ox;
                    row_pixel(m, img, oy, ox);
                    aptr[i] = p.in2 + ((int64_t)(img * p.in2_hp + ((oy * p.in2_mul) >> p.in2_shr) + p.in2_ring) * p.in2_wp +
                                       ((ox * p.in2_mul) >> p.in2_shr) + p.in2_ring) * p.in2_c + c_off_i;
                }
            }
        }
    };
    auto end_step = [&]() {  // after the last load of a K-step: advance the walk (steps are always issued in K order)
        kb += 64;
        if (++t_c == w_cin) {
            t_c = 0;
            kb += w_s - w_cin * 64;
            if (++t_s == w_kw) {
                t_s = 0;
                kb += w_r - w_kw * w_s;
            }
        }
    };
    auto stage = [&](int ks, int buf) {
        begin_step(ks);
#pragma unroll
        for (int j = 0; j < LOADS; ++j) issue_load(j, ks, buf);
        end_step();
    };

    // ---- compute state
    const int wm = w / WN, wn = w - wm * WN;
    const int frow = lane & 15;
    const int swz = (lane >> 1) & 7;  // (row>>1)&7 for every fragment row this lane reads
    const int a_lane_off = (wm * TM + frow) * 128;
    // UNEVEN: the last wave row owns FM - 1 fragments (static_assert below): only fragment FM - 1 is conditional, on one wave-uniform flag
    constexpr int LAST_FM = BM / 16 - (WM - 1) * FM;
    static_assert(!UNEVEN || LAST_FM == FM - 1, "uneven wave rows differ by exactly one fragment");
    const bool full_row = !UNEVEN || wm != WM - 1;
    const int b_lane_off = A_BYTES + (wn * TN + frow) * 128;

    floatx4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    // ---- main loop.  Software pipeline over HALF K-steps (32 deep), fragments double-buffered in registers:
    //
    //   iteration s:   read fragments (s, half 1) -> f[1]  |  MFMAs of (s, half 0) out of f[0]   + LDS-DMA
    //                  wait: K-step s+1 landed; barrier B(s+1)
    //                  read fragments (s+1, half 0) -> f[0] |  MFMAs of (s, half 1) out of f[1]  + LDS-DMA
    //
    // Every ds_read is issued one half-step before its MFMAs, so the LDS latency hides behind the matrix pipe of the SAME wave.
    // (Round 1 read a half-step's 8 fragments and then multiplied: all waves of a workgroup leave the barrier together, read
    // together and multiply together, so nobody covered anybody's ~250 read cycles: with every load removed the launches still
    // took twice their MFMA time, tools/conv_experiment.sh.)  The barrier sits in the MIDDLE of a K-step: B(s+1) publishes K-step
    // s+1 and, since every wave has by then finished reading both halves of K-step s, frees buffer s % STAGES for K-step
    // s + STAGES.  With 3 stages those loads are spread over the MFMAs of the two half-steps after the barrier and have until
    // B(s+3); with 2 stages they all leave right after the barrier and have one K-step, as before.
    auto read_frags = [&](const char* sb, int kk, half8 (&af)[FM], half8 (&bf)[FN]) {
#ifdef VTD_CONV_EXPERIMENT
        if (p.dbg == 6 && sb != smem) return;
#endif
        const int phys = (((lane >> 4) + 4 * kk) ^ swz) * 16;
#pragma unroll
        for (int j = 0; j < FM; ++j) {
            if (UNEVEN && j == FM - 1 && !full_row) { af[j] = af[0]; continue; }  // (keeps the register defined; its MFMAs are skipped)
            af[j] = *(const half8*)(sb + a_lane_off + j * 2048 + phys);
        }
#pragma unroll
        for (int i = 0; i < FN; ++i) bf[i] = *(const half8*)(sb + b_lane_off + i * 2048 + phys);
    };
    // the MFMAs of one half K-step out of (af, bf); `nload` LDS-DMA instructions [j0, j0 + nload) of K-step `lks` are issued BETWEEN
    // them: a global_load_lds costs the wave ~60-100 issue cycles, which hide behind the matrix pipe only if they are spread over it
    auto mfma_half = [&](const half8 (&af)[FM], const half8 (&bf)[FN], auto nload_c, int lks, int lbuf, int j0) {
        constexpr int nload = decltype(nload_c)::value;
        constexpr int NM = FM * FN;
        int q = 0, issued = 0;
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int j = 0; j < FM; ++j) {
                if (!UNEVEN || j < FM - 1 || full_row) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[i], af[j], acc[i][j], 0, 0, 0);
                ++q;
                if (issued < nload && q >= (issued + 1) * NM / (nload + 1)) {
                    issue_load(j0 + issued, lks, lbuf);
                    ++issued;
                }
            }
    };

    const int nk = p.K >> 6;
    constexpr bool PIPE = FM * FN < 32;  // 8 x 4 fragment tiles (256 x 256 on 8 waves): 128 accumulator registers leave no room for a second fragment set
    if constexpr (!PIPE) {
        // plain two-buffer loop: wait, barrier, next K-step's loads leave, then two half-steps of read-then-multiply (the partner wave
        // on the SIMD covers the read latency); what this shape buys is pieces per MFMA, not issue overlap
        static_assert(STAGES == 2 && !UNEVEN, "big register tiles: two LDS stages");
        half8 af[FM], bf[FN];
        stage(0, 0);
        int buf = 0;
        for (int ks = 0; ks < nk; ++ks) {
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();  // K-step ks landed for every wave; everyone finished reading the other buffer
            if (ks + 1 < nk) stage(ks + 1, buf ^ 1);
            const char* sb = smem + buf * STAGE;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                read_frags(sb, kk, af, bf);
                mfma_half(af, bf, IntC<0>{}, 0, 0, 0);
            }
            buf ^= 1;
        }
    } else {
        constexpr int L0 = (LOADS + 1) / 2, L1 = LOADS - L0;
        // D = K-steps a load is issued ahead of its use.  2 stages: 1 (all loads of K-step s+2 leave right after the barrier that frees their
        // buffer).  3 and more: STAGES - 1, spread over the MFMAs (second part of K-step s+D under half 0, first part of K-step s+D+1 under
        // half 1).  Deep rings (6 stages of the 128 x 64 tile = 144 KB) are for launches with a handful of workgroups and a long K, where a
        // K-step is a memory round trip and nothing else (the TrOCR decoder's GEMMs at 272 rows: 48 workgroups, 16 K-steps).
        constexpr int D = STAGES >= 3 ? STAGES - 1 : 1;
        constexpr int P2 = STAGES >= 3 ? L1 : 0;
        constexpr int P4 = STAGES >= 3 ? L0 : LOADS;
        static_assert((D - 1) * LOADS + L0 < 64, "vmcnt immediate");
        half8 fa[2][FM], fb[2][FN];

        // prologue: K-steps 0 .. D-1 entirely (2 stages: 0 and 1), with 3 and more stages also the first part of K-step D
        if constexpr (STAGES >= 3) {
    #pragma unroll
            for (int a = 0; a < D; ++a)
                if (a < nk) stage(a, a);
            if (nk > D) {
                begin_step(D);
    #pragma unroll
                for (int j = 0; j < L0; ++j) issue_load(j, D, D);
                wait_vmcnt<(D - 1) * LOADS + L0>();
            } else if (STAGES == 3 && nk > 1) {
                wait_vmcnt<LOADS>();
            } else {
                wait_vmcnt<0>();
            }
        } else {
            stage(0, 0);
            if (nk > 1) { stage(1, 1); wait_vmcnt<LOADS>(); } else wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();
        read_frags(smem, 0, fa[0], fb[0]);

        // one iteration; HAS2 / HAS4: the loads of part 2 / part 4 exist (compile-time, so the load / MFMA interleave stays pinned)
        auto iteration = [&](int s, int buf, auto has2_c, auto has4_c) {
            constexpr bool HAS2 = decltype(has2_c)::value != 0, HAS4 = decltype(has4_c)::value != 0;
            const char* sb = smem + buf * STAGE;
            int b1 = buf + 1; b1 = b1 >= STAGES ? b1 - STAGES : b1;   // buffer of K-step s+1
            const int bD = buf == 0 ? STAGES - 1 : buf - 1;         // buffer of K-step s+D (3 and more stages)
            read_frags(sb, 1, fa[1], fb[1]);
            if constexpr (HAS2 && P2 > 0) {
                mfma_half(fa[0], fb[0], IntC<P2>{}, s + D, bD, L0);
                end_step();
                __builtin_amdgcn_sched_group_barrier(0x100, FM + FN, 0);  // the fragment reads first,
                pin_loads_between_mfmas<FM * FN, P2>();                    // then MFMA groups with one LDS-DMA after each
            } else {
                mfma_half(fa[0], fb[0], IntC<0>{}, 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, FM + FN, 0);
                pin_loads_between_mfmas<FM * FN, 0>();
            }
            if (s + 1 >= nk) {  // last K-step: nothing to publish or prefetch
                mfma_half(fa[1], fb[1], IntC<0>{}, 0, 0, 0);
                return;
            }
            // K-step s+1 has landed when only the loads of K-steps s+2 .. s+D (all issued by now) are outstanding
            if constexpr (HAS2 && STAGES >= 3) wait_vmcnt<(D - 1) * LOADS>(); else wait_vmcnt<0>();
            // This wave's reads of K-step s must be complete before the barrier (the buffer is refilled behind it).  Asking for the
            // registers makes hipcc place that wait HERE, where nothing younger is outstanding; left alone it waits for them in front of
            // the MFMAs below as lgkmcnt(0), which would also sit out the fragment reads issued just before -- the prefetch would be lost.
    #pragma unroll
            for (int j = 0; j < FM; ++j) asm volatile("" ::"v"(fa[1][j]));
    #pragma unroll
            for (int i = 0; i < FN; ++i) asm volatile("" ::"v"(fb[1][i]));
    #ifdef VTD_CONV_EXPERIMENT
            if (p.dbg != 4)
    #endif
            __builtin_amdgcn_s_barrier();
            if constexpr (HAS4 && STAGES == 2) stage(s + 2, buf);  // one K-step of slack only: the loads leave first
            read_frags(smem + b1 * STAGE, 0, fa[0], fb[0]);
            if constexpr (HAS4 && STAGES >= 3) {
                begin_step(s + D + 1);
                mfma_half(fa[1], fb[1], IntC<P4>{}, s + D + 1, buf, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, FM + FN, 0);
                pin_loads_between_mfmas<FM * FN, P4>();
            } else {
                mfma_half(fa[1], fb[1], IntC<0>{}, 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, FM + FN, 0);
                pin_loads_between_mfmas<FM * FN, 0>();
            }
        };

        int buf = 0, s = 0;
        // steady state: both load groups exist (3 stages: K-steps s+2 and s+3; 2 stages: K-step s+2)
        for (; s + STAGES < nk; ++s) {
            iteration(s, buf, IntC<1>{}, IntC<1>{});
            buf = buf + 1 == STAGES ? 0 : buf + 1;
        }
        if (STAGES >= 3 && s + D < nk) {  // K-step s+D is the last one: its second part still has to leave
            iteration(s, buf, IntC<1>{}, IntC<0>{});
            buf = buf + 1 == STAGES ? 0 : buf + 1;
            ++s;
        }
        for (; s < nk; ++s) {  // drain: nothing left to fetch
            iteration(s, buf, IntC<0>{}, IntC<0>{});
            buf = buf + 1 == STAGES ? 0 : buf + 1;
        }
    }

    // ---- register epilogue.  The LDS round trip below (fp32 tile out, 8-channel rows back in, two workgroup barriers, a pixel
    // table) was 20-37 % of these launches (tools/conv_experiment.sh, VTD_CONV_DEBUG=5).  With the weight rows permuted as above a
    // lane already holds 8 consecutive channels of each of its FM pixels per pair of fragment blocks: (acc + bias) + residual, ReLU,
    // one fp16 rounding -- the same operations in the same order as below, so both paths give identical bits -- and one 16-byte
    // store; the four lane groups of a pixel write 64 contiguous bytes.  No barrier: a wave leaves as soon as its own MFMAs are done.
    if constexpr (!CLASSED && !HEAD_TILE && (FN % 2) == 0) {
        if (direct) {
            const int fq = lane >> 4;
            int64_t o_off[FM], r_off[FM];
            bool ok[FM];
#pragma unroll
            for (int j = 0; j < FM; ++j) {
                int m = m0 + wm * TM + j * 16 + frow;
                ok[j] = m < p.M && (!UNEVEN || j < FM - 1 || full_row);
                m = ok[j] ? m : p.M - 1;
                // pooled: the lane that holds a window's first row stores the window's maximum at the POOLED pixel (magic_* and the output
                // tensor are those of the pooled map: vtd_launch_conv)
                if (p.pool_log2) { ok[j] = ok[j] && (frow & ((1 << p.pool_log2) - 1)) == 0; m >>= p.pool_log2; }
                const int e_howo = p.pool_log2 ? p.pool_hqwq : howo, e_wo = p.pool_log2 ? p.pool_wq : p.wo;
                const int img = (int)(((uint64_t)(uint32_t)m * p.magic_howo) >> 40);
                const int rem = m - img * e_howo;
                const int oy = (int)(((uint64_t)(uint32_t)rem * p.magic_wo) >> 40), ox = rem - oy * e_wo;
                o_off[j] = ((int64_t)(img * p.out_hp + oy + p.out_ring) * p.out_wp + ox + p.out_ring) * p.out_c;
                r_off[j] = ((int64_t)(img * p.res_hp + (oy >> p.res_shift) + p.res_ring) * p.res_wp + (ox >> p.res_shift) + p.res_ring) * p.cout;
            }
            const bool has_res = (p.flags & EPI_RESIDUAL) != 0, relu = (p.flags & EPI_RELU) != 0;
#pragma unroll
            for (int k = 0; k < FN / 2; ++k) {
                const int ch = n0 + wn * TN + 32 * k + 8 * fq;
                if (ch >= p.cout) continue;  // cout % 8 == 0: a chunk is valid as a whole
                half8 resv[FM];
                if (has_res) {
#pragma unroll
                    for (int j = 0; j < FM; ++j) resv[j] = *(const half8*)(p.res + r_off[j] + ch);
                }
                const floatx4 b0 = *(const floatx4*)(p.bias + ch), b1 = *(const floatx4*)(p.bias + ch + 4);
#pragma unroll
                for (int j = 0; j < FM; ++j) {
                    floatx4 v0 = acc[2 * k][j] + b0, v1 = acc[2 * k + 1][j] + b1;
                    half8 hv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float a0 = v0[e], a1 = v1[e];
                        if (has_res) { a0 += (float)resv[j][e]; a1 += (float)resv[j][4 + e]; }
                        if (relu) { a0 = a0 > 0.f ? a0 : 0.f; a1 = a1 > 0.f ? a1 : 0.f; }
                        if (p.pool_log2) {
                            // max over the window's rows = lanes frow ^ 1 (and frow ^ 2): quad-permute DPP, no LDS.  Rounding to fp16 is
                            // monotone, so max-then-round equals the separate pool kernel's round-then-max bit for bit
                            a0 = fmaxf(a0, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, a0), 0xB1, 0xf, 0xf, true)));
                            a1 = fmaxf(a1, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, a1), 0xB1, 0xf, 0xf, true)));
                            if (p.pool_log2 == 2) {
                                a0 = fmaxf(a0, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, a0), 0x4E, 0xf, 0xf, true)));
                                a1 = fmaxf(a1, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, a1), 0x4E, 0xf, 0xf, true)));
                            }
                        }
                        hv[e] = (half_t)a0;
                        hv[4 + e] = (half_t)a1;
                    }
#ifdef VTD_CONV_EXPERIMENT
                    if (p.dbg == 7) { asm volatile("" ::"v"(hv)); continue; }  // everything but the stores
#endif
                    if (ok[j]) *(half8*)((half_t*)p.out + o_off[j] + ch) = hv;
                }
            }
#ifdef VTD_CONV_EXPERIMENT
            if (p.dbg == 9) wait_vmcnt<0>();  // does a wave that ends with stores in flight hold its slot anyway?
#endif
            return;
        }
    }

    if constexpr (DIRECT_ONLY) return;  // (unreachable: the launcher admits these tile shapes only with the register epilogue)
#ifdef VTD_CONV_EXPERIMENT
    if (p.dbg == 5) {  // no epilogue: keep the accumulators alive with one store per wave
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int j = 0; j < FM; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (t == 123.456f) ((float*)p.out)[0] = t;
        return;
    }
#endif
    // ---- epilogue phase 1: accumulators -> fp32 tile in LDS (weights were the A operand: a lane holds 4
    //      consecutive channels (lane>>4)*4.. of pixel lane&15), plus the pixel coordinates of the tile rows
    __syncthreads();  // all waves are done with the staging buffers
    int* pix = (int*)(smem + BM * EPI_ROW);  // [BM][2]: image (or -1 past M), oy | ox << 16
    if (tid < BM) {
        int img = -1, oy = 0, ox = 0;
        if constexpr (CLASSED) {
            const uint32_t pk = p.plist[cl_lt * BM + tid];
            if (pk != 0xffffffffu) { img = cl_img; oy = (int)(pk & 0xffff); ox = (int)(pk >> 16); }
        } else {
            const int m = m0 + tid;
            if (m < p.M) {
                img = m / howo;
                const int rem = m - img * howo;
                oy = rem / p.wo;
                ox = rem - oy * p.wo;
            }
        }
        pix[2 * tid] = img;
        pix[2 * tid + 1] = oy | (ox << 16);
    }
    {
        const int chq = (lane >> 4) * 4;
#pragma unroll
        for (int j = 0; j < FM; ++j)
#pragma unroll
            for (int i = 0; i < FN; ++i)
                *(floatx4*)(smem + (wm * TM + j * 16 + frow) * EPI_ROW + (wn * TN + i * 16 + chq) * 4) = acc[i][j];
    }
    __syncthreads();

    if constexpr (HEAD_TILE) {
        // ---- fused DB-head tail (EPI_HEAD_FINAL): one thread = one (input pixel, 2x2 block) = 64 channels of the
        // ConvT1 output -> BN/ReLU -> ConvT(64->1) -> sigmoid -> a 2x2 patch of the probability map.
        // A quad of lanes shares one (pixel, block): lane q of the quad owns channels 16q..16q+15, keeps the matching
        // 4x16 slice of the final weights in registers and the quad combines its partial sums with two shuffles.
        float* hb = (float*)(smem + BM * EPI_ROW + BM * 8);  // [256] bias of this GEMM (BN folded)
        for (int i = tid; i < 256; i += NT) hb[i] = p.bias[i];
        const int qd = tid & 3;
        float w6[4][16];
#pragma unroll
        for (int o4 = 0; o4 < 4; ++o4)
#pragma unroll
            for (int c = 0; c < 16; ++c) w6[o4][c] = p.head_w[o4 * 64 + qd * 16 + c];
        __syncthreads();
        for (int item = tid >> 2; item < BM * 4; item += NT / 4) {
            const int row = item >> 2, blk = item & 3;
            const int img = pix[2 * row];
            const int yx = pix[2 * row + 1];
            const int oy = yx & 0xffff, ox = yx >> 16;
            float o[4] = {0.f, 0.f, 0.f, 0.f};
            const char* src = smem + row * EPI_ROW + blk * 256 + qd * 64;
            const float* bsrc = hb + blk * 64 + qd * 16;
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                const floatx4 v = *(const floatx4*)(src + c4 * 16);
                const floatx4 bv = *(const floatx4*)(bsrc + c4 * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float a = v[e] + bv[e];
                    a = a > 0.f ? a : 0.f;
#pragma unroll
                    for (int o4 = 0; o4 < 4; ++o4) o[o4] += a * w6[o4][c4 * 4 + e];
                }
            }
#pragma unroll
            for (int o4 = 0; o4 < 4; ++o4) {
                o[o4] += __shfl_xor(o[o4], 1);
                o[o4] += __shfl_xor(o[o4], 2);
            }
            if (img < 0 || qd != 0) continue;
            const int W4 = 4 * p.wo;
            float* dst = p.prob_out + ((int64_t)img * 4 * p.ho + 4 * oy + 2 * (blk >> 1)) * W4 + 4 * ox + 2 * (blk & 1);
            *(float2*)dst = make_float2(1.f / (1.f + expf(-(o[0] + p.head_b))), 1.f / (1.f + expf(-(o[1] + p.head_b))));
            *(float2*)(dst + W4) = make_float2(1.f / (1.f + expf(-(o[2] + p.head_b))), 1.f / (1.f + expf(-(o[3] + p.head_b))));
        }
        return;
    }

    // ---- epilogue phase 2: one thread = 8 consecutive channels of one pixel
    constexpr int CPR = BN / 8;  // 16-byte output chunks per tile row
    static_assert(NT % CPR == 0, "a thread keeps its channel chunk across iterations");
    const int cc = tid % CPR;
    const int ch = n0 + cc * 8;
    if (ch < p.cout) {
        float bias[8];
        {
            const float4 b0 = *(const float4*)(p.bias + ch), b1 = *(const float4*)(p.bias + ch + 4);
            bias[0] = b0.x; bias[1] = b0.y; bias[2] = b0.z; bias[3] = b0.w;
            bias[4] = b1.x; bias[5] = b1.y; bias[6] = b1.z; bias[7] = b1.w;
        }
        int oc = ch, dy = 0, dx = 0;
        if (p.flags & EPI_PIXEL_SHUFFLE) {
            const int blk = ch / p.ps_cout;
            oc = ch - blk * p.ps_cout;
            dy = blk >> 1;
            dx = blk & 1;
        }
        constexpr int ROWS_PER_IT = NT / CPR;
        constexpr int ITERS = BM / ROWS_PER_IT;
        // issue every residual read of this thread up front: one round trip instead of one per row
        half8 resv[ITERS];
        if (p.flags & EPI_RESIDUAL) {
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int row = tid / CPR + it * ROWS_PER_IT;
                const int img = pix[2 * row];
                const int yx = pix[2 * row + 1];
                const int oy = yx & 0xffff, ox = yx >> 16;
                const int64_t ro = ((int64_t)((img < 0 ? 0 : img) * p.res_hp + (oy >> p.res_shift) + p.res_ring) * p.res_wp +
                                    (ox >> p.res_shift) + p.res_ring) * p.cout + ch;
                resv[it] = *(const half8*)(p.res + ro);
            }
        }
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int row = tid / CPR + it * ROWS_PER_IT;
            const int img = pix[2 * row];
            if (img < 0) continue;
            const int yx = pix[2 * row + 1];
            const int oy = yx & 0xffff, ox = yx >> 16;
            const floatx4 v0 = *(const floatx4*)(smem + row * EPI_ROW + cc * 32);
            const floatx4 v1 = *(const floatx4*)(smem + row * EPI_ROW + cc * 32 + 16);
            float v[8] = {v0[0] + bias[0], v0[1] + bias[1], v0[2] + bias[2], v0[3] + bias[3],
                          v1[0] + bias[4], v1[1] + bias[5], v1[2] + bias[6], v1[3] + bias[7]};
            if constexpr (CLASSED) {  // border-class bias (which taps of the composed window fall inside the image)
                const int yc = oy == 0 ? 0 : oy == 1 ? 1 : oy == p.img_h - 2 ? 3 : oy == p.img_h - 1 ? 4 : 2;
                const int xc = ox == 0 ? 0 : ox == 1 ? 1 : ox == p.img_w - 2 ? 3 : ox == p.img_w - 1 ? 4 : 2;
                const float* bt = p.bias_tab + (yc * 5 + xc) * p.cout + ch;
                const float4 b0 = *(const float4*)bt, b1 = *(const float4*)(bt + 4);
                v[0] = v0[0] + b0.x; v[1] = v0[1] + b0.y; v[2] = v0[2] + b0.z; v[3] = v0[3] + b0.w;
                v[4] = v1[0] + b1.x; v[5] = v1[1] + b1.y; v[6] = v1[2] + b1.z; v[7] = v1[3] + b1.w;
            }
            if (p.flags & EPI_RESIDUAL) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += (float)resv[it][e];
            }
            if (p.flags & EPI_RELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
            }
            if (p.flags & EPI_GELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752f));
            }
            if (p.flags & EPI_OUT_F16) {
                half8 hv;
#pragma unroll
                for (int e = 0; e < 8; ++e) hv[e] = (half_t)v[e];
                if (p.seg_cols > 0) {
                    const int sg = ch / p.seg_cols;
                    *(half8*)((half_t*)p.seg_out[sg] + (int64_t)(m0 + row) * p.seg_ldc[sg] + (ch - sg * p.seg_cols)) = hv;
                } else {
                    *(half8*)((half_t*)p.out + (int64_t)(m0 + row) * p.ldc + ch) = hv;
                }
            } else if (p.flags & EPI_OUT_F32) {
                float* o = (float*)p.out + (int64_t)(m0 + row) * p.ldc + ch;
                if (ch + 7 < p.cout) {
                    *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
                    *(float4*)(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
                } else {
                    for (int e = 0; e < 8 && ch + e < p.cout; ++e) o[e] = v[e];
                }
            } else {
                const int py = (p.flags & EPI_PIXEL_SHUFFLE) ? 2 * oy + dy : oy;
                const int px = (p.flags & EPI_PIXEL_SHUFFLE) ? 2 * ox + dx : ox;
                const int64_t oo = ((int64_t)(img * p.out_hp + py + p.out_ring) * p.out_wp + px + p.out_ring) * p.out_c + oc;
                half8 hv;
#pragma unroll
                for (int e = 0; e < 8; ++e) hv[e] = (half_t)v[e];
                *(half8*)((half_t*)p.out + oo) = hv;
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int STAGES, bool CLASSED = false, bool DUAL = false>
int launch_cfg(const ConvParams& p, hipStream_t stream) {
    const int tiles_m = CLASSED ? p.M / BM : (p.M + BM - 1) / BM;  // (classed: vtd_launch_conv set M for this tile height)
    const int tiles_n = p.cout_pad / BN;
    constexpr int stage_bytes = STAGES * (BM + BN) * 128;
    constexpr bool direct_only = (BM / 16) % WM != 0 || BM * (BN * 4 + 16) > 160 * 1024;  // register epilogue only: no fp32 staging tile
    constexpr int epi_bytes = direct_only ? 0 : BM * (BN * 4 + 16) + BM * 8 + (BN == 256 ? 1024 : 0);
    constexpr int lds = stage_bytes > epi_bytes ? stage_bytes : epi_bytes;
    static_assert(lds <= 160 * 1024, "LDS budget");
    static bool attr_done = false;
    if (!attr_done) {
        VTD_HIP_CHECK(hipFuncSetAttribute((const void*)conv_igemm_kernel<BM, BN, WM, WN, STAGES, CLASSED, DUAL>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_done = true;
    }
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, STAGES, CLASSED, DUAL>), dim3(tiles_m * tiles_n), dim3(WM * WN * 64), lds, stream, p, tiles_n);
    return -(int)hipGetLastError();
}

// plain tile shape, with or without a second K segment (ConvParams::in2 without a pixel list)
template <int BM, int BN, int WM, int WN, int STAGES>
int launch_plain(const ConvParams& p, hipStream_t stream) {
    return p.in2 ? launch_cfg<BM, BN, WM, WN, STAGES, false, true>(p, stream) : launch_cfg<BM, BN, WM, WN, STAGES>(p, stream);
}

}  // namespace

// ---- tile configurations.  Which one wins depends on the layer (M, N, K, how memory-bound it is), so the network
// graphs time the valid candidates once per batch size (vtd_api.cpp: autotune) instead of guessing.
//   id  tile      waves stages  LDS      blocks/CU
//   0   256x128   8     3       144 KB   1      K-heavy, N >= 128: least L2->LDS traffic per FLOP
//   1   128x128   4     2        68 KB   2
//   2   128x128   4     3        96 KB   1
//   3   256x64    4     2        80 KB   2
//   4   256x64    4     3       120 KB   1
//   5   128x64    4     2        48 KB   3      short-K / store-bound layers: more blocks in flight per CU
//   6   128x64    4     3        72 KB   2
//   7   64x256    4     2        80 KB   2      only for the fused DB-head tail (EPI_HEAD_FINAL needs all 256 columns)
//   8/9   128x64  4     2/3               only for the classed dual-source op (pixel-list tiles of 128 rows)
//   10/11 256x64  4     2/3               same op, pixel lists cut into 256-row tiles: 17 % fewer LDS-DMA pieces per FLOP
//   12  208x128   8 (2x4) 3     126 KB   1      tile heights that fill whole rounds of the 256 CUs where 256 rows do not: 51 200 rows
//   13  272x128   8 (2x4) 3     150 KB   1      x 256 channels = 494 tiles of 208 (two rounds at 0.81 of the 256-row tile's time each)
//                                               instead of 400 of 256 (two rounds for 1.56); 69 632 rows = exactly 256 tiles of 272 per
//                                               128 channels.  Wave rows of 7 + 6 / 9 + 8 fragments; register epilogue only
//   14  256x128   16 (4x4) 3    144 KB   1      the 256 x 128 tile on FOUR waves per SIMD (64 x 32 per wave, 3 LDS-DMA instructions per wave
//                                               and K-step instead of 6).  Measured equal to configuration 0 on every layer (tools/gpu_tiles.sh),
//                                               as was a 208-row tile with 12 % fewer MFMAs: a K-step of this kernel costs what its 48 one-KB
//                                               LDS-DMA instructions per CU cost (~40 cycles each), whoever issues them
//   15  256x256   8 (2x4) 2     128 KB   1      64 LDS-DMA pieces per K-step for twice the MFMAs of the 256 x 128 tile (32 per 256 x 128
//                                               equivalent instead of 48): for launches with enough 256 x 256 tiles.  Register epilogue only
//   16  128x64    4     6       144 KB   1      six LDS stages: five K-steps of loads in flight, for launches of a few workgroups with a long K
int vtd_conv_num_configs() { return 17; }

// plain NHWC fp16 output with bias / residual / ReLU only: what the register epilogue (and so configurations 12 / 13) can finish
static bool epi_direct_eligible(const ConvParams& p) {
    const uint64_t howo = (uint64_t)p.ho * (uint64_t)p.wo;
    return !p.plist && !(p.flags & ~(EPI_RELU | EPI_RESIDUAL)) && !(p.cout & 7) && (uint64_t)p.M * howo < (1ull << 40) && (!(p.flags & EPI_RESIDUAL) || p.res);
}

bool vtd_conv_config_valid(const ConvParams& p, int cfg) {
    if (p.flags & EPI_HEAD_FINAL) return cfg == 7;
    if (p.plist) return ((cfg == 8 || cfg == 9) || ((cfg == 10 || cfg == 11) && p.plist_b && p.tile_combo_b && p.tiles_per_img_b > 0)) && p.cout_pad % 64 == 0;
    if (p.in2 && (cfg == 3 || cfg == 4 || cfg == 15 || cfg == 16)) return false;  // a second K segment: the tile shapes instantiated for it
    switch (cfg) {
        case 0: case 1: case 2: case 14: return p.cout_pad % 128 == 0;
        case 3: case 4: case 5: case 6: case 16: return p.cout_pad % 64 == 0;
        case 12: case 13: case 15: {
            const char* e = getenv("VTD_EPI_DIRECT");
            return p.cout_pad % (cfg == 15 ? 256 : 128) == 0 && epi_direct_eligible(p) && !(e && e[0] == '0');
        }
        default: return false;
    }
}

int vtd_conv_default_config(const ConvParams& p) {
    if (p.flags & EPI_HEAD_FINAL) return 7;
    if (p.plist) return 9;
    if (p.cout_pad % 128 == 0) {
        const int64_t big_tiles = (int64_t)((p.M + 255) / 256) * (p.cout_pad / 128);
        return (big_tiles >= 512 && p.K >= 512) ? 0 : 1;
    }
    return p.K >= 1024 ? 3 : 5;
}

// Host entry used by the network graphs in vtd_api.cpp.  Shapes are validated here: a mismatch must
// never reach the kernel (an out-of-bounds gather can take the whole node down).
int vtd_launch_conv(const ConvParams& p_in, int cfg, hipStream_t stream) {
    static const int dbg = [] { const char* e = getenv("VTD_CONV_DEBUG"); return e ? atoi(e) : 0; }();
    ConvParams p = p_in;
    p.dbg = dbg;
    if (p.M <= 0 || p.K <= 0 || (p.K & 63) || p.cout <= 0 || (p.cout_pad & 63) || p.cout > p.cout_pad) return -1001;
    if ((p.cout & 7) && !(p.flags & EPI_OUT_F32)) return -1002;
    if ((p.flags & EPI_OUT_F32) && (p.ldc & 3)) return -1003;
    if ((p.flags & EPI_OUT_F16) && ((p.ldc & 7) || (p.cout & 7))) return -1003;
    if (p.seg_cols) {
        if (!(p.flags & EPI_OUT_F16) || p.seg_cols < 256 || (p.seg_cols & 255) || p.cout % p.seg_cols || p.cout / p.seg_cols > 3) return -1010;
        for (int i = 0; i < p.cout / p.seg_cols; ++i)
            if (!p.seg_out[i] || (p.seg_ldc[i] & 7) || p.seg_ldc[i] < p.seg_cols) return -1010;
    }
    if ((p.flags & EPI_PIXEL_SHUFFLE) && (p.ps_cout & 7)) return -1004;
    if (p.ho >= 65536 || p.wo >= 32768) return -1005;
    if ((p.flags & EPI_HEAD_FINAL) && (p.cout != 256 || p.cout_pad != 256 || !p.head_w || !p.prob_out)) return -1008;
    if (p.plist && (!p.tile_combo || !p.in2 || !p.bias_tab || p.tiles_per_img <= 0 || p.M % 128 || p.seg1_steps <= 0 ||
                    p.cin_steps2 <= 0 || p.kw2 <= 0 || p.stride != 1 || (p.flags & ~EPI_RELU))) return -1009;
    if (!p.plist && !p.in2 && (p.cin_steps <= 0 || p.kw <= 0 || p.K != p.cin_steps * 64 * p.kw * (p.K / (p.cin_steps * 64 * p.kw)))) return -1006;
    if (!p.plist && p.in2) {  // second K segment: whole taps of the first window, then a 1x1 window of in2 scaled by in2_mul >> in2_shr
        const int k1 = p.seg1_steps * 64;
        if (p.cin_steps <= 0 || p.kw <= 0 || k1 <= 0 || k1 % (p.cin_steps * 64 * p.kw) || p.cin_steps2 <= 0 || p.kw2 != 1 ||
            p.K != k1 + p.cin_steps2 * 64 || p.in2_c < p.cin_steps2 * 64 || p.in2_mul < 1 || p.in2_mul > 2 || p.in2_shr < 0 || p.in2_shr > 1 ||
            p.in2_ring < 0 || p.in2_hp <= 0 || p.in2_wp <= 0)
            return -1011;
        // the farthest pixel the second segment touches lies inside the padded second tensor
        if ((((p.ho - 1) * p.in2_mul) >> p.in2_shr) + p.in2_ring >= p.in2_hp || (((p.wo - 1) * p.in2_mul) >> p.in2_shr) + p.in2_ring >= p.in2_wp) return -1011;
    }
    if (cfg < 0) cfg = vtd_conv_default_config(p);
    if (!vtd_conv_config_valid(p, cfg)) return -1007;
    const bool pooled = p.pool_pw != 0;
    if (pooled) {  // 2 x pool_pw max-pool behind the ReLU, taken in the register epilogue (window-major GEMM rows)
        if ((p.pool_pw != 1 && p.pool_pw != 2) || p.plist || p.in2 || p.flags != EPI_RELU || (p.ho & 1) || p.wo % p.pool_pw || !epi_direct_eligible(p) ||
            cfg == 7)
            return -1012;
        p.pool_log2 = p.pool_pw == 2 ? 2 : 1;
        p.pool_wq = p.wo / p.pool_pw;
        p.pool_hqwq = (p.ho / 2) * p.pool_wq;
    } else {
        p.pool_log2 = 0;
    }
    {   // register epilogue: plain NHWC fp16 output with bias / residual / ReLU only, and exact magic divisions for m -> (img, oy, ox)
        const char* e = getenv("VTD_EPI_DIRECT");  // tests: 0 = the LDS epilogue everywhere (read per launch so a test can flip it)
        const bool allow = pooled || !(e && e[0] == '0');   // (a pooled conv has no other epilogue)
        const uint64_t wo_e = pooled ? (uint64_t)p.pool_wq : (uint64_t)p.wo;
        const uint64_t howo = pooled ? (uint64_t)p.pool_hqwq : (uint64_t)p.ho * (uint64_t)p.wo;
        p.epi_direct = 0;
        // m -> (image, row, column) by multiplication: exact for every m < M when M x divisor < 2^40 (the error of ceil(2^40 / d) is
        // below d / 2^40 per unit of m).  Used by the loader of every launch that qualifies, and by the register epilogue.
        p.magic_ok = 0;
        if (!p.plist && (uint64_t)p.M * howo < (1ull << 40) && p.M > 0) {
            p.magic_ok = 1;
            p.magic_wo = ((1ull << 40) + wo_e - 1) / wo_e;
            p.magic_howo = ((1ull << 40) + howo - 1) / howo;
        }
        if (allow && epi_direct_eligible(p) && p.magic_ok) p.epi_direct = 1;
    }
    switch (cfg) {
        case 0: return launch_plain<256, 128, 4, 2, 3>(p, stream);
        case 1: return launch_plain<128, 128, 2, 2, 2>(p, stream);
        case 2: return launch_plain<128, 128, 2, 2, 3>(p, stream);
        case 3: return launch_cfg<256, 64, 4, 1, 2>(p, stream);
        case 4: return launch_cfg<256, 64, 4, 1, 3>(p, stream);
        case 5: return launch_plain<128, 64, 2, 2, 2>(p, stream);
        case 6: return launch_plain<128, 64, 2, 2, 3>(p, stream);
        case 7: return launch_cfg<64, 256, 1, 4, 2>(p, stream);
        case 8: return launch_cfg<128, 64, 2, 2, 2, true>(p, stream);
        case 9: return launch_cfg<128, 64, 2, 2, 3, true>(p, stream);
        case 14: return launch_plain<256, 128, 4, 4, 3>(p, stream);
        case 15: return launch_cfg<256, 256, 2, 4, 2>(p, stream);
        case 16: return launch_cfg<128, 64, 2, 2, 6>(p, stream);
        case 12: return launch_plain<208, 128, 2, 4, 3>(p, stream);
        case 13: return launch_plain<272, 128, 2, 4, 3>(p, stream);
        default: {
            // 256-row classed tiles: same images, the other cut of the pixel lists
            const int n_img = p.M / (p.tiles_per_img * 128);
            p.plist = p.plist_b; p.tile_combo = p.tile_combo_b; p.tiles_per_img = p.tiles_per_img_b;
            p.M = n_img * p.tiles_per_img_b * 256;
            return cfg == 10 ? launch_cfg<256, 64, 4, 1, 2, true>(p, stream) : launch_cfg<256, 64, 4, 1, 3, true>(p, stream);
        }
    }
}
