// Sustained dense-fp16 MFMA rate and shader clock of the GPU this runs on: a bare v_mfma_f32_16x16x32_f16 loop
// (operands in registers, 4 independent accumulator chains per wave, W waves per SIMD) timed with HIP events and
// with s_memtime.  Build + run:  hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int CH>
__global__ void mfma_chains(int iters, float* out, unsigned long long* cyc) {
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f - threadIdx.x * 0.002f); }
    floatx4 c[CH];
    for (int k = 0; k < CH; ++k) c[k] = floatx4{0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < CH; ++k) c[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[k], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0;
    for (int k = 0; k < CH; ++k) acc += c[k][k & 3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int CH>
void run_chains(int waves_per_simd) {
    const int iters = 50000, blocks = 256, threads = 256 * waves_per_simd;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, blocks * threads * 4); hipMalloc(&cyc, blocks * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mfma_chains<CH><<<blocks, threads>>>(100, out, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    mfma_chains<CH><<<blocks, threads>>>(iters, out, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < 256; ++i) avg += (double)h[i]; avg /= 256;
    const double flops = (double)blocks * (threads / 64) * iters * (double)CH * 16 * 16 * 32 * 2;
    printf("chains %2d waves/SIMD %d: %7.1f TFLOP/s; clock %.2f GHz; cycles per MFMA per wave %.1f\n", CH, waves_per_simd,
           flops / (ms * 1e-3) / 1e12, avg / (ms * 1e6), avg / ((double)iters * CH));
    hipFree(out); hipFree(cyc);
}

// MFMAs fed from LDS the way the convolution kernels do it: per half K-step a wave reads 4 A + 4 B fragments (ds_read_b128,
// conflict-free rows) for a 64 x 64 register tile = 16 MFMAs; the reads of half-step h+1 are issued before the MFMAs of h.
__global__ void mfma_lds(int iters, float* out, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) ((float*)lds)[i] = (float)(i & 15) * 0.01f;
    __syncthreads();
    const int lane = threadIdx.x & 63, fr = lane & 15, fq = lane >> 4;
    const char* base = lds + (threadIdx.x >> 6) * 8192;
    floatx4 c[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) c[i][j] = floatx4{0, 0, 0, 0};
    half8 fa[2][4], fb[2][4];
    auto load = [&](int h, half8 (&a)[4], half8 (&b)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = *(const half8*)(base + ((j * 16 + fr) * 128 + ((fq ^ ((fr >> 1) & 7)) << 4)) % 4096 + (h & 1) * 64);
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = *(const half8*)(base + 4096 + ((j * 16 + fr) * 128 + ((fq ^ ((fr >> 1) & 7)) << 4)) % 4096 + (h & 1) * 64);
    };
    load(0, fa[0], fb[0]);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        load(1, fa[1], fb[1]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[0][i], fa[0][j], c[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
        __builtin_amdgcn_sched_barrier(0);
        load(0, fa[0], fb[0]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[1][i], fa[1][j], c[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc += c[i][j][(i + j) & 3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

void run_lds(int waves_per_simd) {
    const int iters = 20000, blocks = 256, threads = 256 * waves_per_simd;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, blocks * threads * 4); hipMalloc(&cyc, blocks * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mfma_lds<<<blocks, threads, 65536>>>(100, out, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    mfma_lds<<<blocks, threads, 65536>>>(iters, out, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < 256; ++i) avg += (double)h[i]; avg /= 256;
    const double flops = (double)blocks * (threads / 64) * iters * 32.0 * 16 * 16 * 32 * 2;
    printf("LDS-fed 64x64 register tile, waves/SIMD %d: %7.1f TFLOP/s; clock %.2f GHz; cycles per MFMA per wave %.1f (per SIMD %.1f)\n",
           waves_per_simd, flops / (ms * 1e-3) / 1e12, avg / (ms * 1e6), avg / ((double)iters * 32), avg / ((double)iters * 32 * waves_per_simd));
    hipFree(out); hipFree(cyc);
}

__global__ void mfma_loop(int iters, float* out, unsigned long long* cyc) {
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f - threadIdx.x * 0.002f); }
    floatx4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    const int iters = 200000;
    for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd) {
        const int blocks = 256, threads = 256 * waves_per_simd;  // one block per CU
        float* out; unsigned long long* cyc;
        hipMalloc(&out, blocks * threads * 4); hipMalloc(&cyc, blocks * 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        mfma_loop<<<blocks, threads>>>(1000, out, cyc);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        mfma_loop<<<blocks, threads>>>(iters, out, cyc);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double avg = 0; for (int i = 0; i < 256; ++i) avg += (double)h[i]; avg /= 256;
        const double flops = (double)blocks * (threads / 64) * iters * 4.0 * 16 * 16 * 32 * 2;
        printf("waves/SIMD %d: %.3f ms, %.1f TFLOP/s dense fp16; s_memtime ticks per wave %.0f -> %.3f ticks/ns; MFMA issue interval %.2f ticks\n",
               waves_per_simd, ms, flops / (ms * 1e-3) / 1e12, avg, avg / (ms * 1e6), avg / (iters * 4.0 * waves_per_simd));
    }
    for (int w = 1; w <= 2; ++w) { run_chains<2>(w); run_chains<8>(w); run_chains<16>(w); }
    for (int w = 1; w <= 4; ++w) run_lds(w);
    return 0;
}
