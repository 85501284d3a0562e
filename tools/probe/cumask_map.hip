// Which physical CU does bit i of a hipExtStreamCreateWithCUMask mask enable?  One launch per single-bit mask; every workgroup reports the
// XCC (die) id and the hardware id register of the CU it ran on.  Built by tools/cumask_map.py (hipcc -shared), run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__global__ void where_kernel(uint32_t* out) {
    if (threadIdx.x == 0) {
        const uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
        const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_REG_HW_ID
        out[2 * blockIdx.x] = xcc;
        out[2 * blockIdx.x + 1] = hw;
    }
    // keep the CU busy for a moment so that concurrent workgroups spread over every enabled CU
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < 20000) {}
}

extern "C" int cumask_where(const uint32_t* mask, int words, int blocks, uint32_t* host_out) {
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, words, mask) != hipSuccess) return -1;
    uint32_t* dev = nullptr;
    if (hipMalloc(&dev, (size_t)blocks * 8) != hipSuccess) return -2;
    hipLaunchKernelGGL(where_kernel, dim3(blocks), dim3(64), 0, s, dev);
    if (hipStreamSynchronize(s) != hipSuccess) return -3;
    if (hipMemcpy(host_out, dev, (size_t)blocks * 8, hipMemcpyDeviceToHost) != hipSuccess) return -4;
    (void)hipFree(dev);
    (void)hipStreamDestroy(s);
    return 0;
}
