#!/usr/bin/env python3
"""dense_gemm.hip alone on the GPU: the encoder pass's five GEMM shapes at a 287-crop batch, TFLOP/s per shape and tile order
(VTD_DGM_ORDER=0 row-major runs per XCD, 1 super-blocks), against a float64 check of a few output entries.

The loop-structure variants (VTD_DGM_VARIANT / DGM_BENCH_VARIANTS) are compiled only into an instrumented library:
    VTD_LIB_VARIANT=dgmexp VTD_EXTRA_HIPCC_FLAGS=-DVTD_DGM_EXPERIMENT python video-text-detection-system_amd/build_native.py
    VTD_LIB_VARIANT=dgmexp python tools/dense_gemm_bench.py
On the product library every "variant" is the shipped kernel (the switch is compiled out)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "video-text-detection-system_amd"), ROOT):
    sys.path.insert(0, p)
import torch
from vtd_amd import _native

lib = C.CDLL(_native.LIB_PATH)
fn = getattr(lib, "_Z21vtd_launch_dense_gemmPKDF16_iS0_iPKfPvilii i P12ihipStream_t".replace(" ", ""), None)
if fn is None:
    import subprocess
    sym = [ln.split()[-1] for ln in subprocess.run(["nm", "-D", _native.LIB_PATH], capture_output=True, text=True).stdout.splitlines()
           if "vtd_launch_dense_gemm" in ln and "dense_gemm_ex" not in ln][0]
    fn = getattr(lib, sym)
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p]
M = 287 * 577
shapes = [("qkv", 2304, 768), ("o", 768, 768), ("fc1", 3072, 768), ("fc1+gelu", 3072, 768), ("fc2", 768, 3072), ("ck/cv", 1024, 768)]
EPI_OUT_F16 = 32
EPI_GELU = 64
g = torch.Generator(device="cuda").manual_seed(0)
for name, N, K in shapes:
    A = (torch.rand((M, K), device="cuda", generator=g) * 2 - 1).half()
    W = (torch.rand((N, K), device="cuda", generator=g) * 2 - 1).half() * 0.05
    bias = torch.zeros(N, device="cuda")
    out = torch.empty((M, N), device="cuda", dtype=torch.float16)
    s = torch.cuda.current_stream()
    out0 = None
    flags = EPI_OUT_F16 | (EPI_GELU if "gelu" in name else 0)
    for order, var in tuple(("1", v) for v in os.environ.get("DGM_BENCH_VARIANTS", "0 8 16 0 8 16").split()):
        os.environ["VTD_DGM_ORDER"] = order
        os.environ["VTD_DGM_VARIANT"] = var
        for _ in range(3):
            rc = fn(A.data_ptr(), K, W.data_ptr(), N, bias.data_ptr(), out.data_ptr(), N, M, N, K, flags, s.cuda_stream)
            assert rc == 0, rc
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn(A.data_ptr(), K, W.data_ptr(), N, bias.data_ptr(), out.data_ptr(), N, M, N, K, flags, s.cuda_stream)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / 10
        rows = torch.tensor([0, 1, 255, 256, 70000, M - 1], device="cuda")
        ref = (A[rows].double() @ W.double().T)
        if flags & EPI_GELU:
            ref = torch.nn.functional.gelu(ref)
        ref = ref.float()
        err = float((out[rows].float() - ref).abs().max())
        if var == "0" and out0 is None:
            out0 = out.clone()
        same = "" if var in ("0", "4", "32", "64", "96") or out0 is None else f"  bitwise == variant 0: {bool(torch.equal(out, out0))}"
        print(f"{name:6s} N={N:5d} K={K:5d} order={order} variant={var}: {ms * 1e3:8.1f} us  {2 * M * N * K / (ms * 1e-3) / 1e12:7.1f} TFLOP/s  max err {err:.3e}{same}")
