#!/usr/bin/env python3
"""How fast does the ROCm library GEMM (torch.nn.functional.linear -> hipBLASLt / rocBLAS) run the encoder pass's dense-layer shapes?
A yardstick for csrc/dense_gemm.hip (tools/dense_gemm_bench.py measures that one): same M = crops x 577, fp16 in, fp32 accumulate."""
import sys
import time

import torch

crops = int(sys.argv[1]) if len(sys.argv) > 1 else 272
M = crops * 577
dev = "cuda"
torch.manual_seed(0)
for name, N, K in (("qkv", 2304, 768), ("out", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)):
    x = (torch.randn(M, K, device=dev) * 0.5).half()
    w = (torch.randn(N, K, device=dev) * 0.05).half()
    b = torch.randn(N, device=dev).half()
    for _ in range(3):
        y = torch.nn.functional.linear(x, w, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        y = torch.nn.functional.linear(x, w, b)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{name}: M={M} N={N} K={K}  {ms * 1e3:8.1f} us  {2.0 * M * N * K / ms / 1e9:7.1f} TFLOP/s")
