#!/usr/bin/env python3
"""Host cost of the enqueue calls (launch-bound check): batch 1 keeps the GPU faster than the host."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "video-text-detection-system_amd"), ROOT]
import ctypes as C
import numpy as np
import torch
from vtd_amd import _native, nets as mynets
from vtd_amd._fixtures import synth, weights
from vtd_amd.engine import DetectorEngine, DeviceFrames, PostProcessor

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sd = weights.margin_detector_state_dict("resnet18", 0)
eng = DetectorEngine("resnet18", sd, max_batch=max(n, 1))
frames = DeviceFrames(np.stack([synth.text_frame(i, 720, 1280)[0] for i in range(n)]))
prob = torch.empty((n, 1, 640, 640), dtype=torch.float32, device="cuda")
pp = PostProcessor(n, 640, 640, 64)
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
lib = eng.lib
def pre(): _native.check(lib.vtd_detector_preprocess(eng.handle, C.c_void_p(frames.tensor.data_ptr()), n, 720, 1280, s))
def fwd(): _native.check(lib.vtd_detector_forward(eng.handle, n, C.c_void_p(prob.data_ptr()), None, s))
def post(): pp.run_device(prob, [1280] * n, [720] * n, 0.5)
for f in (pre, fwd, post):
    f(); torch.cuda.synchronize()
for name, f in (("preprocess", pre), ("forward", fwd), ("postprocess", post)):
    iters = 50
    t0 = time.perf_counter()
    for _ in range(iters):
        f()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:12s} host enqueue {1e6 * (t1 - t0) / iters:8.1f} us/call   (drain after loop {1e6 * (t2 - t1):8.1f} us total)")
print("ops in detector graph:", lib.vtd_detector_num_ops(eng.handle))
