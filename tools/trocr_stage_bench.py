#!/usr/bin/env python3
"""The Transformer recogniser's two stages alone on the GPU, on the bench's own crops (32 x 720p frames, ResNet-18 boxes): wall time of the
encoder pass and of the greedy decode, each repeated with nothing else running."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "video-text-detection-system_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np
import torch

os.environ.setdefault("VTD_TROCR_SEEDED", "0")
os.environ.setdefault("VTD_TROCR_MAX_CROPS", "1024")
from vtd_amd._fixtures import synth, weights
from vtd_amd.engine import DeviceFrames
from vtd_amd.pipeline import VideoTextPipeline

B = int(os.environ.get("B", "32"))
reps = int(os.environ.get("REPS", "3"))
frames = np.stack([synth.text_frame(100 + i)[0] for i in range(B)])
pipe = VideoTextPipeline(use_transformer_ocr=True, backbone="resnet18", batch_size=B)
pipe.detector.model._max_batch = B
pipe.detector.model.load_state_dict(weights.margin_detector_state_dict("resnet18", 0))
batch = DeviceFrames(frames)
dets = pipe.detector.detect_batch(batch, 0.5)
boxes = [(i, *d["bbox"]) for i, ds in enumerate(dets) for d in ds if d["bbox"][2] > d["bbox"][0] and d["bbox"][3] > d["bbox"][1]]
eng = pipe.recognizer.model.engine()
print(f"{len(boxes)} crops")
with eng.lock:
    for rep in range(reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = eng.encode_crops(batch, boxes, slot=0)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        ids, _ = eng.generate_current(n, slot=0)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if rep:
            print(f"encoder pass {1e3 * (t1 - t0):7.1f} ms   decode {1e3 * (t2 - t1):7.1f} ms ({eng.last_steps} steps)")
