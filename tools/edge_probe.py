#!/usr/bin/env python3
"""Where does the host block in the first steps after a drained pipeline?  The default line's three-deep loop (submit_detection(i),
submit_recognition(i-1), collect(i-2)) with host timers around the pieces of TextDetector.submit_batch."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "video-text-detection-system_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np
import torch

from vtd_amd import engine as eng_mod
from vtd_amd._fixtures import synth, weights
from vtd_amd.engine import DeviceFrames
from vtd_amd.pipeline import VideoTextPipeline

B = 32
frames = [np.stack([synth.text_frame(100 + 32 * k + i)[0] for i in range(B)]) for k in range(4)]
pipe = VideoTextPipeline(use_transformer_ocr=False, backbone="resnet18", batch_size=B)
pipe.detector.max_detections = 64
pipe.detector.model._max_batch = B
pipe.detector.model.load_state_dict(weights.margin_detector_state_dict("resnet18", 0))
pipe.recognizer.model.load_state_dict(weights.crnn_state_dict(0)) if hasattr(weights, "crnn_state_dict") else None
batches = [DeviceFrames(f) for f in frames]
log = []


def timed(obj, name, label):
    fn = getattr(obj, name)

    def wrapper(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            log.append((label, 1e3 * (time.perf_counter() - t)))
    setattr(obj, name, wrapper)


timed(pipe.detector.model, "forward", "model")
timed(eng_mod.PINNED, "take", "pinned")
state = {"det": None, "q": [], "k": 0}


def step():
    t0 = time.perf_counter()
    job = pipe.submit_detection(batches[state["k"] % 4]); state["k"] += 1
    t1 = time.perf_counter()
    if state["det"] is not None:
        state["q"].append(pipe.submit_recognition(state["det"]))
    t2 = time.perf_counter()
    while len(state["q"]) > 1:
        pipe.collect(state["q"].pop(0))
    state["det"] = job
    return 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (time.perf_counter() - t2)


def drain():
    if state["det"] is not None:
        state["q"].append(pipe.submit_recognition(state["det"])); state["det"] = None
    while state["q"]:
        pipe.collect(state["q"].pop(0))


for rep in range(3):
    for _ in range(6):
        step()
    drain()
    torch.cuda.synchronize()
    if os.environ.get("PROBE_SLEEP"):
        time.sleep(float(os.environ["PROBE_SLEEP"]))
    log.clear()
    out = []
    for i in range(6):
        n0 = len(log)
        a, b, c = step()
        out.append("%.2f/%.2f/%.2f [%s]" % (a, b, c, " ".join("%s %.2f" % x for x in log[n0:] if x[1] > 0.05)))
    drain()
    torch.cuda.synchronize()
    print("rep %d: " % rep + " | ".join(out))
