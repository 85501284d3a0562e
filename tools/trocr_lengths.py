#!/usr/bin/env python3
"""How long are the greedy sequences of the bench's TrOCR workload?  One configs[2]-shaped batch (32 x 720p, ResNet-18 boxes) through the
Transformer recogniser on the bench's seeded weights; prints the histogram of generated lengths (tokens incl. </s>) -- the share of
(row, step) pairs a decoder that stops finished rows would still compute."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "video-text-detection-system_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np

os.environ.setdefault("VTD_TROCR_SEEDED", "0")
os.environ.setdefault("VTD_TROCR_MAX_CROPS", "512")
from vtd_amd._fixtures import synth, weights
from vtd_amd.engine import DeviceFrames
from vtd_amd.pipeline import VideoTextPipeline

B = 32
frames = np.stack([synth.text_frame(100 + i)[0] for i in range(B)])
pipe = VideoTextPipeline(use_transformer_ocr=True, backbone="resnet18", batch_size=B)
pipe.detector.model.load_state_dict(weights.margin_detector_state_dict("resnet18", 0))
batch = DeviceFrames(frames)
dets = pipe.detector.detect_batch(batch, 0.5)
boxes = [(i, *d["bbox"]) for i, ds in enumerate(dets) for d in ds if d["bbox"][2] > d["bbox"][0] and d["bbox"][3] > d["bbox"][1]]
ids = pipe.recognizer.model.recognize_boxes_ids(batch, boxes)
lens = np.array([len(s) - 1 for s in ids])
steps = int(lens.max())
print(f"{len(boxes)} crops; generated tokens: min {lens.min()} median {int(np.median(lens))} mean {lens.mean():.1f} max {steps}")
print("histogram (tokens: rows):", {int(k): int(v) for k, v in zip(*np.unique(lens, return_counts=True))})
print(f"row-steps a stopping decoder computes: {int(lens.sum())} of {len(lens) * steps} = {lens.sum() / (len(lens) * steps):.2f}")
