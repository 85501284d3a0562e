#!/usr/bin/env python3
"""Race screen for dense_gemm.hip (staged operands behind counted waits: a buffer read one barrier early is wrong only now and then):
the encoder pass of N crops repeated REPS times while a second stream keeps the CUs and the memory system busy with work of varying
length; every repetition's encoder states must equal the first one's bit for bit.  Also checks the first pass against the implicit-GEMM
path within the parity tolerance."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "video-text-detection-system_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np
import torch

from vtd_amd._fixtures import synth, weights
from vtd_amd.engine import DeviceFrames, TrOCREngine
from vtd_amd.trocr_spec import BASE_PRINTED as S

N = int(os.environ.get("N", "96"))
REPS = int(os.environ.get("REPS", "60"))
eng = TrOCREngine(S, weights.trocr_state_dict(S, seed=0), max_crops=N)
crops = [synth.glyph_crop(3000 + i) for i in range(N)]
frames = np.zeros((N, 720, 1280, 3), np.uint8)
boxes = []
for i, c in enumerate(crops):
    h, w = c.shape[:2]
    frames[i, 10:10 + h, 20:20 + w] = c
    boxes.append((i, 20, 10, 20 + w, 10 + h))
dev = DeviceFrames(frames)


def run():
    with eng.lock:
        eng.encode_crops(dev, boxes)
    return eng.read_tap("encoder", N)


os.environ["VTD_DENSE_GEMM"] = "0"
ref = run()
os.environ["VTD_DENSE_GEMM"] = "1"
first = run()
print("crops", N, "max |dense - implicit| on the encoder states:", float(np.abs(first - ref).max()))
side = torch.cuda.Stream()
noise = torch.randn(8192, 8192, device="cuda")
bad = 0
rng = np.random.default_rng(0)
for rep in range(REPS):
    with torch.cuda.stream(side):
        for _ in range(int(rng.integers(1, 12))):
            noise = noise * 1.0001 + 0.5
        if rep % 3 == 0:
            noise = (noise[:4096, :4096] @ noise[:4096, :4096].T).repeat(2, 2) * 1e-6
    again = run()
    if not np.array_equal(again, first):
        bad += 1
        d = np.argwhere(again != first)
        print("repetition", rep, "differs in", len(d), "values; first at", d[0].tolist())
print("repetitions", REPS, "mismatching", bad)
sys.exit(1 if bad else 0)
