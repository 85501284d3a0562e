#!/usr/bin/env python3
"""Debug aid: greedy ids of the base architecture with and without live-row compaction against the fp32 oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "video-text-detection-system_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np, torch
from oracle import trocr as otrocr
from vtd_amd._fixtures import synth, weights
from vtd_amd.engine import TrOCREngine, trim_generated
from vtd_amd.trocr_spec import BASE_PRINTED as S
sd = weights.trocr_state_dict(S, seed=0)
eng = TrOCREngine(S, sd, max_crops=16)
px = torch.stack([otrocr.preprocess(synth.glyph_crop(950 + i), S) for i in range(14)])
res = {}
for mode in ("1", "0", "1"):
    os.environ["VTD_TROCR_COMPACT"] = mode
    ids, _ = eng.generate_pixels(px)
    print("mode", mode, "steps", eng.last_steps, "lens", [len(r) for r in trim_generated(ids, S)])
    res[mode] = ids.numpy().copy()
torch.set_num_threads(16)
oid, olg = otrocr.generate(otrocr.encode(px, sd, S), sd, S)
top2 = olg.topk(2, dim=2).values
gap = (top2[..., 0] - top2[..., 1]).numpy()
oid = oid.numpy()
for i in range(14):
    n = int((oid[i, 1:] != S.pad_token_id).sum())
    a, b = res["1"][i, :oid.shape[1]], res["0"][i, :oid.shape[1]]
    print(i, "oracle len", n + 1, "min gap %.4f" % gap[i, :n].min(), "compact==oracle", bool((a == oid[i]).all()), "padded==oracle", bool((b == oid[i]).all()),
          "first diff compact/padded", int(np.argmax(res["1"][i] != res["0"][i])) if (res["1"][i] != res["0"][i]).any() else -1)
    if not (a == oid[i]).all() or not (b == oid[i]).all():
        print("   oracle ", oid[i].tolist()); print("   compact", res["1"][i].tolist()); print("   padded ", res["0"][i].tolist())
