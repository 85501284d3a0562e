"""Pins the CPU restatement of the Transformer recogniser (oracle/trocr.py) against vectors produced by the locally
installed transformers classes (tests/golden/make_golden_trocr.py): encoder states, every step's logits and the greedy
ids on the reduced architecture; greedy ids + spot values on the full trocr-base-printed architecture."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import cstages
from oracle import trocr as otrocr
from vtd_amd._fixtures import synth, weights
from vtd_amd.trocr_spec import BASE_PRINTED, TINY, hf4_key, hf5_key


def test_key_mapping_round_trip():
    sd = weights.trocr_state_dict(TINY, seed=3)
    for k in sd:
        assert hf4_key(hf5_key(k)) == k
    assert hf5_key("encoder.encoder.layer.3.attention.attention.query.weight") == "encoder.layers.3.attention.q_proj.weight"
    assert hf5_key("encoder.encoder.layer.0.output.dense.bias") == "encoder.layers.0.mlp.fc2.bias"
    assert hf5_key("decoder.model.decoder.layers.1.fc2.weight") == "decoder.model.decoder.layers.1.fc2.weight"
    # the published trocr-base-printed parameter count (VisionEncoderDecoderModel minus the unused pooler)
    n = sum(v.numel() for v in weights.trocr_state_dict(BASE_PRINTED, seed=0).values()) if os.environ.get("VTD_SLOW_TESTS") else None
    assert n is None or n > 330_000_000


def test_processor_resize_is_pillow(golden_dir):
    """TrOCRProcessor's resize = PIL.Image.resize((384,384), BILINEAR): the C restatement is bit-exact on crop-sized inputs
    (up- and down-scaling, both axes different)."""
    from PIL import Image
    for seed, (h, w) in enumerate(((37, 211), (58, 398), (20, 60), (500, 700), (384, 384), (1, 9))):
        img = np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)
        want = np.asarray(Image.fromarray(img).resize((384, 384), Image.BILINEAR))
        assert np.array_equal(cstages.pil_resize_bilinear(img, 384, 384), want), (h, w)


def test_tiny_architecture_matches_transformers(golden_dir):
    g = np.load(os.path.join(golden_dir, "trocr_tiny.npz"))
    sd = weights.trocr_state_dict(TINY, seed=3, w_std=0.025, cross_gain=4.0)
    x = torch.stack([otrocr.preprocess(synth.glyph_crop(600 + i), TINY) for i in range(12)])
    enc = otrocr.encode(x, sd, TINY)
    assert enc.shape == (12, 37, 128)
    np.testing.assert_allclose(enc.numpy(), g["enc"], rtol=0, atol=2e-5)
    ids, logits = otrocr.generate(enc, sd, TINY)
    want = g["ids"]
    assert ids.shape[1] == want.shape[1] and np.array_equal(ids.numpy(), want)
    steps = want.shape[1] - 1
    live = (want[:, :-1] != 1) | (np.arange(steps)[None] == 0)         # steps run before a row finished
    err = np.abs(logits.numpy()[:, :steps] - g["logits"])[live]
    assert float(err.max()) <= 2e-4
    assert len({tuple(r) for r in want.tolist()}) >= 3 and (want == 2).sum(axis=1).min() >= 1
    # per-image calls (the reference's usage) give the same rows without padding
    single = otrocr.recognize_ids([synth.glyph_crop(600), synth.glyph_crop(603)], sd, TINY)
    for row, i in zip(single, (0, 3)):
        assert row == [t for t in want[i].tolist() if t != 1][:len(row)]


@pytest.mark.skipif(not os.environ.get("VTD_SLOW_TESTS"), reason="full-size ViT-base encoder on CPU (~1 min); set VTD_SLOW_TESTS=1")
def test_base_architecture_ids_match_transformers(golden_dir):
    man = json.load(open(os.path.join(golden_dir, "trocr_manifest.json")))["base"]
    g = np.load(os.path.join(golden_dir, "trocr_base.npz"))
    sd = weights.trocr_state_dict(BASE_PRINTED, seed=0)
    rows = man["rows"][:3]
    x = torch.stack([otrocr.preprocess(synth.glyph_crop(r["seed"]), BASE_PRINTED) for r in rows])
    enc = otrocr.encode(x, sd, BASE_PRINTED)
    np.testing.assert_allclose(enc[0, 0].numpy(), g["enc_cls"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(enc[0, [1, 100, 576]].numpy(), g["enc_rows"], rtol=0, atol=5e-5)
    for i, r in enumerate(rows):
        ids, logits = otrocr.generate(enc[i:i + 1], sd, BASE_PRINTED)
        assert ids[0].tolist() == r["ids"]
        if i == 0:
            np.testing.assert_allclose(logits[0, 0].numpy(), g["first_logits"], rtol=0, atol=2e-4)


def test_base_golden_is_well_posed(golden_dir):
    man = json.load(open(os.path.join(golden_dir, "trocr_manifest.json")))["base"]
    assert len(man["rows"]) >= 8 and man["distinct_sequences"] >= 6
    for r in man["rows"]:
        assert r["min_gap"] >= 0.015 and r["ids"][0] == 2 and r["ids"][-1] == 2 and 3 <= len(r["ids"]) <= 50
