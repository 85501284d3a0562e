"""GPU parity: crop+resize (bit-exact), CRNN (fp16 MFMA convs, fp32-gate LSTM) and the CTC decode against the
CPU oracle and the golden vectors produced by the reference's own CRNN / _decode_prediction."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import cstages
from oracle import nets as onets
from oracle import pipeline as opipe
from vtd_amd import nets as mynets
from vtd_amd._fixtures import synth, weights
from vtd_amd.vocab import build_vocab, id_to_char_table

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def crnn(hip):
    from vtd_amd.engine import RecognizerEngine
    sd = weights.calibrated_crnn_state_dict(11)
    eng = RecognizerEngine(97, sd, max_crops=64)
    yield eng, sd
    eng.close()


@pytest.fixture(scope="module")
def margin_crnn(hip):
    from vtd_amd.engine import RecognizerEngine
    sd = weights.margin_crnn_state_dict(11)
    eng = RecognizerEngine(97, sd, max_crops=64)
    yield eng, sd
    eng.close()


def _g1(golden_dir):
    """G1 golden + the scale every tolerance hangs on: the smallest max|logit difference| between two inputs."""
    g = np.load(os.path.join(golden_dir, "crnn_g1.npz"))
    ref = g["logits"]
    pair = min(float(np.abs(ref[i] - ref[j]).max()) for i in range(len(ref)) for j in range(i))
    return g, ref, pair


def _taps(eng, n):
    """h0/h1 come back as [n,512,1,31] (NCHW view of [n,31,512])"""
    return {"cnn": eng.read_tap("cnn", n), "h0": eng.read_tap("h0", n)[:, :, 0].transpose(0, 2, 1),
            "h1": eng.read_tap("h1", n)[:, :, 0].transpose(0, 2, 1)}


def test_crop_resize_bit_exact(crnn):
    from vtd_amd.engine import DeviceFrames
    eng, _ = crnn
    rng = np.random.default_rng(0)
    frames = np.stack([synth.text_frame(3)[0], rng.integers(0, 256, (720, 1280, 3), dtype=np.uint8)])
    boxes = [(0, 100, 50, 400, 120), (1, 0, 0, 1280, 720), (1, 37, 11, 165, 43), (0, 500, 300, 756, 364),  # last: exact 2x -> area path
             (1, 1000, 600, 1280, 720), (0, 10, 10, 21, 22), (1, 5, 5, 6, 700), (0, 640, 100, 1279, 101)]
    with eng.lock:
        n = eng.load_crops(DeviceFrames(frames), boxes)
    got = eng.read_tap("resized", n)
    for i, (f, x1, y1, x2, y2) in enumerate(boxes):
        exp = cstages.cv_resize_linear(frames[f][y1:y2, x1:x2], 128, 32)
        assert np.array_equal(got[i].astype(np.uint8), exp), i


def test_crnn_logits_and_taps_vs_reference_golden(crnn, golden_dir):
    """G1: logits / conv features / both LSTM layer outputs of the reference's own CRNN(97) (fp32) on BN-calibrated
    weights and glyph crops.  Tolerance for the fp16-MFMA path: 1/100 of the smallest inter-input logit variation
    (the network is input-dependent by O(1): asserted), taps to 1 % of their range."""
    eng, sd = crnn
    g, ref, pair = _g1(golden_dir)
    assert pair > 1.0
    x = torch.from_numpy(synth.glyph_batch(21, 8))
    logits = eng.forward_logits(x).cpu().numpy()
    taps = _taps(eng, 8)
    err = float(np.abs(logits - ref).max())
    print("logit max abs err", err, "tolerance", pair / 100, "inter-input variation", pair)
    assert err <= pair / 100
    for name in ("cnn", "h0", "h1"):
        want = g[name].astype(np.float32)
        rel = float(np.abs(taps[name] - want).max() / np.abs(want).max())
        between = min(float(np.abs(want[i] - want[j]).max()) for i in range(8) for j in range(i)) / float(np.abs(want).max())
        print(name, "rel err", rel, "inter-input variation (rel)", between)
        assert rel <= 1e-2 and between >= 20 * rel, name
    # arg-max agrees wherever the reference's logit gap is above twice the tolerance (and such steps are the majority)
    top2 = np.sort(ref, axis=2)[..., -2:]
    sure = (top2[..., 1] - top2[..., 0]) >= 2 * pair / 100
    assert sure.mean() > 0.8
    assert np.array_equal(logits.argmax(2)[sure], ref.argmax(2)[sure])


def test_crnn_negative_controls(crnn, golden_dir):
    """The parity assertions above can fail: a zero crop, a permuted batch and a bias-only answer are all far outside
    the tolerance; and the HIP path itself reproduces the reference's zero-input logits."""
    eng, sd = crnn
    g, ref, pair = _g1(golden_dir)
    tol = pair / 100
    z = eng.forward_logits(torch.zeros(1, 3, 32, 128)).cpu().numpy()
    assert float(np.abs(z - g["zero_logits"]).max()) <= tol
    assert float(np.abs(ref - z).max(axis=(1, 2)).min()) > 50 * tol          # ignoring the input would be caught
    assert float(np.abs(ref - ref[::-1]).max()) > 50 * tol                     # mixing up crops would be caught
    bias_only = np.broadcast_to(sd["classifier.bias"].numpy(), ref.shape)
    assert float(np.abs(ref - bias_only).max()) > 50 * tol                     # a bias-only kernel would be caught


def test_crnn_batch_independence(crnn, golden_dir):
    eng, _ = crnn
    g, ref, pair = _g1(golden_dir)
    x = torch.from_numpy(synth.glyph_batch(21, 8))
    one = np.concatenate([eng.forward_logits(x[i:i + 1]).cpu().numpy() for i in (0, 5)])
    assert float(np.abs(one - ref[[0, 5]]).max()) <= pair / 100


def test_margin_crnn_strings_vs_reference_golden(margin_crnn, golden_dir):
    """G1m: margin-carrier weights.  Every crop is well-posed by construction (top-1 margin >= 0.9), the strings are the
    ones the reference's CRNN + softmax + _decode_prediction produced, they differ from crop to crop, and the carrier
    LSTM units (edge detectors / latches through W_hh, both directions, both layers) sit on their saturated levels."""
    from vtd_amd.engine import ctc_greedy_decode
    eng, sd = margin_crnn
    g = np.load(os.path.join(golden_dir, "crnn_g1_margin.npz"))
    exp = json.load(open(os.path.join(golden_dir, "manifest.json")))["crnn_g1_margin"]["decoded"]
    x = torch.from_numpy(g["x_u8"]).permute(0, 3, 1, 2).float() / 255.0
    n = x.shape[0]
    logits = eng.forward_logits(x)
    taps = _taps(eng, n)
    car = [0, 1, 2, 256, 257, 258]
    assert float(np.abs(taps["h0"][:, :, car] - g["h0_carrier"]).max()) <= 5e-3
    assert float(np.abs(taps["h1"][:, :, car] - g["h1_carrier"]).max()) <= 5e-3
    lg = logits.cpu().numpy()
    assert np.isfinite(lg).all()
    ref = g["logits"]
    top2 = np.sort(ref, axis=2)[..., -2:]
    gap = float((top2[..., 1] - top2[..., 0]).min())
    err = float(np.abs(lg - ref).max())
    print("margin logits: max abs err", err, "min top-2 logit gap", gap)
    assert err <= gap / 20
    got = ctc_greedy_decode(logits, id_to_char_table(build_vocab()))
    assert [t for t, _ in got] == [e["text"] for e in exp]                      # identical strings, every crop
    assert max(abs(c - e["confidence"]) for (_, c), e in zip(got, exp)) <= 2e-3
    assert len({t for t, _ in got}) >= 5


def test_crnn_from_crops_matches_oracle_strings(margin_crnn):
    """crop + resize + CRNN + decode on device vs the oracle's recognize_batch: well-posedness is asserted, not assumed."""
    from vtd_amd.engine import DeviceFrames, ctc_greedy_decode
    eng, sd = margin_crnn
    frame, rects = synth.text_frame(5)
    boxes = []
    for r in rects:
        hw = 0.5 * (abs(np.cos(r["angle"])) * r["length"] + abs(np.sin(r["angle"])) * r["thick"])
        hh = 0.5 * (abs(np.sin(r["angle"])) * r["length"] + abs(np.cos(r["angle"])) * r["thick"])
        boxes.append((0, int(max(0, r["cx"] - hw)), int(max(0, r["cy"] - hh)), int(min(1280, r["cx"] + hw)), int(min(720, r["cy"] + hh))))
    boxes += [(0, 100 + 40 * i, 60 + 50 * i, 300 + 60 * i, 100 + 55 * i) for i in range(6)]
    logits = eng.forward_crops(DeviceFrames(frame), boxes)
    got = ctc_greedy_decode(logits, id_to_char_table(build_vocab()))
    crops = [frame[y1:y2, x1:x2] for (_, x1, y1, x2, y2) in boxes]
    exp, probs = opipe.recognize_batch(crops, sd, return_probs=True)
    top2 = np.sort(probs, axis=2)[..., -2:]
    well_posed = (top2[..., 1] - top2[..., 0]).min(axis=1) >= 1e-2
    assert well_posed.mean() >= 0.9
    for (t, c), e, ok in zip(got, exp, well_posed):
        if ok:
            assert t == e["text"]
            assert abs(c - e["confidence"]) <= 2e-3
    assert len({e["text"] for e in exp}) >= 2


def test_ctc_decode_golden_quirks(hip, golden_dir):
    """G4: the reference's _decode_prediction outputs on hand-built and random probability tables."""
    from vtd_amd.engine import ctc_greedy_decode
    inputs = np.load(os.path.join(golden_dir, "decode_inputs.npz"))
    expected = json.load(open(os.path.join(golden_dir, "decode_expected.json")))
    table = id_to_char_table(build_vocab())
    for name, exp in expected.items():
        p = torch.from_numpy(inputs[name])
        text, conf = ctc_greedy_decode(p.unsqueeze(0), table, apply_softmax=False)[0]
        assert text == exp["text"], name
        assert abs(conf - exp["confidence"]) <= 1e-6, name
        if name.startswith("rand"):  # rows are a softmax output: the fused-softmax path on log p must agree
            t2, c2 = ctc_greedy_decode(torch.log(p).unsqueeze(0), table)[0]
            assert t2 == exp["text"] and abs(c2 - exp["confidence"]) <= 1e-5


def test_text_recognizer_surface_and_mock_seam(hip):
    from unittest.mock import patch
    from vtd_amd.recognizer import TextRecognizer
    rec = TextRecognizer(use_transformer=False, max_crops=16)
    assert isinstance(rec.model, mynets.CRNN) and len(rec.vocab) == 97 and rec.use_transformer is False
    img = np.random.default_rng(0).integers(0, 255, (480, 640, 3), dtype=np.uint8)
    with patch.object(rec.model, "forward") as fwd:  # tests/test_models.py:88-98
        fwd.return_value = torch.rand(2, 10, len(rec.vocab))
        res = rec.recognize_batch([img, img])
        assert fwd.called and len(res) == 2
        for r in res:
            assert isinstance(r["text"], str) and isinstance(r["confidence"], float)
    rec.model.load_state_dict(weights.margin_crnn_state_dict(11))
    crop = synth.text_frame(100)[0][560:690, 170:510]
    res = rec.recognize(crop)
    exp = opipe.recognize_batch([crop], rec.model.state_dict())[0]
    assert res["text"] == exp["text"] != "" and abs(res["confidence"] - exp["confidence"]) < 2e-3
    assert rec.recognize(None) == {"text": "", "confidence": 0.0}
    assert rec.recognize_batch([np.zeros((0, 5, 3), np.uint8)]) == [{"text": "", "confidence": 0.0}]
    # use_transformer=True constructs too (tests/test_gpu_trocr.py drives it)


@pytest.mark.parametrize("mode", ["1", "3"])
def test_crnn_halo_convs_forced(hip, golden_dir, monkeypatch, mode):
    """The CRNN's stride-1 3x3 layers on the halo-tile kernel (8x32 pixel blocks on the 8x32 maps, 16x16 on 16x64) wherever it
    applies, not only where the autotune happens to pick it: same golden logits, same tolerance."""
    from vtd_amd.engine import RecognizerEngine
    monkeypatch.setenv("VTD_FORCE_HALO", mode)  # 1: first-generation halo kernel, 3: hand-pipelined conv_halo64
    sd = weights.calibrated_crnn_state_dict(11)
    eng = RecognizerEngine(97, sd, max_crops=8)
    try:
        g, ref, pair = _g1(golden_dir)
        x = torch.from_numpy(synth.glyph_batch(21, 8))
        logits = eng.forward_logits(x).cpu().numpy()
    finally:
        eng.close()
    err = float(np.abs(logits - ref).max())
    print("halo-forced logit max abs err", err, "tolerance", pair / 100)
    assert err <= pair / 100


@pytest.mark.parametrize("ncrops", [1, 7, 37, 272])
def test_pools_fused_into_the_conv_epilogues_are_bit_identical(hip, monkeypatch, ncrops):
    """The MaxPool2d((2,2)) behind conv2 and the MaxPool2d((2,1)) behind conv4 / conv6 (text_recognizer.py:17-22) ride in those
    convolutions' register epilogues: GEMM rows run window-major, so a pooling window sits in neighbouring lanes of one accumulator
    fragment and its maximum is two quad-permute DPP steps (option fuse_pools, default 1; three launches fewer, the un-pooled maps are
    never written).  max commutes with the monotone fp16 rounding, so conv features and logits must equal the separate-pool graph's BIT
    FOR BIT -- at crop counts that give partial tiles, on the shipped kernel selection and on every tile shape that can finish from
    registers (forced), including the uneven 208- / 272-row tiles and the 256 x 256 one."""
    from vtd_amd.engine import RecognizerEngine
    sd = weights.calibrated_crnn_state_dict(11)
    x = torch.from_numpy(np.concatenate([synth.glyph_batch(500 + i, 8) for i in range((ncrops + 7) // 8)])[:ncrops])
    monkeypatch.setenv("VTD_HALO_CONV", "0")   # the un-fused graph on the implicit GEMM too: same K order on both sides

    def run(fuse, cfg=None):
        if cfg is None: monkeypatch.delenv("VTD_FORCE_CONV_CFG", raising=False)
        else: monkeypatch.setenv("VTD_FORCE_CONV_CFG", str(cfg))
        eng = RecognizerEngine(97, sd, max_crops=max(ncrops, 8), options={"fuse_pools": fuse})
        try:
            logits = eng.forward_logits(x).cpu().numpy()
            return logits, eng.read_tap("cnn", ncrops)
        finally:
            eng.close()

    want_logits, want_cnn = run(0)
    assert float(np.abs(want_cnn).max()) > 0.1 and len({want_logits[i].tobytes() for i in range(ncrops)}) == ncrops   # the fixture looks at its input
    for cfg in (None, 0, 1, 3, 5, 12, 13, 14, 15) if ncrops in (7, 272) else (None,):
        logits, cnn = run(1, cfg)
        assert np.array_equal(cnn, want_cnn), (ncrops, cfg)
        assert np.array_equal(logits, want_logits), (ncrops, cfg)
