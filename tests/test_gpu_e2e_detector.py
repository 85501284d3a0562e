"""End-to-end detector parity on margin-controlled fixtures: TextDetector (HIP) vs the CPU
oracle on the same synthetic 720p / 1080p frames.  With the decision taken on the 1/255 lattice the boxes must
be identical (IoU = 1 >= 0.99), confidences within 2e-3 (fp16 map vs fp32 map)."""
import numpy as np
import pytest
import torch

from oracle import pipeline as opipe
from vtd_amd._fixtures import synth, weights

pytestmark = pytest.mark.gpu


def iou(a, b):
    x1, y1, x2, y2 = max(a[0], b[0]), max(a[1], b[1]), min(a[2], b[2]), min(a[3], b[3])
    inter = max(0, x2 - x1) * max(0, y2 - y1)
    ua = (a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter
    return inter / ua if ua else 0.0


@pytest.fixture(scope="module")
def det18(hip):
    from vtd_amd.detector import TextDetector
    d = TextDetector(backbone="resnet18", max_batch=8)
    sd = weights.margin_detector_state_dict("resnet18", 0)
    d.model.load_state_dict(sd)
    return d, sd


def _compare(got, exp):
    assert len(got) == len(exp) and len(exp) >= 4
    for g, e in zip(got, exp):
        assert iou(g["bbox"], e["bbox"]) >= 0.99
        assert g["bbox"] == e["bbox"] and g["polygon"] == e["polygon"]
        assert abs(g["confidence"] - e["confidence"]) <= 2e-3
        assert all(isinstance(v, int) for v in g["bbox"]) and isinstance(g["confidence"], float)


@pytest.mark.parametrize("shape,seed", [((720, 1280), 0), ((720, 1280), 101), ((1080, 1920), 1000), ((480, 640), 7)])
def test_detect_single_frame_matches_oracle(det18, shape, seed):
    d, sd = det18
    frame, _ = synth.text_frame(seed, *shape)
    exp, ref_map = opipe.detect(frame, sd, "resnet18", 0.5, return_map=True)
    assert float(np.abs(ref_map - 0.5).min()) > 0.4  # the fixture really has margin
    _compare(d.detect(frame, 0.5), exp)


def test_detect_batch_32_frames_config2(det18):
    """BASELINE config 2 (B=32 720p, detector only) at reduced batch to keep the oracle side short."""
    d, sd = det18
    frames = [synth.text_frame(100 + i)[0] for i in range(8)]
    got = d.detect_batch(frames, 0.5)
    for f, g in zip(frames, got):
        _compare(g, opipe.detect(f, sd, "resnet18", 0.5))


def test_reference_seams_and_error_convention(det18):
    from unittest.mock import patch
    d, _ = det18
    assert d.device in ("cuda", "cpu")
    assert d.detect(None) == [] and d.detect(np.array([])) == []          # tests/test_models.py:39-46
    assert d.detect(np.zeros((48, 64), np.uint8)) == []                   # grayscale ends in the exception path
    img = np.random.default_rng(0).integers(0, 255, (480, 640, 3), dtype=np.uint8)
    with patch.object(d.model, "forward") as fwd:                         # tests/test_models.py:30-37
        fwd.return_value = {"probability": torch.ones(1, 1, 160, 160) * 0.8, "threshold": torch.ones(1, 1, 160, 160) * 0.5}
        dets = d.detect(img, 0.5)
        from oracle import cstages
        exp = cstages.postprocess(np.full((160, 160), 0.8, np.float32), 640, 480, 0.5)
        assert fwd.called and len(dets) == 1 and dets[0]["bbox"] == exp[0]["bbox"] and dets[0]["polygon"] == exp[0]["polygon"]
    pm = np.random.default_rng(1).random((160, 160))                       # float64 map, tests/test_models.py:48-58
    assert [x["bbox"] for x in d._post_process(pm, 640, 480, 0.5)] == [x["bbox"] for x in cstages.postprocess(pm, 640, 480, 0.5)]
    rgb = d.transform(img[..., ::-1])
    assert tuple(rgb.shape) == (3, 640, 640)
    assert np.array_equal(rgb.numpy(), opipe.preprocess(img)[0].numpy().astype(np.float16).astype(np.float32))
