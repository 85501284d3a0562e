"""BASELINE config 3 at reduced batch: the full VideoTextPipeline (HIP detector + recogniser, batched fast path)
against the CPU oracle's reference-shaped pipeline on the same synthetic 720p frames: identical boxes and
polygons, identical strings (where the oracle's top-1 margin is >= 1e-2), confidences within 2e-3."""
import asyncio
import json

import numpy as np
import pytest

from oracle import pipeline as opipe
from vtd_amd import nets as mynets
from vtd_amd._fixtures import synth, weights

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pipeline(hip):
    from vtd_amd.pipeline import VideoTextPipeline
    p = VideoTextPipeline(use_transformer_ocr=False, backbone="resnet18", batch_size=8)
    det_sd = weights.margin_detector_state_dict("resnet18", 0)
    rec_sd = weights.margin_crnn_state_dict(11)   # every crop well-posed by construction: strings are asserted
    p.detector.model.load_state_dict(det_sd)
    p.recognizer.model.load_state_dict(rec_sd)
    return p, det_sd, rec_sd


def _same(got, exp, rec_sd, frame, texts=None):
    """Identical boxes; identical strings on every well-posed crop, and >= 90 % of the crops must be well-posed (oracle
    top-1 probability margin >= 1e-2 at every timestep).  `texts` collects the strings so callers can assert diversity."""
    assert len(got) == len(exp)
    well_posed = 0
    for g, e in zip(got, exp):
        assert g["bbox"] == e["bbox"]
        assert abs(g["detection_confidence"] - e["detection_confidence"]) <= 2e-3
        x1, y1, x2, y2 = e["bbox"]
        _, probs = opipe.recognize_batch([frame[y1:y2, x1:x2]], rec_sd, return_probs=True)
        top2 = np.sort(probs[0], axis=1)[:, -2:]
        if (top2[:, 1] - top2[:, 0]).min() >= 1e-2:
            well_posed += 1
            assert g["text"] == e["text"]
            assert abs(g["recognition_confidence"] - e["recognition_confidence"]) <= 2e-3
        if texts is not None:
            texts.append(e["text"])
    assert well_posed >= 0.9 * len(exp)


def test_frame_batch_fast_path_matches_oracle(pipeline):
    p, det_sd, rec_sd = pipeline
    frames = [synth.text_frame(100 + i)[0] for i in range(4)]
    info = [(i, i / 10.0) for i in range(4)]
    assert p._fast_path_ok(frames)
    got = asyncio.run(p._process_frame_batch(frames, info, "/tmp"))
    exp = opipe.process_frame_batch(frames, info, det_sd, "resnet18", rec_sd, 0.5)
    json.dumps(got)
    assert [g["frame_number"] for g in got] == [0, 1, 2, 3]
    texts = []
    for g, e, f in zip(got, exp, frames):
        assert g["timestamp"] == e["timestamp"]
        _same(g["detections"], e["detections"], rec_sd, f, texts)
        for gd, ed in zip(g["detections"], e["detections"]):
            assert gd["polygon"] == ed["polygon"]
    assert len(texts) >= 20 and len(set(texts)) >= 4 and all(texts)   # the strings carry information


def test_single_frame_route_and_mixed_sizes_on_the_device_path(pipeline):
    p, det_sd, rec_sd = pipeline
    frame = synth.text_frame(7, 1080, 1920)[0]
    got = p.process_single_frame(frame)
    exp = opipe.process_single_frame(frame, det_sd, "resnet18", rec_sd, 0.5)
    _same(got["detections"], exp["detections"], rec_sd, frame)
    assert all("polygon" not in d for d in got["detections"])
    # frames of different sizes in one batch (the reference takes any mix, pipeliine.py:96-101): grouped by shape, one device pass
    # per group, results back in frame order; only non-uint8 / non-HxWx3 frames leave the device path
    mixed = [synth.text_frame(1, 720, 1280)[0], synth.text_frame(2, 480, 640)[0], synth.text_frame(3, 720, 1280)[0]]
    assert p._fast_path_ok(mixed)
    assert not p._fast_path_ok([mixed[0], mixed[1].astype(np.float32)]) and not p._fast_path_ok([mixed[0][..., 0]])
    p.route_counts = {"device": 0, "reference": 0}
    out = asyncio.run(p._process_frame_batch(mixed, [(0, 0.0), (1, 0.1), (2, 0.2)], "/tmp"))
    assert p.route_counts == {"device": 3, "reference": 0} and [o["frame_number"] for o in out] == [0, 1, 2]
    for o, f in zip(out, mixed):
        e = opipe.process_frame_batch([f], [(0, 0.0)], det_sd, "resnet18", rec_sd, 0.5)[0]
        _same(o["detections"], e["detections"], rec_sd, f)


def test_mock_seams_switch_off_the_fast_path(pipeline):
    from unittest.mock import patch
    p, _, _ = pipeline
    frames = [synth.text_frame(100)[0]] * 2
    with patch.object(p.detector, "detect") as det, patch.object(p.recognizer, "recognize") as rec:
        det.return_value = [{"bbox": [50, 80, 200, 120], "confidence": 0.8}]
        rec.return_value = {"text": "TEST TEXT", "confidence": 0.9}
        assert not p._fast_path_ok(frames)
        out = asyncio.run(p._process_frame_batch(frames, [(0, 0.0), (1, 0.1)], "/tmp"))
        assert det.call_count == 2 and rec.call_count == 2
        assert out[0]["detections"][0]["text"] == "TEST TEXT" and out[0]["detections"][0]["polygon"] == []


def test_pipelined_halves_equal_the_one_shot_pass(pipeline):
    """submit_detection / submit_recognition / collect with two batches in flight == process_device_batch."""
    from vtd_amd.engine import DeviceFrames
    p, _, _ = pipeline
    a = DeviceFrames([synth.text_frame(100 + i)[0] for i in range(3)])
    b = DeviceFrames([synth.text_frame(200 + i)[0] for i in range(3)])
    exp_a, exp_b = p.process_device_batch(a), p.process_device_batch(b)
    ja = p.submit_detection(a)
    jb = p.submit_detection(b)          # second detector pass enqueued before the first is collected
    ja = p.submit_recognition(ja)
    jb = p.submit_recognition(jb)
    assert p.collect(ja) == exp_a and p.collect(jb) == exp_b


def test_upload_stream_batches_equal_resident_batches(pipeline):
    """Frames uploaded from pinned host memory on their own stream (DeviceFrames(stream=...)): the detector, its side-stream
    post-process and the recogniser order themselves behind the upload event; results equal the resident-batch pass, with
    several uploads in flight."""
    import torch
    from vtd_amd.engine import DeviceFrames
    p, _, _ = pipeline
    up = torch.cuda.Stream()
    host = [torch.from_numpy(np.stack([synth.text_frame(300 + 10 * k + i)[0] for i in range(3)])).pin_memory() for k in range(3)]
    exp = [p.process_device_batch(DeviceFrames(h.numpy())) for h in host]
    jobs = [p.submit_detection(DeviceFrames(h, stream=up)) for h in host]   # three uploads + detector passes enqueued
    jobs = [p.submit_recognition(j) for j in jobs]
    assert [p.collect(j) for j in jobs] == exp


@pytest.mark.parametrize("ahead", ["1", "0"])
def test_process_video_pipelined_loop_matches_batch_pass(pipeline, tmp_path, monkeypatch, ahead):
    """process_video on a raw-frame source: batches ride the device pipeline (upload stream, detector, post-process / recogniser
    streams; host frames enter the detector one job after their copy was issued, VTD_STAGE_AHEAD=0: in the same push) and come back
    in frame order with exactly the results of the one-shot batch pass."""
    from vtd_amd.engine import DeviceFrames
    monkeypatch.setenv("VTD_STAGE_AHEAD", ahead)
    p, _, _ = pipeline
    frames = np.stack([synth.text_frame(400 + i)[0] for i in range(11)])
    path = tmp_path / "clip.npy"
    np.save(path, frames)
    (tmp_path / "clip.npy.json").write_text(json.dumps({"fps": 10.0}))
    old_bs = p.batch_size
    p.batch_size = 4   # 4 + 4 + 3 frames: three batches in flight, the last one short
    try:
        out = asyncio.run(p.process_video(str(path), str(tmp_path)))
    finally:
        p.batch_size = old_bs
    assert out["status"] == "success" and len(out["results"]) == 11
    assert [r["frame_number"] for r in out["results"]] == list(range(11))
    exp = []
    for s in range(0, 11, 4):
        exp += p.process_device_batch(DeviceFrames(frames[s:s + 4]), [(i, i / 10.0) for i in range(s, min(s + 4, 11))])
    for got, want in zip(out["results"], exp):
        assert got["detections"] == want["detections"]
    assert out["summary"]["total_frames"] == 11
