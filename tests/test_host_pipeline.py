"""Host logic of VideoTextPipeline on CPU with the reference's own mock seams: rows H1-H3 of SURVEY 8(a), pinned by
tests/golden/pipeline_harness.json (produced by the reference's unmodified pipeliine.py).  No GPU: the pipeline
object is built with __new__ + attribute injection exactly as the survey's recipe does for the reference."""
import asyncio
import json
import os
from concurrent.futures import ThreadPoolExecutor
from unittest.mock import Mock

import numpy as np
import pytest

from vtd_amd.pipeline import VideoTextPipeline


@pytest.fixture
def harness(golden_dir):
    return json.load(open(os.path.join(golden_dir, "pipeline_harness.json")))


@pytest.fixture
def pipe():
    p = VideoTextPipeline.__new__(VideoTextPipeline)
    p.confidence_threshold = 0.5
    p.batch_size = 16
    p.executor = ThreadPoolExecutor(max_workers=4)
    p.detector = Mock()
    p.recognizer = Mock()
    return p


DETS = [{"bbox": [50, 80, 200, 120], "confidence": 0.8, "polygon": [[50, 80], [200, 80], [200, 120], [50, 120]]},
        {"bbox": [10, 10, 10, 40], "confidence": 0.7}]


def test_h1_process_single_frame(pipe, harness):
    pipe.detector.detect.return_value = DETS
    pipe.recognizer.recognize.return_value = {"text": " TEST TEXT ", "confidence": 0.9}
    frame = np.zeros((480, 640, 3), np.uint8)
    assert pipe.process_single_frame(frame) == harness["H1"]
    assert list(pipe.recognizer.recognize.call_args[0][0].shape) == harness["H1_crop_shape"]  # the zero-width crop is skipped
    pipe.detector.detect.assert_called_once_with(frame, 0.5)
    pipe.detector.detect.side_effect = Exception("boom")
    assert pipe.process_single_frame(frame) == harness["H1_error"]
    pipe.detector.detect.side_effect = None
    pipe.detector.detect.return_value = []
    assert pipe.process_single_frame(frame) == {"detections": []}


def test_h2_process_frame_batch(pipe, harness):
    frame = np.zeros((480, 640, 3), np.uint8)
    pipe.detector.detect.return_value = DETS
    pipe.recognizer.recognize.return_value = {"text": " TEST TEXT ", "confidence": 0.9}
    got = asyncio.run(pipe._process_frame_batch([frame, frame], [(0, 0.0), (1, 0.1)], "/tmp"))
    assert got == harness["H2"]
    pipe.detector.detect.return_value = [{"bbox": [50, 80, 200, 120], "confidence": 0.8}]
    assert asyncio.run(pipe._process_frame_batch([frame], [(5, 0.5)], "/tmp")) == harness["H2_no_polygon"]
    pipe.detector.detect.return_value = []
    assert asyncio.run(pipe._process_frame_batch([frame], [(6, 0.6)], "/tmp")) == harness["H2_no_detections"]


def test_h3_generate_summary(pipe, harness):
    got = pipe._generate_summary(harness["H3_input"], 2.0, 3)
    exp = dict(harness["H3"])
    assert set(got.pop("detected_texts")) == set(exp.pop("detected_texts"))
    assert got == exp
    assert pipe._generate_summary([], 0.0, 0) == harness["H3_empty"]
    json.dumps(got)  # the summary goes through Celery's JSON serializer


def test_process_video_loop_batches_progress_and_failure(pipe, tmp_path):
    from vtd_amd.video import VideoProcessor
    frames = np.zeros((90, 48, 64, 3), np.uint8)  # 90 frames @30 fps, the shape of tests/test_integration.py:17-35
    path = str(tmp_path / "clip.npy")
    np.save(path, frames)
    pipe.video_processor = VideoProcessor()
    pipe.batch_size = 4
    pipe.detector.detect.return_value = [{"bbox": [5, 5, 40, 30], "confidence": 0.9, "polygon": []}]
    pipe.recognizer.recognize.return_value = {"text": "HELLO WORLD", "confidence": 0.95}
    calls = []

    async def progress(p, done, total):
        calls.append((p, done, total))

    out = asyncio.run(pipe.process_video(path, str(tmp_path), progress))
    assert out["status"] == "success"
    assert out["video_info"] == {"fps": 30.0, "frame_count": 90, "width": 64, "height": 48, "duration": 3.0, "format": ".npy"}
    assert len(out["results"]) == 30  # sampling rule: every int(30/10) = 3rd frame
    assert [r["frame_number"] for r in out["results"]] == list(range(30))
    assert abs(out["results"][1]["timestamp"] - 0.1) < 1e-9
    assert out["summary"]["total_detections"] == 30 and out["summary"]["detected_texts"] == ["HELLO WORLD"]
    assert calls and calls[0] == (4 / 90, 4, 90) and all(c[1] % 4 == 0 for c in calls)
    json.dumps(out)
    bad = asyncio.run(pipe.process_video(str(tmp_path / "missing.npy"), str(tmp_path)))
    assert bad["status"] == "success" and bad["results"] == [] and bad["video_info"] == {}  # same as the reference: errors are logged
    pipe.video_processor = Mock()
    pipe.video_processor.get_video_info.side_effect = RuntimeError("decoder exploded")
    failed = asyncio.run(pipe.process_video(path, str(tmp_path)))
    assert failed == {"status": "failed", "error": "decoder exploded", "results": []}


def test_product_has_no_cpu_fallback_and_overlay_imports():
    """Without a GPU the model classes must refuse to construct (no silent CPU path)."""
    import torch
    from vtd_amd import _native
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from vtd_amd.detector import TextDetector
    from vtd_amd.recognizer import TextRecognizer
    with pytest.raises(_native.NativeError):
        TextDetector()
    with pytest.raises(_native.NativeError):
        TextRecognizer(use_transformer=False)
    with pytest.raises(_native.NativeError):
        VideoTextPipeline(use_transformer_ocr=False)
    import app.ml as overlay  # the names the Celery worker and the reference tests import
    from app.ml.inference.pipeline import VideoTextPipeline as P2
    from app.ml.models.text_detector import DBNet, TextDetector as T2
    from app.ml.models.text_recognizer import CRNN, TextRecognizer as R2
    assert P2 is VideoTextPipeline and T2 is TextDetector and R2 is TextRecognizer and overlay.DBNet is DBNet and overlay.CRNN is CRNN


def test_parameter_containers_deepcopy_and_pickle():
    """DBNet / CRNN are ordinary modules to their users: EMA copies (copy.deepcopy), whole-model torch.save and spawn pickling
    must work; the engine lock and the native handle stay behind and are rebuilt by the copy at first use."""
    import copy
    import io
    import pickle

    import torch

    from vtd_amd.nets import CRNN, DBNet
    for m in (DBNet("resnet18"), CRNN(97)):
        m.load_state_dict(m.state_dict())                 # a model that has been through the public API (version bumped)
        dup, rt = copy.deepcopy(m), pickle.loads(pickle.dumps(m))
        buf = io.BytesIO()
        torch.save(m, buf)
        for other in (dup, rt):
            assert other._engine is None and other._engine_version == -1
            assert other._engine_lock is not m._engine_lock
            assert other._version == m._version
            for (k, a), (k2, b) in zip(m.state_dict().items(), other.state_dict().items()):
                assert k == k2 and torch.equal(a, b) and a.data_ptr() != b.data_ptr()
        dup.mark_dirty()
        assert dup._version == m._version + 1             # the copy's bookkeeping is its own


def test_quiet_gc_hands_the_heap_back_and_respects_a_host_freeze(monkeypatch):
    """process_video runs its loop with the long-lived heap frozen out of the cyclic collector (result dicts are cycle-free; full passes
    over modules and state dicts cost 12 % of the sustained rate) and undoes that on exit -- unless the host application froze objects
    itself, in which case nothing is handed back behind its back.  VTD_QUIET_GC=0 leaves the collector alone."""
    import gc

    from vtd_amd.pipeline import quiet_gc
    assert gc.get_freeze_count() == 0
    with quiet_gc():
        assert gc.get_freeze_count() > 0 and gc.isenabled()
    assert gc.get_freeze_count() == 0
    monkeypatch.setenv("VTD_QUIET_GC", "0")
    with quiet_gc():
        assert gc.get_freeze_count() == 0
    monkeypatch.delenv("VTD_QUIET_GC")
    # nested / concurrent entries: only the LAST one to leave hands the heap back
    outer, inner = quiet_gc(), quiet_gc()
    outer.__enter__()
    inner.__enter__()
    outer.__exit__(None, None, None)         # the first call to finish must not thaw the heap under the other
    assert gc.get_freeze_count() > 0
    inner.__exit__(None, None, None)
    assert gc.get_freeze_count() == 0
    gc.freeze()                          # the host's own freeze (e.g. a pre-fork server)
    try:
        before = gc.get_freeze_count()
        junk = [[] for _ in range(5000)]     # alive at entry: a freeze of ours on top of the host's would move these in for good
        with quiet_gc():
            assert gc.get_freeze_count() <= before   # nothing added to the permanent generation (it may shrink: frozen objects
        assert 0 < gc.get_freeze_count() <= before   # are still freed by reference counting) and nothing handed back either
        del junk
    finally:
        gc.unfreeze()


def test_mixed_size_batches_come_back_in_frame_order(pipe):
    """_pipeline_push groups a batch by frame shape (one device pass per size, cut to the engine's max_batch), the jobs ride the
    three-deep pipeline, and every pushed batch is handed back whole and in frame order when its last group retires.  Host logic only:
    the device halves are stubbed."""
    class _Eng:
        max_batch = 2

    class _Model:
        def engine(self):
            return _Eng()

    pipe.detector.model = _Model()
    pipe._upload = object()
    pipe._bind_device = lambda: None
    pipe._stage = lambda chunk: (list(chunk), None)
    pipe.submit_detection = lambda batch: {"batch": batch}
    pipe._try_recognition = lambda job: job.setdefault("rec", True)
    pipe.collect = lambda job, info: [{"frame_number": n, "timestamp": t, "detections": [], "shape": tuple(f.shape)}
                                      for (n, t), f in zip(info, job["batch"])]
    sizes = [(4, 6), (8, 6), (4, 6), (4, 6), (8, 6), (2, 2), (4, 6)]
    frames = [np.zeros((h, w, 3), np.uint8) for h, w in sizes]
    assert [len(g) for g in pipe._shape_groups(frames)] == [4, 2, 1]
    out = pipe._pipeline_push(frames[:5], [(i, i / 10) for i in range(5)])      # 3 + 2 frames -> jobs of 2, 1, 2
    out += pipe._pipeline_push(frames[5:], [(i, i / 10) for i in range(5, 7)])  # 2 more jobs
    out += pipe._pipeline_drain()
    assert [r["frame_number"] for r in out] == list(range(7))
    assert [r["shape"][:2] for r in out] == sizes
    assert pipe.route_counts == {"device": 7, "reference": 0}


def test_failing_first_group_of_a_mixed_batch_yields_every_frame_once(pipe):
    """Round-3 advisor finding: with two shape groups in one pushed batch and the FIRST group's staging raising, that job retired at
    once while it was the batch's only counted job, the batch looked complete and its frames came back twice ([0, 2, 0, 1, 2, 3],
    all 'success').  Every frame must come back exactly once, in order, the failed group's with empty detections -- also when the
    failing job is the last group, and when the detector enqueue (not the staging) is what fails."""
    class _Eng:
        max_batch = 8

    class _Model:
        def engine(self):
            return _Eng()

    def run(fail_shape, fail_in):
        pipe.__dict__.pop("_inflight", None)
        pipe.__dict__.pop("route_counts", None)
        pipe.detector.model = _Model()
        pipe._upload = object()
        pipe._bind_device = lambda: None

        def stage(chunk):
            if fail_in == "stage" and chunk[0].shape[0] == fail_shape:
                raise RuntimeError("staging failed")
            return list(chunk), None

        def detect(batch):
            if fail_in == "detect" and batch[0].shape[0] == fail_shape:
                raise RuntimeError("enqueue failed")
            return {"batch": batch}

        pipe._stage, pipe.submit_detection = stage, detect
        pipe._try_recognition = lambda job: job.setdefault("rec", True)
        pipe.collect = lambda job, info: [{"frame_number": n, "timestamp": t, "detections": ["box"]} for n, t in info]
        sizes = [(4, 6), (8, 6), (4, 6), (8, 6)]                      # two shape groups: frames {0, 2} and {1, 3}
        frames = [np.zeros((h, w, 3), np.uint8) for h, w in sizes]
        out = pipe._pipeline_push(frames, [(i, i / 10) for i in range(4)])
        out += pipe._pipeline_push(frames, [(i, i / 10) for i in range(4, 8)])
        out += pipe._pipeline_drain()
        assert [r["frame_number"] for r in out] == list(range(8)), (fail_shape, fail_in, [r["frame_number"] for r in out])
        for r in out:
            failed = sizes[r["frame_number"] % 4][0] == fail_shape
            assert r["detections"] == ([] if failed else ["box"]), (fail_shape, fail_in, r)

    for fail_in in ("stage", "detect"):
        run(4, fail_in)   # the first group of every batch fails
        run(8, fail_in)   # the last group fails


def test_swallowed_errors_are_counted(pipe):
    """The reference's error convention turns a failing detector batch into empty detections and a failing recogniser into empty strings
    (log and go on).  A caller that must not read such a run as a result -- bench.py, which once posted 662 frames/s for a recogniser that
    failed every pass -- finds the count in ``error_counts``."""
    class _Eng:
        max_batch = 8

    class _Model:
        def engine(self):
            return _Eng()

    pipe.__dict__.pop("_inflight", None)
    pipe.__dict__.pop("error_counts", None)
    pipe.detector.model = _Model()
    pipe._upload = object()
    pipe._bind_device = lambda: None
    pipe._stage = lambda chunk: (list(chunk), None)
    calls = {"n": 0}

    def detect(batch):
        calls["n"] += 1
        if calls["n"] == 2:
            raise RuntimeError("enqueue failed")
        return {"batch": batch}

    pipe.submit_detection = detect
    pipe._try_recognition = lambda job: job.setdefault("rec", True)

    def collect(job, info):
        if int(job["batch"][0][0, 0, 0]) == 3:
            raise RuntimeError("collection failed")
        return [{"frame_number": n, "timestamp": t, "detections": ["box"]} for n, t in info]

    pipe.collect = collect
    out = []
    for k in range(5):
        out += pipe._pipeline_push([np.full((4, 6, 3), k, np.uint8)], [(k, 0.0)])
    out += pipe._pipeline_drain()
    assert [r["frame_number"] for r in out] == list(range(5))
    assert [bool(r["detections"]) for r in out] == [True, False, True, False, True]
    assert pipe.error_counts == {"detection": 1, "collection": 1}


def test_host_frames_enter_the_detector_one_interval_after_their_copy(pipe, monkeypatch):
    """Host frames are staged (pinned buffer + copy on the upload stream) when pushed and enter the detector when the NEXT job is
    pushed, so the detector never waits for its own batch's copy; every stage advances one job per push, results stay whole and in
    order, the drain flushes a batch that never got its interval, and VTD_STAGE_AHEAD=0 restores detect-at-push.  Host logic only."""
    from vtd_amd import engine

    class _Eng:
        max_batch = 8

    class _Model:
        def engine(self):
            return _Eng()

    class _Pinned:
        released = []

        def release(self, h):
            self.released.append(h)

    pinned = _Pinned()
    monkeypatch.setattr(engine, "PINNED", pinned)
    log = []
    pipe.detector.model = _Model()
    pipe._upload = object()
    pipe._bind_device = lambda: None

    def stage(chunk):
        k = int(chunk[0][0, 0, 0])
        log.append(("stage", k))
        return list(chunk), f"pinned{k}"

    def detect(batch):
        log.append(("detect", int(batch[0][0, 0, 0])))
        return {"batch": batch}

    def recognise(job):
        if "rec" not in job and not job.get("failed"):
            log.append(("recognise", int(job["batch"][0][0, 0, 0])))
            job["rec"] = True

    def collect(job, info):
        log.append(("collect", int(job["batch"][0][0, 0, 0])))
        return [{"frame_number": n, "timestamp": t, "detections": []} for n, t in info]

    pipe._stage, pipe.submit_detection, pipe._try_recognition, pipe.collect = stage, detect, recognise, collect
    batches = [[np.full((4, 6, 3), k, np.uint8) for _ in range(2)] for k in range(5)]
    outs = [pipe._pipeline_push(b, [(2 * k, 0.0), (2 * k + 1, 0.0)]) for k, b in enumerate(batches)]
    assert [len(o) for o in outs] == [0, 0, 0, 2, 2]                      # batch k comes out with push k + 3
    assert log[:3] == [("stage", 0), ("stage", 1), ("detect", 0)]         # batch 0 enters the detector behind batch 1's copy
    assert log.index(("detect", 3)) > log.index(("stage", 4)) and log.index(("recognise", 2)) > log.index(("detect", 3))
    rest = pipe._pipeline_drain()
    assert [r["frame_number"] for o in outs for r in o] + [r["frame_number"] for r in rest] == list(range(10))
    assert ("detect", 4) in log and log.index(("detect", 4)) > log.index(("collect", 1))   # the drain enqueues the batch still staged
    assert sorted(pinned.released) == [f"pinned{k}" for k in range(5)]

    log.clear()
    monkeypatch.setenv("VTD_STAGE_AHEAD", "0")
    outs = [pipe._pipeline_push(b, [(2 * k, 0.0), (2 * k + 1, 0.0)]) for k, b in enumerate(batches[:3])]
    assert log[:2] == [("stage", 0), ("detect", 0)] and [len(o) for o in outs] == [0, 0, 2]
    assert len(pipe._pipeline_drain()) == 4
