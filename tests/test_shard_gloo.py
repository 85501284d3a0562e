"""N>1 path on CPU (gloo, world size 2, 127.0.0.1):

* record blocks in the `vtd_detection` layout vtd_postproc_run writes ([F, MAX_DET, 16] int32: bbox, polygon, float32
  confidence / area bits, first pixel -- include/vtd.h) are sharded, all-gathered and merged back in frame order, and
  `records_to_dicts` of the merged blocks equals `records_to_dicts` of the originals;
* the whole-result block codec round-trips every result shape the pipeline produces;
* `VideoTextPipeline.process_video` in its rank-aware mode (frame i -> rank i mod W, one gather per round, rank 0 owns the
  merged ordered result) returns exactly the single-process result -- driven through the reference's own mock seams
  (detector.detect / recognizer.recognize), so it runs without a GPU.
"""
import asyncio
import json
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vtd_amd import shard

N_FRAMES, MAX_DET = 7, 4


def _postproc_records(frame_idx):
    """One frame's record block exactly as vtd_postproc_run lays it out, via the layout code's own inverse."""
    rng = np.random.default_rng(1000 + frame_idx)
    n = frame_idx % 3 + (1 if frame_idx == 5 else 0)
    rec = np.zeros((MAX_DET, 16), np.int32)
    for k in range(n):
        x1, y1 = int(rng.integers(0, 600)), int(rng.integers(0, 300))
        rec[k, 0:4] = [x1, y1, x1 + int(rng.integers(11, 300)), y1 + int(rng.integers(11, 100))]
        rec[k, 4:12] = rng.integers(0, 640, 8)
        rec[k, 12] = np.float32(rng.random()).view(np.int32)          # confidence bits
        rec[k, 13] = np.float32(100 + 1000 * rng.random()).view(np.int32)   # area bits
        rec[k, 14:16] = rng.integers(0, 640, 2)
    return rec, n


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _init(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _records_worker(rank, world, port, out_dir):
    _init(rank, world, port)
    mine = shard.frames_of_rank(N_FRAMES, rank, world)
    per_rank = (N_FRAMES + world - 1) // world
    records = torch.zeros((per_rank, MAX_DET, 16), dtype=torch.int32)
    counts = torch.zeros((per_rank,), dtype=torch.int32)
    for i, g in enumerate(mine):
        rec, n = _postproc_records(g)
        records[i], counts[i] = torch.from_numpy(rec), n
    rec_all, cnt_all = shard.gather_detections(records, counts)
    merged = shard.merge_by_frame(rec_all, cnt_all, N_FRAMES)
    torch.save([m.clone() for m in merged], os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_gather_of_postproc_records(tmp_path):
    from vtd_amd.engine import records_to_dicts
    mp.spawn(_records_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert shard.frames_of_rank(7, 0, 2) == [0, 2, 4, 6] and shard.frames_of_rank(7, 1, 2) == [1, 3, 5]
    for rank in range(2):  # every rank ends up with the full, ordered result
        merged = torch.load(os.path.join(tmp_path, f"rank{rank}.pt"))
        assert len(merged) == N_FRAMES
        for g, block in enumerate(merged):
            exp, n = _postproc_records(g)
            assert block.shape[0] == n and np.array_equal(block.numpy(), exp[:n])
            got = records_to_dicts(block.numpy(), debug=True)
            assert got == records_to_dicts(exp[:n], debug=True)
            for d in got:
                json.dumps(d)
                assert isinstance(d["confidence"], float) and all(isinstance(v, int) for v in d["bbox"])


def _sample_results():
    return [
        {"frame_number": 0, "timestamp": 0.0, "detections": []},
        {"frame_number": 3, "timestamp": 0.30000000000000004, "detections": [
            {"bbox": [50, 80, 200, 120], "text": " TEST TEXT ", "detection_confidence": 0.8, "recognition_confidence": 0.9,
             "polygon": [[50, 80], [200, 80], [200, 120], [50, 120]]},
            {"bbox": [1, 2, 30, 40], "text": "", "detection_confidence": float(np.float32(0.123456)), "recognition_confidence": 0.0,
             "polygon": []},
            {"bbox": [0, 0, 1279, 719], "text": "~\\\"{|}", "detection_confidence": 1.0, "recognition_confidence": 0.9853782176971435,
             "polygon": [[0, 0], [639, 0], [639, 639], [0, 639]]}]},
        {"frame_number": 7, "timestamp": 123.456, "detections": [
            {"bbox": [5, 5, 50, 50], "text": "x" * 31, "detection_confidence": 0.5, "recognition_confidence": 0.25,
             "polygon": [[1, 2], [3, 4], [5, 6], [7, 8]]}]},
    ]


def test_result_block_round_trip():
    res = _sample_results()
    det, txt = shard.required_caps(res)
    assert (det, txt) == (3, 31)
    blk = shard.pack_results(res, frames=5, max_det=8, text_cap=32)
    assert blk.dtype == np.int32 and blk.shape == shard.block_shape(5, 8, 32)
    back = shard.unpack_results(blk)
    assert back == res
    assert json.dumps(back) == json.dumps(res)
    for bad in (dict(frames=2, max_det=8, text_cap=32), dict(frames=5, max_det=2, text_cap=32), dict(frames=5, max_det=8, text_cap=8)):
        try:
            shard.pack_results(res, **bad)
        except OverflowError:
            continue
        raise AssertionError(f"no overflow for {bad}")


# ---- rank-aware process_video through the reference's mock seams ------------------------------------------------------
N_VIDEO = 23


class _Source:
    """VideoProcessor seam: 23 frames of 48x64; frame k is filled with k so the stand-in detector can tell frames apart."""

    def get_video_info(self, path):
        return {"fps": 10.0, "frame_count": N_VIDEO, "width": 64, "height": 48, "duration": N_VIDEO / 10.0, "format": ".npy"}

    async def extract_frames_generator(self, path, target_fps=10):
        for k in range(N_VIDEO):
            yield np.full((48, 64, 3), k, np.uint8), k, k / 10.0
            await asyncio.sleep(0)


def _detect(frame, threshold=0.5):
    k = int(frame[0, 0, 0])
    return [{"bbox": [j, j, 20 + j + k, 30 + j], "confidence": 0.5 + 0.01 * j + 0.001 * k,
             "polygon": [[j, j], [20 + j + k, j], [20 + j + k, 30 + j], [j, 30 + j]]} for j in range(k % 4)]


def _recognize(crop):
    return {"text": f"w{crop.shape[1]}h{crop.shape[0]}", "confidence": 0.9 - 0.001 * crop.shape[1]}


def _mock_pipeline():
    from concurrent.futures import ThreadPoolExecutor
    from unittest.mock import Mock
    from vtd_amd.pipeline import VideoTextPipeline
    p = VideoTextPipeline.__new__(VideoTextPipeline)
    p.confidence_threshold, p.batch_size = 0.5, 4
    p.executor = ThreadPoolExecutor(max_workers=4)
    p.detector, p.recognizer = Mock(), Mock()
    p.detector.detect.side_effect = _detect
    p.recognizer.recognize.side_effect = _recognize
    p.video_processor = _Source()
    return p


def _video_worker(rank, world, port, out_dir):
    _init(rank, world, port)
    p = _mock_pipeline()
    progress = []

    async def cb(frac, done, total):
        progress.append((done, total))

    out = asyncio.run(p.process_video("clip.npy", out_dir, cb))
    out["_progress"] = progress
    out["_detect_calls"] = p.detector.detect.call_count
    json.dump(out, open(os.path.join(out_dir, f"video_rank{rank}.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()


def test_rank_aware_process_video_equals_single_process(tmp_path):
    single = asyncio.run(_mock_pipeline().process_video("clip.npy", str(tmp_path)))
    assert single["status"] == "success" and len(single["results"]) == N_VIDEO and "shard" not in single
    mp.spawn(_video_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0 = json.load(open(tmp_path / "video_rank0.json"))
    r1 = json.load(open(tmp_path / "video_rank1.json"))
    assert r0["status"] == r1["status"] == "success"
    assert r0["shard"] == {"rank": 0, "world_size": 2} and r1["shard"] == {"rank": 1, "world_size": 2}
    assert r0["results"] == json.loads(json.dumps(single["results"]))       # merged, ordered, identical
    assert [fr["frame_number"] for fr in r0["results"]] == list(range(N_VIDEO))
    assert r1["results"] == []
    # the work really was split: rank r ran detect on frames r, r+2, ...
    assert r0["_detect_calls"] == 12 and r1["_detect_calls"] == 11
    for key in ("total_frames", "frames_with_text", "total_detections", "unique_texts", "avg_detection_confidence",
                "avg_recognition_confidence"):
        assert r0["summary"][key] == single["summary"][key], key
    assert r0["_progress"] == r1["_progress"] and r0["_progress"][-1][0] in (16, N_VIDEO)


def _failing_worker(rank, world, port, out_dir, fail_rank, fail_frame):
    import time
    _init(rank, world, port)
    p = _mock_pipeline()
    if rank == fail_rank:
        def boom(frame, threshold=0.5):
            if int(frame[0, 0, 0]) == fail_frame:
                raise RuntimeError("injected decode / device failure")
            return _detect(frame, threshold)
        p.detector.detect.side_effect = boom
    t0 = time.time()
    out = asyncio.run(p.process_video("clip.npy", out_dir))
    out["_seconds"] = time.time() - t0
    json.dump(out, open(os.path.join(out_dir, f"fail_rank{rank}.json"), "w"))
    dist.barrier()    # both ranks are still alive and in step after the abort
    dist.destroy_process_group()


def test_a_failing_rank_takes_every_rank_out_of_the_loop_together(tmp_path):
    """A rank whose loop raises (here: detect on frame 9, in the third round) tells its peers through the error flag of the next
    capacity all_reduce: every rank returns the reference's 'failed' dict within seconds, nobody waits for the backend timeout
    (30 min on gloo), and the process group is still usable afterwards.  Same when the failure hits in the very last round."""
    for fail_rank, fail_frame in ((1, 9), (0, 22)):
        mp.spawn(_failing_worker, args=(2, _free_port(), str(tmp_path), fail_rank, fail_frame), nprocs=2, join=True)
        for rank in range(2):
            out = json.load(open(tmp_path / f"fail_rank{rank}.json"))
            assert out["status"] == "failed" and out["results"] == [] and out["_seconds"] < 60, out
            assert ("injected" in out["error"]) == (rank == fail_rank)
            assert ("peer rank failed" in out["error"]) == (rank != fail_rank)


def test_every_rank_returns_rank0_summary(tmp_path):
    mp.spawn(_video_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0 = json.load(open(tmp_path / "video_rank0.json"))
    r1 = json.load(open(tmp_path / "video_rank1.json"))
    assert r1["summary"] == r0["summary"] and r0["summary"]["total_detections"] > 0


# ---- kernel-selection tables travel from rank 0 to every rank -----------------------------------------------------------
class _FakeEngine:
    def __init__(self, text):
        self.text = text

    def tuning_text(self):
        return self.text

    def set_tuning(self, text):
        self.text = text


def _tuning_worker(rank, world, port, out_dir):
    _init(rank, world, port)
    engines = [_FakeEngine(f"conv|a|n4 {3 + rank}\n"), _FakeEngine(f"conv|b|n8 {100 + rank}\n")]
    shard.sync_tuning(engines)
    json.dump([e.text for e in engines], open(os.path.join(out_dir, f"tuning_rank{rank}.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()


def test_rank0_tuning_table_reaches_every_rank(tmp_path):
    mp.spawn(_tuning_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    t0 = json.load(open(tmp_path / "tuning_rank0.json"))
    t1 = json.load(open(tmp_path / "tuning_rank1.json"))
    assert t0 == t1 == ["conv|a|n4 3\n", "conv|b|n8 100\n"]


def test_frame_source_skips_the_frames_of_other_ranks(tmp_path):
    """VideoProcessor.set_shard: numbering and timestamps stay whole-clip, the frames of other ranks come out as None without being
    read, set_shard(0, 1) restores the plain source (the rank-aware process_video sets and resets it)."""
    from vtd_amd.video import VideoProcessor
    frames = np.arange(9 * 4 * 6 * 3, dtype=np.uint8).reshape(9, 4, 6, 3)
    clip = str(tmp_path / "c.npy")
    np.save(clip, frames)
    open(clip + ".json", "w").write(json.dumps({"fps": 30.0}))   # interval 3: source frames 0, 3, 6
    vp = VideoProcessor()
    whole = list(vp.extract_frames_at_fps(clip, 10))
    assert [(n, round(t, 6)) for _, n, t in whole] == [(0, 0.0), (1, 0.1), (2, 0.2)] and all(f is not None for f, _, _ in whole)
    vp.set_shard(1, 2)
    mine = list(vp.extract_frames_at_fps(clip, 10))
    assert [(f is None, n, round(t, 6)) for f, n, t in mine] == [(True, 0, 0.0), (False, 1, 0.1), (True, 2, 0.2)]
    assert np.array_equal(mine[1][0], frames[3])
    vp.set_shard(0, 1)
    assert all(f is not None for f, _, _ in vp.extract_frames_at_fps(clip, 10))
