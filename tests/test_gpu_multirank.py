"""N>1 on the GPU box's single card: two ranks share GPU 0 over gloo (RCCL refuses two ranks on one device; the data path is
identical, only the transport of the one all-gather differs).

* `python bench.py --gpus 2` (self-launched ranks) runs the product pipeline on both ranks, gathers the detection records of every
  step and prints one JSON line with the whole-job rate.
* `VideoTextPipeline.process_video` in its rank-aware mode on real device pipelines: rank 0's merged result equals the
  single-process result of the same clip, frame for frame."""
import asyncio
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from vtd_amd._fixtures import synth, weights

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    e.update(kw)
    return e


def test_bench_two_ranks_on_one_gpu(hip):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "8",
                        "--cpu-seconds", "0", "--no-profile"], env=_env(VTD_DIST_BACKEND="gloo", VTD_MAX_CROPS="128"),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 16 and out["scaling"] == "weak"
    assert out["value"] > 0 and out["config"]["detections_last_step_rank0"] > 0


_RANK = r"""
import asyncio, json, os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
import torch, torch.distributed as dist
from vtd_amd._fixtures import weights
from vtd_amd.pipeline import VideoTextPipeline
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
if world > 1:
    dist.init_process_group("gloo", rank=rank, world_size=world)
p = VideoTextPipeline(use_transformer_ocr=False, backbone="resnet18", batch_size=4)
p.detector.model.load_state_dict(weights.margin_detector_state_dict("resnet18", 0))
p.recognizer.model.load_state_dict(weights.margin_crnn_state_dict(11))
out = asyncio.run(p.process_video(sys.argv[3], os.path.dirname(sys.argv[3])))
json.dump(out, open(sys.argv[3] + f".rank{rank}of{world}.json", "w"))
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
"""


def test_rank_aware_process_video_on_device(hip, tmp_path):
    frames = np.stack([synth.text_frame(900 + i)[0] for i in range(13)])
    clip = str(tmp_path / "clip.npy")
    np.save(clip, frames)
    open(clip + ".json", "w").write(json.dumps({"fps": 10.0}))
    args = [sys.executable, "-c", _RANK, os.path.join(ROOT, "video-text-detection-system_amd"), ROOT, clip]
    single = subprocess.run(args, env=_env(RANK="0", WORLD_SIZE="1"), capture_output=True, text=True, timeout=600)
    assert single.returncode == 0, single.stderr[-3000:]
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = str(s.getsockname()[1]); s.close()
    procs = [subprocess.Popen(args, env=_env(RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=port),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-2000:] for o in outs]
    want = json.load(open(clip + ".rank0of1.json"))
    r0 = json.load(open(clip + ".rank0of2.json"))
    r1 = json.load(open(clip + ".rank1of2.json"))
    assert want["status"] == r0["status"] == r1["status"] == "success"
    assert [fr["frame_number"] for fr in r0["results"]] == list(range(13))
    assert r0["results"] == want["results"] and r1["results"] == []
    assert sum(len(fr["detections"]) for fr in r0["results"]) > 50
    assert r0["shard"] == {"rank": 0, "world_size": 2}
    assert r0["summary"]["total_detections"] == want["summary"]["total_detections"]


def test_vtd_gather_c_entry_single_rank(hip):
    """include/vtd_comm.h: unique id -> communicator -> all-gather of a record block on the caller's stream, world size 1 (the only
    RCCL world one GPU can host: the two-rank tests above run over gloo); the block comes back unchanged and the handle reports
    its rank / world size."""
    import ctypes as C
    import torch
    from vtd_amd import _native_comm
    lib = _native_comm.load()
    ident = (C.c_ubyte * _native_comm.ID_BYTES)()
    _native_comm.check(lib.vtd_comm_unique_id(ident), "vtd_comm_unique_id")
    comm = C.c_void_p()
    _native_comm.check(lib.vtd_comm_create(ident, 0, 1, C.byref(comm)), "vtd_comm_create")
    try:
        assert lib.vtd_comm_rank(comm) == 0 and lib.vtd_comm_world_size(comm) == 1
        rec = torch.randint(-2 ** 31, 2 ** 31 - 1, (32, 64, 16), dtype=torch.int32, device="cuda")
        out = torch.zeros_like(rec)
        s = torch.cuda.current_stream()
        _native_comm.check(lib.vtd_gather(comm, C.c_void_p(rec.data_ptr()), rec.numel(), C.c_void_p(out.data_ptr()), C.c_void_p(s.cuda_stream)), "vtd_gather")
        s.synchronize()
        assert torch.equal(out, rec)
        assert lib.vtd_gather(comm, None, 4, C.c_void_p(out.data_ptr()), None) != 0
    finally:
        lib.vtd_comm_destroy(comm)
