"""N>1 path on CPU: two gloo ranks shard a frame list, 'detect' with a deterministic stand-in, gather the padded
record blocks and rebuild the global per-frame order (the same functions bench.py uses over RCCL)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vtd_amd import shard

N_FRAMES, MAX_DET = 7, 4


def _fake_records(frame_idx):
    n = frame_idx % 3 + (1 if frame_idx == 5 else 0)
    rec = torch.zeros((MAX_DET, 16), dtype=torch.int32)
    for k in range(n):
        rec[k, :4] = torch.tensor([frame_idx, k, frame_idx * 10 + k, 99])
    return rec, n


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard.frames_of_rank(N_FRAMES, rank, world)
    per_rank = (N_FRAMES + world - 1) // world
    records = torch.zeros((per_rank, MAX_DET, 16), dtype=torch.int32)
    counts = torch.zeros((per_rank,), dtype=torch.int32)
    for i, g in enumerate(mine):
        records[i], counts[i] = _fake_records(g)
    rec_all, cnt_all = shard.gather_detections(records, counts)
    merged = shard.merge_by_frame(rec_all, cnt_all, N_FRAMES)
    torch.save([m.clone() for m in merged], os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_gather(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert shard.frames_of_rank(7, 0, 2) == [0, 2, 4, 6] and shard.frames_of_rank(7, 1, 2) == [1, 3, 5]
    for rank in range(2):  # every rank ends up with the full, ordered result
        merged = torch.load(os.path.join(tmp_path, f"rank{rank}.pt"))
        assert len(merged) == N_FRAMES
        for g, block in enumerate(merged):
            exp, n = _fake_records(g)
            assert block.shape[0] == n and torch.equal(block, exp[:n])
