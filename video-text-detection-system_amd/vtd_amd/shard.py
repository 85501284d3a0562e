"""Multi-GPU layout of the path: frames are independent, so rank r of W owns frames r, r+W, r+2W, ...
(weights replicated, no data-path collective).  The single exchange is the gather of the fixed-size detection
records (+ counts) so that the rank owning the result list sees every frame: one all_gather of a small padded block
per batch step -- latency-bound, ~100 KB, never a ring all-reduce.  Works on any torch.distributed backend
("nccl" = RCCL over xGMI on the GPUs; "gloo" in the CPU tests).
"""
import torch
import torch.distributed as dist


def frames_of_rank(n_frames, rank, world):
    """Global frame indices owned by `rank` (round-robin keeps decode order interleaved across GPUs)."""
    return list(range(rank, n_frames, world))


def gather_detections(records, counts, group=None):
    """records: [F, MAX_DET, 16] int32, counts: [F] int32 for this rank's F frames (same F on every rank, pad with
    count 0).  Returns ([W, F, MAX_DET, 16], [W, F]) on every rank."""
    world = dist.get_world_size(group)
    f = records.shape[0]
    # concatenated-along-dim-0 output form: the one every backend (RCCL and gloo) accepts
    rec_out = torch.empty((world * f,) + tuple(records.shape[1:]), dtype=records.dtype, device=records.device)
    cnt_out = torch.empty((world * f,), dtype=counts.dtype, device=counts.device)
    dist.all_gather_into_tensor(rec_out, records.contiguous(), group=group)
    dist.all_gather_into_tensor(cnt_out, counts.contiguous(), group=group)
    return rec_out.view((world, f) + tuple(records.shape[1:])), cnt_out.view(world, f)


def merge_by_frame(rec_all, cnt_all, n_frames):
    """Undo the round-robin sharding: list over global frame index of that frame's [count,16] record block."""
    world, per_rank = cnt_all.shape
    out = []
    for g in range(n_frames):
        r, i = g % world, g // world
        c = int(cnt_all[r, i])
        out.append(rec_all[r, i, :min(c, rec_all.shape[2])])
    return out
