"""Multi-GPU layout of the path: frames are independent, so rank r of W owns frames r, r+W, r+2W, ...
(weights replicated, no data-path collective).  The single exchange is the gather of fixed-size result blocks so that
the rank owning the result list sees every frame: one all_gather of a small padded block per round -- latency-bound,
never a ring all-reduce.  Works on any torch.distributed backend ("nccl" = RCCL over xGMI on the GPUs; "gloo" in the
CPU tests).

Two block formats:

* detection records only (``gather_detections``): the ``vtd_detection`` records ``[F, MAX_DET, 16] int32`` + counts
  exactly as ``vtd_postproc_run`` leaves them in HBM (include/vtd.h) -- what bench.py exchanges every step on the
  post-process stream.
* whole per-frame results (``pack_results`` / ``unpack_results``): what ``VideoTextPipeline.process_video`` exchanges
  in its rank-aware mode (BASELINE configs[3]): per frame a header row (valid, count, frame_number, timestamp bits) and
  per detection bbox, polygon, both confidences (float64 bits, exact for any Python float) and the text as code
  points.  ``unpack_results(pack_results(x)) == x`` for every result list the pipeline can produce.
"""
import numpy as np
import torch
import torch.distributed as dist

# int32 columns of one result row
_BBOX, _POLY, _NPOLY, _DCONF, _RCONF, _TLEN, _TEXT = 0, 4, 12, 13, 15, 17, 18
HDR_VALID, HDR_COUNT, HDR_FRAME, HDR_TS = 0, 1, 2, 3


def frames_of_rank(n_frames, rank, world):
    """Global frame indices owned by `rank` (round-robin keeps decode order interleaved across GPUs)."""
    return list(range(rank, n_frames, world))


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def comm_device():
    """Where collective buffers live: HBM for RCCL, host memory for gloo."""
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def sync_tuning(engines, group=None, src=0):
    """Rank `src`'s kernel-selection tables (whatever its engines measured on top of the shipped table) become every
    rank's: call it after rank `src` has run its priming pass and before the others run theirs, so all ranks launch the
    same kernels and produce bit-identical maps.  `engines`: list of DetectorEngine / RecognizerEngine, same order on
    every rank.  Returns the table texts."""
    texts = [e.tuning_text() for e in engines] if dist.get_rank(group) == src else [None] * len(engines)
    dist.broadcast_object_list(texts, src=src, group=group)
    if dist.get_rank(group) != src:
        for e, t in zip(engines, texts):
            e.set_tuning(t)
    return texts


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


# ---------------------------------------------------------------------------------------------- whole-result blocks
def block_shape(frames, max_det, text_cap):
    return (frames, max_det + 1, _TEXT + text_cap)


def _f64_bits(values):
    return np.asarray(values, dtype=np.float64).reshape(-1).view(np.int32).reshape(-1, 2)


def pack_results(results, frames, max_det, text_cap=48):
    """Per-frame result dicts (pipeliine.py:135-139 schema) -> int32 block [frames, max_det+1, 18+text_cap]; row 0 of a
    frame is its header.  Frames beyond len(results) are marked invalid.  Raises OverflowError when a frame has more
    than max_det detections or a text longer than text_cap (the caller then re-packs with larger caps on every rank)."""
    blk = np.zeros(block_shape(frames, max_det, text_cap), np.int32)
    if len(results) > frames:
        raise OverflowError("more frames than the block holds")
    for i, fr in enumerate(results):
        dets = fr["detections"]
        if len(dets) > max_det:
            raise OverflowError(f"{len(dets)} detections > block capacity {max_det}")
        hdr = blk[i, 0]
        hdr[HDR_VALID], hdr[HDR_COUNT], hdr[HDR_FRAME] = 1, len(dets), int(fr["frame_number"])
        hdr[HDR_TS:HDR_TS + 2] = _f64_bits([fr["timestamp"]])[0]
        if not dets:
            continue
        rows = blk[i, 1:1 + len(dets)]
        rows[:, _BBOX:_BBOX + 4] = np.asarray([d["bbox"] for d in dets], np.int32)
        for k, d in enumerate(dets):
            poly = d.get("polygon", [])
            if poly:
                flat = np.asarray(poly, np.int32).reshape(-1)
                if flat.size != 8:
                    raise OverflowError("polygon is not 4 points")
                rows[k, _POLY:_POLY + 8] = flat
                rows[k, _NPOLY] = 4
            codes = [ord(ch) for ch in d["text"]]
            if len(codes) > text_cap:
                raise OverflowError(f"text of {len(codes)} characters > block capacity {text_cap}")
            rows[k, _TLEN] = len(codes)
            rows[k, _TEXT:_TEXT + len(codes)] = codes
        rows[:, _DCONF:_DCONF + 2] = _f64_bits([d["detection_confidence"] for d in dets])
        rows[:, _RCONF:_RCONF + 2] = _f64_bits([d["recognition_confidence"] for d in dets])
    return blk


def unpack_results(blk):
    """Inverse of pack_results: the valid frames of a block, in block order."""
    out = []
    for i in range(blk.shape[0]):
        hdr = blk[i, 0]
        if not hdr[HDR_VALID]:
            continue
        n = int(hdr[HDR_COUNT])
        rows = np.ascontiguousarray(blk[i, 1:1 + n])
        ts = float(np.ascontiguousarray(hdr[HDR_TS:HDR_TS + 2]).view(np.float64)[0])
        dets = []
        if n:
            bbox = rows[:, _BBOX:_BBOX + 4].tolist()
            poly = rows[:, _POLY:_POLY + 8].reshape(n, 4, 2).tolist()
            npoly = rows[:, _NPOLY].tolist()
            dconf = np.ascontiguousarray(rows[:, _DCONF:_DCONF + 2]).view(np.float64).reshape(-1).tolist()
            rconf = np.ascontiguousarray(rows[:, _RCONF:_RCONF + 2]).view(np.float64).reshape(-1).tolist()
            tlen = rows[:, _TLEN].tolist()
            text = rows[:, _TEXT:].tolist()
            for k in range(n):
                dets.append({"bbox": bbox[k], "text": "".join(map(chr, text[k][:tlen[k]])), "detection_confidence": dconf[k],
                             "recognition_confidence": rconf[k], "polygon": poly[k] if npoly[k] else []})
        out.append({"frame_number": int(hdr[HDR_FRAME]), "timestamp": ts, "detections": dets})
    return out


def required_caps(results):
    """(max detections per frame, longest text) of a result list -- what a block must hold."""
    det = max((len(fr["detections"]) for fr in results), default=0)
    txt = max((len(d["text"]) for fr in results for d in fr["detections"]), default=0)
    return det, txt


class ShardAbort(RuntimeError):
    """A peer rank reported a failure at a sequence point: every rank leaves the video loop together."""


class ResultGather:
    """One all_gather of a result block per round of W x batch_size frames, asynchronous on its own stream; blocks are
    turned back into dicts when the round is retired (one round later, so the collective overlaps the next round's
    compute).  Capacities are agreed first with a tiny all_reduce(MAX), so no frame is ever truncated (the reference has
    no cap on detections per frame).

    Failure protocol: every sequence point (``submit``, ``finish``, ``abort``) opens with the SAME 4-int all_reduce(MAX)
    whose last slot is an error flag.  A rank that fails anywhere in its loop calls ``abort()`` -- one such all_reduce with
    the flag raised -- and leaves; the peers meet it at their next sequence point, see the flag and raise ``ShardAbort``
    instead of entering the all_gather, so nobody is left waiting in a collective until the backend's timeout."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.device = comm_device()
        self.stream = torch.cuda.Stream() if self.device.type == "cuda" else None
        self.pending = []
        self.closed = False

    def _agree(self, need):
        """all_reduce(MAX) of (frames, detections, text length, error flag); raises ShardAbort when any rank raised the flag."""
        caps = torch.tensor(need, dtype=torch.int32).to(self.device)
        dist.all_reduce(caps, op=dist.ReduceOp.MAX, group=self.group)
        frames, max_det, text_cap, flag = (int(v) for v in caps.cpu().tolist())   # small sync: the agreed capacities
        if flag:
            self.closed = True
            raise ShardAbort("a peer rank failed; leaving the sharded video loop on every rank")
        return frames, max_det, text_cap

    def submit(self, results):
        """results: the per-frame dicts this rank finished since the last round (possibly none).  Every rank calls this
        once per round, in the same order."""
        # packed to this rank's own needs BEFORE the first collective: whatever can raise here raises while the peers can
        # still be told (abort); between the all_reduce and the all_gather nothing can fail any more
        own_det, own_txt = required_caps(results)
        own = pack_results(results, max(len(results), 1), max(own_det, 1), max(own_txt, 1))
        ctx = torch.cuda.stream(self.stream) if self.stream is not None else _null()
        with ctx:
            frames, max_det, text_cap = self._agree((len(results), own_det, own_txt, 0))
            frames, max_det, text_cap = max(frames, 1), _round_up(max(max_det, 1), 8), _round_up(max(text_cap, 1), 16)
            blk = np.zeros(block_shape(frames, max_det, text_cap), np.int32)
            blk[:own.shape[0], :own.shape[1], :own.shape[2]] = own
            local = torch.from_numpy(blk)
            if self.device.type == "cuda":
                local = local.pin_memory().to(self.device, non_blocking=True)
            out = torch.empty((self.world,) + tuple(local.shape), dtype=torch.int32, device=self.device)
            dist.all_gather_into_tensor(out.view(-1, *local.shape[1:]), local, group=self.group)
            host = out.to("cpu", non_blocking=True) if self.device.type == "cuda" else out
            ev = None
            if self.stream is not None:
                ev = torch.cuda.Event()
                ev.record()
        self.pending.append((host, ev, local, out))

    def abort(self):
        """This rank failed: raise the flag at the peers' next sequence point (no-op once a peer's flag has been seen or the
        loop has finished -- nobody is waiting then)."""
        if self.closed:
            return
        self.closed = True
        ctx = torch.cuda.stream(self.stream) if self.stream is not None else _null()
        with ctx:
            caps = torch.tensor((0, 0, 0, 1), dtype=torch.int32).to(self.device)
            dist.all_reduce(caps, op=dist.ReduceOp.MAX, group=self.group)

    def finish(self, payload=None, src=0):
        """Last sequence point: all ranks agree that nobody failed, then rank `src`'s `payload` (the summary of the merged
        result) is broadcast so every rank returns the same one."""
        ctx = torch.cuda.stream(self.stream) if self.stream is not None else _null()
        with ctx:
            self._agree((0, 0, 0, 0))
            box = [payload if self.rank == src else None]
            dist.broadcast_object_list(box, src=src, group=self.group)
        self.closed = True
        return box[0]

    def retire(self, keep_last=0):
        """Results of every finished round except the newest `keep_last`, ordered by frame number inside a round."""
        done = []
        while len(self.pending) > keep_last:
            host, ev, _, _ = self.pending.pop(0)
            if ev is not None:
                ev.synchronize()
            merged = []
            for r in range(self.world):
                merged += unpack_results(host[r].numpy())
            merged.sort(key=lambda fr: fr["frame_number"])
            done += merged
        return done


def _round_up(v, m):
    return (v + m - 1) // m * m


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
