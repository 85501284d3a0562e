"""``VideoTextPipeline`` with the reference's surface (app/ml/inference/pipeliine.py:17-210; imported by the Celery
worker as ``app.ml.inference.pipeline``, app/tasks/video_processing.py:12,33-37).

Same constructor, mutable ``confidence_threshold`` / ``batch_size`` attributes, ``process_video`` (async, async
progress callback), ``_process_frame_batch``, ``process_single_frame``, ``_generate_summary``, result schema (plain
Python containers -- the dict goes through Celery's JSON serializer and a JSON column) and error conventions.

What changes underneath: the reference fans each batch out as N=1 ``detect`` calls on four threads and recognises
crops one at a time; here a batch is ONE device pass -- fused preprocess -> DBNet -> post-process for all frames,
then every crop of the batch through crop/resize -> CRNN -> CTC decode -- with frames, maps, boxes and crops
resident in HBM.  The reference-shaped route is kept and taken whenever the test seams are in use
(``detector.detect`` / ``recognizer.recognize`` patched or replaced, tests/test_models.py:115-141,
tests/test_integration.py:54-79 of the reference) or a frame is not an HxWx3 uint8 array.  Frames of different sizes in one
batch (the reference takes any mix, pipeliine.py:96-101; BASELINE configs[4] alternates 720p / 1080p) stay on the device path:
the batch is grouped by frame shape, each group is one device pass, and the results are re-interleaved in frame order.
"""
import asyncio
import contextlib
import gc
import os
import logging
import threading
import time
from concurrent.futures import ThreadPoolExecutor
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch

logger = logging.getLogger(__name__)


_QUIET_GC_LOCK = threading.Lock()
_QUIET_GC_DEPTH = 0   # entries of quiet_gc() that hold OUR freeze (nested or concurrent process_video calls)


@contextlib.contextmanager
def quiet_gc():
    """The frame loop allocates a few thousand small containers per batch (result dicts: plain, cycle-free, freed by reference
    counting), which keeps tripping CPython's cyclic collector -- and every full pass rescans the whole long-lived heap (modules,
    state dicts, staging buffers).  Measured on the bench's sustained leg: 9.6 k -> 12.4 k frames/s with the long-lived heap moved out
    of the collector's sight.  ``gc.freeze()`` (CPython >= 3.7) does exactly that for everything alive at loop entry; young objects
    are still collected.  VTD_QUIET_GC=0 switches it off.

    Process-global state, so: when the host application has frozen objects of its own (the prefork ``gc.freeze()`` pattern) this does
    NOTHING -- the host made its choice, and a freeze of ours on top could never be undone without thawing the host's objects too
    (every call would leave more of the heap permanently uncollectable).  Nested and concurrent entries are counted: the first one
    freezes, only the last one to leave unfreezes (the first call to finish must not thaw the heap under the others)."""
    global _QUIET_GC_DEPTH
    if os.environ.get("VTD_QUIET_GC", "1") == "0" or not hasattr(gc, "freeze"):
        yield
        return
    with _QUIET_GC_LOCK:
        if _QUIET_GC_DEPTH == 0 and gc.get_freeze_count() != 0:
            ours = False               # the host's freeze: leave the collector's state alone
        else:
            ours = True
            if _QUIET_GC_DEPTH == 0:
                gc.collect()
                gc.freeze()
            _QUIET_GC_DEPTH += 1
    try:
        yield
    finally:
        if ours:
            with _QUIET_GC_LOCK:
                _QUIET_GC_DEPTH -= 1
                if _QUIET_GC_DEPTH == 0:
                    gc.unfreeze()


def _is_overridden(obj, name, owner_cls):
    """True when `name` was patched on the instance or the object is not the native class at all."""
    if owner_cls is None or not isinstance(obj, owner_cls):
        return True
    if name in vars(obj):
        return True
    return getattr(type(obj), name, None) is not getattr(owner_cls, name, None)


class VideoTextPipeline:
    def __init__(self,
                 detector_path: Optional[str] = None,
                 recognizer_path: Optional[str] = None,
                 use_transformer_ocr: bool = True,
                 confidence_threshold: float = 0.5,
                 batch_size: int = 16,
                 backbone: Optional[str] = None):
        from .detector import TextDetector
        from .recognizer import TextRecognizer
        from .video import ImageProcessor, VideoProcessor
        self.detector = TextDetector(detector_path, backbone=backbone)
        self.recognizer = TextRecognizer(recognizer_path, use_transformer=use_transformer_ocr)
        self.video_processor = VideoProcessor()
        self.image_processor = ImageProcessor()
        self.confidence_threshold = confidence_threshold
        self.batch_size = batch_size
        self.executor = ThreadPoolExecutor(max_workers=4)
        self._device_index = torch.cuda.current_device() if torch.cuda.is_available() else None

    # ---------------------------------------------------------------------------------- video loop
    async def process_video(self, video_path: str, output_dir: str, progress_callback=None) -> Dict[str, Any]:
        """pipeliine.py:34-91.  When torch.distributed is initialised with W > 1 ranks (one process per GPU) the loop runs in
        its rank-aware mode (BASELINE configs[3]): sampled frame i belongs to rank i mod W, every rank pushes
        only its own frames through its device pipeline, and once per round of W x batch_size frames the ranks exchange
        their finished results in ONE all_gather of a padded block (vtd_amd/shard.py, RCCL on its own stream).  Rank 0
        returns the merged result ordered by frame number -- identical to the single-GPU result; the other ranks return
        rank 0's summary (broadcast at the end) and video_info with an empty 'results' list.  Every rank walks the whole clip
        (numbering and timestamps are whole-clip); a source with ``set_shard`` (vtd_amd.video.VideoProcessor) hands the frames of
        other ranks out as None without reading them, any other source is simply filtered here.  A rank
        that fails tells the others at their next sequence point (shard.ResultGather: error flag in the capacity
        all_reduce), so all ranks return 'failed' together instead of waiting in a collective."""
        gather = None
        quiet = contextlib.ExitStack()
        try:
            from . import shard
            quiet.enter_context(quiet_gc())
            start_time = time.time()
            self._bind_device()
            video_info = self.video_processor.get_video_info(video_path)
            frames = self.video_processor.extract_frames_generator(video_path)
            all_results: List[Dict] = []
            frame_count = 0
            total_frames = video_info.get("frame_count", 0)
            pending_frames, pending_info = [], []
            world, rank = 1, 0
            if shard.is_distributed() and os.environ.get("VTD_SHARD_VIDEO", "1") != "0":
                import torch.distributed as dist
                world, rank = dist.get_world_size(), dist.get_rank()
                gather = shard.ResultGather()
                if hasattr(self.video_processor, "set_shard"):   # frames of other ranks come out as None, unread (numbering stays whole-clip)
                    self.video_processor.set_shard(rank, world)
                    quiet.callback(self.video_processor.set_shard, 0, 1)
            seen = 0          # frames of the current round seen by every rank (W x batch_size closes a round)

            loop = asyncio.get_event_loop()

            async def flush(last=False):
                # Batches of uint8 frames ride the three-deep device pipeline (upload stream -> detector -> post-process /
                # recogniser streams, see _pipeline_push), one device pass per frame size: results come back two or three batches
                # later, in frame order.  Anything else drains the pipeline first and takes the reference-shaped route.
                nonlocal frame_count, seen
                done = []
                if pending_frames:
                    if self._fast_path_ok(pending_frames):
                        done = await loop.run_in_executor(self.executor, self._pipeline_push, list(pending_frames), list(pending_info))
                    else:
                        done = await loop.run_in_executor(self.executor, self._pipeline_drain)
                        done += await self._process_frame_batch(pending_frames, pending_info, output_dir)
                if last:
                    done += await loop.run_in_executor(self.executor, self._pipeline_drain)
                if gather is None:
                    all_results.extend(done)
                    frame_count += len(pending_frames)
                else:  # every rank enters the collective once per round, whatever it carries
                    await loop.run_in_executor(self.executor, gather.submit, done)
                    merged = await loop.run_in_executor(self.executor, gather.retire, 0 if last else 1)
                    if rank == 0:
                        all_results.extend(merged)
                    frame_count += seen
                    seen = 0
                pending_frames.clear()
                pending_info.clear()

            async for frame, frame_number, timestamp in frames:
                if gather is None or (frame_count + seen) % world == rank:
                    pending_frames.append(frame)
                    pending_info.append((frame_number, timestamp))
                seen += 1
                if (len(pending_frames) >= self.batch_size) if gather is None else (seen >= world * self.batch_size):
                    await flush()
                    if progress_callback:
                        progress = frame_count / total_frames if total_frames > 0 else 0
                        await progress_callback(progress, frame_count, total_frames)
            await flush(last=True)
            if gather is not None:
                all_results.sort(key=lambda fr: fr["frame_number"])
            processing_time = time.time() - start_time
            summary = self._generate_summary(all_results, processing_time, frame_count)
            if gather is not None:   # last sequence point: nobody failed -> every rank returns rank 0's summary
                summary = await loop.run_in_executor(self.executor, gather.finish, summary if rank == 0 else None)
            out = {"status": "success", "results": all_results, "summary": summary, "video_info": video_info}
            if gather is not None:
                out["shard"] = {"rank": rank, "world_size": world}
            return out
        except Exception as e:
            logger.error(f"Video processing failed: {e}")
            self._abandon_pipeline()
            if gather is not None and not gather.closed:
                try:   # tell the peers at their next sequence point (they return 'failed' too)
                    gather.abort()
                except Exception as e2:
                    logger.error(f"Could not notify the peer ranks: {e2}")
            return {"status": "failed", "error": str(e), "results": []}
        finally:
            quiet.close()

    def _bind_device(self):
        """Executor threads start on HIP device 0 whatever the constructing thread selected: re-select this pipeline's GPU
        (one process per GPU; LOCAL_RANK picks it) at every thread entry."""
        dev = getattr(self, "_device_index", None)
        if dev is not None and torch.cuda.is_available():
            torch.cuda.set_device(dev)

    # ---- device pipeline of the video loop: batch i uploads and detects while batch i-1 is recognised and batch i-2 is collected
    @staticmethod
    def _shape_groups(frames):
        """[(positions of the frames of one shape)] in order of first appearance: one device pass per group."""
        groups = {}
        for pos, f in enumerate(frames):
            groups.setdefault(tuple(f.shape), []).append(pos)
        return list(groups.values())

    def _stage(self, chunk):
        """A group's frames as one resident batch: host arrays go through a pinned staging buffer and the upload stream (the
        copy overlaps the previous batch's compute); frames that already live in HBM are stacked there."""
        from .engine import PINNED, DeviceFrames
        if torch.is_tensor(chunk[0]):
            return DeviceFrames(torch.stack(list(chunk))), None
        host = PINNED.take((len(chunk),) + tuple(chunk[0].shape), torch.uint8)  # pinned staging: the copy is truly asynchronous
        staged = host.numpy()
        for i, f in enumerate(chunk):
            staged[i] = f
        return DeviceFrames(host, stream=self._upload), host

    def _pipeline_push(self, frames, frame_info) -> List[Dict]:
        """One batch into the device pipeline; returns the result dicts of whole batches that have come out (in order).

        Every job (one shape group of a batch, cut to the engine's max_batch) moves one stage per job pushed behind it:
        staged (host frames only: in the pinned buffer, its copy in flight on the upload stream) -> detector enqueued -> recogniser
        enqueued (needs the boxes on the host: by then the detector has had a whole interval) -> collected.  Host frames enter the
        detector one interval AFTER their copy was issued, so the detector stream never waits for PCIe and the next batch's copy
        runs beside the current batch's detector pass (bench.py --upload: 10.2 k frames/s against 9.1 k when every batch waits for
        its own copy first; VTD_STAGE_AHEAD=0 restores that order).  Frames that already live in HBM skip the staged stage."""
        self._bind_device()
        if getattr(self, "_inflight", None) is None:
            self._inflight = []
        if getattr(self, "_upload", None) is None:
            self._upload = torch.cuda.Stream()
        ahead = os.environ.get("VTD_STAGE_AHEAD", "1") != "0"
        out = []
        cap = getattr(self.detector.model.engine(), "max_batch", len(frames))
        # results of one pushed batch are handed back together, in frame order, once its last group retires (jobs retire in
        # submission order, so batches stay in order too)
        # ("left" is the batch's TOTAL job count from the start: a first job that fails at once and retires while it is the only one
        # in flight must not look like the batch's last -- its frames would be handed back twice, round-3 advisor finding)
        jobs = [positions[start:start + cap] for positions in self._shape_groups(frames) for start in range(0, len(positions), cap)]
        token = {"left": len(jobs), "parts": []}
        for idx in jobs:
            tick = self.__dict__["_tick"] = self.__dict__.get("_tick", 0) + 1
            job = {"info": [frame_info[i] for i in idx], "host": None, "token": token, "pos": idx}
            try:
                batch, job["host"] = self._stage([frames[i] for i in idx])
                if ahead and job["host"] is not None:
                    job["staged"] = batch
                else:
                    self._enqueue_detection(job, batch, tick)
            except Exception as e:
                # the reference's detect() swallows its errors and yields [] for that frame (text_detector.py:139-141): a batch
                # that cannot be enqueued degrades to empty detections for its frames, the video goes on
                logger.error(f"Detection failed: {e}")
                self._count_error("detection")
                job["failed"] = True
            self._route_count("device", len(idx))
            self._inflight.append(job)
            for older in self._inflight[:-1]:            # GPU work first ...
                if "staged" in older:                   # its copy left one interval ago
                    self._enqueue_detection(older, older.pop("staged"), tick)
            for older in self._inflight[:-1]:            # ... then what makes the host wait (the boxes of a batch)
                if "rec" not in older and not older.get("failed") and older.get("det_tick", tick) < tick:
                    self._try_recognition(older)
                    older["rec_tick"] = tick
            # a job retires `lag` pushes after its recogniser was submitted: 1 for the CRNN; the Transformer recogniser asks for more
            # (engine.TrOCREngine.pipeline_lag) so that the pass BEHIND the one being decoded is already queued and its encoder pass
            # runs beside that decode
            lag = self._recognizer_lag()
            while self._inflight and (self._inflight[0].get("failed") or self._inflight[0].get("rec_tick", tick) + lag <= tick):
                out += self._retire(self._inflight.pop(0))
        return out

    def _recognizer_lag(self):
        try:
            if getattr(self.recognizer, "use_transformer", False) and os.environ.get("VTD_TROCR_LAG", "") != "":
                return max(1, int(os.environ["VTD_TROCR_LAG"]))
            if getattr(self.recognizer, "use_transformer", False):
                return max(1, int(getattr(self.recognizer.model.engine(), "pipeline_lag", 1)))
        except Exception:
            pass
        return 1

    def _enqueue_detection(self, job, batch, tick):
        try:
            job.update(self.submit_detection(batch))
            job["det_tick"] = tick
        except Exception as e:
            logger.error(f"Detection failed: {e}")
            self._count_error("detection")
            job["failed"] = True

    def _count_error(self, kind):
        """The reference's error convention swallows failures (log + empty result); a caller that must not mistake them for results
        (bench.py) reads them here: {'detection': n, 'recognition': n, 'collection': n}."""
        counts = self.__dict__.setdefault("error_counts", {})
        counts[kind] = counts.get(kind, 0) + 1

    def _route_count(self, route, n):
        counts = self.__dict__.setdefault("route_counts", {"device": 0, "reference": 0})
        counts[route] += n

    def _try_recognition(self, job):
        if "rec" in job or job.get("failed"):
            return
        try:
            self.submit_recognition(job)
        except Exception as e:
            logger.error(f"Detection failed: {e}")
            self._count_error("detection")
            job["failed"] = True

    def _retire(self, job) -> List[Dict]:
        from .engine import PINNED
        if "staged" in job:   # drain: a batch that never got its interval
            self._enqueue_detection(job, job.pop("staged"), self.__dict__.get("_tick", 0))
        self._try_recognition(job)
        res = None
        if not job.get("failed"):
            try:
                res = self.collect(job, job["info"])
            except Exception as e:
                logger.error(f"Batch collection failed: {e}")
                self._count_error("collection")
        if res is None:
            res = [{"frame_number": num, "timestamp": ts, "detections": []} for num, ts in job["info"]]
        if job.get("host") is not None:
            PINNED.release(job["host"])
        token = job["token"]
        token["parts"] += zip(job["pos"], res)
        token["left"] -= 1
        if token["left"]:
            return []
        parts, token["parts"] = token["parts"], []
        return [r for _, r in sorted(parts, key=lambda t: t[0])]

    def _pipeline_drain(self) -> List[Dict]:
        self._bind_device()
        out = []
        while getattr(self, "_inflight", None):
            out += self._retire(self._inflight.pop(0))
        return out

    def _abandon_pipeline(self):
        try:
            torch.cuda.synchronize()
        except Exception:
            pass
        self._inflight = []
        try:   # tickets the Transformer recogniser still holds (each keeps a whole frame batch alive) must not ride into the next video
            if getattr(self.recognizer, "use_transformer", False):
                self.recognizer.model.engine().discard_queue()
        except Exception as e:
            logger.error(f"Could not discard the recogniser's queue: {e}")

    # ---------------------------------------------------------------------------------- batches
    def _fast_path_ok(self, frames) -> bool:
        """The device path takes any mix of frame sizes; only the test seams (detect / recognize patched or replaced) and frames
        that are not HxWx3 uint8 arrays (or resident uint8 tensors) go the reference-shaped way."""
        try:
            from .detector import TextDetector
            from .recognizer import TextRecognizer
        except Exception:
            return False
        if _is_overridden(self.detector, "detect", TextDetector) or _is_overridden(self.recognizer, "recognize", TextRecognizer):
            return False
        if not frames:
            return False

        def ok(f):
            shape = getattr(f, "shape", None)
            if shape is None or len(shape) != 3 or shape[2] != 3 or shape[0] <= 0 or shape[1] <= 0:
                return False
            return f.dtype == torch.uint8 and f.is_cuda if torch.is_tensor(f) else getattr(f, "dtype", None) == np.uint8
        return all(ok(f) for f in frames)

    def process_device_batch(self, batch, frame_info=None) -> List[Dict]:
        """The batched device pass on frames that are already resident in HBM (a ``DeviceFrames``): fused
        preprocess -> DBNet -> post-process for every frame, then all crops of the batch through crop/resize ->
        CRNN -> CTC decode.  Two small device->host reads (boxes, decoded text); everything else stays on the GPU."""
        n = batch.n
        frame_info = frame_info or [(i, 0.0) for i in range(n)]
        detections = self.detector.detect_batch(batch, self.confidence_threshold)
        boxes, owners = [], []
        for i, dets in enumerate(detections):
            for j, det in enumerate(dets):
                x1, y1, x2, y2 = det["bbox"]
                if x2 > x1 and y2 > y1:  # `cropped_text.size == 0` -> skipped (pipeliine.py:122-123)
                    boxes.append((i, x1, y1, x2, y2))
                    owners.append((i, j))
        texts = self.recognizer.recognize_boxes(batch, boxes) if boxes else []
        per_frame = [[] for _ in range(n)]
        for (i, j), rec in zip(owners, texts):
            det = detections[i][j]
            per_frame[i].append({"bbox": det["bbox"], "text": rec["text"], "detection_confidence": det["confidence"],
                                 "recognition_confidence": rec["confidence"], "polygon": det.get("polygon", [])})
        return [{"frame_number": frame_info[i][0], "timestamp": frame_info[i][1], "detections": per_frame[i]} for i in range(n)]

    # ---- the same pass split into enqueue / collect halves so consecutive batches overlap: while the host reads the
    # boxes of batch i and launches its recogniser, the GPU already runs the detector of batch i+1
    def submit_detection(self, batch):
        return {"batch": batch, "det": self.detector.submit_batch(batch, self.confidence_threshold)}

    def submit_recognition(self, job):
        detections = self.detector.finish_batch(job["det"])
        boxes, owners = [], []
        for i, dets in enumerate(detections):
            for j, det in enumerate(dets):
                x1, y1, x2, y2 = det["bbox"]
                if x2 > x1 and y2 > y1:
                    boxes.append((i, x1, y1, x2, y2))
                    owners.append((i, j))
        # The recogniser of batch i runs on its own HIP stream: it is independent of the detector of batch i+1 that is
        # already queued on the caller's stream, and its narrow kernels (LSTM recurrence: 17 workgroups, CTC decode, crop)
        # then fill CUs the detector's convolutions leave idle between launches instead of serialising behind them.
        stream = self._recognizer_stream()
        try:
            if stream is None:
                rec = self.recognizer.submit_boxes(job["batch"], boxes)
            else:
                stream.wait_event(job["det"]["event"])  # frames uploaded + detector done with them, in stream order
                with torch.cuda.stream(stream):
                    rec = self.recognizer.submit_boxes(job["batch"], boxes)
        except Exception as e:
            logger.error(f"CRNN batch recognition failed: {e}")
            self._count_error("recognition")
            rec, job["rec_failed"] = None, True
        job.update(detections=detections, owners=owners, rec=rec)
        return job

    def _recognizer_stream(self):
        if os.environ.get("VTD_REC_STREAM", "1") == "0":
            return None
        if getattr(self, "_rec_stream", None) is None:
            self._rec_stream = torch.cuda.Stream()
        return self._rec_stream

    def collect(self, job, frame_info=None) -> List[Dict]:
        n = job["batch"].n
        frame_info = frame_info or [(i, 0.0) for i in range(n)]
        try:
            if job.get("rec_failed"):
                raise RuntimeError("recogniser submission failed")
            texts = self.recognizer.finish_boxes(job["rec"])
        except Exception as e:   # text_recognizer.py:138-140: log, empty text, zero confidence, detections kept
            logger.error(f"CRNN batch recognition failed: {e}")
            self._count_error("recognition")
            texts = [{"text": "", "confidence": 0.0}] * len(job["owners"])
        per_frame = [[] for _ in range(n)]
        for (i, j), rec in zip(job["owners"], texts):
            det = job["detections"][i][j]
            per_frame[i].append({"bbox": det["bbox"], "text": rec["text"], "detection_confidence": det["confidence"],
                                 "recognition_confidence": rec["confidence"], "polygon": det.get("polygon", [])})
        return [{"frame_number": frame_info[i][0], "timestamp": frame_info[i][1], "detections": per_frame[i]} for i in range(n)]

    def _batched_device_pass(self, frames, frame_info) -> List[Dict]:
        from .engine import DeviceFrames
        self._bind_device()
        parts = []
        cap = getattr(self.detector.model.engine(), "max_batch", len(frames))
        for positions in self._shape_groups(frames):   # one device pass per frame size, results back in frame order
            for start in range(0, len(positions), cap):
                idx = positions[start:start + cap]
                chunk = [frames[i] for i in idx]
                batch = DeviceFrames(torch.stack(chunk) if torch.is_tensor(chunk[0]) else chunk)
                parts += zip(idx, self.process_device_batch(batch, [frame_info[i] for i in idx]))
                self._route_count("device", len(idx))
        return [r for _, r in sorted(parts, key=lambda t: t[0])]

    async def _process_frame_batch(self, frames: List[np.ndarray], frame_info: List[Tuple], output_dir: str) -> List[Dict]:
        loop = asyncio.get_event_loop()
        if self._fast_path_ok(frames):
            return await loop.run_in_executor(self.executor, self._batched_device_pass, list(frames), list(frame_info))
        # reference-shaped route (pipeliine.py:96-141): N=1 detect per frame on the pool, one recognize per crop
        self._route_count("reference", len(frames))
        tasks = [loop.run_in_executor(self.executor, self.detector.detect, frame, self.confidence_threshold) for frame in frames]
        batch_detections = await asyncio.gather(*tasks)
        results = []
        for (frame_number, timestamp), frame, detections in zip(frame_info, frames, batch_detections):
            regions = []
            for detection in detections or []:
                x1, y1, x2, y2 = detection["bbox"]
                crop = frame[y1:y2, x1:x2]
                if crop.size == 0:
                    continue
                text = self.recognizer.recognize(crop)
                regions.append({"bbox": detection["bbox"], "text": text["text"], "detection_confidence": detection["confidence"],
                                "recognition_confidence": text["confidence"], "polygon": detection.get("polygon", [])})
            results.append({"frame_number": frame_number, "timestamp": timestamp, "detections": regions})
        return results

    def process_single_frame(self, frame: np.ndarray) -> Dict[str, Any]:
        try:
            detections = self.detector.detect(frame, self.confidence_threshold)
            if not detections:
                return {"detections": []}
            regions = []
            for detection in detections:
                x1, y1, x2, y2 = detection["bbox"]
                crop = frame[y1:y2, x1:x2]
                if crop.size == 0:
                    continue
                text = self.recognizer.recognize(crop)
                regions.append({"bbox": detection["bbox"], "text": text["text"], "detection_confidence": detection["confidence"],
                                "recognition_confidence": text["confidence"]})
            return {"detections": regions}
        except Exception as e:
            logger.error(f"Single frame processing failed: {e}")
            return {"detections": [], "error": str(e)}

    # ---------------------------------------------------------------------------------- summary
    def _generate_summary(self, results: List[Dict], processing_time: float, frame_count: int) -> Dict[str, Any]:
        dets = [d for frame in results for d in frame["detections"]]
        if dets:
            avg_det = float(np.mean([d["detection_confidence"] for d in dets]))
            avg_rec = float(np.mean([d["recognition_confidence"] for d in dets]))
        else:
            avg_det = avg_rec = 0.0
        texts = {d["text"].strip() for d in dets if d["text"].strip()}
        return {
            "total_frames": frame_count,
            "frames_with_text": sum(1 for frame in results if frame["detections"]),
            "total_detections": len(dets),
            "unique_texts": len(texts),
            "detected_texts": list(texts),
            "avg_detection_confidence": avg_det,
            "avg_recognition_confidence": avg_rec,
            "processing_time_seconds": processing_time,
            "fps_processed": frame_count / processing_time if processing_time > 0 else 0,
        }
