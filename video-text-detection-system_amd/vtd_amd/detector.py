"""``TextDetector`` with the reference's call surface (app/ml/models/text_detector.py:88-178), running on
the HIP engine.  Same constructor, attributes (``model``, ``device``, ``transform``), ``detect`` /
``_post_process`` / ``load_model`` signatures, result schema and swallow-and-log error convention; plus
batched extensions (``detect_batch``) that keep frames, probability maps and detections in HBM.

Mock seams kept alive (tests/test_models.py:30,148,174 of the reference): ``detect`` routes through
``self.model(...)`` (so ``patch.object(detector.model, 'forward')`` takes effect) and accepts whatever
map size that returns.
"""
import logging
import os
import threading

import numpy as np
import torch

from . import _native
from .engine import PINNED, DeviceFrames, PostProcessor, copy_to_pinned
from .nets import DBNet

logger = logging.getLogger(__name__)


class FusedTransform:
    """Stand-in for the reference's ``transforms.Compose([ToPILImage, Resize((640,640)), ToTensor, Normalize])``
    (text_detector.py:99-104): callable on an RGB HWC uint8 array, returns the normalised ``[3,640,640]``
    float tensor -- computed by the fused preprocess kernel (it takes BGR, so the channel swap is undone here)."""

    def __init__(self, detector):
        self._detector = detector

    def __call__(self, image_rgb):
        bgr = np.ascontiguousarray(np.asarray(image_rgb)[..., ::-1])
        eng = self._detector.model.engine()
        with eng.lock:
            eng._set_input(DeviceFrames(bgr))
        return torch.from_numpy(eng.read_tap("input", 1)[0])


class TextDetector:
    def __init__(self, model_path: str = None, device: str = None, backbone: str = None, max_batch: int = None,
                 max_detections: int = 1024):
        _native.require()  # no GPU / no library -> raise here, never a silent CPU path
        self.device = device or ("cuda" if torch.cuda.is_available() else "cpu")
        backbone = backbone or os.environ.get("VTD_BACKBONE", "resnet50")
        state = torch.random.get_rng_state()
        try:
            torch.manual_seed(0)  # reference default fetches ImageNet weights (A3); here: deterministic init
            self.model = DBNet(backbone)
        finally:
            torch.random.set_rng_state(state)
        if max_batch:
            self.model._max_batch = max_batch
        if model_path:
            self.load_model(model_path)
        self.model.eval()
        self.transform = FusedTransform(self)
        self.max_detections = max_detections
        self._pp = {}
        self._pp_lock = threading.Lock()

    def load_model(self, model_path: str):
        try:
            checkpoint = torch.load(model_path, map_location="cpu", weights_only=True)
            self.model.load_state_dict(checkpoint["model_state_dict"])
            logger.info(f"Model loaded from {model_path}")
        except Exception as e:
            logger.error(f"Failed to load model: {e}")
            raise

    # ---- post-process -------------------------------------------------------------------------
    def _postprocessor(self, h, w, batch):
        key = (h, w)
        with self._pp_lock:
            pp = self._pp.get(key)
            if pp is None or pp.max_batch < batch:
                # a smaller workspace that another thread already fetched stays alive through that thread's reference and is
                # destroyed with it (PostProcessor.__del__): never close a handle someone may be about to use
                pp = PostProcessor(max(batch, 1), h, w, self.max_detections)
                self._pp[key] = pp
            return pp

    def _post_process_maps(self, prob, widths, heights, threshold):
        """prob: [n,h,w] float32 cuda tensor -> list (per frame) of detection dicts."""
        n, h, w = prob.shape
        return self._postprocessor(h, w, n).run(prob, widths, heights, threshold)

    def _post_process(self, prob_map, orig_width: int, orig_height: int, threshold: float):
        """text_detector.py:143-178 for one map; accepts any 2-D float array (tests feed 160x160 float64)."""
        arr = prob_map.detach() if torch.is_tensor(prob_map) else torch.from_numpy(np.ascontiguousarray(prob_map))
        if arr.dim() != 2:
            raise ValueError("prob_map must be 2-D")
        prob = arr.to("cuda", torch.float32).unsqueeze(0)
        return self._post_process_maps(prob, [int(orig_width)], [int(orig_height)], float(threshold))[0]

    # ---- detection ----------------------------------------------------------------------------
    def detect(self, image: np.ndarray, confidence_threshold: float = 0.5):
        try:
            original_height, original_width = image.shape[:2]
            if image.ndim != 3 or image.shape[2] != 3:
                raise ValueError("expected an HxWx3 BGR frame")  # the reference's transform raises here too
            output = self.model(DeviceFrames(image))
            prob = output["probability"]
            prob = prob if torch.is_tensor(prob) else torch.as_tensor(np.asarray(prob))
            return self._post_process(prob[0, 0], original_width, original_height, confidence_threshold)
        except Exception as e:
            logger.error(f"Detection failed: {e}")
            return []

    # ---- asynchronous batched path (software pipelining across batches) ------------------------------------------
    def submit_batch(self, batch: DeviceFrames, confidence_threshold: float = 0.5):
        """Enqueue preprocess -> DBNet -> post-process for a resident batch plus an asynchronous copy of the detection
        records to pinned host memory; nothing synchronises.  ``finish_batch(ticket)`` returns the per-frame dicts."""
        n = batch.n
        prob = self.model(batch)["probability"]
        pp = self._postprocessor(prob.shape[-2], prob.shape[-1], n)
        # The post-process (a dependent chain of narrow kernels) runs on its own HIP stream behind an event: the caller's
        # stream is free for the next batch's preprocess + DBNet at once and the two overlap on the GPU.
        side = self._post_stream()
        if side is not None:
            done = torch.cuda.Event()
            done.record()
            side.wait_event(done)
        with pp.lock, torch.cuda.stream(side if side is not None else torch.cuda.current_stream()):
            records = torch.empty((n, pp.max_out, 16), dtype=torch.int32, device="cuda")
            counts = torch.empty((n,), dtype=torch.int32, device="cuda")
            pp.run_device(prob.reshape(n, prob.shape[-2], prob.shape[-1]), [batch.width] * n, [batch.height] * n,
                          confidence_threshold, records, counts)
            host_rec = PINNED.take((n, pp.max_out, 16))
            host_cnt = PINNED.take((n,))
            copy_to_pinned(host_rec, records)
            copy_to_pinned(host_cnt, counts)
            ev = torch.cuda.Event()
            ev.record()
        return {"rec": host_rec, "cnt": host_cnt, "event": ev, "max_out": pp.max_out, "keep": (prob, records, counts, batch)}

    def _post_stream(self):
        if os.environ.get("VTD_POST_STREAM", "1") == "0":
            return None
        if getattr(self, "_side_stream", None) is None:
            self._side_stream = torch.cuda.Stream()
        return self._side_stream

    @staticmethod
    def finish_batch(ticket):
        from .engine import records_to_dicts
        ticket["event"].synchronize()
        rec, cnt = ticket["rec"].numpy(), ticket["cnt"].numpy()
        if len(cnt) and int(cnt.max()) > ticket["max_out"]:   # the reference has no cap: say so instead of dropping silently
            logger.warning(f"{int((cnt > ticket['max_out']).sum())} frame(s) exceed max_detections={ticket['max_out']} "
                           f"(up to {int(cnt.max())} components); extra detections dropped -- raise TextDetector.max_detections")
        out = [records_to_dicts(rec[i, :min(int(cnt[i]), ticket["max_out"])]) for i in range(len(cnt))]
        PINNED.release(ticket["rec"])
        PINNED.release(ticket["cnt"])
        return out

    def detect_batch(self, frames, confidence_threshold: float = 0.5):
        """Batched fast path: ``frames`` is a list/array of equally sized BGR frames or a ``DeviceFrames``.
        One fused preprocess + DBNet + post-process launch sequence; never raises (returns [] per frame)."""
        try:
            batch = frames if isinstance(frames, DeviceFrames) else DeviceFrames(frames)
            n = batch.n
            prob = self.model(batch)["probability"]
            return self._post_process_maps(prob.reshape(n, prob.shape[-2], prob.shape[-1]), [batch.width] * n,
                                           [batch.height] * n, confidence_threshold)
        except Exception as e:
            logger.error(f"Batched detection failed: {e}")
            n = len(frames) if hasattr(frames, "__len__") else getattr(frames, "n", 0)
            return [[] for _ in range(n)]
