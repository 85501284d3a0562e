"""Device engines: thin owners of the native handles (include/vtd.h) plus the torch tensors that
serve as their caller-owned device buffers.  torch is plumbing here (allocation, streams); every
FLOP is issued by libvtd_hip.so.
"""
import ctypes as C
import logging
import os
import sys
import threading

import numpy as np
import torch

from . import _native

logger = logging.getLogger(__name__)

DEFAULT_MAX_BATCH = int(os.environ.get("VTD_MAX_BATCH", "32"))

# Kernel-selection table shipped with the package (include/vtd.h: vtd_*_set_tuning): chosen on an MI355X by tools/tune_table.py,
# loaded into every engine so that all processes / ranks run the same kernels.  VTD_TUNING_FILE points at another table,
# VTD_TUNING=0 ignores tables (every shape is then decided by the in-process timing contest).
TUNING_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuning", "gfx950.txt")


def shipped_tuning_text():
    if os.environ.get("VTD_TUNING", "1") == "0":
        return ""
    path = os.environ.get("VTD_TUNING_FILE", TUNING_FILE)
    try:
        with open(path) as f:
            return f.read()
    except OSError:
        return ""


class _Tunable:
    """get / set of a handle's kernel-selection table (kind = 'detector' | 'recognizer')."""

    _kind = None

    def set_tuning(self, text):
        """Replaces table entries; under the engine lock, so a forward / generate in another thread never sees its resolved launch
        slots cleared half way."""
        if text:
            fn = getattr(self.lib, f"vtd_{self._kind}_set_tuning")
            with self.lock:
                _native.check(fn(self.handle, text.encode()), f"vtd_{self._kind}_set_tuning")

    def tuning_text(self):
        fn = getattr(self.lib, f"vtd_{self._kind}_get_tuning")
        need = fn(self.handle, None, 0)
        buf = C.create_string_buffer(int(need))
        fn(self.handle, buf, need)
        return buf.value.decode()

    def tuning(self):
        return {k: int(v) for k, v in (line.rsplit(" ", 1) for line in self.tuning_text().splitlines() if line)}

    @property
    def tuning_measured(self):
        """True when some kernel choice of this engine came from its own timing contest (not from a table)."""
        return bool(getattr(self.lib, f"vtd_{self._kind}_tuning_measured")(self.handle))


def _stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class PinnedPool:
    """Recycles pinned host staging buffers (hipHostMalloc costs far more than the copies they serve).  A buffer is
    handed out again only after `release`, which callers do once the event guarding its last copy has completed."""

    def __init__(self):
        self._free = {}
        self._lock = threading.Lock()

    def take(self, shape, dtype=torch.int32):
        key = (tuple(shape), dtype)
        with self._lock:
            lst = self._free.get(key)
            if lst:
                return lst.pop()
        return torch.empty(shape, dtype=dtype).pin_memory()

    def release(self, t):
        with self._lock:
            self._free.setdefault((tuple(t.shape), t.dtype), []).append(t)


PINNED = PinnedPool()


def copy_to_pinned(dst, src):
    """Device tensor -> pinned host tensor on the current stream, by a copy kernel (include/vtd.h vtd_copy_to_pinned_host).  The
    asynchronous memcpy of the detection records -- a few KB on the post-process side stream -- was seen to block the host for a whole
    detector pass, once per drained pipeline; a launch never waits for the GPU.  Falls back to the plain
    asynchronous copy if the buffer is not mapped."""
    nbytes = src.numel() * src.element_size()
    if nbytes >= 16 and dst.is_pinned() and src.is_contiguous() and dst.is_contiguous() and dst.numel() * dst.element_size() == nbytes and \
            _native.require().vtd_copy_to_pinned_host(src.data_ptr(), dst.data_ptr(), nbytes, torch.cuda.current_stream().cuda_stream) == 0:
        return
    dst.copy_(src, non_blocking=True)


class DeviceFrames:
    """A batch of equally sized uint8 BGR HWC frames resident in HBM ([n,H,W,3] cuda tensor)."""

    def __init__(self, frames, stream=None):
        """`stream`: upload on that HIP stream (give it pinned host memory and the copy overlaps the previous batch's
        compute); consumers order themselves behind `self.ready`.  Default: the caller's current stream."""
        if isinstance(frames, np.ndarray):
            frames = torch.from_numpy(np.ascontiguousarray(frames))
        elif isinstance(frames, (list, tuple)):
            frames = torch.from_numpy(np.ascontiguousarray(np.stack(frames)))
        if frames.dim() == 3:
            frames = frames.unsqueeze(0)
        if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[-1] != 3:
            raise ValueError("frames must be uint8 [n,H,W,3] (BGR)")
        self.ready = None
        if stream is None or frames.is_cuda:
            self.tensor = frames.to("cuda", non_blocking=True).contiguous()
        else:
            with torch.cuda.stream(stream):
                self.tensor = frames.contiguous().to("cuda", non_blocking=True)
                self.ready = torch.cuda.Event()
                self.ready.record()

    def wait_ready(self):
        """Make the current stream wait for an upload that was issued on another stream (no-op otherwise)."""
        if self.ready is not None:
            torch.cuda.current_stream().wait_event(self.ready)

    @property
    def n(self):
        return self.tensor.shape[0]

    @property
    def height(self):
        return self.tensor.shape[1]

    @property
    def width(self):
        return self.tensor.shape[2]


class DetectorEngine(_Tunable):
    """DBNet on the GPU: reference checkpoint tensors in, [n,1,640,640] probability maps out."""

    _kind = "detector"

    def __init__(self, backbone, state_dict, max_batch=None, options=None):
        self.lib = _native.require()
        self.max_batch = max_batch or DEFAULT_MAX_BATCH
        self.backbone = backbone
        self.lock = threading.Lock()  # a handle is single-stream; detect() may be entered from 4 threads
        h = C.c_void_p()
        _native.check(self.lib.vtd_detector_create(backbone.encode(), self.max_batch, C.byref(h)), "vtd_detector_create")
        self.handle = h
        try:
            for name, value in (options or {}).items():
                _native.check(self.lib.vtd_detector_set_option(h, name.encode(), int(value)), f"vtd_detector_set_option({name})")
            for key, value in state_dict.items():
                if key.endswith("num_batches_tracked"):
                    continue
                arr = np.ascontiguousarray(value.detach().cpu().float().numpy())
                _native.check(self.lib.vtd_detector_set_tensor(h, key.encode(), arr.ctypes.data, arr.size),
                              f"vtd_detector_set_tensor({key})")
            _native.check(self.lib.vtd_detector_finalize(h, _stream_ptr()), "vtd_detector_finalize")
            self.set_tuning(shipped_tuning_text())
        except Exception:
            self.close()
            raise

    def close(self):
        if getattr(self, "handle", None):
            self.lib.vtd_detector_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def macs_per_frame(self):
        return int(self.lib.vtd_detector_macs_per_frame(self.handle))

    def _set_input(self, x):
        if isinstance(x, DeviceFrames):
            n = x.n
            x.wait_ready()
            _native.check(self.lib.vtd_detector_preprocess(self.handle, C.c_void_p(x.tensor.data_ptr()), n, x.height, x.width,
                                                           _stream_ptr()), "vtd_detector_preprocess")
            return n, x
        if not torch.is_tensor(x) or x.dim() != 4 or tuple(x.shape[1:]) != (3, 640, 640):
            raise ValueError("DBNet input must be [n,3,640,640] float or a DeviceFrames batch")
        xd = x.to("cuda", torch.float32).contiguous()
        _native.check(self.lib.vtd_detector_set_input_nchw(self.handle, C.c_void_p(xd.data_ptr()), xd.shape[0], _stream_ptr()),
                      "vtd_detector_set_input_nchw")
        return xd.shape[0], xd

    def forward(self, x, want_threshold=False):
        with self.lock:
            n, keep = self._set_input(x)
            prob = torch.empty((n, 1, 640, 640), dtype=torch.float32, device="cuda")
            thr = torch.empty_like(prob) if want_threshold else None
            _native.check(self.lib.vtd_detector_forward(self.handle, n, C.c_void_p(prob.data_ptr()),
                                                        C.c_void_p(thr.data_ptr()) if thr is not None else None, _stream_ptr()),
                          "vtd_detector_forward")
            del keep
            return {"probability": prob, "threshold": thr}

    def read_tap(self, name, n):
        shapes = {"input": (3, 640, 640), "stem": (64, 320, 320), "pool": (64, 160, 160), "p2": (256, 160, 160), "head1": (64, 160, 160),
                  "head2": (64, 320, 320)}
        if name not in shapes:
            wide = self.backbone == "resnet50"
            idx = int(name[1]) - 2
            ch = (64, 128, 256, 512)[idx] * (4 if wide else 1)
            shapes[name] = (ch, 160 >> idx, 160 >> idx)
        c, h, w = shapes[name]
        out = np.empty((n, c, h, w), np.float32)
        with self.lock:
            _native.check(self.lib.vtd_detector_read_tap(self.handle, name.encode(), n, out.ctypes.data, out.size, _stream_ptr()),
                          f"vtd_detector_read_tap({name})")
        return out


class PostProcessor:
    """``TextDetector._post_process`` on the GPU for a batch of probability maps (vtd_postproc_*)."""

    def __init__(self, max_batch, map_h=640, map_w=640, max_out=1024):
        self.lib = _native.require()
        self.max_batch, self.h, self.w, self.max_out = max_batch, map_h, map_w, max_out
        self.lock = threading.Lock()
        h = C.c_void_p()
        _native.check(self.lib.vtd_postproc_create(max_batch, map_h, map_w, max_out, C.byref(h)), "vtd_postproc_create")
        self.handle = h
        self.records = torch.empty((max_batch, max_out, 16), dtype=torch.int32, device="cuda")
        self.counts = torch.empty((max_batch,), dtype=torch.int32, device="cuda")

    def close(self):
        if getattr(self, "handle", None):
            self.lib.vtd_postproc_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run_device(self, prob, orig_w, orig_h, threshold, records=None, counts=None):
        """Enqueue only; returns (records[n,max_out,16] int32 view, counts[n]) device tensors.  Pass `records` /
        `counts` to keep several batches in flight (the default buffers are overwritten by the next call)."""
        n = prob.shape[0]
        records = self.records if records is None else records
        counts = self.counts if counts is None else counts
        if prob.dtype != torch.float32 or not prob.is_cuda or tuple(prob.shape[-2:]) != (self.h, self.w):
            raise ValueError("probability maps must be float32 cuda tensors of the workspace's map size")
        prob = prob.reshape(n, self.h, self.w).contiguous()
        ow = np.ascontiguousarray(orig_w, dtype=np.int32)
        oh = np.ascontiguousarray(orig_h, dtype=np.int32)
        _native.check(self.lib.vtd_postproc_run(self.handle, C.c_void_p(prob.data_ptr()), n, ow.ctypes.data, oh.ctypes.data,
                                                float(threshold), C.c_void_p(records.data_ptr()),
                                                C.c_void_p(counts.data_ptr()), _stream_ptr()), "vtd_postproc_run")
        self._keep = (prob, ow, oh)
        return records[:n], counts[:n]

    def run(self, prob, orig_w, orig_h, threshold, debug=False):
        with self.lock:
            rec, cnt = self.run_device(prob, orig_w, orig_h, threshold)
            cnt = cnt.cpu().numpy()
            if int(cnt.max(initial=0)) > self.max_out:
                import logging
                logging.getLogger(__name__).warning(f"frame with {int(cnt.max())} components exceeds max_detections={self.max_out}; "
                                                    "extra detections dropped")
            kmax = int(min(cnt.max(initial=0), self.max_out))
            rec = rec[:, :kmax].cpu().numpy() if kmax else np.zeros((len(cnt), 0, 16), np.int32)
        return [records_to_dicts(rec[i, :min(int(cnt[i]), self.max_out)], debug) for i in range(len(cnt))]


def records_to_dicts(rec, debug=False):
    """vtd_detection records -> the reference's detection dicts (text_detector.py:172-176): plain Python ints/floats."""
    if len(rec) == 0:
        return []
    bbox = rec[:, 0:4].tolist()
    poly = rec[:, 4:12].reshape(-1, 4, 2).tolist()
    conf = np.ascontiguousarray(rec[:, 12]).view(np.float32).tolist()
    out = [{"bbox": b, "confidence": c, "polygon": p} for b, c, p in zip(bbox, conf, poly)]
    if debug:
        area = np.ascontiguousarray(rec[:, 13]).view(np.float32).tolist()
        first = rec[:, 14:16].tolist()
        for d, a, f in zip(out, area, first):
            d["_area"] = a
            d["_first"] = tuple(f)
    return out


def detector_profile(engine):
    """[(description, total_ms, calls, total_macs)] per launch slot of a DetectorEngine (after set_profiling(1))."""
    lib, out = engine.lib, []
    name = C.create_string_buffer(160)
    ms, calls, macs = C.c_double(), C.c_int64(), C.c_double()
    for i in range(lib.vtd_detector_num_ops(engine.handle)):
        _native.check(lib.vtd_detector_get_profile(engine.handle, i, name, 160, C.byref(ms), C.byref(calls), C.byref(macs),
                                                   _stream_ptr()), "vtd_detector_get_profile")
        out.append((name.value.decode(), ms.value, calls.value, macs.value))
    return out


class RecognizerEngine(_Tunable):
    """CRNN on the GPU: crops (or reference-format [n,3,32,128] tensors) in, [n,31,V] logits / decoded text out."""

    _kind = "recognizer"

    T = 31

    def __init__(self, vocab_size, state_dict, max_crops=None, options=None):
        self.lib = _native.require()
        self.vocab_size = vocab_size
        self.max_crops = max_crops or int(os.environ.get("VTD_MAX_CROPS", "512"))
        self.lock = threading.Lock()
        h = C.c_void_p()
        _native.check(self.lib.vtd_recognizer_create(vocab_size, self.max_crops, C.byref(h)), "vtd_recognizer_create")
        self.handle = h
        try:
            # VTD_RECOGNIZER_OPTIONS="fuse_pools=0": build options for A/B measurements (include/vtd.h: vtd_recognizer_set_option)
            env_opts = {k: int(v) for k, v in (kv.split("=") for kv in os.environ.get("VTD_RECOGNIZER_OPTIONS", "").split(",") if kv)}
            for name, value in {**env_opts, **(options or {})}.items():
                _native.check(self.lib.vtd_recognizer_set_option(h, name.encode(), int(value)), f"vtd_recognizer_set_option({name})")
            for key, value in state_dict.items():
                if key.endswith("num_batches_tracked"):
                    continue
                arr = np.ascontiguousarray(value.detach().cpu().float().numpy())
                _native.check(self.lib.vtd_recognizer_set_tensor(h, key.encode(), arr.ctypes.data, arr.size),
                              f"vtd_recognizer_set_tensor({key})")
            _native.check(self.lib.vtd_recognizer_finalize(h, _stream_ptr()), "vtd_recognizer_finalize")
            self.set_tuning(shipped_tuning_text())
        except Exception:
            self.close()
            raise

    def close(self):
        if getattr(self, "handle", None):
            self.lib.vtd_recognizer_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def macs_per_crop(self):
        return int(self.lib.vtd_recognizer_macs_per_crop(self.handle))

    def _forward_current(self, n):
        logits = torch.empty((n, self.T, self.vocab_size), dtype=torch.float32, device="cuda")
        _native.check(self.lib.vtd_recognizer_forward(self.handle, n, C.c_void_p(logits.data_ptr()), _stream_ptr()),
                      "vtd_recognizer_forward")
        return logits

    def forward_logits(self, x):
        """CRNN.forward on a reference-format tensor [n,3,32,128] (any device) -> [n,31,V] f32 cuda."""
        if not torch.is_tensor(x) or x.dim() != 4 or tuple(x.shape[1:]) != (3, 32, 128):
            raise ValueError("CRNN input must be [n,3,32,128] float")
        outs = []
        with self.lock:
            for i in range(0, x.shape[0], self.max_crops):
                xd = x[i:i + self.max_crops].to("cuda", torch.float32).contiguous()
                _native.check(self.lib.vtd_recognizer_set_input_nchw(self.handle, C.c_void_p(xd.data_ptr()), xd.shape[0], _stream_ptr()),
                              "vtd_recognizer_set_input_nchw")
                outs.append(self._forward_current(xd.shape[0]))
        return outs[0] if len(outs) == 1 else torch.cat(outs)

    def load_crops(self, frames, boxes):
        """K6 + conv1 for boxes [(frame, x1, y1, x2, y2), ...] on a DeviceFrames batch; call with the lock held."""
        b = boxes if torch.is_tensor(boxes) else torch.as_tensor(np.asarray(boxes, dtype=np.int32).reshape(-1, 5))
        b = b.to("cuda", torch.int32).contiguous()
        n = b.shape[0]
        frames.wait_ready()
        _native.check(self.lib.vtd_recognizer_crop_resize(self.handle, C.c_void_p(frames.tensor.data_ptr()), frames.n, frames.height,
                                                          frames.width, C.c_void_p(b.data_ptr()), n, _stream_ptr()),
                      "vtd_recognizer_crop_resize")
        self._keep = (b, frames)
        return n

    def forward_crops(self, frames, boxes):
        with self.lock:
            n = self.load_crops(frames, boxes)
            return self._forward_current(n)

    def submit_decode(self, frames, boxes_np, id2char_dev, blank_id=0):
        """Enqueue crop/resize -> CRNN -> softmax+CTC decode for boxes_np ([k,5] int32) and an asynchronous copy of the
        decoded records to pinned host memory.  Returns a ticket for ``finish_decode`` (nothing synchronises here)."""
        k = int(boxes_np.shape[0])
        host_boxes = PINNED.take((k, 5))
        host_boxes.numpy()[...] = boxes_np
        with self.lock:
            dev_boxes = host_boxes.to("cuda", non_blocking=True)
            n = self.load_crops(frames, dev_boxes)
            logits = self._forward_current(n)
            out = torch.empty((k, 2 + self.T), dtype=torch.int32, device="cuda")
            _native.check(self.lib.vtd_ctc_greedy_decode(C.c_void_p(logits.data_ptr()), k, self.T, self.vocab_size,
                                                         C.c_void_p(id2char_dev.data_ptr()), blank_id, 1, C.c_void_p(out.data_ptr()),
                                                         _stream_ptr()), "vtd_ctc_greedy_decode")
            host = PINNED.take((k, 2 + self.T))
            copy_to_pinned(host, out)
            ev = torch.cuda.Event()
            ev.record()
        return {"host": host, "event": ev, "keep": (host_boxes, dev_boxes, logits, out, frames)}

    @staticmethod
    def finish_decode(ticket):
        ticket["event"].synchronize()
        out = decode_records_to_text(ticket["host"].numpy())
        PINNED.release(ticket["host"])
        PINNED.release(ticket["keep"][0])
        return out

    def read_tap(self, name, n):
        shape = (n, 32, 128, 3) if name == "resized" else (n, 512, 1, 31)
        out = np.empty(shape, np.float32)
        with self.lock:
            _native.check(self.lib.vtd_recognizer_read_tap(self.handle, name.encode(), n, out.ctypes.data, out.size, _stream_ptr()),
                          f"vtd_recognizer_read_tap({name})")
        return out


def ctc_greedy_decode(logits, id2char, blank_id=0, apply_softmax=True):
    """[softmax +] the reference's greedy decode on the GPU.  logits: [n,T,V] float tensor (any device), logits when
    apply_softmax else probabilities; id2char: list of code points / -1.  Returns [(text, confidence)]."""
    lib = _native.require()
    lg = logits.detach().to("cuda", torch.float32).contiguous()
    n, T, V = lg.shape
    table = torch.tensor(list(id2char)[:V] + [-1] * max(0, V - len(id2char)), dtype=torch.int32, device="cuda")
    out = torch.empty((n, 2 + T), dtype=torch.int32, device="cuda")
    _native.check(lib.vtd_ctc_greedy_decode(C.c_void_p(lg.data_ptr()), n, T, V, C.c_void_p(table.data_ptr()), blank_id,
                                            1 if apply_softmax else 0, C.c_void_p(out.data_ptr()), _stream_ptr()),
                  "vtd_ctc_greedy_decode")
    return decode_records_to_text(out.cpu().numpy())


def decode_records_to_text(host):
    """[n][2+T] int32 decode records -> [(text, confidence)]."""
    conf = np.ascontiguousarray(host[:, 1]).view(np.float32).tolist()
    lens = host[:, 0].tolist()
    rows = host[:, 2:].tolist()
    return [("".join(map(chr, rows[i][:lens[i]])), conf[i]) for i in range(len(lens))]


_MASKED_STREAMS = {}   # (device, total CUs, decode CUs) -> (encoder stream, decode stream): process-wide, like torch's own stream pool
_MASKED_LOCK = threading.Lock()


def _masked_stream_pair(lib, device, total, dec_cus):
    """Two HIP streams with disjoint CU masks (include/vtd.h: vtd_stream_create_masked), wrapped for torch.

    What a mask bit means on this part was measured, not assumed (tools/cumask_map.py on an MI355X, ROCm 7.2): bit i enables CU
    i // 8 of XCC (die) i % 8, and a die whose bits are ALL zero is not masked at all -- every one of its CUs stays enabled.  So a
    partition must leave every die some CUs on both sides: the decode gets CUs 0 .. dec_cus / 8 - 1 of every die, the encoder pass the
    other ones (an evenly 'strided' mask, the first thing tried, emptied whole dies on one side and partitioned nothing).  Equal
    shares per die also keep the persistent dense GEMM's one-workgroup-per-CU grid balanced (dense_gemm.hip sizes its grid by the
    stream's mask).  The pair is created once per process and shared by every engine on the device: torch's allocators remember the
    streams a block was used on (a pinned buffer's free records an event on each of them), so a stream handed to torch must outlive
    every tensor that met it -- these are never destroyed (tools/cumask_probe.py: a process exits cleanly with them alive)."""
    dies = 8
    if total % dies:
        raise _native.NativeError(f"{total} CUs do not split over {dies} dies")
    per_die = total // dies
    dc = min(max(1, dec_cus // dies), per_die - 1)
    key = (device, total, dc)
    with _MASKED_LOCK:
        if key not in _MASKED_STREAMS:
            words = (total + 31) // 32
            streams = []
            for want_dec in (False, True):
                m = (C.c_uint32 * words)()
                for die in range(dies):
                    for cu in range(per_die):
                        if (cu < dc) == want_dec:
                            bit = cu * dies + die
                            m[bit // 32] |= 1 << (bit % 32)
                h = C.c_void_p()
                _native.check(lib.vtd_stream_create_masked(m, words, C.byref(h)), "vtd_stream_create_masked")
                streams.append(torch.cuda.ExternalStream(h.value))
            _MASKED_STREAMS[key] = tuple(streams)
        return _MASKED_STREAMS[key] + (dc * dies,)


class TrOCREngine(_Tunable):
    """The Transformer recogniser on the GPU (include/vtd.h: vtd_trocr_*): crops of resident frames, or the
    ``pixel_values`` tensor the reference hands to ``generate``, in; greedy token ids out."""

    _kind = "trocr"

    def __init__(self, spec, state_dict, max_crops=None, slots=None, xattn=None):
        from .trocr_spec import hf4_key
        self.lib = _native.require()
        self.spec = spec
        self.max_crops = max_crops or int(os.environ.get("VTD_TROCR_MAX_CROPS", "256"))   # rows per encoder pass / decode (~20 MB of HBM each:
                                                                                            # workspaces + caches + a 0.89 MB slot; 47 MB with the 28 MB slot of xattn=False)
        self.lock = threading.Lock()
        cfg = _native.TrocrConfig(spec.image_size, spec.patch_size, spec.enc_hidden, spec.enc_layers, spec.enc_heads, spec.enc_ffn,
                                  int(spec.enc_qkv_bias), spec.enc_ln_eps, spec.dec_hidden, spec.dec_layers, spec.dec_heads, spec.dec_ffn,
                                  spec.vocab_size, spec.max_positions, spec.dec_ln_eps, spec.decoder_start_token_id, spec.eos_token_id,
                                  spec.pad_token_id, spec.max_length)
        h = C.c_void_p()
        _native.check(self.lib.vtd_trocr_create(C.byref(cfg), self.max_crops, C.byref(h)), "vtd_trocr_create")
        self.handle = h
        try:
            # one encoder-output slot unless the overlapped order (two passes in flight) or a caller asks for the second one
            want_slots = slots or (2 if os.environ.get("VTD_TROCR_OVERLAP", "0") == "1" else 1)
            _native.check(self.lib.vtd_trocr_set_option(h, b"slots", int(want_slots)), "vtd_trocr_set_option(slots)")
            # decoder cross-attention on the raw encoder states (csrc/trocr_xattn.hip; include/vtd.h option "xattn") unless asked for the
            # reference's per-layer key / value projections: xattn=False, or VTD_TROCR_XATTN=0
            want_xattn = (os.environ.get("VTD_TROCR_XATTN", "1") != "0") if xattn is None else bool(xattn)
            _native.check(self.lib.vtd_trocr_set_option(h, b"xattn", int(want_xattn)), "vtd_trocr_set_option(xattn)")
            for key, value in state_dict.items():
                arr = np.ascontiguousarray(value.detach().cpu().float().numpy())
                _native.check(self.lib.vtd_trocr_set_tensor(h, hf4_key(key).encode(), arr.ctypes.data, arr.size), f"vtd_trocr_set_tensor({key})")
            _native.check(self.lib.vtd_trocr_finalize(h, _stream_ptr()), "vtd_trocr_finalize")
            self.set_tuning(shipped_tuning_text())
        except Exception:
            self.close()
            raise
        self.tokens = int(self.lib.vtd_trocr_encoder_tokens(h))
        self.logits_stride = int(self.lib.vtd_trocr_logits_stride(h))
        self.slots = int(self.lib.vtd_trocr_num_slots(h))
        form = C.c_int()
        _native.check(self.lib.vtd_trocr_get_option(h, b"xattn", C.byref(form)), "vtd_trocr_get_option(xattn)")
        self.xattn = bool(form.value)   # the form that runs (a geometry the kernel does not cover keeps the key / value form)
        self._next_slot = 0
        self._queue = []          # tickets whose crops are not staged yet (submit_crops / finish)
        self._qlock = threading.Lock()
        self._passes = []         # encoded passes that wait for their decode, oldest first
        self._setup_overlap()

    def close(self):
        if getattr(self, "_worker", None) is not None and self._worker.is_alive() and not sys.is_finalizing():
            self._generation = getattr(self, "_generation", 0) + 1
            self._jobs.put(None)
            self._worker.join(timeout=60)
        if getattr(self, "handle", None):
            if not sys.is_finalizing():   # work may still be queued on the shared encoder / decode streams
                try:
                    torch.cuda.synchronize()
                except Exception:
                    pass
            self.lib.vtd_trocr_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def macs_per_crop(self):
        return int(self.lib.vtd_trocr_macs_per_crop(self.handle))

    @property
    def last_steps(self):
        """Decoder steps the last generate call enqueued (it stops once every row has emitted </s>)."""
        return int(self.lib.vtd_trocr_last_steps(self.handle))

    def encode_pixels(self, pixel_values, slot=0):
        """[n,3,S,S] float (any device): ViT encoder + cross-attention keys / values into `slot`.  Call with the lock held."""
        s = self.spec.image_size
        if not torch.is_tensor(pixel_values) or pixel_values.dim() != 4 or tuple(pixel_values.shape[1:]) != (3, s, s):
            raise ValueError(f"pixel_values must be [n,3,{s},{s}] float")
        x = pixel_values.to("cuda", torch.float32).contiguous()
        _native.check(self.lib.vtd_trocr_encode_pixels_slot(self.handle, slot, C.c_void_p(x.data_ptr()), x.shape[0], _stream_ptr()),
                      "vtd_trocr_encode_pixels")
        self._keep = x
        return x.shape[0]

    def encode_crops(self, frames, boxes, slot=0):
        """boxes [(frame, x1, y1, x2, y2), ...] of a DeviceFrames batch into `slot`.  Call with the lock held."""
        b = np.ascontiguousarray(np.asarray(boxes, dtype=np.int32).reshape(-1, 5))
        frames.wait_ready()
        _native.check(self.lib.vtd_trocr_encode_crops_slot(self.handle, slot, C.c_void_p(frames.tensor.data_ptr()), frames.n, frames.height,
                                                           frames.width, b.ctypes.data, b.shape[0], _stream_ptr()), "vtd_trocr_encode_crops")
        self._keep = (b, frames)
        return b.shape[0]

    def _enqueue_generate(self, n, slot, max_length=None, forced=None, want_logits=False):
        """Greedy decode of the n crops encoded into `slot`, on the CURRENT stream.  Returns device tensors (ids, logits | None); the
        call itself returns once all but the last two decoder steps have run (the handle paces itself two steps behind the GPU)."""
        max_length = max_length or self.spec.max_length
        ids = torch.empty((n, max_length), dtype=torch.int32, device="cuda")
        logits = torch.zeros((n, max_length - 1, self.logits_stride), dtype=torch.float32, device="cuda") if want_logits else None
        fdev, flen = None, 0
        if forced is not None:
            fdev = torch.as_tensor(np.asarray(forced), dtype=torch.int32).to("cuda").contiguous()
            flen = fdev.shape[1]
        _native.check(self.lib.vtd_trocr_generate_slot(self.handle, slot, n, max_length, C.c_void_p(fdev.data_ptr()) if fdev is not None else None,
                                                       flen, C.c_void_p(ids.data_ptr()), C.c_void_p(logits.data_ptr()) if logits is not None else None,
                                                       _stream_ptr()), "vtd_trocr_generate")
        self._keep_gen = (fdev,)
        return ids, logits

    def generate_current(self, n, max_length=None, forced=None, want_logits=False, slot=0):
        """Greedy decode of the n crops encoded last into `slot`.  Returns (ids [n,max_length] int32 cpu tensor, logits or None)."""
        ids, logits = self._enqueue_generate(n, slot, max_length, forced, want_logits)
        out = ids.cpu()
        return out, (logits[..., :self.spec.vocab_size].cpu() if logits is not None else None)

    def generate_pixels(self, pixel_values, **kw):
        outs = []
        with self.lock:
            for i in range(0, pixel_values.shape[0], self.max_crops):
                n = self.encode_pixels(pixel_values[i:i + self.max_crops])
                outs.append(self.generate_current(n, **kw))
        ids = torch.cat([o[0] for o in outs])
        logits = torch.cat([o[1] for o in outs]) if outs[0][1] is not None else None
        return ids, logits

    def generate_crops(self, frames, boxes, **kw):
        outs = []
        with self.lock:
            for i in range(0, len(boxes), self.max_crops):
                n = self.encode_crops(frames, boxes[i:i + self.max_crops])
                outs.append(self.generate_current(n, **kw)[0])
        return torch.cat(outs)

    # ---- pipelined use: recogniser batches are decoupled from detector batches ---------------------------------------------------
    # A decode step costs about the same whether 30 or 300 rows are live (~136 dependent launches), so crops are worth collecting:
    # submit_crops only QUEUES a ticket; the GPU work starts when some ticket's result is asked for (finish) or the queue would
    # overflow the workspace.  Queued tickets are then cut into PASSES of `pass_tickets` tickets (VTD_TROCR_PASS_TICKETS, default 2);
    # every pass is ONE encoder pass into an encoder-output slot and ONE decode.
    #
    # A caller keeps `pipeline_lag` = pass_tickets - 1 tickets in flight behind the one it asks for (VideoTextPipeline._pipeline_push and
    # bench.py do), so that a whole pass is queued when its first ticket is finished; with fewer in flight the passes are simply smaller.
    #
    # Overlap (VTD_TROCR_OVERLAP=1, off by default): the handle has two slots, and the encoder pass of pass k+1 can run BESIDE the
    # decode of pass k -- each on a stream of its own whose kernels are confined to a disjoint part of the chip (include/vtd.h:
    # vtd_stream_create_masked; VTD_TROCR_DEC_CUS of the 256 CUs for the decode, the rest for the encoder pass; pipeline_lag is then
    # 2 pass_tickets - 1).  Measured twice, lost twice: on plain streams (round 3) the decoder's ~6.7 k small
    # dependent launches queued for CU slots behind the encoder's wide ones (10 -> 50 us each) and the encoder pass doubled; with the
    # chip really partitioned (round 4: 32 / 64 / 96 / 128 CUs for the decode) every decode launch finds its CUs free, but the decode's
    # kernels are sized to finish in ONE round on 256 CUs -- on a quarter of the chip each takes several -- and the encoder pass loses
    # the same share: 135 - 238 frames/s against 279 back to back on the ResNet-18 line.  The mode stays as a tested option.
    def _setup_overlap(self):
        self.pass_tickets = max(1, int(os.environ.get("VTD_TROCR_PASS_TICKETS", "2")))
        self.overlap = False
        self._enc_stream = self._dec_stream = None
        if os.environ.get("VTD_TROCR_OVERLAP", "0") == "1" and self.slots >= 2:
            try:
                total = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
                dec_cus = min(max(8, int(os.environ.get("VTD_TROCR_DEC_CUS", "80"))), total - 8)
                self._enc_stream, self._dec_stream, dec_cus = _masked_stream_pair(self.lib, torch.cuda.current_device(), total, dec_cus)
                self.overlap = True
                self.dec_cus, self.enc_cus = dec_cus, total - dec_cus
            except Exception as e:   # no CU masks on this stack: the back-to-back order
                logger.warning(f"TrOCREngine: CU-masked streams unavailable ({e}); encoder pass and decode run back to back")
                self._enc_stream = self._dec_stream = None
        # Asynchronous passes (VTD_TROCR_ASYNC=1): a full pass (pass_tickets tickets) goes to a worker thread, which stages, encodes and
        # decodes it while the submitting thread keeps feeding the detector -- a pass blocks its host thread for its whole (host-paced)
        # decode, during which the caller's thread otherwise submits nothing: the detector's launches of the next pass's batches then fall
        # into the decode's launch-bound tail instead of running while the recogniser's streams sit idle.  The caller keeps two passes of
        # tickets in flight (pipeline_lag = 2 pass_tickets - 1).
        self.async_passes = os.environ.get("VTD_TROCR_ASYNC", "0") == "1" and not self.overlap
        self._jobs = self._worker = None
        self._generation = 0
        self.pipeline_lag = 2 * self.pass_tickets - 1 if (self.overlap or self.async_passes) else max(1, self.pass_tickets - 1)

    def submit_crops(self, frames, boxes):
        """Queue the crops `boxes` ([(frame, x1, y1, x2, y2), ...]) of a resident frame batch; returns a ticket for ``finish``.  Nothing
        is enqueued on the GPU unless the queue has to be flushed to make room (VTD_TROCR_MERGE=0: every ticket runs on its own,
        at once)."""
        b = np.ascontiguousarray(np.asarray(boxes, dtype=np.int32).reshape(-1, 5))
        ticket = {"boxes": b, "frames": frames, "stream": torch.cuda.current_stream(), "parts": None}
        if self.async_passes and os.environ.get("VTD_TROCR_MERGE", "1") != "0":
            # (the engine lock is the worker's for the whole of a pass: the queue has a lock of its own, so submitting never waits for one)
            job = None
            with self._qlock:
                self._queue.append(ticket)
                if len(self._queue) >= self.pass_tickets:
                    group, self._queue = self._queue[:self.pass_tickets], self._queue[self.pass_tickets:]
                    for t in group:
                        t["ready"] = threading.Event()
                    job = (self._generation, group, torch.cuda.current_device())
            if job is not None:
                self._start_worker()
                self._jobs.put(job)
            return ticket
        with self.lock:
            room = self.max_crops * (self.slots if self.overlap else 1)
            if sum(len(t["boxes"]) for t in self._queue) + len(b) > room:
                self._flush()
            with self._qlock:
                self._queue.append(ticket)
            if os.environ.get("VTD_TROCR_MERGE", "1") == "0":
                self._flush()
        return ticket

    def _start_worker(self):
        if self._worker is None or not self._worker.is_alive():
            import queue as _queue
            self._jobs = _queue.Queue()
            self._worker = threading.Thread(target=self._worker_loop, name="vtd-trocr-pass", daemon=True)
            self._worker.start()

    def _worker_loop(self):
        jobs = self._jobs
        while True:
            job = jobs.get()
            if job is None:
                return
            generation, group, device = job
            try:
                torch.cuda.set_device(device)
                with self.lock:
                    if generation == self._generation and getattr(self, "handle", None):   # (discard_queue / close bump the generation)
                        self._run_groups([group], group)
            except Exception as e:   # finish() reports it per ticket (parts stay None), as for a synchronous pass that raised
                logger.error(f"TrOCREngine: recogniser pass failed: {e}")
            finally:
                for t in group:
                    t["ready"].set()

    def _flush(self):
        """Everything queued is cut into passes and their encoder passes are enqueued (lock held).  Back-to-back mode decodes every pass
        at once; overlap mode leaves the decodes to ``finish`` (and runs the oldest ones only when the slots run out).  Tickets larger
        than the workspace are cut into passes of max_crops rows; every ticket ends up with a list of (pass, spans)."""
        with self._qlock:
            queue, self._queue = self._queue, []
        if not queue:
            return
        merge = os.environ.get("VTD_TROCR_MERGE", "1") != "0"
        groups, k = [], (self.pass_tickets if merge else 1)
        for i in range(0, len(queue), k):
            groups.append(queue[i:i + k])
        self._run_groups(groups, queue)

    def _run_groups(self, groups, queue):
        """Stage + encode (+ decode, back to back) the ticket groups, one or more passes each (lock held)."""
        if getattr(self, "_passes", None) is None:
            self._passes = []
        parts = {id(t): [] for t in queue}
        for group in groups:
            rows = [(t, i) for t in group for i in range(len(t["boxes"]))]
            caller = group[-1]["stream"]
            for start in range(0, len(rows), self.max_crops):
                chunk = rows[start:start + self.max_crops]
                while len(self._passes) >= self.slots:       # both slots hold passes that wait for their decode: run the oldest
                    self._decode_pass(self._passes[0])
                slot = self._next_slot
                self._next_slot = (slot + 1) % self.slots
                enc = self._enc_stream if self.overlap else caller
                if self.overlap:
                    enc.wait_stream(caller)                  # the frames the crops come out of (detector / upload order)
                spans = {}
                with torch.cuda.stream(enc):
                    # runs of consecutive rows of one ticket are staged with one processor launch each
                    off, kk = 0, 0
                    while kk < len(chunk):
                        t, i0 = chunk[kk]
                        k2 = kk
                        while k2 < len(chunk) and chunk[k2][0] is t:
                            k2 += 1
                        n = k2 - kk
                        fr = t["frames"]
                        fr.wait_ready()
                        bb = np.ascontiguousarray(t["boxes"][i0:i0 + n])
                        _native.check(self.lib.vtd_trocr_stage_crops_slot(self.handle, slot, C.c_void_p(fr.tensor.data_ptr()), fr.n, fr.height, fr.width,
                                                                          bb.ctypes.data, n, off, _stream_ptr()), "vtd_trocr_stage_crops")
                        spans.setdefault(id(t), []).append((off, i0, n))
                        off += n
                        kk = k2
                    _native.check(self.lib.vtd_trocr_encode_staged_slot(self.handle, slot, off, _stream_ptr()), "vtd_trocr_encode_staged")
                pas = {"slot": slot, "rows": off, "caller": caller, "host": None, "event": None, "ids": None, "users": 0, "decoded": False}
                for t in group:
                    if id(t) in spans:
                        parts[id(t)].append((pas, spans[id(t)]))
                        pas["users"] += 1
                self._passes.append(pas)
                if not self.overlap:
                    self._decode_pass(pas)
        for t in queue:             # only now are the tickets marked as run: an exception above leaves them unmarked
            t["parts"] = parts[id(t)]   # (a ticket keeps its frames until it is finished: the processor launches read them asynchronously)

    def _decode_pass(self, pas):
        """Greedy decode of one encoded pass (lock held): host-paced, returns when all but the last two steps have run."""
        if pas["decoded"]:
            return
        stream = self._dec_stream if self.overlap else self.decode_stream(pas["caller"])
        host = None
        try:
            with torch.cuda.stream(stream):
                ids, _ = self._enqueue_generate(pas["rows"], pas["slot"])
                host = PINNED.take(tuple(ids.shape))
                copy_to_pinned(host, ids)
                ev = torch.cuda.Event()
                ev.record()
            pas.update(host=host, event=ev, ids=ids)
        except Exception:
            if host is not None:
                PINNED.release(host)
            pas["failed"] = True
            raise
        finally:
            pas["decoded"] = True
            if pas in self._passes:
                self._passes.remove(pas)

    def decode_stream(self, stream=None):
        """Back-to-back mode: the stream a pass's decode runs on is the one its encoder pass was enqueued on.  (VTD_TROCR_DEC_STREAM=1:
        a plain high-priority stream of its own -- the round-3 experiment that lost, kept for A/B runs.)"""
        if os.environ.get("VTD_TROCR_DEC_STREAM", "0") != "1":
            return stream if stream is not None else torch.cuda.current_stream()
        if self._dec_stream is None:
            self._dec_stream = torch.cuda.Stream(priority=-1)
        return self._dec_stream

    def finish(self, ticket):
        """ids [n, max_length] int32 (cpu) of a ticket, rows in the order of its boxes.  Flushes the queue when the ticket is still in
        it (every ticket queued by then gets its encoder pass enqueued: the passes behind this ticket's run beside its decode).
        Host-blocking: returns when this ticket's decode has finished."""
        ready = ticket.get("ready")
        if ready is not None:
            ready.wait()               # its pass runs (or ran) on the worker thread: encoded and decoded when the event is set
        else:
            with self.lock:
                if ticket["parts"] is None:
                    self._flush()
                if ticket["parts"] is not None:
                    for pas, _ in ticket["parts"]:
                        while not pas["decoded"]:          # passes decode in the order they were encoded
                            self._decode_pass(self._passes[0])
        if ticket["parts"] is None:    # its pass raised half way (the exception went to whoever triggered the flush)
            raise _native.NativeError("the recogniser pass this ticket was queued for failed")
        n = len(ticket["boxes"])
        out = torch.empty((n, self.spec.max_length), dtype=torch.int32)
        try:
            for pas, spans in ticket["parts"]:
                if pas.get("failed") or pas["event"] is None:
                    raise _native.NativeError("the recogniser pass this ticket was queued for failed")
                pas["event"].synchronize()
                for off, i0, cnt in spans:
                    out[i0:i0 + cnt] = pas["host"][off:off + cnt]
        finally:
            for pas, _ in ticket["parts"]:
                pas["users"] -= 1
                if pas["users"] == 0 and pas["host"] is not None:
                    PINNED.release(pas["host"])
                    pas["host"] = pas["ids"] = None
            ticket["parts"] = []
            ticket["frames"] = None
        return out

    def discard_queue(self):
        """Drop every ticket that has not been finished (an abandoned video): queued tickets release their frame batches, encoded
        passes are forgotten (their slots are reused in order; the handle's events keep the GPU side consistent)."""
        with self.lock:
            self._generation = getattr(self, "_generation", 0) + 1   # passes still waiting for the worker are dropped when it gets to them
            with self._qlock:
                for t in self._queue:
                    t["frames"] = None
                    t["parts"] = None
                self._queue = []
            for pas in getattr(self, "_passes", None) or []:
                pas["decoded"] = pas["failed"] = True
            self._passes = []

    def set_profiling(self, mode):
        _native.check(self.lib.vtd_trocr_set_profiling(self.handle, int(mode)), "vtd_trocr_set_profiling")

    def profile(self):
        """(total ms, launches, summed row counts) of the bracketed cross-attention launches since the last call."""
        ms, calls, rows = C.c_double(), C.c_int64(), C.c_int64()
        stream = self._dec_stream if self._dec_stream is not None else torch.cuda.current_stream()   # get_profile waits for it
        torch.cuda.synchronize()
        _native.check(self.lib.vtd_trocr_get_profile(self.handle, C.byref(ms), C.byref(calls), C.byref(rows), C.c_void_p(stream.cuda_stream)),
                      "vtd_trocr_get_profile")
        return ms.value, calls.value, rows.value

    def gemm_profile(self):
        """(total ms, launches, executed FLOPs) of the encoder pass's dense-GEMM launches bracketed since the last call (profiling mode bit 1)."""
        ms, calls, flops = C.c_double(), C.c_int64(), C.c_double()
        torch.cuda.synchronize()
        _native.check(self.lib.vtd_trocr_get_gemm_profile(self.handle, C.byref(ms), C.byref(calls), C.byref(flops), _stream_ptr()),
                      "vtd_trocr_get_gemm_profile")
        return ms.value, calls.value, flops.value

    def read_tap(self, name, n):
        s = self.spec
        shape = (n, 3, s.image_size, s.image_size) if name == "pixel_values" else (n, self.tokens, s.enc_hidden)
        out = np.empty(shape, np.float32)
        with self.lock:
            _native.check(self.lib.vtd_trocr_read_tap(self.handle, name.encode(), n, out.ctypes.data, out.size, _stream_ptr()),
                          f"vtd_trocr_read_tap({name})")
        return out


def trim_generated(ids, spec):
    """Rows of generate()'s id matrix -> lists cut after <eos> (the start token, which is also id 2, stays)."""
    out = []
    for row in ids.tolist():
        seq = [row[0]]
        for tok in row[1:]:
            seq.append(tok)
            if tok == spec.eos_token_id:
                break
        while len(seq) > 1 and seq[-1] == spec.pad_token_id:
            seq.pop()
        out.append(seq)
    return out
