"""Parameter containers for DBNet and CRNN with the reference's state-dict key names.

The reference checkpoints are ``{'model_state_dict': sd}`` with keys laid out by
``nn.Sequential(*resnet.children()[:-2])`` / ``FeaturePyramidNetwork`` / ``DBHead``
(app/ml/models/text_detector.py:12-86) and ``CRNN`` (app/ml/models/text_recognizer.py:12-37);
SURVEY.md Appendix C lists them.  These modules reproduce exactly those keys and
shapes so ``load_state_dict(strict=True)`` accepts a reference checkpoint, but they
hold *parameters only*: ``forward`` hands the work to the HIP engine
(``vtd_amd.engine``), never to torch ops.

Repairs relative to the as-shipped reference: the
``'resnet18'`` channel plan exists (A2) and nothing is fetched from the network (A3).
"""
from collections import OrderedDict

import os
import threading

import torch
import torch.nn as nn


def _conv_bn(cin, cout, k, stride, pad):
    return nn.Conv2d(cin, cout, k, stride, pad, bias=False), nn.BatchNorm2d(cout)


class _Residual(nn.Module):
    """Shared shell of the two ResNet v1.5 block kinds (parameters only)."""

    expansion = 1

    def forward(self, x):  # pragma: no cover - compute lives in the HIP engine
        raise RuntimeError("ResNet blocks are parameter containers; run DBNet.forward")


class BasicBlock(_Residual):
    expansion = 1

    def __init__(self, cin, width, stride):
        super().__init__()
        self.conv1, self.bn1 = _conv_bn(cin, width, 3, stride, 1)
        self.conv2, self.bn2 = _conv_bn(width, width, 3, 1, 1)
        self.stride = stride
        if stride != 1 or cin != width:
            self.downsample = nn.Sequential(*_conv_bn(cin, width, 1, stride, 0))


class Bottleneck(_Residual):
    expansion = 4

    def __init__(self, cin, width, stride):
        super().__init__()
        self.conv1, self.bn1 = _conv_bn(cin, width, 1, 1, 0)
        self.conv2, self.bn2 = _conv_bn(width, width, 3, stride, 1)  # v1.5: stride on the 3x3
        self.conv3, self.bn3 = _conv_bn(width, width * 4, 1, 1, 0)
        self.stride = stride
        if stride != 1 or cin != width * 4:
            self.downsample = nn.Sequential(*_conv_bn(cin, width * 4, 1, stride, 0))


_PLANS = {
    # name: (block class, blocks per stage, C5 channels)
    "resnet18": (BasicBlock, (2, 2, 2, 2), 512),
    "resnet50": (Bottleneck, (3, 4, 6, 3), 2048),
}


def make_trunk(name):
    """ResNet children()[:-2] as an 8-slot Sequential: 0 conv7x7/s2, 1 BN, 2 ReLU,
    3 maxpool, 4..7 the four stages (text_detector.py:17-19)."""
    block, counts, _ = _PLANS[name]
    slots = [nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True),
             nn.MaxPool2d(3, 2, 1)]
    cin = 64
    for stage, (n, width) in enumerate(zip(counts, (64, 128, 256, 512))):
        blocks = []
        for b in range(n):
            stride = 2 if (b == 0 and stage > 0) else 1
            blocks.append(block(cin, width, stride))
            cin = width * block.expansion
        slots.append(nn.Sequential(*blocks))
    return nn.Sequential(*slots)


def trunk_out_channels(name):
    return _PLANS[name][2]


class FeaturePyramidNetwork(nn.Module):
    """Same parameters as text_detector.py:31-41; the wiring (B.3 of SURVEY.md) is in the engine."""

    def __init__(self, in_channels):
        super().__init__()
        self.inner_blocks = nn.ModuleList(nn.Conv2d(in_channels >> i, 256, 1) for i in range(4))
        self.layer_blocks = nn.ModuleList(nn.Conv2d(256, 256, 3, padding=1) for _ in range(4))


def _db_branch(c):
    q = c // 4
    return nn.Sequential(
        nn.Conv2d(c, q, 3, padding=1), nn.BatchNorm2d(q), nn.ReLU(inplace=True),
        nn.ConvTranspose2d(q, q, 2, stride=2), nn.BatchNorm2d(q), nn.ReLU(inplace=True),
        nn.ConvTranspose2d(q, 1, 2, stride=2), nn.Sigmoid())


class DBHead(nn.Module):
    def __init__(self, in_channels):
        super().__init__()
        self.probability_head = _db_branch(in_channels)
        self.threshold_head = _db_branch(in_channels)


class _EngineOwner:
    """Engine bookkeeping shared by DBNet and CRNN.  The lock and the native handle are process-local: they are left out of
    pickles and deep copies (``copy.deepcopy(model)``, ``torch.save(model)``, multiprocessing spawn) and rebuilt on demand, so a
    copy packs its own engine from its own parameters at first use."""

    def _init_engine_state(self):
        self._engine = None
        self._engine_version = -1
        self._version = 0
        self._engine_lock = threading.Lock()

    def __getstate__(self):
        state = self.__dict__.copy()
        state["_engine"] = None
        state["_engine_version"] = -1
        state.pop("_engine_lock", None)
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        self._engine = None
        self._engine_version = -1
        self._engine_lock = threading.Lock()

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__getstate__().items():
            new.__dict__[k] = copy.deepcopy(v, memo)
        new._engine_lock = threading.Lock()
        return new

    # any in-place parameter update through the public API invalidates the packed copy
    def load_state_dict(self, state_dict, strict=True, **kw):
        try:
            return super().load_state_dict(state_dict, strict=strict, **kw)
        finally:   # torch copies the matching tensors before it raises on a mismatch: the packed copy is stale either way
            self._version += 1

    def mark_dirty(self):
        self._version += 1


class DBNet(_EngineOwner, nn.Module):
    """DBNet detector network (text_detector.py:12-29), compute on the HIP engine.

    ``forward(x)`` accepts what the reference's ``TextDetector`` feeds it -- a
    normalised float NCHW tensor ``[B,3,640,640]`` -- or a ``DeviceFrames`` batch
    produced by the fused preprocess kernel, and returns
    ``{'probability': [B,1,640,640] f32 cuda tensor, 'threshold': same or None}``.
    """

    def __init__(self, backbone="resnet50", pretrained=False, compute_threshold=False):
        super().__init__()
        if backbone not in _PLANS:
            raise ValueError(f"unknown backbone {backbone!r}; expected one of {sorted(_PLANS)}")
        self.backbone_name = backbone
        self.backbone = make_trunk(backbone)
        self.fpn = FeaturePyramidNetwork(trunk_out_channels(backbone))
        self.head = DBHead(256)
        # the reference computes the threshold map and never reads it at inference
        # (text_detector.py:128); off by default, same kernels when switched on
        self.compute_threshold = compute_threshold
        self._init_engine_state()

    def engine(self):
        """The native engine for the current parameters, built once under a lock (detect() is entered from four pool threads,
        pipeliine.py:32,96-101: without it a first mixed-size batch would build four engines).  A replaced engine is only
        dereferenced, never closed here: a thread still inside it keeps it alive and its handle is destroyed with the last
        reference."""
        from . import engine as _e
        with self._engine_lock:
            if self._engine is None or self._engine_version != self._version:
                version = self._version
                # VTD_DETECTOR_OPTIONS="fuse_fpn_head=0,fuse_stem_pool=0": build options for A/B measurements (include/vtd.h)
                opts = {k: int(v) for k, v in (kv.split("=") for kv in os.environ.get("VTD_DETECTOR_OPTIONS", "").split(",") if kv)}
                self._engine = _e.DetectorEngine(self.backbone_name, self.state_dict(), getattr(self, "_max_batch", None),
                                                 options=opts or None)
                self._engine_version = version
            return self._engine

    def forward(self, x):
        return self.engine().forward(x, want_threshold=self.compute_threshold)


class CRNN(_EngineOwner, nn.Module):
    """CRNN recogniser parameters (text_recognizer.py:12-37): 7 conv(+BN+ReLU) with the four
    pools, 2-layer bidirectional LSTM(512->256), Linear(512->vocab).  ``forward`` takes
    ``[B,3,32,128]`` float (BGR/255, text_recognizer.py:118-119) and returns ``[B,31,V]``
    logits as an f32 cuda tensor."""

    def __init__(self, vocab_size, hidden_size=256, num_layers=2):
        super().__init__()
        if hidden_size != 256 or num_layers != 2:
            raise ValueError("the HIP recogniser is specialised for hidden_size=256, num_layers=2")
        plan = [(3, 64, 3, 1, "p22"), (64, 128, 3, 1, "p22"), (128, 256, 3, 1, None),
                (256, 256, 3, 1, "p21"), (256, 512, 3, 1, None), (512, 512, 3, 1, "p21"),
                (512, 512, 2, 0, None)]
        mods = []
        for cin, cout, k, pad, pool in plan:
            mods += [nn.Conv2d(cin, cout, k, 1, pad), nn.BatchNorm2d(cout), nn.ReLU(True)]
            if pool == "p22":
                mods.append(nn.MaxPool2d(2, 2))
            elif pool == "p21":
                mods.append(nn.MaxPool2d((2, 1), (2, 1)))
        self.cnn = nn.Sequential(*mods)
        self.rnn = nn.LSTM(512, hidden_size, num_layers, batch_first=True, bidirectional=True)
        self.classifier = nn.Linear(hidden_size * 2, vocab_size)
        self.vocab_size = vocab_size
        self._init_engine_state()

    def engine(self):
        from . import engine as _e
        with self._engine_lock:   # see DBNet.engine
            if self._engine is None or self._engine_version != self._version:
                version = self._version
                self._engine = _e.RecognizerEngine(self.vocab_size, self.state_dict(), getattr(self, "_max_crops", None))
                self._engine_version = version
            return self._engine

    def forward(self, x):
        return self.engine().forward_logits(x)


def seeded_state_dict(module_factory, seed):
    """Deterministic default-init weights (torch's own init under a fixed seed) plus
    non-trivial BatchNorm statistics, so BN folding is actually exercised."""
    gen_state = torch.random.get_rng_state()
    try:
        torch.manual_seed(seed)
        m = module_factory()
        g = torch.Generator().manual_seed(seed + 7919)
        sd = OrderedDict()
        for k, v in m.state_dict().items():
            if k.endswith("running_mean"):
                v = torch.randn(v.shape, generator=g) * 0.05
            elif k.endswith("running_var"):
                v = 1.0 + 0.2 * torch.rand(v.shape, generator=g)
            elif k.endswith("num_batches_tracked"):
                v = v.clone()
            elif v.dim() == 1 and _is_bn_gamma(k, m):
                v = 1.0 + 0.1 * torch.randn(v.shape, generator=g)
            sd[k] = v.clone()
        return sd
    finally:
        torch.random.set_rng_state(gen_state)


def _is_bn_gamma(key, module):
    if not key.endswith(".weight"):
        return False
    owner = module
    for part in key.split(".")[:-1]:
        owner = getattr(owner, part) if not part.isdigit() else owner[int(part)]
    return isinstance(owner, nn.BatchNorm2d)
