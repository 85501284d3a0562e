"""Deterministic synthetic weights (no checkpoints ship with the reference and nothing can be fetched).

* ``default_state_dict``   seeded torch default init (+ randomised BN statistics): tensor-level tolerance tests.
* ``margin_detector_state_dict``  "margin weights" (SURVEY.md section 8d): the full dense network is still
  evaluated, but one carrier channel computes a *decision with margin* early, on values that are exact
  multiples of 1/255, and is then passed through untouched to the probability map.  With default-init weights
  ~2 % of pixels sit within 1e-3 of the threshold, so box/IoU parity between an fp16 GPU path and an fp32 CPU
  path would be a coin flip; with these weights every pixel is >= 0.4 away from the decision boundary at the
  point where the decision is taken, while every layer still runs at full size.

  Carrier path (R18; R50 analogous through its bottlenecks):
    stem    ch0 = luminance of the centre tap            -> n/255, n integer (the resized frame is uint8)
    pool    max over 3x3 keeps it on the 1/255 lattice
    layer1  block0: t = relu(255*x0 - 127.5)   block1: B = 2*relu(t) - 2*relu(t - 0.5)  in {0, ~1}
    FPN     only the C2 lateral feeds channel 0; smooth / head conv / ConvT pass it with unit centre taps
    ConvT6  logit = 16*B - 8
"""
from collections import OrderedDict

import torch

from . import nets

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def default_state_dict(kind, seed=0, **kw):
    factory = {"resnet18": lambda: nets.DBNet("resnet18"), "resnet50": lambda: nets.DBNet("resnet50"),
               "crnn": lambda: nets.CRNN(kw.get("vocab_size", 97))}[kind]
    return nets.seeded_state_dict(factory, seed)


def _plain_init(backbone, seed):
    state = torch.random.get_rng_state()
    try:
        torch.manual_seed(seed)
        sd = OrderedDict((k, v.clone()) for k, v in nets.DBNet(backbone).state_dict().items())
    finally:
        torch.random.set_rng_state(state)
    return sd


def _bn_identity(sd, prefix, ch, beta=0.0):
    sd[prefix + ".weight"][ch] = 1.0
    sd[prefix + ".bias"][ch] = beta
    sd[prefix + ".running_mean"][ch] = 0.0
    sd[prefix + ".running_var"][ch] = 1.0 - 1e-5  # gamma / sqrt(var + eps) == 1 exactly


def _row(sd, key, ch, taps=None):
    """Zero every weight feeding output channel `ch`, then set the listed (in_ch, r, s, value) taps."""
    w = sd[key]
    w[ch].zero_()
    for cin, r, s, val in (taps or ()):
        w[ch, cin, r, s] = val


def margin_detector_state_dict(backbone="resnet18", seed=0, gain=16.0):
    sd = _plain_init(backbone, seed)
    r50 = backbone == "resnet50"

    # stem: channel 0 = luminance of the centre pixel (un-normalised: sum_c std_c/3 * x_c + mean(mean_c) = v/255 for gray frames)
    for ch in range(4):
        _row(sd, "backbone.0.weight", ch)
        _bn_identity(sd, "backbone.1", ch)
    for c in range(3):
        sd["backbone.0.weight"][0, c, 3, 3] = STD[c] / 3.0
    sd["backbone.1.bias"][0] = sum(MEAN) / 3.0

    l1 = "backbone.4."
    if not r50:
        # block0: ch1 = relu(255*x0 - 127.5)
        _row(sd, l1 + "0.conv1.weight", 1, [(0, 1, 1, 255.0)])
        _bn_identity(sd, l1 + "0.bn1", 1, beta=-127.5)
        for ch in (0, 1, 2, 3):
            _row(sd, l1 + "0.conv2.weight", ch, [(1, 1, 1, 1.0)] if ch == 1 else None)
            _bn_identity(sd, l1 + "0.bn2", ch)
        # block1: ch2 = 2*relu(t) - 2*relu(t - 0.5)
        _row(sd, l1 + "1.conv1.weight", 1, [(1, 1, 1, 1.0)])
        _bn_identity(sd, l1 + "1.bn1", 1)
        _row(sd, l1 + "1.conv1.weight", 2, [(1, 1, 1, 1.0)])
        _bn_identity(sd, l1 + "1.bn1", 2, beta=-0.5)
        for ch in (0, 1, 2, 3):
            _row(sd, l1 + "1.conv2.weight", ch, [(1, 1, 1, 2.0), (2, 1, 1, -2.0)] if ch == 2 else None)
            _bn_identity(sd, l1 + "1.bn2", ch)
    else:
        # block0 (with 64->256 downsample): ch1 = relu(255*x0 - 127.5); ch2 stays 0
        _row(sd, l1 + "0.conv1.weight", 1, [(0, 0, 0, 255.0)])
        _bn_identity(sd, l1 + "0.bn1", 1, beta=-127.5)
        _row(sd, l1 + "0.conv2.weight", 1, [(1, 1, 1, 1.0)])
        _bn_identity(sd, l1 + "0.bn2", 1)
        for ch in (0, 1, 2, 3):
            _row(sd, l1 + "0.conv3.weight", ch, [(1, 0, 0, 1.0)] if ch == 1 else None)
            _bn_identity(sd, l1 + "0.bn3", ch)
            _row(sd, l1 + "0.downsample.0.weight", ch)
            _bn_identity(sd, l1 + "0.downsample.1", ch)
        # block1: ch2 = 2*relu(t) - 2*relu(t - 0.5)
        _row(sd, l1 + "1.conv1.weight", 1, [(1, 0, 0, 1.0)])
        _bn_identity(sd, l1 + "1.bn1", 1)
        _row(sd, l1 + "1.conv1.weight", 2, [(1, 0, 0, 1.0)])
        _bn_identity(sd, l1 + "1.bn1", 2, beta=-0.5)
        for ch in (1, 2):
            _row(sd, l1 + "1.conv2.weight", ch, [(ch, 1, 1, 1.0)])
            _bn_identity(sd, l1 + "1.bn2", ch)
        for ch in (0, 1, 2, 3):
            _row(sd, l1 + "1.conv3.weight", ch, [(1, 0, 0, 2.0), (2, 0, 0, -2.0)] if ch == 2 else None)
            _bn_identity(sd, l1 + "1.bn3", ch)
        # block2: identity on the carrier
        for ch in (0, 1, 2, 3):
            _row(sd, l1 + "2.conv3.weight", ch)
            _bn_identity(sd, l1 + "2.bn3", ch)

    # FPN: channel 0 <- C2 carrier only
    for i in range(4):
        _row(sd, f"fpn.inner_blocks.{i}.weight", 0, [(2, 0, 0, 1.0)] if i == 3 else None)
        sd[f"fpn.inner_blocks.{i}.bias"][0] = 0.0
    _row(sd, "fpn.layer_blocks.3.weight", 0, [(0, 1, 1, 1.0)])
    sd["fpn.layer_blocks.3.bias"][0] = 0.0

    for head in ("head.probability_head.", "head.threshold_head."):
        _row(sd, head + "0.weight", 0, [(0, 1, 1, 1.0)])
        sd[head + "0.bias"][0] = 0.0
        _bn_identity(sd, head + "1", 0)
        w3 = sd[head + "3.weight"]  # [cin, cout, 2, 2]
        w3[:, 0] = 0.0
        w3[0, 0] = 1.0
        sd[head + "3.bias"][0] = 0.0
        _bn_identity(sd, head + "4", 0)
        w6 = sd[head + "6.weight"]  # [64, 1, 2, 2]
        w6.mul_(0.25)
        w6[0, 0] = gain
        sd[head + "6.bias"][0] = -gain / 2.0
    return sd
