"""fp32 torch-CPU restatement of the reference networks, driven by a reference-format state dict.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Functional on purpose: it shares no code with
the product's parameter containers, only the state-dict key contract.

  dbnet_forward   DBNet.forward             app/ml/models/text_detector.py:25-29, with the one
                                            documented repair: the FPN taps C2..C5
  trunk_taps      resnet children()[:-2]    text_detector.py:17-19 (torchvision 0.16.1 ResNet v1.5,
                                            absent here -> PARITY UNPINNED, restated from the public
                                            definition)
  fpn_forward     FeaturePyramidNetwork     text_detector.py:43-56 (intended multi-scale wiring)
  db_branch       DBHead.probability_head   text_detector.py:61-70
  crnn_forward    CRNN.forward              app/ml/models/text_recognizer.py:29-37
"""
import torch
import torch.nn.functional as F

_STAGES = {"resnet18": ("basic", (2, 2, 2, 2)), "resnet50": ("bottleneck", (3, 4, 6, 3))}


def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        False, 0.0, 1e-5)


def _block(x, sd, p, kind, stride):
    idt = x
    if kind == "basic":
        y = F.relu(_bn(F.conv2d(x, sd[p + ".conv1.weight"], None, stride, 1), sd, p + ".bn1"))
        y = _bn(F.conv2d(y, sd[p + ".conv2.weight"], None, 1, 1), sd, p + ".bn2")
    else:
        y = F.relu(_bn(F.conv2d(x, sd[p + ".conv1.weight"]), sd, p + ".bn1"))
        y = F.relu(_bn(F.conv2d(y, sd[p + ".conv2.weight"], None, stride, 1), sd, p + ".bn2"))
        y = _bn(F.conv2d(y, sd[p + ".conv3.weight"]), sd, p + ".bn3")
    if (p + ".downsample.0.weight") in sd:
        idt = _bn(F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride), sd, p + ".downsample.1")
    return F.relu(y + idt)


def trunk_taps(x, sd, backbone, prefix="backbone."):
    kind, counts = _STAGES[backbone]
    y = F.relu(_bn(F.conv2d(x, sd[prefix + "0.weight"], None, 2, 3), sd, prefix + "1"))
    y = F.max_pool2d(y, 3, 2, 1)
    taps = []
    for stage, n in enumerate(counts):
        for b in range(n):
            stride = 2 if (b == 0 and stage > 0) else 1
            y = _block(y, sd, f"{prefix}{4 + stage}.{b}", kind, stride)
        taps.append(y)
    return taps  # C2, C3, C4, C5


def fpn_forward(taps, sd, prefix="fpn."):
    c2, c3, c4, c5 = taps
    feats = [c5, c4, c3, c2]
    last = F.conv2d(feats[0], sd[prefix + "inner_blocks.0.weight"], sd[prefix + "inner_blocks.0.bias"])
    for i in range(1, 4):
        lat = F.conv2d(feats[i], sd[prefix + f"inner_blocks.{i}.weight"], sd[prefix + f"inner_blocks.{i}.bias"])
        last = lat + F.interpolate(last, scale_factor=2, mode="nearest")
    return F.conv2d(last, sd[prefix + "layer_blocks.3.weight"], sd[prefix + "layer_blocks.3.bias"], 1, 1)


def db_branch(p2, sd, prefix, logits=False):
    y = F.conv2d(p2, sd[prefix + "0.weight"], sd[prefix + "0.bias"], 1, 1)
    y = F.relu(_bn(y, sd, prefix + "1"))
    y = F.conv_transpose2d(y, sd[prefix + "3.weight"], sd[prefix + "3.bias"], 2)
    y = F.relu(_bn(y, sd, prefix + "4"))
    y = F.conv_transpose2d(y, sd[prefix + "6.weight"], sd[prefix + "6.bias"], 2)
    return y if logits else torch.sigmoid(y)


@torch.no_grad()
def dbnet_forward(x, sd, backbone, want_threshold=False, return_taps=False):
    taps = trunk_taps(x, sd, backbone)
    p2 = fpn_forward(taps, sd)
    out = {"probability": db_branch(p2, sd, "head.probability_head."),
           "threshold": db_branch(p2, sd, "head.threshold_head.") if want_threshold else None}
    if return_taps:
        out["taps"] = taps
        out["p2"] = p2
    return out


_CRNN_CONVS = ((0, 1, 1, "p22"), (4, 5, 1, "p22"), (8, 9, 1, None), (11, 12, 1, "p21"), (15, 16, 1, None),
               (18, 19, 1, "p21"), (22, 23, 0, None))


def crnn_cnn(x, sd):
    y = x
    for ci, bi, pad, pool in _CRNN_CONVS:
        y = F.relu(_bn(F.conv2d(y, sd[f"cnn.{ci}.weight"], sd[f"cnn.{ci}.bias"], 1, pad), sd, f"cnn.{bi}"))
        if pool == "p22":
            y = F.max_pool2d(y, 2, 2)
        elif pool == "p21":
            y = F.max_pool2d(y, (2, 1), (2, 1))
    return y  # [B,512,1,31]


def _lstm_dir(seq, w_ih, w_hh, b_ih, b_hh, reverse):
    # PyTorch gate order i,f,g,o; zero initial state (text_recognizer.py:26,34)
    B, T, _ = seq.shape
    H = w_hh.shape[1]
    h = seq.new_zeros(B, H)
    c = seq.new_zeros(B, H)
    xs = seq @ w_ih.t() + (b_ih + b_hh)
    out = seq.new_zeros(B, T, H)
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        g = xs[:, t] + h @ w_hh.t()
        i, f, gg, o = g.chunk(4, dim=1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        out[:, t] = h
    return out


@torch.no_grad()
def crnn_forward(x, sd, return_cnn=False, return_layers=False):
    """return_cnn: (logits, cnn features); return_layers: (logits, cnn features, [h0, h1]) with the two LSTM layer
    outputs as [B,31,512] (forward units first, as nn.LSTM concatenates them)."""
    feat = crnn_cnn(x, sd)
    b, c, h, w = feat.shape
    seq = feat.reshape(b, c * h, w).permute(0, 2, 1)
    layers = []
    for layer in range(2):
        fw = _lstm_dir(seq, sd[f"rnn.weight_ih_l{layer}"], sd[f"rnn.weight_hh_l{layer}"],
                       sd[f"rnn.bias_ih_l{layer}"], sd[f"rnn.bias_hh_l{layer}"], False)
        bw = _lstm_dir(seq, sd[f"rnn.weight_ih_l{layer}_reverse"], sd[f"rnn.weight_hh_l{layer}_reverse"],
                       sd[f"rnn.bias_ih_l{layer}_reverse"], sd[f"rnn.bias_hh_l{layer}_reverse"], True)
        seq = torch.cat([fw, bw], dim=2)
        layers.append(seq)
    logits = seq @ sd["classifier.weight"].t() + sd["classifier.bias"]
    if return_layers:
        return logits, feat, layers
    return (logits, feat) if return_cnn else logits
