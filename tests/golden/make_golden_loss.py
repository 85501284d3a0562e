"""Generates tests/golden/dbloss.npz from the REFERENCE's own loss code.

Runs only in the build container (needs /root/reference).  The unmodified app/ml/training/trainer.py is exec'd with an inert stub for
pytorch_lightning (absent in this image; only its names are needed at import time: LightningModule as a base class) -- DiceLoss is a
plain nn.Module and nn.BCELoss is torch's.  The vectors are the four scalars TextDetectionLightningModule.training_step computes
(trainer.py:52-56, evaluated with the reference's DiceLoss instance and nn.BCELoss exactly as that method does) on seeded maps,
including exact 0 / 1 probabilities (torch's -100 clamp) and a map size that is not a multiple of four.

    python tests/golden/make_golden_loss.py
"""
import importlib.machinery
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def load_trainer():
    pl = _stub("pytorch_lightning", LightningModule=torch.nn.Module, Trainer=object)
    pl.callbacks = _stub("pytorch_lightning.callbacks")
    pl.loggers = _stub("pytorch_lightning.loggers")
    path = os.path.join(REF, "app/ml/training/trainer.py")
    mod = types.ModuleType("ref_trainer")
    mod.__file__ = path
    sys.modules["ref_trainer"] = mod
    with open(path) as f:
        exec(compile(f.read(), path, "exec"), mod.__dict__)
    return mod


def cases():
    g = torch.Generator().manual_seed(20260)
    out = {}
    for name, shape in (("b2_64x80", (2, 1, 64, 80)), ("b1_37x53", (1, 1, 37, 53)), ("b3_640", (3, 1, 640, 640))):
        prob = torch.sigmoid(torch.randn(shape, generator=g) * 3.0)
        thresh = torch.sigmoid(torch.randn(shape, generator=g))
        prob_t = (torch.rand(shape, generator=g) < 0.2).float()
        thresh_t = 0.3 + 0.4 * torch.rand(shape, generator=g)
        if name == "b2_64x80":   # saturated outputs: log(0) on both sides, clamped at -100 by torch
            prob.view(-1)[:7] = torch.tensor([0.0, 1.0, 0.0, 1.0, 1e-30, 1.0 - 6e-8, 0.5])
            prob_t.view(-1)[:7] = torch.tensor([1.0, 0.0, 0.0, 1.0, 1.0, 0.0, 1.0])
            thresh.view(-1)[:2] = torch.tensor([0.0, 1.0])
        out[name] = (prob, thresh, prob_t, thresh_t)
    return out


def main():
    tr = load_trainer()
    dice_loss, bce_loss = tr.DiceLoss(), torch.nn.BCELoss()     # TextDetectionLightningModule.__init__, trainer.py:39-40
    blob = {}
    for name, (prob, thresh, prob_t, thresh_t) in cases().items():
        outputs = {"probability": prob, "threshold": thresh}
        targets = {"probability_map": prob_t, "threshold_map": thresh_t}
        prob_loss = bce_loss(outputs["probability"], targets["probability_map"])          # trainer.py:52-56, verbatim order
        thresh_loss = bce_loss(outputs["threshold"], targets["threshold_map"])
        d = dice_loss(outputs["probability"], targets["probability_map"])
        total = prob_loss + thresh_loss + d
        small = prob.numel() <= 20000
        if small:   # inputs of the small cases are stored; the 640 x 640 case is regenerated from its seed (tests check the recipe's hash)
            blob[name + "/prob"], blob[name + "/thresh"] = prob.numpy(), thresh.numpy()
            blob[name + "/prob_t"], blob[name + "/thresh_t"] = prob_t.numpy(), thresh_t.numpy()
        blob[name + "/expected"] = np.array([float(prob_loss), float(thresh_loss), float(d), float(total)], np.float64)
        blob[name + "/shape"] = np.array(prob.shape)
        print(name, blob[name + "/expected"])
    np.savez_compressed(os.path.join(HERE, "dbloss.npz"), **blob)


if __name__ == "__main__":
    main()
