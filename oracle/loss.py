"""CPU restatement of the reference's training loss (TEST INFRASTRUCTURE: only tests/, smoke() and bench.py's cpu_baseline leg may import
this package; the product path never does).

Follows app/ml/training/trainer.py:48-56 (training_step: two nn.BCELoss terms + DiceLoss, summed left to right) and :130-142 (DiceLoss),
in float64 numpy so that it is the exact value the float32 implementations approximate.  Pinned by tests/golden/dbloss.npz, which
tests/golden/make_golden_loss.py produced by running the reference's own DiceLoss class and torch's nn.BCELoss (the reference's choice)."""
import numpy as np


def bce(pred, target):
    """nn.BCELoss(reduction='mean'): mean of -(t * max(log p, -100) + (1 - t) * max(log(1 - p), -100)) (torch clamps both logs at -100)."""
    p = np.asarray(pred, np.float64).reshape(-1)
    t = np.asarray(target, np.float64).reshape(-1)
    with np.errstate(divide="ignore"):
        l0 = np.maximum(np.log(p), -100.0)
        l1 = np.maximum(np.log1p(-p), -100.0)
    return float(np.mean(-(t * l0 + (1.0 - t) * l1)))


def dice(pred, target, smooth=1e-5):
    """trainer.py:135-142."""
    p = np.asarray(pred, np.float64).reshape(-1)
    t = np.asarray(target, np.float64).reshape(-1)
    inter = float((p * t).sum())
    return 1.0 - (2.0 * inter + smooth) / (float(p.sum()) + float(t.sum()) + smooth)


def detection_loss(outputs, targets, smooth=1e-5):
    """trainer.py:52-56."""
    prob = bce(outputs["probability"], targets["probability_map"])
    thresh = bce(outputs["threshold"], targets["threshold_map"])
    d = dice(outputs["probability"], targets["probability_map"], smooth)
    return {"prob_loss": prob, "thresh_loss": thresh, "dice_loss": d, "loss": prob + thresh + d}
