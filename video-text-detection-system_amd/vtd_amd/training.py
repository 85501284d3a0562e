"""Training-side surface of the reference (app/ml/training/trainer.py), first slice: the loss its training / validation step computes.

``DiceLoss`` has the reference's constructor and ``forward(pred, target)`` (trainer.py:130-142); ``detection_loss(outputs, targets)``
is the four-scalar body of ``TextDetectionLightningModule.training_step`` / ``validation_step`` (trainer.py:48-56, 66-71):

    prob_loss   = nn.BCELoss()(outputs['probability'], targets['probability_map'])
    thresh_loss = nn.BCELoss()(outputs['threshold'],   targets['threshold_map'])
    dice_loss   = DiceLoss()(outputs['probability'],   targets['probability_map'])
    total_loss  = prob_loss + thresh_loss + dice_loss

Both run as ONE HBM-bound HIP pass over the maps (include/vtd.h: vtd_dbloss_forward; csrc/dbloss.hip) -- forward only: the scalars
are plain (detached) float32 tensors.  Backward, AdamW and ReduceLROnPlateau (trainer.py:107-128) are not built yet; like every
other product entry there is no CPU fallback (tensors must live on the GPU, the library must be present).
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _native


def _f32_cuda(t, name):
    if not torch.is_tensor(t) or not t.is_cuda:
        raise _native.NativeError(f"{name}: a CUDA (HIP) tensor is required -- the loss has no CPU path")
    t = t.detach()
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _run(prob, thresh, prob_t, thresh_t, smooth, want_sums=False):
    lib = _native.require()
    prob, prob_t = _f32_cuda(prob, "pred"), _f32_cuda(prob_t, "target")
    if prob.numel() != prob_t.numel() or prob.numel() == 0:
        raise ValueError(f"pred and target must have the same, non-zero number of elements ({prob.numel()} vs {prob_t.numel()})")
    if (thresh is None) != (thresh_t is None):
        raise ValueError("threshold map and threshold target come together")
    if thresh is not None:
        thresh, thresh_t = _f32_cuda(thresh, "threshold"), _f32_cuda(thresh_t, "threshold target")
        if thresh.numel() != prob.numel() or thresh_t.numel() != prob.numel():
            raise ValueError("threshold maps must have the probability map's element count")
    ws = torch.empty(int(lib.vtd_dbloss_workspace_bytes()), dtype=torch.uint8, device=prob.device)
    out = torch.empty(4, dtype=torch.float32, device=prob.device)
    sums = torch.empty(5, dtype=torch.float64, device=prob.device) if want_sums else None
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    _native.check(lib.vtd_dbloss_forward(ptr(prob), ptr(thresh), ptr(prob_t), ptr(thresh_t), prob.numel(), float(smooth), ptr(ws), ptr(out),
                                         ptr(sums), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "vtd_dbloss_forward")
    return out, sums


class DiceLoss(nn.Module):
    """trainer.py:130-142: ``1 - (2 sum(pred * target) + smooth) / (sum(pred) + sum(target) + smooth)`` over the flattened maps."""

    def __init__(self, smooth: float = 1e-5):
        super(DiceLoss, self).__init__()
        self.smooth = smooth

    def forward(self, pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        out, _ = _run(pred, None, target, None, self.smooth)
        return out[2]


def detection_loss(outputs, targets, smooth: float = 1e-5, want_sums: bool = False):
    """The body of training_step / validation_step (trainer.py:48-56): {'loss', 'prob_loss', 'thresh_loss', 'dice_loss'} as 0-d float32
    tensors on the device (one pass over the four maps, one 16-byte result).  ``outputs`` is DBNet's output dict
    ({'probability', 'threshold'}), ``targets`` the batch's {'probability_map', 'threshold_map'}."""
    out, sums = _run(outputs["probability"], outputs["threshold"], targets["probability_map"], targets["threshold_map"], smooth, want_sums)
    res = {"prob_loss": out[0], "thresh_loss": out[1], "dice_loss": out[2], "loss": out[3]}
    if want_sums:
        res["sums"] = sums
    return res
