"""app.ml.training.trainer (reference: app/ml/training/trainer.py) -> MI355X implementation, first slice: the loss (forward)."""
from vtd_amd.training import DiceLoss, detection_loss  # noqa: F401
