"""app.ml.utils.preprocessing (reference: app/ml/utils/preprocessing.py) -- frame-source seam only."""
from vtd_amd.video import VideoProcessor, ImageProcessor  # noqa: F401
