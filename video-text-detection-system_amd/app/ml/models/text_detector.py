"""app.ml.models.text_detector (reference: app/ml/models/text_detector.py) -> MI355X implementation."""
from vtd_amd.detector import TextDetector  # noqa: F401
from vtd_amd.nets import DBNet, DBHead, FeaturePyramidNetwork  # noqa: F401
