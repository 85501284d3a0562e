"""app.ml.models.text_recognizer (reference: app/ml/models/text_recognizer.py) -> MI355X implementation."""
from vtd_amd.nets import CRNN  # noqa: F401
from vtd_amd.recognizer import TextRecognizer, TransformerRecognizer  # noqa: F401
