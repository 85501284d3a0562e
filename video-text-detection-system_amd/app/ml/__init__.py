"""Same public names as the reference's app/ml/__init__.py:1-22 (training symbols are out of scope)."""
from .models.text_detector import TextDetector, DBNet
from .models.text_recognizer import TextRecognizer, CRNN, TransformerRecognizer
from .inference.pipeline import VideoTextPipeline
from .utils.preprocessing import VideoProcessor, ImageProcessor

__version__ = "1.0.0"
__all__ = ["TextDetector", "DBNet", "TextRecognizer", "CRNN", "TransformerRecognizer", "VideoTextPipeline",
           "VideoProcessor", "ImageProcessor"]
