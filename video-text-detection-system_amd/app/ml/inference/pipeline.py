"""app.ml.inference.pipeline -- the module name every caller imports (the reference file is misspelt
`pipeliine.py`)."""
from vtd_amd.pipeline import VideoTextPipeline  # noqa: F401
