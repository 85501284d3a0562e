"""Drop-in overlay: put this directory ahead of the reference's on PYTHONPATH (or copy `app/ml` over it) and the
Celery worker's `from app.ml.inference.pipeline import VideoTextPipeline` resolves to the MI355X path."""
