"""Golden vectors for the result sink: the reference's CSV / XML exporters
(/root/reference/app/services/processing_service.py:59-137) run in this container on a fixed result dict.

    python tests/golden/make_golden_export.py

The module is executed from its source file with stand-ins for what its import block needs but the two exporters never
touch (cv2, celery, app.config); only data is written: export_input.json, export_expected.csv, export_expected.xml.
"""
import asyncio
import importlib.machinery
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def main():
    class _Celery:
        def __init__(self, *a, **k):
            pass

        def config_from_object(self, *a, **k):
            pass

    _stub("cv2")
    _stub("celery", Celery=_Celery)
    for pkg in ("refsvc", "refsvc.services"):
        _stub(pkg).__path__ = []
    _stub("refsvc.config", settings=types.SimpleNamespace(celery_broker_url="", celery_result_backend=""))
    path = os.path.join(REF, "app/services/processing_service.py")
    mod = types.ModuleType("refsvc.services.processing_service")
    mod.__file__ = path
    mod.__package__ = "refsvc.services"
    sys.modules[mod.__name__] = mod
    with open(path) as f:
        exec(compile(f.read(), path, "exec"), mod.__dict__)
    svc = mod.ProcessingService()

    data = {
        "status": "success",
        "results": [
            {"frame_number": 0, "timestamp": 0.0, "detections": [
                {"bbox": [10, 20, 110, 52], "text": "HELLO, world", "detection_confidence": 0.9375, "recognition_confidence": 0.5,
                 "polygon": [[5, 10], [55, 10], [55, 26], [5, 26]]},
                {"bbox": [300, 400, 420, 440], "text": 'quote " and <tag> & amp', "detection_confidence": 0.75,
                 "recognition_confidence": 0.123456789, "polygon": []}]},
            {"frame_number": 3, "timestamp": 0.3, "detections": []},
            {"frame_number": 6, "timestamp": 0.6000000000000001, "detections": [
                {"bbox": [0, 0, 1280, 720], "text": "", "detection_confidence": 1.0, "recognition_confidence": 0.0, "polygon": []},
                {"bbox": [1, 2, 3, 4], "text": "line\nbreak", "detection_confidence": 0.5, "recognition_confidence": 0.25}]},
        ],
        "summary": {"total_frames": 3, "frames_with_text": 2, "total_detections": 4, "unique_texts": 3,
                    "detected_texts": ["HELLO, world", "line\nbreak"], "avg_detection_confidence": 0.796875,
                    "avg_recognition_confidence": 0.21836419725, "processing_time_seconds": 1.5, "fps_processed": 2.0},
    }
    json.dump(data, open(os.path.join(HERE, "export_input.json"), "w"), indent=1)
    csv_text = asyncio.run(svc.export_results_csv(data))
    xml_text = asyncio.run(svc.export_results_xml(data))
    with open(os.path.join(HERE, "export_expected.csv"), "w", newline="") as f:
        f.write(csv_text)
    with open(os.path.join(HERE, "export_expected.xml"), "w", newline="") as f:
        f.write(xml_text)
    # degenerate inputs the exporters accept
    json.dump({"csv_empty": asyncio.run(svc.export_results_csv({})), "xml_empty": asyncio.run(svc.export_results_xml({}))},
              open(os.path.join(HERE, "export_expected_empty.json"), "w"), indent=1)
    print("csv bytes", len(csv_text), "xml bytes", len(xml_text))


if __name__ == "__main__":
    main()
