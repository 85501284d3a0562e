"""Result sink of the path: the on-wire forms the reference builds from a ``process_video`` result dict.

* ``export_results_csv`` / ``export_results_xml``: same text, byte for byte, as the reference's exporters
  (app/services/processing_service.py:59-137) for the same dict -- pinned by tests/golden/export_expected.* which were
  produced by running the reference's functions (tests/golden/make_golden_export.py).  Same error convention: any failure
  is logged and an empty string returned.
* ``detections_table``: the bulk / columnar hand-off the survey asks for instead of one object per detection: flat numpy
  columns (frame index, timestamp, bbox, confidences, text offsets into one UTF-8 blob) that a database bulk insert or an
  Arrow / Parquet writer takes in one call.
"""
import csv
import io
import logging
import xml.etree.ElementTree as ET
from typing import Any, Dict

import numpy as np

logger = logging.getLogger(__name__)

CSV_HEADER = ("frame_number", "timestamp", "text", "bbox_x1", "bbox_y1", "bbox_x2", "bbox_y2", "detection_confidence",
              "recognition_confidence")


def _rows(results_data):
    """(frame_number, timestamp, detection dict) for every detection, frames in order."""
    for frame in results_data.get("results", []):
        number, stamp = frame.get("frame_number", 0), frame.get("timestamp", 0.0)
        for det in frame.get("detections", []):
            yield number, stamp, det


def _csv_records(results_data):
    for number, stamp, det in _rows(results_data):
        box = det.get("bbox", [0, 0, 0, 0])
        yield (number, stamp, det.get("text", ""), box[0], box[1], box[2], box[3], det.get("detection_confidence", 0.0),
               det.get("recognition_confidence", 0.0))


def export_results_csv(results_data: Dict[str, Any]) -> str:
    """processing_service.py:59-91: the csv module's default dialect does the quoting (the standard library is the
    specification here, so it is used, not imitated)."""
    try:
        out = io.StringIO()
        writer = csv.writer(out)
        writer.writerow(CSV_HEADER)
        writer.writerows(_csv_records(results_data))
        return out.getvalue()
    except Exception as e:
        logger.error(f"CSV export failed: {e}")
        return ""


def _build(parent, spec):
    """spec = (tag, {attr: value}, text or None, [child specs]) -> ElementTree nodes under parent."""
    tag, attrs, text, children = spec
    node = ET.Element(tag) if parent is None else ET.SubElement(parent, tag)
    for key, value in attrs.items():
        node.set(key, value if isinstance(value, str) else str(value))
    if text is not None:
        node.text = text
    for child in children:
        _build(node, child)
    return node


def _object_spec(det):
    x1, y1, x2, y2 = det.get("bbox", [0, 0, 0, 0])[:4]
    corners = ((x1, y1), (x2, y1), (x2, y2), (x1, y2))  # clockwise from the top-left corner
    return ("object", {"transcription": det.get("text", ""),
                       "detection_confidence": str(det.get("detection_confidence", 0.0)),
                       "recognition_confidence": str(det.get("recognition_confidence", 0.0))}, None,
            [("Point", {"x": str(x), "y": str(y)}, None, []) for x, y in corners])


def export_results_xml(results_data: Dict[str, Any]) -> str:
    """processing_service.py:93-137: <video_text_detection><summary/><frames><frame><object><Point/>x4 ... serialised by
    ElementTree itself (escaping, attribute order = insertion order, ' />' for empty elements)."""
    try:
        summary = ("summary", {}, None, [(key, {}, str(value), []) for key, value in results_data.get("summary", {}).items()])
        frames = ("frames", {}, None,
                  [("frame", {"number": str(fr.get("frame_number", 0)), "timestamp": str(fr.get("timestamp", 0.0))}, None,
                    [_object_spec(det) for det in fr.get("detections", [])]) for fr in results_data.get("results", [])])
        return ET.tostring(_build(None, ("video_text_detection", {}, None, [summary, frames])), encoding="unicode")
    except Exception as e:
        logger.error(f"XML export failed: {e}")
        return ""


def detections_table(results_data: Dict[str, Any]) -> Dict[str, np.ndarray]:
    """Columnar view of every detection of a result dict: one numpy array per column, texts as one UTF-8 blob with offsets."""
    rows = list(_rows(results_data))
    n = len(rows)
    table = {
        "frame_number": np.fromiter((r[0] for r in rows), dtype=np.int64, count=n),
        "timestamp": np.fromiter((r[1] for r in rows), dtype=np.float64, count=n),
        "bbox": np.array([r[2].get("bbox", [0, 0, 0, 0])[:4] for r in rows], dtype=np.int32).reshape(n, 4),
        "detection_confidence": np.fromiter((r[2].get("detection_confidence", 0.0) for r in rows), dtype=np.float32, count=n),
        "recognition_confidence": np.fromiter((r[2].get("recognition_confidence", 0.0) for r in rows), dtype=np.float32, count=n),
    }
    encoded = [r[2].get("text", "").encode("utf-8") for r in rows]
    offsets = np.zeros(n + 1, dtype=np.int64)
    np.cumsum([len(b) for b in encoded], out=offsets[1:])
    table["text_offsets"] = offsets
    table["text_utf8"] = np.frombuffer(b"".join(encoded), dtype=np.uint8).copy()
    return table
