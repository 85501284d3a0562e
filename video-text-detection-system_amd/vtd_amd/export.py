"""Result sink of the path: the on-wire forms the reference builds from a ``process_video`` result dict.

* ``export_results_csv`` / ``export_results_xml``: same text, byte for byte, as the reference's exporters
  (app/services/processing_service.py:59-137) for the same dict -- pinned by tests/golden/export_expected.* which were
  produced by running the reference's functions (tests/golden/make_golden_export.py).  Same error convention: any failure
  is logged and an empty string returned.
* ``detections_table``: the bulk / columnar hand-off the survey asks for instead of one object per detection: flat numpy
  columns (frame index, timestamp, bbox, confidences, text offsets into one UTF-8 blob) that a database bulk insert or an
  Arrow / Parquet writer takes in one call.
"""
import io
import logging
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


def _csv_field(value):
    """csv.writer's default dialect (excel, QUOTE_MINIMAL, CRLF rows): quote when the text holds a comma, a quote or a line
    break, double embedded quotes; numbers go through str() (floats by repr, as the csv module does)."""
    text = "" if value is None else (repr(value) if isinstance(value, float) else str(value))
    if any(ch in text for ch in ',"\r\n'):
        return '"' + text.replace('"', '""') + '"'
    return text


def export_results_csv(results_data: Dict[str, Any]) -> str:
    try:
        out = io.StringIO()
        out.write(",".join(CSV_HEADER) + "\r\n")
        for number, stamp, det in _rows(results_data):
            box = det.get("bbox", [0, 0, 0, 0])
            fields = (number, stamp, det.get("text", ""), box[0], box[1], box[2], box[3], det.get("detection_confidence", 0.0),
                      det.get("recognition_confidence", 0.0))
            out.write(",".join(_csv_field(f) for f in fields) + "\r\n")
        return out.getvalue()
    except Exception as e:
        logger.error(f"CSV export failed: {e}")
        return ""


def _xml_text(s):
    return s.replace("&", "&amp;").replace("<", "&lt;").replace(">", "&gt;")


def _xml_attr(s):
    # ElementTree's attribute escaping: markup characters, the double quote, and line breaks / tabs as character references
    s = _xml_text(s).replace('"', "&quot;")
    return s.replace("\r", "&#13;").replace("\n", "&#10;").replace("\t", "&#09;")


def _element(tag, attrs="", body=None):
    if body is None or body == "":
        return f"<{tag}{attrs} />"
    return f"<{tag}{attrs}>{body}</{tag}>"


def export_results_xml(results_data: Dict[str, Any]) -> str:
    try:
        summary = "".join(_element(key, body=_xml_text(str(value))) for key, value in results_data.get("summary", {}).items())
        frames = []
        for frame in results_data.get("results", []):
            objects = []
            for det in frame.get("detections", []):
                x1, y1, x2, y2 = det.get("bbox", [0, 0, 0, 0])[:4]
                points = "".join(f'<Point x="{_xml_attr(str(x))}" y="{_xml_attr(str(y))}" />' for x, y in ((x1, y1), (x2, y1), (x2, y2), (x1, y2)))
                attrs = (f' transcription="{_xml_attr(det.get("text", ""))}"'
                         f' detection_confidence="{_xml_attr(str(det.get("detection_confidence", 0.0)))}"'
                         f' recognition_confidence="{_xml_attr(str(det.get("recognition_confidence", 0.0)))}"')
                objects.append(_element("object", attrs, points))
            attrs = f' number="{_xml_attr(str(frame.get("frame_number", 0)))}" timestamp="{_xml_attr(str(frame.get("timestamp", 0.0)))}"'
            frames.append(_element("frame", attrs, "".join(objects)))
        return _element("video_text_detection", body=_element("summary", body=summary) + _element("frames", body="".join(frames)))
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
