"""Result sink: CSV / XML exporters byte-identical to the reference's on the golden input produced by the
reference's own functions (tests/golden/make_golden_export.py), plus the columnar table."""
import json
import os

import numpy as np

from vtd_amd import export


def _golden(golden_dir, name, mode="r"):
    with open(os.path.join(golden_dir, name), mode, newline="" if mode == "r" else None) as f:
        return f.read()


def test_csv_and_xml_match_the_reference_byte_for_byte(golden_dir):
    data = json.loads(_golden(golden_dir, "export_input.json"))
    assert export.export_results_csv(data) == _golden(golden_dir, "export_expected.csv")
    assert export.export_results_xml(data) == _golden(golden_dir, "export_expected.xml")
    empty = json.loads(_golden(golden_dir, "export_expected_empty.json"))
    assert export.export_results_csv({}) == empty["csv_empty"]
    assert export.export_results_xml({}) == empty["xml_empty"]


def test_export_error_convention():
    # the reference logs and returns "" on any failure (processing_service.py:88-90,133-135)
    assert export.export_results_csv({"results": [{"detections": [{"bbox": None}]}]}) == ""
    assert export.export_results_xml({"results": 5}) == ""


def test_columnar_table_round_trip(golden_dir):
    data = json.loads(_golden(golden_dir, "export_input.json"))
    t = export.detections_table(data)
    flat = [(f["frame_number"], f["timestamp"], d) for f in data["results"] for d in f["detections"]]
    assert t["frame_number"].tolist() == [r[0] for r in flat]
    assert t["bbox"].tolist() == [r[2]["bbox"] for r in flat]
    assert np.allclose(t["detection_confidence"], [r[2]["detection_confidence"] for r in flat])
    blob = t["text_utf8"].tobytes()
    texts = [blob[t["text_offsets"][i]:t["text_offsets"][i + 1]].decode("utf-8") for i in range(len(flat))]
    assert texts == [r[2]["text"] for r in flat]
    assert export.detections_table({})["bbox"].shape == (0, 4)
