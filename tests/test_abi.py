"""The C-ABI library builds, loads and exports every symbol include/vtd.h declares (no compute calls)."""
import ctypes
import os
import re

from vtd_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "vtd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vtd_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__
    __graft_entry__.build()
    lib = ctypes.CDLL(_native.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 10
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/vtd.h but not exported"
    assert sorted(_native.SIGNATURES) == declared, "python binding table and header disagree"


def test_error_strings_and_argument_validation_without_device():
    lib = _native.load()
    assert lib.vtd_version().startswith(b"vtd_hip")
    assert lib.vtd_strerror(0) == b"ok"
    assert b"state dict" in lib.vtd_strerror(-1103)
    h = ctypes.c_void_p()
    assert lib.vtd_detector_create(b"vgg16", 1, ctypes.byref(h)) == -1100  # only the two reference plans
    assert lib.vtd_detector_create(b"resnet18", 0, ctypes.byref(h)) == -1100
    assert lib.vtd_detector_create(b"resnet18", 2, ctypes.byref(h)) == 0
    buf = (ctypes.c_float * 4)()
    assert lib.vtd_detector_set_tensor(h, b"not.a.key", buf, 4) == -1101
    assert lib.vtd_detector_set_tensor(h, b"backbone.1.num_batches_tracked", buf, 1) == 0
    assert lib.vtd_detector_forward(h, 1, buf, None, None) == -1104  # not finalized
    lib.vtd_detector_destroy(h)
