"""ctypes front-end of oracle/csrc/oracle.c (test infrastructure only)."""
import ctypes as C

import numpy as np

from . import build as _build

_lib = None


class OrcDet(C.Structure):
    _fields_ = [("bbox", C.c_int * 4), ("poly", C.c_int * 8), ("conf", C.c_float), ("area", C.c_float),
                ("first_x", C.c_int), ("first_y", C.c_int)]


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_build.build())
        _lib.orc_pil_resize_bilinear.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_void_p, C.c_int, C.c_int]
        _lib.orc_cv_resize_linear_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_void_p, C.c_int, C.c_int]
        _lib.orc_postprocess.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                         C.POINTER(OrcDet), C.c_int]
        _lib.orc_min_area_box.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    return _lib


def pil_resize_bilinear(img_u8, out_h=640, out_w=640):
    """Pillow ``Image.resize((out_w,out_h), BILINEAR)`` on an HxWx3 uint8 array (text_detector.py:101)."""
    assert img_u8.dtype == np.uint8 and img_u8.ndim == 3 and img_u8.shape[2] == 3
    if img_u8.strides[2] != 1 or img_u8.strides[1] != 3:
        img_u8 = np.ascontiguousarray(img_u8)
    out = np.empty((out_h, out_w, 3), np.uint8)
    lib().orc_pil_resize_bilinear(img_u8.ctypes.data, img_u8.shape[0], img_u8.shape[1], img_u8.strides[0],
                                  out.ctypes.data, out_h, out_w)
    return out


def cv_resize_linear(img_u8, out_w=128, out_h=32):
    """``cv2.resize(img, (out_w, out_h))`` restatement (text_recognizer.py:118); accepts crop views."""
    assert img_u8.dtype == np.uint8 and img_u8.ndim == 3 and img_u8.shape[2] == 3
    if img_u8.strides[2] != 1 or img_u8.strides[1] != 3:
        img_u8 = np.ascontiguousarray(img_u8)
    out = np.empty((out_h, out_w, 3), np.uint8)
    rc = lib().orc_cv_resize_linear_u8(img_u8.ctypes.data, img_u8.shape[0], img_u8.shape[1], img_u8.strides[0],
                                       out.ctypes.data, out_h, out_w)
    if rc != 0:
        raise ValueError("empty source image")
    return out


def postprocess(prob_map, orig_width, orig_height, threshold, max_det=4096, with_debug=False):
    """``TextDetector._post_process`` restatement (text_detector.py:143-178)."""
    prob = np.ascontiguousarray(prob_map, dtype=np.float32)
    assert prob.ndim == 2
    buf = (OrcDet * max_det)()
    n = lib().orc_postprocess(prob.ctypes.data, prob.shape[0], prob.shape[1], int(orig_width), int(orig_height),
                              float(threshold), buf, max_det)
    out = []
    for i in range(min(n, max_det)):
        d = buf[i]
        poly = [[d.poly[2 * j], d.poly[2 * j + 1]] for j in range(4)]
        rec = {"bbox": [int(v) for v in d.bbox], "confidence": float(d.conf), "polygon": poly}
        if with_debug:
            rec["_area"] = float(d.area)
            rec["_first"] = (int(d.first_x), int(d.first_y))
        out.append(rec)
    return out


def min_area_box(points_xy):
    pts = np.ascontiguousarray(points_xy, dtype=np.int32).reshape(-1, 2)
    box = np.zeros(8, np.int32)
    lib().orc_min_area_box(pts.ctypes.data, len(pts), box.ctypes.data)
    return box.reshape(4, 2)
