"""Oracle byte/geometry stages: Pillow resample pinned against PIL itself (G6); the OpenCV-derived
post-process restatement (PARITY UNPINNED vs cv2) cross-checked against independent scipy
constructions and closed-form cases."""
import numpy as np
import pytest
from PIL import Image

from oracle import cstages
from vtd_amd._fixtures import synth


@pytest.mark.parametrize("shape", [(720, 1280), (1080, 1920), (480, 640), (640, 640), (300, 500), (1000, 37)])
def test_g6_pil_resize_bit_exact(shape):
    rng = np.random.default_rng(shape[0])
    img = rng.integers(0, 256, (*shape, 3), dtype=np.uint8)
    ref = np.asarray(Image.fromarray(img).resize((640, 640), Image.BILINEAR))
    got = cstages.pil_resize_bilinear(img, 640, 640)
    assert np.array_equal(got, ref)


def test_pil_resize_on_structured_frame():
    frame, _ = synth.text_frame(0, 720, 1280)
    rgb = np.ascontiguousarray(frame[..., ::-1])
    ref = np.asarray(Image.fromarray(rgb).resize((640, 640), Image.BILINEAR))
    assert np.array_equal(cstages.pil_resize_bilinear(rgb), ref)


def test_cv_resize_identity_constant_and_area_path():
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (32, 128, 3), dtype=np.uint8)
    assert np.array_equal(cstages.cv_resize_linear(img), img)  # same size -> all taps on pixel centres
    const = np.full((57, 211, 3), 173, np.uint8)
    assert np.all(cstages.cv_resize_linear(const) == 173)
    big = rng.integers(0, 256, (64, 256, 3), dtype=np.uint8)  # exact 2x -> 2x2 box average, round half up
    exp = (big.reshape(32, 2, 128, 2, 3).astype(np.int32).sum(axis=(1, 3)) + 2) >> 2
    assert np.array_equal(cstages.cv_resize_linear(big), exp.astype(np.uint8))
    # crops are views into a frame (pipeliine.py:121): strided input must equal the contiguous copy
    frame = rng.integers(0, 256, (200, 300, 3), dtype=np.uint8)
    view = frame[20:77, 31:250]
    assert np.array_equal(cstages.cv_resize_linear(view), cstages.cv_resize_linear(view.copy()))
    # monotone ramp stays monotone and within range
    ramp = np.repeat(np.linspace(0, 255, 300).astype(np.uint8)[None, :, None], 40, 0).repeat(3, 2)
    out = cstages.cv_resize_linear(ramp)
    assert np.all(np.diff(out[0, :, 0].astype(int)) >= 0)


def _rect_map(h, w, y0, y1, x0, x1, lo=0.1, hi=0.9):
    m = np.full((h, w), lo, np.float32)
    m[y0:y1, x0:x1] = hi
    return m


def test_postprocess_axis_aligned_rectangle():
    m = _rect_map(640, 640, 100, 141, 200, 401)  # 201 x 41 pixels
    dets = cstages.postprocess(m, 1280, 720, 0.5, with_debug=True)
    assert len(dets) == 1
    d = dets[0]
    assert d["_area"] == 200 * 40  # contour through pixel centres: (w-1)(h-1)
    assert sorted(map(tuple, d["polygon"])) == sorted([(200, 100), (400, 100), (400, 140), (200, 140)])
    assert d["bbox"] == [int(200 * 1280 / 640), int(100 * 720 / 640), int(400 * 1280 / 640), int(140 * 720 / 640)]
    x1, y1, x2, y2 = d["bbox"]  # text_detector.py:169-170: the box is re-projected with floor division
    ref_conf = float(np.mean(m[y1 * 640 // 720:y2 * 640 // 720, x1 * 640 // 1280:x2 * 640 // 1280]))
    assert abs(d["confidence"] - ref_conf) < 1e-6
    assert d["_first"] == (200, 100)


def test_postprocess_filters_and_strict_threshold():
    m = _rect_map(640, 640, 10, 21, 10, 21)  # 11x11 -> area 100 -> kept by area, w,h = 10 -> dropped by size
    assert cstages.postprocess(m, 640, 640, 0.5) == []
    m = _rect_map(640, 640, 10, 22, 10, 22)  # 12x12 -> area 121, size 11 > 10
    assert len(cstages.postprocess(m, 640, 640, 0.5)) == 1
    m = _rect_map(640, 640, 10, 20, 10, 30)  # 20x10 -> area 19*9=171 but h=9 -> dropped
    assert cstages.postprocess(m, 640, 640, 0.5) == []
    m = _rect_map(640, 640, 10, 20, 10, 21)  # 11x10 -> area 90 < 100
    assert cstages.postprocess(m, 640, 640, 0.5) == []
    m = _rect_map(640, 640, 50, 100, 50, 100, lo=0.2, hi=0.5)  # p == thr is NOT foreground
    assert cstages.postprocess(m, 640, 640, 0.5) == []
    assert len(cstages.postprocess(m, 640, 640, np.float32(0.49999))) == 1


def test_postprocess_external_only_and_order():
    m = np.full((640, 640), 0.1, np.float32)
    m[100:300, 100:300] = 0.9
    m[150:250, 150:250] = 0.1  # hole
    m[180:220, 180:220] = 0.9  # island inside the hole: not an external contour
    m[400:450, 50:120] = 0.9   # second external component, discovered later in raster order
    m[400:450, 500:600] = 0.9  # third, same first row, further right
    dets = cstages.postprocess(m, 640, 640, 0.5, with_debug=True)
    assert [d["_first"] for d in dets] == [(500, 400), (50, 400), (100, 100)]  # reverse discovery order
    assert dets[2]["_area"] == 199 * 199  # the hole does not reduce the outer border's area
    # frame-border contact and the virtual 1-px zero padding
    m = _rect_map(640, 640, 0, 30, 0, 640)
    d = cstages.postprocess(m, 640, 640, 0.5, with_debug=True)
    assert len(d) == 1 and d[0]["_area"] == 639 * 29 and d[0]["bbox"] == [0, 0, 639, 29]


def test_postprocess_area_identity_and_externality_vs_scipy():
    """Independent construction: external components = 8-connected components of the complement of the
    frame-connected (4-connected) background; traced-border area = #2x2 blocks fully inside the filled
    component + half the blocks with exactly three pixels inside."""
    from scipy import ndimage
    rng = np.random.default_rng(7)
    for trial in range(6):
        h, w = (96, 128) if trial < 4 else (160, 160)
        base = ndimage.gaussian_filter(rng.standard_normal((h, w)), 2.0 + trial % 3)
        m = (base > np.quantile(base, 0.62)).astype(np.float32) * 0.8 + 0.1
        dets = cstages.postprocess(m, 640, 640, 0.5, with_debug=True)
        mask = m > 0.5
        bg = ~mask
        bg_pad = np.pad(bg, 1, constant_values=True)
        lab_bg, _ = ndimage.label(bg_pad)  # 4-connected
        outside = (lab_bg == lab_bg[0, 0])[1:-1, 1:-1]
        filled = ~outside
        lab, n = ndimage.label(filled, structure=np.ones((3, 3)))
        expect = {}
        for k in range(1, n + 1):
            comp = lab == k
            c = np.pad(comp, 1).astype(np.int32)
            q = c[:-1, :-1] + c[1:, :-1] + c[:-1, 1:] + c[1:, 1:]
            area = float((q == 4).sum() + 0.5 * (q == 3).sum())
            ys, xs = np.nonzero(comp)
            first = (int(xs[ys == ys.min()].min()), int(ys.min()))
            expect[first] = area
        got = {d["_first"]: d["_area"] for d in dets}
        for first, area in got.items():
            assert first in expect and expect[first] == area
        # every sufficiently large component that also passes the box-size filter must be reported
        assert all(a >= 100 for a in got.values())
        assert len(got) <= sum(1 for a in expect.values() if a >= 100)


def test_min_area_box_rotated_rectangles():
    """Rasterised rotated rectangles: the recovered box must wrap the pixels tightly (<= 1.5 px slack)."""
    from scipy.spatial import ConvexHull
    rng = np.random.default_rng(11)
    for _ in range(40):
        cx, cy = rng.uniform(150, 490, 2)
        wl, hl = rng.uniform(60, 250), rng.uniform(14, 60)
        th = np.deg2rad(rng.uniform(-80, 80))
        yy, xx = np.mgrid[0:640, 0:640]
        u = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th)
        v = -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
        mask = (np.abs(u) <= wl / 2) & (np.abs(v) <= hl / 2)
        ys, xs = np.nonzero(mask)
        pts = np.stack([xs, ys], 1)
        box = cstages.min_area_box(pts).astype(np.float64)
        # all pixel centres inside the (truncated) box grown by 1.5 px; box area close to hull min-area
        c = box.mean(0)
        e0, e1 = box[1] - box[0], box[3] - box[0]
        l0, l1 = np.linalg.norm(e0), np.linalg.norm(e1)
        d0 = e0 / l0 if l0 >= l1 else e1 / l1  # long side; int truncation skews thin boxes, so use its normal
        a = (pts - c) @ d0
        b = (pts - c) @ np.array([-d0[1], d0[0]])
        if l0 < l1:
            l0, l1 = l1, l0
        assert np.all(np.abs(a) <= l0 / 2 + 1.5) and np.all(np.abs(b) <= l1 / 2 + 1.5)
        hull = pts[ConvexHull(pts).vertices].astype(np.float64)
        best = np.inf
        for i in range(len(hull)):
            d = hull[(i + 1) % len(hull)] - hull[i]
            d /= np.linalg.norm(d)
            n = np.array([-d[1], d[0]])
            best = min(best, np.ptp(hull @ d) * np.ptp(hull @ n))
        assert abs(l0 * l1 - best) <= 1.5 * (l0 + l1)  # corners are truncated to ints: <= 1 px per side
