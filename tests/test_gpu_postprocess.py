"""GPU parity: the HIP post-process (union-find CCL + lattice area + row-extent hull + calipers) against the
C oracle (Suzuki border following + shoelace + hull of the traced contour).  Integer outputs must be
bit-exact and in the same order; the float confidence within 1e-6."""
import numpy as np
import pytest
import torch

from oracle import cstages
from vtd_amd._fixtures import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["frame-kernel", "ten-launches"])
def pp_mode(request, monkeypatch):
    """Every case runs on the ten launches (default) and on the one-workgroup-per-frame kernel that can replace them (VTD_PP_FUSED=1):
    both must match the oracle bit for bit, so they match one another."""
    monkeypatch.setenv("VTD_PP_FUSED", "1" if request.param == "frame-kernel" else "0")


def _check(pp, maps, sizes, thr):
    prob = torch.from_numpy(np.stack(maps)).cuda()
    got = pp.run(prob, [s[0] for s in sizes], [s[1] for s in sizes], thr, debug=True)
    total = 0
    for i, m in enumerate(maps):
        exp = cstages.postprocess(m, sizes[i][0], sizes[i][1], thr, max_det=pp.max_out, with_debug=True)
        assert len(got[i]) == len(exp), (i, len(got[i]), len(exp))
        for g, e in zip(got[i], exp):
            assert g["_first"] == e["_first"]
            assert g["_area"] == e["_area"], (g, e)
            assert g["polygon"] == e["polygon"], (g, e)
            assert g["bbox"] == e["bbox"], (g, e)
            if np.isnan(e["confidence"]):
                assert np.isnan(g["confidence"])
            else:
                assert abs(g["confidence"] - e["confidence"]) <= 1e-6
        total += len(exp)
    return total


@pytest.fixture(scope="module")
def pp640(hip):
    from vtd_amd.engine import PostProcessor
    p = PostProcessor(8, 640, 640, max_out=4096)
    yield p
    p.close()


def test_margin_maps_rotated_rectangles(pp640):
    maps = [synth.margin_prob_map(100 + i) for i in range(8)]
    sizes = [(1280, 720)] * 4 + [(1920, 1080)] * 2 + [(640, 480), (640, 640)]
    assert _check(pp640, maps, sizes, 0.5) >= 30


def test_blobs_with_holes_and_islands(pp640):
    from scipy import ndimage
    rng = np.random.default_rng(42)
    maps = []
    for t in range(8):
        base = ndimage.gaussian_filter(rng.standard_normal((640, 640)), 3.0 + t)
        maps.append(((base > np.quantile(base, 0.55 + 0.04 * (t % 3))) * 0.8 + 0.1).astype(np.float32))
    assert _check(pp640, maps, [(1280, 720)] * 8, 0.5) > 20


def test_constructed_topologies(pp640):
    m = [np.full((640, 640), 0.1, np.float32) for _ in range(8)]
    m[0][100:300, 100:300] = 0.9; m[0][150:250, 150:250] = 0.1; m[0][180:220, 180:220] = 0.9  # ring + island
    m[1][:] = 0.9                                                                              # everything foreground
    m[2][0:40, 0:640] = 0.9; m[2][600:640, 0:100] = 0.9; m[2][200:400, 639] = 0.9               # frame contact, 1-px column
    for k in range(20):                                                                        # diagonal staircase (8-conn only)
        m[3][100 + 12 * k:112 + 12 * k, 100 + 12 * k:112 + 12 * k] = 0.9
    m[4][300:320, 50:600] = 0.9; m[4][100:500, 300:320] = 0.9                                  # plus sign
    yy, xx = np.mgrid[0:640, 0:640]
    m[5][(xx - 320) ** 2 + (yy - 320) ** 2 <= 200 ** 2] = 0.9                                  # disc: many hull vertices
    m[5][(xx - 320) ** 2 + (yy - 320) ** 2 <= 150 ** 2] = 0.1
    m[5][(xx - 320) ** 2 + (yy - 320) ** 2 <= 100 ** 2] = 0.9
    m[6][10:21, 10:21] = 0.9; m[6][10:22, 40:52] = 0.9; m[6][10:20, 70:90] = 0.9; m[6][50:62, 10:23] = 0.9  # filter edges
    m[7][((xx + yy) % 2 == 0) & (xx > 200) & (xx < 300) & (yy > 200) & (yy < 300)] = 0.9      # checkerboard (holes are 4-conn)
    _check(pp640, m, [(1280, 720)] * 8, 0.5)
    # threshold strictness and NaNs
    t = [np.full((640, 640), 0.5, np.float32), np.full((640, 640), np.nan, np.float32)]
    t[0][100:200, 100:300] = np.float32(0.5000001)
    _check(pp640, t, [(640, 640)] * 2, 0.5)


def test_reference_style_random_maps_160(hip):
    """tests/test_models.py:48-58,170-183 feed 160x160 uniform-random maps at several thresholds."""
    from vtd_amd.engine import PostProcessor
    pp = PostProcessor(4, 160, 160, max_out=512)
    try:
        rng = np.random.default_rng(0)
        for thr in (0.3, 0.5, 0.7, 0.9):
            maps = [rng.random((160, 160)).astype(np.float32) for _ in range(4)]
            _check(pp, maps, [(640, 480)] * 4, thr)
        maps = [np.full((160, 160), 0.8, np.float32)] * 2  # test_models.py:148-152
        _check(pp, maps, [(640, 480)] * 2, 0.5)
    finally:
        pp.close()


def test_noise_stress_many_components(pp640):
    rng = np.random.default_rng(5)
    maps = [(rng.random((640, 640)) < q).astype(np.float32) * 0.8 + 0.1 for q in (0.2, 0.45, 0.55, 0.7)]
    _check(pp640, maps, [(1920, 1080)] * 4, 0.5)
