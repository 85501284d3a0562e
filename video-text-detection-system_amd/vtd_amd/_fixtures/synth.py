"""Seeded synthetic inputs for the BASELINE configs.

Frames are uint8 BGR HWC, C-contiguous: black background, 6-12 white filled rotated rectangles
(text-line proxies: 40-400 px long, 16-48 px high, angle in [-30,30] degrees) and N(0,4) noise.
Throughput runs use uniformly random frames (conv cost is content independent).
"""
import numpy as np


def text_frame(seed, height=720, width=1280, n_boxes=None, noise_sigma=4.0):
    rng = np.random.default_rng(seed)
    img = np.zeros((height, width), np.float32)
    n = int(rng.integers(6, 13)) if n_boxes is None else n_boxes
    rects = []
    occupied = np.zeros((height, width), bool)
    tries = 0
    while len(rects) < n and tries < 200:
        tries += 1
        length = float(rng.uniform(40, 400))
        thick = float(rng.uniform(16, 48))
        ang = float(np.deg2rad(rng.uniform(-30, 30)))
        cx = float(rng.uniform(0.12 * width, 0.88 * width))
        cy = float(rng.uniform(0.12 * height, 0.88 * height))
        half = 0.5 * np.hypot(length, thick) + 2
        x0, x1 = int(max(0, cx - half)), int(min(width, cx + half + 1))
        y0, y1 = int(max(0, cy - half)), int(min(height, cy + half + 1))
        yy, xx = np.mgrid[y0:y1, x0:x1].astype(np.float32)
        u = (xx - cx) * np.cos(ang) + (yy - cy) * np.sin(ang)
        v = -(xx - cx) * np.sin(ang) + (yy - cy) * np.cos(ang)
        m = (np.abs(u) <= length / 2) & (np.abs(v) <= thick / 2)
        grown = (np.abs(u) <= length / 2 + 12) & (np.abs(v) <= thick / 2 + 12)
        if (occupied[y0:y1, x0:x1] & grown).any():
            continue  # keep text lines separated so they stay distinct components after the 640^2 resize
        occupied[y0:y1, x0:x1] |= grown
        img[y0:y1, x0:x1][m] = 255.0
        rects.append({"cx": cx, "cy": cy, "length": length, "thick": thick, "angle": ang})
    img = img + rng.normal(0.0, noise_sigma, img.shape).astype(np.float32)
    frame = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    return np.ascontiguousarray(np.repeat(frame[:, :, None], 3, axis=2)), rects


def random_frames(seed, n, height=720, width=1280):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, (n, height, width, 3), dtype=np.uint8)


def margin_prob_map(seed, h=640, w=640, n_boxes=8, lo=0.1, hi=0.9):
    """Synthetic probability map with margin: rotated rectangles at `hi` on `lo` with a 1-px linear
    ramp edge (post-process-only parity fixtures)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    cover = np.zeros((h, w), np.float32)
    placed = 0
    tries = 0
    while placed < n_boxes and tries < 200:
        tries += 1
        length = float(rng.uniform(0.06 * w, 0.45 * w))
        thick = float(rng.uniform(0.03 * h, 0.09 * h))
        ang = float(np.deg2rad(rng.uniform(-40, 40)))
        cx, cy = float(rng.uniform(0.15 * w, 0.85 * w)), float(rng.uniform(0.15 * h, 0.85 * h))
        u = (xx - cx) * np.cos(ang) + (yy - cy) * np.sin(ang)
        v = -(xx - cx) * np.sin(ang) + (yy - cy) * np.cos(ang)
        d = np.maximum(np.abs(u) - length / 2, np.abs(v) - thick / 2)  # signed distance-ish
        c = np.clip(0.5 - d, 0.0, 1.0)
        if (cover[(d < 6)] > 0).any():
            continue
        cover = np.maximum(cover, c)
        placed += 1
    return (lo + (hi - lo) * cover).astype(np.float32)


def glyph_crop(seed, height=None, width=None, noise_sigma=4.0):
    """One text-line-like crop (uint8 BGR, gray): white ground with dark vertical / horizontal strokes of random
    size and position plus N(0,4) noise.  Content differs strongly from seed to seed, which is what the recogniser's
    tensor-level parity tests need (uniform-noise images all look alike to a CRNN)."""
    rng = np.random.default_rng(seed)
    h = int(rng.integers(20, 60)) if height is None else height
    w = int(rng.integers(60, 400)) if width is None else width
    img = np.full((h, w), 255.0, np.float32)
    x = int(rng.integers(2, 10))
    while x < w - 6:
        sw = int(rng.integers(2, 7))
        kind = int(rng.integers(0, 4))
        y0 = int(rng.integers(0, max(1, h // 3)))
        y1 = int(rng.integers(2 * h // 3, h))
        if kind == 0:
            img[y0:y1, x:x + sw] = 0
        elif kind == 1:
            img[y0:y0 + sw, x:x + 3 * sw] = 0
        elif kind == 2:
            img[y1 - sw:y1, x:x + 3 * sw] = 0
            img[y0:y1, x:x + sw] = 0
        else:
            mid = (y0 + y1) // 2
            img[mid:mid + sw, x:x + 3 * sw] = 0
        x += int(rng.integers(6, 20))
    img += rng.normal(0.0, noise_sigma, img.shape).astype(np.float32)
    g = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    return np.ascontiguousarray(np.repeat(g[:, :, None], 3, axis=2))


def glyph_batch(seed, n):
    """[n,3,32,128] float32 CRNN inputs in the reference's format (text_recognizer.py:118-119: BGR/255, CHW) made of
    32x128 glyph crops (no resize involved)."""
    x = np.stack([glyph_crop(seed * 1000 + i, 32, 128) for i in range(n)])
    return np.ascontiguousarray(x.transpose(0, 3, 1, 2)).astype(np.float32) / np.float32(255.0)
