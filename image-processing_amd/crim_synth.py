"""Seeded synthetic X-ray-like images and masks for the benchmark and the parity tests.

The reference ships no benchmark inputs; these follow the recipe fixed for this project's
measurements: a smooth field 120 + 70*sin(x/23+p1)*cos(y/31+p2), three soft ellipses, Gaussian
noise (sigma 12), clipped to uint8; "grey" replicates the plane to three channels, "colour"
applies a fixed false-colour map plus per-channel noise (sigma 8).  Masks stay at least
``margin`` pixels away from every border so that target patches are never clipped in timed runs.
"""
import numpy as np


def xray_like(height, width, seed, colour=False):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:height, 0:width].astype(np.float64)
    p1, p2 = rng.random(2) * 2 * np.pi
    f = 120.0 + 70.0 * np.sin(x / 23.0 + p1) * np.cos(y / 31.0 + p2)
    for _ in range(3):
        cy, cx = rng.random() * height, rng.random() * width
        ay, ax = (0.08 + 0.2 * rng.random()) * height, (0.08 + 0.2 * rng.random()) * width
        amp = rng.uniform(-50, 50)
        f += amp * np.exp(-(((y - cy) / ay) ** 2 + ((x - cx) / ax) ** 2))
    f += rng.normal(0.0, 12.0, (height, width))
    g = np.clip(np.rint(f), 0, 255).astype(np.uint8)
    if not colour:
        return np.ascontiguousarray(np.repeat(g[:, :, None], 3, axis=2))
    t = g.astype(np.float64) / 255.0
    b = 255.0 * np.clip(1.5 - np.abs(4 * t - 1), 0, 1)
    gg = 255.0 * np.clip(1.5 - np.abs(4 * t - 2), 0, 1)
    r = 255.0 * np.clip(1.5 - np.abs(4 * t - 3), 0, 1)
    img = np.stack([b, gg, r], axis=2) + rng.normal(0.0, 8.0, (height, width, 3))
    return np.ascontiguousarray(np.clip(np.rint(img), 0, 255).astype(np.uint8))


def blob_mask(height, width, seed, fraction=0.10, margin=9):
    """One filled disc plus two rectangles covering about `fraction` of the frame."""
    rng = np.random.default_rng(seed + 7919)
    m = np.zeros((height, width), np.uint8)
    target = int(round(fraction * height * width))
    yy, xx = np.mgrid[0:height, 0:width]
    lo, hi_y, hi_x = margin, height - margin, width - margin
    rad = max(2, int(np.sqrt(0.45 * target / np.pi)))
    cy = int(rng.integers(lo + rad, max(lo + rad + 1, hi_y - rad)))
    cx = int(rng.integers(lo + rad, max(lo + rad + 1, hi_x - rad)))
    m[(yy - cy) ** 2 + (xx - cx) ** 2 <= rad * rad] = 1
    for _ in range(64):
        left = target - int(m.sum())
        if left <= 0:
            break
        side_h = max(2, int(np.sqrt(left / 2.0) * rng.uniform(0.7, 1.3)))
        side_w = max(2, min(left // side_h if side_h else 2, hi_x - lo - 1))
        side_h = min(side_h, hi_y - lo - 1)
        y0 = int(rng.integers(lo, max(lo + 1, hi_y - side_h)))
        x0 = int(rng.integers(lo, max(lo + 1, hi_x - side_w)))
        m[y0:y0 + side_h, x0:x0 + side_w] = 1
    m[:margin] = 0
    m[-margin:] = 0
    m[:, :margin] = 0
    m[:, -margin:] = 0
    return m


def square_hole_mask(height, width, side):
    """Centred side x side hole (the "25 % contiguous hole" of the large-image configs)."""
    m = np.zeros((height, width), np.uint8)
    y0, x0 = (height - side) // 2, (width - side) // 2
    m[y0:y0 + side, x0:x0 + side] = 1
    return m
