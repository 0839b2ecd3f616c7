"""Synthetic 1080p frames for the pipeline benchmark: a low-frequency
colour gradient + N(0,8) noise background with K real face crops pasted at seeded, non-overlapping
positions and sizes in [80,320] px.  The crops are the reference's own data/*.png pictures that
already ship as detector fixtures under tests/golden/images."""
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
FACE_FILES = ["041bc30432964f95871d4c223eba8f7c.png", "318c7ec3b94b451c813a5665cfcfbda3.png",
              "33f2891da9694198a67aabd1660517c3.png"]


def _faces():
    from PIL import Image
    d = os.path.join(os.path.dirname(_HERE), "tests", "golden", "images")
    return [Image.open(os.path.join(d, f)).convert("RGB") for f in FACE_FILES]


def make_frames(n_frames=64, faces_per_frame=8, height=1080, width=1920, seed=0):
    """-> (n_frames, H, W, 3) uint8 RGB, list of per-frame pasted boxes."""
    from PIL import Image
    rng = np.random.default_rng(seed)
    faces = _faces()
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float32)
    frames = np.empty((n_frames, height, width, 3), dtype=np.uint8)
    truth = []
    for f in range(n_frames):
        ph = rng.uniform(0, 2 * np.pi, size=3)
        bg = np.stack([96 + 48 * np.sin(xx / width * 2.1 + ph[c]) * np.cos(yy / height * 1.7 + ph[c] * 0.5)
                       for c in range(3)], axis=-1)
        bg += rng.normal(0, 8, size=bg.shape).astype(np.float32)
        img = np.clip(bg, 0, 255).astype(np.uint8)
        boxes = []
        tries = 0
        while len(boxes) < faces_per_frame and tries < 200:
            tries += 1
            s = int(rng.integers(80, 321))
            x0 = int(rng.integers(0, width - s))
            y0 = int(rng.integers(0, height - s))
            if any(not (x0 + s + 8 < b[0] or b[2] + 8 < x0 or y0 + s + 8 < b[1] or b[3] + 8 < y0) for b in boxes):
                continue
            face = faces[(f * faces_per_frame + len(boxes)) % len(faces)].resize((s, s), Image.BICUBIC)
            img[y0:y0 + s, x0:x0 + s] = np.asarray(face)
            boxes.append((x0, y0, x0 + s, y0 + s))
        frames[f] = img
        truth.append(boxes)
    return frames, truth
