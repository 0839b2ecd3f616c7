import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def load_image(name):
    from PIL import Image
    return np.asarray(Image.open(os.path.join(GOLDEN, "images", name)).convert("RGB"))


def seeded_normal(shape, seed):
    import torch
    g = torch.Generator().manual_seed(int(seed))
    return torch.randn(shape, generator=g, dtype=torch.float32)


def mtcnn_state_dicts():
    import torch
    d = os.path.join(REPO, "vn_celeb_face_recognition_amd", "weights_mtcnn")
    return tuple(torch.load(os.path.join(d, n + ".pt"), weights_only=True) for n in ("pnet", "rnet", "onet"))
