import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "vag-nmt_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """Returns (meta dict, P dict of torch tensors, raw npz dict)."""
    z = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    meta = json.loads(bytes(z.pop("meta")).decode())
    P = {k[2:]: torch.from_numpy(v) for k, v in z.items() if k.startswith("P/")}
    return meta, P, z


GOLDEN_CASES = ["mm_dot_tied_s0_f32", "mm_dot_tied_s0_f64", "mm_mlp_untied_s1_f32", "mm_mlp_untied_s1_f64",
                "text_tied_s0_f32", "text_untied_s1_f64", "mm_dot_tied_mid_f32", "mm_dot_full_len_f32"]


@pytest.fixture(scope="session")
def golden_loader():
    return load_golden
