"""Seeded random-shape sweep of the whole training path against the CPU oracle: odd batch sizes, lengths, vocabulary
sizes, hidden sizes (multiples of 4, the library's alignment unit), both attention scores, tied/untied output
embedding, multimodal/text-only, teacher forcing/free running.  Loss and every parameter gradient are compared."""
import random

import pytest
import torch

from test_gpu_edge_and_full import make, run_both

pytestmark = pytest.mark.gpu


def _case(seed):
    r = random.Random(seed)
    B = r.choice([1, 2, 3, 5, 7, 16, 17, 33])
    Ts = r.randint(1, 11)
    Tt = r.randint(1, 9)
    E = 4 * r.randint(2, 9)
    H = 4 * r.randint(2, 13)
    S = 4 * r.randint(2, 9)
    I = 4 * r.randint(4, 40)
    Vs = r.randint(8, 90)
    Vt = r.randint(8, 130)
    lens = sorted([r.randint(1, Ts) for _ in range(B)], reverse=True)
    lens[0] = Ts
    return dict(B=B, Ts=Ts, Tt=Tt, E=E, H=H, S=S, I=I, Vs=Vs, Vt=Vt, lens=lens, attn=r.choice(["dot", "mlp"]),
                tied=r.random() < 0.5, kind=r.choice(["mm", "mm", "text"]), teacher=r.random() < 0.7)


@pytest.mark.parametrize("seed", list(range(24)))
def test_random_shapes_match_oracle(seed):
    c = _case(1000 + seed)
    m, src, tgt, im = make(c["Vs"], c["Vt"], c["I"], c["E"], c["H"], c["S"], c["B"], c["Ts"], c["Tt"], c["lens"],
                           seed=seed, attn=c["attn"], tied=c["tied"], kind=c["kind"])
    run_both(m, src, c["lens"], tgt, im if c["kind"] == "mm" else None, teacher=c["teacher"], check_grads=c["teacher"],
             kind=c["kind"])
