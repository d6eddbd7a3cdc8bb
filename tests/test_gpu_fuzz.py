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


@pytest.mark.parametrize("seed", list(range(10)))
def test_random_decode_matches_oracle(seed):
    """Greedy and beam search (graph replay and launch by launch) against the oracle's restatement of
    models/...V11.py:179-337 on random small configurations."""
    import numpy as np
    from oracle import vag_oracle as O
    r = random.Random(5000 + seed)
    B = r.choice([1, 2, 3, 5, 8, 16])
    Ts = r.randint(1, 13)
    E, H, S, I = 4 * r.randint(2, 8), 4 * r.randint(3, 12), 4 * r.randint(2, 8), 4 * r.randint(4, 20)
    Vs, Vt = r.randint(10, 80), r.randint(12, 150)
    k = r.choice([2, 3, 5, 12])
    max_len = r.randint(2, 19)
    lens = sorted([r.randint(1, Ts) for _ in range(B)], reverse=True)
    lens[0] = Ts
    m, src, _, im = make(Vs, Vt, I, E, H, S, B, Ts, 3, lens, seed=seed, attn=r.choice(["dot", "mlp"]),
                         tied=r.random() < 0.5)
    P = {n: p.detach().clone() for n, p in m.named_parameters()}
    want_g = O.greedy_decode(P, src, lens, im, max_length=max_len, attn=m.attn_model)
    want_b, want_scores = O.beam_search(P, src, lens, im, beam_size=k, max_length=max_len, return_scores=True,
                                        attn=m.attn_model)
    mg = m.cuda().eval()
    for graph in (True, False):
        mg.decode_graph = graph
        got_g = [[int(t) for t in h] for h in mg.beamsearch_decode(src.cuda(), lens, im.cuda(), 1, max_len)]
        got_b = [[int(t) for t in h] for h in mg.beamsearch_decode(src.cuda(), lens, im.cuda(), k, max_len)]
        assert got_g == want_g, (graph, "greedy")
        assert np.allclose(mg.last_beam_scores.cpu().numpy(), want_scores.numpy(), rtol=1e-4, atol=1e-4), (graph, "scores")
        assert got_b == want_b, (graph, "beam")
