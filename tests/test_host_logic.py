"""CPU: host-side behaviour of the drop-in modules (API surface, parameter names, pickling, loud failure on CPU)."""
import inspect
import io

import pytest
import torch

from conftest import load_golden


def models():
    from machine_translation_vision.models import NMT_AttentionImagine_Seq2Seq_Beam_V11, NMT_Seq2Seq_Beam_V2
    return NMT_AttentionImagine_Seq2Seq_Beam_V11, NMT_Seq2Seq_Beam_V2


def test_import_surface_matches_reference_names():
    from machine_translation_vision.models import NMT_AttentionImagine_Seq2Seq_Beam_V11, NMT_Seq2Seq_Beam_V2  # noqa
    from machine_translation_vision.layers import LIUMCVC_Encoder, NMT_Decoder, VSE_Imagine_Enc  # noqa
    from machine_translation_vision.losses import PairwiseRankingLoss, ImageRetrievalRankingLoss  # noqa
    from machine_translation_vision.utils.utils import l2norm  # noqa


def test_constructor_and_forward_signatures():
    V11, V2 = models()
    assert list(inspect.signature(V11.__init__).parameters)[1:] == [
        "src_size", "tgt_size", "im_feats_size", "src_embedding_size", "tgt_embedding_size", "hidden_size",
        "shared_embedding_size", "loss_w", "beam_size", "attn_model", "n_layers", "dropout_ctx", "dropout_emb",
        "dropout_out", "dropout_rnn_enc", "dropout_rnn_dec", "dropout_im_emb", "dropout_txt_emb", "activation_vse",
        "tied_emb", "init_split"]
    assert list(inspect.signature(V11.forward).parameters)[1:] == [
        "src_var", "src_lengths", "tgt_var", "im_var", "teacher_force_ratio", "max_length", "criterion_mt",
        "criterion_vse"]
    assert list(inspect.signature(V11.beamsearch_decode).parameters)[1:] == [
        "src_var", "src_lengths", "im_var", "beam_size", "max_length", "tgt_var"]
    assert list(inspect.signature(V2.__init__).parameters)[1:] == [
        "src_size", "tgt_size", "src_embedding_size", "tgt_embedding_size", "hidden_size", "beam_size", "n_layers",
        "dropout_ctx", "dropout_emb", "dropout_out", "dropout_rnn", "tied_emb"]
    assert list(inspect.signature(V2.forward).parameters)[1:] == [
        "src_var", "src_lengths", "tgt_var", "teacher_force_ratio", "max_length", "criterion"]


@pytest.mark.parametrize("name", ["mm_dot_tied_s0_f32", "mm_mlp_untied_s1_f32", "text_tied_s0_f32"])
def test_parameter_names_and_shapes_equal_the_reference(name):
    V11, V2 = models()
    meta, P, _ = load_golden(name)
    Vs, Vt, I, E, H, S, B, Ts, Tt = meta["dims"]
    if meta["kind"] == "mm":
        m = V11(Vs, Vt, I, E, E, H, S, 0.99, attn_model=meta["attn"], tied_emb=meta["tied"])
    else:
        m = V2(Vs, Vt, E, E, H, tied_emb=meta["tied"])
    ours = {n: tuple(p.shape) for n, p in m.named_parameters()}
    ref = {n: tuple(p.shape) for n, p in P.items()}
    assert ours == ref
    assert list(ours) == list(ref)      # same registration order as well
    missing, unexpected = m.load_state_dict(P, strict=False)
    assert not unexpected and set(missing) <= {"decoder.out.weight"}
    if meta["tied"]:
        assert m.decoder.out.weight is m.decoder.embedding.weight


def test_reset_parameters_statistics():
    V11, _ = models()
    torch.manual_seed(0)
    m = V11(500, 600, 256, 64, 64, 128, 96, 0.99, tied_emb=True)
    w = m.decoder.gru_2.weight_hh_l0          # kaiming_normal_: std = sqrt(2 / fan_in)
    assert abs(float(w.std()) - (2.0 / 128) ** 0.5) < 0.01
    assert float(m.encoder.embedding.weight[0].norm()) > 0.1      # the pad row is re-initialised too (V11.py:77-80)
    assert float(m.decoder.W1.bias.abs().max()) == 0.0            # bias_zero
    assert m.decoder.attn.v.dim() == 1


def test_whole_module_pickle_roundtrip():
    V11, _ = models()
    m = V11(50, 60, 96, 16, 16, 24, 20, 0.99, tied_emb=True)
    buf = io.BytesIO()
    torch.save(m, buf)                      # the reference checkpoints whole modules (nmt_multimodal_beam_DE.py:492)
    buf.seek(0)
    m2 = torch.load(buf, weights_only=False)
    for (n1, p1), (n2, p2) in zip(m.named_parameters(), m2.named_parameters()):
        assert n1 == n2 and torch.equal(p1, p2)
    assert m2.shared_embedding_size == 20 and m2.tgt_size == 60 and m2.hidden_size == 24


def test_cpu_tensors_fail_loudly_no_fallback():
    from vagnmt_hip._lib import VagError
    from machine_translation_vision.losses import PairwiseRankingLoss
    V11, _ = models()
    m = V11(50, 60, 96, 16, 16, 24, 20, 0.99)
    crit = torch.nn.NLLLoss(weight=torch.ones(60), reduction="none")
    with pytest.raises(VagError):
        m(torch.ones(2, 3, dtype=torch.long), [3, 3], torch.ones(2, 3, dtype=torch.long), torch.zeros(2, 96),
          criterion_mt=crit, criterion_vse=PairwiseRankingLoss(0.1))
    with pytest.raises(VagError):
        PairwiseRankingLoss(0.1)(torch.zeros(2, 4), torch.zeros(2, 4))
    with pytest.raises(VagError):
        m.beamsearch_decode(torch.ones(2, 3, dtype=torch.long), [3, 3], torch.zeros(2, 96), 3, 5)


def test_optimizer_grouping_follows_reference_rule():
    from vagnmt_hip.trainer import flat_layout, param_groups
    V11, _ = models()
    m = V11(50, 60, 96, 16, 16, 24, 20, 0.99, tied_emb=True)
    named = list(m.named_parameters())
    g = param_groups(named)
    assert [x[0] for x in g] == ["weight", "bias"]
    assert "decoder.attn.v" in g[0][1]            # 1-D but no 'bias' in its name -> weight-decayed (DE.py:306)
    assert all("bias" in n for n in g[1][1])
    g4 = param_groups(named, vse_separate=True)
    assert [x[3] for x in g4] == [1.0, 1.0, 0.5, 0.5]
    assert all("vse_imagine" in n for n in g4[2][1] + g4[3][1])
    groups, offs, seg, n = flat_layout(named)
    assert seg[0] == 0 and seg[-1] == n and all(o % 64 == 0 for o in offs.values())
    assert sum(len(x[1]) for x in groups) == len(named)


def test_eos_cut():
    from machine_translation_vision.models._seq2seq import Seq2SeqBase
    assert Seq2SeqBase._cut([[5, 6, 3, 7], [3, 1], [4, 4, 4]]) == [[5, 6], [], [4, 4, 4]]


def test_bucket_sampler_reproduces_reference_batches():
    """tests/golden/sampler.npz: batches of the reference's own BucketBatchSampler under numpy seed 5."""
    import os
    import numpy as np
    from conftest import GOLDEN
    from machine_translation_vision.samplers import BucketBatchSampler
    z = dict(np.load(os.path.join(GOLDEN, "sampler.npz")))
    lengths = list(z["lengths"])
    for bs in (16, 64):
        smp = BucketBatchSampler(lengths, bs)
        assert len(smp) == int(z["bs%d/n_batches" % bs][0])
        np.random.seed(5)
        batches = [np.asarray(b) for b in smp]
        assert [len(b) for b in batches] == list(z["bs%d/sizes" % bs])
        assert np.array_equal(np.concatenate(batches), z["bs%d/flat" % bs])
        for b in batches:                                   # every batch: equal lengths, at most bs samples
            assert len(set(lengths[i] for i in b)) == 1 and len(b) <= bs
        assert sorted(np.concatenate(batches).tolist()) == list(range(len(lengths)))   # each sample exactly once


def test_checkpoint_roundtrip_with_optimizer_state(tmp_path):
    from vagnmt_hip.checkpoint import load_checkpoint, save_checkpoint
    from vagnmt_hip.trainer import TrainStep
    V11, _ = models()
    torch.manual_seed(0)
    m = V11(50, 60, 96, 16, 16, 24, 20, 0.99, tied_emb=True)
    ts = TrainStep(m, None, None, use_graph=False)              # flat buffers work on any device
    ts.fp.m.copy_(torch.randn_like(ts.fp.m))
    ts.fp.v.copy_(torch.rand_like(ts.fp.v))
    ts.step_count.fill_(17)
    ts.set_lr(8e-5)
    path = str(tmp_path / "ck.pt")
    save_checkpoint(path, m, ts, extra={"epoch": 3})
    torch.manual_seed(1)
    m2 = V11(50, 60, 96, 16, 16, 24, 20, 0.99, tied_emb=True)
    ts2 = TrainStep(m2, None, None, use_graph=False)
    ck = load_checkpoint(path, m2, ts2)
    assert ck["extra"] == {"epoch": 3} and int(ts2.step_count) == 17 and ts2.lr == 8e-5
    for (n1, p1), (n2, p2) in zip(m.named_parameters(), m2.named_parameters()):
        assert n1 == n2 and torch.equal(p1, p2)
    for n, p in ts.fp.named:                          # (the 256-byte slot padding between parameters is not state)
        o, k = ts.fp.offsets[n], p.numel()
        o2 = ts2.fp.offsets[n]
        assert torch.equal(ts.fp.m[o:o + k], ts2.fp.m[o2:o2 + k]) and torch.equal(ts.fp.v[o:o + k], ts2.fp.v[o2:o2 + k])
    # a bare reference-named state_dict and a pickled module load as well
    torch.save({k: v for k, v in m.state_dict().items()}, str(tmp_path / "sd.pt"))
    m3 = V11(50, 60, 96, 16, 16, 24, 20, 0.99, tied_emb=True)
    load_checkpoint(str(tmp_path / "sd.pt"), m3)
    assert torch.equal(m3.decoder.attn.v, m.decoder.attn.v)
    # parameters are still views of the flat buffer after loading (the fused optimiser keeps working)
    assert m2.decoder.attn.v.data_ptr() == ts2.fp.flat[ts2.fp.offsets["decoder.attn.v"]:].data_ptr()


def test_grouped_launch_plan_fills_the_block_slots_and_orders_longest_first():
    """vag_gemm_group_plan (host only): the decoder's weight-gradient group of configs[1] -- five accumulating products over
    K = Tt*B = 2560 and one that overwrites a scratch -- used to be cut into 528 equal blocks, a second round of the 512 block
    slots for 16 of them.  The plan must (a) slice every product so that no slice is shorter than 256, (b) order products by
    slice length, longest first, (c) beat the one-common-split plan in the list-scheduling model the planner itself uses."""
    import ctypes as C
    import heapq
    from vagnmt_hip import _lib as L

    def plan(shapes):
        n = len(shapes)
        M = (C.c_int64 * n)(*[s[0] for s in shapes]); N = (C.c_int64 * n)(*[s[1] for s in shapes])
        K = (C.c_int64 * n)(*[s[2] for s in shapes]); acc = (C.c_int * n)(*[s[3] for s in shapes])
        split, order = (C.c_int * n)(), (C.c_int * n)()
        assert L.lib().vag_gemm_group_plan(n, M, N, K, acc, split, order) == 0
        return list(split), list(order)

    def makespan(shapes, split, order):
        slots = [0.0] * 512
        heapq.heapify(slots)
        for i in order:
            Mi, Ni, Ki, _ = shapes[i]
            steps = -(-(-(-Ki // 32)) // split[i])
            for _ in range(-(-Mi // 128) * -(-Ni // 128) * split[i]):
                heapq.heappush(slots, heapq.heappop(slots) + steps + 4.0)
        return max(slots)

    H, C2, E, R = 512, 1024, 256, 2560
    shapes = [(3 * H, H, R, 1), (C2, H, R, 1), (3 * H, C2, R, 0), (3 * H, H, R, 1), (3 * H, E, R, 1), (C2, C2, R, 1)]
    split, order = plan(shapes)
    assert sorted(order) == list(range(len(shapes)))
    for (Mi, Ni, Ki, _), s in zip(shapes, split):
        assert 1 <= s <= max(1, Ki // 256)
    lens = [-(-(-(-shapes[i][2] // 32)) // split[i]) for i in order]
    assert lens == sorted(lens, reverse=True)
    tiles = sum(-(-m // 128) * -(-n // 128) for m, n, _, _ in shapes)
    common = max(1, (512 + tiles // 2) // tiles)                       # the first version's rule: aim at ~512 blocks
    old = [common if a else 1 for (_, _, _, a) in shapes]
    assert makespan(shapes, split, order) <= makespan(shapes, old, list(range(len(shapes))))
    # heterogeneous K, nothing may be sliced (all overwrite, K < 512): the plan is only an order
    shapes2 = [(2560, 1024, 1024, 0), (2560, 1536, 256, 0), (2560, 1536, 1024, 0)]
    split2, order2 = plan(shapes2)
    assert split2[1] == 1 and shapes2[order2[-1]][2] == 256
    # bad arguments come back as -EINVAL
    assert L.lib().vag_gemm_group_plan(0, None, None, None, None, None, None) == -22


def test_flat_layout_with_explicit_groups_and_the_shims_optimizer_checks():
    """The ``train`` shim hands TrainStep the caller's torch.optim.Adam groups (nmt_multimodal_beam_DE.py:303-332): every group is
    split into its non-encoder / encoder part, its own weight decay travels as a number; the optimiser checks are host logic."""
    import importlib
    from machine_translation_vision.models import NMT_AttentionImagine_Seq2Seq_Beam_V11
    from vagnmt_hip.trainer import flat_layout
    m = NMT_AttentionImagine_Seq2Seq_Beam_V11(50, 60, 96, 16, 16, 24, 20, 0.99, tied_emb=True)
    named = [(n, p) for n, p in m.named_parameters()]
    w = [n for n, _ in named if "bias" not in n]
    b = [n for n, _ in named if "bias" in n]
    segs, offs, seg_off, total = flat_layout(named, groups=[("g0", w, 1e-5, 1.0), ("g1", b, 0.0, 1.0)])
    assert [s[0] for s in segs] == ["g0", "g1", "g0/encoder", "g1/encoder"]
    assert [s[2] for s in segs] == [1e-5, 0.0, 1e-5, 0.0]
    assert sorted(offs) == sorted(n for n, _ in named) and seg_off[-1] == total
    ref = flat_layout(named)                       # the by-name grouping gives the same offsets
    assert ref[1] == offs and ref[2] == seg_off
    T = importlib.import_module("train")
    def groups():                                  # fresh dicts: an optimiser writes its defaults into them
        return [{"params": [p for n, p in named if "bias" not in n], "weight_decay": 1e-5},
                {"params": [p for n, p in named if "bias" in n]}]
    opt = torch.optim.Adam(groups(), lr=4e-4)
    assert T._plain_adam(opt) and T._covers(m, opt)
    assert not T._plain_adam(torch.optim.Adam(groups(), lr=4e-4, amsgrad=True))
    assert not T._plain_adam(torch.optim.SGD(groups(), lr=0.1))
    assert not T._covers(m, torch.optim.Adam(groups()[:1], lr=4e-4))
    vw = torch.ones(60)
    vw[0] = 0
    # CPU tensors never reach the fused driver; the literal sequence then fails loudly in the HIP operators (no CPU fallback)
    d, existing = T._driver(m, opt, torch.nn.NLLLoss(weight=vw, reduction="none"), None, 1.0, 1.0)
    assert d is None and existing is None


def test_three_bucket_layout_cuts_after_the_decoder_and_after_the_visual_grounding():
    """TrainStep(three_buckets=True): [head + decoder + attn_e | vse_imagine.* + decoderini.* | encoder.*], each a contiguous range of the
    flat buffer in the order the backward pass finishes them (vag_train_step phases 1|16, 32, 4); the default layout is unchanged."""
    from machine_translation_vision.models import NMT_AttentionImagine_Seq2Seq_Beam_V11
    from vagnmt_hip.trainer import flat_layout, is_late, is_mid
    m = NMT_AttentionImagine_Seq2Seq_Beam_V11(50, 60, 96, 16, 16, 24, 20, 0.99, tied_emb=True)
    named = [(n, p) for n, p in m.named_parameters()]
    for vs in (False, True):
        segs, offs, seg_off, total = flat_layout(named, vse_separate=vs, three_buckets=True)
        kinds = [2 if s[0].endswith("/encoder") else 1 if s[0].endswith("/vse+init") else 0 for s in segs]
        assert kinds == sorted(kinds) and set(kinds) == {0, 1, 2}
        for (name, names, _, _), kind in zip(segs, kinds):
            for n in names:
                assert (2 if is_late(n) else 1 if is_mid(n) else 0) == kind, (name, n)
        assert sorted(offs) == sorted(n for n, _ in named) and seg_off[-1] == total
        two = flat_layout(named, vse_separate=vs)
        assert two[3] == total and all(not s[0].endswith("/vse+init") for s in two[0])


def test_whole_module_pickle_drops_decode_caches_and_gradient_routing():
    """torch.save(model) as the reference's trainer does it right after an evaluation pass (nmt_multimodal_beam_DE.py:491-520): what
    decoding cached on the module (captured graphs do not pickle) and the step driver's gradient routing on the parameters stay out."""
    V11, _ = models()
    m = V11(50, 60, 96, 16, 16, 24, 20, 0.99, tied_emb=True)

    class Unpicklable:
        def __reduce__(self):
            raise TypeError("a captured HIP graph does not pickle")
    m._decode_cache = {("beam", 16): {"graph": Unpicklable()}}
    m._decode_wcache = {1: {"prep": torch.zeros(3)}}
    from vagnmt_hip.trainer import FlatParams
    fp = FlatParams(m)                                   # (CPU tensors: the flat re-homing is host logic)
    p0 = next(m.parameters())
    assert hasattr(p0, "_vag_grad")
    buf = io.BytesIO()
    torch.save(m, buf)
    assert buf.tell() < 3 * fp.n * 4                     # the values once, not the gradient buffer as well
    buf.seek(0)
    m2 = torch.load(buf, weights_only=False)
    assert not hasattr(m2, "_decode_cache") and not hasattr(m2, "_decode_wcache")
    for (n1, p1), (n2, p2) in zip(m.named_parameters(), m2.named_parameters()):
        assert n1 == n2 and torch.equal(p1, p2) and not hasattr(p2, "_vag_grad")
    assert hasattr(m, "_decode_cache")                   # (the live module keeps its caches)


def test_sharded_optimizer_layout_is_equal_shards_of_padded_buffers():
    """TrainStep(zero1=True) (SURVEY 8e option for train.py:46-49): the four flat buffers are ALLOCATED to a multiple of world x 64
    floats so that reduce-scatter / all-gather cut them into equal, 256-byte aligned shards; the parameters' views and the segment
    table are those of the replicated layout (pure host logic: no GPU, no process group)."""
    import torch
    from machine_translation_vision.models import NMT_AttentionImagine_Seq2Seq_Beam_V11
    from vagnmt_hip.trainer import FlatParams, TrainStep
    torch.manual_seed(0)
    m0 = NMT_AttentionImagine_Seq2Seq_Beam_V11(50, 60, 96, 16, 16, 24, 20, 0.99, tied_emb=True)
    ref = FlatParams(m0)
    assert ref.n_alloc == ref.n
    for world in (2, 3, 8):
        torch.manual_seed(0)
        m = NMT_AttentionImagine_Seq2Seq_Beam_V11(50, 60, 96, 16, 16, 24, 20, 0.99, tied_emb=True)
        fp = FlatParams(m, pad_to=world * 64)
        assert fp.n == ref.n and fp.offsets == ref.offsets and fp.seg_off == ref.seg_off
        assert fp.n_alloc % (world * 64) == 0 and 0 <= fp.n_alloc - fp.n < world * 64
        assert fp.flat.numel() == fp.n and fp.flat_alloc.numel() == fp.n_alloc
        assert fp.flat.data_ptr() == fp.flat_alloc.data_ptr() and fp.grad.data_ptr() == fp.grad_alloc.data_ptr()
        assert float(fp.flat_alloc[fp.n:].abs().sum()) == 0.0
        assert torch.equal(fp.flat, ref.flat)                       # same values at the same offsets
        size = fp.n_alloc // world
        assert size % 64 == 0
        covered = 0
        for rank in range(world):                                   # what TrainStep._shard hands vag_clip_adam_shard
            lo = min(rank * size, fp.n)
            hi = min(lo + size, fp.n)
            assert lo % 4 == 0 and lo <= hi
            covered += hi - lo
        assert covered == fp.n
    ts = TrainStep(m0, None, None, use_graph=False, zero1=True, backend=object())
    assert ts.zero1 and ts._shard()[:2] == (0, ts.fp.n)
    import pytest
    with pytest.raises(ValueError):
        TrainStep(m0, None, None, zero1=True, comm=object(), backend=object())
