"""ctypes binding of libvagnmt.so (include/vag_nmt.h).

The library is the product: there is NO CPU or pure-torch fallback.  If the shared object is missing, or a
tensor is not a contiguous fp32/int64 HIP tensor, the call raises."""
import ctypes as C
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(_HERE)                       # .../vag-nmt_amd
CSRC = os.path.join(PKG_ROOT, "csrc")
LIB_PATH = os.environ.get("VAG_LIB") or os.path.join(PKG_ROOT, "lib", "libvagnmt.so")      # VAG_LIB: A/B builds of the library

P = C.c_void_p
I64 = C.c_int64
I32 = C.c_int
F = C.c_float


class GruW(C.Structure):
    _fields_ = [("w_ih", P), ("w_hh", P), ("b_ih", P), ("b_hh", P)]


class DecW(C.Structure):
    _fields_ = [("emb", P), ("gru1", GruW), ("attn_h", P), ("attn_v", P), ("c2h", P), ("gru2", GruW)]


class HeadW(C.Structure):
    _fields_ = [("w1", P), ("b1", P), ("w2", P), ("b2", P), ("w3", P), ("b3", P), ("out_w", P), ("out_b", P)]


class ModelW(C.Structure):
    """vag_model_w / vag_model_g (identical layout: one pointer per parameter)."""
    _fields_ = [("enc_emb", P), ("enc_fw", GruW), ("enc_bw", GruW),
                ("im_w", P), ("im_b", P), ("txt_w", P), ("txt_b", P), ("ctx2ctx", P), ("emb2ctx", P), ("mlp_w", P),
                ("ini_w", P), ("ini_b", P), ("attn_e", P), ("dec", DecW), ("head", HeadW)]


class StepCfg(C.Structure):
    """vag_step_cfg"""
    _fields_ = [(n, I64) for n in ("B", "Ts", "Tt", "Es", "Et", "H", "S", "I", "V", "ldl")] + \
               [(n, I32) for n in ("multimodal", "attn_method", "activation_vse", "rank_kind", "free_run", "storage")] + \
               [(n, F) for n in ("margin", "loss_w", "init_split", "p_emb", "p_ctx", "p_out")] + [("loss_ring", I32), ("guard", P)]


# name -> (restype, argtypes); mirrors include/vag_nmt.h declaration by declaration
PROTOS = {
    "vag_version": (I32, []),
    "vag_gemm_f32": (I32, [I64, I64, I64, F, P, I64, I64, P, I64, I64, F, P, I64, P, I32, P]),
    "vag_linear_fwd": (I32, [I64, I64, I64, P, P, P, I32, P, P]),
    "vag_linear_bwd": (I32, [I64, I64, I64, P, P, P, P, I32, P, I32, P, P, P]),
    "vag_embed_fwd": (I32, [P, I64, P, I64, P, P]),
    "vag_embed_bwd": (I32, [P, I64, P, I64, P, P]),
    "vag_bigru_ws_floats": (I64, [I64, I64, I64, I64]),
    "vag_bigru_seq_fwd": (I32, [P, P, P, GruW, GruW, F, F, P, I64, I64, I64, I64, P, P, P, P]),
    "vag_bigru_seq_bwd": (I32, [P, P, GruW, GruW, F, F, P, I64, I64, I64, I64, P, P, P, GruW, GruW, P]),
    "vag_gru_cell_fwd": (I32, [P, P, P, P, I64, I64, P, P, P]),
    "vag_gru_cell_bwd": (I32, [P, P, P, P, P, P, I64, I64, P, P, P, P]),
    "vag_attn_keys_proj": (I32, [P, P, I64, I64, P, P]),
    "vag_bahdanau_attn_fwd": (I32, [P, P, P, P, P, I64, I64, I64, I64, P, P, P, P]),
    "vag_attn_keys_proj_bwd": (I32, [P, P, P, I64, I64, P, I32, P, P]),
    "vag_cgru_ws_floats": (I64, [I64, I64, I64, I64, I64]),
    "vag_cgru_attn_decode_seq_fwd": (I32, [P, P, P, P, P, DecW, I64, I64, I64, I64, I64, I64, P, P, P, P, I32,
                                           C.POINTER(HeadW), F, P, P, P, I64, P]),
    "vag_cgru_free_supported": (I32, [I64, I64, I64, I64, I64, I64]),
    "vag_cgru_free_tables_floats": (I64, [I64, I64, I64, I64, I64, I64]),
    "vag_cgru_attn_decode_free_fwd": (I32, [P, P, P, P, P, DecW, I64, I64, I64, I64, I64, I64, P, P, P, P,
                                            C.POINTER(HeadW), F, P, P, P, I64, P, P]),
    "vag_cgru_bwd_scratch_floats": (I64, [I64, I64, I64, I64, I64]),
    "vag_cgru_attn_decode_seq_bwd": (I32, [P, P, P, P, P, DecW, I64, I64, I64, I64, I64, I64, P, P, P, P, P, P, P, P,
                                           I32, P, P, DecW, P, P]),
    "vag_cgru_attn_decode_seq_bwd_loop": (I32, [P, P, P, P, P, DecW, I64, I64, I64, I64, I64, I64, P, P, P, P, P, P, P, P,
                                                I32, P, P, P, P]),
    "vag_cgru_attn_decode_seq_bwd_weights": (I32, [P, P, DecW, I64, I64, I64, I64, I64, P, P, P, P, P, DecW, P, P]),
    "vag_cgru_step_scratch_floats": (I64, [I64, I64, I64, I64]),
    "vag_cgru_prep_floats": (I64, [I64]),
    "vag_cgru_prepare": (I32, [DecW, I64, P, P]),
    "vag_cgru_attn_decode_step": (I32, [P, P, P, I64, P, P, DecW, P, I64, I64, I64, I64, P, P, P, P, P, P]),
    "vag_cgru_decode_keys_floats": (I64, [I64, I64, I64, I64]),
    "vag_cgru_decode_keys": (I32, [P, P, P, I64, I64, I64, I64, P, P]),
    "vag_cgru_decode_tables_floats": (I64, [I64, I64, I64]),
    "vag_cgru_decode_tables": (I32, [DecW, P, I64, I64, I64, P, P]),
    "vag_cgru_attn_decode_step_h": (I32, [P, P, P, P, I64, I64, P, P, DecW, P, I64, I64, I64, I64, P, P, P, P, P, P]),
    "vag_head_logp_step_h": (I32, [P, P, P, P, P, HeadW, I64, I64, I64, I64, P, I64, P, P, P]),
    "vag_head_logits_step_h": (I32, [P, P, P, P, P, HeadW, I64, I64, I64, I64, P, I64, P, P, P]),
    "vag_head_ce_seq_fwd": (I32, [P, P, P, HeadW, P, P, I64, I64, I64, I64, I64, F, P, I32, P, P, I64, P, P, P, P, P]),
    "vag_head_ce_seq_bwd": (I32, [P, P, P, HeadW, P, P, I64, I64, I64, I64, I64, F, P, P, P, I64, P, P, P, P, P, P,
                                  HeadW, P, P]),
    "vag_head_ce_seq_bwd_data": (I32, [HeadW, P, P, I64, I64, I64, I64, I64, F, P, P, P, I64, P, P, P, P, P, P, P, P]),
    "vag_head_bwd_weights": (I32, [P, P, P, I64, I64, I64, I64, P, P, I64, P, HeadW, P]),
    "vag_head_logp_seq_fwd": (I32, [P, P, P, HeadW, I64, I64, I64, I64, F, P, P, P, I64, P]),
    "vag_head_logp_seq_bwd": (I32, [P, P, P, HeadW, I64, I64, I64, I64, F, P, P, P, P, I64, P, P, P, HeadW, P, P]),
    "vag_l2norm_fwd": (I32, [P, I64, I64, P, P, P]),
    "vag_l2norm_bwd": (I32, [P, P, P, P, I64, I64, P, P]),
    "vag_head_logp_step": (I32, [P, P, P, HeadW, I64, I64, I64, I64, P, I64, P, P, P]),
    "vag_img_proj_l2_fwd": (I32, [P, P, P, I64, I64, I64, I32, P, P, P, P]),
    "vag_img_proj_l2_bwd": (I32, [P, P, P, P, P, P, I64, I64, I64, I32, P, P, P, P]),
    "vag_imagine_ws_floats": (I64, [I64, I64, I64, I64, I32]),
    "vag_imagine_attn_ctx_fwd": (I32, [P, P, P, P, P, P, I32, I64, I64, I64, I64, P, P, P, P]),
    "vag_imagine_attn_ctx_bwd": (I32, [P, P, P, P, P, P, I32, I64, I64, I64, I64, P, P, P, P, I32, P, P, P, P, P]),
    "vag_rank_loss_fwd": (I32, [P, P, I64, I64, F, I32, P, P, P, P]),
    "vag_rank_loss_bwd": (I32, [P, P, P, P, I64, I64, P, P, P]),
    "vag_gather_rows_i64": (I32, [P, I64, P, I64, I64, P, P]),
    "vag_retrieval_ranks": (I32, [P, P, I64, I64, P, P, P]),
    "vag_dec_init_fwd": (I32, [P, P, P, F, P, P, I64, I64, I64, I64, P, P, P]),
    "vag_dec_init_bwd": (I32, [P, P, P, F, P, P, I64, I64, I64, I64, P, I32, P, P, P, P, P]),
    "vag_beam_scratch_bytes": (I64, [I64, I64, I64, I64]),
    "vag_beam_step": (I32, [P, I64, P, P, I64, I64, P, P, I64, I64, I64, I64, P, P, P]),
    "vag_beam_step_dev": (I32, [P, I64, P, P, P, I64, P, P, P, I64, I64, I64, I64, P, P, P]),
    "vag_head_logits_parts_count": (I64, [HeadW, I64, I64, I64]),
    "vag_head_logits_step": (I32, [P, P, P, HeadW, I64, I64, I64, I64, P, I64, P, P, P]),
    "vag_beam_step_logits_dev": (I32, [P, I64, P, I64, P, P, P, I64, P, P, P, I64, I64, I64, I64, P, P, P]),
    "vag_beam_finish": (I32, [P, P, I64, I64, I64, I64, P, P, P]),
    "vag_clip_adam_flat": (I32, [P, P, P, P, I64, I32, C.POINTER(I64), C.POINTER(F), C.POINTER(F), F, F, F, F, F, I32, P,
                                 P, P, P, P]),
    "vag_clip_adam_shard": (I32, [P, P, P, P, I64, I32, C.POINTER(I64), C.POINTER(F), C.POINTER(F), F, F, F, F, F, I32, P,
                                  P, P, P, I64, I64, I32, P, P]),
    "vag_step_ws_floats": (I64, [C.POINTER(StepCfg)]),
    "vag_step_ws_offset": (I64, [C.POINTER(StepCfg), I32]),
    "vag_train_step": (I32, [C.POINTER(StepCfg), C.POINTER(ModelW), C.POINTER(ModelW), P, P, P, P, P, P, P, P, P, I32, P]),
    "vag_copy4": (I32, [C.POINTER(P), C.POINTER(P), C.POINTER(I64), I32, P]),
    "vag_set_operator_context": (I32, [P, I32]),
    "vag_set_option": (I32, [C.c_char_p, I64]),
    "vag_recurrence_supported": (I32, [I32, I64, I64, I64, I64]),
    "vag_persistent_timeouts": (I32, []),
    "vag_set_operator_guard": (I32, [P]),
    "vag_recurrence_time": (I32, [I32, P, P]),
    "vag_gemm_group_plan": (I32, [I32, P, P, P, P, P, P]),
    "vag_comm_unique_id": (I32, [P]),
    "vag_comm_init": (I32, [P, I32, I32, P]),
    "vag_comm_allreduce": (I32, [P, P, I64, P]),
    "vag_comm_size": (I32, [P]),
    "vag_comm_destroy": (I32, [P]),
    "vag_recurrence_sync_words": (I64, [I32, I64, I64]),
    "vag_cgru_recurrence_fwd": (I32, [P, P, P, P, DecW, P, P, P, I64, I64, I64, I64, P, P, P, P, P, P, P, P, P]),
    "vag_derived_floats": (I64, [I64]),
    "vag_derive_weights": (I32, [DecW, P, P, I64, I32, P, P]),
    "vag_cgru_ws_offset": (I64, [I64, I64, I64, I64, I64, I32]),
    "vag_dropout_mask": (I32, [P, I32, I64, F, P, P]),
    "vag_rng_advance": (I32, [P, P]),
}

_lib = None


def use_lab_build():
    """Measurement tools only (tools/exp_dec_phases.py, list_gemms.py): build and load ../lib/libvagnmt_lab.so (`make LAB=1`:
    phase timestamps inside the persistent decoder kernels, the "gemm_debug" / "dec_stamps" / "dec_bwd_stamps" options)
    instead of the product library.  Call before the first lib()."""
    global LIB_PATH
    assert _lib is None, "use_lab_build() must come before the library is loaded"
    r = subprocess.run(["make", "-C", CSRC, "-j8", "LAB=1"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building libvagnmt_lab.so failed:\n" + r.stderr[-2000:])
    LIB_PATH = os.path.join(os.path.dirname(LIB_PATH), "libvagnmt_lab.so")
    return LIB_PATH


def build(verbose=False):
    """Compile libvagnmt.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-C", CSRC, "-j8"], capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout[-4000:])
        print(r.stderr[-4000:])
    if r.returncode != 0:
        raise RuntimeError("building libvagnmt.so failed")
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libvagnmt.so not found at %s: build it with `make -C %s` (or __graft_entry__.build()). "
                "There is no CPU fallback for the VAG-NMT hot path." % (LIB_PATH, CSRC))
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOS.items():
            fn = getattr(L, name)          # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class VagError(RuntimeError):
    pass


def check(rc, what=""):
    if rc != 0:
        kind = "argument/shape error" if rc < 0 else "hipError_t"
        raise VagError("libvagnmt %s failed: %s %d" % (what, kind, rc))


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class capture:
    """``with capture(graph):`` -- stream capture into a torch.cuda.CUDAGraph on a side stream, errors confined to the capturing
    thread, Python's garbage collector held off for the duration.

    Not ``torch.cuda.graph(graph)``: that context manager runs ``torch.cuda.synchronize(); gc.collect(); torch.cuda.empty_cache()``
    on entry -- ~25 ms per capture here, and every cached block of the allocator handed back to the driver, so the steps after a
    capture pay ``hipMalloc`` again.  A bucketed batch stream meets a new (B, Ts, Tt) shape every few steps: with those captures an
    epoch-shaped run was 4-5x SLOWER than eager launches (profiles/r05_exp_stream.txt).  The training step's captured regions
    allocate nothing (static workspaces), so none of the three is needed there.  The DECODE graphs (models/_seq2seq.py) do allocate
    inside their captures -- the per-step outputs of ``ops.decode_step_h`` / ``head_logits_step`` -- which works because a capture
    opens a private memory pool; pass ``pool=`` (one ``torch.cuda.graph_pool_handle()`` per model) so that all decode shapes of a
    model draw from ONE pool whose blocks are reused from capture to capture instead of one pool per captured shape.
    The collector stays off because a collection that runs INSIDE a capture can free tensors whose storage the caching allocator
    must first fence on another stream (anything that went through ``record_stream``, e.g. gradient buckets handed to an exchange
    stream): that event record on a non-capturing stream aborts the process (found as a silent abort ~100 tests after a
    data-parallel test, always inside a decode-graph capture)."""
    _streams = {}

    def __init__(self, graph, pool=None):
        self._graph = graph
        self._pool = pool
        self._gc = False
        self._ctx = None

    def __enter__(self):
        import gc
        dev = torch.cuda.current_device()
        side = capture._streams.get(dev)
        if side is None:
            side = capture._streams[dev] = torch.cuda.Stream(device=dev)
        self._gc = gc.isenabled()
        if self._gc:
            gc.disable()
        try:
            side.wait_stream(torch.cuda.current_stream())
            self._ctx = torch.cuda.stream(side)
            self._ctx.__enter__()
            try:
                if self._pool is not None:
                    self._graph.capture_begin(pool=self._pool, capture_error_mode="thread_local")
                else:
                    self._graph.capture_begin(capture_error_mode="thread_local")
            except BaseException:
                self._ctx.__exit__(None, None, None)
                raise
        except BaseException:
            if self._gc:          # the capture never began (already capturing, allocator error): __exit__ will not run
                gc.enable()
            raise
        return self

    def __exit__(self, *exc):
        import gc
        try:
            self._graph.capture_end()
        finally:
            try:
                self._ctx.__exit__(*exc)
                torch.cuda.current_stream().wait_stream(capture._streams[torch.cuda.current_device()])
            finally:
                if self._gc:
                    gc.enable()
        return False


def ptr(t, dtype=torch.float32):
    """Device pointer of a contiguous HIP tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise VagError("the VAG-NMT HIP path needs tensors on the GPU (got a %s tensor); there is no CPU fallback"
                       % t.device)
    if t.dtype != dtype:
        raise VagError("expected %s, got %s" % (dtype, t.dtype))
    if not t.is_contiguous():
        raise VagError("expected a contiguous tensor")
    return t.data_ptr()


def gru_w(w_ih, w_hh, b_ih, b_hh):
    return GruW(ptr(w_ih), ptr(w_hh), ptr(b_ih), ptr(b_hh))


def call(name, *args):
    check(getattr(lib(), name)(*args), name)


def set_option(name, value):
    """Debug / tuning option of the library by name (include/vag_nmt.h: vag_set_option)."""
    check(lib().vag_set_option(name.encode(), int(value)), "vag_set_option(%s)" % name)
