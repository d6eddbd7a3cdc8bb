"""CPU: the C-ABI library loads and exports every symbol include/vag_nmt.h declares (no compute calls)."""
import os
import re

from conftest import ROOT


def header_functions():
    src = open(os.path.join(ROOT, "include", "vag_nmt.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vag_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from vagnmt_hip import _lib
    assert os.path.exists(_lib.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    L = _lib.lib()
    names = header_functions()
    assert len(names) >= 35
    for n in names:
        assert hasattr(L, n), "libvagnmt.so does not export " + n
    assert L.vag_version() >= 100


def test_binding_table_matches_header():
    from vagnmt_hip import _lib
    assert sorted(_lib.PROTOS) == header_functions()


def test_workspace_size_queries_are_pure_host_calls():
    from vagnmt_hip import _lib
    L = _lib.lib()
    assert L.vag_bigru_ws_floats(64, 40, 256, 512) > 64 * 40 * 2 * 512
    assert L.vag_cgru_ws_floats(64, 40, 40, 256, 512) > 0
    assert L.vag_cgru_bwd_scratch_floats(64, 40, 40, 256, 512) > 0
    assert L.vag_beam_scratch_bytes(16, 12, 9391, 80) > 0
    assert L.vag_imagine_ws_floats(64, 40, 1024, 512, 1) > L.vag_imagine_ws_floats(64, 40, 1024, 512, 0)
    # round 4: scratch of the one-launch free-running decoder and of the hoisted decoding step
    r = lambda n: (n + 63) // 64 * 64      # noqa: E731
    assert L.vag_cgru_free_tables_floats(64, 40, 40, 256, 512, 9391) == \
        r(9391 * 1536) + r(9391 * 256) + r(64 * 40 * 256) + r(2 * 40 * 64 * 64)
    assert L.vag_cgru_decode_keys_floats(16, 40, 256, 512) == r(16 * 40 * 1536) + r(16 * 40 * 256)
    assert L.vag_cgru_decode_tables_floats(9391, 256, 512) == r(9391 * 1536) + r(9391 * 256)
    # the step workspace holds the free-running tables for shapes the one-launch form can take, and only for those
    c = _lib.StepCfg()
    c.B, c.Ts, c.Tt, c.Es, c.Et, c.H, c.S, c.I, c.V, c.ldl = 64, 40, 40, 256, 256, 512, 512, 2048, 9391, 9392
    c.multimodal, c.rank_kind = 1, 0
    import ctypes as C
    a = L.vag_step_ws_floats(C.byref(c))
    c.B = 65
    b = L.vag_step_ws_floats(C.byref(c))
    c.B, c.loss_ring = 64, 256
    assert L.vag_step_ws_floats(C.byref(c)) == a          # (the result ring lives in the caller's `losses`, not in the workspace)
    assert a > 0 and b > 0 and a - b * 64 // 65 > 9391 * 1536


def test_argument_errors_are_negative_codes_not_crashes():
    from vagnmt_hip import _lib
    L = _lib.lib()
    # NULL pointers / bad shapes must come back as -EINVAL before anything touches a device
    assert L.vag_linear_fwd(4, 4, 4, None, None, None, 0, None, None) == -22
    assert L.vag_bigru_seq_fwd(None, None, None, _lib.GruW(), _lib.GruW(), 0.0, 0.0, None, 1, 1, 4, 4, None, None, None,
                               None) == -22
    assert L.vag_rank_loss_fwd(None, None, 4, 4, 0.1, 0, None, None, None, None) == -22
    # round 4 entry points
    assert L.vag_cgru_attn_decode_free_fwd(None, None, None, None, None, _lib.DecW(), 64, 40, 40, 256, 512, 9391, None, None, None,
                                           None, None, 0.0, None, None, None, 0, None, None) == -22
    assert L.vag_cgru_decode_keys(None, None, None, 16, 40, 256, 512, None, None) == -22
    assert L.vag_cgru_decode_tables(_lib.DecW(), None, 9391, 256, 512, None, None) == -22
    assert L.vag_cgru_attn_decode_step_h(None, None, None, None, 0, 1, None, None, _lib.DecW(), None, 16, 40, 256, 512, None, None,
                                         None, None, None, None) == -22
    assert L.vag_head_logits_step(None, None, None, _lib.HeadW(), 192, 256, 512, 9391, None, 9392, None, None, None) == -22
    assert L.vag_beam_step_logits_dev(None, 9392, None, 147, None, None, None, 80, None, None, None, 16, 12, 9391, 512, None, None,
                                      None) == -22
    assert L.vag_head_logits_parts_count(_lib.HeadW(), 192, 256, 9391) == 0        # no weights: nothing to take
    assert L.vag_cgru_free_supported(64, 40, 40, 256, 512, 9391) in (0, 1)          # (0 without a device: the CU count decides)


def test_host_side_under_sanitizers():
    """`make -f asan.mk` (csrc/asan.mk; a recipe of its own, kept off the GPU boxes by .gpurunignore): the host pass of every translation unit under AddressSanitizer + UBSan, the device code untouched
    (host-side sanitizer build; GPU sanitizers do not exist on this pool).  The ABI and host-logic tests of this
    directory run against that library in a child process with the sanitizer runtime preloaded: the argument checks, workspace layout
    arithmetic, group planner and the thread-local request state are what it watches.  Any report makes the child exit non-zero."""
    import subprocess
    import sys
    import pytest
    csrc = os.path.join(ROOT, "vag-nmt_amd", "csrc")
    if not os.path.isfile(os.path.join(csrc, "asan.mk")):
        pytest.skip("csrc/asan.mk does not travel to GPU boxes (.gpurunignore): sanitizer builds run on the CPU box only")
    r = subprocess.run(["make", "-C", csrc, "-f", "asan.mk", "asan", "-j8"], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    so = os.path.join(ROOT, "vag-nmt_amd", "lib", "libvagnmt_asan.so")
    rt = subprocess.run(["/opt/rocm/lib/llvm/bin/clang", "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True,
                        text=True).stdout.strip()
    assert os.path.isfile(so) and os.path.isfile(rt), (so, rt)
    env = dict(os.environ, VAG_LIB=so, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    code = "\n".join([
        "import sys, ctypes as C",
        "sys.path[:0] = [%r, %r]" % (os.path.join(ROOT, "tests"), os.path.join(ROOT, "vag-nmt_amd")),
        "import test_abi as T",
        "from vagnmt_hip import _lib",
        "assert _lib.LIB_PATH.endswith('libvagnmt_asan.so')",
        "T.test_library_exports_every_declared_symbol()",
        "T.test_workspace_size_queries_are_pure_host_calls()",
        "T.test_argument_errors_are_negative_codes_not_crashes()",
        "L = _lib.lib()",
        "# the grouped-product planner (host code with its own scratch vectors) over many random groups",
        "import random",
        "random.seed(0)",
        "for _ in range(300):",
        "    n = random.randint(1, 12)",
        "    M = (C.c_int64 * n)(*[random.choice([64, 512, 1536, 2560, 9391]) for _ in range(n)])",
        "    N = (C.c_int64 * n)(*[random.choice([64, 256, 512, 1024, 3072]) for _ in range(n)])",
        "    K = (C.c_int64 * n)(*[random.choice([64, 256, 2560, 5120]) for _ in range(n)])",
        "    acc = (C.c_int * n)(*[random.randint(0, 1) for _ in range(n)])",
        "    split, order = (C.c_int * n)(), (C.c_int * n)()",
        "    assert L.vag_gemm_group_plan(n, M, N, K, acc, split, order) == 0",
        "    assert sorted(order) == list(range(n)) and all(s >= 1 for s in split)",
        "for name in ('persistent', 'leaf_queue', 'head_chunk', 'persist_spin_limit', 'persistent_enc_bwd'):",
        "    assert L.vag_set_option(name.encode(), 1) == 0",
        "assert L.vag_set_option(b'no_such_option', 1) == -22",
        "L.vag_set_option(b'head_chunk', -1); L.vag_set_option(b'persist_spin_limit', 0)",
        "print('SANITIZED-OK')"])
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SANITIZED-OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr, r.stderr[-3000:]
