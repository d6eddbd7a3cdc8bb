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


def test_argument_errors_are_negative_codes_not_crashes():
    from vagnmt_hip import _lib
    L = _lib.lib()
    # NULL pointers / bad shapes must come back as -EINVAL before anything touches a device
    assert L.vag_linear_fwd(4, 4, 4, None, None, None, 0, None, None) == -22
    assert L.vag_bigru_seq_fwd(None, None, None, _lib.GruW(), _lib.GruW(), 0.0, 0.0, None, 1, 1, 4, 4, None, None, None,
                               None) == -22
    assert L.vag_rank_loss_fwd(None, None, 4, 4, 0.1, 0, None, None, None, None) == -22
