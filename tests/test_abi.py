# -*- coding: utf-8 -*-
"""The C-ABI library loads on a machine without a GPU and exports exactly what include/trs.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    txt = open(os.path.join(ROOT, "include", "trs.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(trs_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_functions():
    fns = header_functions()
    assert "trs_score_fwd_bwd" in fns and "trs_topk" in fns and len(fns) >= 15


def test_library_exports_every_declared_symbol():
    from torchrecsys_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in header_functions():
        assert hasattr(lib, name), f"{name} declared in include/trs.h but not exported"


def test_binding_matches_header():
    from torchrecsys_amd import _lib
    assert sorted(_lib.PROTOTYPES) == header_functions()
    lib = _lib.load()
    assert lib.trs_abi_version() == _lib.ABI_VERSION
    assert lib.trs_topk_workspace_bytes(100_000, 10) > 0  # host-only helper: callable without a GPU


def test_struct_layout_matches_header():
    """ctypes mirrors of trs_tables / trs_batch have the C layout (x86-64 SysV: natural alignment)."""
    from torchrecsys_amd import _lib
    assert ctypes.sizeof(_lib.TrsTables) == 4 * 8 + 8 * 8 + 8 * 8 + 2 * 8 + 8 * 8 + 2 * 4
    assert ctypes.sizeof(_lib.TrsBatch) == 5 * 8 + 8 + 4 + 4 + 8
    assert _lib.TrsTables.D.offset == 4 * 8 + 8 * 8 + 8 * 8 + 2 * 8 + 8 * 8
    assert _lib.TrsBatch.err_flag_dev.offset == 56


def test_bad_arguments_return_error_codes_without_a_gpu():
    """Argument validation happens on the host before any launch."""
    from torchrecsys_amd import _lib
    lib = _lib.load()
    T, B = _lib.TrsTables(), _lib.TrsBatch()
    assert lib.trs_score_forward(1, ctypes.byref(T), ctypes.byref(B), None, None, None) == -1
    assert b"NULL" in lib.trs_last_error() or b"table" in lib.trs_last_error()
    assert lib.trs_topk(None, 0, 1, None, None, 0, None) == -1
    assert lib.trs_sample_neg(None, 3, 10, 5, 0, 0, None, None) == -1
    with pytest.raises(_lib.TrsError):
        _lib.check(-1, "x")


def test_tuning_knobs_are_set_through_the_abi_not_the_environment():
    """trs_tuning_set (include/trs.h): every knob of csrc/trs_common.h TrsTuning is settable and resettable by name; an
    unknown name is an error; no source file of the library calls getenv outside the one-time read in api.cpp."""
    from torchrecsys_amd import _lib
    lib = _lib.load()
    src = open(os.path.join(ROOT, "torchrecsys_amd", "csrc", "trs_common.h")).read()
    body = src[src.index("struct TrsTuning {"):src.index("TrsTuning& trs_tuning();")]
    import re
    names = re.findall(r"// TRS_([A-Z0-9_]+) ", body)
    assert len(names) >= 10
    for n in names:
        assert lib.trs_tuning_set(n.encode(), 1, 0) == 0, n
        assert lib.trs_tuning_set(n.encode(), 0, 1) == 0, n
    assert lib.trs_tuning_set(b"NO_SUCH_KNOB", 1, 0) < 0 and b"unknown knob" in lib.trs_last_error()
    with _lib.tuning(GEMM16_TILE=256):
        pass
    csrc = os.path.join(ROOT, "torchrecsys_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".h", ".cpp")) and f != "api.cpp":
            assert "getenv(" not in open(os.path.join(csrc, f)).read(), f
    assert open(os.path.join(csrc, "api.cpp")).read().count("getenv(") == 1


def _header_struct_fields(name):
    """Member names of `typedef struct NAME { ... } NAME;` in include/trs.h, in declaration order."""
    import re
    src = open(os.path.join(ROOT, "include", "trs.h")).read()
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), src, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    out = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        # "float *a, *b"  /  "const void* keys[TRS_MAX_META]"  /  "int32_t x"
        first, *rest = decl.split(",")
        out.append(re.sub(r"\[.*\]", "", first.split()[-1]).lstrip("*"))
        out += [re.sub(r"\[.*\]", "", r_.strip()).lstrip("*").strip() for r_ in rest]
    return out


def test_ctypes_structs_follow_the_header_member_by_member():
    """A ctypes mirror with a member missing or out of order would shift every device pointer after it by one slot."""
    import re
    from torchrecsys_amd import _lib
    for cname, mirror in (("trs_train_args", _lib.TrsTrainArgs), ("trs_opt", _lib.TrsOpt),
                          ("trs_meta_stage", _lib.TrsMetaStage), ("trs_tables", _lib.TrsTables),
                          ("trs_batch", _lib.TrsBatch), ("trs_sampler", _lib.TrsSampler)):
        assert _header_struct_fields(cname) == [f[0] for f in mirror._fields_], cname
    src = open(os.path.join(ROOT, "include", "trs.h")).read()
    assert int(re.search(r"#define TRS_ABI_VERSION (\d+)", src).group(1)) == _lib.ABI_VERSION
    assert _lib.load().trs_train_steps_sgd(None, None) == -1  # validated on the host, no launch
