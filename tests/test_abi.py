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


def test_one_launch_step_kernels_keep_four_waves_per_simd(tmp_path):
    """Register guard on the shipped code objects (no GPU, no recompilation): the one-launch step kernels of the c4 /
    c2 row shapes (FM, 16-byte lanes, whole rows: 32 or 16 lanes per triple) need at most 128 VGPRs — four waves per
    SIMD; at 133-135 the same kernel ran 2 us per step slower (profiles/EXPERIMENTS.md) — and no scratch."""
    import re
    import shutil
    import subprocess
    llvm = "/opt/rocm/lib/llvm/bin"
    tools = [os.path.join(llvm, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")]
    if not all(os.path.exists(t) for t in tools):
        pytest.skip("ROCm LLVM tools not found")
    from torchrecsys_amd import _lib
    fat = tmp_path / "fat.bin"
    subprocess.run([tools[0], "-O", "binary", "--only-section=.hip_fatbin", _lib.LIB_PATH, str(fat)], check=True)
    data = fat.read_bytes()
    offs = [m.start() for m in re.finditer(re.escape(b"__CLANG_OFFLOAD_BUNDLE__"), data)]
    assert offs, "no offload bundle in libtrs_hip.so"
    found = {}
    for k, o in enumerate(offs):  # one bundle per translation unit
        part, co = tmp_path / "b.bin", tmp_path / "co.o"
        part.write_bytes(data[o:offs[k + 1] if k + 1 < len(offs) else len(data)])
        r = subprocess.run([tools[1], "--type=o", "--unbundle", f"--input={part}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], capture_output=True)
        if r.returncode:
            continue
        notes = subprocess.run([tools[2], "--notes", str(co)], capture_output=True, text=True).stdout
        for blk in notes.split("- .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk)
            vg = re.search(r"\.vgpr_count:\s+(\d+)", blk)
            sc = re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk)
            if name and vg and sc:
                found[name.group(1)] = (int(vg.group(1)), int(sc.group(1)))
    shutil.rmtree(tmp_path, ignore_errors=True)
    hot = {n: v for n, v in found.items()
           if re.search(r"fwd_stage_kernelILi1ELi4ELi(32|16)ELi1ELi0ELb1ELi3ELi0ELi[013]E", n)}
    assert len(hot) == 6, sorted(hot)
    for n, (vgpr, scratch) in hot.items():
        assert vgpr <= 128 and scratch == 0, (n, vgpr, scratch)
