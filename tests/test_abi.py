"""The C ABI cannot drift silently: a C translation unit compiled with gcc against include/miseg_hip.h reports sizeof / offsetof of every
params struct and every field; they must equal the ctypes mirror in mi-seg_amd/hip/lib.py (which names the same fields, so a renamed or
missing field fails the compile), and the library's own miseg_abi_struct_size / miseg_abi_version must agree with both.  CPU only."""
import ctypes as C
import os
import re
import subprocess

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "miseg_hip.h")
DEBUG_HEADER = os.path.join(ROOT, "include", "miseg_hip_debug.h")      # measurement / experiment entry points: not the product ABI


def _lib():
    from mi_seg_amd.hip import lib
    return lib


def test_every_header_struct_has_a_ctypes_mirror():
    lib = _lib()
    declared = set(re.findall(r"\}\s*(miseg_\w+);", open(HEADER).read()))
    mirrored = {v for v in lib.C_NAMES.values() if v}
    assert declared == mirrored - lib.DEBUG_STRUCTS, (sorted(declared - mirrored), sorted(mirrored - declared))
    assert set(re.findall(r"\}\s*(miseg_\w+);", open(DEBUG_HEADER).read())) == lib.DEBUG_STRUCTS


def _prototypes(path):
    """names of the functions a header declares (a line that starts with a return type and `miseg_name(`)"""
    text = re.sub(r"/\*.*?\*/", "", open(path).read(), flags=re.S)
    return set(re.findall(r"^(?:int|size_t|void|const char\*)\s+(miseg_\w+)\s*\(", text, flags=re.M))


def test_every_prototype_is_bound_and_the_debug_entry_points_stay_out_of_the_product_header():
    """include/miseg_hip.h == hip/lib.py::PROTOS and include/miseg_hip_debug.h == DEBUG_PROTOS, name for name; the library exports all of
    them (load() binds every name of both tables)"""
    lib = _lib()
    assert _prototypes(HEADER) == set(lib.PROTOS), (sorted(_prototypes(HEADER) - set(lib.PROTOS)), sorted(set(lib.PROTOS) - _prototypes(HEADER)))
    assert _prototypes(DEBUG_HEADER) == set(lib.DEBUG_PROTOS), (sorted(_prototypes(DEBUG_HEADER) ^ set(lib.DEBUG_PROTOS)))
    assert not (set(lib.PROTOS) & set(lib.DEBUG_PROTOS))
    so = lib.load()
    assert so.miseg_prof_available() == 0                      # the product library is linked without --wrap=hipLaunchKernel ...
    assert so.miseg_prof_arm(3) == -2 and b"libmiseg_hip_prof.so" in so.miseg_last_error()
    assert so.miseg_prof_arm(-1) == 0
    assert os.path.exists(lib.PROF_LIB_PATH)                   # ... the measurement build beside it carries the hook
    with lib.profiling_library() as prof:
        assert prof.miseg_prof_available() == 1 and prof.miseg_source_digest() == so.miseg_source_digest()
    assert lib.load() is so


def test_a_library_older_than_its_sources_is_refused():
    """the .so carries the sha256 of the sources it was compiled from; load() compares it with the tree (csrc/build.py::source_digest)"""
    lib = _lib()
    so = lib.load()
    assert so.miseg_source_digest().decode() == lib.source_digest()
    import pytest
    keep = lib.source_digest
    try:
        lib.source_digest = lambda: "0" * 64
        with pytest.raises(lib.MisegHipError, match="other sources"):
            lib._open(lib.LIB_PATH)
    finally:
        lib.source_digest = keep


def test_struct_layouts_match_a_c_compile_of_the_header(tmp_path):
    lib = _lib()
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', f'#include "{DEBUG_HEADER}"', "int main(void) {"]
    for t, cname in lib.C_NAMES.items():
        lines.append(f'  printf("S {cname} %zu\\n", sizeof({cname}));')
        for fname, _ in t._fields_:
            lines.append(f'  printf("F {cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  printf("V %d\\n", MISEG_ABI_VERSION);', "  return 0;", "}"]
    src = tmp_path / "abi.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "abi"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-o", str(exe), str(src)], check=True, capture_output=True, text=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    by_name = {v: k for k, v in lib.C_NAMES.items()}
    seen = 0
    for ln in out:
        w = ln.split()
        if not w:
            continue
        if w[0] == "S":
            assert C.sizeof(by_name[w[1]]) == int(w[2]), f"sizeof({w[1]}): C {w[2]}, ctypes {C.sizeof(by_name[w[1]])}"
        elif w[0] == "F":
            assert getattr(by_name[w[1]], w[2]).offset == int(w[3]), f"offsetof({w[1]}, {w[2]}): C {w[3]}, ctypes {getattr(by_name[w[1]], w[2]).offset}"
            seen += 1
        elif w[0] == "V":
            assert int(w[1]) == lib.ABI_VERSION
    assert seen > 300


def test_library_reports_the_same_sizes_and_version():
    lib = _lib()
    so = lib.load()                       # load() itself refuses a version / size mismatch
    assert so.miseg_abi_version() == lib.ABI_VERSION
    for t, cname in lib.C_NAMES.items():
        assert so.miseg_abi_struct_size(cname.encode()) == C.sizeof(t), cname
    assert so.miseg_abi_struct_size(b"no_such_struct") == 0
    buf = C.create_string_buffer(16)
    assert so.miseg_device_arch(buf, 16) == 0 and buf.value == b"gfx950"


def test_struct_size_field_is_checked():
    """the structs added in ABI version 2 carry their own size: a caller built against another header gets BADARG, not a wild read"""
    lib = _lib()
    so = lib.load()
    p = lib.Stitch()
    p.struct_size = C.sizeof(lib.Stitch) - 8
    assert so.miseg_stitch_windows(C.byref(p), None) == -1
    assert b"struct_size" in so.miseg_last_error()


def _integration_stub():
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"```python\n(import ctypes as C, torch\n.*?)```", md, flags=re.S)
    assert m, "INTEGRATION.md lost its per-op binding stub"
    return m.group(1)


def test_integration_md_stub_matches_the_binding():
    """the stub a reference maintainer would copy out of INTEGRATION.md is executed as written (its own asserts compare its structs with
    the library's sizeof) and its field lists must equal the ones of hip/lib.py, which the gcc test above ties to the header"""
    lib = _lib()
    ns = {}
    cwd = os.getcwd()
    os.chdir(ROOT)                      # the stub opens the library by its in-tree relative path
    try:
        exec(compile(_integration_stub(), "INTEGRATION.md", "exec"), ns)
    finally:
        os.chdir(cwd)
    for mine, theirs in ((ns["Stats"], lib.InstnormStats), (ns["Apply"], lib.InstnormApply)):
        assert [f[0] for f in mine._fields_] == [f[0] for f in theirs._fields_]
        assert C.sizeof(mine) == C.sizeof(theirs)
        for f in mine._fields_:
            assert getattr(mine, f[0]).offset == getattr(theirs, f[0]).offset, f[0]
