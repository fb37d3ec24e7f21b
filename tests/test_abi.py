"""CPU suite, part 2: the C-ABI library builds, loads and exports exactly what include/saccot.h declares —
no compute call is made (there is no GPU here), and without a GPU the library must fail loudly."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    # the drop-in surface (saccot.h) and the test / tuning hooks (saccot_debug.h)
    text = open(os.path.join(ROOT, "include", "saccot.h")).read() + open(os.path.join(ROOT, "include", "saccot_debug.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sc_[a-z_]+)\s*\(", text)))


def test_header_exports_present(pkg):
    names = _declared()
    assert len(names) >= 16 and "sc_register" in names and "sc_hypothesize_device" in names
    L = pkg.load_library()
    for n in names:
        assert hasattr(L, n), f"libsaccot.so does not export {n}"
    assert sorted(pkg.api.EXPORTS) == names


def test_header_is_plain_c():
    src = '#include "saccot.h"\nint main(void){ sc_params p; sc_default_params(&p); return (int)p.size; }\n'
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"),
                        "-x", "c", "-"], input=src.encode(), capture_output=True)
    assert r.returncode == 0, r.stderr.decode()


def test_struct_layouts_match(pkg):
    L = pkg.load_library()
    p = pkg.ScParams()
    L.sc_default_params(C.byref(p))
    assert p.size == C.sizeof(pkg.ScParams) == 64
    assert (p.t_cmp, p.max_triangles, p.shard_world, p.shard_block) == (pytest.approx(0.9), 50000, 1, 1024)
    # sc_stats: compile a probe against the header and compare sizeof
    exe = os.path.join(ROOT, "tests", ".abi_probe")
    src = '#include <stdio.h>\n#include "saccot.h"\nint main(void){printf("%zu %zu", sizeof(sc_params), sizeof(sc_stats));return 0;}\n'
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-x", "c", "-", "-o", exe], input=src.encode(), check=True)
    try:
        a, b = subprocess.check_output([exe]).decode().split()
    finally:
        os.remove(exe)
    assert int(a) == C.sizeof(pkg.ScParams) and int(b) == C.sizeof(pkg.ScStats)


def test_product_reads_no_environment(pkg):
    """SURVEY §5 "config / flags: one POD sc_params; no env vars": no getenv in the product sources, none imported by
    the shared library; the knobs the tests drive go through sc_set_debug, whose struct layout is checked here."""
    pk = os.path.join(ROOT, "sac-cot_amd", "csrc")
    for f in os.listdir(pk):
        if f.endswith((".hip", ".hpp", ".cpp", ".h")):
            assert "getenv" not in open(os.path.join(pk, f)).read(), f
    und = subprocess.check_output(["nm", "-D", "--undefined-only", pkg.api.LIB_PATH]).decode()
    assert "getenv" not in und
    exe = os.path.join(ROOT, "tests", ".abi_probe_dbg")
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "saccot_debug.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu", sizeof(sc_debug), offsetof(sc_debug, sample_edges), offsetof(sc_debug, compat_rows), sizeof(sc_debug_info), offsetof(sc_debug_info, filter_recounts), offsetof(sc_debug, lanes_per_edge), offsetof(sc_debug, gram_guard_fail), offsetof(sc_debug_info, n_frames));return 0;}\n'
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-x", "c", "-", "-o", exe], input=src.encode(), check=True)
    try:
        a, b, c, d, e, f, g, h = (int(x) for x in subprocess.check_output([exe]).decode().split())
    finally:
        os.remove(exe)
    D, I = pkg.api.ScDebug, pkg.api.ScDebugInfo
    assert (a, b, c) == (C.sizeof(D), D.sample_edges.offset, D.compat_rows.offset)
    assert (d, e) == (C.sizeof(I), I.filter_recounts.offset)
    assert (f, g, h) == (D.lanes_per_edge.offset, D.gram_guard_fail.offset, I.n_frames.offset)
    assert len(D._fields_) <= 30   # (the lab's knobs left the product's struct)


def test_version_and_strerror(pkg):
    L = pkg.load_library()
    assert L.sc_version() >> 16 == 0
    assert L.sc_strerror(0) == b"ok" and b"hypothesis" in L.sc_strerror(-5) and L.sc_strerror(-99) == b"unknown status"


def test_no_gpu_means_loud_failure(pkg):
    """The product has no CPU fallback: without a HIP device sc_create must fail (SC_EHIP) and the Python
    layer must raise.  (Skipped on a GPU box, where the GPU suite exercises the real path.)"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.SacCotError) as e:
        pkg.Registrar(0)
    assert e.value.status == -3


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under sac-cot_amd/ may import, load or link it."""
    pk = os.path.join(ROOT, "sac-cot_amd")
    for d, _, files in os.walk(pk):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(d, f)).read()
                code = "\n".join(line for line in text.splitlines()
                                 if not line.lstrip().startswith(("//", "#", "*", "/*", '"""')))
                assert "libsaccot_oracle" not in code and "saccot_oracle" not in code, f
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
    out = subprocess.check_output(["readelf", "-d", os.path.join(pk, "libsaccot.so")]).decode()
    assert "oracle" not in out


def test_mex_gateway_compiles_against_the_header(pkg, tmp_path):
    """integration/saccot_mex.cpp cannot run here — no MATLAB / Octave in the image — but it must at least
    meet a compiler: against the declarations-only tests/mex_stub/mex.h and include/saccot.h, warnings as errors, so that
    drift between the gateway and the C ABI breaks THIS test.  (The GPU suite goes further and runs its mexFunction from a
    stand-in runtime: tests/test_gpu_cabi_example.py.)"""
    root = ROOT
    obj = str(tmp_path / "saccot_mex.o")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-c", "-fPIC",
                           "-I", os.path.join(root, "tests", "mex_stub"), "-I", os.path.join(root, "include"),
                           os.path.join(root, "integration", "saccot_mex.cpp"), "-o", obj])
    syms = subprocess.check_output(["nm", "-g", "--defined-only", obj], text=True)
    assert " T mexFunction" in syms
    undefined = subprocess.check_output(["nm", "-u", obj], text=True)
    used = sorted({ln.split()[-1] for ln in undefined.splitlines() if ln.split() and ln.split()[-1].startswith("sc_")})
    assert used, undefined
    L = pkg.load_library()
    for s in used:  # every ABI symbol the gateway binds is exported by the library
        assert hasattr(L, s), s
