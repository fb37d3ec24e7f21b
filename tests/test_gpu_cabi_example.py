"""GPU suite: the C ABI from COMPILED code — integration/example_register.cpp is built with g++ against
include/saccot.h and sac-cot_amd/libsaccot.so (no Python, no torch in that process) and must reproduce the CPU
restatement's result on a correspondence file."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--devices", "0"], ["--loopback", "3"]])
def test_cpp_program_links_and_matches_the_oracle(pkg, O, tmp_path, extra):
    """plain sc_register; the native multi-device entry with one device (no RCCL call); and its loopback form (three
    ranks on this one GPU: the whole sharded orchestration from compiled code)"""
    exe = str(tmp_path / "example_register")
    lib_dir = os.path.join(ROOT, "sac-cot_amd")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "integration", "example_register.cpp"), "-L", lib_dir, "-lsaccot",
                           f"-Wl,-rpath,{lib_dir}", "-o", exe])
    cfg, scene = pkg.synth.make_config_scene("C1")
    path = str(tmp_path / "corr.txt")
    pkg.corrio.save_correspondences(path, scene.src, scene.tgt)          # %.9g: float32 round-trips exactly
    out = subprocess.run([exe, path, repr(float(np.float32(cfg.tau))), str(cfg.T)] + extra, capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 0, out.stderr
    head, rline, tline = out.stdout.strip().splitlines()
    fields = dict(zip(head.split()[0::2], head.split()[1::2]))
    ref = O.register(scene.src, scene.tgt, threads=1, **cfg.params())
    assert int(fields["n"]) == cfg.n and int(fields["edges"]) == ref["edges"]
    assert int(fields["best_rank"]) == ref["best_rank"] and int(fields["inliers"]) == int(ref["mask"].sum())
    R = np.array(rline.split()[1:], dtype=np.float32).reshape(3, 3)
    t = np.array(tline.split()[1:], dtype=np.float32)
    assert np.array_equal(R, ref["R"]) and np.array_equal(t, ref["t"])   # %.9g prints float32 exactly


@pytest.mark.gpu
@pytest.mark.parametrize("refine", [0, 1])
def test_mex_gateway_runs_under_a_stand_in_runtime(pkg, O, tmp_path, refine):
    """integration/saccot_mex.cpp, compiled against tests/mex_stub/mex.h and RUN from tests/mex_stub/mex_host.cpp — a
    stand-in for the MATLAB runtime (column-major N x 3 singles in, a parameter struct, [R, t, inl] out).  Not MATLAB: what
    it pins is the gateway's own logic — SoA layout, parameter fields, the row-major -> column-major R, the logical mask —
    against the CPU restatement."""
    exe = str(tmp_path / "mex_host")
    lib_dir = os.path.join(ROOT, "sac-cot_amd")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "tests", "mex_stub"),
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "integration", "saccot_mex.cpp"),
                           os.path.join(ROOT, "tests", "mex_stub", "mex_host.cpp"), "-L", lib_dir, "-lsaccot",
                           f"-Wl,-rpath,{lib_dir}", "-o", exe])
    cfg, scene = pkg.synth.make_config_scene("C1")
    path = str(tmp_path / "corr.txt")
    pkg.corrio.save_correspondences(path, scene.src, scene.tgt)
    out = subprocess.run([exe, path, repr(float(np.float32(cfg.tau))), str(cfg.T), str(refine)], capture_output=True,
                         text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    rline, tline, mline = out.stdout.strip().splitlines()
    ref = O.register(scene.src, scene.tgt, threads=1, **cfg.params())
    if refine:  # 'refine' = 1 -> SC_FLAG_REFINE: the fp64 least-squares refit over the winner's inliers
        done, Rt = O.refine(scene.src, scene.tgt, ref["mask"], np.concatenate([ref["R"].ravel(), ref["t"]]))
        assert done
        ref = dict(ref, R=Rt[:9].reshape(3, 3), t=Rt[9:])
    R = np.array(rline.split()[1:], dtype=np.float32).reshape(3, 3)
    t = np.array(tline.split()[1:], dtype=np.float32)
    assert np.array_equal(R, ref["R"]) and np.array_equal(t, ref["t"])
    mask = np.frombuffer(mline.split("mask")[1].encode(), dtype=np.uint8) - ord("0")
    assert np.array_equal(mask, ref["mask"]) and int(mline.split()[3]) == int(ref["mask"].sum())
