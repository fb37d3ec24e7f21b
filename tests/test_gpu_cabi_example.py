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
