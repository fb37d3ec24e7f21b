"""Build recipe for libsaccot.so: hipcc, gfx950 only, in-tree (the .so travels with the repo snapshot)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsaccot.so")
SOURCES = ["sc_compat.hip", "sc_tri.hip", "sc_score.hip", "sc_sort.hip", "sc_capi.hip", "sc_multi.hip"]
HEADERS = ["sc_arith.hpp", "sc_block.hpp", "sc_kernels.hpp", "sc_gramref.hpp", os.path.join("..", "..", "include", "saccot.h"),
           os.path.join("..", "..", "include", "saccot_debug.h")]
# -ffp-contract=off: the canonical arithmetic (sc_arith.hpp) fuses only where it says fmaf.
# -amdgpu-mfma-vgpr-form: MFMA results land in VGPRs (unified register file on gfx950), no v_accvgpr_read copies.
# -fno-slp-vectorize / -fno-vectorize: the vectorizers pack the fp32 chains into v_pk_* + v_mov shuffles, slower than plain VALU on gfx950.
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-slp-vectorize", "-fno-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result"]


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False, ablations: bool = False) -> str:
    """ablations: also compile the scheduling / timing-only variants of the C2 filter kernels (-DSC_ABLATIONS; the lab
    build tools/pmc_gram_variants.sh and tools/ab_stage.py sweeps use).  The default build — what ships and what the tests
    load — holds variant 0 only; the two builds keep their objects in different directories and share the .so name, so an
    ablation build must be followed by a default build before anything else runs."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(CSRC, "build_ablations" if ablations else "build")
    flags = FLAGS + (["-DSC_ABLATIONS"] if ablations else [])
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            cmd = [hipcc] + flags + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
    tag = os.path.join(CSRC, "build", ".linked_from")  # which object set the .so was last linked from
    os.makedirs(os.path.dirname(tag), exist_ok=True)
    last = open(tag).read().strip() if os.path.exists(tag) else ""
    if force or _stale(LIB, objs) or last != objdir:
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl", "-lpthread"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        with open(tag, "w") as f:
            f.write(objdir)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True, ablations="--ablations" in sys.argv))
