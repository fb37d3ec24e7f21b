"""GPU suite, part 2: the HIP path against the COMMITTED golden vectors (tests/golden/*.npz) — no oracle call here,
so this also passes judgement on the fixtures themselves from the other side."""
import hashlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", ["micro64", "c0"])
def test_hip_path_reproduces_golden(pkg, reg, name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    s, tc, tau, ml = (float(x) for x in g["params"])
    kw = dict(sigma=s, t_cmp=tc, tau=tau, min_len=ml, max_triangles=int(g["T"]), rank_mode=0)
    p = pkg.make_params(**kw)
    S, bits, deg = reg.compat(g["src"], g["tgt"], p)
    assert np.array_equal(bits, g["bits"]) and np.array_equal(deg, g["deg"])
    assert hashlib.sha256(S.tobytes()).digest() == g["S_sha256"].tobytes()
    tri, key, total, edges = reg.triangles(g["src"], g["tgt"], p)
    assert (total, edges) == (int(g["tri_total"]), int(g["edges"]))
    assert np.array_equal(tri, g["tri"]) and np.array_equal(key, g["key"])
    Rt = reg.kabsch(g["src"], g["tgt"], p, tri)
    assert Rt.tobytes() == g["Rt"].tobytes()
    cnt, k = reg.score(g["src"], g["tgt"], p, Rt)
    assert np.array_equal(cnt, g["cnt"])
    assert (k >> 32, 0xFFFFFFFF - (k & 0xFFFFFFFF)) == (int(g["best_count"]), int(g["best_rank"]))
    out = reg.register(g["src"], g["tgt"], **kw)
    assert out["status"] == 0 and np.array_equal(out["mask"], g["mask"])
    assert out["R"].tobytes() == g["R"].tobytes() and out["t"].tobytes() == g["t"].tobytes()
