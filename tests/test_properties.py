"""CPU suite, part 3: property tests (SURVEY §4 tier 3) of the CPU restatement and of the host-side sharding logic,
driven by `hypothesis` (random sizes, seeds, parameters and partitions instead of a handful of fixed seeds).
PARITY UNPINNED by the reference (README.md:1-2 only): these pin the restatement against its own algebra."""
import numpy as np
from hypothesis import HealthCheck, given, settings, strategies as st

SET = dict(deadline=None, max_examples=20, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])


def _scene(pkg, n, rho, seed, tau=0.05):
    return pkg.synth.make_scene(n, rho, 1.0, tau, seed)


@settings(**SET)
@given(n=st.integers(40, 220), rho=st.floats(0.3, 0.9), seed=st.integers(0, 10_000), data=st.data())
def test_permuting_the_correspondences_permutes_the_mask(pkg, O, n, rho, seed, data):
    """Relabelling the correspondences relabels the graph: same number of edges and triangles, the same best inlier
    count, and — when that count is attained by one hypothesis only — the same (R,t) bits and the permuted mask."""
    sc = _scene(pkg, n, rho, seed)
    kw = dict(sigma=0.05, t_cmp=0.9, tau=0.05, min_len=0.05, max_triangles=10 ** 7)     # every triangle: no cut-off ties
    a = O.register(sc.src, sc.tgt, threads=1, **kw)
    perm = np.array(data.draw(st.permutations(list(range(n)))))
    b = O.register(sc.src[perm], sc.tgt[perm], threads=1, **kw)
    assert (a["rc"], a["edges"], a["tri_total"], a["best_count"]) == (b["rc"], b["edges"], b["tri_total"], b["best_count"])
    if a["rc"] == 0:
        assert int(a["mask"].sum()) == int(b["mask"].sum()) == a["best_count"]
        S, bits, deg = O.compat(sc.src, sc.tgt, 0.05, 0.9, 0.05, 0.05)
        tri, _, _ = O.triangles(S, bits, deg, 10 ** 7, 0)
        cnt = O.score(sc.src, sc.tgt, O.kabsch3(sc.src, sc.tgt, tri), 0.05)
        if int((cnt == cnt.max()).sum()) == 1:                     # a unique winner: the triangle is the same point set
            assert np.array_equal(b["mask"], a["mask"][perm])


@settings(**SET)
@given(n=st.integers(30, 200), seed=st.integers(0, 10_000), ang=st.floats(-3.0, 3.0), tx=st.floats(-2, 2))
def test_noise_free_all_inlier_scene_recovers_ground_truth(pkg, O, n, seed, ang, tx):
    """q = R p + t exactly (up to fp32 rounding of q): every pair is compatible with residual ~0, every hypothesis of
    a non-degenerate triangle is the ground truth, all n correspondences are inliers."""
    rng = np.random.default_rng(seed)
    src = rng.uniform(-0.5, 0.5, (n, 3)).astype(np.float32)
    c, s = np.cos(ang), np.sin(ang)
    R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
    tgt = (src.astype(np.float64) @ R.T + np.array([tx, 0.3, -0.2])).astype(np.float32)
    out = O.register(src, tgt, threads=1, sigma=0.05, t_cmp=0.9, tau=0.05, min_len=0.05, max_triangles=500)
    assert out["rc"] == 0 and out["best_count"] == n and out["mask"].all()
    assert pkg.synth.rotation_error_deg(out["R"], R) < 0.2 and np.linalg.norm(out["t"] - np.array([tx, 0.3, -0.2])) < 5e-3


@settings(**SET)
@given(n=st.integers(60, 300), seed=st.integers(0, 10_000), ang=st.floats(-3.0, 3.0))
def test_rigid_motion_of_the_target_keeps_the_graph(pkg, O, n, seed, ang):
    """Edges depend on pairwise lengths only: moving the target cloud rigidly may flip only pairs within fp32 rounding
    of a threshold."""
    sc = _scene(pkg, n, 0.4, seed)
    _, bits, _ = O.compat(sc.src, sc.tgt, 0.05, 0.9, 0.05, 0.05, want_S=False)
    c, s = np.cos(ang), np.sin(ang)
    Rz = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
    tgt2 = (sc.tgt.astype(np.float64) @ Rz.T + np.array([0.3, -0.2, 0.1])).astype(np.float32)
    _, bits2, _ = O.compat(sc.src, tgt2, 0.05, 0.9, 0.05, 0.05, want_S=False)
    diff = int(np.unpackbits((bits ^ bits2).view(np.uint8)).sum()); total = int(np.unpackbits(bits.view(np.uint8)).sum())
    assert diff <= 0.004 * total + 6


@settings(**SET)
@given(t_eff=st.integers(0, 5000), block=st.integers(1, 700), world=st.integers(1, 9))
def test_block_dealing_partitions_the_list(pkg, t_eff, block, world):
    """Stage C's dealing of the selected list (blocks of `block` positions, round-robin over `world` ranks): the ranks'
    index sets partition 0 .. t_eff - 1, the counts agree with shard.local_count, and the local order is ascending."""
    seen = []
    for r in range(world):
        idx = pkg.shard.local_indices(t_eff, block, r, world)
        assert len(idx) == pkg.shard.local_count(t_eff, block, r, world)
        assert np.all(np.diff(idx) > 0) and np.all((idx // block) % world == r)
        seen.append(idx)
    allidx = np.sort(np.concatenate(seen)) if seen else np.array([], np.int64)
    assert np.array_equal(allidx, np.arange(t_eff))


@settings(**SET)
@given(st.lists(st.tuples(st.integers(0, 2 ** 20), st.integers(0, 2 ** 32 - 1), st.integers(0, 2 ** 31)), min_size=1, max_size=9))
def test_key_pair_reduction_is_a_lexicographic_max(pkg, triples):
    """The two-step reduction of the winner key pairs (max K0, then max K1 among the ranks attaining it) picks: most
    inliers, then best ranking key, then lowest position — whatever the order of the ranks."""
    pairs = [pkg.shard.encode_pair(c, k, p) for c, k, p in triples]
    k0, k1 = pkg.shard.reduce_pairs(pairs)
    live = [(c, k, p) for c, k, p in triples if c > 0]
    if not live:
        assert (k0, k1) == (0, 0)
        return
    best = max(live, key=lambda x: (x[0], x[1], -x[2]))
    assert pkg.shard.decode_pair(k0, k1) == best
    assert pkg.shard.reduce_pairs(pairs[::-1]) == (k0, k1)
