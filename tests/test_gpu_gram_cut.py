"""GPU: the Gram filter's reference frame and its cut (sc_score.hip, "the frame" / "the cut"; sc_gramref.hpp).

Stage C2's Gram filter evaluates the squared residual in the frame of a reference hypothesis that 64 voters elect, and a
hypothesis near that frame only looks at the correspondences near it — every other test is decided by the triangle inequality.
None of this may change a count: each case below compares EVERY count with the plain fp32 kernel's (sc_debug.score_filter = 1)
on inputs built to sit on the cut's edges — two rigid motions in one scene (one mode is the frame, the other is far from it),
correspondences spread across the near / far boundary, hypotheses spread across the near / not-near boundary, no consensus at all,
NaN voters, fewer hypotheses than voters — and the whole path with the vote taken early (by the counting pass) and late.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _params(tau, T):
    return dict(sigma=tau, t_cmp=0.9, tau=tau, min_len=tau, max_triangles=T, rank_mode=0)


def _rot(axis, angle):
    axis = np.asarray(axis, dtype=np.float64); axis = axis / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)


def _random_rotations(rng, T):
    q = rng.normal(size=(T, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    return np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                     2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                     2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], axis=1).reshape(T, 3, 3)


def _perturbed(rng, R, t, T, ang, dt, centre):
    """T motions around (R, t): rotation by up to `ang` about a random axis through `centre`, translation off by up to `dt`."""
    out = np.empty((T, 12))
    for h in range(T):
        dR = _rot(rng.normal(size=3), float(rng.uniform(0, ang)))
        Rh = dR @ R
        th = (dR @ (R @ centre + t) - Rh @ centre) + rng.uniform(-1, 1, 3) * dt   # the image of `centre` stays put, up to dt
        out[h, :9] = Rh.ravel(); out[h, 9:] = th
    return out


def _both(pkg, src, tgt, prm, Rt, **knobs):
    reg = pkg.Registrar(0); plain = pkg.Registrar(0)
    try:
        reg.set_debug(score_filter=3, **knobs); plain.set_debug(score_filter=1)
        c_g, k_g = reg.score(src, tgt, prm, Rt)
        info = reg.debug_last()
        c_p, k_p = plain.score(src, tgt, prm, Rt)
        return c_g, k_g, c_p, k_p, info
    finally:
        reg.close(); plain.close()


def test_two_rigid_motions_one_is_the_frame_the_other_far_from_it(pkg):
    rng = np.random.default_rng(41)
    n, T, tau = 4096, 36864, 0.03
    p = rng.uniform(-1, 1, (n, 3))
    RA, tA = _rot([1, 2, 3], 0.7), np.array([0.3, -0.2, 0.1])
    RB, tB = _rot([-2, 1, 0.5], 2.1), np.array([-0.8, 0.5, 0.9])
    q = rng.uniform(-2, 2, (n, 3))
    a = rng.random(n) < 0.30; b = ~a & (rng.random(n) < 0.35)
    q[a] = p[a] @ RA.T + tA + rng.normal(size=(int(a.sum()), 3)) * tau / 3
    q[b] = p[b] @ RB.T + tB + rng.normal(size=(int(b.sum()), 3)) * tau / 3
    c0 = np.zeros(3)
    Rt = np.concatenate([_perturbed(rng, RA, tA, T // 2, 0.01, 2 * tau, c0), _perturbed(rng, RB, tB, T // 4, 0.01, 2 * tau, c0),
                         np.concatenate([_random_rotations(rng, T // 4).reshape(-1, 9), rng.uniform(-1, 1, (T // 4, 3))], axis=1)])
    Rt = Rt[rng.permutation(T)].astype(np.float32)
    src, tgt = p.astype(np.float32), q.astype(np.float32)
    c_g, k_g, c_p, k_p, info = _both(pkg, src, tgt, pkg.make_params(**_params(tau, T)), Rt)
    assert info["c2_kernel"] == 2 and np.array_equal(c_g, c_p) and k_g == k_p, info
    assert c_p.max() > 0.2 * n                                            # (mode A's hypotheses do see their inliers)
    # one mode is near the frame, the other (and the random quarter) is not; the near correspondences are that mode's
    assert 0.2 * T < info["gram_near_hyp"] < 0.6 * T and 0.1 * n < info["gram_near_corr"] < 0.6 * n, info
    print("two motions:", {k: info[k] for k in info if k.startswith("gram_") or k.startswith("filter_")})


def test_correspondences_and_hypotheses_spread_across_both_boundaries_of_the_cut(pkg):
    """|V| (the distance of a correspondence from the frame's prediction) spread over 0 .. 3 x the near radius (8 + 1.05) tau, and
    hypotheses spread over 0 .. 3 x the near reach of 8 tau: whatever falls on which side, every count is the fp32 kernel's — also
    with the radius at 1/16 tau (nothing is near), at 4 tau and at 32 tau (sc_debug.gram_kappa_q4)."""
    rng = np.random.default_rng(42)
    n, T, tau = 5000, 28160, 0.02
    p = rng.uniform(-1, 1, (n, 3))
    R, t = _rot([0.2, -1, 0.4], 1.1), np.array([0.1, 0.2, -0.3])
    u = rng.normal(size=(n, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    d = rng.uniform(0, 3 * 9.05 * tau, n)
    d[: n // 4] = np.abs(rng.normal(size=n // 4)) * tau / 2                # a quarter are true inliers
    q = p @ R.T + t + u * d[:, None]
    # reach = |dM|_F |half diagonal| + |tau'| ~ 2.45 x angle + |translation offset|: both up to ~1.5 x the near reach of 8 tau
    Rt = _perturbed(rng, R, t, T, 0.08, 0.08, np.zeros(3)).astype(np.float32)
    src, tgt = p.astype(np.float32), q.astype(np.float32)
    prm = pkg.make_params(**_params(tau, T))
    seen = []
    for kq4 in (0, 1, 64, 512):
        c_g, k_g, c_p, k_p, info = _both(pkg, src, tgt, prm, Rt, gram_kappa_q4=kq4)
        assert info["c2_kernel"] == 2 and np.array_equal(c_g, c_p) and k_g == k_p, (kq4, info)
        seen.append((kq4, info["gram_near_hyp"], info["gram_near_corr"]))
    by = {k: (h, c) for k, h, c in seen}
    assert 0.03 * T < by[0][0] < 0.97 * T and 0.2 * n < by[0][1] < 0.8 * n, seen   # the default radius really cuts through both
    assert by[1][0] <= 1 and by[64][0] < by[0][0] < by[512][0] and by[64][1] < by[0][1] <= by[512][1], seen
    print("boundaries:", seen)


def test_no_consensus_nan_voters_and_fewer_hypotheses_than_voters(pkg):
    rng = np.random.default_rng(43)
    n, tau = 3000, 0.05
    p = rng.uniform(-1, 1, (n, 3)); R, t = _rot([1, 0, 1], 0.4), np.array([0.5, 0.0, -0.1])
    q = p @ R.T + t + rng.normal(size=(n, 3)) * tau / 3
    out = rng.random(n) < 0.5
    q[out] = rng.uniform(-2, 2, (int(out.sum()), 3))
    src, tgt = p.astype(np.float32), q.astype(np.float32)
    # (a) every hypothesis random: no two voters agree, the frame is one of them, nothing but itself is near
    T = 45056
    Rt = np.concatenate([_random_rotations(rng, T).reshape(-1, 9), rng.uniform(-1, 1, (T, 3))], axis=1).astype(np.float32)
    c_g, k_g, c_p, k_p, info = _both(pkg, src, tgt, pkg.make_params(**_params(tau, T)), Rt)
    assert np.array_equal(c_g, c_p) and k_g == k_p and info["gram_near_hyp"] < 0.01 * T, info
    # (b) the voters' positions (every T / 64-th hypothesis) hold NaN / inf / zero matrices, the rest is the truth: the box-centre frame
    Rt = np.tile(np.concatenate([R.ravel(), t]).astype(np.float32), (T, 1))
    Rt[:, 9:] += rng.normal(size=(T, 3)).astype(np.float32) * tau / 4
    voters = (np.arange(64, dtype=np.int64) * T) // 64
    Rt[voters[0::2], 4] = np.nan; Rt[voters[1::2], 10] = np.inf
    c_g, k_g, c_p, k_p, info = _both(pkg, src, tgt, pkg.make_params(**_params(tau, T)), Rt)
    assert np.array_equal(c_g, c_p) and k_g == k_p and info["gram_ref"] == 0xFFFFFFFF, info
    assert c_p.max() > 0.4 * n
    # (b') ... or all-zero "rotations": finite, they agree with each other perfectly and win the vote — the frame (a half turn
    # rebuilt from the zero matrix) is a valid rotation all the same; every true hypothesis is far from it
    Rt[voters] = 0.0
    c_g, k_g, c_p, k_p, info = _both(pkg, src, tgt, pkg.make_params(**_params(tau, T)), Rt)
    assert np.array_equal(c_g, c_p) and k_g == k_p and info["gram_ref"] != 0xFFFFFFFF and info["gram_near_hyp"] < 0.01 * T, info
    # (c) fewer hypotheses than voters (the filter forced on a tiny call), and exactly one
    for T2 in (1, 7, 63, 300):
        Rt2 = np.tile(np.concatenate([R.ravel(), t]).astype(np.float32), (T2, 1))
        Rt2[:, 9:] += rng.normal(size=(T2, 3)).astype(np.float32) * tau
        c_g, k_g, c_p, k_p, info = _both(pkg, src, tgt, pkg.make_params(**_params(tau, T2)), Rt2)
        assert info["c2_kernel"] == 2 and np.array_equal(c_g, c_p) and k_g == k_p, (T2, info)


@pytest.mark.parametrize("name", ["C2", "C4"])
def test_whole_path_vote_under_the_counting_pass_equals_vote_after_the_selection(pkg, name):
    """sc_register: the frame voted by an extra workgroup of stage B's counting pass among the estimating sample's best triangles
    (default) and by a launch of its own among the selection (sc_debug.gram_ref_late) are different frames; the outputs are not."""
    cfg, scene = pkg.synth.make_config_scene(name)
    r = pkg.Registrar(0)
    try:
        a = r.register(scene.src, scene.tgt, **cfg.params()); da = r.debug_last()
        a2 = r.register(scene.src, scene.tgt, **cfg.params()); da2 = r.debug_last()      # (host-free repetition)
        r.set_debug(gram_ref_late=1)
        b = r.register(scene.src, scene.tgt, **cfg.params()); db = r.debug_last()
        r.set_debug(score_filter=1)
        c = r.register(scene.src, scene.tgt, **cfg.params())
        assert da["c2_kernel"] == 2 and db["c2_kernel"] == 2 and da2["fast_path"] == 1, (da, da2, db)
        for x in (a2, b, c):
            assert np.array_equal(a["mask"], x["mask"]) and np.array_equal(a["R"].view(np.uint32), x["R"].view(np.uint32))
            assert a["stats"]["best_rank"] == x["stats"]["best_rank"] and a["stats"]["best_count"] == x["stats"]["best_count"]
        for d in (da, da2, db):   # either way most hypotheses are near the frame and few correspondences are
            assert d["gram_ref"] != 0xFFFFFFFF and d["gram_near_hyp"] > 0.5 * d["gram_rows"] and d["gram_near_corr"] < 0.4 * cfg.n, d
        print(name, "early:", {k: da[k] for k in da if k.startswith("gram_n") or k.startswith("gram_r")}, "late:",
              {k: db[k] for k in db if k.startswith("gram_n") or k.startswith("gram_r")})
    finally:
        r.close()


@pytest.mark.parametrize("block", range(3))
def test_whole_path_on_random_scenes_equals_the_fp32_kernel(pkg, block):
    """sc_register on 5 x 3 random synthetic scenes — size, inlier ratio, tau against the extent, T (all big enough for the host to
    pick a matrix-pipe filter) drawn at random — with everything at its default (estimated bound, fused edge kernel, vote under the
    counting pass, Gram filter in the voted frame with its cut) against the same call with the plain fp32 scoring kernel and the
    certifying sample: identical mask, motion bits, winner rank and count; and the default did run the Gram filter in most draws."""
    rng = np.random.default_rng(7100 + block)
    gram = 0
    a = pkg.Registrar(0); b = pkg.Registrar(0)
    try:
        b.set_debug(score_filter=1, no_estimate=1)
        for it in range(5):
            n = int(rng.integers(3000, 9001))
            rho = float(rng.uniform(0.06, 0.4))
            L = float(10.0 ** rng.uniform(-1, 2))
            tau = L * float(10.0 ** rng.uniform(np.log10(0.006), np.log10(0.05)))
            T = int(((1 << 27) // n + int(rng.integers(1, 40000))) // 256 * 256)
            sc = pkg.synth.make_scene(n, rho, L, tau, int(rng.integers(1, 1 << 30)))
            kw = dict(sigma=tau, t_cmp=0.9, tau=tau, min_len=tau, max_triangles=T, rank_mode=0)
            ra = a.register(sc.src, sc.tgt, **kw); da = a.debug_last()
            rb = b.register(sc.src, sc.tgt, **kw)
            tag = (block, it, n, rho, L, tau, T, da)
            assert ra["status"] == rb["status"], tag
            assert np.array_equal(ra["mask"], rb["mask"]) and np.array_equal(ra["R"].view(np.uint32), rb["R"].view(np.uint32)), tag
            assert np.array_equal(ra["t"].view(np.uint32), rb["t"].view(np.uint32)), tag
            assert ra["stats"]["best_rank"] == rb["stats"]["best_rank"] and ra["stats"]["best_count"] == rb["stats"]["best_count"], tag
            gram += int(da["c2_kernel"] == 2)
        assert gram >= 3, gram
    finally:
        a.close(); b.close()
