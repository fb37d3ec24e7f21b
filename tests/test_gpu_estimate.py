"""GPU: stage B pruned by an ESTIMATED bound (sc_tri.hip 3c).

sc_register / sc_register_device(_async) prune the graph by a bound guessed from a 1-in-64 sample of its triangles instead
of one certified by a 4 M-key sample; the select verifies the guess (it must find T keys at or above it) and a call whose
guess was too high is repeated with a certifying sample.  Either way the outputs are the certified path's, bit for bit —
and that path is what tests/test_gpu_parity.py pins to the CPU restatement.
"""
import numpy as np
import pytest

from conftest import nan_equal_bits

pytestmark = pytest.mark.gpu

KEYS = ("edges", "tri_kept", "tri_scored", "best_rank", "best_count")


def _same(a, b):
    return (a["status"] == b["status"] and np.array_equal(a["mask"], b["mask"]) and nan_equal_bits(a["R"], b["R"])
            and nan_equal_bits(a["t"], b["t"]) and all(a["stats"][k] == b["stats"][k] for k in KEYS))


@pytest.mark.parametrize("name", ["C1", "C2", "C3", "C4"])
def test_estimated_bound_is_verified_and_changes_nothing_but_the_work(pkg, O, name):
    cfg, scene = pkg.synth.make_config_scene(name)
    r = pkg.Registrar(0)
    try:
        est = r.register(scene.src, scene.tgt, **cfg.params())
        assert r.debug_last()["prune_bound"] == 1, r.debug_last()
        again = r.register(scene.src, scene.tgt, **cfg.params())       # the host-free repetition estimates too
        d = r.debug_last()
        assert d["prune_bound"] == 1 and d["fast_path"] == 1 and _same(again, est)
        r.set_debug(no_estimate=1)
        cert = r.register(scene.src, scene.tgt, **cfg.params())
        assert r.debug_last()["prune_bound"] == 0
        assert _same(est, cert)
        assert cfg.T <= est["stats"]["tri_total"]
        print(name, "triangles enumerated: estimated bound", est["stats"]["tri_total"], "certified", cert["stats"]["tri_total"])
    finally:
        r.close()
    if name != "C3":   # (the oracle's C3 pass is minutes of host time; test_gpu_parity.py covers that shape)
        ref = O.register(scene.src, scene.tgt, threads=8, **cfg.params())
        assert np.array_equal(est["mask"], ref["mask"]) and est["stats"]["best_rank"] == ref["best_rank"]
        assert nan_equal_bits(np.concatenate([est["R"].ravel(), est["t"]]), np.concatenate([ref["R"].ravel(), ref["t"]]))


@pytest.mark.parametrize("name", ["C1", "C2"])
def test_estimate_that_is_too_high_is_caught_and_the_call_repeated(pkg, name):
    """est_margin_pct = 1 aims the bound at the key of rank T / 100: the pruned graph then holds far fewer than T triangles
    above it, the select reports the shortfall, the call is repeated with a certifying sample — same outputs — and the
    context certifies for a while (64 calls after the first failure, twice as many after each further one)."""
    import torch
    dev = torch.device("cuda:0")
    cfg, scene = pkg.synth.make_config_scene(name)
    ref_r = pkg.Registrar(0); ref_r.set_debug(no_estimate=1)
    ref = ref_r.register(scene.src, scene.tgt, **cfg.params())
    ref_r.close()
    r = pkg.Registrar(0)
    try:
        r.set_debug(est_margin_pct=1)
        got = r.register(scene.src, scene.tgt, **cfg.params())
        assert r.debug_last()["prune_bound"] == 2, r.debug_last()
        assert _same(got, ref) and got["stats"]["tri_total"] == ref["stats"]["tri_total"]
        nxt = r.register(scene.src, scene.tgt, **cfg.params())          # held off: no second failure right away
        assert r.debug_last()["prune_bound"] == 0 and _same(nxt, ref)
        if name == "C1":
            # r04c: ... but not for ever.  The first failure costs 64 certifying calls; then the context estimates again (here the
            # knob makes that fail again: 128 certifying calls follow, and so on up to 4096)
            seen = [0]
            for _ in range(63 + 1 + 128 + 1):
                r.register(scene.src, scene.tgt, **cfg.params())
                seen.append(r.debug_last()["prune_bound"])
            assert seen[:64] == [0] * 64 and seen[64] == 2 and seen[65:193] == [0] * 128 and seen[193] == 2, seen
        # the same through the host-free form: a good call, then the knob makes the REPEATED shape's estimate fail
        r.set_debug()                                                   # (also forgets the failure)
        ds = torch.from_numpy(scene.src).to(dev); dt = torch.from_numpy(scene.tgt).to(dev)
        p = pkg.make_params(**cfg.params())
        outs = []
        for k in range(3):
            d_Rt = torch.zeros(12, dtype=torch.float32, device=dev); d_mask = torch.zeros(cfg.n, dtype=torch.uint8, device=dev)
            rc, st = r.register_device(ds.data_ptr(), dt.data_ptr(), cfg.n, p, d_Rt.data_ptr(), d_mask.data_ptr())
            torch.cuda.synchronize()
            outs.append((rc, st, d_Rt.cpu().numpy(), d_mask.cpu().numpy(), r.debug_last()))
        assert [o[4]["prune_bound"] for o in outs] == [1, 1, 1] and [o[4]["fast_path"] for o in outs] == [0, 1, 1]
        for rc, st, Rt, mask, _ in outs:
            assert rc == ref["status"] and np.array_equal(mask, ref["mask"]) and st["best_rank"] == ref["stats"]["best_rank"]
            assert nan_equal_bits(Rt, np.concatenate([ref["R"].ravel(), ref["t"]]))
    finally:
        r.close()


def test_estimate_on_graphs_with_few_or_tied_triangles(pkg, O):
    """Fewer triangles than T (the sample never reaches its count: no pruning at all), massive ties (all weights equal:
    every key in one bin), T far above and far below the sample's resolution."""
    r = pkg.Registrar(0)
    try:
        n, tau = 2000, 0.02
        for rho, L, T in ((0.01, 6.0, 10000), (0.20, 1.0, 50), (0.20, 1.0, 3_000_000), (0.0, 1.0, 10000)):
            sc = pkg.synth.make_scene(n, rho, L, tau, 77)
            kw = dict(sigma=tau, t_cmp=0.9, tau=tau, min_len=tau, max_triangles=T, rank_mode=0)
            got = r.register(sc.src, sc.tgt, **kw)
            ref = O.register(sc.src, sc.tgt, threads=8, **kw)
            assert got["status"] == ref["rc"] and np.array_equal(got["mask"], ref["mask"]), (rho, L, T)
            assert got["stats"]["best_rank"] == ref["best_rank"] and got["stats"]["tri_kept"] == ref["t_eff"], (rho, L, T)
        # massive ties: a lattice moved rigidly, sigma huge -> every edge weight rounds to the same value
        g = np.stack(np.meshgrid(np.arange(12), np.arange(12), np.arange(6), indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
        src = g * 0.5
        tgt = src + np.float32(0.25)
        kw = dict(sigma=1e4, t_cmp=0.9, tau=0.01, min_len=0.1, max_triangles=5000, rank_mode=0)
        got = r.register(src, tgt, **kw)
        ref = O.register(src, tgt, threads=8, **kw)
        assert got["status"] == ref["rc"] and np.array_equal(got["mask"], ref["mask"])
        assert got["stats"]["best_rank"] == ref["best_rank"] and got["stats"]["tri_kept"] == ref["t_eff"]
    finally:
        r.close()


@pytest.mark.parametrize("name", ["C0", "C1", "C2", "C4"])
def test_fused_edge_kernel_equals_the_separate_launches(pkg, O, name):
    """launch_edge_build — row statistics, CSR offsets (from the deg+ stage A accumulates), edge list and the estimating
    sample in one launch — against the three launches it replaces on the hot path (sc_debug.no_edge_build): same edges,
    same bound (the fused sample recomputes the weights it cannot look up yet: bit-identical, so even the number of
    triangles enumerated agrees), same outputs; first call, host-free repetition, a different size in between."""
    cfg, scene = pkg.synth.make_config_scene(name)
    cfg0, scene0 = pkg.synth.make_config_scene("C0")
    a = pkg.Registrar(0)
    b = pkg.Registrar(0); b.set_debug(no_edge_build=1)
    try:
        for k in range(3):
            ra = a.register(scene.src, scene.tgt, **cfg.params())
            rb = b.register(scene.src, scene.tgt, **cfg.params())
            assert _same(ra, rb) and ra["stats"]["tri_total"] == rb["stats"]["tri_total"], (name, k)
            assert a.debug_last()["prune_bound"] == b.debug_last()["prune_bound"]
            if k == 1:
                x = a.register(scene0.src, scene0.tgt, **cfg0.params())
                y = b.register(scene0.src, scene0.tgt, **cfg0.params())
                assert _same(x, y)
    finally:
        a.close(); b.close()
    ref = O.register(scene.src, scene.tgt, threads=8, **cfg.params())
    assert np.array_equal(ra["mask"], ref["mask"]) and ra["stats"]["best_rank"] == ref["best_rank"] and ra["stats"]["edges"] == ref["edges"]


def _ties_scene():
    """a lattice moved rigidly under a huge sigma: every edge weight rounds to the same value -> every key ties"""
    g = np.stack(np.meshgrid(np.arange(12), np.arange(12), np.arange(6), indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    src = g * 0.5
    return src, src + np.float32(0.25), dict(sigma=1e4, t_cmp=0.9, tau=0.01, min_len=0.1, rank_mode=0)


@pytest.mark.parametrize("T", [5000, 37, 3_000_000])
def test_the_select_with_every_key_tied_equals_the_oracle(pkg, O, T):
    """the two select rounds + count + write on a lattice moved rigidly under a huge sigma — every edge weight rounds to the same
    value, every key ties, the threshold's `need_eq` decides the whole selection by ordinal — with T below, far below and above
    the number of triangles; the waited call and its host-free repetition"""
    src, tgt, kw = _ties_scene()
    kw = dict(kw, max_triangles=T)
    r = pkg.Registrar(0)
    try:
        outs = [r.register(src, tgt, **kw) for _ in range(2)]
    finally:
        r.close()
    ref = O.register(src, tgt, threads=8, **kw)
    for o in outs:
        assert o["status"] == ref["rc"] and np.array_equal(o["mask"], ref["mask"])
        assert o["stats"]["best_rank"] == ref["best_rank"] and o["stats"]["tri_kept"] == ref["t_eff"]


@pytest.mark.parametrize("n,rho,L,tau,T", [(8200, 0.12, 3.0, 0.1, 30000), (10007, 0.10, 8.0, 0.2, 60000), (12345, 0.08, 20.0, 0.5, 5000)])
def test_fused_edge_kernel_on_rows_beyond_128_words(pkg, O, n, rho, L, tau, T):
    """edge_build_kernel<5> (bit rows of 129 .. 320 words; C3 is the only BASELINE shape there): ragged sizes, against the
    separate launches and the oracle."""
    sc = pkg.synth.make_scene(n, rho, L, tau, 4242 + n)
    kw = dict(sigma=tau, t_cmp=0.9, tau=tau, min_len=tau, max_triangles=T, rank_mode=0)
    a = pkg.Registrar(0)
    b = pkg.Registrar(0); b.set_debug(no_edge_build=1)
    try:
        ra = a.register(sc.src, sc.tgt, **kw); rb = b.register(sc.src, sc.tgt, **kw)
        ra2 = a.register(sc.src, sc.tgt, **kw)
        assert _same(ra, rb) and _same(ra2, rb) and ra["stats"]["tri_total"] == rb["stats"]["tri_total"]
    finally:
        a.close(); b.close()
    ref = O.register(sc.src, sc.tgt, threads=8, **kw)
    assert ra["status"] == ref["rc"] and np.array_equal(ra["mask"], ref["mask"]) and ra["stats"]["edges"] == ref["edges"]
    assert ra["stats"]["best_rank"] == ref["best_rank"] and ra["stats"]["tri_kept"] == ref["t_eff"]


def test_replicated_ranks_prune_by_the_estimated_bound_too(pkg):
    """SC_FLAG_EST_BOUND on sc_hypothesize_device (r04b; stages A and B replicated, stage C dealt over 3 ranks): same winner, mask and
    motion as the certified form; a bound that is too high comes back as SC_EBOUND from the finalize call, on the rank that made it as on
    every other, and the repeat without the flag succeeds; the _begin / _end pair (a SHARED, certifying sample) refuses the flag."""
    import torch
    dev = torch.device("cuda:0")
    cfg, scene = pkg.synth.make_config_scene("C2")
    world, block = 3, 1024
    d_src = torch.from_numpy(scene.src).to(dev); d_tgt = torch.from_numpy(scene.tgt).to(dev)
    d_keys = torch.zeros(2 * world, dtype=torch.int64, device=dev)
    d_Rt = torch.zeros(12, dtype=torch.float32, device=dev); d_mask = torch.zeros(cfg.n, dtype=torch.uint8, device=dev)
    r = pkg.Registrar(0)
    try:
        def run(flags):
            for k in range(world):
                p = pkg.make_params(shard_rank=k, shard_world=world, shard_block=block, flags=flags, **cfg.params())
                r.hypothesize_device(d_src.data_ptr(), d_tgt.data_ptr(), cfg.n, p, d_keys.data_ptr() + 16 * k)
            rc, st = r.finalize_gathered_device(d_keys.data_ptr(), world, d_Rt.data_ptr(), d_mask.data_ptr())
            torch.cuda.synchronize()
            return rc, st, d_Rt.cpu().numpy().copy(), d_mask.cpu().numpy().copy(), r.debug_last()
        rc0, st0, Rt0, m0, d0 = run(0)
        assert rc0 == 0 and d0["prune_bound"] == 0
        rc1, st1, Rt1, m1, d1 = run(pkg.SC_FLAG_EST_BOUND)
        assert rc1 == 0 and d1["prune_bound"] == 1, d1
        assert np.array_equal(m0, m1) and np.array_equal(Rt0.view(np.uint32), Rt1.view(np.uint32))
        assert st0["best_rank"] == st1["best_rank"] and st0["best_count"] == st1["best_count"]
        assert st1["tri_total"] < st0["tri_total"]                         # (the estimated bound prunes harder)
        r.set_debug(est_margin_pct=1)                                      # aims at rank T / 100: too high
        rc2, _, _, _, d2 = run(pkg.SC_FLAG_EST_BOUND)
        assert rc2 == pkg.SC_EBOUND and d2["prune_bound"] == 2, (rc2, d2)
        rc3, st3, Rt3, m3, d3 = run(0)                                     # the repeat the caller owes
        assert rc3 == 0 and np.array_equal(m0, m3) and np.array_equal(Rt0.view(np.uint32), Rt3.view(np.uint32))
        r.set_debug()
        p = pkg.make_params(shard_rank=0, shard_world=world, shard_block=block, flags=pkg.SC_FLAG_EST_BOUND, **cfg.params())
        d_hist = torch.zeros(pkg.SC_HIST_WORDS, dtype=torch.int32, device=dev)
        with pytest.raises(pkg.SacCotError) as e:
            r.hypothesize_begin_device(d_src.data_ptr(), d_tgt.data_ptr(), cfg.n, p, d_hist.data_ptr())
        assert e.value.status == pkg.SC_EINVAL
    finally:
        r.close()
