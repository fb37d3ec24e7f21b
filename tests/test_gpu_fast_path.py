"""GPU: the host-free form of sc_register_device (include/saccot.h, sc_register_device_async / sc_wait).

A call that repeats the shape of the previous call on its context enqueues the whole chain without waiting for stage B's
two counts; sc_wait validates and, when a count outgrew what the launches covered (or the graph had fewer triangles than
T), repeats the call the waiting way.  Either way every output must be the waiting path's, bit for bit — and that path
is what tests/test_gpu_parity.py pins to the CPU restatement.
"""
import numpy as np
import pytest

from conftest import nan_equal_bits

pytestmark = pytest.mark.gpu


def _dev(torch, scene, dev):
    return torch.from_numpy(scene.src).to(dev), torch.from_numpy(scene.tgt).to(dev)


def _same(a, b):
    return (np.array_equal(a["mask"], b["mask"]) and nan_equal_bits(a["Rt"], b["Rt"]) and a["rc"] == b["rc"]
            and all(a["st"][k] == b["st"][k] for k in ("edges", "tri_total", "tri_kept", "tri_scored", "best_rank", "best_count")))


def _run(torch, r, ds, dt, n, p, dev):
    d_Rt = torch.zeros(12, dtype=torch.float32, device=dev)
    d_mask = torch.full((n,), 7, dtype=torch.uint8, device=dev)
    rc, st = r.register_device(ds.data_ptr(), dt.data_ptr(), n, p, d_Rt.data_ptr(), d_mask.data_ptr())
    torch.cuda.synchronize()
    return dict(rc=rc, st=st, Rt=d_Rt.cpu().numpy(), mask=d_mask.cpu().numpy(), fast=r.debug_last()["fast_path"])


@pytest.mark.parametrize("name", ["C1", "C2", "C4"])
def test_host_free_repeats_are_the_waited_call(pkg, O, name):
    import torch
    dev = torch.device("cuda:0")
    cfg, scene = pkg.synth.make_config_scene(name)
    ds, dt = _dev(torch, scene, dev)
    p = pkg.make_params(**cfg.params())
    r = pkg.Registrar(0)
    try:
        r.set_stream(torch.cuda.current_stream().cuda_stream)
        first = _run(torch, r, ds, dt, cfg.n, p, dev)
        assert first["fast"] == 0                                   # a first call waits
        for _ in range(3):
            again = _run(torch, r, ds, dt, cfg.n, p, dev)
            assert again["fast"] == 1, "a repeated shape did not take the host-free form: " + r._lib.sc_last_error(r._h).decode()
            assert _same(again, first)
        assert first["st"]["bytes_moved"] > 4 * cfg.n * cfg.n      # (dense S alone is 4 n^2)
        r.set_debug(no_fast=1)
        for _ in range(2):
            waited = _run(torch, r, ds, dt, cfg.n, p, dev)
            assert waited["fast"] == 0 and _same(waited, first)
    finally:
        r.close()
    ref = O.register(scene.src, scene.tgt, threads=8, **cfg.params())
    assert np.array_equal(first["mask"], ref["mask"]) and first["st"]["best_rank"] == ref["best_rank"]
    assert nan_equal_bits(first["Rt"], np.concatenate([ref["R"].ravel(), ref["t"]]))


def test_host_free_call_that_fails_validation_is_repeated(pkg, O):
    """Same n and parameters, different graphs: far fewer triangles than T, outliers only, an edge count far beyond what
    the last call's launches cover.  Each is detected at sc_wait and repeated; results equal a fresh context's."""
    import torch
    dev = torch.device("cuda:0")
    n, tau, T = 2000, 0.02, 10000
    kw = dict(sigma=tau, t_cmp=0.9, tau=tau, min_len=tau, max_triangles=T, rank_mode=0)
    p = pkg.make_params(**kw)
    scenes = dict(regular=pkg.synth.make_scene(n, 0.20, 1.0, tau, 1001),
                  sparse=pkg.synth.make_scene(n, 0.01, 6.0, tau, 55),      # 6053 edges, 310 triangles: fewer than T
                  tiny=pkg.synth.make_scene(n, 0.01, 10.0, tau, 55),       # 3690 edges: below the size that is pruned at all
                  outliers=pkg.synth.make_scene(n, 0.0, 1.0, tau, 56),     # 27 481 edges, 4870 triangles
                  dense=pkg.synth.make_scene(n, 0.60, 1.0, tau, 57))       # ~9 x the edges
    r = pkg.Registrar(0)
    fresh = {}
    try:
        r.set_stream(torch.cuda.current_stream().cuda_stream)
        for nm, sc in scenes.items():
            f = pkg.Registrar(0)
            f.set_stream(torch.cuda.current_stream().cuda_stream)
            ds, dt = _dev(torch, sc, dev)
            fresh[nm] = _run(torch, f, ds, dt, n, p, dev)
            f.close()
        assert fresh["sparse"]["st"]["tri_kept"] < T and fresh["dense"]["st"]["edges"] > 3 * fresh["regular"]["st"]["edges"]
        # (scene, the fast_path state(s) the call must report; None: either)
        seq = [("regular", 0), ("regular", 1), ("sparse", 2), ("regular", 0), ("regular", 1), ("tiny", 2), ("regular", 0),
               ("regular", 1), ("outliers", 2), ("regular", 0), ("regular", 1), ("dense", 2), ("dense", 1), ("dense", 1),
               ("regular", None), ("regular", None), ("regular", 1), ("sparse", 2)]
        for nm, ef in seq:
            ds, dt = _dev(torch, scenes[nm], dev)
            got = _run(torch, r, ds, dt, n, p, dev)
            assert ef is None or got["fast"] == ef, (nm, got["fast"], ef)
            assert _same(got, fresh[nm]), nm
    finally:
        r.close()
    for nm in ("sparse", "tiny", "outliers"):
        ref = O.register(scenes[nm].src, scenes[nm].tgt, threads=8, **kw)
        assert fresh[nm]["rc"] == ref["rc"] and np.array_equal(fresh[nm]["mask"], ref["mask"])
        assert fresh[nm]["st"]["best_rank"] == ref["best_rank"] and fresh[nm]["st"]["tri_kept"] == ref["t_eff"]


def test_async_two_contexts_on_one_stream(pkg):
    """A stream of frames: two contexts bound to the same stream alternate, each call enqueued before the previous one is
    waited for.  Every frame's outputs equal the synchronous call's; at most one call may be outstanding per context."""
    import torch
    dev = torch.device("cuda:0")
    cfg, scene = pkg.synth.make_config_scene("C1")
    cfg0, scene0 = pkg.synth.make_config_scene("C0")
    p, p0 = pkg.make_params(**cfg.params()), pkg.make_params(**cfg0.params())
    ds, dt = _dev(torch, scene, dev)
    ds0, dt0 = _dev(torch, scene0, dev)
    regs = [pkg.Registrar(0), pkg.Registrar(0)]
    try:
        st = torch.cuda.current_stream().cuda_stream
        for r in regs:
            r.set_stream(st)
        base = _run(torch, regs[0], ds, dt, cfg.n, p, dev)
        base0 = _run(torch, regs[0], ds0, dt0, cfg0.n, p0, dev)
        outs = [(torch.zeros(12, dtype=torch.float32, device=dev), torch.zeros(cfg.n, dtype=torch.uint8, device=dev)) for _ in regs]
        for r, o in zip(regs, outs):                                 # warm both contexts on the big shape
            r.register_device(ds.data_ptr(), dt.data_ptr(), cfg.n, p, o[0].data_ptr(), o[1].data_ptr())
        results, n_fast = [], 0
        regs[0].register_device_async(ds.data_ptr(), dt.data_ptr(), cfg.n, p, outs[0][0].data_ptr(), outs[0][1].data_ptr())
        with pytest.raises(pkg.SacCotError):                         # a second outstanding call on the same context
            regs[0].register_device_async(ds.data_ptr(), dt.data_ptr(), cfg.n, p, outs[0][0].data_ptr(), outs[0][1].data_ptr())
        for k in range(1, 9):
            cur, prev = k & 1, (k - 1) & 1
            regs[cur].register_device_async(ds.data_ptr(), dt.data_ptr(), cfg.n, p, outs[cur][0].data_ptr(), outs[cur][1].data_ptr())
            rc, s = regs[prev].wait()
            n_fast += regs[prev].debug_last()["fast_path"] == 1
            results.append((rc, s, outs[prev][0].cpu().numpy().copy(), outs[prev][1].cpu().numpy().copy()))
        rc, s = regs[0].wait()
        results.append((rc, s, outs[0][0].cpu().numpy().copy(), outs[0][1].cpu().numpy().copy()))
        assert n_fast >= 6
        for rc, s, Rt, mask in results:
            assert rc == base["rc"] and s["best_rank"] == base["st"]["best_rank"] and s["edges"] == base["st"]["edges"]
            assert np.array_equal(mask, base["mask"]) and nan_equal_bits(Rt, base["Rt"])
        with pytest.raises(pkg.SacCotError):                         # nothing outstanding
            regs[1].wait()
        # a different shape in between is simply a waited call; the shape after it waits once, then is host-free again
        got0 = _run(torch, regs[1], ds0, dt0, cfg0.n, p0, dev)
        assert got0["fast"] == 0 and _same(got0, base0)
        a = _run(torch, regs[1], ds, dt, cfg.n, p, dev)
        b = _run(torch, regs[1], ds, dt, cfg.n, p, dev)
        assert (a["fast"], b["fast"]) == (0, 1) and _same(a, base) and _same(b, base)
    finally:
        for r in regs:
            r.close()


def test_host_free_through_the_host_entry_and_with_flags(pkg, O):
    """sc_register (host arrays) sits on the same machinery; refinement, the truncated scores and SC_FLAG_NO_DENSE_S ride
    along; a flag or parameter that differs from the last call's makes the call wait."""
    cfg, scene = pkg.synth.make_config_scene("C1")
    r = pkg.Registrar(0)
    try:
        for extra in (dict(), dict(flags=pkg.SC_FLAG_REFINE), dict(score_mode=1), dict(flags=pkg.SC_FLAG_NO_DENSE_S)):
            kw = dict(cfg.params(), **extra)
            a = r.register(scene.src, scene.tgt, **kw)
            assert r.debug_last()["fast_path"] == 0, extra
            b = r.register(scene.src, scene.tgt, **kw)
            assert r.debug_last()["fast_path"] == 1, extra
            assert np.array_equal(a["mask"], b["mask"]) and nan_equal_bits(a["R"], b["R"]) and nan_equal_bits(a["t"], b["t"])
            assert a["stats"]["best_rank"] == b["stats"]["best_rank"] and a["stats"]["best_count"] == b["stats"]["best_count"]
            ref = O.register(scene.src, scene.tgt, threads=8, score_mode=extra.get("score_mode", 0), **cfg.params())
            assert np.array_equal(b["mask"], ref["mask"]) and b["stats"]["best_rank"] == ref["best_rank"]
    finally:
        r.close()


def test_a_frame_enqueued_ahead_picks_stage_c2s_kernel_by_the_last_calls_maxima(pkg):
    """A frame enqueued while the previous one still runs (a second context on the stream) reaches the choice of stage C2's
    kernel before its own staging kernel has published the coordinate maxima: it decides by the words of the last call of
    its context — the same shape — instead of blind (blind, every such frame of C2 ran the linear filter: 53 us against 27).
    Whatever it picks, the outputs are the waited call's."""
    import torch
    dev = torch.device("cuda:0")
    cfg, scene = pkg.synth.make_config_scene("C2")
    p = pkg.make_params(**cfg.params())
    ds, dt = _dev(torch, scene, dev)
    regs = [pkg.Registrar(0), pkg.Registrar(0)]
    try:
        st = torch.cuda.current_stream().cuda_stream
        for r in regs:
            r.set_stream(st)
        base = _run(torch, regs[0], ds, dt, cfg.n, p, dev)
        assert regs[0].debug_last()["c2_kernel"] == 2              # (what the waited call chooses at C2: the Gram filter)
        outs = [(torch.zeros(12, dtype=torch.float32, device=dev), torch.zeros(cfg.n, dtype=torch.uint8, device=dev)) for _ in regs]
        for r, o in zip(regs, outs):
            for _ in range(2):
                r.register_device(ds.data_ptr(), dt.data_ptr(), cfg.n, p, o[0].data_ptr(), o[1].data_ptr())
        kernels = []
        regs[0].register_device_async(ds.data_ptr(), dt.data_ptr(), cfg.n, p, outs[0][0].data_ptr(), outs[0][1].data_ptr())
        for k in range(1, 13):
            cur, prev = k & 1, (k - 1) & 1
            regs[cur].register_device_async(ds.data_ptr(), dt.data_ptr(), cfg.n, p, outs[cur][0].data_ptr(), outs[cur][1].data_ptr())
            rc, s = regs[prev].wait()
            d = regs[prev].debug_last()
            kernels.append((d["fast_path"], d["c2_kernel"]))
            assert rc == base["rc"] and s["best_rank"] == base["st"]["best_rank"] and s["best_count"] == base["st"]["best_count"]
            assert np.array_equal(outs[prev][1].cpu().numpy(), base["mask"]) and nan_equal_bits(outs[prev][0].cpu().numpy(), base["Rt"])
        regs[0].wait()
        assert all(kd == (1, 2) for kd in kernels[1:]), kernels    # (the first frame was enqueued into an idle stream)
    finally:
        for r in regs:
            r.close()


def test_replicated_ranks_enqueue_host_free_and_finalize_in_two_halves(pkg):
    """r04c: sc_hypothesize_device with SC_FLAG_EST_BOUND enqueues a repeated shape without a host wait (stages A and B
    replicated on every rank), sc_finalize_gathered_device_async + sc_wait split the finalize step — two ranks of a job emulated
    on one GPU, two frames in flight per rank (two contexts each, one stream).  Every frame's winner, motion and mask are the
    single-GPU call's; a frame whose counts outgrow what the host-free launches covered comes back as SC_EBOUND from sc_wait and
    the repeat without the flag succeeds."""
    import torch
    dev = torch.device("cuda:0")
    n, tau, T, world, block = 2000, 0.02, 10000, 2, 500
    kw = dict(sigma=tau, t_cmp=0.9, tau=tau, min_len=tau, max_triangles=T, rank_mode=0)
    scenes = dict(regular=pkg.synth.make_scene(n, 0.20, 1.0, tau, 1001), dense=pkg.synth.make_scene(n, 0.60, 1.0, tau, 57))
    st_ = torch.cuda.current_stream().cuda_stream
    single = pkg.Registrar(0); single.set_stream(st_)
    ctx = [[pkg.Registrar(0), pkg.Registrar(0)] for _ in range(world)]          # ctx[rank][frame parity]
    try:
        dsc = {nm: _dev(torch, sc, dev) for nm, sc in scenes.items()}
        base = {nm: _run(torch, single, dsc[nm][0], dsc[nm][1], n, pkg.make_params(**kw), dev) for nm in scenes}
        for pr in ctx:
            for g in pr:
                g.set_stream(st_)
        prm = [pkg.make_params(shard_rank=r, shard_world=world, shard_block=block, flags=pkg.SC_FLAG_EST_BOUND, **kw) for r in range(world)]
        prm0 = [pkg.make_params(shard_rank=r, shard_world=world, shard_block=block, **kw) for r in range(world)]
        keys = [torch.zeros(2 * world, dtype=torch.int64, device=dev) for _ in range(2)]   # the "gathered" pairs of a frame
        outs = [[(torch.zeros(12, dtype=torch.float32, device=dev), torch.zeros(n, dtype=torch.uint8, device=dev)) for _ in range(2)] for _ in range(world)]

        def enqueue(k, nm, params):
            i = k & 1
            for r in range(world):    # (on real ranks these run side by side and an all-gather fills keys[i])
                ctx[r][i].hypothesize_device(dsc[nm][0].data_ptr(), dsc[nm][1].data_ptr(), n, params[r], keys[i].data_ptr() + 16 * r)
            for r in range(world):
                ctx[r][i].finalize_gathered_device_async(keys[i].data_ptr(), world, outs[r][i][0].data_ptr(), outs[r][i][1].data_ptr())

        def collect(k, nm):
            i = k & 1
            res = []
            for r in range(world):
                rc, s = ctx[r][i].wait()
                res.append((rc, s, ctx[r][i].debug_last()["fast_path"]))
            return res

        def check(res, nm, k):
            i = k & 1
            torch.cuda.synchronize()
            for r, (rc, s, _) in enumerate(res):
                assert rc == base[nm]["rc"] and s["best_rank"] == base[nm]["st"]["best_rank"] and s["best_count"] == base[nm]["st"]["best_count"], (nm, k, r)
                assert np.array_equal(outs[r][i][1].cpu().numpy(), base[nm]["mask"]) and nan_equal_bits(outs[r][i][0].cpu().numpy(), base[nm]["Rt"])

        fast = []
        enqueue(0, "regular", prm)
        for k in range(1, 8):
            enqueue(k, "regular", prm)
            res = collect(k - 1, "regular")
            check(res, "regular", k - 1)
            fast.append([f for _, _, f in res])
        res = collect(7, "regular"); check(res, "regular", 7)
        assert fast[0] == [0, 0] and fast[1] == [0, 0] and all(f == [1, 1] for f in fast[2:]), fast   # (a context's first call waits)
        with pytest.raises(pkg.SacCotError):                       # one call outstanding per context
            ctx[0][0].finalize_gathered_device_async(keys[0].data_ptr(), world, outs[0][0][0].data_ptr(), outs[0][0][1].data_ptr())
            ctx[0][0].finalize_gathered_device_async(keys[0].data_ptr(), world, outs[0][0][0].data_ptr(), outs[0][0][1].data_ptr())
        with pytest.raises(pkg.SacCotError):                       # ... whatever the entry
            ctx[0][0].hypothesize_device(dsc["regular"][0].data_ptr(), dsc["regular"][1].data_ptr(), n, prm[0], keys[0].data_ptr())
        with pytest.raises(pkg.SacCotError):
            ctx[0][0].register(scenes["regular"].src, scenes["regular"].tgt, **kw)
        ctx[0][0].wait()
        # ~9 x the edges under the same shape: the host-free launches do not cover it -> SC_EBOUND on every rank, then the repeat
        enqueue(0, "dense", prm)
        res = collect(0, "dense")
        assert [rc for rc, _, _ in res] == [pkg.SC_EBOUND] * world, res
        enqueue(0, "dense", prm0)
        res = collect(0, "dense"); check(res, "dense", 0)
        assert [f for _, _, f in res] == [0, 0]
    finally:
        single.close()
        for pr in ctx:
            for g in pr:
                g.close()


def test_entries_refuse_a_context_with_an_outstanding_call(pkg):
    """while a call of sc_register_device_async is outstanding EVERY entry that would start work on (or reconfigure)
    the context answers SC_EINVAL — the stage hooks, the later halves of the phase APIs, sc_set_debug, sc_set_stream — and the
    outstanding call is none the worse for it."""
    import torch
    dev = torch.device("cuda:0")
    cfg, scene = pkg.synth.make_config_scene("C1")
    ds, dt = _dev(torch, scene, dev)
    p = pkg.make_params(**cfg.params())
    r = pkg.Registrar(0)
    try:
        r.set_stream(torch.cuda.current_stream().cuda_stream)
        first = _run(torch, r, ds, dt, cfg.n, p, dev)
        d_Rt = torch.zeros(12, dtype=torch.float32, device=dev)
        d_mask = torch.full((cfg.n,), 7, dtype=torch.uint8, device=dev)
        d_hist = torch.zeros(pkg.SC_HIST_WORDS, dtype=torch.int32, device=dev)
        d_key = torch.zeros(2, dtype=torch.int64, device=dev)
        r.register_device_async(ds.data_ptr(), dt.data_ptr(), cfg.n, p, d_Rt.data_ptr(), d_mask.data_ptr())   # host-free: outstanding
        tries = {
            "sc_compat_host": lambda: r.compat(scene.src, scene.tgt, p, want_S=False),
            "sc_triangles_host": lambda: r.triangles(scene.src, scene.tgt, p),
            "sc_mask_host": lambda: r.mask(scene.src, scene.tgt, p, np.zeros(12, np.float32)),
            "sc_hypothesize_end_device": lambda: r.hypothesize_end_device(d_hist.data_ptr(), d_key.data_ptr()),
            "sc_hypothesize_device": lambda: r.hypothesize_device(ds.data_ptr(), dt.data_ptr(), cfg.n, p, d_key.data_ptr()),
            "sc_shard_edges_device": lambda: r.shard_edges_device(d_hist.data_ptr()),
            "sc_shard_select_device": lambda: r.shard_select_device(d_hist.data_ptr(), d_hist.data_ptr()),
            "sc_shard_score_device": lambda: r.shard_score_device(d_hist.data_ptr(), d_key.data_ptr()),
            "sc_set_debug": lambda: r.set_debug(no_fast=1),
            "sc_set_stream": lambda: r.set_stream(None),
            "sc_register_device_async": lambda: r.register_device_async(ds.data_ptr(), dt.data_ptr(), cfg.n, p, d_Rt.data_ptr(), d_mask.data_ptr()),
        }
        for name, call in tries.items():
            with pytest.raises(pkg.SacCotError) as ei:
                call()
            assert ei.value.status == pkg.SC_EINVAL, (name, ei.value.status)
        rc, st = r.wait()
        torch.cuda.synchronize()
        got = dict(rc=rc, st=st, Rt=d_Rt.cpu().numpy(), mask=d_mask.cpu().numpy())
        assert r.debug_last()["fast_path"] == 1 and _same(got, first)
        again = _run(torch, r, ds, dt, cfg.n, p, dev)   # and the context is as usable as before
        assert _same(again, first)
    finally:
        r.close()


def test_a_host_free_enqueue_that_does_not_fit_the_workspace_cap_runs_the_waited_way(pkg):
    """the allocations of a host-free call are sized by its COVERS; under a workspace cap that only just holds the waited
    call they fail — the call then runs the waited way and returns what that returns (SC_OK), not SC_ENOMEM."""
    import torch
    dev = torch.device("cuda:0")
    cfg, scenes = pkg.synth.make_stream_scenes("C1", 3)
    p = pkg.make_params(**cfg.params())
    r = pkg.Registrar(0)
    try:
        r.set_stream(torch.cuda.current_stream().cuda_stream)
        outs = []
        for s in scenes:   # waited / host-free under the default cap: what the context holds afterwards is what these calls need
            ds, dt = _dev(torch, s, dev)
            outs.append(_run(torch, r, ds, dt, cfg.n, p, dev))
        held = outs[-1]["st"]["workspace_bytes"]
        big = pkg.synth.make_scene(cfg.n, cfg.rho * 1.6, cfg.L, cfg.tau, cfg.seed + 77)   # more edges than anything seen so far
        ds, dt = _dev(torch, big, dev)
        tight = pkg.make_params(max_workspace=held + (1 << 20), **cfg.params())
        got = _run(torch, r, ds, dt, cfg.n, tight, dev)
        loose = pkg.Registrar(0)
        try:
            loose.set_stream(torch.cuda.current_stream().cuda_stream)
            ref = _run(torch, loose, ds, dt, cfg.n, p, dev)
        finally:
            loose.close()
        assert got["rc"] == 0 and _same(got, ref), (got["rc"], got["st"], ref["st"])
    finally:
        r.close()
