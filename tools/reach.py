"""Experiment (r04): how much of stage C2 a triangle-inequality cut against a REFERENCE hypothesis would skip.
For every ranked hypothesis h: dM = R0^T R_h - I, tau_h = R0^T (t_h - t0); reach_h = ||dM||_2 Pmax + |tau_h|;
a correspondence with |V_i| = |R0^T (q_i - t0) - p_i| > reach_h + tau cannot be an inlier of h."""
import sys, json
import numpy as np
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import __graft_entry__ as ge
pkg = ge.load_package()

def run(name):
    cfg, sc = pkg.synth.make_config_scene(name)
    prm = pkg.make_params(**cfg.params())
    r = pkg.Registrar(0)
    tri, key, total, edges = r.triangles(sc.src, sc.tgt, prm)
    Rt = r.kabsch(sc.src, sc.tgt, prm, tri).astype(np.float64)
    out = r.register(sc.src, sc.tgt, **cfg.params())
    r.close()
    T = Rt.shape[0]; n = cfg.n
    p = sc.src.astype(np.float64); q = sc.tgt.astype(np.float64)
    cP = 0.5 * (p.max(0) + p.min(0)); P = p - cP
    Pmax = np.linalg.norm(P, axis=1).max()
    res = {"config": name, "T": T, "n": n, "tau": cfg.tau, "best_rank": int(out["stats"]["best_rank"]), "best_count": int(out["stats"]["best_count"])}
    for refname, ref in (("rank0", 0), ("winner", int(out["stats"]["best_rank"]))):
        R0 = Rt[ref, :9].reshape(3, 3); t0 = Rt[ref, 9:]
        u, s, vt = np.linalg.svd(R0); R0 = u @ vt
        Rh = Rt[:, :9].reshape(T, 3, 3)
        M = np.einsum("ji,hjk->hik", R0, Rh)            # R0^T R_h
        dM = M - np.eye(3)
        nrm = np.linalg.norm(dM, ord=2, axis=(1, 2))
        th = (Rt[:, 9:] - t0) @ R0                       # R0^T (t_h - t0)
        # with the common centre cP: residual = dM P + (tau_h + dM cP) - V
        th_c = th + np.einsum("hij,j->hi", dM, cP)
        reach = nrm * Pmax + np.linalg.norm(th_c, axis=1)
        V = (q - t0) @ R0 - p
        Vn = np.sort(np.linalg.norm(V, axis=1))
        lim = reach + cfg.tau * 1.01
        cnt = np.searchsorted(Vn, lim, side="right")     # correspondences a hypothesis must look at
        # wave granularity: 32 consecutive hypotheses share the largest reach; unit granularity 256 correspondences
        Tw = (T // 32) * 32
        wl = lim[:Tw].reshape(-1, 32).max(1)
        wc = np.ceil(np.searchsorted(Vn, wl, side="right") / 256.0) * 256
        # workgroup granularity: 256 consecutive hypotheses
        Tg = (T // 256) * 256
        gl = lim[:Tg].reshape(-1, 256).max(1)
        gc = np.ceil(np.searchsorted(Vn, gl, side="right") / 256.0) * 256
        res[refname] = {
            "reach_over_tau_pct": [float(np.percentile(reach / cfg.tau, x)) for x in (10, 50, 90, 99)],
            "dM_norm_pct": [float(np.percentile(nrm, x)) for x in (10, 50, 90, 99)],
            "V_near_frac(|V|<=8tau)": float((Vn <= 8 * cfg.tau).mean()),
            "work_frac_per_hyp": float(cnt.sum() / (T * n)),
            "work_frac_per_wave32_unit256": float(wc.sum() * 32 / (Tw * n)),
            "work_frac_per_group256_unit256": float(gc.sum() * 256 / (Tg * n)),
            "hyps_seeing_all": float((cnt >= n).mean()),
        }
    return res

for name in sys.argv[1:]:
    print(json.dumps(run(name)), flush=True)
