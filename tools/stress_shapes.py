#!/usr/bin/env python3
"""r05 development: two contexts on one stream, random streams of frames over several SHAPES and scenes through
sc_register_device_async / sc_wait only; every result against the first result of its (shape, scene); at the first wrong one the
recent history of both contexts is printed (what ran before it, how it was enqueued, what the launches covered).
python tools/stress_shapes.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as ge

pkg = ge.load_package()
if os.environ.get("SC_SOAK_LIB"):
    pkg.api.LIB_PATH = os.environ["SC_SOAK_LIB"]
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
shapes = []
for name, T, count in (("C2", 50000, 16), ("C2", 200000, 4), ("C1", 10000, 16), ("C1", 3000, 4), ("C0", 200, 4)):
    cfg, scs = pkg.synth.make_stream_scenes(name, count)
    kw = cfg.params(); kw["max_triangles"] = T
    shapes.append((f"{name}/T={T}", cfg.n, kw, [(torch.from_numpy(x.src).to(dev), torch.from_numpy(x.tgt).to(dev)) for x in scs]))
rng = np.random.default_rng(int(os.environ.get("SEED", "0")))
first, hist = {}, [[], []]
calls = bad = 0
t0 = time.time()
with torch.cuda.stream(stream):
    pair = [pkg.Registrar(0), pkg.Registrar(0)]
    for g in pair:
        g.set_stream(stream.cuda_stream)
    ctr = [dict(n_fast_ok=0, n_fast_repeat=0, n_est_fail=0, n_frames=0) for _ in pair]
    while time.time() - t0 < budget and not bad:
        sname, n, kw, devs = shapes[int(rng.integers(len(shapes)))]
        nb, k0 = int(rng.integers(1, 12)), int(rng.integers(len(devs)))
        pb = pkg.make_params(**kw)
        fRt = torch.zeros(nb, 12, dtype=torch.float32, device=dev)
        fmask = torch.zeros(nb, n, dtype=torch.uint8, device=dev)
        sts = []
        try:
            for k in range(nb + 1):
                if k < nb:
                    a, b = devs[(k0 + k) % len(devs)]
                    pair[k & 1].register_device_async(a.data_ptr(), b.data_ptr(), n, pb, fRt[k].data_ptr(), fmask[k].data_ptr())
                    hist[k & 1].append([sname, (k0 + k) % len(devs), None])
                if k >= 1:
                    sts.append(pair[(k - 1) & 1].wait())
        except Exception as ex:
            print("EXCEPTION", ex, "in", sname, "frame", k, flush=True)
            bad += 1
        stream.synchronize()
        hRt, hmask = fRt.cpu().numpy(), fmask.cpu().numpy()
        for k, (rc, st) in enumerate(sts):
            key = f"{sname}/scene{(k0 + k) % len(devs)}"
            sig = (rc, st["edges"], st["tri_kept"], st["best_rank"], st["best_count"], hRt[k].tobytes(), hmask[k].tobytes())
            # (hist entries are appended at enqueue time: entry of frame k of this burst on context k & 1)
            if key not in first:
                first[key] = sig
            elif first[key] != sig:
                bad += 1
                print("MISMATCH", key, "frame", k, "of", nb, sig[:5], "vs", first[key][:5], "tri_total", st["tri_total"], flush=True)
            calls += 1
        if bad:
            for i, g in enumerate(pair):
                try:
                    d = g.debug_last()
                except Exception as ex:
                    d = str(ex)
                print(f"context {i}: last calls (shape, scene): {hist[i][-8:]}\n   debug_last: {d}\n   last_error: {g._lib.sc_last_error(g._h).decode()}", flush=True)
print(f"stress: {calls} calls, {bad} bad, {time.time() - t0:.0f} s; " + " | ".join(str({k: g.debug_last()[k] for k in ('n_frames', 'n_fast_ok', 'n_fast_repeat', 'n_est_ok', 'n_est_fail', 'cover_edges', 'cover_triangles')}) for g in pair if not bad))
sys.exit(1 if bad else 0)
