#!/usr/bin/env python3
"""Soak: one context (two for the bursts), thousands of calls over alternating problem sizes and eight forms of the call (r04c: rings of overlapping frames; r05: streams of DISTINCT frames of one shape) — the two phase-1 forms of
the phase API (certified pruning bound, host waits in the middle) and sc_register_device (r04: estimated bound, fused edge
kernel, host-free enqueue of a repeated shape); every result must be byte-identical to the first one of its configuration
(tickets, polled read-backs, speculative launches and the validate-and-repeat paths are exercised on buffers left over from
other sizes); r04c: BURSTS of 2 - 6 frames of one configuration as a stream — frame k + 1 enqueued before frame k is waited for, two
contexts on the stream — through sc_register_device_async / sc_wait and through host-free sc_hypothesize_device +
sc_finalize_gathered_device_async / sc_wait.   python tools/soak.py [seconds] [--big]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as ge

pkg = ge.load_package()
if os.environ.get("SC_SOAK_LIB"):   # (A/B of library builds: tools/r5)
    pkg.api.LIB_PATH = os.environ["SC_SOAK_LIB"]
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
big = "--big" in sys.argv   # also C3 and C4 (the Gram filter's cut at its largest shapes; ~1 ms per call)
dev = torch.device("cuda", 0)
reg = pkg.Registrar(0)
regB = pkg.Registrar(0)
stream = torch.cuda.Stream(device=dev)
cases = []
for name, T in (("C0", 200), ("C1", 10000), ("C2", 50000), ("C1", 3000), ("C2", 200000)) + ((("C3", 200000), ("C4", 500000)) if big else ()):
    cfg, sc = pkg.synth.make_config_scene(name)
    kw = cfg.params(); kw["max_triangles"] = T
    cases.append((f"{name}/T={T}", cfg.n, kw, torch.from_numpy(sc.src).to(dev), torch.from_numpy(sc.tgt).to(dev)))
# form 7 (r05): streams of DISTINCT frames — families of 16 scenes of one shape whose counts differ from frame to frame (synth.make_stream_scenes)
families = []
for name, T in (("C1", 10000), ("C2", 50000)):
    cfg, scs = pkg.synth.make_stream_scenes(name, 16)
    kw = cfg.params(); kw["max_triangles"] = T
    families.append((f"{name}/T={T}", cfg.n, kw, [(torch.from_numpy(x.src).to(dev), torch.from_numpy(x.tgt).to(dev)) for x in scs]))
distinct = [0, 0]   # streams of distinct frames, frames in them
first = {}
forms = [0, 0, 0]   # sc_register_device calls by sc_debug_last.fast_path: waited / host-free / host-free then repeated
calls = mism = 0
t0 = time.time()
rng = np.random.default_rng(0)
with torch.cuda.stream(stream):
    reg.set_stream(stream.cuda_stream)
    regB.set_stream(stream.cuda_stream)
    pair = [reg, regB]
    bursts = [0, 0]
    d_keys2 = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(2)]
    d_Rt2 = [torch.zeros(12, dtype=torch.float32, device=dev) for _ in range(2)]
    d_key = torch.zeros(2, dtype=torch.int64, device=dev)
    d_hist = torch.zeros(pkg.SC_HIST_WORDS, dtype=torch.int32, device=dev)
    d_Rt = torch.zeros(12, dtype=torch.float32, device=dev)
    # form 6 (r04c): frames OVERLAPPING on the GPU — a ring of three contexts, each on a stream of its own
    ring = [pkg.Registrar(0) for _ in range(3)]
    ring_streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
    for g_, s_ in zip(ring, ring_streams):
        g_.set_stream(s_.cuda_stream)
    ring_Rt = [torch.zeros(12, dtype=torch.float32, device=dev) for _ in range(3)]
    overlapped = 0
    while time.time() - t0 < budget:
        name, n, kw, s, t = cases[int(rng.integers(len(cases)))]
        form = int(rng.integers(8))
        if form == 7:
            # a stream of 4 - 24 DISTINCT frames of one shape through the two contexts of the stream (what bench.py times): every frame
            # must equal the first result of ITS scene — whatever the frames before it left in the contexts' covers and buffers
            fam, n, kw, devs = families[int(rng.integers(len(families)))]
            nb, k0 = int(rng.integers(4, 25)), int(rng.integers(16))
            pb = pkg.make_params(**kw)
            fRt = torch.zeros(nb, 12, dtype=torch.float32, device=dev)
            fmask = torch.zeros(nb, n, dtype=torch.uint8, device=dev)
            sts = []
            for k in range(nb + 1):
                if k < nb:
                    a, b = devs[(k0 + k) % 16]
                    pair[k & 1].register_device_async(a.data_ptr(), b.data_ptr(), n, pb, fRt[k].data_ptr(), fmask[k].data_ptr())
                if k >= 1:
                    sts.append(pair[(k - 1) & 1].wait())
                    if sts[-1][0] != 0:   # (none of these scenes lacks a hypothesis: say what the context knows, while it knows it)
                        g_ = pair[(k - 1) & 1]
                        print("UNEXPECTED STATUS", sts[-1][0], fam, "scene", (k0 + k - 1) % 16, "frame", k - 1, sts[-1][1], g_.debug_last(),
                              g_._lib.sc_last_error(g_._h).decode(), flush=True)
            stream.synchronize()
            hRt, hmask = fRt.cpu().numpy(), fmask.cpu().numpy()
            for k, (rc, st) in enumerate(sts):
                key = f"{fam}/scene{(k0 + k) % 16}"
                sig = (rc, st["edges"], st["tri_kept"], st["best_rank"], st["best_count"], hRt[k].tobytes(), hmask[k].tobytes())
                if key not in first:
                    first[key] = sig
                elif first[key] != sig:
                    mism += 1
                    print("MISMATCH", key, "stream of distinct frames", k, sig[:5], "vs", first[key][:5], flush=True)
                calls += 1
            distinct[0] += 1; distinct[1] += nb
            continue
        if form == 6:
            stream.synchronize()   # (the inputs were uploaded on `stream`; the ring's streams do not wait for it)
            nb = int(rng.integers(3, 10))
            pb = pkg.make_params(**kw)
            rmask = [torch.zeros(n, dtype=torch.uint8, device=dev) for _ in range(3)]
            for k in range(nb + 2):
                if k < nb:
                    i = k % 3
                    ring[i].register_device_async(s.data_ptr(), t.data_ptr(), n, pb, ring_Rt[i].data_ptr(), rmask[i].data_ptr())
                if k >= 2:
                    i = (k - 2) % 3
                    rc, st = ring[i].wait()
                    ring_streams[i].synchronize()
                    sig = (rc, st["edges"], st["tri_kept"], st["best_rank"], st["best_count"],
                           ring_Rt[i].cpu().numpy().tobytes(), rmask[i].cpu().numpy().tobytes())
                    if name not in first:
                        first[name] = sig
                    elif first[name] != sig:
                        mism += 1
                        print("MISMATCH", name, "overlapped frames", k - 2, sig[:5], "vs", first[name][:5], flush=True)
                    calls += 1
            overlapped += 1
            continue   # 0 hypothesize + finalize, 1 its split-sample form, 2 sc_register_device, 3 form 0 with SC_FLAG_EST_BOUND (r04b), 4 / 5 bursts (r04c)
        if form >= 4:
            # a burst of frames as a stream: 4 = sc_register_device_async, 5 = hypothesize (estimated bound: host-free from its
            # context's second frame on) + finalize in two halves; every frame must equal the configuration's first result
            nb = int(rng.integers(2, 7))
            pb = pkg.make_params(flags=pkg.SC_FLAG_EST_BOUND if form == 5 else 0, **kw)
            masks = [torch.zeros(n, dtype=torch.uint8, device=dev) for _ in range(2)]

            def enq(k):
                i = k & 1
                if form == 4:
                    pair[i].register_device_async(s.data_ptr(), t.data_ptr(), n, pb, d_Rt2[i].data_ptr(), masks[i].data_ptr())
                else:
                    pair[i].hypothesize_device(s.data_ptr(), t.data_ptr(), n, pb, d_keys2[i].data_ptr())
                    pair[i].finalize_gathered_device_async(d_keys2[i].data_ptr(), 1, d_Rt2[i].data_ptr(), masks[i].data_ptr())
            enq(0)
            for k in range(1, nb + 1):
                if k < nb:
                    enq(k)
                i = (k - 1) & 1
                rc, st = pair[i].wait()
                if rc == pkg.SC_EBOUND:   # (form 5 only: the caller's repeat, the certifying way)
                    p0 = pkg.make_params(**kw)
                    pair[i].hypothesize_device(s.data_ptr(), t.data_ptr(), n, p0, d_keys2[i].data_ptr())
                    rc, st = pair[i].finalize_gathered_device(d_keys2[i].data_ptr(), 1, d_Rt2[i].data_ptr(), masks[i].data_ptr())
                    bursts[1] += 1
                # (the copies below run on this stream, behind frame k: frame k - 1's outputs are complete, frame k + 1 reuses them)
                sig = (rc, st["edges"], st["tri_kept"], st["best_rank"], st["best_count"],
                       d_Rt2[i].cpu().numpy().tobytes(), masks[i].cpu().numpy().tobytes())
                if name not in first:
                    first[name] = sig
                elif first[name] != sig:
                    mism += 1
                    print("MISMATCH", name, ("burst register", "burst hypothesize")[form - 4], k - 1, sig[:5], "vs", first[name][:5], flush=True)
                calls += 1
            bursts[0] += 1
            continue
        split = form == 1
        p = pkg.make_params(flags=pkg.SC_FLAG_EST_BOUND if form == 3 else 0, **kw)
        d_mask = torch.zeros(n, dtype=torch.uint8, device=dev)
        if form == 2:
            rc, st = reg.register_device(s.data_ptr(), t.data_ptr(), n, p, d_Rt.data_ptr(), d_mask.data_ptr())
            forms[reg.debug_last()["fast_path"]] += 1
        else:
            if split:
                reg.hypothesize_begin_device(s.data_ptr(), t.data_ptr(), n, p, d_hist.data_ptr())
                reg.hypothesize_end_device(d_hist.data_ptr(), d_key.data_ptr())
            else:
                reg.hypothesize_device(s.data_ptr(), t.data_ptr(), n, p, d_key.data_ptr())
            rc, st = reg.finalize_device(d_key.data_ptr(), d_Rt.data_ptr(), d_mask.data_ptr())
        stream.synchronize()
        # (tri_total — the triangles ENUMERATED — depends on the pruning bound, certified or estimated: not part of the result)
        sig = (rc, st["edges"], st["tri_kept"], st["best_rank"], st["best_count"],
               d_Rt.cpu().numpy().tobytes(), d_mask.cpu().numpy().tobytes())
        if name not in first:
            first[name] = sig
        elif first[name] != sig:
            mism += 1
            print("MISMATCH", name, ("plain", "split", "register", "plain + estimated bound")[form], sig[:5], "vs", first[name][:5], flush=True)
        calls += 1
        if calls % 2000 == 0:
            print(f"{calls} calls, {mism} mismatches, {time.time() - t0:.0f} s", flush=True)
print(f"soak: {calls} calls over {len(first)} configurations in {time.time() - t0:.0f} s, {mism} mismatches; "
      f"sc_register_device calls: {forms[0]} waited, {forms[1]} host-free, {forms[2]} host-free and repeated; {bursts[0]} bursts of streamed frames "
      f"({bursts[1]} frames came back SC_EBOUND and were repeated); {overlapped} rings of frames overlapping on three streams; "
      f"{distinct[0]} streams of DISTINCT frames ({distinct[1]} frames; the two contexts of the stream: "
      + ", ".join(f"{k_} {sum(g_.debug_last()[k_] for g_ in pair)}" for k_ in ("n_frames", "n_fast_ok", "n_fast_repeat", "n_est_ok", "n_est_fail")) + ")")
sys.exit(1 if mism else 0)
