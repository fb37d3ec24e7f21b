#!/usr/bin/env python3
"""Per-rank step time of an N-GPU job, emulated on ONE GPU (no multi-GPU box is available to the builder).

    python tools/emulate_world.py [--config C3] [--scaling strong|weak] [--mode shard_ab|replicated] [--nodense] 1 2 4 8

shard_ab   (what bench.py runs for N > 1): every rank is a context of its own; the ranks share the
           exchange buffers.  One untimed pass of ALL ranks fills them; then each rank's phases are timed alone, K times,
           with everything the other ranks contribute already in place.  A device copy of the received bytes stands in
           for each collective (bit rows, candidate blobs) and a 1 KiB copy for the histogram all-reduce — a copy inside
           one HBM is NOT an xGMI transfer: the table lists the bytes per collective so the real cost can be priced
           (7 links x ~153 GB/s per GPU, direct all-gather).  The step of the job is bounded by the SLOWEST rank: max and
           mean over the ranks are printed.
replicated (round 1's form): A and B on every rank, only stage C sharded.
strong scaling: T_total = the config's T (BASELINE configs[3] / [4] as specified); weak: T per GPU fixed."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as ge

ap = argparse.ArgumentParser()
ap.add_argument("worlds", nargs="*", type=int, default=[1, 2, 4, 8])
ap.add_argument("--config", default="C3")
ap.add_argument("--scaling", choices=("strong", "weak"), default="strong")
ap.add_argument("--mode", choices=("shard_ab", "replicated"), default="shard_ab")
ap.add_argument("--nodense", action="store_true")
ap.add_argument("--streamed", action="store_true",
                help="replicated mode: a rank's frames as a STREAM (r04c: host-free sc_hypothesize_device, sc_finalize_gathered_device_async, "
                     "frame k + 1 enqueued before frame k is waited for — two contexts on one stream), as bench.py times them")
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--certified", action="store_true",
                help="shard_ab without SC_FLAG_EST_BOUND: certifying sample shared by the ranks + the 1 KiB histogram all-reduce "
                     "(r03's form: four collectives per step instead of three)")
ap.add_argument("--estimate", action="store_true", help="SC_FLAG_EST_BOUND also on graphs of 8192 correspondences and more (the host layers do not)")
ap.add_argument("--no-latency", action="store_true", help="skip the measurement of a collective's latency (below)")
args = ap.parse_args()

pkg = ge.load_package()
cfg, scene = pkg.synth.make_config_scene(args.config)
dev = torch.device("cuda", 0)
d_src = torch.from_numpy(scene.src).to(dev); d_tgt = torch.from_numpy(scene.tgt).to(dev)
flags = pkg.SC_FLAG_NO_DENSE_S if args.nodense else 0
est = 0 if (args.certified or (cfg.n >= 8192 and not args.estimate)) else pkg.SC_FLAG_EST_BOUND   # (the host layers' rule: shard.py)
K = args.iters

# What ONE collective costs before any byte moves (VERDICT r03 #6c): RCCL on a ONE-rank communicator of this GPU — the enqueue,
# RCCL's own kernel and its completion on the stream, no link involved — timed as the stream time K back-to-back calls add.
# A real N-rank collective cannot be cheaper than this; the table adds (collectives per step) x this figure in a column of its
# own.  The transfer itself is priced from the bytes (7 links x ~153 GB/s) in DESIGN.md, not here.
coll_us = {}
if not args.no_latency:
    import torch.distributed as dist
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    small = torch.zeros(2, dtype=torch.int64, device=dev); big = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
    h = torch.zeros(256, dtype=torch.int32, device=dev)
    def lat(fn, n=200):
        for _ in range(20): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
    coll_us = {"all_gather 16 B": lat(lambda: dist.all_gather_into_tensor(small, small)),
               "all_gather 8 MB": lat(lambda: dist.all_gather_into_tensor(big, big)),
               "all_reduce 1 KiB": lat(lambda: dist.all_reduce(h))}
    print("# one collective on a ONE-rank RCCL communicator of this GPU (enqueue + RCCL kernel, no link): " +
          ", ".join(f"{k} {v:.1f} us" for k, v in coll_us.items()))
    dist.destroy_process_group()
print(f"# {args.config}: N={cfg.n} T={cfg.T} scaling={args.scaling} mode={args.mode} dense_S={not args.nodense}")
base = None
for world in args.worlds:
    kw = cfg.params()
    kw["max_triangles"] = cfg.T if args.scaling == "strong" else cfg.T * world
    T_total = kw["max_triangles"]
    block = max(64, min(1024, T_total // (world * 8)))
    ps = [pkg.make_params(shard_rank=r, shard_world=world, shard_block=block, flags=flags, **kw) for r in range(world)]
    d_Rt = torch.zeros(12, dtype=torch.float32, device=dev); d_mask = torch.zeros(cfg.n, dtype=torch.uint8, device=dev)
    d_keys = torch.zeros(2 * world, dtype=torch.int64, device=dev)
    if args.mode == "replicated" or world == 1:
        reg = pkg.Registrar(0); reg.set_stream(torch.cuda.current_stream().cuda_stream)
        # (replicated, world > 1: stage B pruned by the estimated bound, as bench.py's replicated form does — SC_FLAG_EST_BOUND on
        # sc_hypothesize_device; --certified keeps the certifying sample)
        if world > 1 and not args.certified:
            ps = [pkg.make_params(shard_rank=r, shard_world=world, shard_block=block, flags=flags | pkg.SC_FLAG_EST_BOUND, **kw) for r in range(world)]
        def step(r):
            reg.hypothesize_device(d_src.data_ptr(), d_tgt.data_ptr(), cfg.n, ps[r], d_keys.data_ptr() + 16 * r)
        for r in range(world):
            step(r)
        torch.cuda.synchronize()
        times = []
        for r in range(world if world <= 2 else 2):  # ranks are statistically alike here: two suffice
            for _ in range(2):
                step(r); reg.finalize_gathered_device(d_keys.data_ptr(), world, d_Rt.data_ptr(), d_mask.data_ptr())
            if args.streamed:
                regB = pkg.Registrar(0); regB.set_stream(torch.cuda.current_stream().cuda_stream)
                pair = [reg, regB]
                outs = [(d_Rt, d_mask), (torch.zeros(12, dtype=torch.float32, device=dev), torch.zeros(cfg.n, dtype=torch.uint8, device=dev))]
                for _ in range(3):
                    regB.hypothesize_device(d_src.data_ptr(), d_tgt.data_ptr(), cfg.n, ps[r], d_keys.data_ptr() + 16 * r)
                    regB.finalize_gathered_device(d_keys.data_ptr(), world, outs[1][0].data_ptr(), outs[1][1].data_ptr())

                def enqueue(k):
                    pair[k & 1].hypothesize_device(d_src.data_ptr(), d_tgt.data_ptr(), cfg.n, ps[r], d_keys.data_ptr() + 16 * r)
                    pair[k & 1].finalize_gathered_device_async(d_keys.data_ptr(), world, outs[k & 1][0].data_ptr(), outs[k & 1][1].data_ptr())
                KS = 4 * K
                torch.cuda.synchronize(); t0 = time.perf_counter()
                enqueue(0)
                for k in range(1, KS + 1):
                    if k < KS:
                        enqueue(k)
                    rc, st = pair[(k - 1) & 1].wait()
                torch.cuda.synchronize(); times.append((time.perf_counter() - t0) / KS)
                regB.close()
                continue
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(K):
                step(r); rc, st = reg.finalize_gathered_device(d_keys.data_ptr(), world, d_Rt.data_ptr(), d_mask.data_ptr())
            torch.cuda.synchronize(); times.append((time.perf_counter() - t0) / K)
        coll = "key pairs 16 B/rank" + (" (streamed)" if args.streamed else "")
        reg.close()
    else:
        regs = [pkg.Registrar(0) for _ in range(world)]
        for g in regs:
            g.set_stream(torch.cuda.current_stream().cuda_stream)
        level = 0
        while True:   # fill pass (repeated with bigger candidate blobs while the library answers SC_ERETRY)
            ps = [pkg.make_params(shard_rank=r, shard_world=world, shard_block=block, flags=flags | est, shard_cand_level=level, **kw)
                  for r in range(world)]
            plan = pkg.shard_plan(ps[0], cfg.n)
            d_bits = torch.zeros(plan.bits_bytes_total // 8, dtype=torch.int64, device=dev)
            d_bits_rx = torch.zeros_like(d_bits)
            d_hist = [torch.zeros(pkg.SC_HIST_WORDS, dtype=torch.int32, device=dev) for _ in range(world)]
            d_cand = torch.zeros(world * plan.cand_bytes_per_rank // 8, dtype=torch.int64, device=dev)
            d_cand_rx = torch.zeros_like(d_cand)
            # every rank, phase by phase (what the collectives would have delivered ends up in the shared buffers)
            for r in range(world):
                regs[r].shard_compat_device(d_src.data_ptr(), d_tgt.data_ptr(), cfg.n, ps[r], d_bits.data_ptr())
            for r in range(world):
                regs[r].shard_edges_device(d_hist[r].data_ptr())
            torch.cuda.synchronize()
            summed = torch.from_numpy((sum(h.cpu().numpy().view(np.uint32).astype(np.uint64) for h in d_hist)
                                       & np.uint64(0xFFFFFFFF)).astype(np.uint32).view(np.int32)).to(dev)
            for r in range(world):
                if not est:
                    d_hist[r].copy_(summed)
                regs[r].shard_select_device(d_hist[r].data_ptr(), d_cand.data_ptr() + r * plan.cand_bytes_per_rank)
            for r in range(world):
                regs[r].shard_score_device(d_cand.data_ptr(), d_keys.data_ptr() + 16 * r)
            rcs = [regs[r].finalize_gathered_device(d_keys.data_ptr(), world, d_Rt.data_ptr(), d_mask.data_ptr())[0]
                   for r in range(world)]
            torch.cuda.synchronize()
            if est and all(rc == pkg.SC_EBOUND for rc in rcs):   # the estimate failed on this input: the certified form
                est = 0
                continue
            if all(rc != pkg.SC_ERETRY for rc in rcs):
                break
            assert all(rc == pkg.SC_ERETRY for rc in rcs)
            level += 1
        hdr = d_cand.cpu().numpy().view(np.uint64).reshape(world, -1)[:, :2].copy()
        cap = (plan.cand_bytes_per_rank - 256) // 20
        rx_bits = (world - 1) * plan.bits_bytes_per_rank // 8       # int64 words a rank receives
        rx_cand_words = plan.cand_bytes_per_rank // 8

        def step(r):
            g = regs[r]
            g.shard_compat_device(d_src.data_ptr(), d_tgt.data_ptr(), cfg.n, ps[r], d_bits.data_ptr())
            d_bits_rx[:rx_bits].copy_(d_bits[:rx_bits])                                  # stand-in: all-gather of the bit rows
            g.shard_edges_device(d_hist[r].data_ptr())
            if not est:
                d_hist[r].copy_(summed)                                                 # stand-in: 1 KiB all-reduce
            g.shard_select_device(d_hist[r].data_ptr(), d_cand.data_ptr() + r * plan.cand_bytes_per_rank)
            for q in range(world):                                                      # stand-in: all-gather of the blobs
                if q != r:                                                              # (header + entries sent)
                    nw = 32 + (int(hdr[q, 1]) * 20 + 7) // 8
                    d_cand_rx[q * rx_cand_words: q * rx_cand_words + nw].copy_(d_cand[q * rx_cand_words: q * rx_cand_words + nw])
            g.shard_score_device(d_cand.data_ptr(), d_keys.data_ptr() + 16 * r)
            return g.finalize_gathered_device(d_keys.data_ptr(), world, d_Rt.data_ptr(), d_mask.data_ptr())
        times = []
        for r in range(world):
            for _ in range(2):
                step(r)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(K):
                rc, st = step(r)
            torch.cuda.synchronize(); times.append((time.perf_counter() - t0) / K)
        sent = hdr[:, 1].astype(np.int64)
        coll = (f"{'estimated bound: 3 collectives' if est else 'certified bound: 4 collectives'}; bit rows {plan.bits_bytes_per_rank/1e6:.2f} MB/rank (rx {(world-1)*plan.bits_bytes_per_rank/1e6:.1f} MB), {'no histogram exchange' if est else 'hist 1 KiB'}, "
                f"candidates sent {sent.min()}..{sent.max()} of cap {cap} at level {level} (x20 B; blob {plan.cand_bytes_per_rank/1e6:.2f} MB), key pairs 16 B; "
                f"enumerated per rank {hdr[:,0].min()}..{hdr[:,0].max()}")
        for g in regs:
            g.close()
    tmax, tmean = max(times), sum(times) / len(times)
    base = base or tmax
    ncoll = 0 if world == 1 else (1 if (args.mode == "replicated") else (3 if est else 4))
    extra = ""
    if coll_us and ncoll:
        add = (coll_us["all_gather 16 B"] + ((coll_us["all_gather 8 MB"] * 2 + (0 if est else coll_us["all_reduce 1 KiB"])) if ncoll > 1 else 0)) * 1e-6
        extra = f" | + {ncoll} collective latencies (measured, single-rank RCCL): {(tmax + add)*1e3:.3f} ms"
    print(f"world={world} T_total={T_total} per-rank step: max {tmax*1e3:.3f} ms mean {tmean*1e3:.3f} ms -> job {T_total/tmax/1e6:.1f} M hyp/s, "
          f"speed-up vs the first row {base/tmax:.2f}x | winner rank={st['best_rank']} inliers={st['best_count']} | {coll}"
          f"{extra} | per rank ms: {' '.join(f'{t*1e3:.3f}' for t in times)}", flush=True)
