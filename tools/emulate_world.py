#!/usr/bin/env python3
"""Per-rank step time at world = W on ONE GPU: stage A + B for T_total = 50k * W (replicated on every rank) and stage C
for rank 0's share, no collective.  What bench.py --gpus W does per rank, minus the all-reduces.  From world 4 on the
pruning sample is sharded as in bench.py: rank 0 samples its share and a device copy of the precomputed summed
histogram (1 KiB) stands in for the all-reduce.  SPLIT=0 / SPLIT=1 in the environment forces either form."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
cfg, scene = pkg.synth.make_config_scene("C2")
dev = torch.device("cuda", 0)
reg = pkg.Registrar(0)
reg.set_stream(torch.cuda.current_stream().cuda_stream)
d_src = torch.from_numpy(scene.src).to(dev); d_tgt = torch.from_numpy(scene.tgt).to(dev)
d_key = torch.zeros(2, dtype=torch.int64, device=dev)
d_Rt = torch.zeros(12, dtype=torch.float32, device=dev); d_mask = torch.zeros(cfg.n, dtype=torch.uint8, device=dev)
worlds = [int(w) for w in sys.argv[1:]] or [1, 2, 4, 8]
base = None
for world in worlds:
    kw = cfg.params(); kw["max_triangles"] = cfg.T * world
    p = pkg.make_params(shard_rank=0, shard_world=world, shard_block=1000, flags=pkg.SC_FLAG_TIMING_HOT, **kw)
    split = os.environ.get("SPLIT", "1" if world >= 4 else "0") == "1"
    d_hist = torch.zeros(pkg.SC_HIST_WORDS, dtype=torch.int32, device=dev)
    summed = torch.zeros_like(d_hist)
    if split:  # what the all-reduce would deliver: the sum of every rank's share
        for r in range(world):
            pr = pkg.make_params(shard_rank=r, shard_world=world, shard_block=1000, **kw)
            reg.hypothesize_begin_device(d_src.data_ptr(), d_tgt.data_ptr(), cfg.n, pr, d_hist.data_ptr())
            summed += d_hist
        torch.cuda.synchronize()
    def one():
        if split:
            reg.hypothesize_begin_device(d_src.data_ptr(), d_tgt.data_ptr(), cfg.n, p, d_hist.data_ptr())
            d_hist.copy_(summed)
            reg.hypothesize_end_device(d_hist.data_ptr(), d_key.data_ptr())
        else:
            reg.hypothesize_device(d_src.data_ptr(), d_tgt.data_ptr(), cfg.n, p, d_key.data_ptr())
        return reg.finalize_device(d_key.data_ptr(), d_Rt.data_ptr(), d_mask.data_ptr())
    for _ in range(3):
        one()
    torch.cuda.synchronize(); t0 = time.perf_counter(); K = 30
    for _ in range(K):
        _, st = one()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    base = base or dt * world  # weak scaling: linear = the first row's rate per rank
    print(f"world={world} T_total={cfg.T*world} per-rank step {dt*1e3:.3f} ms -> job {cfg.T*world/dt/1e6:.1f} M hyp/s (vs linear from the first row: {base/worlds[0]/dt*100:.0f}%)  tri_enum={st['tri_total']} score={st['us_score']:.0f}us scored={st['tri_scored']} sample={'sharded' if split else 'replicated'}")
