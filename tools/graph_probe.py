#!/usr/bin/env python3
"""r05 probe: what would ONE frame cost as a HIP graph?  The host-free enqueue of sc_register_device_async is captured (torch's
stream capture around the call: every launch of the library lands in the capturing stream) and the graph replayed back to back on
the same frame, against the same frame enqueued launch by launch (two contexts, one stream: bench.py's `stream_identical_frames`).
TIMING ONLY: a replay repeats the captured launch arguments (the scan's look-back epoch among them), which the library's protocol
does not allow — nothing here checks results beyond the first collected frame.   python tools/graph_probe.py [C2] [replays]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
name = sys.argv[1] if len(sys.argv) > 1 else "C2"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 300
cfg, scene = pkg.synth.make_config_scene(name)
dev = torch.device("cuda", 0)
d_src = torch.from_numpy(scene.src).to(dev); d_tgt = torch.from_numpy(scene.tgt).to(dev)
outs = [(torch.zeros(12, dtype=torch.float32, device=dev), torch.zeros(cfg.n, dtype=torch.uint8, device=dev)) for _ in range(2)]
p = pkg.make_params(**cfg.params())
regs = [pkg.Registrar(0) for _ in range(2)]
s = torch.cuda.Stream(device=dev)
with torch.cuda.stream(s):
    for g, o in zip(regs, outs):
        g.set_stream(s.cuda_stream)
        for _ in range(12):
            g.register_device(d_src.data_ptr(), d_tgt.data_ptr(), cfg.n, p, o[0].data_ptr(), o[1].data_ptr())
    torch.cuda.synchronize()
    # ---- launch by launch: frame k + 1 enqueued before frame k is waited for
    for rep in range(3):
        t0 = time.perf_counter()
        regs[0].register_device_async(d_src.data_ptr(), d_tgt.data_ptr(), cfg.n, p, outs[0][0].data_ptr(), outs[0][1].data_ptr())
        for f in range(1, K + 1):
            if f < K:
                regs[f & 1].register_device_async(d_src.data_ptr(), d_tgt.data_ptr(), cfg.n, p, outs[f & 1][0].data_ptr(), outs[f & 1][1].data_ptr())
            rc, st = regs[(f - 1) & 1].wait()
        torch.cuda.synchronize()
        print(f"{name} launch by launch: {(time.perf_counter() - t0) / K * 1e3:.4f} ms per frame (winner rank {st['best_rank']}, {st['best_count']} inliers, fast_path {regs[0].debug_last()['fast_path']})", flush=True)
# ---- the same enqueue captured once, replayed
graph = torch.cuda.CUDAGraph()
g0 = regs[0]
with torch.cuda.graph(graph, stream=s, capture_error_mode="relaxed"):
    g0.register_device_async(d_src.data_ptr(), d_tgt.data_ptr(), cfg.n, p, outs[0][0].data_ptr(), outs[0][1].data_ptr())
with torch.cuda.stream(s):
    graph.replay()
torch.cuda.synchronize()
rc, st = g0.wait()
print(f"{name} captured frame collected: rc {rc} winner rank {st['best_rank']}, {st['best_count']} inliers, fast_path {g0.debug_last()['fast_path']}", flush=True)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.cuda.stream(s):
        for _ in range(K):
            graph.replay()
    torch.cuda.synchronize()
    print(f"{name} one graph per frame, replayed back to back: {(time.perf_counter() - t0) / K * 1e3:.4f} ms per frame", flush=True)
