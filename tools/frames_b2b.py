"""Lab: where does a stream of frames lose time?  python tools/frames_b2b.py <sync|b2b> [C2] [nodense]"""
import os, sys, time; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
mode = sys.argv[1]; cfgname = sys.argv[2] if len(sys.argv) > 2 else "C2"; nodense = "nodense" in sys.argv
dev = torch.device("cuda", 0)
cfg, scene = pkg.synth.make_config_scene(cfgname)
d_src = torch.from_numpy(scene.src).to(dev); d_tgt = torch.from_numpy(scene.tgt).to(dev)
st = torch.cuda.current_stream().cuda_stream
NC = 4 if mode.endswith("4s") else (3 if mode.endswith("3s") else 2)
regs = [pkg.Registrar(0) for _ in range(NC)]
outs = [(torch.zeros(12, dtype=torch.float32, device=dev), torch.zeros(cfg.n, dtype=torch.uint8, device=dev)) for _ in range(NC)]
streams = [torch.cuda.Stream(device=dev) for _ in range(NC)]
for i, r in enumerate(regs): r.set_stream(streams[i].cuda_stream if mode.endswith("s") and mode != "sync" else st)
p = pkg.make_params(flags=(pkg.SC_FLAG_NO_DENSE_S if nodense else 0) | (pkg.SC_FLAG_TIMING_HOT if "hot" in sys.argv else 0), **cfg.params())
for r, o in zip(regs, outs):
    for _ in range(5): r.register_device(d_src.data_ptr(), d_tgt.data_ptr(), cfg.n, p, o[0].data_ptr(), o[1].data_ptr())
K = 400
torch.cuda.synchronize(); t0 = time.perf_counter()
if mode == "sync":
    for k in range(K): regs[0].register_device(d_src.data_ptr(), d_tgt.data_ptr(), cfg.n, p, outs[0][0].data_ptr(), outs[0][1].data_ptr())
elif mode == "sync2":   # synchronous, but the two contexts alternate
    for k in range(K): regs[k & 1].register_device(d_src.data_ptr(), d_tgt.data_ptr(), cfg.n, p, outs[k & 1][0].data_ptr(), outs[k & 1][1].data_ptr())
elif mode.endswith("s") and mode != "sync":   # NC frames in flight, each context on a stream of its own: the frames OVERLAP on the GPU
    for k in range(NC - 1):
        regs[k].register_device_async(d_src.data_ptr(), d_tgt.data_ptr(), cfg.n, p, outs[k][0].data_ptr(), outs[k][1].data_ptr())
    for k in range(NC - 1, K):
        regs[k % NC].register_device_async(d_src.data_ptr(), d_tgt.data_ptr(), cfg.n, p, outs[k % NC][0].data_ptr(), outs[k % NC][1].data_ptr())
        regs[(k - NC + 1) % NC].wait()
    for k in range(K - NC + 1, K):
        regs[k % NC].wait()
else:
    regs[0].register_device_async(d_src.data_ptr(), d_tgt.data_ptr(), cfg.n, p, outs[0][0].data_ptr(), outs[0][1].data_ptr())
    for k in range(1, K):
        regs[k & 1].register_device_async(d_src.data_ptr(), d_tgt.data_ptr(), cfg.n, p, outs[k & 1][0].data_ptr(), outs[k & 1][1].data_ptr())
        regs[(k - 1) & 1].wait()
    regs[(K - 1) & 1].wait()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"{mode} {cfgname} {'nodense' if nodense else 'dense'}{' hot' if 'hot' in sys.argv else ''}: {dt / K * 1e3:.4f} ms per call, fast_path {[r.debug_last()['fast_path'] for r in regs]}", flush=True)
