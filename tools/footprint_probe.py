#!/usr/bin/env python3
"""r05 probe: in the kernel trace the memory-heavy kernels of a STREAMED frame run ~0.5 - 1.2 us longer each than the same kernels of a
waited frame (~6 us per frame).  Two things differ: the stream keeps the GPU busy without a pause, and it alternates TWO contexts
(2 x ~130 MB of workspace, of which 2 x 100 MB are the dense matrices S) where the waited form re-uses ONE.  This separates them:
waited frames on one context, waited frames alternating two contexts, waited frames alternating two contexts WITHOUT the dense matrix.
python tools/footprint_probe.py [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 400
cfg, scenes = pkg.synth.make_stream_scenes("C2", 32)
dev = torch.device("cuda", 0)
d = [(torch.from_numpy(s.src).to(dev), torch.from_numpy(s.tgt).to(dev)) for s in scenes]
Rt = torch.zeros(12, dtype=torch.float32, device=dev); mask = torch.zeros(cfg.n, dtype=torch.uint8, device=dev)
s = torch.cuda.Stream(device=dev)
with torch.cuda.stream(s):
    for flags, label in ((0, "dense S"), (pkg.SC_FLAG_NO_DENSE_S, "no dense S")):
        p = pkg.make_params(flags=flags, **cfg.params())
        for nctx in (1, 2, 4):
            regs = [pkg.Registrar(0) for _ in range(nctx)]
            for g in regs:
                g.set_stream(s.cuda_stream)
            for f in range(16 * nctx):
                regs[f % nctx].register_device(d[f % 32][0].data_ptr(), d[f % 32][1].data_ptr(), cfg.n, p, Rt.data_ptr(), mask.data_ptr())
            res = []
            for rep in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for f in range(K):
                    regs[f % nctx].register_device(d[f % 32][0].data_ptr(), d[f % 32][1].data_ptr(), cfg.n, p, Rt.data_ptr(), mask.data_ptr())
                torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / K * 1e3)
            print(f"{label}: waited frames alternating {nctx} context(s): {' '.join('%.4f' % r for r in res)} ms per frame", flush=True)
            for g in regs:
                g.close()
