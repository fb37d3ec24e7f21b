#!/usr/bin/env python3
"""A/B of sc_debug knobs / flags on one GPU: per-stage device times (SC_FLAG_TIMING pass) and the untimed-bracket step
time for each variant.   python tools/ab_stage.py C2 C3 -- name:key=val,key=val[,flags=N] ...
Example:  python tools/ab_stage.py C2 -- base: st4:compat_store_mode=1 nt:compat_store_mode=2 nodense:flags=32"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
args = sys.argv[1:]
cfgs = args[:args.index("--")] if "--" in args else ["C2"]
variants = args[args.index("--") + 1:] if "--" in args else ["base:"]
dev = torch.device("cuda", 0)
keys = ("us_stage", "us_compat", "us_triangles", "us_trikeys", "us_kabsch", "us_score", "us_argmax", "us_mask")
for name in cfgs:
    cfg, scene = pkg.synth.make_config_scene(name)
    d_src = torch.from_numpy(scene.src).to(dev); d_tgt = torch.from_numpy(scene.tgt).to(dev)
    d_Rt = torch.zeros(12, dtype=torch.float32, device=dev); d_mask = torch.zeros(cfg.n, dtype=torch.uint8, device=dev)
    for v in variants:
        vname, _, spec = v.partition(":")
        knobs = dict(kv.split("=") for kv in spec.split(",") if kv)
        flags = int(knobs.pop("flags", 0))
        reg = pkg.Registrar(0)
        reg.set_stream(torch.cuda.current_stream().cuda_stream)
        if knobs:
            reg.set_debug(**{k: int(x) for k, x in knobs.items()})
        p_hot = pkg.make_params(flags=flags, **cfg.params())
        p_diag = pkg.make_params(flags=flags | pkg.SC_FLAG_TIMING, **cfg.params())
        for _ in range(3):
            reg.register_device(d_src.data_ptr(), d_tgt.data_ptr(), cfg.n, p_hot, d_Rt.data_ptr(), d_mask.data_ptr())
        torch.cuda.synchronize(); K = 20; t0 = time.perf_counter()
        for _ in range(K):
            reg.register_device(d_src.data_ptr(), d_tgt.data_ptr(), cfg.n, p_hot, d_Rt.data_ptr(), d_mask.data_ptr())
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
        acc = {k: 0.0 for k in keys}
        for _ in range(10):
            _, st = reg.register_device(d_src.data_ptr(), d_tgt.data_ptr(), cfg.n, p_diag, d_Rt.data_ptr(), d_mask.data_ptr())
            for k in keys:
                acc[k] += st[k] / 10
        print(f"{name} {vname:12s} step {dt*1e3:7.4f} ms | " + " ".join(f"{k[3:]}={acc[k]:7.1f}" for k in keys) +
              f" | rank={st['best_rank']} cnt={st['best_count']} tri={st['tri_total']}", flush=True)
        reg.close()
