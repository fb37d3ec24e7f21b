import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
dev = torch.device("cuda:0")
cfg, scene = pkg.synth.make_config_scene("C4")
ds = torch.from_numpy(scene.src).to(dev); dt = torch.from_numpy(scene.tgt).to(dev)
p = pkg.make_params(**cfg.params())
for nofast in (1, 0):
    r = pkg.Registrar(0)
    r.set_stream(torch.cuda.current_stream().cuda_stream)
    if nofast: r.set_debug(no_fast=1)
    d_Rt = torch.zeros(12, dtype=torch.float32, device=dev); d_mask = torch.zeros(cfg.n, dtype=torch.uint8, device=dev)
    for k in range(4):
        rc, st = r.register_device(ds.data_ptr(), dt.data_ptr(), cfg.n, p, d_Rt.data_ptr(), d_mask.data_ptr())
        print("nofast", nofast, "call", k, "fast", r.debug_last()["fast_path"], st["edges"], st["tri_total"], st["best_rank"], r._lib.sc_last_error(r._h).decode(), flush=True)
    r.close()
