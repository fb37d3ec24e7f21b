import sys; sys.path.insert(0, ".")
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
dev = torch.device("cuda", 0)
cfg, scene = pkg.synth.make_config_scene("C2")
d_src = torch.from_numpy(scene.src).to(dev); d_tgt = torch.from_numpy(scene.tgt).to(dev)
d_Rt = torch.zeros(12, dtype=torch.float32, device=dev); d_mask = torch.zeros(cfg.n, dtype=torch.uint8, device=dev)
reg = pkg.Registrar(0); reg.set_stream(torch.cuda.current_stream().cuda_stream)
p = pkg.make_params(flags=pkg.SC_FLAG_TIMING_HOT, **cfg.params())
for i in range(4):
    rc, st = reg.register_device(d_src.data_ptr(), d_tgt.data_ptr(), cfg.n, p, d_Rt.data_ptr(), d_mask.data_ptr())
    d = reg.debug_last()
    print(st["us_score"], d["us_c2_filter"], d["c2_kernel"], pkg.api.C.sizeof(pkg.api.ScDebugInfo), reg._lib.sc_last_error(reg._h) if hasattr(reg._lib, "sc_last_error") else "")
