import sys; sys.path.insert(0, ".")
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
dev = torch.device("cuda", 0)
cfg, scene = pkg.synth.make_config_scene(sys.argv[1] if len(sys.argv) > 1 else "C2")
d_src = torch.from_numpy(scene.src).to(dev); d_tgt = torch.from_numpy(scene.tgt).to(dev)
d_Rt = torch.zeros(12, dtype=torch.float32, device=dev); d_mask = torch.zeros(cfg.n, dtype=torch.uint8, device=dev)
reg = pkg.Registrar(0); reg.set_stream(torch.cuda.current_stream().cuda_stream)
p = pkg.make_params(**cfg.params())
for i in range(4):
    reg.register_device(d_src.data_ptr(), d_tgt.data_ptr(), cfg.n, p, d_Rt.data_ptr(), d_mask.data_ptr())
reg.set_debug(filter_lds_queue=127)
for i in range(3):
    reg.register_device(d_src.data_ptr(), d_tgt.data_ptr(), cfg.n, p, d_Rt.data_ptr(), d_mask.data_ptr())
    torch.cuda.synchronize(); print("----", flush=True)
