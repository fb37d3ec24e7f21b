"""development: per-kernel time of one kernel with sc_debug.reserved[0] (dbg_stop) = k.  python tools/r4/stop.py C2 0 1 2 3"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, ctypes as C
import __graft_entry__ as ge
pkg = ge.load_package()
name = sys.argv[1]; stops = [int(x) for x in sys.argv[2:]]
dev = torch.device("cuda:0")
cfg, scene = pkg.synth.make_config_scene(name)
ds = torch.from_numpy(scene.src).to(dev); dt = torch.from_numpy(scene.tgt).to(dev)
d_Rt = torch.zeros(12, dtype=torch.float32, device=dev); d_mask = torch.zeros(cfg.n, dtype=torch.uint8, device=dev)
p = pkg.make_params(**cfg.params())
for stop in stops:
    r = pkg.Registrar(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
    d = pkg.api.ScDebug(size=C.sizeof(pkg.api.ScDebug), compact_self_max=-1, scan_self_max=-1)
    d.reserved[0] = stop; d.no_fast = 1
    for kv in os.environ.get("KNOBS", "").split(","):
        if kv: setattr(d, kv.split("=")[0], int(kv.split("=")[1]))
    r._check(r._lib.sc_set_debug(r._h, C.byref(d)))
    for _ in range(12):
        try: r.register_device(ds.data_ptr(), dt.data_ptr(), cfg.n, p, d_Rt.data_ptr(), d_mask.data_ptr())
        except Exception as e: pass
    torch.cuda.synchronize(); r.close()
