import sys, time
sys.path.insert(0, ".")
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
dev = torch.device("cuda", 0)
for name in sys.argv[1:]:
    cfg, scene = pkg.synth.make_config_scene(name)
    d_src = torch.from_numpy(scene.src).to(dev); d_tgt = torch.from_numpy(scene.tgt).to(dev)
    d_Rt = torch.zeros(12, dtype=torch.float32, device=dev); d_mask = torch.zeros(cfg.n, dtype=torch.uint8, device=dev)
    for sp in (3, 5, 6, 7, 8):
        reg = pkg.Registrar(0); reg.set_stream(torch.cuda.current_stream().cuda_stream)
        reg.set_debug(filter_splits=sp)
        p = pkg.make_params(flags=pkg.SC_FLAG_TIMING, **cfg.params())
        acc = 0.0
        for it in range(6):
            _, st = reg.register_device(d_src.data_ptr(), d_tgt.data_ptr(), cfg.n, p, d_Rt.data_ptr(), d_mask.data_ptr())
            if it >= 2: acc += st["us_score"] / 4
        d = reg.debug_last()
        print(name, "splits", sp, "score us %.1f" % acc, {k: d[k] for k in ("filter_splits", "filter_undecided", "filter_recounts", "gram_near_corr", "gram_near_hyp")}, flush=True)
        reg.close()
