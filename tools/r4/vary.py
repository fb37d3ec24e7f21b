import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package()
cfg, scene = pkg.synth.make_config_scene("C2")
n = cfg.n
for knobs in ({}, {"no_edge_build": 1}, {"no_estimate": 1}, {"pad_": 1}):
    reg = pkg.Registrar(0)
    if knobs: reg.set_debug(**knobs)
    seq = [n, int(0.8 * n), int(0.6 * n), int(0.9 * n)]
    for m in seq: reg.register(scene.src[:m], scene.tgt[:m], **cfg.params())
    out = []
    for _ in range(3):
        for m in seq:
            t0 = time.perf_counter(); r = reg.register(scene.src[:m], scene.tgt[:m], **cfg.params()); dt = time.perf_counter() - t0
            d = reg.debug_last()
            out.append(f"{m}:{dt*1e3:.2f}ms/f{d['fast_path']}/p{d['prune_bound']}/E{r['stats']['edges']}/M{r['stats']['tri_total']}")
    print(knobs, " ".join(out), flush=True)
    reg.close()
