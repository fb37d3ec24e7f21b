import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
for name in sys.argv[1:]:
    cfg, scene = pkg.synth.make_config_scene(name)
    reg = pkg.Registrar(0)
    p = pkg.make_params(**cfg.params())
    tri, key, total, edges = reg.triangles(scene.src, scene.tgt, p)
    Rt = reg.kabsch(scene.src, scene.tgt, p, tri)
    for flt in (2, 3):
        reg.set_debug(score_filter=flt)
        cnt, k = reg.score(scene.src, scene.tgt, p, Rt)
        print(name, "filter", flt, reg.debug_last(), "tests", cfg.T * cfg.n, flush=True)
    reg.close()
