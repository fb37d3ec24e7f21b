#!/usr/bin/env python3
"""Register one correspondence file on the GPU:  python tools/register_file.py corr.txt --tau 0.1 [--T 50000]
Prints R, t, the inlier count and writes the inlier mask next to the input (<file>.inliers.txt) with --save-mask."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--tau", type=float, required=True, help="inlier distance, in the units of the file")
    ap.add_argument("--sigma", type=float, default=None, help="rigidity sigma (default: tau)")
    ap.add_argument("--t-cmp", type=float, default=0.9)
    ap.add_argument("--min-len", type=float, default=None, help="minimum segment length (default: tau)")
    ap.add_argument("--T", type=int, default=50000, help="triangle hypotheses scored")
    ap.add_argument("--refine", action="store_true", help="fp64 least-squares refit over the winner's inliers")
    ap.add_argument("--save-mask", action="store_true")
    a = ap.parse_args()
    pkg = ge.load_package()
    src, tgt = pkg.corrio.load_correspondences(a.path)
    reg = pkg.Registrar(0)  # raises without a GPU: there is no CPU fallback
    out = reg.register(src, tgt, sigma=a.sigma or a.tau, t_cmp=a.t_cmp, tau=a.tau, min_len=a.min_len or a.tau,
                       max_triangles=a.T, flags=pkg.SC_FLAG_REFINE if a.refine else 0)
    np.set_printoptions(precision=7, suppress=True)
    print(f"n = {len(src)}  status = {out['status']}  inliers = {int(out['mask'].sum())}  "
          f"edges = {out['stats']['edges']}  triangles kept = {out['stats']['tri_kept']}")
    print("R =\n", out["R"], "\nt =", out["t"])
    if a.save_mask:
        np.savetxt(a.path + ".inliers.txt", out["mask"], fmt="%d")
    reg.close()
    return 0 if out["status"] == 0 else 1


if __name__ == "__main__":
    raise SystemExit(main())
