#!/usr/bin/env python3
"""r05 development: tools/soak.py with every call on every context logged (method, n, T, flags, status), stopping at the first
mismatch / exception with the last calls of each context printed."""
import collections, os, sys, traceback
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
ALL = []
_orig_init = pkg.Registrar.__init__
def _init(self, *a, **k):
    _orig_init(self, *a, **k)
    self._log = collections.deque(maxlen=14)
    ALL.append(self)
pkg.Registrar.__init__ = _init
def _wrap(name):
    f = getattr(pkg.Registrar, name)
    def w(self, *a, **k):
        prm = [x for x in a if isinstance(x, pkg.ScParams)]
        n = [x for x in a if isinstance(x, int) and 3 <= x <= 100000]
        tag = (name, n[0] if n else None, (prm[0].max_triangles, prm[0].flags) if prm else None)
        try:
            r = f(self, *a, **k)
        except Exception as ex:
            self._log.append(tag + ("EXC " + str(ex)[:120],))
            raise
        rc = r[0] if isinstance(r, tuple) else None
        extra = None
        if isinstance(r, tuple) and isinstance(r[1], dict):
            extra = (r[1].get("edges"), r[1].get("tri_total"), r[1].get("tri_kept"))
        self._log.append(tag + (rc, extra))
        return r
    setattr(pkg.Registrar, name, w)
for m in ("register_device", "register_device_async", "wait", "hypothesize_device", "hypothesize_begin_device", "hypothesize_end_device",
          "finalize_device", "finalize_gathered_device", "finalize_gathered_device_async"):
    _wrap(m)
def dump():
    for i, g in enumerate(ALL):
        print(f"--- context {i}:", flush=True)
        for e in g._log:
            print("     ", e, flush=True)
        try:
            print("      debug_last:", g.debug_last(), "|", g._lib.sc_last_error(g._h).decode(), flush=True)
        except Exception as ex:
            print("      debug_last failed:", ex, flush=True)
src = open(os.path.join(ROOT, "tools", "soak.py")).read()
src = src.replace("mism += 1\n", "mism += 1; dump(); budget = 0\n")
sys.argv = [os.path.join(ROOT, "tools", "soak.py")] + sys.argv[1:]
try:
    exec(compile(src, "soak.py", "exec"), {"__name__": "__main__", "__file__": os.path.join(ROOT, "tools", "soak.py"), "dump": dump})
except SystemExit:
    raise
except BaseException:
    traceback.print_exc()
    dump()
    sys.exit(2)
