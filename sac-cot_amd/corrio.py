"""Loaders for correspondence dumps.  The reference (/root/reference/README.md:1-2) defines no file
format, so these are the generic shapes registration pipelines write: one correspondence per row,
`x y z x' y' z'` (source point, matched target point).

  .txt / .csv / .xyz   six numeric columns, whitespace- or comma-separated, `#` comments and one header line allowed
  .npy                 array of shape (n, 6) or (2, n, 3)
  .npz                 arrays `src` and `tgt` of shape (n, 3)

Everything is returned as two C-contiguous float32 arrays of shape (n, 3) — the SC_AOS layout of include/saccot.h.
Host-side convenience only: nothing here is on the measured path."""
from __future__ import annotations

import os

import numpy as np


def _finite_pair(src: np.ndarray, tgt: np.ndarray, where: str):
    src = np.ascontiguousarray(src, dtype=np.float32)
    tgt = np.ascontiguousarray(tgt, dtype=np.float32)
    if src.ndim != 2 or src.shape[1] != 3 or src.shape != tgt.shape:
        raise ValueError(f"{where}: expected two (n, 3) arrays, got {src.shape} and {tgt.shape}")
    if src.shape[0] < 3:
        raise ValueError(f"{where}: at least 3 correspondences are needed, got {src.shape[0]}")
    if not (np.isfinite(src).all() and np.isfinite(tgt).all()):
        raise ValueError(f"{where}: non-finite coordinate")
    return src, tgt


def load_correspondences(path: str):
    """Read a correspondence file; returns (src, tgt), float32 (n, 3) each."""
    ext = os.path.splitext(path)[1].lower()
    if ext == ".npz":
        with np.load(path, allow_pickle=False) as z:
            if "src" not in z or "tgt" not in z:
                raise ValueError(f"{path}: an .npz needs arrays 'src' and 'tgt'")
            return _finite_pair(z["src"], z["tgt"], path)
    if ext == ".npy":
        a = np.load(path, allow_pickle=False)
        if a.ndim == 2 and a.shape[1] == 6:
            return _finite_pair(a[:, :3], a[:, 3:], path)
        if a.ndim == 3 and a.shape[0] == 2 and a.shape[2] == 3:
            return _finite_pair(a[0], a[1], path)
        raise ValueError(f"{path}: expected shape (n, 6) or (2, n, 3), got {a.shape}")
    rows = []
    with open(path, "r", encoding="utf-8", errors="replace") as f:
        for ln, line in enumerate(f, 1):
            line = line.split("#", 1)[0].replace(",", " ").replace(";", " ").strip()
            if not line:
                continue
            parts = line.split()
            try:
                vals = [float(p) for p in parts[:6]]
            except ValueError:
                if not rows:  # one header line
                    continue
                raise ValueError(f"{path}:{ln}: not numeric: {line[:60]!r}") from None
            if len(vals) != 6:
                raise ValueError(f"{path}:{ln}: expected 6 columns (x y z x' y' z'), got {len(parts)}")
            rows.append(vals)
    if not rows:
        raise ValueError(f"{path}: no correspondences")
    a = np.asarray(rows, dtype=np.float64)
    return _finite_pair(a[:, :3], a[:, 3:], path)


def save_correspondences(path: str, src: np.ndarray, tgt: np.ndarray) -> None:
    """Write (src, tgt) in the format the extension names (.txt/.csv, .npy as (n, 6), .npz)."""
    src, tgt = _finite_pair(src, tgt, "save_correspondences")
    ext = os.path.splitext(path)[1].lower()
    if ext == ".npz":
        np.savez(path, src=src, tgt=tgt)
    elif ext == ".npy":
        np.save(path, np.hstack([src, tgt]))
    else:
        sep = "," if ext == ".csv" else " "
        np.savetxt(path, np.hstack([src, tgt]), fmt="%.9g", delimiter=sep, header="x y z x' y' z'")
