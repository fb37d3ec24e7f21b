"""Multi-GPU layer: one process per GPU, hypotheses sharded, one 8-byte max all-reduce.

Stages A and B are replicated (deterministic, so every rank holds the identical ranked list); stage C scores this
rank's blocks of the ranked list (blocks of `shard_block` triangles dealt round-robin, so every rank sees the same
mix of high- and low-ranked triangles).  The winner key K = (count << 32) | (0xFFFFFFFF - rank_index) embeds the
global rank index, so after `all_reduce(MAX)` every rank decodes the same winner and re-solves it locally — no
broadcast.  The collective is torch.distributed's: backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU tests.
"""
from __future__ import annotations

import numpy as np


def global_index(l: np.ndarray | int, block: int, rank: int, world: int):
    """Global rank index of local hypothesis l (mirror of shard_global_index in csrc/sc_score.hip)."""
    l = np.asarray(l, dtype=np.int64)
    return ((l // block) * world + rank) * block + (l % block)


def local_count(t_eff: int, block: int, rank: int, world: int) -> int:
    n, gb = 0, rank
    while gb * block < t_eff:
        n += min((gb + 1) * block, t_eff) - gb * block
        gb += world
    return n


def local_indices(t_eff: int, block: int, rank: int, world: int) -> np.ndarray:
    """Global rank indices scored by `rank`, in local order."""
    return global_index(np.arange(local_count(t_eff, block, rank, world)), block, rank, world)


def encode_key(count: int, rank_index: int) -> int:
    """Single-stage key of a RANKED list (stage hook sc_score_host, oracle so_best_key)."""
    return 0 if count == 0 else (int(count) << 32) | (0xFFFFFFFF - int(rank_index))


def decode_key(key: int) -> tuple[int, int]:
    """-> (count, rank_index)"""
    return int(key) >> 32, 0xFFFFFFFF - (int(key) & 0xFFFFFFFF)


def encode_pair(count: int, ranking_key: int, position: int) -> tuple[int, int]:
    """Two-stage winner key of the hot path (include/saccot.h, sc_hypothesize_device)."""
    if count == 0:
        return 0, 0
    return (int(count) << 32) | (int(ranking_key) & 0xFFFFFFFF), 0xFFFFFFFF - int(position)


def decode_pair(k0: int, k1: int) -> tuple[int, int, int]:
    """-> (count, ranking_key, position)"""
    return int(k0) >> 32, int(k0) & 0xFFFFFFFF, 0xFFFFFFFF - (int(k1) & 0xFFFFFFFF)


def reduce_pairs(pairs) -> tuple[int, int]:
    """Host-side reference of the two-step reduction over ranks: max K0, then max K1 among the ranks attaining it."""
    k0 = max(int(p[0]) for p in pairs)
    k1 = max((int(p[1]) for p in pairs if int(p[0]) == k0), default=0)
    return k0, (k1 if k0 else 0)


def allreduce_best(key2):
    """In-place two-step MAX all-reduce of the int64 key-pair tensor (2 elements) over the default process group.
    Values are < 2^63 (count < 2^31), so the signed int64 view orders like the unsigned keys.  No host sync."""
    import torch
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        mine = key2[0:1].clone()
        dist.all_reduce(key2[0:1], op=dist.ReduceOp.MAX)
        key2[1:2] = torch.where(mine == key2[0:1], key2[1:2], torch.zeros_like(key2[1:2]))
        dist.all_reduce(key2[1:2], op=dist.ReduceOp.MAX)
    return key2


def allreduce_hist(d_hist) -> None:
    """Sum the pruning-sample histograms of all ranks in place (include/saccot.h, sc_hypothesize_begin_device).
    d_hist: int32 tensor of SC_HIST_WORDS entries (the counts are < 2^31: at most 5T/8 * 64 * 64 sampled triangles
    would be needed to overflow — the bins are u32 on the device and the int32 view adds the same bit patterns).
    One 1 KiB SUM all-reduce; no-op without an initialised process group or with world size 1."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    dist.all_reduce(d_hist, op=dist.ReduceOp.SUM)


def _flat_allgather_ok() -> bool:
    """Which all-gather form this process group takes, decided from the BACKEND NAME — never by trying one form and
    catching its exception: a RuntimeError raised on one rank only (an asynchronous RCCL error, a watchdog timeout)
    would send that rank into a different collective from its peers and desynchronise the job instead of ending it.
    "nccl" (= RCCL on ROCm) has all_gather_into_tensor, also with the output aliasing the input slice; gloo gets the
    list form."""
    import torch.distributed as dist
    return "nccl" in str(dist.get_backend()).lower()


def allgather_best(key2, out):
    """ONE all-gather of every rank's 16-byte winner key pair into `out` (int64 tensor of 2 * world entries, rank r's
    pair at out[2r : 2r + 2]) for Registrar.finalize_gathered_device, instead of the two dependent MAX all-reduces of
    allreduce_best.  Without a process group (or world 1) the pair is copied to out[0:2].  No host sync."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if _flat_allgather_ok():
            dist.all_gather_into_tensor(out, key2)
        else:
            dist.all_gather(list(out.view(-1, 2).unbind(0)), key2)
    else:
        out[0:2].copy_(key2)
    return out


# ---------------------------------------------------------------------------------------------------------
# Stages A and B sharded too: host side of the phase API of include/saccot.h, one process per GPU.
# ---------------------------------------------------------------------------------------------------------
def allgather_inplace(buf, rank: int, world: int):
    """In-place all-gather of a flat tensor made of `world` equal slices: rank r contributes slice r (which it has
    already written) and receives the others.  RCCL (backend "nccl") takes the aliased form directly; backends
    without all_gather_into_tensor (gloo) get a list of views and a copy of the own slice.  No host sync."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or world == 1:
        return buf
    per = buf.numel() // world
    mine = buf[rank * per:(rank + 1) * per]
    if _flat_allgather_ok():
        dist.all_gather_into_tensor(buf, mine)
    else:
        dist.all_gather([buf[r * per:(r + 1) * per] for r in range(world)], mine.clone())
    return buf


class ShardedStep:
    """One rank's buffers and phase sequence for the fully sharded path:
         compat (row block) -> all-gather bit rows -> edges + sample [share -> all-reduce histogram: only without
         SC_FLAG_EST_BOUND] -> own top-T -> all-gather candidate blobs -> merge + C1 + C2 -> all-gather key pairs -> finalize.
    Every collective is torch.distributed's on the context's stream (backend "nccl" = RCCL over xGMI; "gloo" in the
    CPU rehearsal).  With world == 1 no collective is issued."""

    def __init__(self, pkg, reg, n: int, params, rank: int, world: int, device):
        import torch
        self.pkg, self.reg, self.n, self.p, self.rank, self.world, self.device = pkg, reg, n, params, rank, world, device
        self.level = int(params.shard_cand_level)   # sticky: raised after SC_ERETRY (candidate blobs too small)
        # SC_FLAG_EST_BOUND: stage B pruned by an ESTIMATED bound — every rank takes the whole (cheap) sample, so the histogram
        # all-reduce is skipped: three collectives per step, not four.  Sticky off after SC_EBOUND (the estimate was too high
        # for this kind of input: every rank gets that status together, the step is repeated with the certifying sample).
        # Only on graphs below 8192 correspondences: there the whole sample costs a rank ~14 us — less than the all-reduce it saves —
        # and the tighter bound shrinks everything after it (emulated C2 weak / 8 ranks 0.34 -> 0.32 ms, C4 strong / 8 0.35 -> 0.31);
        # at C3 (N = 20 000) the replicated sample costs more than a shared certifying one (0.69 vs 0.61 ms per rank-step at 8 ranks).
        self.estimate = n < 8192
        plan = pkg.shard_plan(self._with_level(params), n)
        self.plan = plan
        self.bits = torch.zeros(plan.bits_bytes_total // 8, dtype=torch.int64, device=device)
        self.hist = torch.zeros(pkg.SC_HIST_WORDS, dtype=torch.int32, device=device)
        self.cand = torch.zeros(world * plan.cand_bytes_per_rank // 8, dtype=torch.int64, device=device)
        self.keys = torch.zeros(2 * world, dtype=torch.int64, device=device)
        self.Rt = torch.zeros(12, dtype=torch.float32, device=device)
        self.mask = torch.zeros(n, dtype=torch.uint8, device=device)

    def bytes_exchanged(self) -> dict:
        """Bytes this rank RECEIVES per step and collective (for pricing against the xGMI links)."""
        w = self.world
        return {"bit_rows": (w - 1) * int(self.plan.bits_bytes_per_rank), "histogram": 1024 if (w > 1 and not self.estimate) else 0,
                "candidates": (w - 1) * int(self.plan.cand_bytes_per_rank), "key_pairs": 16 * (w - 1)}

    def _with_level(self, p):
        q = type(p).from_buffer_copy(p)
        q.shard_cand_level = self.level
        if self.estimate and self.world > 1:
            q.flags |= self.pkg.SC_FLAG_EST_BOUND
        else:
            q.flags &= ~self.pkg.SC_FLAG_EST_BOUND
        return q

    def step(self, d_src: int, d_tgt: int, params=None):
        """One call of the path.  SC_ERETRY (a candidate blob was too small; every rank sees the same blobs, so every rank
        gets it together) raises the blob level for good, reallocates the blobs and runs the call again."""
        import torch
        p = params or self.p
        for _ in range(20):
            rc, st = self._step_once(d_src, d_tgt, self._with_level(p))
            if rc == self.pkg.SC_EBOUND:   # (every rank alike) the estimated bound failed: certify from now on
                self.estimate = False
                continue
            if rc != self.pkg.SC_ERETRY:
                return rc, st
            self.level += 1
            self.plan = self.pkg.shard_plan(self._with_level(p), self.n)   # the parameters actually in use
            self.cand = torch.zeros(self.world * self.plan.cand_bytes_per_rank // 8, dtype=torch.int64, device=self.device)
        return rc, st

    def _step_once(self, d_src: int, d_tgt: int, p):
        r, w, reg = self.rank, self.world, self.reg
        reg.shard_compat_device(d_src, d_tgt, self.n, p, self.bits.data_ptr())
        allgather_inplace(self.bits, r, w)
        reg.shard_edges_device(self.hist.data_ptr())
        if not (p.flags & self.pkg.SC_FLAG_EST_BOUND):   # (with the flag every rank already holds the whole sample)
            allreduce_hist(self.hist)
        reg.shard_select_device(self.hist.data_ptr(), self.cand.data_ptr() + r * int(self.plan.cand_bytes_per_rank))
        allgather_inplace(self.cand, r, w)
        reg.shard_score_device(self.cand.data_ptr(), self.keys.data_ptr() + 16 * r)
        allgather_inplace(self.keys, r, w)
        return reg.finalize_gathered_device(self.keys.data_ptr(), w, self.Rt.data_ptr(), self.mask.data_ptr())


class ReplicatedStream:
    """One rank of a job whose ranks REPLICATE stages A and B (graphs below 8192 correspondences: what `bench.py --gpus N` runs by
    default), registering a STREAM of frames: frame f + 1 is enqueued — host-free sc_hypothesize_device (SC_FLAG_EST_BOUND) on the
    second of two contexts bound to one stream, ONE all-gather of the 16-byte key pairs, sc_finalize_gathered_device_async —
    before frame f's sc_wait delivers its status.  SC_EBOUND (every rank alike: the stages are replicated and deterministic, and the
    ranks' contexts have seen the same calls in the same order) means this frame again without the flag; two FAILED ESTIMATES
    (sc_debug_last.prune_bound == 2, as opposed to a host-free call whose covers were outgrown) switch the flag off for good.
    `pair`: two Registrar-like objects; `ptr`: how a buffer becomes what they take (tensor.data_ptr() for the library; the CPU test of
    this class passes the tensors themselves to stand-ins built on the CPU restatement: tests/test_shard_gloo.py)."""

    def __init__(self, pkg, pair, n: int, params, world: int, make_i64, ptr=lambda t: t.data_ptr()):
        self.pkg, self.pair, self.n, self.world, self.ptr = pkg, pair, n, world, ptr
        self.p_plain = type(params).from_buffer_copy(params)
        self.p_plain.flags &= ~pkg.SC_FLAG_EST_BOUND
        self.p_est = type(params).from_buffer_copy(params)
        self.p_est.flags |= pkg.SC_FLAG_EST_BOUND
        self.keys = [make_i64(2), make_i64(2)]
        self.alls = [make_i64(2 * world), make_i64(2 * world)]
        self.estimate, self.est_fails, self.redone = True, 0, 0
        self._frame = {}

    def _run(self, i, params, src, tgt):
        g = self.pair[i]
        g.hypothesize_device(self.ptr(src), self.ptr(tgt), self.n, params, self.ptr(self.keys[i]))
        allgather_best(self.keys[i], self.alls[i])   # ONE collective (16 bytes per rank), in stream order

    def enqueue(self, f: int, src, tgt, Rt, mask) -> None:
        i = f & 1
        self._frame[i] = (src, tgt, Rt, mask)
        self._run(i, self.p_est if self.estimate else self.p_plain, src, tgt)
        self.pair[i].finalize_gathered_device_async(self.ptr(self.alls[i]), self.world, self.ptr(Rt), self.ptr(mask))

    def collect(self, f: int):
        """frame f's status and statistics; its (R, t) and mask are complete"""
        i = f & 1
        rc, st = self.pair[i].wait()
        if rc == self.pkg.SC_EBOUND:
            rc, st = self._again(i)
        return rc, st

    def waited(self, i: int, src, tgt, Rt, mask):
        """one frame on context i, complete on return (warm-up; the `waited` leg of bench.py)"""
        self._frame[i] = (src, tgt, Rt, mask)
        self._run(i, self.p_est if self.estimate else self.p_plain, src, tgt)
        rc, st = self.pair[i].finalize_gathered_device(self.ptr(self.alls[i]), self.world, self.ptr(Rt), self.ptr(mask))
        if rc == self.pkg.SC_EBOUND:
            rc, st = self._again(i)
        return rc, st

    def _again(self, i: int):
        """SC_EBOUND came back: the frame once more, the certifying (and waiting) way — on every rank alike"""
        self.redone += 1
        if self.pair[i].debug_last()["prune_bound"] == 2:   # (off the fast path: sc_debug_last synchronises)
            self.est_fails += 1
            if self.est_fails >= 2:
                self.estimate = False
        src, tgt, Rt, mask = self._frame[i]
        self._run(i, self.p_plain, src, tgt)
        return self.pair[i].finalize_gathered_device(self.ptr(self.alls[i]), self.world, self.ptr(Rt), self.ptr(mask))
