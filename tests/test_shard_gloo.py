"""CPU suite, part 3: the multi-GPU layer without GPUs.

Host logic (round-robin block dealing, key encode/decode) is checked directly; the N > 1 path — every rank scores its
share of the replicated ranked list, one MAX all-reduce of the 8-byte key, every rank decodes the same winner — runs
as two real processes over torch.distributed's gloo backend, with the CPU restatement standing in for the kernels
(tests may call the oracle; the product never does)."""
import os
import socket

import numpy as np
import pytest


@pytest.mark.parametrize("t_eff,block,world", [(0, 1024, 1), (1, 1024, 4), (1023, 256, 3), (50000, 1000, 8),
                                               (400000, 1000, 8), (777, 64, 5), (4096, 1024, 4)])
def test_block_dealing_partitions_the_ranked_list(pkg, t_eff, block, world):
    sh = pkg.shard
    seen = []
    for r in range(world):
        idx = sh.local_indices(t_eff, block, r, world)
        assert len(idx) == sh.local_count(t_eff, block, r, world)
        assert np.all(idx < t_eff) and np.all(np.diff(idx) > 0)
        seen.append(idx)
    allidx = np.sort(np.concatenate(seen)) if seen else np.zeros(0)
    assert np.array_equal(allidx, np.arange(t_eff))                   # every hypothesis scored exactly once
    if t_eff >= block * world:
        first = [s[0] // block for s in seen]
        assert first == list(range(world))                              # every rank starts at the top of the list


def test_pair_reduction_picks_the_best_ranked_winner(pkg):
    sh = pkg.shard
    # rank 0: 9 inliers, key 70, position 5;  rank 1: 9 inliers, key 90, position 17;  rank 2: 9 / 90 / 11
    pairs = [sh.encode_pair(9, 70, 5), sh.encode_pair(9, 90, 17), sh.encode_pair(9, 90, 11), sh.encode_pair(0, 99, 0)]
    assert sh.decode_pair(*sh.reduce_pairs(pairs)) == (9, 90, 11)   # most inliers, best key, lowest position
    assert sh.reduce_pairs([(0, 0), (0, 0)]) == (0, 0)
    assert sh.decode_pair(*sh.reduce_pairs([sh.encode_pair(3, 1, 0), sh.encode_pair(4, 0, 99)])) == (4, 0, 99)


def test_key_encoding_orders_like_the_spec(pkg):
    sh = pkg.shard
    assert sh.encode_key(0, 5) == 0
    assert sh.decode_key(sh.encode_key(734, 22370)) == (734, 22370)
    assert sh.encode_key(10, 99) > sh.encode_key(9, 0)                  # count dominates
    assert sh.encode_key(10, 3) > sh.encode_key(10, 4)                  # ties -> better (lower) rank index
    assert sh.encode_key(2**31 - 1, 0) < 2**63                          # fits torch int64


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_hist(rank, words):
    """A rank's share of the sample histogram (any int32 content will do: the reduction is a plain sum; one bin is
    chosen so that the u32 sum passes 2^31 and must wrap like the device's counters)."""
    h = ((np.arange(words, dtype=np.int64) * (rank + 3) + 7 * rank) % 1000).astype(np.int32)
    h[5] = np.int32(0x7FFFFFF0 if rank == 0 else 0x40)
    return h


def _worker(rank, world, port, block, out_dir):
    import sys
    import torch
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import __graft_entry__ as ge
    pkg = ge.load_package(); O = ge.load_oracle()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg, sc = pkg.synth.make_config_scene("C0")
    kw = cfg.params()
    # stages A + B replicated on every rank (deterministic -> identical ranked lists)
    S, bits, deg = O.compat(sc.src, sc.tgt, kw["sigma"], kw["t_cmp"], kw["min_len"], kw["tau"])
    tri, wkey, _ = O.triangles(S, bits, deg, kw["max_triangles"], 0)
    mine = pkg.shard.local_indices(len(tri), block, rank, world)
    Rt = O.kabsch3(sc.src, sc.tgt, tri[mine])
    cnt = O.score(sc.src, sc.tgt, Rt, kw["tau"])
    # this rank's winner pair: most inliers, then best ranking key, then lowest position (include/saccot.h)
    best = (0, 0)
    for c, g in zip(cnt, mine):
        cand = pkg.shard.encode_pair(int(c), int(wkey[g]), int(g))
        if cand > best:
            best = cand
    key = torch.tensor(best, dtype=torch.int64)
    gathered = torch.zeros(2 * world, dtype=torch.int64)
    pkg.shard.allgather_best(key, gathered)                             # the one-collective form (16 bytes per rank)
    pkg.shard.allreduce_best(key)                                       # the two 8-byte collectives of the path
    assert pkg.shard.reduce_pairs([tuple(int(x) for x in gathered[2 * r:2 * r + 2]) for r in range(world)]) == \
        (int(key[0]), int(key[1]))                                      # both forms agree on every rank
    count, _, widx = pkg.shard.decode_pair(int(key[0]), int(key[1]))
    Rt_w = O.kabsch3(sc.src, sc.tgt, tri[widx:widx + 1])[0]            # every rank re-solves the winner locally
    mask = O.mask(sc.src, sc.tgt, Rt_w, kw["tau"])
    # the split phase 1 (sc_hypothesize_begin_device / _end_device): per-rank sample histograms are summed in place
    hist = torch.from_numpy(_fake_hist(rank, pkg.SC_HIST_WORDS))
    pkg.shard.allreduce_hist(hist)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), count=count, widx=widx, Rt=Rt_w, mask=mask, scored=len(mine),
             hist=hist.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,block", [(2, 32), (2, 1024)])
def test_two_ranks_gloo_agree_with_single_rank(pkg, O, tmp_path, world, block):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(world, port, block, str(tmp_path)), nprocs=world, join=True)
    cfg, sc = pkg.synth.make_config_scene("C0")
    ref = O.register(sc.src, sc.tgt, threads=1, **cfg.params())
    outs = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    assert sum(int(o["scored"]) for o in outs) == ref["t_eff"]
    for o in outs:
        assert (int(o["count"]), int(o["widx"])) == (ref["best_count"], ref["best_rank"])
        assert o["Rt"].tobytes() == np.concatenate([ref["R"].ravel(), ref["t"]]).astype(np.float32).tobytes()
        assert np.array_equal(o["mask"], ref["mask"])
    want = sum(_fake_hist(r, pkg.SC_HIST_WORDS).view(np.uint32).astype(np.uint64) for r in range(world))
    for o in outs:                                                     # summed histogram, identical on every rank
        assert np.array_equal(o["hist"].view(np.uint32).astype(np.uint64), want & np.uint64(0xFFFFFFFF))


# ---------------------------------------------------------------------------------------------------------
# stages A and B sharded: the collectives' host logic on CPU tensors over gloo, the per-rank
# device work restated with the oracle — bit rows by row block, each rank's own top-T of a contiguous row
# range, blobs in the library's layout, merged in rank order with "ties -> lowest position"
# ---------------------------------------------------------------------------------------------------------
def _rank_rows(n, world):
    """contiguous row ranges (any partition into ascending contiguous ranges is valid for the merge; the library
    balances them by a work estimate on the device)"""
    cuts = [0] + [int(round(n * (1 - (1 - (r + 1) / world) ** 0.5))) for r in range(world - 1)] + [n]
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def _worker_ab(rank, world, port, out_dir, level):
    import sys
    import torch
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import __graft_entry__ as ge
    pkg = ge.load_package(); O = ge.load_oracle()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg, sc = pkg.synth.make_config_scene("C0")
    kw = cfg.params()
    if level < 0:
        kw["max_triangles"] = 6000                      # enough for 1024-entry blobs to be too small
    T = kw["max_triangles"]
    p = pkg.make_params(shard_rank=rank, shard_world=world, **kw)
    plan = pkg.shard_plan(p, cfg.n)                     # needs no GPU
    R, W = plan.rows_per_rank, plan.words_per_row
    S, bits, deg = O.compat(sc.src, sc.tgt, kw["sigma"], kw["t_cmp"], kw["min_len"], kw["tau"])
    # phase 1: this rank's row block of the bit rows into its slice of the shared buffer, then the in-place all-gather
    buf = torch.zeros(plan.bits_bytes_total // 8, dtype=torch.int64)
    r0, r1 = min(rank * R, cfg.n), min(rank * R + R, cfg.n)
    view = buf.numpy().view(np.uint64).reshape(world * R, W)
    view[r0:r1] = bits[r0:r1]
    pkg.shard.allgather_inplace(buf, rank, world)
    assert np.array_equal(view[:cfg.n], bits)
    tri_all, key_all, total = O.triangles(S, bits, deg, 10 ** 9, 0)       # every triangle, ranked
    lo, hi = _rank_rows(cfg.n, world)[rank]
    own = np.nonzero((tri_all[:, 0] >= lo) & (tri_all[:, 0] < hi))[0]     # ranked order restricted to my rows
    retries = 0
    while True:
        # phase 3: own best of a contiguous row range, as a blob in the library's layout; a blob holds `cap` entries
        # (max(2T/world, 4096) << level, at most T rounded up): a list cut below T says so in its header
        plan = pkg.shard_plan(pkg.make_params(shard_rank=rank, shard_world=world, shard_cand_level=level, **kw), cfg.n)
        blob_words = plan.cand_bytes_per_rank // 8
        cap = (plan.cand_bytes_per_rank - 256) // 20
        assert cap % 1024 == 0 and cap <= (T + 1023) // 1024 * 1024
        want_q = min(T, cap)
        mine = own[:want_q]                                                   # my best ...
        kstar_q = int(key_all[mine[-1]]) if len(mine) else 0
        order = np.lexsort((tri_all[mine, 2], tri_all[mine, 1], tri_all[mine, 0]))
        mine = mine[order]                                                    # ... in (i,j,k) order
        cand = torch.zeros(world * blob_words, dtype=torch.int64)
        blob = cand.numpy().view(np.uint8)[rank * blob_words * 8:(rank + 1) * blob_words * 8]
        hdr = blob[:256].view(np.uint64); keys = blob[256:256 + 4 * cap].view(np.uint32)
        recs = blob[256 + 4 * cap:].view(np.uint32).reshape(cap, 4)
        hdr[0], hdr[1], hdr[2] = len(own), len(mine), kstar_q
        hdr[5] = 1 if (want_q < T and len(own) > want_q) else 0
        keys[:len(mine)] = key_all[mine]
        recs[:len(mine), :3] = tri_all[mine]; recs[:len(mine), 3] = key_all[mine]
        pkg.shard.allgather_inplace(cand, rank, world)
        # phase 4: merge — concatenation in rank order, exact threshold, ties to the lowest position
        allk, allt, cut = [], [], []
        for r in range(world):
            b = cand.numpy().view(np.uint8)[r * blob_words * 8:(r + 1) * blob_words * 8]
            h = b[:256].view(np.uint64)
            ns = int(h[1])
            if h[5]:
                cut.append(int(h[2]))
            allk.append(b[256:256 + 4 * cap].view(np.uint32)[:ns].copy())
            allt.append(b[256 + 4 * cap:].view(np.uint32).reshape(cap, 4)[:ns, :3].copy())
        allk = np.concatenate(allk); allt = np.concatenate(allt)
        assert np.all(np.lexsort((allt[:, 2], allt[:, 1], allt[:, 0])) == np.arange(len(allt)))   # global (i,j,k) order
        want = min(T, len(allk))
        kstar = int(np.sort(allk)[::-1][want - 1])
        # the exactness check (merge_check_kernel): every cut list's threshold must lie strictly below the merged one —
        # what such a rank kept back then cannot belong to the global top-T.  Same blobs everywhere: same verdict.
        if all(kstar > kq for kq in cut):
            break
        level += 1; retries += 1
    take = np.nonzero(allk > kstar)[0]
    ties = np.nonzero(allk == kstar)[0][: want - len(take)]
    sel = np.sort(np.concatenate([take, ties]))
    np.savez(os.path.join(out_dir, f"ab{rank}.npz"), tri=allt[sel], key=allk[sel], total=np.int64(total),
             retries=np.int64(retries))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,level", [(2, 0), (3, 0), (2, -8)])
def test_sharded_A_and_B_host_logic_gloo(pkg, O, tmp_path, world, level):
    """Every rank ends with the same selection, and it is the restatement's top-T as a set (in (i,j,k) order) — also when
    the candidate blobs start too small (level -8: 1024 entries) and the ranks have to agree on running again."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker_ab, args=(world, port, str(tmp_path), level), nprocs=world, join=True)
    cfg, sc = pkg.synth.make_config_scene("C0")
    kw = cfg.params()
    if level < 0:
        kw["max_triangles"] = 6000
    S, bits, deg = O.compat(sc.src, sc.tgt, kw["sigma"], kw["t_cmp"], kw["min_len"], kw["tau"])
    tri, key, total = O.triangles(S, bits, deg, kw["max_triangles"], 0)
    order = np.lexsort((tri[:, 2], tri[:, 1], tri[:, 0]))
    outs = [np.load(tmp_path / f"ab{r}.npz") for r in range(world)]
    for o in outs:
        assert np.array_equal(o["tri"], tri[order]) and np.array_equal(o["key"], key[order])
        assert int(o["retries"]) == int(outs[0]["retries"])
    if level < 0:
        assert total > 6000 and int(outs[0]["retries"]) >= 1


# ---------------------------------------------------------------------------------------------------------
# the STREAMED replicated form (what `bench.py --gpus N` runs below 8192 correspondences): shard.ReplicatedStream — two frames in
# flight per rank, one all-gather of the key pairs per frame, SC_EBOUND answered by every rank with the same repeat — over gloo,
# with stand-ins for the two contexts that restate a rank's share of a frame with the CPU restatement
# ---------------------------------------------------------------------------------------------------------
class _StandInContext:
    """What ReplicatedStream needs of a Registrar, on CPU tensors: hypothesize_device (this rank's key pair of the frame),
    finalize_gathered_device(_async) + wait (reduce the gathered pairs, re-solve the winner, mask), debug_last.  `fail`: frames (by
    the number of hypothesize calls with SC_FLAG_EST_BOUND seen so far) whose estimate 'fails': SC_EBOUND from the finalize step."""

    def __init__(self, pkg, O, rank, world, block, kw, fail=()):
        self.pkg, self.O, self.rank, self.world, self.block, self.kw, self.fail = pkg, O, rank, world, block, kw, set(fail)
        self.est_calls, self.pending, self.last_bound = 0, None, 0

    def hypothesize_device(self, src, tgt, n, p, key):
        O, pkg = self.O, self.pkg
        src, tgt = src.numpy(), tgt.numpy()
        est = bool(p.flags & pkg.SC_FLAG_EST_BOUND)
        self.bad = est and self.est_calls in self.fail
        self.est_calls += est
        S, bits, deg = O.compat(src, tgt, self.kw["sigma"], self.kw["t_cmp"], self.kw["min_len"], self.kw["tau"])
        tri, wkey, _ = O.triangles(S, bits, deg, self.kw["max_triangles"], 0)
        mine = pkg.shard.local_indices(len(tri), self.block, self.rank, self.world)
        cnt = O.score(src, tgt, O.kabsch3(src, tgt, tri[mine]), self.kw["tau"])
        best = (0, 0)
        for c, g in zip(cnt, mine):
            best = max(best, pkg.shard.encode_pair(int(c), int(wkey[g]), int(g)))
        key[0], key[1] = best
        self.frame = (src, tgt, tri)
        return {}

    def _finish(self, alls, Rt, mask):
        if self.bad:
            self.last_bound = 2
            return self.pkg.SC_EBOUND, {}
        self.last_bound = 1
        src, tgt, tri = self.frame
        k0, k1 = self.pkg.shard.reduce_pairs([tuple(int(x) for x in alls[2 * r:2 * r + 2]) for r in range(self.world)])
        count, _, widx = self.pkg.shard.decode_pair(k0, k1)
        Rt_w = self.O.kabsch3(src, tgt, tri[widx:widx + 1])[0]
        Rt.copy_(__import__("torch").from_numpy(Rt_w))
        mask.copy_(__import__("torch").from_numpy(self.O.mask(src, tgt, Rt_w, self.kw["tau"])))
        return 0, dict(best_rank=widx, best_count=count)

    def finalize_gathered_device(self, alls, world, Rt, mask):
        return self._finish(alls.clone(), Rt, mask)

    def finalize_gathered_device_async(self, alls, world, Rt, mask):
        self.pending = (alls.clone(), Rt, mask)   # (the gathered pairs as they are NOW: the other context's frame overwrites nothing of this one)

    def wait(self):
        a, self.pending = self.pending, None
        return self._finish(*a)

    def debug_last(self):
        return dict(prune_bound=self.last_bound)


def _worker_stream(rank, world, port, out_dir):
    import sys
    import torch
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import __graft_entry__ as ge
    pkg = ge.load_package(); O = ge.load_oracle()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg, scenes = pkg.synth.make_stream_scenes("C0", 5)
    kw = cfg.params()
    p = pkg.make_params(shard_rank=rank, shard_world=world, shard_block=32, **kw)
    # context 0's second and context 1's first estimating call 'fail' (frames 2 and 1): one repeat each, then — two failed estimates —
    # the stream certifies from there on; the same on every rank, as on the GPU (replicated, deterministic stages)
    pair = [_StandInContext(pkg, O, rank, world, 32, kw, fail=(1,)), _StandInContext(pkg, O, rank, world, 32, kw, fail=(0,))]
    rs = pkg.shard.ReplicatedStream(pkg, pair, cfg.n, p, world, lambda k: torch.zeros(k, dtype=torch.int64), ptr=lambda t: t)
    src = [torch.from_numpy(s.src.copy()) for s in scenes]; tgt = [torch.from_numpy(s.tgt.copy()) for s in scenes]
    frames = 9
    Rt = torch.zeros(frames, 12); mask = torch.zeros(frames, cfg.n, dtype=torch.uint8)
    out = []
    rs.enqueue(0, src[0], tgt[0], Rt[0], mask[0])
    for f in range(1, frames + 1):
        if f < frames:
            k = f % len(scenes)
            rs.enqueue(f, src[k], tgt[k], Rt[f], mask[f])
        rc, st = rs.collect(f - 1)
        out.append((rc, st["best_rank"], st["best_count"]))
    np.savez(os.path.join(out_dir, f"stream{rank}.npz"), out=np.array(out), Rt=Rt.numpy(), mask=mask.numpy(),
             redone=rs.redone, estimate=int(rs.estimate), est_fails=rs.est_fails)
    dist.barrier()
    dist.destroy_process_group()


def test_replicated_ranks_stream_distinct_frames_over_gloo(pkg, O, tmp_path):
    """bench.py's N > 1 default on CPU: two ranks, nine frames over five distinct scenes, two frames in flight per rank, the key
    pairs all-gathered over gloo; two frames come back SC_EBOUND on every rank (a 'failed estimate' each) and are repeated without
    the flag, after which the stream stops estimating.  Every frame equals the single-rank restatement of ITS scene."""
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_worker_stream, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    cfg, scenes = pkg.synth.make_stream_scenes("C0", 5)
    refs = [O.register(s.src, s.tgt, threads=1, **cfg.params()) for s in scenes]
    outs = [np.load(tmp_path / f"stream{r}.npz") for r in range(world)]
    for o in outs:
        assert int(o["redone"]) == 2 and int(o["est_fails"]) == 2 and int(o["estimate"]) == 0
        for f in range(9):
            ref = refs[f % 5]
            assert tuple(o["out"][f]) == (0, ref["best_rank"], ref["best_count"]), (f, o["out"][f])
            assert o["Rt"][f].astype(np.float32).tobytes() == np.concatenate([ref["R"].ravel(), ref["t"]]).astype(np.float32).tobytes()
            assert np.array_equal(o["mask"][f], ref["mask"])


def test_bench_parent_ends_the_ranks_when_one_dies():
    """bench.py --gpus N without a launcher supervises the ranks it starts: rank 1 exits with a code at once,
    the others sleep as if stuck in a rendezvous — the parent must return that code promptly instead of waiting on them."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["SC_BENCH_SUPERVISION_TEST"] = "1"
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, timeout=120)
    assert r.returncode == 7, (r.returncode, r.stderr.decode()[-500:])
    assert time.monotonic() - t0 < 60
