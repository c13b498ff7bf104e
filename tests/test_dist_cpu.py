"""N>1 path on CPU: two gloo ranks exchange their template shards' raw match records with the same all-gather +
host merge the GPU job uses (linemod_pose_estimation_amd/dist.py); the result must equal the unsharded match.
The per-rank records come from the oracle here (tests may use it as the stand-in producer: there is no GPU in this
container); on the GPU box tests/test_gpu_parity.py checks the real sharded contexts."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker_frame_groups(rank, world, port, q):
    """Two gloo ranks as TWO FRAME GROUPS of one template shard each (G = 2, R = 1): rank g produces the records of its slice of
    the batch with frame indices local to the slice; the grouped merge must equal the unsharded per-frame result."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from linemod_pose_estimation_amd import synth
    from linemod_pose_estimation_amd.dist import allgather_blocks, make_block, block_bytes
    from linemod_pose_estimation_amd.detector import merge_gathered
    from oracle import oracle as o
    bank = synth.make_bank(40, seed=31, size_range=(30.0, 70.0), classes=["a", "b"])
    n = 3                                                   # uneven: group 0 takes frame 0, group 1 frames 1 and 2
    frames = [synth.make_scene(bank, 320, 240, seed=32 + f)[0] for f in range(n)]
    det = o.OracleDetector(bank)
    first, last = rank * n // world, (rank + 1) * n // world
    K = 4096
    finals, recs = [], []
    for f, src in enumerate(frames):
        finals.append(det.match(src, 78.0))
        if first <= f < last:
            raw = det.last_raw()
            raw["frame"] = f - first
            recs.append(raw)
    mine = np.concatenate(recs)
    gathered = allgather_blocks(torch.from_numpy(make_block(mine, K)))
    merged = merge_gathered(gathered.contiguous().numpy().reshape(-1), world, block_bytes(K), K, n, frame_groups=world)
    ok = True
    for f in range(n):
        ok = ok and len(merged[f]) == len(finals[f]) and all(np.array_equal(merged[f][k], finals[f][k]) for k in finals[f].dtype.names)
    q.put((rank, bool(ok), len(mine), [len(m) for m in merged]))
    dist.barrier()
    dist.destroy_process_group()


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from linemod_pose_estimation_amd import synth
    from linemod_pose_estimation_amd.dist import allgather_blocks, make_block, block_bytes
    from linemod_pose_estimation_amd.detector import merge_gathered
    from oracle import oracle as o
    bank = synth.make_bank(40, seed=31, size_range=(30.0, 70.0), classes=["a", "b"])
    frames = [synth.make_scene(bank, 320, 240, seed=32 + f)[0] for f in range(2)]
    det = o.OracleDetector(bank)
    shard = bank.shard(rank, world)
    K = 4096
    rec_all = []
    finals = []
    for f, src in enumerate(frames):
        finals.append(det.match(src, 78.0))
        raw = det.last_raw()
        raw["frame"] = f
        keep = np.zeros(len(raw), bool)
        for ci, cid in enumerate(sorted(shard)):
            b, e = shard[cid]
            keep |= (raw["class_index"] == ci) & (raw["template_id"] >= b) & (raw["template_id"] < e)
        rec_all.append(raw[keep])
    mine = np.concatenate(rec_all)
    blk = torch.from_numpy(make_block(mine, K))
    gathered = allgather_blocks(blk)
    counts = gathered[:, :8].contiguous().numpy().view(np.uint32)[:, 1]
    merged = merge_gathered(gathered.contiguous().numpy().reshape(-1), world, block_bytes(K), K, len(frames))
    ok = int(counts[rank]) == len(mine)
    for f in range(len(frames)):
        ok = ok and len(merged[f]) == len(finals[f]) and all(np.array_equal(merged[f][k], finals[f][k]) for k in finals[f].dtype.names)
    q.put((rank, bool(ok), int(counts[rank]), [len(m) for m in merged]))
    dist.barrier()
    dist.destroy_process_group()


def _spawn(target):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = []
    import queue as _q
    import time
    t0 = time.time()
    while len(res) < world and time.time() - t0 < 300:
        try:
            res.append(q.get(timeout=2))
        except _q.Empty:
            if any(p.exitcode not in (None, 0) for p in procs):
                break
    for p in procs:
        p.join(timeout=60)
        if p.is_alive():
            p.terminate()
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert len(res) == world
    return res


def test_two_rank_allgather_merge_equals_unsharded():
    res = _spawn(_worker)
    assert all(r[1] for r in res), res
    assert all(r[2] > 0 for r in res), res          # both shards contributed records
    assert res[0][3] == res[1][3] and sum(res[0][3]) > 0


def test_two_ranks_as_two_frame_groups_merge_equals_unsharded():
    res = _spawn(_worker_frame_groups)
    assert all(r[1] for r in res), res
    assert all(r[2] > 0 for r in res), res          # both frame groups contributed records
    assert res[0][3] == res[1][3] and all(n > 0 for n in res[0][3])


def test_grouped_merge_on_a_grid_of_blocks():
    """lmx_merge_gathered_groups on host-built blocks, no process group: G x R grids (1x4, 2x2, 4x1, 3x2), batches that do not divide evenly and
    batches with fewer frames than frame groups (idle groups contribute empty blocks); every grid must give the unsharded per-frame result."""
    from linemod_pose_estimation_amd import synth
    from linemod_pose_estimation_amd.dist import make_block, block_bytes
    from linemod_pose_estimation_amd.detector import merge_gathered
    from oracle import oracle as o
    bank = synth.make_bank(36, seed=61, size_range=(30.0, 70.0), classes=["a", "b"])
    frames = [synth.make_scene(bank, 320, 240, seed=62 + f)[0] for f in range(5)]
    det = o.OracleDetector(bank)
    finals, raws = [], []
    for src in frames:
        finals.append(det.match(src, 76.0))
        raws.append(det.last_raw().copy())
    assert sum(len(r) for r in raws) > 20
    K = 2048
    for G, R in ((1, 4), (2, 2), (4, 1), (3, 2)):
        for n in (5, 3, 1):
            blocks = []
            for k in range(G * R):
                g, r = k // R, k % R
                first, last = g * n // G, (g + 1) * n // G
                shard = bank.shard(r, R)
                recs = []
                for f in range(first, last):
                    raw = raws[f].copy()
                    raw["frame"] = f - first
                    keep = np.zeros(len(raw), bool)
                    for ci, cid in enumerate(sorted(shard)):
                        b, e = shard[cid]
                        keep |= (raw["class_index"] == ci) & (raw["template_id"] >= b) & (raw["template_id"] < e)
                    recs.append(raw[keep])
                blocks.append(make_block(np.concatenate(recs) if recs else raws[0][:0], K))
            merged = merge_gathered(np.concatenate(blocks), G * R, block_bytes(K), K, n, frame_groups=G)
            for f in range(n):
                assert len(merged[f]) == len(finals[f]) and all(np.array_equal(merged[f][k], finals[f][k]) for k in finals[f].dtype.names), (G, R, n, f)
    import pytest
    from linemod_pose_estimation_amd import _lib
    with pytest.raises(_lib.LmxError, match="frame groups"):
        merge_gathered(np.concatenate(blocks), 6, block_bytes(K), K, 1, frame_groups=4)


def test_shard_ranges_cover_bank():
    from linemod_pose_estimation_amd import synth
    bank = synth.make_bank(11, seed=1, size_range=(20.0, 30.0), classes=["x", "y"])
    for world in (1, 2, 3, 8):
        for cid in ("x", "y"):
            spans = [bank.shard(r, world)[cid] for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == 11
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
