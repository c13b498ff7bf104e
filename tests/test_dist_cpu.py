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


def test_two_rank_allgather_merge_equals_unsharded():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
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
    assert all(r[1] for r in res), res
    assert all(r[2] > 0 for r in res), res          # both shards contributed records
    assert res[0][3] == res[1][3] and sum(res[0][3]) > 0


def test_shard_ranges_cover_bank():
    from linemod_pose_estimation_amd import synth
    bank = synth.make_bank(11, seed=1, size_range=(20.0, 30.0), classes=["x", "y"])
    for world in (1, 2, 3, 8):
        for cid in ("x", "y"):
            spans = [bank.shard(r, world)[cid] for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == 11
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
