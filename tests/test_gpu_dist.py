"""ShardedMatcher (linemod_pose_estimation_amd/dist.py) on the GPU box: the RCCL path with a one-rank process group
(the box has one GPU; N = 2,4,8 are the driver's to launch) and a two-rank job that shares the card with gloo as
the transport -- both must reproduce the oracle's unsharded result."""
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["LMX_ROOT"])
import numpy as np, torch, torch.distributed as dist
from linemod_pose_estimation_amd import synth
from linemod_pose_estimation_amd.dist import ShardedMatcher
from oracle import oracle as o
backend = os.environ["LMX_BACKEND"]
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
if backend == "nccl":
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
else:
    dist.init_process_group("gloo")
bank = synth.make_bank(70, seed=71, size_range=(30.0, 80.0), classes=["a", "b"])
frames = [synth.make_scene(bank, 320, 240, seed=72 + f)[0] for f in range(3)]
sm = ShardedMatcher(bank, 320, 240, max_batch=3, gather_capacity=8192)
sm.upload(frames)
for rep in range(2):
    outs = sm.step(3, 77.0)
# pipelined: as many batches in flight as the matcher allows, different batch sizes and thresholds, results oldest first
plan = [(3, 77.0), (2, 85.0), (1, 77.0), (3, 90.0), (3, 77.0), (2, 77.0)]
got, queued = [], 0
for n, thr in plan:
    if queued == sm.depth:
        got.append(sm.finish()); queued -= 1
    sm.submit(n, thr); queued += 1
while queued:
    got.append(sm.finish()); queued -= 1
od = o.OracleDetector(bank)
for (n, thr), res in zip(plan, got):
    assert len(res) == n
    for f in range(n):
        ref = od.match(frames[f], thr)
        assert len(res[f]) == len(ref), (n, thr, f, len(res[f]), len(ref))
        for k in ref.dtype.names:
            assert np.array_equal(res[f][k], ref[k]), (n, thr, f, k)
for f in range(3):
    ref = od.match(frames[f], 77.0)
    assert len(ref) > 5 and len(outs[f]) == len(ref), (f, len(outs[f]), len(ref))
    for k in ref.dtype.names:
        assert np.array_equal(outs[f][k], ref[k]), (f, k)
# gather capacity far too small: every rank sees the same headers, regrows and repeats the exchange; only rank 0 merges
sm2 = ShardedMatcher(bank, 320, 240, max_batch=3, gather_capacity=16, result_ranks=(0,))
sm2.upload(frames)
got, queued = [], 0
for n, thr in plan:
    if queued == sm2.depth:
        got.append(sm2.finish()); queued -= 1
    sm2.submit(n, thr); queued += 1
while queued:
    got.append(sm2.finish()); queued -= 1
assert sm2.regrows >= 1 and sm2.capacity > 16, (sm2.regrows, sm2.capacity)
for (n, thr), res in zip(plan, got):
    if rank != 0:
        assert res is None
        continue
    for f in range(n):
        ref = od.match(frames[f], thr)
        assert len(res[f]) == len(ref), (n, thr, f, len(res[f]), len(ref))
        for k in ref.dtype.names:
            assert np.array_equal(res[f][k], ref[k]), (n, thr, f, k)
# the ranks as frame groups (G x R = world x 1 and, for two ranks, also 1 x 2 above): every rank holds the whole bank and takes its share of
# each batch; batches that do not divide evenly, a batch with fewer frames than groups (idle ranks export an empty block), a prepared batch
for G in sorted({world, 1}):
    sm3 = ShardedMatcher(bank, 320, 240, max_batch=3, gather_capacity=16 if G > 1 else 8192, frame_groups=G)
    assert (sm3.G, sm3.R) == (G, world // G) and sm3.det.max_batch == (3 + G - 1) // G
    from linemod_pose_estimation_amd import Detector
    plan3 = [(3, 77.0), (1, 77.0), (2, 85.0), (3, 90.0), (3, 77.0), (1, 85.0), (2, 77.0), (3, 77.0)]
    got, queued = [], 0
    for i, (n, thr) in enumerate(plan3):
        if queued == sm3.depth:
            got.append(sm3.finish()); queued -= 1
        sm3.upload(Detector.prepare_batch(frames[:n]) if i % 2 else frames[:n])
        sm3.submit(n, thr); queued += 1
    while queued:
        got.append(sm3.finish()); queued -= 1
    for (n, thr), res in zip(plan3, got):
        assert len(res) == n
        for f in range(n):
            ref = od.match(frames[f], thr)
            assert len(res[f]) == len(ref), (G, n, thr, f, len(res[f]), len(ref))
            for k in ref.dtype.names:
                assert np.array_equal(res[f][k], ref[k]), (G, n, thr, f, k)
    if G > 1:
        assert sm3.regrows >= 1
        st = sm3.det.stats()
        assert st["candidates"] > 0        # the release path feeds lmx_ctx_stats too (it reported 0 for sharded callers before)
dist.barrier()
dist.destroy_process_group()
print("RANK%d OK" % rank)
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, backend, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   LMX_ROOT=ROOT, LMX_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o_) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and ("RANK%d OK" % r) in o_, o_[-2000:]


def test_one_rank_rccl(tmp_path):
    _run(1, "nccl", tmp_path)


def test_two_ranks_one_gpu_gloo(tmp_path):
    _run(2, "gloo", tmp_path)


# ---- lmx_group_* from C++ (tests/cpp/group_main.cpp), no Python in the data path ------------------------------------------------
_GROUP = {}


def _group_fixture(tmp_path_factory):
    """group_main built once, the two-class bank as yml, three frames as raw bytes, the oracle's answers per (frame, threshold)."""
    if _GROUP:
        return _GROUP
    import numpy as np
    from linemod_pose_estimation_amd import NativeBank, _lib, synth
    from oracle import oracle as o
    d = tmp_path_factory.mktemp("group")
    exe = str(d / "group_main")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "group_main.cpp"),
                           "-o", exe, "-L", _lib.CSRC, "-llmx", "-Wl,-rpath," + _lib.CSRC, "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    bank = synth.make_bank(70, seed=171, size_range=(30.0, 80.0), classes=["a", "b"])
    yml = d / "bank.yml"
    NativeBank.from_bank(bank).save_yaml(yml)
    frames = [synth.make_scene(bank, 320, 240, seed=172 + f)[0] for f in range(3)]
    (d / "frames.raw").write_bytes(b"".join(np.ascontiguousarray(fr[0]).tobytes() + np.ascontiguousarray(fr[1]).tobytes() for fr in frames))
    od = o.OracleDetector(bank)
    refs, raw_tids = {}, {}
    for thr in (77.0, 85.0):
        for f in range(3):
            refs[(f, thr)] = od.match(frames[f], thr)
            raw_tids[(f, thr)] = od.last_raw()["template_id"].copy()
    _GROUP.update(exe=exe, yml=str(yml), frames=str(d / "frames.raw"), refs=refs, raw_tids=raw_tids, n_per_class=70)
    return _GROUP


def _records_per_rank(fx, world, frames_thr):
    """Raw (pre-unique) records every rank of a `world`-way shard produces for the given (frame, threshold) pairs."""
    n = fx["n_per_class"]
    counts = [0] * world
    for key in frames_thr:
        for t in fx["raw_tids"][key]:
            for r in range(world):
                if (r * n) // world <= int(t) < ((r + 1) * n) // world:
                    counts[r] += 1
    return counts


def _run_group(fx, members, capacity, collective, devices, mode, max_candidates=0, frame_groups=0):
    return subprocess.run([fx["exe"], fx["yml"], str(members), str(capacity), "320", "240", "77", "3", fx["frames"], collective, devices, mode, str(max_candidates),
                           str(frame_groups)], capture_output=True, text=True, timeout=600)


def _parse_group(stdout):
    import numpy as np
    lines = [l for l in stdout.strip().splitlines() if l.startswith("batch ") or l.startswith("group of")]   # RCCL prints a banner on stdout
    assert lines and lines[-1].startswith("group of"), stdout[-2000:]
    got = {}
    for l in lines[:-1]:
        head, rest = l.split(":")
        _, b, _, f = head.split()
        v = rest.split()
        got.setdefault((int(b), int(f)), []).append((int(v[0]), int(v[1]), np.float32(float(v[2])), int(v[3]), int(v[4])))
    words = lines[-1].replace(",", "").split()
    return got, {"size": int(words[2]), "collective": words[4], "depth": int(words[6]), "frame_groups": int(words[9]), "capacity": int(words[-1])}


def _check_frame(got, ref):
    assert len(got) == len(ref), (len(got), len(ref))
    for g, r in zip(got, ref):
        assert g == (r["x"], r["y"], r["similarity"], r["class_index"], r["template_id"]), (g, r)


def test_device_group_from_cpp_one_member_rccl_grows_its_gather_blocks(tmp_path_factory):
    """RCCL from C++ with the one GPU the box has: gather_capacity 4 is far too small for the batch, so the first exchange reports the
    true counts in its headers and the group re-allocates and repeats it (two-phase fallback); results equal the oracle either way."""
    fx = _group_fixture(tmp_path_factory)
    need = _records_per_rank(fx, 1, [(f, 77.0) for f in range(3)])[0]
    for capacity in (4, 8192):
        res = _run_group(fx, 1, capacity, "rccl", "distinct", "batch")
        assert res.returncode == 0, res.stderr[-2000:]
        got, info = _parse_group(res.stdout)
        assert info["size"] == 1 and info["collective"] == "rccl"
        assert info["capacity"] >= (need if capacity == 4 else capacity)
        for b in range(2):
            for f in range(3):
                _check_frame(got.get((b, f), []), fx["refs"][(f, 77.0)])


@pytest.mark.parametrize("members,capacity", [(2, 512), (3, 512), (8, 256), (2, 4), (8, 8192)])
def test_device_group_world_gt_1_on_one_gpu(tmp_path_factory, members, capacity):
    """lmx_group_* with 2, 3 and 8 members that SHARE device 0 (devices[] repeats the id; RCCL refuses that, the peer-copy collective
    does not): ncclCommInitAll aside, everything a multi-GPU group does runs here -- per-member contexts with shard_rank = i, one
    staging copy fanned out to every member, per-member enqueue / export on the host thread pool, the all-gather, the merge in rank
    order.  Capacities are chosen so that rank 0's block FITS while a higher rank's does not: the regrow is triggered by a non-zero
    rank's header (the test checks that premise from the oracle's raw records)."""
    fx = _group_fixture(tmp_path_factory)
    per_rank = _records_per_rank(fx, members, [(f, 77.0) for f in range(3)])
    if capacity in (256, 512):
        assert per_rank[0] <= capacity < max(per_rank[1:]), per_rank
    res = _run_group(fx, members, capacity, "peer", "same", "batch")
    assert res.returncode == 0, res.stderr[-2000:]
    got, info = _parse_group(res.stdout)
    assert info["size"] == members and info["collective"] == "peer_copy"
    assert info["capacity"] >= max(per_rank) if capacity < max(per_rank) else info["capacity"] == capacity
    for b in range(2):
        for f in range(3):
            _check_frame(got.get((b, f), []), fx["refs"][(f, 77.0)])


@pytest.mark.parametrize("members,capacity", [(3, 256), (8, 8192)])
def test_device_group_split_phase_pipeline(tmp_path_factory, members, capacity):
    """upload / submit / finish with the group's full depth in flight (device lanes on: six batches), different frames, batch sizes and
    thresholds per batch.  With capacity 256 some batches overflow their blocks while later batches are already queued: the batch is
    re-exchanged from the members' output slots (lmx_ctx_export_oldest_on) and the other ring entries grow when they come round."""
    fx = _group_fixture(tmp_path_factory)
    res = _run_group(fx, members, capacity, "peer", "same", "pipeline")
    assert res.returncode == 0, res.stderr[-2000:]
    got, info = _parse_group(res.stdout)
    assert info["depth"] == 6 and info["size"] == members
    n_batches = 3 * info["depth"] + 1
    worst = 0
    for b in range(n_batches):
        n, thr = 3 - (b % 3), 77.0 + 8.0 * (b % 2)
        worst = max(worst, max(_records_per_rank(fx, members, [((i + b) % 3, thr) for i in range(n)])))
        for i in range(n):
            _check_frame(got.get((b, i), []), fx["refs"][((i + b) % 3, thr)])
    if capacity == 256:
        assert worst > 256 and info["capacity"] >= worst
    else:
        assert info["capacity"] == capacity


def test_device_group_refusals(tmp_path_factory):
    """A member whose CANDIDATE list overflowed cannot be repaired by a larger gather block (the scoring kernel dropped candidates): the
    group reports LMX_ERR_OVERFLOW instead of an incomplete result.  RCCL with a repeated device id is refused with a pointer to the
    peer-copy collective."""
    fx = _group_fixture(tmp_path_factory)
    res = _run_group(fx, 3, 8192, "peer", "same", "batch", max_candidates=8)
    assert res.returncode == 1 and "candidate list overflow" in res.stderr, (res.returncode, res.stderr[-1000:])
    res = _run_group(fx, 2, 8192, "rccl", "same", "batch")
    assert res.returncode == 1 and "appears twice" in res.stderr, (res.returncode, res.stderr[-1000:])


def test_device_group_python_wrapper_four_members_share_the_gpu():
    """dist.DeviceGroup (ctypes over lmx_group_*): four members on device 0, pipelined, against the oracle."""
    import numpy as np
    from linemod_pose_estimation_amd import synth
    from linemod_pose_estimation_amd.dist import DeviceGroup
    from oracle import oracle as o
    bank = synth.make_bank(50, seed=271, size_range=(30.0, 80.0))
    frames = [synth.make_scene(bank, 320, 240, seed=272 + f)[0] for f in range(4)]
    od = o.OracleDetector(bank)
    refs = [od.match(fr, 80.0) for fr in frames]
    g = DeviceGroup(bank, 320, 240, 4, devices=[0, 0, 0, 0], max_batch=4, gather_capacity=64, collective="peer_copy")
    assert g.size == 4 and g.collective == "peer_copy" and g.depth == 6
    outs, queued = [], 0
    for b in range(10):
        if queued == g.depth:
            outs.append(g.finish(4)); queued -= 1
        g.upload(frames[b % 4:] + frames[:b % 4])
        g.submit(4, 80.0); queued += 1
    while queued:
        outs.append(g.finish(4)); queued -= 1
    for b, res in enumerate(outs):
        for i in range(4):
            ref = refs[(i + b) % 4]
            assert len(res[i]) == len(ref) and all(np.array_equal(res[i][k], ref[k]) for k in ref.dtype.names), (b, i)
    g.close()


def test_device_group_misuse_is_reported():
    """Order of calls: submit before any upload, more submits than the depth, finish with nothing in flight, a batch size that
    differs from the submitted one -- each an error with a message, none a crash; the group keeps working afterwards."""
    import numpy as np
    from linemod_pose_estimation_amd import synth, _lib
    from linemod_pose_estimation_amd.dist import DeviceGroup
    from oracle import oracle as o
    bank = synth.make_bank(20, seed=371, size_range=(30.0, 70.0))
    frames = [synth.make_scene(bank, 320, 240, seed=372 + f)[0] for f in range(2)]
    g = DeviceGroup(bank, 320, 240, 2, devices=[0, 0], max_batch=2, collective="peer_copy", overlap=False)
    assert g.depth == 2
    with pytest.raises(_lib.LmxError, match="nothing uploaded"):
        g.submit(2, 80.0)
    with pytest.raises(_lib.LmxError, match="nothing submitted"):
        g.finish(2)
    g.upload(frames)
    g.submit(2, 80.0)
    g.submit(1, 80.0)
    with pytest.raises(_lib.LmxError, match="already in flight"):
        g.submit(2, 80.0)
    with pytest.raises(_lib.LmxError, match="submitted with 2"):
        g.finish(1)
    od = o.OracleDetector(bank)
    a, b = g.finish(2), g.finish(1)
    for f in range(2):
        ref = od.match(frames[f], 80.0)
        assert len(a[f]) == len(ref) and all(np.array_equal(a[f][k], ref[k]) for k in ref.dtype.names)
    ref = od.match(frames[0], 80.0)
    assert len(b[0]) == len(ref) and all(np.array_equal(b[0][k], ref[k]) for k in ref.dtype.names)
    with pytest.raises(_lib.LmxError, match="rccl|RCCL|appears twice"):
        DeviceGroup(bank, 320, 240, 2, devices=[0, 0], max_batch=2, collective="rccl")
    g.close()


@pytest.mark.parametrize("members,frame_groups,capacity,mode", [(2, 2, 8192, "batch"), (4, 2, 64, "pipeline"), (3, 3, 8192, "pipeline"), (8, 4, 32, "pipeline"), (8, 8, 8192, "batch")])
def test_device_group_frame_groups_from_cpp_on_one_gpu(tmp_path_factory, members, frame_groups, capacity, mode):
    """the G x R member grid of lmx_group_* from C++ (tests/cpp/group_main.cpp), members sharing device 0.  Member (g, r)
    takes frames [g*n/G, (g+1)*n/G) and template shard r of R; the pipeline mode submits batches of 3, 2, 1 frames, so with G = 3, 4 or 8
    some frame groups are idle in some batches (their members enqueue nothing and contribute an empty block), batches divide unevenly,
    and small gather capacities make blocks regrow while later batches are queued.  Every frame of every batch equals the oracle."""
    fx = _group_fixture(tmp_path_factory)
    res = _run_group(fx, members, capacity, "peer", "same", mode, frame_groups=frame_groups)
    assert res.returncode == 0, res.stderr[-2000:]
    got, info = _parse_group(res.stdout)
    assert info["size"] == members and info["frame_groups"] == frame_groups
    if mode == "batch":
        for b in range(2):
            for f in range(3):
                _check_frame(got.get((b, f), []), fx["refs"][(f, 77.0)])
    else:
        for b in range(3 * info["depth"] + 1):
            n, thr = 3 - (b % 3), 77.0 + 8.0 * (b % 2)
            for i in range(n):
                _check_frame(got.get((b, i), []), fx["refs"][((i + b) % 3, thr)])
    if capacity < 8192:
        assert info["capacity"] > capacity        # a member's block overflowed and the exchange was repeated
    # a grid that does not exist is refused, not rounded
    bad = _run_group(fx, 3, 8192, "peer", "same", "batch", frame_groups=2)
    assert bad.returncode == 1 and "frame groups" in bad.stderr, (bad.returncode, bad.stderr[-500:])


def test_device_group_frame_groups_python_wrapper_and_refusals():
    """dist.DeviceGroup with frame_groups: 4 members as 2 x 2 and as 4 x 1 on device 0, five frames per batch (uneven split), hipGraph members;
    a submit for fewer frames than were uploaded is refused when the frames were dealt to groups at upload time."""
    import numpy as np
    from linemod_pose_estimation_amd import synth, _lib
    from linemod_pose_estimation_amd.dist import DeviceGroup
    from oracle import oracle as o
    bank = synth.make_bank(50, seed=471, size_range=(30.0, 80.0), classes=["a", "b"])
    frames = [synth.make_scene(bank, 320, 240, seed=472 + f)[0] for f in range(5)]
    od = o.OracleDetector(bank)
    refs = [od.match(fr, 80.0) for fr in frames]
    assert sum(len(r) for r in refs) > 10
    for G, hipgraph in ((2, False), (4, True)):
        g = DeviceGroup(bank, 320, 240, 4, devices=[0] * 4, max_batch=5, gather_capacity=64, collective="peer_copy", frame_groups=G, hipgraph=hipgraph)
        assert g.size == 4 and g.frame_groups == G
        outs, queued = [], 0
        for b in range(9):
            if queued == g.depth:
                outs.append(g.finish(5)); queued -= 1
            g.upload(frames[b % 5:] + frames[:b % 5])
            g.submit(5, 80.0); queued += 1
        while queued:
            outs.append(g.finish(5)); queued -= 1
        for b, res in enumerate(outs):
            for i in range(5):
                ref = refs[(i + b) % 5]
                assert len(res[i]) == len(ref) and all(np.array_equal(res[i][k], ref[k]) for k in ref.dtype.names), (G, b, i)
        g.upload(frames)
        with pytest.raises(_lib.LmxError, match="whole upload"):
            g.submit(3, 80.0)
        g.close()


def test_configs3_assembled_50k_bank_eight_members_equal_the_whole_bank_oracle():
    """VERDICT r3 item 2, BASELINE configs[3] assembled once: the 50 000-template bank over EIGHT members (sharing device 0: peer-copy
    collective), 640x480, 2 frames, threshold 92 -- every shard is scored by its own context, the eight blocks are exchanged and merged, and
    the result equals OracleDetector(bank).match for the WHOLE bank, matches and order.  Gather capacity 2: rank 0 (no records in these
    frames) fits, ranks 1, 3 and 5 (three records each) do not, so the regrow is decided by a non-zero rank's header.  Then the same bank
    as 8 frame groups x 1 shard (configs[4]'s decomposition: every member holds all 50 000 templates, eight frames are dealt one per member)
    and as a 2 x 4 grid, at threshold 88 (hundreds of records)."""
    import numpy as np
    from golden_util import bank_50k
    from linemod_pose_estimation_amd import synth
    from linemod_pose_estimation_amd.dist import DeviceGroup
    from oracle import oracle as o
    bank = bank_50k()
    frames = [synth.make_scene(bank, 640, 480, seed=6000 + f, row_pad=0, texture=0.6)[0] for f in range(8)]
    od = o.OracleDetector(bank)

    def check(res, fr, thr):
        for f, src in enumerate(fr):
            ref = od.match(src, thr)
            assert len(res[f]) == len(ref), (f, len(res[f]), len(ref))
            for k in ref.dtype.names:
                assert np.array_equal(res[f][k], ref[k]), (f, k)
    # configs[3]: 1 x 8, two frames (seeds 6001, 6002), threshold 92
    two = frames[1:3]
    per_rank = np.zeros(8, int)
    for src in two:
        od.match(src, 92.0)
        per_rank += np.bincount(od.last_raw()["template_id"] * 8 // 50000, minlength=8)
    assert per_rank[0] <= 2 < per_rank[1:].max(), per_rank
    g = DeviceGroup(bank, 640, 480, 8, devices=[0] * 8, max_batch=2, gather_capacity=2, collective="peer_copy")
    assert g.size == 8 and g.frame_groups == 1
    g.upload(two)
    g.submit(2, 92.0)
    res = g.finish(2)
    assert sum(len(r) for r in res) >= 6
    check(res, two, 92.0)
    assert g.gather_capacity() >= per_rank.max() > 2
    g.close()
    # configs[4]'s decomposition: 8 x 1 on eight frames, and 2 x 4; threshold 88 as well (12-36 matches per frame)
    for G, members in ((8, 8), (2, 8)):
        g = DeviceGroup(bank, 640, 480, members, devices=[0] * members, max_batch=8, gather_capacity=16, collective="peer_copy", frame_groups=G, overlap=False)
        for thr in (92.0, 88.0):
            g.upload(frames)
            g.submit(8, thr)
            check(g.finish(8), frames, thr)
        g.close()


WORKER_50K = r'''
import os, sys
sys.path.insert(0, os.environ["LMX_ROOT"])
sys.path.insert(0, os.path.join(os.environ["LMX_ROOT"], "tests"))
import numpy as np, torch, torch.distributed as dist
from linemod_pose_estimation_amd import synth
from linemod_pose_estimation_amd.dist import ShardedMatcher
from oracle import oracle as o
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
bank = synth.make_bank(50000, seed=20250217)
frames = [synth.make_scene(bank, 640, 480, seed=6000 + f, row_pad=0, texture=0.6)[0] for f in (1, 2)]
od = o.OracleDetector(bank)
refs = {thr: [od.match(fr, thr) for fr in frames] for thr in (92.0, 88.0)}
for G in (1, 2):
    sm = ShardedMatcher(bank, 640, 480, max_batch=2, gather_capacity=2, frame_groups=G)
    sm.upload(frames)
    for thr in (92.0, 88.0):
        res = sm.step(2, thr)
        for f in range(2):
            ref = refs[thr][f]
            assert len(res[f]) == len(ref), (G, thr, f, len(res[f]), len(ref))
            for k in ref.dtype.names:
                assert np.array_equal(res[f][k], ref[k]), (G, thr, f, k)
    assert sm.regrows >= 1
    del sm
dist.barrier()
dist.destroy_process_group()
print("RANK%d OK" % rank)
'''


def test_configs3_two_gloo_ranks_share_the_gpu_on_the_50k_bank(tmp_path):
    """The same assembly through the one-process-per-GPU front end: ShardedMatcher under gloo, two ranks on the one GPU, the 50 000-template
    bank as two template shards (G = 1) and as two frame groups (G = 2); merged result == whole-bank oracle on every rank."""
    script = tmp_path / "worker50k.py"
    script.write_text(WORKER_50K)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LMX_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, (p, o_) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and ("RANK%d OK" % r) in o_, o_[-2000:]
