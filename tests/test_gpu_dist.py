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


def test_device_group_from_cpp_matches_oracle_and_grows_its_gather_blocks(tmp_path):
    """lmx_group_* (RCCL from C++, no Python in the data path): a C++ program creates a single-process group over the box's
    GPU, matches a batch and prints the matches.  gather_capacity 4 is far too small for the batch, so the first exchange
    reports the true counts in its headers and the group re-allocates and repeats it (two-phase fallback); the results must
    equal the oracle either way.  (World sizes above one need more GPUs than this box has: the driver's multi-GPU run.)"""
    import numpy as np
    from linemod_pose_estimation_amd import NativeBank, _lib, synth
    from oracle import oracle as o
    exe = str(tmp_path / "group_main")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "group_main.cpp"),
                           "-o", exe, "-L", _lib.CSRC, "-llmx", "-Wl,-rpath," + _lib.CSRC, "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    bank = synth.make_bank(70, seed=171, size_range=(30.0, 80.0), classes=["a", "b"])
    yml = tmp_path / "bank.yml"
    NativeBank.from_bank(bank).save_yaml(yml)
    frames = [synth.make_scene(bank, 320, 240, seed=172 + f)[0] for f in range(3)]
    (tmp_path / "frames.raw").write_bytes(b"".join(np.ascontiguousarray(fr[0]).tobytes() + np.ascontiguousarray(fr[1]).tobytes() for fr in frames))
    od = o.OracleDetector(bank)
    refs = [od.match(fr, 77.0) for fr in frames]
    assert sum(len(r) for r in refs) > 8
    for gather_capacity in (4, 8192):
        res = subprocess.run([exe, str(yml), "1", str(gather_capacity), "320", "240", "77", "3", str(tmp_path / "frames.raw")], capture_output=True, text=True)
        assert res.returncode == 0, res.stderr
        lines = res.stdout.strip().splitlines()
        while not lines[0].startswith("group of"):       # RCCL prints its version banner on stdout
            lines.pop(0)
        cap = int(lines[0].split()[-1])
        assert lines[0].startswith("group of 1, gather capacity") and cap >= gather_capacity
        if gather_capacity == 4:
            assert cap >= max(len(r) for r in refs) and cap > 4        # grown to hold every record of the batch
        got = {f: [] for f in range(3)}
        for l in lines[1:]:
            head, rest = l.split(":")
            got[int(head.split()[1])].append(rest.split())
        for f in range(3):
            assert len(got[f]) == len(refs[f])
            for g, r in zip(got[f], refs[f]):
                assert (int(g[0]), int(g[1]), int(g[3]), int(g[4])) == (r["x"], r["y"], r["class_index"], r["template_id"])
                assert np.float32(float(g[2])) == r["similarity"]
