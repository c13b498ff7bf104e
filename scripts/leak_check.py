"""Resource check on a GPU box: device memory and host RSS after repeated context creation / destruction, after thousands of pipelined
steps, host-frame uploads, single calls, device groups and bank loads: everything must come back to where it started.
usage: python scripts/leak_check.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
from linemod_pose_estimation_amd import synth, Detector  # noqa: E402
from linemod_pose_estimation_amd.dist import DeviceGroup  # noqa: E402


def rss_mb():
    with open("/proc/self/status") as f:
        for line in f:
            if line.startswith("VmRSS"):
                return int(line.split()[1]) / 1024.0
    return 0.0


def dev_used_mb():
    free, total = torch.cuda.mem_get_info()
    return (total - free) / 2 ** 20


def main():
    bank = synth.make_bank(600, seed=11)
    frames = [synth.make_scene(bank, 640, 480, seed=20 + f, row_pad=0)[0] for f in range(16)]
    host = [[np.array(s, copy=True) for s in fr] for fr in frames]

    def cycle():
        det = Detector(bank, 640, 480, device=0, max_batch=16, overlap=True)
        det.upload(frames)
        bench.run_pipelined(det, 40, 16, 90.0)
        bench.run_pipelined(det, 20, 16, 90.0, uploads=[Detector.prepare_batch(host)])
        det.match(host[0], 90.0)
        det.close()
        g = DeviceGroup(bank, 640, 480, 3, devices=[0, 0, 0], max_batch=4, collective="peer_copy")
        g.upload(host[:4])
        g.submit(4, 90.0)
        g.finish(4)
        g.close()

    for _ in range(3):
        cycle()
    torch.cuda.synchronize()
    r0, d0 = rss_mb(), dev_used_mb()
    for i in range(40):
        cycle()
    torch.cuda.synchronize()
    r1, d1 = rss_mb(), dev_used_mb()
    print("40 create/use/destroy cycles: host RSS %.0f -> %.0f MB, device memory in use %.0f -> %.0f MB" % (r0, r1, d0, d1))
    det = Detector(bank, 640, 480, device=0, max_batch=16, overlap=True)
    det.upload(frames)
    bench.run_pipelined(det, 200, 16, 90.0)
    r2, d2 = rss_mb(), dev_used_mb()
    bench.run_pipelined(det, 6000, 16, 90.0)
    for i in range(400):
        det.match(host[i % 16], 90.0)
    r3, d3 = rss_mb(), dev_used_mb()
    print("6000 pipelined steps + 400 single calls on one context: host RSS %.0f -> %.0f MB, device %.0f -> %.0f MB" % (r2, r3, d2, d3))
    det.close()
    ok = abs(r1 - r0) < 64 and abs(d1 - d0) < 64 and abs(r3 - r2) < 32 and abs(d3 - d2) < 8
    print("leak check", "ok" if ok else "FAILED")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
