"""The reference's per-request flow through the C++ `cv::linemod` facade (tests/cpp/cv_facade_main.cpp `requests`): every request calls
readLinemod(<templates.yml>) -- cv::FileStorage, Detector::read, readClass per class -- into a NEW detector and matches one frame, as
src/linemod_ensenso_detect_3_mult_detect_service.cpp:1784-1786 does.  Builds the caller with g++, writes a 3000-template bank as yml and
one frame as raw files, prints the median milliseconds of readLinemod and of match().  usage (GPU box): python scripts/cpp_request_flow.py [n_templates]"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from linemod_pose_estimation_amd import _lib, synth, NativeBank  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    tmp = tempfile.mkdtemp(prefix="lmx_cpp_requests_")
    exe = os.path.join(tmp, "cv_facade_main")
    subprocess.check_call(["g++", "-std=c++11", "-O2", "-I", os.path.join(ROOT, "tests", "cpp", "cv_standin"), "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "cv_facade_main.cpp"), "-o", exe, "-pthread", "-L", _lib.CSRC, "-llmx", "-Wl,-rpath," + _lib.CSRC,
                           "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    for mods in (("ColorGradient",), ("ColorGradient", "DepthNormal")):
        bank = synth.make_bank(n, modalities=mods, seed=20250214)
        yml = os.path.join(tmp, "templates_%d.yml" % len(mods))
        NativeBank.from_bank(bank).save_yaml(yml)
        src = synth.make_scene(bank, 640, 480, seed=4000, row_pad=0)[0]
        files = []
        for i, a in enumerate(src):
            f = os.path.join(tmp, "src%d_%d.raw" % (len(mods), i))
            np.ascontiguousarray(a).tofile(f)
            files.append(f)
        env = dict(os.environ)
        env.setdefault("LMX_NORMAL_LUT", "")
        for mode in ("requests", "requests_cached"):
            res = subprocess.run([exe, mode, yml, "640", "480", "92", "30"] + files, capture_output=True, text=True, env=env)
            print("%d templates, %s (%.1f MB of yml), %s: %s" % (n, "+".join(mods), os.path.getsize(yml) / 1e6,
                                                                 "readLinemod via FileStorage" if mode == "requests" else "Detector::load (cached)",
                                                                 (res.stdout.strip().replace("\n", " | ") or res.stderr.strip()[-300:])))


if __name__ == "__main__":
    main()
