#!/usr/bin/env python3
"""Timing experiments: build copies of liblmx.so with one stage of a kernel compiled out (results are WRONG; only the
kernel time is of interest) into variants/.  usage: build_variants.py color|depth|spread|refine|b1half|score|wpb|dqunroll  (kernel-side experiments only: lmx_kernels.hip)"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cs = os.path.join(root, "linemod_pose_estimation_amd", "csrc")
src = open(os.path.join(cs, "lmx_kernels.hip")).read()
which = sys.argv[1]
if which == "color":
    reps = [("  // A\n  if (x0 >= 5", "  // A\n  if (!(LMX_EXP_SKIP & 1)) if (x0 >= 5"),
            ("  if (pyr_dst != nullptr) {\n    const int Hd", "  if (pyr_dst != nullptr && !(LMX_EXP_SKIP & 2)) {\n    const int Hd"),
            ("  if (tid < 3 * (IS / 4) * (SH / 5)) {", "  if (!(LMX_EXP_SKIP & 4)) if (tid < 3 * (IS / 4) * (SH / 5)) {"),
            ("  for (int i = tid; i < SH * (SW / 2); i += 256) {", "  if (!(LMX_EXP_SKIP & 8)) for (int i = tid; i < SH * (SW / 2); i += 256) {"),
            ("    if (x0 >= 2 && x0 + CQ_TW + 2 <= W && y0 >= 2 && y0 + CQ_TH + 2 <= H) stage_d(std::true_type{});\n    else stage_d(std::false_type{});",
             "    if (LMX_EXP_SKIP & 16) {} else if (x0 >= 2 && x0 + CQ_TW + 2 <= W && y0 >= 2 && y0 + CQ_TH + 2 <= H) stage_d(std::true_type{});\n    else stage_d(std::false_type{});"),
            ("  // E\n  {", "  // E\n  if (!(LMX_EXP_SKIP & 32)) {")]
    names = {"NONE": 0, "A": 1, "P": 2, "B": 4, "C": 8, "D": 16, "E": 32}
elif which == "depth":
    reps = [("  IntT f[8], md[8];\n  const IntT thr = difference_threshold;", "  if (LMX_EXP_SKIP & 1) return (int)((dl[0] + dl[1] + dl[2] + dl[3] + dl[4] + dl[5] + dl[6] + dl[7]) & 7) + 1;\n  IntT f[8], md[8];\n  const IntT thr = difference_threshold;"),
            ("  float s = sqrtf(nx * nx + ny * ny + nz * nz);\n  if (!(s > 0)) return 0;\n  float inv = 1.0f / s;",
             "  float s = (LMX_EXP_SKIP & 2) ? (nx * nx + ny * ny + nz * nz) : sqrtf(nx * nx + ny * ny + nz * nz);\n  if (!(s > 0)) return 0;\n  float inv = (LMX_EXP_SKIP & 2) ? __builtin_amdgcn_rsqf(s) : 1.0f / s;"),
            ("    unsigned long long p = cnt;\n    p += p << 6; p += p << 12; p += p << 24; p += p << 48;\n    const int med = 9 - __popcll(((p + 19ull * ONES) >> 5) & ONES);",
             "    unsigned long long p = cnt;\n    if (!(LMX_EXP_SKIP & 4)) { p += p << 6; p += p << 12; p += p << 24; p += p << 48; }\n    const int med = (LMX_EXP_SKIP & 8) ? (int)(s_oh[seg * RPS + j + 2][lx + 2] >> 7) & 7 : 9 - __popcll(((p + 19ull * ONES) >> 5) & ONES);")]
    names = {"NONE": 0, "NOLSQ": 1, "FASTNORM": 2, "NOPREFIX": 4, "NOMEDIAN": 12}
elif which == "spread":
    reps = [("  for (int j = tid; j < Wd; j += 256) {\n    uint32_t d[RI];", "  if (!(LMX_EXP_SKIP & 1)) for (int j = tid; j < Wd; j += 256) {\n    uint32_t d[RI];"),
            ("  for (int i = tid; i < T * W4; i += 256) {\n    int ly = i / W4, j = i - ly * W4;\n    const uint32_t* p = s_v", "  if (!(LMX_EXP_SKIP & 2)) for (int i = tid; i < T * W4; i += 256) {\n    int ly = i / W4, j = i - ly * W4;\n    const uint32_t* p = s_v"),
            ("  if (ls != nullptr) {  // finer level: one dword", "  if (LMX_EXP_SKIP & 4) return;\n  if (ls != nullptr) {  // finer level: one dword"),
            ("        *reinterpret_cast<uint32_t*>(out + (size_t)(4 * h + 0) * g.nib_ori_stride) =", "        if (!(LMX_EXP_SKIP & 8)) *reinterpret_cast<uint32_t*>(out + (size_t)(4 * h + 0) * g.nib_ori_stride) =")]
    names = {"V": 1, "H": 2, "OUT": 4, "ST0": 8, "NONE": 0}
elif which == "dqunroll":
    reps = []
    names = {"1": "-DLMX_DQ_UNROLL=1", "2": "-DLMX_DQ_UNROLL=2", "3": "-DLMX_DQ_UNROLL=3", "5": "-DLMX_DQ_UNROLL=5"}
elif which == "wpb":
    reps = []
    names = {"1": "-DLMX_SC_WPB=1", "2": "-DLMX_SC_WPB=2", "4": "-DLMX_SC_WPB=4", "8": "-DLMX_SC_WPB=8", "16": "-DLMX_SC_WPB=16"}
elif which == "refine":
    reps = [("              v[u] = load_u32_unaligned(ls + (size_t)((a & 0x1fffffffu) + lane_off));", "              v[u] = (LMX_EXP_SKIP & 1) ? (a + lane_off) * 0x9e3779b9u : load_u32_unaligned(ls + (size_t)((a & 0x1fffffffu) + lane_off));"),
            ("              acc += response4(v[u], *reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(s_masks) + row_of[u]), c7);",
             "              acc += (LMX_EXP_SKIP & 2) ? (v[u] & row_of[u]) : response4(v[u], *reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(s_masks) + row_of[u]), c7);")]
    names = {"NONE": 0, "NOLOAD": 1, "NORESP": 2, "NOBOTH": 3}
elif which == "b1half":
    # estimates for a cheaper first block.  HALF: in block 0 the second chunk re-reads the first chunk's addresses (same L1 traffic,
    # half the L2 -> L1 line fills).  NOLOAD: in block 0 the second chunk takes the first chunk's registers (half the loads).
    reps = [("        for (int i = 0; i < SB_BLOCK - 1; ++i) v[k][i] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)lane_off[k], (int)off[i], 0);",
             "        for (int i = 0; i < SB_BLOCK - 1; ++i) { if ((LMX_EXP_SKIP & 2) && b == 0 && k > 0) v[k][i] = v[0][i]; else v[k][i] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)lane_off[((LMX_EXP_SKIP & 1) && b == 0) ? 0 : k], (int)off[i], 0); }")]
    names = {"NONE": 0, "HALF": 1, "NOLOAD": 2}
elif which == "align":
    # would line-aligned loads be cheaper?  ALIGNED rounds every feature's byte offset down to 128 (wrong data, similar statistics)
    reps = [("    for (int i = 0; i < SB_BLOCK - 1; ++i) off[i] = blk[i];\n    const uint32_t meta = blk[SB_BLOCK - 1];\n    uint32_t v[NCH][SB_BLOCK - 1];",
             "    for (int i = 0; i < SB_BLOCK - 1; ++i) off[i] = LMX_EXP_SKIP ? (blk[i] & ~127u) : blk[i];\n    const uint32_t meta = blk[SB_BLOCK - 1];\n    uint32_t v[NCH][SB_BLOCK - 1];")]
    names = {"NONE": 0, "ALIGNED": 1}
elif which == "score":
    reps = []
    names = {"gu3": "-DLMX_SC8_GU=3", "gu4": "-DLMX_SC8_GU=4", "gu5": "-DLMX_SC8_GU=5", "gu6": "-DLMX_SC8_GU=6", "gu8": "-DLMX_SC8_GU=8"}
else:
    raise SystemExit("unknown")
for a, b in reps:
    assert src.count(a) == 1, a
    src = src.replace(a, b)
tmp = os.path.join(cs, "_exp_kernels.hip")
open(tmp, "w").write(src)
os.makedirs(os.path.join(root, "variants"), exist_ok=True)
# the variant = the modified kernels file compiled like the Makefile does, linked with the objects of the regular build
subprocess.check_call(["make", "-C", cs])
others = [os.path.join(cs, o) for o in ("lmx_f2.o", "lmx_api.o", "lmx_yaml.o", "lmx_train.o", "lmx_group.o", "lmx_hostcopy.o")]
procs = []
for n, bit in names.items():
    obj = os.path.join(root, "variants", "kernels_%s_%s.o" % (which, n))
    out = os.path.join(root, "variants", "liblmx_%s_%s.so" % (which, n))
    flags = ["-DLMX_EXP_SKIP=%d" % bit] if isinstance(bit, int) else bit.split()
    cmd = "/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I%s -I%s %s -c -o %s %s && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o %s %s %s -ldl && rm %s" % (
        os.path.join(root, "include"), cs, " ".join(flags), obj, tmp, out, obj, " ".join(others), obj)
    procs.append(subprocess.Popen(cmd, shell=True))
    if len(procs) >= 3:
        for p in procs: p.wait()
        procs = []
for p in procs: p.wait()
os.remove(tmp)
print(sorted(os.listdir(os.path.join(root, "variants"))))
