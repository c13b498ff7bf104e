#!/usr/bin/env python3
"""Static instruction mix of one kernel of liblmx's device code, split at its barriers: VALU / LDS / VMEM / SALU counts per stage, the VALU count
weighted by what the issue-rate microbenchmark measured on gfx950 (profiles/r04_valu_issue_microbench.txt: plain VOP1 / VOP2 adds, logic, constant
shifts and v_bitop3 ~2.3 cycles per wave64 instruction per SIMD; multiplies, dot products, packed math, v_perm / v_alignbit, 3-operand VOP3 integer
ops, v_min / v_max, v_ffbl, shifts by a VGPR amount, DPP / SDWA forms and compares ~4.2).  Loops are counted once (pass trip counts with --mult).
usage: isa_mix.py <kernel name substring> [--mult stage=trips,...] [--src csrc/lmx_kernels.hip]
Round 4 used it to find where k_color_quantize's 1811 VALU instructions per wave went (stage x trip count reproduced the PMC figure)."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HALF = re.compile(r"^(v_perm_b32|v_alignbit|v_alignbyte|v_mul_|v_mad_|v_dot|v_pk_|v_add3|v_and_or|v_or3|v_lshl_add|v_lshl_or|v_add_lshl|v_bfe|v_bfi|v_bfm|v_cvt|v_readlane|v_readfirstlane|"
                  r"v_writelane|v_sad|v_med3|v_min|v_max|v_xad|v_lshlrev_b64|v_lshrrev_b64|v_mbcnt|v_div|v_rcp|v_sqrt|v_rsq|v_ldexp|v_trunc|v_rndne|v_floor|v_ffb|v_bcnt|v_cmp|v_pack|v_addc|v_subb)")


def cost(ins, line):
    if "dpp" in line or "sdwa" in line:
        return 4.2
    if ins.startswith("v_bitop3"):
        return 2.3
    if HALF.match(ins):
        return 4.2
    if re.match(r"^v_(lshlrev|lshrrev|ashrrev)_b?i?(32|16)", ins) and re.search(r",\s*v\d+,\s*v\d+\s*$", line):   # shift amount in a VGPR
        return 4.2
    return 2.3


def main():
    name = sys.argv[1]
    src = os.path.join(ROOT, "linemod_pose_estimation_amd", "csrc", "lmx_kernels.hip")
    mult = {}
    for i, a in enumerate(sys.argv):
        if a == "--src":
            src = sys.argv[i + 1]
        if a == "--mult":
            mult = {k: float(v) for k, v in (kv.split("=") for kv in sys.argv[i + 1].split(","))}
    cs = os.path.dirname(src)
    out = os.path.join(tempfile.mkdtemp(), "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"), "-I", cs,
                           "--cuda-device-only", "-S", "-o", out, src], stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and name in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    meta = {}
    for l in lines[end:end + 200]:
        m = re.match(r"\s*\.amdhsa_(next_free_vgpr|group_segment_fixed_size|next_free_sgpr)\s+(\d+)", l)
        if m:
            meta[m.group(1)] = int(m.group(2))
    print(lines[start].split(":")[0], meta)
    segs, cur = [], []
    for l in lines[start:end + 1]:
        cur.append(l)
        if "s_barrier" in l:
            segs.append(cur)
            cur = []
    segs.append(cur)
    tv = tc = 0
    for k, seg in enumerate(segs):
        n = collections.Counter()
        nv = nl = nm = ns = 0
        cyc = 0.0
        for l in seg:
            t = l.strip().split()
            if not t or t[0].startswith(";") or t[0].startswith("."):
                continue
            ins = t[0]
            if ins.startswith("v_"):
                nv += 1
                cyc += cost(ins, l)
                n[re.sub(r"_e(32|64)$", "", ins)] += 1
            elif ins.startswith("ds_"):
                nl += 1
            elif ins.startswith(("global_", "buffer_", "flat_", "scratch_")):
                nm += 1
            elif ins.startswith("s_"):
                ns += 1
        m = mult.get(str(k), 1.0)
        tv += nv * m
        tc += cyc * m
        print("stage %d x%-4g VALU %5d  weighted cycles %7.0f  LDS %4d  VMEM %3d  SALU %4d  | %s" % (k, m, nv, cyc, nl, nm, ns, " ".join("%s:%d" % kv for kv in n.most_common(7))))
    print("sum (x trip counts given): VALU %.0f, weighted cycles %.0f" % (tv, tc))


if __name__ == "__main__":
    main()
