#!/usr/bin/env python3
"""Static VALU cost of a kernel per source line: python tools/isa_by_line.py KERNEL_REGEX [EXTRA hipcc flags]
Compiles kernels.hip with line tables, takes the kernel whose mangled name matches, and adds up, per (file, line), the vector
instructions weighted by their measured issue cost on gfx950 (profiles/r03_valu_rate.txt; f64 arithmetic = 1, 32-bit integer = 0.45,
v_mad_u64_u32 = 1.45, v_rcp/rsq_f64 = 3.3, ...).  Static counts: a line inside a loop counts once; still, the trip loop's blocks each run
about once per trip, so the table shows where a trip's instructions are."""
import collections, os, re, subprocess, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pat = re.compile(sys.argv[1])
extra = sys.argv[2:]
# RMD_ISA_CLASS=moves: count only moves, selects and compares (v_mov*, v_cndmask*, v_cmp*, v_readlane / v_writelane), one each
only_moves = os.environ.get("RMD_ISA_CLASS") == "moves"
out = "/tmp/isa_by_line.s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-DRMD_DIAG=0", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
                "-gline-tables-only", "-I/opt/rocm/include", "-S", "--cuda-device-only", "-o", out, os.path.join(root, "raymond_amd/csrc/kernels.hip")] + extra,
               check=True, stderr=subprocess.DEVNULL)
files, cost, count = {}, collections.Counter(), collections.Counter()
inside, cur = False, None
def weight(m):
    if m.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")): return 3.3
    if m.startswith("v_mad_u64_u32"): return 1.45
    if m.startswith(("v_mul_lo_u32", "v_mul_hi_u32")): return 1.0
    if "f64" in m or m.startswith(("v_mov_b64", "v_lshl_add_u64", "v_lshrrev_b64", "v_lshlrev_b64", "v_cndmask_b64")): return 1.0
    return 0.45
for line in open(out):
    m = re.match(r"\s*\.file\s+(\d+)\s+\"[^\"]*\"\s+\"([^\"]+)\"", line) or re.match(r"\s*\.file\s+(\d+)\s+\"([^\"]+)\"", line)
    if m:
        files[int(m.group(1))] = os.path.basename(m.group(2))
    if re.match(r"^_Z\w+:", line):
        inside = bool(pat.search(line))
        continue
    if line.startswith(".Lfunc_end"):
        inside = False
    if not inside:
        continue
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", line)
    if m:
        cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    m = re.match(r"\s+(v_\w+)", line)
    if m and cur:
        if only_moves:
            if not m.group(1).startswith(("v_mov", "v_cndmask", "v_cmp", "v_readlane", "v_writelane", "v_accvgpr")): continue
            cost[cur] += 1.0
            count[cur] += 1
            continue
        cost[cur] += weight(m.group(1))
        count[cur] += 1
total = sum(cost.values())
print("total static VALU cost %.0f f64-equivalents, %d instructions" % (total, sum(count.values())))
src = {}
for (f, l), c in cost.most_common(60):
    path = os.path.join(root, "raymond_amd/csrc", f)
    if f not in src and os.path.exists(path):
        src[f] = open(path).read().split("\n")
    text = src[f][l - 1].strip()[:110] if f in src and l <= len(src[f]) else ""
    print("%6.1f %4d  %s:%d  %s" % (c, count[(f, l)], f, l, text))
