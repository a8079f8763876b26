#!/usr/bin/env python3
"""Where a kernel's instructions are, loop by loop: python tools/isa_loops.py KERNEL_REGEX [--freq LOOP=PER_SAMPLE ...] [EXTRA hipcc flags]

Compiles kernels.hip with line tables, takes the kernel whose mangled name matches, finds the natural loops of its assembly (control-flow graph of the
assembly, back edges to a dominating block, loop bodies by reachability) and adds up, per loop (innermost only: an instruction belongs to the innermost loop
around it), the instructions by class — f64 arithmetic, 32/64-bit integer, moves / selects / compares, other vector, scalar, branches, LDS,
vector memory — together with the source lines the loop's instructions come from (so that a loop can be recognised).  `--freq L3=1.04` gives
loop L3's bodies per sample (from the DIAG event counters, tools/diag_counts.sh); the table then also shows instructions per sample and the
totals can be held against the SQ_INSTS_* counters of profiles/pmc_latest.json.  Static analysis: a loop body's blocks are taken to run once
per iteration (branches inside a body are not weighted)."""
import collections, os, re, subprocess, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
pat = re.compile(args.pop(0))
freq, extra, cold_lines, detail, detail_re = {}, [], set(), {}, {}
while args:
    a = args.pop(0)
    if a == "--freq":
        k, v = args.pop(0).split("=")
        freq[k] = float(v)
    elif a == "--detail":  # LOOP[:CLASS,...[:MNEMONIC_REGEX]]: the loop's instructions per source line (of those classes / mnemonics)
        d = args.pop(0).split(":")
        detail[d[0]] = set(d[1].split(",")) if len(d) > 1 and d[1] else None
        if len(d) > 2: detail_re[d[0]] = re.compile(d[2])
    elif a == "--cold":  # FILE:LINE,...: basic blocks whose vector instructions mostly come from these lines are not counted (fallback paths behind a ballot)
        for c in args.pop(0).split(","):
            f, l = c.split(":")
            cold_lines.add((f, int(l)))
    else:
        extra.append(a)
out = "/tmp/isa_loops.s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-DRMD_DIAG=0", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
                "-gline-tables-only", "-I/opt/rocm/include", "-S", "--cuda-device-only", "-o", out, os.path.join(root, "raymond_amd/csrc/kernels.hip")] + extra,
               check=True, stderr=subprocess.DEVNULL)

def classify(m):
    if m.startswith("v_"):
        if m.startswith(("v_mov", "v_cndmask", "v_cmp", "v_readlane", "v_writelane", "v_readfirstlane", "v_accvgpr", "v_pk_mov", "v_swap", "v_permlane")): return "move"
        if m.startswith(("v_add_f64", "v_mul_f64", "v_fma_f64", "v_fmac_f64", "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "v_pk_add_f64", "v_pk_mul_f64", "v_pk_fma_f64")): return "f64"
        if "f64" in m: return "f64x"  # max / min / ldexp / frexp / trunc / rndne / fract / cvt: f64 operands, not arithmetic the counters class as add / mul / fma
        if m.startswith("v_cvt"): return "cvt"
        return "int"
    if m.startswith("ds_"): return "lds"
    if m.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if m.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_call")): return "branch"
    if m.startswith(("s_load", "s_buffer_load", "s_store")): return "smem"
    if m.startswith(("s_waitcnt", "s_nop", "s_sleep", "s_setprio", "s_barrier", "s_endpgm")): return "wait"
    if m.startswith("s_"): return "salu"
    return "other"

files, insts, labels = {}, [], {}
inside, cur = False, None
for line in open(out):
    m = re.match(r"\s*\.file\s+(\d+)\s+\"[^\"]*\"\s+\"([^\"]+)\"", line) or re.match(r"\s*\.file\s+(\d+)\s+\"([^\"]+)\"", line)
    if m: files[int(m.group(1))] = os.path.basename(m.group(2))
    if re.match(r"^_Z\w+:", line):
        inside = bool(pat.search(line))
        continue
    if line.startswith(".Lfunc_end"): inside = False
    if not inside: continue
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", line)
    if m:
        cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    m = re.match(r"^(\.?L\w+):", line)
    if m:
        labels[m.group(1)] = len(insts)
        continue
    m = re.match(r"^(\d+):", line)  # local labels of the inline assembly blocks ("1:" ... "s_cbranch 1b")
    if m:
        labels["local%s@%d" % (m.group(1), len(insts))] = len(insts)
        continue
    m = re.match(r"\s+([vsdgbf]\w+)\s*(.*)", line)
    if m and not m.group(1).startswith(("s_code_end",)):
        insts.append((m.group(1), m.group(2), cur))

# control-flow graph: blocks begin at labels and after branches; natural loops = back edges u -> h with h dominating u
def target_of(i):
    mn, ops, _ = insts[i]
    t = ops.split()[-1] if ops else ""
    if t in labels: return labels[t]
    m = re.match(r"(\d+)([bf])$", t)
    if m:
        c = [v for k, v in labels.items() if k.startswith("local%s@" % m.group(1))]
        c = [v for v in c if (v <= i if m.group(2) == "b" else v > i)]
        if c: return max(c) if m.group(2) == "b" else min(c)
    return None
starts = {0} | set(labels.values())
for i, (mn, ops, _) in enumerate(insts):
    if mn.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")): starts.add(i + 1)
starts = sorted(x for x in starts if x < len(insts))
block_of = {}
for bi, st in enumerate(starts):
    en = starts[bi + 1] if bi + 1 < len(starts) else len(insts)
    for i in range(st, en): block_of[i] = bi
nb = len(starts)
succ = [[] for _ in range(nb)]
for bi, st in enumerate(starts):
    en = (starts[bi + 1] if bi + 1 < nb else len(insts)) - 1
    mn = insts[en][0]
    if mn.startswith(("s_cbranch", "s_branch")):
        t = target_of(en)
        if t is not None and t < len(insts): succ[bi].append(block_of[t])
    if not mn.startswith(("s_branch", "s_endpgm", "s_setpc")) and bi + 1 < nb: succ[bi].append(bi + 1)
pred = [[] for _ in range(nb)]
for u in range(nb):
    for v in succ[u]: pred[v].append(u)
# dominators (iterative, bit sets as Python ints)
full = (1 << nb) - 1
dom = [full] * nb
dom[0] = 1
changed = True
while changed:
    changed = False
    for v in range(1, nb):
        d = full
        for u in pred[v]: d &= dom[u]
        d |= 1 << v
        if d != dom[v]: dom[v], changed = d, True
body_of = {}
for u in range(nb):
    for h in succ[u]:
        if dom[u] >> h & 1:  # back edge
            body = body_of.setdefault(h, {h})
            stack = [u]
            while stack:
                x = stack.pop()
                if x in body: continue
                body.add(x)
                stack.extend(pred[x])
loops = sorted(body_of.items(), key=lambda kv: starts[kv[0]])  # (header block, set of blocks)
def innermost(i):
    b, best = block_of[i], None
    for n, (h, body) in enumerate(loops):
        if b in body and (best is None or len(body) < len(loops[best][1])): best = n
    return best
def depth(n):
    return sum(1 for (h2, b2) in loops if loops[n][1] <= b2) - 1
classes = ["f64", "f64x", "int", "cvt", "move", "salu", "branch", "lds", "vmem", "smem", "wait"]
tab = collections.defaultdict(collections.Counter)
lines = collections.defaultdict(collections.Counter)
cold_blocks = set()
per_line = collections.defaultdict(lambda: collections.defaultdict(collections.Counter))
for bi, st in enumerate(starts):
    en = starts[bi + 1] if bi + 1 < nb else len(insts)
    v = [insts[i][2] for i in range(st, en) if insts[i][0].startswith("v_")]
    if len(v) >= 4 and sum(1 for loc in v if loc in cold_lines) > 0.8 * len(v): cold_blocks.add(bi)
n_cold = 0
for i, (mn, ops, loc) in enumerate(insts):
    if block_of[i] in cold_blocks:
        n_cold += 1
        continue
    n = innermost(i)
    tab[n][classify(mn)] += 1
    name = "top" if n is None else "L%d" % n
    if name in detail and (detail[name] is None or classify(mn) in detail[name]) and (name not in detail_re or detail_re[name].search(mn)): per_line[name][loc][mn] += 1
    if loc and mn.startswith("v_"): lines[n][loc] += 1
print("kernel: %d instructions, %d loops; %d instructions in %d cold blocks left out" % (len(insts), len(loops), n_cold, len(cold_blocks)))
print("%-6s %-5s %6s | " % ("loop", "depth", "insts") + " ".join("%6s" % c for c in classes) + " | lines of most of its vector instructions")
tot = collections.Counter()
order = [None] + list(range(len(loops)))
for n in order:
    name = "top" if n is None else "L%d" % n
    c = tab[n]
    if not sum(c.values()): continue
    where = ", ".join("%s:%d(%d)" % (f.replace(".hpp", ""), l, k) for (f, l), k in lines[n].most_common(4))
    print("%-6s %-5s %6d | " % (name, "" if n is None else depth(n), sum(c.values())) + " ".join("%6d" % c[k] for k in classes) + " | " + where)
    if name in freq:
        for k in classes: tot[k] += c[k] * freq[name]
        print("%-6s x %-8g per sample: " % ("", freq[name]) + " ".join("%s %.2f" % (k, c[k] * freq[name]) for k in classes if c[k]))
if freq:
    print("sum over the weighted loops, per sample: " + " ".join("%s %.2f" % (k, tot[k]) for k in classes))
    print("  vector total %.2f (f64 %.2f, f64-other + moves %.2f, int %.2f, cvt %.2f)" % (sum(tot[k] for k in ("f64", "f64x", "int", "cvt", "move")), tot["f64"], tot["f64x"] + tot["move"], tot["int"], tot["cvt"]))
for name, d in per_line.items():
    print("---- %s, %s per source line" % (name, ",".join(sorted(detail[name])) if detail[name] else "all instructions"))
    srcs = {}
    for loc, c in sorted(d.items(), key=lambda kv: -sum(kv[1].values()))[:45]:
        text = ""
        if loc:
            path = os.path.join(root, "raymond_amd/csrc", loc[0])
            if loc[0] not in srcs and os.path.exists(path): srcs[loc[0]] = open(path).read().split("\n")
            if loc[0] in srcs and 0 < loc[1] <= len(srcs[loc[0]]): text = srcs[loc[0]][loc[1] - 1].strip()[:90]
        print("%4d  %s:%s  %s   | %s" % (sum(c.values()), loc[0] if loc else "?", loc[1] if loc else "?", " ".join("%s×%d" % kv for kv in c.most_common(4)), text))
