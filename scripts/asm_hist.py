#!/usr/bin/env python3
"""Development tool: instruction histogram of one kernel in a hipcc -S listing.
usage: asm_hist.py listing.s <mangled-name-substring> [--blocks]
Prints per basic block (label) the instruction count by class, then the biggest blocks."""
import collections
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and pat in l.split(":")[0])
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
blocks, cur = collections.OrderedDict(), "entry"
blocks[cur] = []
for l in lines[start + 1:end]:
    s = l.strip()
    if not s or s.startswith(";") or s.startswith("."):
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            cur = m.group(1)
            blocks[cur] = []
        continue
    blocks[cur].append(s.split()[0])


def cls(op):
    if op.startswith("v_pk_"): return "v_pk"
    if op.startswith(("v_mov", "v_accvgpr")): return "v_mov"
    if op.startswith("v_cndmask"): return "v_cndmask"
    if op.startswith(("v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32")): return "v_f32"
    if op.startswith(("v_min", "v_max")): return "v_minmax"
    if op.startswith(("v_log", "v_sqrt", "v_rcp", "v_exp")): return "v_trans"
    if op.startswith("v_"): return "v_other"
    if op.startswith("ds_"): return "ds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem:" + op.split("_")[1]
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if op.startswith("s_"): return "salu"
    return op


tot = collections.Counter()
rows = []
for name, ops in blocks.items():
    c = collections.Counter(cls(o) for o in ops)
    tot.update(c)
    rows.append((len(ops), name, c))
print("kernel:", lines[start].split(":")[0][:120])
print("total instructions", sum(tot.values()), dict(tot))
for n, name, c in sorted(rows, reverse=True)[:8]:
    print(f"{name:12s} {n:5d}", dict(c))
if "--ops" in sys.argv:
    big = max(rows)[1]
    print(collections.Counter(blocks[big]).most_common(40))
