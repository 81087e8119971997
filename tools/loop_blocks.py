#!/usr/bin/env python3
"""CPU only: the basic blocks of the innermost loop around a marker instruction in one kernel of vr_kernels.hip, from the
compiler's assembly -- which blocks the loop holds, how many instructions each has (vector / scalar / memory), and which
source lines they come from.  The union kernel is bound by the instructions its waves issue, so this is the figure a change
to the lane code is judged by before it goes to the GPU.

usage: python tools/loop_blocks.py [--kernel SUBSTR] [--marker OPCODE] [--nth N] [extra -D flags...]
  default: the N-th (0) `flat_load_dword` of vr_k_union<UnionNarrow>: the node fetch of the descend loop."""
import collections
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "volumerenderer_amd", "csrc")
args = sys.argv[1:]
kernel, marker, nth = "vr_k_unionINS_11UnionNarrow", "flat_load_dword", 0
while args and args[0] in ("--kernel", "--marker", "--nth"):
    if args[0] == "--kernel":
        kernel = args[1]
    elif args[0] == "--marker":
        marker = args[1]
    else:
        nth = int(args[1])
    args = args[2:]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
       "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-gpu-flush-denormals-to-zero", "-S", "--cuda-device-only", "-gline-tables-only",
       "-o", "/dev/stdout", "vr_kernels.hip"] + args
asm = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True).stdout.splitlines()
files = {}
for l in asm:
    m = re.match(r'\t\.file\t(\d+) "([^"]*)"(?: "([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
i0 = [i for i, l in enumerate(asm) if re.match(r"_Z\w*" + re.escape(kernel) + r"\w*:", l)][0]
i1 = [i for i, l in enumerate(asm) if l.startswith(".Lfunc_end") and i > i0][0]
blocks = collections.OrderedDict()
cur = "entry"
blocks[cur] = {"ins": [], "lines": collections.Counter(), "succ": []}
loc = None
for l in asm[i0 + 1:i1]:
    m = re.match(r"\t\.loc\t(\d+) (\d+) ", l)
    if m:
        loc = "%s:%s" % (files.get(int(m.group(1)), "?").replace("vr_trace_union.h", "U").replace("vr_kernels.hip", "K").replace("vr_trace_core.h", "C"), m.group(2))
        continue
    m = re.match(r"(\.LBB\d+_\d+):", l)
    if m:
        cur = m.group(1)
        blocks[cur] = {"ins": [], "lines": collections.Counter(), "succ": []}
        continue
    if l.startswith("\t") and not l.startswith("\t.") and not l.lstrip().startswith(";"):
        blocks[cur]["ins"].append(l.strip())
        blocks[cur]["lines"][loc] += 1
names = list(blocks)
for i, n in enumerate(names):
    b = blocks[n]
    fall = True
    for ins in b["ins"]:
        op = ins.split()[0]
        if op.startswith("s_cbranch"):
            b["succ"].append(ins.split()[1])
        elif op == "s_branch":
            b["succ"].append(ins.split()[1])
            fall = False
        elif op in ("s_endpgm",):
            fall = False
    if fall and i + 1 < len(names):
        b["succ"].append(names[i + 1])
pred = collections.defaultdict(list)
for n, b in blocks.items():
    for t in b["succ"]:
        pred[t].append(n)
with_marker = [n for n in names if any(ins.split()[0] == marker for ins in blocks[n]["ins"])]
head = with_marker[nth]


def reach(start, forward=True):
    seen, todo = set(), [start]
    while todo:
        n = todo.pop()
        for t in (blocks[n]["succ"] if forward else pred[n]):
            if t in blocks and t not in seen:
                seen.add(t)
                todo.append(t)
    return seen


def scc_of(start, allowed):
    """blocks of `allowed` that `start` reaches and that reach it, staying inside `allowed`"""
    def go(forward):
        seen, todo = set(), [start]
        while todo:
            n = todo.pop()
            for t in (blocks[n]["succ"] if forward else pred[n]):
                if t in allowed and t not in seen:
                    seen.add(t)
                    todo.append(t)
        return seen
    return (go(True) & go(False)) | {start}


# The innermost loop around the marker block: start from the strongly connected region that holds it (the kernel's main loop) and
# take away, again and again, the header of the outermost loop left -- the block of the region that comes first in the layout --
# as long as the marker block stays on a cycle without it.
idx = {n: i for i, n in enumerate(names)}
scc = scc_of(head, set(names))
inner = set(scc)
while True:
    first = min(inner, key=lambda n: idx[n])
    if first == head:
        break
    smaller = scc_of(head, inner - {first})
    if len(smaller) <= 1 and head not in blocks[head]["succ"]:
        break
    inner = smaller
tot = collections.Counter()
print("kernel %s: innermost loop around %s #%d = %d blocks (strongly connected region: %d blocks)" % (kernel, marker, nth, len(inner), len(scc)))
for n in names:
    if n not in inner:
        continue
    b = blocks[n]["ins"]
    v = sum(1 for x in b if x.startswith("v_"))
    sc = sum(1 for x in b if x.startswith("s_"))
    mem = [x.split()[0] for x in b if x.startswith(("ds_", "flat_", "global_", "buffer_"))]
    tot["all"] += len(b); tot["v"] += v; tot["s"] += sc; tot["mem"] += len(mem)
    print("  %-12s %3d  valu %3d salu %3d  %-44s %s" % (n, len(b), v, sc, " ".join(mem)[:44], " ".join("%s(%d)" % kv for kv in blocks[n]["lines"].most_common(3))))
print("  total %d instructions in the loop's blocks: valu %d, salu %d, memory %d" % (tot["all"], tot["v"], tot["s"], tot["mem"]))
