#!/usr/bin/env python3
"""Static instruction mix of one expand_grid_kernel instantiation between the phase markers.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DMPLX_PHASE_MARK -S --cuda-device-only \
          -o /tmp/grid_mark.s motion_primitive_library_amd/csrc/expand_grid_kernel.hip
    python profiles/micro/isa_phase_count.py /tmp/grid_mark.s 3 2 0 0 [region ...]        # D K YAW POT

Counts are STATIC (per appearance in the listing, attributed to the last marker seen in file order): a map of
where the instructions are, not the dynamic count (that is SQ_INSTS_VALU, micro/valu_phase_split.sh).  Needs no GPU.
With region names (PT3 ...) the opcode histogram of those regions is printed too.
"""
import collections
import re
import sys


def classify(op):
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "lane"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_setprio", "s_sleep")):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def function_lines(path, D, K, Y, P):
    name = "expand_grid_kernelILi%dELi%dELb%dELb%dEEE" % (D, K, Y, P)
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and name in l.split(":")[0])
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    return lines[start + 1:end]


def main():
    path, D, K, Y, P = sys.argv[1], *[int(x) for x in sys.argv[2:6]]
    region = "pre"
    agg = collections.OrderedDict()
    ops = collections.defaultdict(collections.Counter)
    for l in function_lines(path, D, K, Y, P):
        t = l.strip()
        m = re.match(r"; PTMARK (\d+)", t)
        if m:
            region = "PT%s" % m.group(1)
            continue
        if not t or t.startswith((";", ".", "//")) or t.endswith(":") or re.match(r"^\.?L?BB\d+_\d+:", t):
            continue
        op = t.split()[0]
        c = classify(op)
        agg.setdefault(region, collections.Counter())[c] += 1
        ops[region][op] += 1
    tot = collections.Counter()
    cols = ("valu", "lane", "salu", "lds", "vmem", "smem", "branch")
    print("%-6s " % "region" + " ".join("%6s" % c for c in cols))
    for r, c in agg.items():
        tot.update(c)
        print("%-6s " % r + " ".join("%6d" % c[k] for k in cols))
    print("%-6s " % "total" + " ".join("%6d" % tot[k] for k in cols))
    for r in sys.argv[6:]:
        print(r, ops[r].most_common(40))


if __name__ == "__main__":
    main()
