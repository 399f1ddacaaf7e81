#!/usr/bin/env python3
"""Rough VGPR pressure profile of one kernel from hipcc's ISA listing (--save-temps): for every instruction of the kernel, in
listing order, the number of VGPRs holding a value that is still read later before being overwritten (straight-line
approximation: loop-carried values are undercounted).  Prints the profile in blocks of N instructions with the labels and a
few mnemonics, so that the region where the pressure peaks can be matched to the source.

    python tools/isa_pressure.py file.s kernel_substring [block=100]
"""
import re
import sys


def regs(tok):
    out = []
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(3) is not None:
            out.append(int(m.group(3)))
        else:
            out += list(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def main():
    path, name = sys.argv[1], sys.argv[2]
    block = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and name in l and l.split(";")[0].rstrip().endswith(":"))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    ins = []
    for l in lines[start + 1:end + 1]:
        t = l.split(";")[0].strip()
        if not t or t.startswith(".") or t.endswith(":"):
            if t.endswith(":"):
                ins.append(("label", t, [], []))
            continue
        op, _, rest = t.partition(" ")
        ops = [o.strip() for o in rest.split(",")]
        stores = op.startswith(("ds_write", "ds_or", "ds_add", "global_store", "scratch_store", "buffer_store", "flat_store", "ds_bpermute_b32x")) \
            or op.startswith("s_") or op.startswith("v_cmp") or op.startswith("global_atomic")
        if op.startswith("v_cmpx") or stores or op in ("v_readlane_b32", "v_readfirstlane_b32"):
            d, u = [], [r for o in ops for r in regs(o)]
            if op in ("v_readlane_b32", "v_readfirstlane_b32"):
                u = [r for o in ops[1:] for r in regs(o)]
        else:
            d = regs(ops[0]) if ops else []
            u = [r for o in ops[1:] for r in regs(o)]
            if op.startswith("v_writelane") or "dpp" in rest or op.startswith(("v_mac", "v_fmac", "v_mov_b32_dpp")):
                u += d
        ins.append((op, t, d, u))
    n = len(ins)
    live_after = [0] * n
    live = set()
    for i in range(n - 1, -1, -1):
        op, t, d, u = ins[i]
        live_after[i] = len(live)
        for r in d:
            live.discard(r)
        for r in u:
            live.add(r)
    k = 0
    for b in range(0, n, block):
        chunk = ins[b:b + block]
        peak = max(live_after[b:b + block])
        labels = [t for op, t, d, u in chunk if op == "label"]
        ops = {}
        for op, t, d, u in chunk:
            if op != "label":
                ops[op] = ops.get(op, 0) + 1
        top = sorted(ops.items(), key=lambda kv: -kv[1])[:6]
        print("%5d  peak %3d  %s  %s" % (b, peak, " ".join("%s:%d" % kv for kv in top), " ".join(labels)[:60]))


if __name__ == "__main__":
    main()
