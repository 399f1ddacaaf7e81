#!/usr/bin/env python3
"""Per-frame instruction budget of gc_encode_persistent_kernel<8, false> from hipcc's ISA listing: where do
the ~409 lane-operations per input sample go against the ~240 the two passes x 8 predictors need?

The frame loop's basic blocks are found from the listing itself: the fourteen `v_mad_i32_i24 v, e, e, acc` of a quantise pass
(gc_encode_core.hpp: total += e * e) mark the passes; the blocks between the loop's labels are classified by what they hold.
Prints static instruction counts per block of the HOT path of one frame (lane = (channel, predictor), CPW = 8: one wave-frame
= 8 channel-frames = 112 samples), the cold block's path for a plain third trip, the helper wave's loop per tile, and the
resulting lane-operations per sample with the measured rates (cold-block rate, short-pass share) filled in.

    python tools/isa_budget_gc.py > profiles/r06_gc_encode_isa_budget.md
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "vgaudio_amd", "csrc", "gc_encode_kernel.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fwrapv", "-fno-fast-math", "-w"]
COLD_RATE = 0.3445          # bench.py signal_sensitivity, synthetic: cold blocks per wave-frame (profiles/r06 bench line)
SHORT_SHARE = 0.70          # wave-frames whose two passes run without the f32 detour (LABNOTES 9.7)


def listing():
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["--save-temps=obj", "-c", SRC, "-o", os.path.join(tmp, "k.o")], check=True, cwd=tmp,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        s = [f for f in os.listdir(tmp) if f.endswith(".s") and "gfx950" in f][0]
        return open(os.path.join(tmp, s)).read().split("\n")


def kernel(lines, name):
    a = next(i for i, l in enumerate(lines) if l.startswith("_Z") and name in l and l.split(";")[0].rstrip().endswith(":"))
    b = next(i for i in range(a, len(lines)) if "s_endpgm" in lines[i])
    out = []
    for l in lines[a + 1:b + 1]:
        t = l.split(";")[0].rstrip()
        if not t.strip() or t.strip().startswith((".p2align", ".", "//")) and not t.strip().endswith(":"):
            continue
        out.append(t.strip())
    return out


def count(block):
    c = collections.Counter()
    for t in block:
        if t.endswith(":"):
            continue
        op = t.split()[0]
        if op.startswith("v_"):
            c["VALU"] += 1
        elif op.startswith(("s_waitcnt", "s_nop")):
            c["wait/nop"] += 1
        elif op.startswith("s_"):
            c["SALU"] += 1
        elif op.startswith("ds_"):
            c["LDS"] += 1
        elif op.startswith(("global_", "scratch_", "buffer_", "flat_")):
            c["VMEM"] += 1
        else:
            c["other"] += 1
    c["total"] = sum(v for k, v in c.items() if k != "total")
    return c


def main():
    L = kernel(listing(), "gc_encode_persistent_kernelILi8ELb0E")
    sq = [i for i, l in enumerate(L) if re.search(r"v_mad_i32_i24 v\d+, (v\d+), \1, v\d+", l)]
    assert len(sq) >= 56, "the listing does not look like the encoder's"
    # first frame of the loop body (row set A): passes 0..55 = short pair (28) + normal pair (28)
    short_a, short_b = sq[0], sq[27]
    normal_a, normal_b = sq[28], sq[55]
    labels = [i for i, l in enumerate(L) if l.endswith(":")]
    prev_label = lambda i: max(j for j in labels if j <= i)
    next_label = lambda i: min(j for j in labels if j > i)
    frame_start = prev_label(prev_label(short_a) - 1)
    # walk back over the small blocks in front of the passes (s1 select, the rare sequential pre-scan) to the frame's first label
    k = labels.index(prev_label(short_a))
    while k > 0 and short_a - labels[k - 1] < 420:
        k -= 1
    frame_start = labels[k]
    short_blk = (prev_label(short_a), next_label(short_b))
    normal_blk = (prev_label(normal_a), next_label(normal_b))
    tail_end = normal_blk[1]
    while not L[tail_end].startswith("s_branch"):
        tail_end += 1
    # (what an execz branch jumps over -- the rare +M / -M tie's sequential pre-scan -- is not on the hot path)
    pre, skip_to = [], None
    for t in L[frame_start:short_blk[0]]:
        if skip_to:
            if t == skip_to + ":":
                skip_to = None
            continue
        pre.append(t)
        if t.startswith("s_cbranch_execz"):
            skip_to = t.split()[1]
    rows = [("row reads (12 x ds_read_b128) + pre-scan of the two history-dependent distances + first scale", count(pre)),
            ("passes B (s1 + 1) and A (s1) without the f32 detour, 28 sample steps", count(L[short_blk[0]:short_blk[1]])),
            ("the same with the conversions (30 % of the wave-frames take these instead)", count(L[normal_blk[0]:normal_blk[1]])),
            ("select the pass the reference ends on, 8-predictor argmin (DPP), winner's history, 4 x ds_write_b128", count(L[normal_blk[1]:tail_end + 1]))]
    print("# GC-ADPCM encoder: instruction budget of one wave-frame (round 6)\n")
    print("`gc_encode_persistent_kernel<8, false>`, static counts from hipcc's gfx950 listing of `vgaudio_amd/csrc/gc_encode_kernel.hip`")
    print("(`tools/isa_budget_gc.py`).  One wave-frame = 64 lanes = 8 channels x 8 predictors = 112 input samples.\n")
    print("| block of the hot frame | total | VALU | SALU | LDS | wait / nop |")
    print("|---|---:|---:|---:|---:|---:|")
    for name, c in rows:
        print("| %s | %d | %d | %d | %d | %d |" % (name, c["total"], c["VALU"], c["SALU"], c["LDS"], c["wait/nop"]))
    hot_valu = rows[0][1]["VALU"] + SHORT_SHARE * rows[1][1]["VALU"] + (1 - SHORT_SHARE) * rows[2][1]["VALU"] + rows[3][1]["VALU"]
    # the cold block: the third pass (short form) is the next cluster of 14 squares behind the hot tail that is a single pass
    third = [c for c in cluster(sq) if len(c) == 14]
    third_pass = count(L[prev_label(third[1][0]):next_label(third[1][-1])]) if len(third) > 1 else count([])
    print("\nCold block (third and later trips; %.3f of the wave-frames on the synthetic set): the third pass alone is %d VALU "
          "instructions for ~1.25 lanes of work; with its entry, loop and the frame's tail repeated behind it a plain visit executes "
          "~290 instructions more than a hot frame (the path was 385 until round 6: the bump loops of the generic resume point and the "
          "literal pass's hoisted set-up ran on every visit -- removing them changed nothing measurable: 146.1 against 145.9 ms).\n"
          % (COLD_RATE, third_pass["VALU"]))
    per_frame = hot_valu + COLD_RATE * (third_pass["VALU"] + 45)
    print("| per wave-frame (VALU) | instructions | lane-ops per input sample |")
    print("|---|---:|---:|")
    for name, v in (("the two passes (70 % short / 30 % with conversions)", SHORT_SHARE * rows[1][1]["VALU"] + (1 - SHORT_SHARE) * rows[2][1]["VALU"]),
                    ("row unpack, pre-scan, first scale", rows[0][1]["VALU"]),
                    ("select / argmin / commit", rows[3][1]["VALU"]),
                    ("cold block, %.3f x (third pass + ~45)" % COLD_RATE, COLD_RATE * (third_pass["VALU"] + 45)),
                    ("encoder wave, sum", per_frame)):
        print("| %s | %.0f | %.0f |" % (name, v, v * 64 / 112))
    print("\nMeasured (profiles/r05_sq_counters.json, same kernel arithmetic): 75.4 G VALU wave-instructions per launch / 105.3 M "
          "wave-frames = 716 per wave-frame = 409 lane-ops per sample.  The difference to the encoder wave's sum is the HELPER wave "
          "(one per two encoder waves: per tile of 4 frames x 16 channels it unpacks the frames, computes 96 pre-scan distances per "
          "frame with v_dot2 -- 576 of its ~950 instructions --, packs and stores the previous tile's winners) and the seams "
          "(0.3 % of the frames re-encoded).")
    print("""
## What the budget says (round 6)

* The two passes are 226 of the 409 lane-operations per sample -- 13.7 VALU instructions per sample step and predictor (two
  multiply-adds for the distance, three for the rounded shift, the nibble clamp, three for the reconstruction, its clamp, the
  error and its square-accumulate, half a max3 / min3 for the overflow): nothing in them is bookkeeping.
* The largest non-pass item is the HELPER's pre-scan: 96 distances per frame at 6 instructions each (v_dot2, the truncating
  division by 2048 in three, the difference, half a max3 / min3) = ~72 VALU per encoder wave-frame, 10 % of the launch.  In
  the numerator's domain (N = 2048 x - P: the distance is ceil(N / 2048) or floor(N / 2048) by the sign of P, both monotone)
  the running extremes cost 3 instructions per distance and decide the first scale unless an extreme lies within one of a
  power of two or of 9 x 2^j -- then the exact scan of that frame runs; costed at ~34 VALU per wave-frame (4.7 %) for a second
  way through the first-scale logic and its emulator, not built.
* The cold block is 53 lane-operations per sample for 1.25 lanes of work per visit.  Its entry was trimmed (385 -> ~290
  instructions a visit: the resume point of the generic path worked out only where it is taken, the literal pass's set-up no
  longer hoisted into every visit) with NO change in time (146.1 against 145.9 ms, same box): what a visit costs is the third
  pass's own dependent chain, a single one where the hot pair interleaves two.  Deferring third trips (LABNOTES 9.5) or
  batching them over a tile's frames was costed again with the roll-back's price: 67-74 instructions per frame against 78.
* Select / argmin / commit: 45 VALU + 34 SALU.  Storing the winner's nibbles under its own mask instead of fourteen selects
  in every lane removes 18 VALU and adds two exec-masked branches to the wave's critical path: 148.5 ms against 146.0.
  The wave is bound by the latency of its own chain as much as by the SIMD's issue slots (0.86-0.88 busy): removing
  instructions that are off the chain buys nothing, adding branches on it costs.
""")


def cluster(idx):
    out, cur = [], [idx[0]]
    for i in idx[1:]:
        if i - cur[-1] < 60:
            cur.append(i)
        else:
            out.append(cur)
            cur = [i]
    out.append(cur)
    return out


if __name__ == "__main__":
    main()
