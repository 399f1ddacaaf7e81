#!/usr/bin/env python3
"""Turns rocprofv3 counter_collection CSVs (one --pmc pass each) into one JSON object per kernel.

    python tools/summarize_pmc.py traffic OUT.json FETCH_DIR WRITE_DIR [FETCH_DIR WRITE_DIR ...]
    python tools/summarize_pmc.py sq OUT.json DIR [DIR ...]

traffic: FETCH_SIZE x 2 (MI355X_MICROARCH.md: gfx950 reports half the bytes of wide coalesced reads; checked against
kernels whose traffic is known) + WRITE_SIZE, in bytes per launch (mean over the vga:: kernels' launches).
sq: mean counter values per launch plus the launch duration, per kernel.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"vga::\w+::(\w+)", name)
    return m.group(1) if m else None


def read(dirname):
    rows = []
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection*.csv"), recursive=True):
        with open(f, newline="") as fh:
            rows += list(csv.DictReader(fh))
    per = defaultdict(lambda: defaultdict(list))          # kernel -> counter -> [values per dispatch]
    dur = defaultdict(dict)
    for r in rows:
        k = short(r["Kernel_Name"])
        if not k:
            continue
        per[k][r["Counter_Name"]].append((r["Dispatch_Id"], float(r["Counter_Value"])))
        dur[k][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    out = {}
    for k, counters in per.items():
        out[k] = {}
        # the launches that count: near-empty ones (e.g. the repair launch that returns at once, which shares the
        # kernel's name) are dropped from every mean -- chosen once, by duration, so that all counters (SQ_WAVES
        # included) average over the same dispatches
        dmax = max(dur[k].values()) if dur[k] else 0.0
        keep = {d for d, t in dur[k].items() if t >= 0.05 * dmax}
        for c, vals in counters.items():
            by = defaultdict(float)
            for d, v in vals:                              # a counter may come in several rows (XCDs): sum per dispatch
                if d in keep:
                    by[d] += v
            out[k][c] = sum(by.values()) / max(len(by), 1)
        kept = [dur[k][d] for d in keep]
        out[k]["_dur_ms"] = sum(kept) / max(len(kept), 1)
        out[k]["_launches"] = len(kept)
        out[k]["_launches_dropped"] = len(dur[k]) - len(kept)
    return out


def main():
    mode, outp, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    res = {}
    if mode == "traffic":
        for fd, wd in zip(dirs[0::2], dirs[1::2]):
            f, w = read(fd), read(wd)
            for k in sorted(set(f) & set(w)):
                if "FETCH_SIZE" not in f[k] or "WRITE_SIZE" not in w[k]:
                    continue
                rd = f[k]["FETCH_SIZE"] * 1024 * 2
                wr = w[k]["WRITE_SIZE"] * 1024
                res[k] = {"FETCH_SIZE_KB_per_launch": f[k]["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": w[k]["WRITE_SIZE"],
                          "hbm_read_bytes_corrected": rd, "hbm_write_bytes": wr, "traffic_bytes_per_launch": rd + wr,
                          "launch_ms_profiled": f[k]["_dur_ms"]}
        res["_note"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes with --kernel-trace; FETCH_SIZE x 2 per the "
                        "gfx950 correction (MI355X_MICROARCH.md), calibrated in round 1 on gc_encode_kernel (reads each PCM sample once)")
    else:
        for d in dirs:
            for k, v in read(d).items():
                res.setdefault(k, {}).update(v)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from vgaudio_amd.build import source_hashes
    res["_csrc_sha256"] = source_hashes()              # bench.py quotes these counters only while the kernels are unchanged
    json.dump(res, open(outp, "w"), indent=1, sort_keys=True)
    print(json.dumps({k: {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items()} if isinstance(v, dict) else v
                      for k, v in res.items()}, indent=1)[:6000])


if __name__ == "__main__":
    main()
