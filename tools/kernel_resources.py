#!/usr/bin/env python3
"""Compiles every .hip file of vgaudio_amd/csrc for gfx950 with -Rpass-analysis=kernel-resource-usage and prints one line per
kernel: VGPRs, AGPRs, SGPRs, spills, scratch bytes per lane, LDS bytes, occupancy -- plus, from the disassembly, how many
scratch_* and v_readlane / v_writelane instructions the kernel holds (the figures behind "where do the
spills sit").  Needs no GPU.

    python tools/kernel_resources.py [file.hip ...] [-D...] > profiles/rNN_kernel_resources.txt
"""
import concurrent.futures
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vgaudio_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fwrapv", "-fno-fast-math", "-w"]
FIELDS = ["VGPRs", "AGPRs", "TotalSGPRs", "VGPRs Spill", "SGPRs Spill", "ScratchSize [bytes/lane]", "LDS Size [bytes/block]",
          "Occupancy [waves/SIMD]"]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return out.splitlines()


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("vga::", "")


def analyse(path, defs):
    with tempfile.TemporaryDirectory() as tmp:
        base = os.path.join(tmp, "k")
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
        from vgaudio_amd.build import FILE_FLAGS              # the product's per-file code generation options
        r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + FILE_FLAGS.get(os.path.basename(path), []) + defs +
                           ["-Rpass-analysis=kernel-resource-usage", "--save-temps=obj", "-c",
                            path, "-o", base + ".o"], capture_output=True, text=True, cwd=tmp)
        if r.returncode:
            raise SystemExit(r.stderr[-3000:])
        kernels, cur = {}, None
        for line in r.stderr.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                cur = kernels.setdefault(m.group(1), {})
                continue
            m = re.search(r":\s{2,}([A-Za-z][^:]*?): (\S+) \[-Rpass", line)
            if m and cur is not None:
                cur[m.group(1)] = m.group(2)
        asm = [f for f in os.listdir(tmp) if f.endswith(".s") and "gfx950" in f]
        counts = {}
        if asm:
            text = open(os.path.join(tmp, asm[0])).read()
            for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end", text, flags=re.M | re.S):
                body = m.group(2)
                counts[m.group(1)] = (len(re.findall(r"^\s+scratch_", body, flags=re.M)),
                                      len(re.findall(r"^\s+v_(?:readlane|writelane)_b32", body, flags=re.M)),
                                      len(re.findall(r"^\s+[vs]_\w+|^\s+(?:ds|global|buffer|flat|scratch)_\w+", body, flags=re.M)))
        return os.path.basename(path), kernels, counts


def main():
    defs = [a for a in sys.argv[1:] if a.startswith("-D")]
    files = [os.path.abspath(a) for a in sys.argv[1:] if not a.startswith("-")]
    if not files:
        files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(len(files), os.cpu_count() or 4)) as pool:
        results = list(pool.map(lambda f: analyse(f, defs), files))
    print("# hipcc %s %s -Rpass-analysis=kernel-resource-usage (gfx950)" % (" ".join(FLAGS[1:]), " ".join(defs)))
    print("%-78s %5s %5s %5s %6s %6s %7s %7s %4s %8s %9s %7s" % ("kernel", "VGPR", "AGPR", "SGPR", "vspill", "sspill", "scratch", "LDS",
                                                                "occ", "scratch_", "lane r/w", "instrs"))
    for fname, kernels, counts in results:
        if not kernels:
            continue
        print("## " + fname)
        names = list(kernels)
        for mangled, nice in zip(names, demangle(names)):
            k = kernels[mangled]
            c = counts.get(mangled, ("-", "-", "-"))
            print("%-78s %5s %5s %5s %6s %6s %7s %7s %4s %8s %9s %7s" % ((short(nice)[:78],) + tuple(k.get(f, "-") for f in FIELDS) + c))


if __name__ == "__main__":
    main()
