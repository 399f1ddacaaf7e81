"""Builds vgaudio_amd/libvgaudio_hip.so for gfx950 with hipcc (in-tree, so the
.so travels to the GPU box with the repo snapshot).

    python -m vgaudio_amd.build [--force]

Every .hip file is compiled to its own object (in parallel; objects live in vgaudio_amd/build/ and are
rebuilt when the source, any header of csrc/ or the public header is newer), then linked.
"""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
OUT = os.path.join(HERE, "libvgaudio_hip.so")

# -ffp-contract=off : RyuJIT never contracts a*b+c into an FMA
# -fwrapv           : C# int arithmetic is unchecked (wraps)
# no fast-math      : IEEE divide / rint / NaN compares are part of the parity contract
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fwrapv",
         "-fno-fast-math", "-Wall", "-Wno-unused-function"]


# Per-file code generation options (hipcc -mllvm ..., ROCm 7.2's LLVM 22), measured on the file's own kernels at the BASELINE
# shapes (round 6, profiles/r06_z_compiler_*.log; same bytes out): the GC-ADPCM encoder's frame loop is one long dependent chain
# per lane, and the post-RA machine scheduler's reordering of it costs 4 %: 144.2-147.0 ms at configs[1] with the defaults, 139.0-
# 140.6 without it, 138.6-138.9 with relaxed-occupancy scheduling and without the pre-RA peepholes as well.  The same options on
# the other kernel files: coefficient kernel 30.4 / 30.1, ADX 17.4 / 17.4 + 7.3 / 7.3, GC decode 8.5-9.3 / 8.5, HCA encode 21.0 /
# 20.7, HCA decode 24.0 / 23.1 -- nothing or worse, so they keep the defaults.
FILE_FLAGS = {
    "gc_encode_kernel.hip": ["-mllvm", "-enable-post-misched=0", "-mllvm", "-amdgpu-schedule-relaxed-occupancy=true",
                             "-mllvm", "-amdgpu-enable-pre-ra-optimizations=0"],
}


FLAGS_KEY = "@compiler-options"


def flags_for(src):
    return FLAGS + FILE_FLAGS.get(os.path.basename(src), [])


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def headers():
    inc = os.path.join(HERE, "..", "include")
    return ([os.path.join(CSRC, f) for f in os.listdir(CSRC) if not f.endswith(".hip")] +
            [os.path.join(inc, f) for f in os.listdir(inc)] + [os.path.abspath(__file__)])


def source_hashes():
    """{file name: sha256} of the kernels' sources (csrc/): profiles/ summaries are stamped with them, and bench.py quotes
    a profile's counters only while the files its kernel is built from still match (a changed kernel silently
    invalidates measured traffic)."""
    import hashlib
    out = {f: hashlib.sha256(open(os.path.join(CSRC, f), "rb").read()).hexdigest()
           for f in sorted(os.listdir(CSRC)) if os.path.isfile(os.path.join(CSRC, f))}
    out[FLAGS_KEY] = hashlib.sha256(repr((FLAGS, sorted(FILE_FLAGS.items()))).encode()).hexdigest()   # the options are part of the code
    return out


# the .hip files that hold a codec's kernels; the headers come from their #include lines (bench.py: is a committed
# profile still about this code?)
KERNEL_FILES = {
    "gc": ["gc_encode_kernel.hip", "gcadpcm_kernels.hip", "gc_decode_kernel.hip"],
    "adx": ["adx_kernels.hip"],
    "hca": ["hca_encode_wave_kernel.hip", "hca_encode_kernel.hip", "hca_decode_kernels.hip"],
}


def kernel_sources(codec):
    """The codec's kernel files and every file of csrc/ they include, directly or not."""
    import re
    seen, todo = [], list(KERNEL_FILES[codec])
    while todo:
        f = todo.pop()
        if f in seen:
            continue
        seen.append(f)
        try:
            text = open(os.path.join(CSRC, f)).read()
        except OSError:
            continue
        for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', text, flags=re.M):
            inc = os.path.basename(inc)
            if os.path.exists(os.path.join(CSRC, inc)):
                todo.append(inc)
    return sorted(seen) + [FLAGS_KEY]


def profile_is_current(profile, codec):
    """profile: a JSON object written by tools/summarize_pmc.py (carries "_csrc_sha256").  A file missing from either side
    counts as changed."""
    stamp = (profile or {}).get("_csrc_sha256")
    if not isinstance(stamp, dict):
        return False
    now = source_hashes()
    return all(f in stamp and f in now and stamp[f] == now[f] for f in kernel_sources(codec))


# A second library whose GC-ADPCM encoder also counts its cold blocks (-DVGA_GC_STATS: five instructions per cold block, 1 ms of
# the configs[1] launch -- not in the product): bench.py's signal_sensitivity block reads the third-trip rates from it.
STATS_OUT = os.path.join(HERE, "libvgaudio_hip_stats.so")
STATS_SRC = "gc_encode_kernel.hip"


def build_stats_variant(verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    src = os.path.join(CSRC, STATS_SRC)
    obj = os.path.join(OBJ, "stats_" + STATS_SRC[:-4] + ".o")
    newest = max([os.path.getmtime(src)] + [os.path.getmtime(h) for h in headers()])
    if os.path.exists(STATS_OUT) and os.path.getmtime(STATS_OUT) >= max(newest, os.path.getmtime(OUT)):
        return STATS_OUT
    if not os.path.exists(obj) or os.path.getmtime(obj) < newest:
        cmd = [hipcc] + flags_for(src) + ["-DVGA_GC_STATS", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    others = [obj_of(s) for s in sources() if os.path.basename(s) != STATS_SRC]
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + others + [obj, "-o", STATS_OUT]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return STATS_OUT


def obj_of(src):
    return os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")


def stale_sources():
    newest_header = max(os.path.getmtime(h) for h in headers())
    out = []
    for s in sources():
        o = obj_of(s)
        if not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), newest_header):
            out.append(s)
    return out


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in sources() + headers())


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    todo = sources() if force else stale_sources()

    def compile_one(src):
        cmd = [hipcc] + flags_for(src) + ["-c", src, "-o", obj_of(src)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with concurrent.futures.ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 4) or 1) as pool:
        list(pool.map(compile_one, todo))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [obj_of(s) for s in sources()] + ["-o", OUT]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    if "--file-flags" in sys.argv:                      # tools/build_variants.sh, tools/kernel_resources.py: the extra options of one file
        print(" ".join(FILE_FLAGS.get(os.path.basename(sys.argv[sys.argv.index("--file-flags") + 1]), [])))
        sys.exit(0)
    build(force="--force" in sys.argv, verbose=True)
    print(OUT)
    if "--stats" in sys.argv:
        print(build_stats_variant(verbose=True))
