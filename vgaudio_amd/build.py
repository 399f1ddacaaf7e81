"""Builds vgaudio_amd/libvgaudio_hip.so for gfx950 with hipcc (in-tree, so the
.so travels to the GPU box with the repo snapshot).

    python -m vgaudio_amd.build [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libvgaudio_hip.so")

# -ffp-contract=off : RyuJIT never contracts a*b+c into an FMA
# -fwrapv           : C# int arithmetic is unchecked (wraps)
# no fast-math      : IEEE divide / rint / NaN compares are part of the parity contract
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fwrapv",
         "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "vgaudio_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + sources() + ["-o", OUT]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(OUT)
