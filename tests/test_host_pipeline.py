"""The host side of the C ABI's pointer entry points (vgaudio_amd/csrc/host_pipeline.hpp: feeder threads, pinned
rings, per-chunk launches, drainer threads) is plain C++ over a few HIP runtime calls.  It is compiled here against a
mock of those calls whose streams are real asynchronous queues (tests/host/mockhip) and run under ThreadSanitizer:
data integrity over ragged shapes, error propagation without hangs, no data races.  CPU only."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("sanitizer", ["thread", "address"])
def test_host_pipeline_against_mock_streams(tmp_path, sanitizer):
    gxx = shutil.which("g++")
    assert gxx, "g++ is part of the image"
    exe = str(tmp_path / f"test_host_pipeline_{sanitizer}")
    src = os.path.join(HERE, "host", "test_host_pipeline.cpp")
    subprocess.run([gxx, "-std=c++17", "-O1", "-g", f"-fsanitize={sanitizer}", "-I", os.path.join(HERE, "host", "mockhip"),
                    src, "-o", exe, "-lpthread"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr
    assert "WARNING: ThreadSanitizer" not in r.stderr and "ERROR: AddressSanitizer" not in r.stderr, r.stderr
