#!/usr/bin/env python3
"""bench.py -- the driver's benchmark contract, one JSON line per run.

    python bench.py --gpus N --steps K --warmup W                  (headline: --codec gc)
    python bench.py --codec adx|hca ...                            (BASELINE configs[2] / configs[3], same contract)
    (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   -- or plain
          python bench.py --gpus N, which starts its own N ranks, one per GPU, and prints rank 0's line)

--codec gc  (default; BASELINE.json's metric): configs[1], 4096 independent mono channels x 48 kHz x 60 s per GPU.
            A step = gc_coefs_kernel (GcAdpcmCoefficients.CalculateCoefficients for every channel) +
            gc_encode_kernel (GcAdpcmEncoder.Encode), inputs resident in HBM.
--codec adx: configs[2], 4096 channels CRI ADX encode + decode round trip.  A step = encode + decode.
--codec hca: configs[3], 1024 stereo streams CRI HCA encode, quality High.  A step = encode (decode is timed beside it).

Channels/streams shard across GPUs with no data-path collective (weak scaling: every rank owns its own units);
with N > 1 the gc line also reports the step with the final RCCL gather of the bitstream to rank 0.
Rank 0 prints ONE JSON line.

roofline    : the dominant kernel; achieved = algorithmic bytes per launch (GC encode 2 + 8/14 B/sample,
              ADX 2.5625 B/sample, HCA 2 + 682/2048 B/channel-sample) / mean HIP-event duration of that launch;
              traffic = HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/), valid for the
              profiled shape only.
cpu_baseline: the oracle (C restatement of the reference, "port") run with the reference's scheduling -- one task per
              channel (Parallel.For, GcAdpcmFormat.cs:65; CriAdxFormat.cs:67) or per stream (HCA has no intra-file
              parallelism, CriHcaFormat.cs:53-81; Cli/Batch.cs:24-25) -- on a bounded sample; the same leg compares its
              output with this run's GPU output for those units, bit for bit.  The oracle is touched nowhere else.
e2e         : (gc, N = 1) one vga_gcadpcm_encode_batch call through the host-pointer C ABI -- what a P/Invoke caller
              gets, pageable host arrays in and out -- next to the PCIe-bound time for the same bytes.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

# before anything initialises HIP: the host pipeline wants more hardware queues than the runtime's default of 4
# (vgaudio_amd/__init__.py does the same; this covers an import order where torch touches the GPU first)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
XGMI_LINK_GBS = 153.0          # one xGMI link, per direction (MI355X_MICROARCH.md: 7 links x ~153 GB/s per GPU, point to point)
ENC_BYTES_PER_SAMPLE = 2.0 + 8.0 / 14.0
COEF_BYTES_PER_SAMPLE = 2.0
PIPE_BYTES_PER_SAMPLE = 4.0 + 8.0 / 14.0
ADX_BYTES_PER_SAMPLE = 2.0 + 18.0 / 32.0
HCA_BYTES_PER_CHANNEL_SAMPLE = 2.0 + 682.0 / 2048.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--codec", choices=["gc", "adx", "hca"], default="gc")
    ap.add_argument("--channels", type=int, default=4096, help="channels per GPU (gc, adx)")
    ap.add_argument("--streams", type=int, default=1024, help="stereo streams per GPU (hca)")
    ap.add_argument("--seconds", type=float, default=60.0, help="audio seconds per channel @48 kHz")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-channels", type=int, default=0, help="units in the CPU baseline sample (0 = auto)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the host-pointer ABI measurement")
    ap.add_argument("--e2e-channels", type=int, default=0, help="channels of the e2e call (0 = as many of --channels as host memory allows)")
    ap.add_argument("--no-mixed", action="store_true", help="gc: skip the mixed-lengths block (ragged batch of files)")
    ap.add_argument("--no-signals", action="store_true", help="gc: skip the signal_sensitivity block (the step on other signal classes)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="gc: skip the short ADX (configs[2]) and HCA (configs[3]) runs appended to the line as `other_configs`")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="process-group backend for N > 1 (nccl = RCCL; gloo moves the gather through host memory: for trying "
                         "the N > 1 path on a box with fewer GPUs than ranks)")
    ap.add_argument("--share-gpu", action="store_true", help="every rank uses cuda:0 (N > 1 plumbing test on a 1-GPU box; implies --backend gloo)")
    return ap.parse_args()


def usable_cpus():
    """CPUs this process may actually use: min(affinity, cgroup quota).  The GPU box reports 256
    logical CPUs but caps the container with cpu.max (e.g. 1600000/100000 = 16 CPUs)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    note = f"{os.cpu_count()} logical CPUs visible"
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            q = max(1, int(int(quota) / int(period) + 0.5))
            if q < n:
                note += f", cgroup cpu.max limits this container to {q}"
                n = q
    except (OSError, ValueError):
        pass
    return n, note


def host_memory_available():
    """bytes this process can still allocate: min(MemAvailable, cgroup headroom)."""
    avail = None
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                avail = int(line.split()[1]) * 1024
    except OSError:
        pass
    try:
        mx = open("/sys/fs/cgroup/memory.max").read().strip()
        cur = int(open("/sys/fs/cgroup/memory.current").read().strip())
        if mx != "max":
            head = int(mx) - cur
            avail = head if avail is None else min(avail, head)
    except (OSError, ValueError):
        pass
    return avail


def load_profile_json(codec, *names):
    """The first of profiles/<names> that was measured on the code as it is now: tools/summarize_pmc.py stamps its
    summaries with the sha256 of every file under csrc/, and a summary whose kernel's files have changed since is not
    quoted (returns (None, why))."""
    from vgaudio_amd.build import profile_is_current
    stale = None
    for name in names:
        try:
            js = json.load(open(os.path.join(ROOT, "profiles", name)))
        except (OSError, ValueError):
            continue
        if profile_is_current(js, codec):
            return js, f"profiles/{name} (source hashes match)"
        stale = stale or name
    return None, (f"profiles/{stale} was measured on other kernel sources: not quoted" if stale else "no committed profile")


class Ctx:
    pass


def setup(args):
    import torch
    import torch.distributed as dist
    cx = Ctx()
    cx.torch, cx.dist = torch, dist
    cx.world = int(os.environ.get("WORLD_SIZE", "1"))
    cx.rank = int(os.environ.get("RANK", "0"))
    cx.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    from vgaudio_amd import _lib, device as vdev, distributed as vdist
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as plain `python bench.py --gpus N`: be the launcher -- N ranks of this same command line, one per GPU
        raise SystemExit(vdist.launch_local_ranks([os.path.abspath(__file__)] + sys.argv[1:], args.gpus))
    if args.gpus > 1 and cx.world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={cx.world}")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback exists)"
    if args.share_gpu:
        args.backend = "gloo"
    device_index = 0 if args.share_gpu else cx.local_rank
    torch.cuda.set_device(device_index)
    cx.dev = torch.device("cuda", device_index)
    cx.lib, cx.vdev, cx.vdist = _lib, vdev, vdist
    cx.L = _lib.lib()
    _lib.check(cx.L.vga_set_device(device_index))
    cx.backend = args.backend
    if cx.world > 1:
        vdist.init(args.backend, cx.dev, timeout_s=300)
    cx.st = lambda: torch.cuda.current_stream().cuda_stream
    return cx


def timed_steps(cx, args, step, n_events):
    """W warm-up steps (after a two-step device wake-up), then exactly K steps between barrier + synchronize;
    returns (seconds over K steps, max over ranks; per-step event lists)."""
    torch, dist = cx.torch, cx.dist
    # device wake-up (part of set-up, not of the W warm-up steps the contract asks for): the first two launches after
    # allocation run ~45 % slow (clock ramp + first-touch TLB fills of the 24 GB input, profiles/r01_c_kernel_trace.csv)
    for _ in range(2):
        step(None)
    for _ in range(args.warmup):
        step(None)
    torch.cuda.synchronize()
    if cx.world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(n_events)] for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(evs[k])
    torch.cuda.synchronize()
    if cx.world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if cx.world > 1:
        elapsed = cx.vdist.max_over_ranks(elapsed, cx.dev)
    return elapsed, evs


def mean_ms(evs, a, b):
    import numpy as np
    return float(np.mean([e[a].elapsed_time(e[b]) for e in evs])) if evs else 0.0


def base_line(args, cx, metric, value, ms_per_step, dtype, workload, config, roofline, cpu):
    out = {"metric": metric, "value": round(value, 2), "unit": "Msamples/s", "n_gpus": cx.world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": dtype, "data": "synthetic",
           "config": dict({"workload": workload, "parallelism": f"units sharded x{cx.world}"}, **config),
           "roofline": roofline, "cpu_baseline": cpu}
    if cpu:
        out["speedup_vs_cpu_baseline"] = round(value / cpu["value"], 2)
    return out


# ====================================================================================================== GC-ADPCM
def encode_benchmarks_single_thread(po, np):
    """The reference's own micro-benchmark shapes (VGAudio.Benchmark/AdpcmBenchmarks/EncodeBenchmarks.cs:8-24: 1 s of a
    48 kHz 440 Hz sine; GenerateCoefs, EncodeAdpcm, both), single thread, on the C restatement."""
    from vgaudio_amd import synth
    x = synth.sine(48000)
    coefs = po.gc_calculate_coefficients(x)

    def per_call(fn):
        fn()
        t0, k = time.perf_counter(), 0
        while time.perf_counter() - t0 < 0.4:
            fn()
            k += 1
        return (time.perf_counter() - t0) / max(k, 1)

    t_coefs = per_call(lambda: po.gc_calculate_coefficients(x))
    t_enc = per_call(lambda: po.gc_encode(x, coefs))
    t_both = per_call(lambda: po.gc_encode(x, po.gc_calculate_coefficients(x)))
    return {"shape": "1 s, 48 kHz, 440 Hz sine (EncodeBenchmarks.cs:8-24), 1 thread, C restatement",
            "GenerateCoefs_ms": round(t_coefs * 1e3, 3), "EncodeAdpcm_ms": round(t_enc * 1e3, 3),
            "GenerateCoefsAndEncode_ms": round(t_both * 1e3, 3),
            "GenerateCoefsAndEncode_Msamples_per_s": round(48000 / t_both / 1e6, 3)}


def pcie_rates(cx):
    """Page-locked copy rates of this box, GB/s: each direction alone (1 GiB) and both at once on two streams."""
    torch = cx.torch
    # PCIe rate of this box from page-locked memory (what the pipeline's rings see), 1 GiB each way
    pin = torch.empty(1 << 30, dtype=torch.uint8).pin_memory()
    dbuf = torch.empty(1 << 30, dtype=torch.uint8, device=cx.dev)
    rates = {}
    for name, (a, b) in (("h2d", (dbuf, pin)), ("d2h", (pin, dbuf))):
        a.copy_(b, non_blocking=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        a.copy_(b, non_blocking=True)
        torch.cuda.synchronize()
        rates[name] = (1 << 30) / (time.perf_counter() - t0) / 1e9
    # both directions at once (two streams): the DMA engines' aggregate, which is what a pipelined call is bound by
    pin2 = torch.empty(1 << 30, dtype=torch.uint8).pin_memory()
    dbuf2 = torch.empty(1 << 30, dtype=torch.uint8, device=cx.dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for timed in (False, True):                                      # first round: the new streams' queues come up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            with torch.cuda.stream(s1):
                dbuf.copy_(pin, non_blocking=True)
            with torch.cuda.stream(s2):
                pin2.copy_(dbuf2, non_blocking=True)
        torch.cuda.synchronize()
        rates["both_directions_aggregate"] = 4 * (1 << 30) / (time.perf_counter() - t0) / 1e9
    del pin, dbuf, pin2, dbuf2
    return rates


def pcie_bound_ms(rates, in_bytes, out_bytes):
    """the larger direction alone, or all bytes at the measured two-way aggregate, whichever is longer"""
    return max(in_bytes / rates["h2d"], out_bytes / rates["d2h"], (in_bytes + out_bytes) / rates["both_directions_aggregate"]) / 1e6


def host_call_e2e(cx, entry_point, call, in_bytes, out_bytes, units, total_samples, identical):
    """Times one host-pointer batch call (best of two after a warm-up call made by the caller) against the PCIe bound."""
    rates = pcie_rates(cx)
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        call()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    same = bool(identical())
    if not same:
        raise SystemExit(f"PARITY FAILURE: {entry_point} and the device-resident path disagree")
    bound = pcie_bound_ms(rates, in_bytes, out_bytes)
    return {"entry_point": entry_point + " (pageable host arrays in and out)", "units": units, "ms": round(best * 1e3, 1),
            "value": round(total_samples / best / 1e6, 1), "unit": "Msamples/s", "host_bytes_in": in_bytes, "host_bytes_out": out_bytes,
            "pcie_pinned_GBps": {k: round(v, 1) for k, v in rates.items()}, "pcie_bound_ms": round(bound, 1),
            "ratio_to_pcie_bound": round(best * 1e3 / bound, 2), "identical_to_device_path": same}


def measure_e2e(cx, args, pcm, n, coefs_dev, adpcm_dev, devices=None):
    """One vga_gcadpcm_encode_batch call through the host-pointer ABI (pageable numpy rows in and out); with `devices`
    the library spreads the call's channels over those GPUs (vga_set_devices: one process, several GPUs)."""
    import numpy as np
    torch, L, lib = cx.torch, cx.L, cx.lib
    if devices:
        lib.check(L.vga_set_devices((C.c_int * len(devices))(*devices), len(devices)))
    nch = pcm.shape[0]
    nb = cx.vdev.gc_byte_count(n)
    want = args.e2e_channels or nch
    avail = host_memory_available()
    per_ch = 2 * n + nb
    note = None
    if avail is not None and want * per_ch > 0.7 * avail:
        fit = max(64, int(0.7 * avail / per_ch) // 64 * 64)
        note = f"host memory allows {fit} of {want} channels ({avail / 2**30:.0f} GiB available)"
        want = min(want, fit)
    want = min(want, nch)
    rates = pcie_rates(cx)
    host = np.empty((want, n), dtype=np.int16)                       # pageable, like a managed short[][]
    for c0 in range(0, want, 256):
        host[c0:c0 + 256] = pcm[c0:c0 + 256, :n].cpu().numpy()
    outs = np.zeros((want, nb), dtype=np.uint8)
    cf = np.zeros(want * 16, dtype=np.int16)
    pp = (lib.i16p * want)(*[host[c].ctypes.data_as(lib.i16p) for c in range(want)])
    op = (lib.u8p * want)(*[outs[c].ctypes.data_as(lib.u8p) for c in range(want)])
    warm = min(want, 64)                                             # first call: pinned rings, code objects
    lib.check(L.vga_gcadpcm_encode_batch(pp, warm, n, 0, 0, cf.ctypes.data_as(lib.i16p), op))
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        lib.check(L.vga_gcadpcm_encode_batch(pp, want, n, 0, 0, cf.ctypes.data_as(lib.i16p), op))
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    st = (C.c_double * 32)()
    nf = L.vga_testing_last_pipeline_stats(st, 32)
    names = ["total", "setup", "feeders_memcpy_sum", "feeders_wait_slot_sum", "feeders_issue_sum", "slowest_feeder", "caller_wait_upload",
             "caller_launch", "caller_tail_sync", "drainers_wait_compute_sum", "drainers_wait_download_sum", "drainers_memcpy_sum",
             "slowest_drainer", "feeders", "drainers", "chunks", "units_per_chunk", "device_alloc", "entry_point",
             "feeders_chunk_boundary_sum", "feeders_final_sync_sum", "drainers_register_sum"]
    breakdown = {k: (int(st[i]) if 13 <= i <= 16 else round(st[i] * 1e3, 1)) for i, k in enumerate(names[:nf])}
    same = bool(np.array_equal(outs, adpcm_dev[:want, :nb].cpu().numpy()) and
                np.array_equal(cf.reshape(want, 16), coefs_dev[:want].cpu().numpy().reshape(want, 16)))
    if not same:
        raise SystemExit("PARITY FAILURE: the host-pointer ABI and the device-resident path disagree")
    in_bytes, out_bytes = want * 2 * n, want * nb
    pcie_ms = pcie_bound_ms(rates, in_bytes, out_bytes)
    e2e = {"entry_point": "vga_gcadpcm_encode_batch (pageable host arrays in and out)", "channels": want,
           "samples_per_channel": n, "ms": round(best * 1e3, 1), "value": round(want * n / best / 1e6, 1), "unit": "Msamples/s",
           "host_bytes_in": in_bytes, "host_bytes_out": out_bytes,
           "pcie_pinned_GBps": {k: round(v, 1) for k, v in rates.items()}, "pcie_bound_ms": round(pcie_ms, 1),
           "ratio_to_pcie_bound": round(best * 1e3 / pcie_ms, 2), "identical_to_device_path": same,
           "host_threads": f"{breakdown.get('feeders', '?')} feeder (direct copies from page-locked caller rows) + "
                           f"{breakdown.get('drainers', '?')} drainers (32 MB ring slots) + caller, 2 kernel lanes (host_pipeline.hpp)",
           "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"), "breakdown_ms": breakdown}
    if note:
        e2e["note"] = note
    if devices:
        lib.check(L.vga_set_devices(None, 0))
        e2e["devices"] = list(devices)
        e2e["entry_point"] += f", spread over {len(devices)} GPUs by vga_set_devices"
        e2e["pcie_bound_ms_all_links"] = round(pcie_ms / len(set(devices)), 1)
        e2e["ratio_to_pcie_bound_all_links"] = round(best * 1e3 / (pcie_ms / len(set(devices))), 2)
    return e2e


def measure_mixed_lengths(cx, args, equal_length_value):
    """The reference's batch path is a Parallel.ForEach over FILES of different lengths (VGAudio.Cli/Batch.cs:24-25 ->
    Convert.cs:19 -> GcAdpcmFormat.EncodeFromPcm16): mono files with log-uniform lengths between 1 s and 120 s, as many
    as hold the samples of configs[1] (4096 x 60 s).  Measured: (1) ONE ragged call, device resident -- the same kernels
    as the headline step, per-channel shapes from tables -- and through the host-pointer ABI (vga_gcadpcm_encode_batch_v);
    (2) the drop-in a maintainer would write first: 16 host threads, each converting whole files with one
    vga_gcadpcm_encode_batch call per file (a sample of the files); (3) the CPU port scheduled per file.  A sample of the
    files is held to the oracle bit for bit."""
    import threading
    import numpy as np
    torch, vdev, L, lib = cx.torch, cx.vdev, cx.L, cx.lib
    rng = np.random.default_rng(0xBA7C4)
    target = 4096 * 2_880_000 if (args.channels == 4096 and args.seconds == 60.0) else int(args.channels * args.seconds * 48000)
    lens, total = [], 0
    while total < target:
        n = int(np.exp(rng.uniform(np.log(48000.0), np.log(120 * 48000.0))))
        lens.append(n)
        total += n
    nfiles = len(lens)
    first_channel = 1 << 20                                # channels no other block of the bench uses
    rb = vdev.GcRaggedBatch(lens, cx.dev)
    pcm = rb.synth(first_channel=first_channel)
    adpcm = rb.alloc_adpcm()
    ws = torch.empty(max(rb.workspace_bytes, 16), dtype=torch.uint8, device=cx.dev)
    torch.cuda.synchronize()
    coefs = None
    for _ in range(2):
        coefs = rb.coefs(pcm, workspace=ws)
        rb.encode(pcm, coefs, out=adpcm)
    reps = 3
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3 * reps)]
    for k in range(reps):
        ev[3 * k].record()
        coefs = rb.coefs(pcm, workspace=ws)
        ev[3 * k + 1].record()
        rb.encode(pcm, coefs, out=adpcm)
        ev[3 * k + 2].record()
    torch.cuda.synchronize()
    coef_ms = float(np.mean([ev[3 * k].elapsed_time(ev[3 * k + 1]) for k in range(reps)]))
    enc_ms = float(np.mean([ev[3 * k + 1].elapsed_time(ev[3 * k + 2]) for k in range(reps)]))
    value = total / ((coef_ms + enc_ms) * 1e-3) / 1e6
    out = {"what": "mono files with log-uniform lengths in [1 s, 120 s] (seeded), as many as hold the samples of the headline "
                   "workload; a step = coefficient search + encode of every file, inputs resident in HBM",
           "reference": "VGAudio.Cli/Batch.cs:24-25 (Parallel.ForEach over files) -> GcAdpcmFormat.EncodeFromPcm16",
           "files": nfiles, "samples": total, "shortest": int(min(lens)), "longest": int(max(lens)),
           "one_ragged_call_device_resident": {
               "entry_points": "vga_gcadpcm_coefs_device_v + vga_gcadpcm_encode_device_v", "coefs_ms": round(coef_ms, 3),
               "encode_ms": round(enc_ms, 3), "ms": round(coef_ms + enc_ms, 3), "value": round(value, 1), "unit": "Msamples/s",
               "ratio_to_equal_length_step": round(value / equal_length_value, 3) if equal_length_value else None}}
    co = coefs.cpu().numpy().reshape(nfiles, 16)
    # ---- a sample of the files against the oracle (and as the CPU port scheduled per file): every 64th file of the
    # length-sorted list up to ~8 s of CPU work
    threads, cpu_note = usable_cpus()
    if not args.no_cpu_baseline:
        from oracle import pyoracle as po                 # the oracle appears in this leg only
        po.lib()
        order = np.argsort(lens)
        pick = [int(i) for i in order[::max(1, nfiles // 160)]]
        budget = int(3e6 * threads)                        # ~ samples the port encodes in a second on these threads
        chosen, acc = [], 0
        for i in pick:
            if acc + lens[i] > 8 * budget:
                break
            chosen.append(i)
            acc += lens[i]
        host_files = [pcm[int(rb.pcm_offsets[i]):int(rb.pcm_offsets[i]) + lens[i]].cpu().numpy() for i in chosen]
        results = [None] * len(chosen)
        nxt = [0]
        lock = threading.Lock()

        def worker():
            while True:
                with lock:
                    j = nxt[0]
                    nxt[0] += 1
                if j >= len(chosen):
                    return
                c = po.gc_calculate_coefficients(host_files[j])
                results[j] = (c, po.gc_encode(host_files[j], c))

        t0 = time.perf_counter()
        ts = [threading.Thread(target=worker) for _ in range(threads)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        dt = time.perf_counter() - t0
        for j, i in enumerate(chosen):
            row = adpcm[int(rb.adpcm_offsets[i]):int(rb.adpcm_offsets[i]) + int(rb.byte_counts[i])].cpu().numpy()
            if not (co[i].tolist() == results[j][0].tolist() and np.array_equal(row, results[j][1])):
                raise SystemExit(f"PARITY FAILURE: ragged call, file {i} ({lens[i]} samples) differs from the CPU restatement")
        out["bit_exact_files_checked"] = len(chosen)
        out["cpu_port_per_file"] = {"value": round(acc / dt / 1e6, 3), "unit": "Msamples/s", "cores": threads,
                                    "sample": f"{len(chosen)} of the files ({acc} samples), one task per file on {threads} threads "
                                              f"({cpu_note}), {dt:.1f} s wall"}
        del host_files, results
    if args.no_e2e:
        rb.close()
        return out
    # ---- through the host-pointer ABI: pageable numpy rows, one per file
    avail = host_memory_available()
    nb = rb.byte_counts
    use = nfiles
    need = np.cumsum(2 * np.asarray(lens, dtype=np.int64) + nb)
    if avail is not None and need[-1] > 0.6 * avail:
        use = max(1, int(np.searchsorted(need, 0.6 * avail)))
        out["note"] = f"host memory allows the first {use} of {nfiles} files through the host-pointer ABI ({avail / 2**30:.0f} GiB available)"
    host = [pcm[int(rb.pcm_offsets[i]):int(rb.pcm_offsets[i]) + lens[i]].cpu().numpy() for i in range(use)]
    outs = [np.zeros(int(nb[i]), dtype=np.uint8) for i in range(use)]
    cf = np.zeros((use, 16), dtype=np.int16)
    counts = np.asarray(lens[:use], dtype=np.int32)
    pp = (lib.i16p * use)(*[a.ctypes.data_as(lib.i16p) for a in host])
    op = (lib.u8p * use)(*[a.ctypes.data_as(lib.u8p) for a in outs])
    cp = counts.ctypes.data_as(C.POINTER(C.c_int))
    samples_used = int(counts.astype(np.int64).sum())
    # the library's cache of device blocks still holds the equal-length call's (44 GB, a few bytes too small for this
    # call's packed rows): start from an empty cache, then one untimed call of the full size fills it for the timed ones
    L.vga_release_cached_memory()
    lib.check(L.vga_gcadpcm_encode_batch_v(pp, cp, use, None, None, cf.ctypes.data_as(lib.i16p), op))                 # warm-up
    rates = pcie_rates(cx)
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        lib.check(L.vga_gcadpcm_encode_batch_v(pp, cp, use, None, None, cf.ctypes.data_as(lib.i16p), op))
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    st = (C.c_double * 32)()
    nf = L.vga_testing_last_pipeline_stats(st, 32)
    names = ["total", "setup", "feeders_memcpy_sum", "feeders_wait_slot_sum", "feeders_issue_sum", "slowest_feeder", "caller_wait_upload",
             "caller_launch", "caller_tail_sync", "drainers_wait_compute_sum", "drainers_wait_download_sum", "drainers_memcpy_sum",
             "slowest_drainer", "feeders", "drainers", "chunks", "units_per_chunk", "device_alloc", "entry_point",
             "feeders_chunk_boundary_sum", "feeders_final_sync_sum", "drainers_register_sum"]
    breakdown = {k: (int(st[i]) if 13 <= i <= 16 else round(st[i] * 1e3, 1)) for i, k in enumerate(names[:nf])}
    same = bool(np.array_equal(cf, co[:use]))
    flat_out = adpcm.cpu().numpy()
    for i in range(use):
        if not same:
            break
        same = np.array_equal(outs[i], flat_out[int(rb.adpcm_offsets[i]):int(rb.adpcm_offsets[i]) + int(nb[i])])
    if not same:
        raise SystemExit("PARITY FAILURE: vga_gcadpcm_encode_batch_v and the device-resident ragged path disagree")
    in_bytes, out_bytes = 2 * samples_used, int(nb[:use].sum())
    bound = pcie_bound_ms(rates, in_bytes, out_bytes)
    out["one_ragged_call_host_pointers"] = {
        "entry_point": "vga_gcadpcm_encode_batch_v (pageable host arrays in and out)", "files": use, "samples": samples_used,
        "ms": round(best * 1e3, 1), "value": round(samples_used / best / 1e6, 1), "unit": "Msamples/s",
        "pcie_bound_ms": round(bound, 1), "ratio_to_pcie_bound": round(best * 1e3 / bound, 2), "identical_to_device_path": True,
        "breakdown_ms": breakdown}
    # ---- the first drop-in: a worker per file (Batch.cs), every file one equal-length call.  A sample of the files (every
    # k-th, so that the length mix is the batch's), 16 host threads.
    step_k = max(1, use // 512)
    sample = list(range(0, use, step_k))
    workers = 16
    nxt = [0]
    lock = threading.Lock()
    errors = []

    def file_worker():
        while True:
            with lock:
                j = nxt[0]
                nxt[0] += 1
            if j >= len(sample):
                return
            i = sample[j]
            p1 = (lib.i16p * 1)(host[i].ctypes.data_as(lib.i16p))
            o1 = (lib.u8p * 1)(outs[i].ctypes.data_as(lib.u8p))
            rc = L.vga_gcadpcm_encode_batch(p1, 1, int(counts[i]), 0, 0, cf[i].ctypes.data_as(lib.i16p), o1)
            if rc:
                errors.append(rc)

    t0 = time.perf_counter()
    ts = [threading.Thread(target=file_worker) for _ in range(workers)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    dt = time.perf_counter() - t0
    done = int(counts[sample].astype(np.int64).sum())
    out["one_call_per_file"] = {"entry_point": "vga_gcadpcm_encode_batch, one channel per call, 16 host threads (pageable host arrays)",
                                "files": len(sample), "samples": done, "ms": round(dt * 1e3, 1), "value": round(done / dt / 1e6, 1),
                                "unit": "Msamples/s", "errors": len(errors)}
    if best:
        out["one_call_per_file"]["ragged_call_speedup"] = round((samples_used / best) / (done / dt), 1)
    del outs
    # ---- the same files through the other two codecs' ragged entry points ("a mixed-lengths measurement for
    # them"): ONE vga_adx_encode_batch_v / vga_hca_encode_batch_v call on pageable host rows (their kernels take one length a
    # launch: the rows travel in length buckets a quarter wide, zero-padded on the device -- DESIGN.md 4.6).  Both codecs are
    # PCIe-bound through host pointers; what a bucketed call loses shows as its ratio to that bound, next to the equal-length
    # calls' (the `e2e` blocks of the adx / hca lines).  A sample of files bit for bit against the oracle.
    try:
        out["other_codecs_one_ragged_call_host_pointers"] = mixed_other_codecs(cx, args, host, counts, rates)
    except SystemExit:
        raise
    except Exception as e:                              # noqa: BLE001 -- the line is worth more than this block
        out["other_codecs_one_ragged_call_host_pointers"] = {"error": f"{type(e).__name__}: {e}"}
    del host
    rb.close()
    return out


def pipeline_summary(L):
    """the calling thread's last pipelined call (vga_testing_last_pipeline_stats): when its upload ended, and the call"""
    st = (C.c_double * 32)()
    L.vga_testing_last_pipeline_stats(st, 32)
    return {"upload_done": round(st[5] * 1e3, 1), "total": round(st[0] * 1e3, 1), "chunks": int(st[15]),
            "drainers_memcpy_sum": round(st[11] * 1e3, 1)}


def mixed_other_codecs(cx, args, host, counts, rates):
    import numpy as np
    L, lib = cx.L, cx.lib
    use = len(host)
    samples = int(np.asarray(counts, dtype=np.int64).sum())
    res = {}
    check = 0 if args.no_cpu_baseline else 8
    if check:
        from oracle import pyoracle as po                 # the checker, after the timed calls
    pp = (lib.i16p * use)(*[a.ctypes.data_as(lib.i16p) for a in host])
    cp = np.asarray(counts, dtype=np.int32).ctypes.data_as(C.POINTER(C.c_int))
    pick = [int(i) for i in np.argsort(counts)[:: max(1, use // 8)]][:8]      # short files first: the oracle's time
    # ---- CRI ADX
    params = (lib.AdxParams * use)()
    for i in range(use):
        L.vga_adx_default_params(C.byref(params[i]))
    sizes = [L.vga_adx_encoded_byte_count(int(counts[i]), C.byref(params[i])) for i in range(use)]
    outs = [np.zeros(max(n_, 1), dtype=np.uint8) for n_ in sizes]
    op = (lib.u8p * use)(*[a.ctypes.data_as(lib.u8p) for a in outs])
    hist = np.zeros(use, dtype=np.int16)
    L.vga_release_cached_memory()
    lib.check(L.vga_adx_encode_batch_v(pp, cp, use, params, op, hist.ctypes.data_as(lib.i16p)))       # warm-up: fills the device cache
    t0 = time.perf_counter()
    lib.check(L.vga_adx_encode_batch_v(pp, cp, use, params, op, hist.ctypes.data_as(lib.i16p)))
    dt = time.perf_counter() - t0
    bound = pcie_bound_ms(rates, 2 * samples, int(sum(sizes)))
    res["adx"] = {"entry_point": "vga_adx_encode_batch_v", "files": use, "samples": samples, "ms": round(dt * 1e3, 1),
                  "value": round(samples / dt / 1e6, 1), "unit": "Msamples/s", "pcie_bound_ms": round(bound, 1),
                  "ratio_to_pcie_bound": round(dt * 1e3 / bound, 2), "pipeline_ms": pipeline_summary(L)}
    for i in pick[:check]:
        if not np.array_equal(outs[i][:sizes[i]], po.adx_encode(host[i], po.adx_params())):
            raise SystemExit(f"PARITY FAILURE: ragged ADX call, file {i} ({int(counts[i])} samples) differs from the CPU restatement")
    res["adx"]["bit_exact_files_checked"] = len(pick[:check])
    del outs, op
    # ---- CRI HCA: every file a mono stream, quality High
    cps = (lib.HcaParamsC * use)()
    infos = (lib.HcaInfoC * use)()
    for i in range(use):
        cps[i] = lib.HcaParamsC(2, 0, 0, 1, 48000, int(counts[i]), 0, 0, 0)
        lib.check(L.vga_hca_encoder_initialize(C.byref(cps[i]), C.byref(infos[i])))
    fsz = [infos[i].frame_count * infos[i].frame_size for i in range(use)]
    outs = [np.zeros(max(n_, 1), dtype=np.uint8) for n_ in fsz]
    op = (lib.u8p * use)(*[a.ctypes.data_as(lib.u8p) for a in outs])
    L.vga_release_cached_memory()
    lib.check(L.vga_hca_encode_batch_v(pp, use, cps, infos, op))
    t0 = time.perf_counter()
    lib.check(L.vga_hca_encode_batch_v(pp, use, cps, infos, op))
    dt = time.perf_counter() - t0
    bound = pcie_bound_ms(rates, 2 * samples, int(sum(fsz)))
    res["hca"] = {"entry_point": "vga_hca_encode_batch_v (every file a mono stream, quality High)", "files": use, "samples": samples,
                  "ms": round(dt * 1e3, 1), "value": round(samples / dt / 1e6, 1), "unit": "Msamples/s",
                  "pcie_bound_ms": round(bound, 1), "ratio_to_pcie_bound": round(dt * 1e3 / bound, 2), "pipeline_ms": pipeline_summary(L)}
    done = 0
    for i in pick[:max(0, check - 4)]:
        rc, _, want = po.hca_encode(host[i][None, :], po.hca_params(1, int(counts[i])))
        if rc != 0 or not np.array_equal(outs[i][:fsz[i]], want.reshape(-1)):
            raise SystemExit(f"PARITY FAILURE: ragged HCA call, file {i} ({int(counts[i])} samples) differs from the CPU restatement")
        done += 1
    res["hca"]["bit_exact_files_checked"] = done
    L.vga_release_cached_memory()
    return res


def measure_signal_sensitivity(cx, args):
    """The step on signals other than the synthetic generator's (vgaudio_amd/signals.py): the encoders'
    work depends on the data -- third trips of the retry loop (GcAdpcmEncoder.cs:127-170), how soon the seams between time
    pieces close -- so every class runs BASELINE configs[1]'s shape device-resident: coefficient search + encode between HIP
    events, the encoder's own counters (vga_testing_gc_encode_stats: wave-frames that took the cold block, seams closed inside
    their piece / left to gc_encode_chain_kernel, frames re-encoded by seam runs), the GC-ADPCM decoder and the ADX
    encoder + decoder on the same rows, and a sample of channels bit for bit against the oracle."""
    import numpy as np
    torch, vdev, L, lib = cx.torch, cx.vdev, cx.L, cx.lib
    from vgaudio_amd import signals
    nch = args.channels
    n = int(round(args.seconds * 48000))
    pcm = vdev.alloc_pcm(nch, n, cx.dev)
    adpcm = vdev.alloc_adpcm(nch, n, cx.dev)
    back = vdev.alloc_pcm(nch, n, cx.dev)
    ws = torch.empty(max(L.vga_gcadpcm_coefs_workspace_bytes(nch, n), 16), dtype=torch.uint8, device=cx.dev)
    p = lib.AdxParams()
    L.vga_adx_default_params(C.byref(p))
    anb = L.vga_adx_encoded_byte_count(n, C.byref(p))
    apitch = (anb + 15) // 16 * 16
    adx = torch.zeros((nch, apitch), dtype=torch.uint8, device=cx.dev)
    hist = torch.zeros(nch, dtype=torch.int16, device=cx.dev)
    status = torch.zeros(1, dtype=torch.int32, device=cx.dev)
    nb = vdev.gc_byte_count(n)
    reps = 2
    check_channels = 0 if args.no_cpu_baseline else 16
    threads, _ = usable_cpus()
    if check_channels:
        from oracle import pyoracle as po                 # the checker, after the timed launches of each class
    raw = (C.c_ulonglong * 8)()
    # the cold-block rates come from the library variant that counts them (the product does not: it costs 1 ms of the launch),
    # in a process of its own -- same classes, same shape
    cold_rates, cold_note = {}, "vgaudio_amd/libvgaudio_hip_stats.so (the product built with -DVGA_GC_STATS)"
    stats_lib = os.path.join(ROOT, "vgaudio_amd", "libvgaudio_hip_stats.so")
    if os.path.exists(stats_lib):
        import subprocess
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "signal_cold_rates.py"), str(nch), repr(args.seconds)],
                               env=dict(os.environ, VGAUDIO_HIP_LIBRARY=stats_lib), capture_output=True, text=True, timeout=300)
            cold_rates = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        except Exception as e:                          # noqa: BLE001 -- a diagnostic: the block is worth more than this field
            cold_note = f"not measured: {type(e).__name__}: {e}"
    else:
        cold_note = "not measured: vgaudio_amd/libvgaudio_hip_stats.so is not built (python -m vgaudio_amd.build --stats)"

    def ev():
        return torch.cuda.Event(enable_timing=True)

    classes = {}
    for cls in ("synthetic",) + tuple(signals.CLASSES):
        if cls == "synthetic":
            vdev.synth_pcm(nch, n, cx.dev, out=pcm)
        else:
            signals.device(cls, nch, n, cx.dev, out=pcm)
        coefs = vdev.gc_coefs(pcm, n, workspace=ws)            # (the first launches on fresh rows: not timed)
        vdev.gc_encode(pcm, n, coefs, out=adpcm)
        have_stats = L.vga_testing_gc_encode_stats(None, 1) == 0
        e = [ev() for _ in range(3 * reps)]
        for r in range(reps):
            e[3 * r].record()
            coefs = vdev.gc_coefs(pcm, n, workspace=ws)
            e[3 * r + 1].record()
            vdev.gc_encode(pcm, n, coefs, out=adpcm)
            e[3 * r + 2].record()
        torch.cuda.synchronize()
        coef_ms = min(e[3 * r].elapsed_time(e[3 * r + 1]) for r in range(reps))
        enc_ms = min(e[3 * r + 1].elapsed_time(e[3 * r + 2]) for r in range(reps))
        st = None
        if have_stats and L.vga_testing_gc_encode_stats(raw, 1) == 0:
            v = [int(x) / reps for x in raw]
            seams = v[0] + v[1]
            st = {"cold_block_wave_frame_rate": cold_rates.get(cls),
                  "seams_per_launch": round(seams), "seams_left_to_chain_kernel": round(v[1]),
                  "channels_walked_by_chain_kernel": round(v[5]),
                  "frames_reencoded_per_seam": round(v[2] / seams, 1) if seams else None,
                  "frames_reencoded_frac": round(v[2] / (nch * ((n + 13) // 14)), 5)}
        a = [ev() for _ in range(4)]
        a[0].record()
        vdev.gc_decode(adpcm, coefs, n, out=back)
        a[1].record()
        lib.check(L.vga_adx_encode_device(pcm.data_ptr(), pcm.stride(0), nch, n, C.byref(p), adx.data_ptr(), apitch, hist.data_ptr(), cx.st()))
        a[2].record()
        lib.check(L.vga_adx_decode_device(adx.data_ptr(), apitch, anb, nch, n, C.byref(p), back.data_ptr(), back.stride(0), status.data_ptr(), cx.st()))
        a[3].record()
        # (second pass of the three: the timed one)
        b = [ev() for _ in range(4)]
        b[0].record()
        vdev.gc_decode(adpcm, coefs, n, out=back)
        b[1].record()
        lib.check(L.vga_adx_encode_device(pcm.data_ptr(), pcm.stride(0), nch, n, C.byref(p), adx.data_ptr(), apitch, hist.data_ptr(), cx.st()))
        b[2].record()
        lib.check(L.vga_adx_decode_device(adx.data_ptr(), apitch, anb, nch, n, C.byref(p), back.data_ptr(), back.stride(0), status.data_ptr(), cx.st()))
        b[3].record()
        torch.cuda.synchronize()
        row = {"coefs_ms": round(coef_ms, 2), "encode_ms": round(enc_ms, 2), "step_ms": round(coef_ms + enc_ms, 2),
               "gc_decode_ms": round(min(a[0].elapsed_time(a[1]), b[0].elapsed_time(b[1])), 2),
               "adx_encode_ms": round(min(a[1].elapsed_time(a[2]), b[1].elapsed_time(b[2])), 2),
               "adx_decode_ms": round(min(a[2].elapsed_time(a[3]), b[2].elapsed_time(b[3])), 2), "encoder": st}
        if check_channels:
            idx = torch.linspace(0, nch - 1, check_channels, device=cx.dev).round().to(torch.int64)
            host = pcm[idx, :n].cpu().numpy()
            wc, wa = po.gc_encode_batch(host, threads=threads)
            want_adx, _ = po.adx_encode_batch(host, po.adx_params(), threads=threads)
            ok = (np.array_equal(coefs[idx].cpu().numpy().reshape(-1, 16), np.asarray(wc).reshape(-1, 16)) and
                  np.array_equal(adpcm[idx, :nb].cpu().numpy(), np.asarray(wa)[:, :nb]) and
                  np.array_equal(adx[idx, :anb].cpu().numpy(), want_adx))
            if not ok:
                raise SystemExit(f"PARITY FAILURE: signal class {cls}: GPU output differs from the CPU restatement")
            row["bit_exact_channels_checked"] = check_channels
        classes[cls] = row
    base = classes["synthetic"]["step_ms"]
    for row in classes.values():
        row["step_vs_synthetic"] = round(row["step_ms"] / base, 3) if base else None
    worst = max(classes, key=lambda k: classes[k]["step_ms"])
    return {"what": f"{nch} channels x {n} samples of each signal class (vgaudio_amd/signals.py), device-resident: coefficient "
                    f"search + encode (best of {reps} after one untimed pass), the encoder's counters per launch, GC-ADPCM decode, "
                    "ADX encode / decode on the same rows; sampled channels bit for bit against the oracle",
            "classes": classes, "cold_block_rate_source": cold_note, "slowest_class": worst, "slowest_step_vs_synthetic": classes[worst]["step_vs_synthetic"],
            "no_class_slower_than_1_5x_synthetic": bool(classes[worst]["step_vs_synthetic"] is not None and classes[worst]["step_vs_synthetic"] <= 1.5)}


def guarded(cx, line, fn, limit_s=240.0):
    """Runs fn() -- the N > 1 extras, which every rank takes part in -- so that rank 0's result line survives them: an
    exception becomes {"error": ...} in place of fn's result, and on rank 0 (line is not None) a watchdog thread prints
    the line with the time-out named and ends the process if fn has not returned after limit_s (a collective waiting
    for a rank that is gone never raises; the process group's own time-out, setup(), aborts the other ranks)."""
    import threading
    timer = None
    if line is not None:
        def give_up():
            out = dict(line, gather={"error": f"the multi-GPU extras did not finish within {limit_s:g} s; the line holds the timed steps only"})
            print(json.dumps(out), flush=True)
            os._exit(0)
        timer = threading.Timer(limit_s, give_up)
        timer.daemon = True
        timer.start()
    try:
        return fn()
    except Exception as e:                              # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"}
    finally:
        if timer is not None:
            timer.cancel()


def run_gc(args, cx):
    import numpy as np
    torch, vdev, L = cx.torch, cx.vdev, cx.L
    nch = args.channels
    n = int(round(args.seconds * 48000))
    first_channel = cx.rank * nch                     # contiguous channel block per GPU
    pcm = vdev.synth_pcm(nch, n, cx.dev, first_channel=first_channel)
    adpcm = vdev.alloc_adpcm(nch, n, cx.dev)
    ws = torch.empty(max(L.vga_gcadpcm_coefs_workspace_bytes(nch, n), 16), dtype=torch.uint8, device=cx.dev)
    torch.cuda.synchronize()
    state = {}

    def step(events):
        if events is not None:
            events[0].record()
        state["coefs"] = vdev.gc_coefs(pcm, n, workspace=ws)
        if events is not None:
            events[1].record()
        vdev.gc_encode(pcm, n, state["coefs"], out=adpcm)
        if events is not None:
            events[2].record()

    def multi_gpu_extras(adpcm, coefs):
        """N > 1 only, after the timed steps: the final gather (alone, and underneath the next step), its digest check, the
        per-shard digests, rank 0's step with the other GPUs idle.  Returns (gather, scaling, adpcm, coefs)."""
        scaling = None
        # the results of all channels end up in one place (GcAdpcmFormat.cs:65-74): ADPCM rows + coefs of
        # every rank gathered to rank 0 over RCCL/xGMI, in channel chunks; timed as steps that include it.
        nb = vdev.gc_byte_count(n)
        g = cx.vdist.BitstreamGather(nch, adpcm.stride(0), cx.dev)
        g.gather(adpcm, coefs)                              # warm-up: communicators, buffers
        torch.cuda.synchronize()
        cx.dist.barrier()
        # the gather by itself (nothing else running on any rank)
        t0 = time.perf_counter()
        g.gather(adpcm, coefs)
        torch.cuda.synchronize()
        cx.dist.barrier()
        ms_alone = cx.vdist.max_over_ranks(time.perf_counter() - t0, cx.dev) * 1e3
        # double-buffered outputs: the gather of step k runs on the process group's stream beside the kernels of
        # step k+1 (a caller converting batch after batch); the last gather is waited for inside the timed region
        bufs = [adpcm, torch.empty_like(adpcm)]
        keep, pending = [], []
        t0 = time.perf_counter()
        for k in range(args.steps):
            cur = bufs[k % 2]
            ck = vdev.gc_coefs(pcm, n, workspace=ws)
            vdev.gc_encode(pcm, n, ck, out=cur)
            for w in pending:
                w.wait()
            pending = g.gather(cur, ck, async_op=True)
            keep = [keep[-1], ck] if keep else [ck]         # coefficient rows stay alive until their gather is done
        for w in pending:
            w.wait()
        torch.cuda.synchronize()
        cx.dist.barrier()
        torch.cuda.synchronize()
        el = cx.vdist.max_over_ranks(time.perf_counter() - t0, cx.dev)
        ms_g = el / max(args.steps, 1) * 1e3
        adpcm, coefs = bufs[(args.steps - 1) % 2] if args.steps else adpcm, (keep[-1] if keep else coefs)
        # one more gather that carries every rank's 64-bit digest of what it sends; rank 0 recomputes the digests over what
        # arrived (BitstreamGather.verify) -- outside the timed region, it reads every row once more
        g.gather(adpcm, coefs, nbytes=nb)
        torch.cuda.synchronize()
        verified = g.verify(adpcm, coefs, nb) if cx.rank == 0 else None
        if verified is not None and verified.startswith("MISMATCH"):
            raise SystemExit("PARITY FAILURE: the rows rank 0 gathered are not the rows the ranks sent: " + verified)
        # every shard's output against the digest committed for it (tests/golden/gc_shard_oracle_digests.json: what the
        # ORACLE produces for the shard's 4096 channels, written in the build container): the first 8-GPU run has expected values
        mine = cx.vdist.rows_digest(adpcm[:nch], nb, coefs[:nch], first_channel)
        want = None
        try:
            gold = json.load(open(os.path.join(ROOT, "tests", "golden", "gc_shard_oracle_digests.json")))   # provenance: the oracle
            if gold["channels_per_shard"] == nch and gold["samples_per_channel"] == n and cx.rank < len(gold["shards"]):
                want = int(gold["shards"][cx.rank]["rows_digest"], 16)
        except (OSError, ValueError, KeyError):
            pass
        ok = torch.tensor([1 if (want is None or want == mine) else 0, 0 if want is None else 1], dtype=torch.int64,
                          device="cpu" if cx.backend == "gloo" else cx.dev)
        cx.dist.all_reduce(ok, op=cx.dist.ReduceOp.MIN)
        shard_digests = ("match the committed per-shard digests" if int(ok[0]) == 1 else "MISMATCH against the committed per-shard digests") \
            if int(ok[1]) == 1 else "no committed digest for this shape"
        if int(ok[1]) == 1 and int(ok[0]) != 1:
            raise SystemExit("PARITY FAILURE: a rank's output differs from the committed digest of its shard")
        # weak scaling inside this job: rank 0 repeats the step while every other GPU is idle
        torch.cuda.synchronize()
        cx.dist.barrier()
        alone_ms = None
        if cx.rank == 0:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = max(2, min(args.steps, 5))
            step(None)
            e0.record()
            for _ in range(reps):
                step(None)
            e1.record()
            torch.cuda.synchronize()
            alone_ms = e0.elapsed_time(e1) / reps
        cx.dist.barrier()
        job = cx.vdist.describe_job(cx.dev)                 # (a collective: every rank calls it)
        peers = cx.world - 1
        gather_gbs = (nch * nb + nch * 32) * peers / (ms_alone * 1e-3) / 1e9 if ms_alone > 0 else 0.0
        gather = {"what": "all ranks' ADPCM rows + coefficients to rank 0 (grouped send/recv in 512-channel chunks; the gather "
                          "of step k overlaps the kernels of step k+1)",
                  "backend": cx.backend + (" (rows staged through host memory)" if g.via_host else ""),
                  "job": job,
                  # rank 0 receives from every peer at once: seven peers = seven xGMI links of ~153 GB/s each on an 8-GPU node
                  "into_rank0_GBps_alone": round(gather_gbs, 2),
                  "xgmi_links_into_rank0": peers, "xgmi_link_GBps": XGMI_LINK_GBS,
                  "frac_of_xgmi_into_rank0": round(gather_gbs / (peers * XGMI_LINK_GBS), 4) if peers else None,
                  "bytes_per_peer": nch * nb + nch * 32, "ms_alone": round(ms_alone, 3),
                  "ms_per_step_with_gather": round(ms_g, 3), "ms_hidden": round(max(0.0, ms_alone - max(0.0, ms_g - ms_per_step)), 3),
                  "ms_exposed": round(max(0.0, ms_g - ms_per_step), 3),
                  "value_with_gather": round(samples_per_step / (ms_g * 1e-3) / 1e6, 2),
                  "verified": verified, "shard_digests": shard_digests}
        if alone_ms:
            scaling = {"rank0_alone_ms_per_step": round(alone_ms, 3), "all_ranks_ms_per_step": round(ms_per_step, 3),
                       "weak_scaling_efficiency": round(alone_ms / ms_per_step, 4),
                       "note": "same job: rank 0's step with the other GPUs idle / the max-over-ranks step with all of them busy"}

        return gather, scaling, adpcm, coefs

    elapsed, evs = timed_steps(cx, args, step, 3)
    coefs = state["coefs"]
    coef_ms, enc_ms = mean_ms(evs, 0, 1), mean_ms(evs, 1, 2)
    ms_per_step = elapsed / max(args.steps, 1) * 1e3
    samples_per_step = nch * n * cx.world
    value = samples_per_step / (ms_per_step * 1e-3) / 1e6

    gather = scaling = None
    if cx.rank != 0:
        if cx.world > 1:
            guarded(cx, None, lambda: multi_gpu_extras(adpcm, coefs))
        return None
    # the other direction (GcAdpcmDecoder.Decode), outside the timed steps: three launches between HIP events
    dec_ms = 0.0
    if cx.world == 1:
        back = vdev.alloc_pcm(nch, n, cx.dev)
        vdev.gc_decode(adpcm, coefs, n, out=back)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            vdev.gc_decode(adpcm, coefs, n, out=back)
        e1.record()
        torch.cuda.synchronize()
        dec_ms = e0.elapsed_time(e1) / 3
        del back
    verified = 0
    enc_bytes = ENC_BYTES_PER_SAMPLE * nch * n
    full = nch == 4096 and n == 2880000
    pmc, pmc_note = load_profile_json("gc", "r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_b_pmc_traffic.json")
    traffic = None
    enc_key = "gc_encode_persistent_kernel"            # from 2048 channels on (gc_encode_kernel.hip); the plain grid below
    if pmc and full:
        try:
            if enc_key not in pmc:
                enc_key = "gc_encode_kernel"
            traffic = round(pmc[enc_key]["traffic_bytes_per_launch"])
        except KeyError:
            pass
    # What actually binds the kernel (LABNOTES.md 4.1): wave-instruction issue.  From the committed SQ counter pass.
    issue = None
    sqj, sq_note = load_profile_json("gc", "r06_sq_counters.json", "r05_sq_counters.json", "r04_sq_counters.json", "r03_sq_counters.json", "r02_b_sq_counters.json")
    if sqj and full:
        try:
            sq = sqj[enc_key if enc_key in sqj else "gc_encode_kernel"]
            # the clock of the profiled launch: SQ_BUSY_CYCLES is summed over the 32 shader engines, each busy for the whole
            # launch.  (Rounds 1-3 divided by the waves' mean lifetime instead -- SQ_WAVE_CYCLES / SQ_WAVES -- which reads
            # as "the clock" only if every wave lives the whole launch; in the plain grid workgroups ended between 106 and
            # 162 ms, so that figure, 0.92, was the issue fraction of a wave's lifetime, not of the launch: 0.79.)
            clock_ghz = sq["SQ_BUSY_CYCLES"] / 32 / (sq["_dur_ms"] * 1e6)
            quads = sq["_dur_ms"] * 1e6 * clock_ghz / 4                # issue slots of one SIMD over the launch
            issue = {"valu_wave_instructions_per_launch": round(sq["SQ_INSTS_VALU"]),
                     "valu_issue_frac": round(sq["SQ_ACTIVE_INST_VALU"] / (1024 * quads), 3),
                     "valu_issue_frac_of_the_waves_lifetime": round(sq["SQ_ACTIVE_INST_VALU"] / (1024 * sq["SQ_WAVE_CYCLES"] / sq["SQ_WAVES"]), 3),
                     "profiled_launch_ms": round(sq["_dur_ms"], 1), "clock_GHz": round(clock_ghz, 2),
                     "note": "SQ_ACTIVE_INST_VALU (quad-cycles) / (1024 SIMDs x launch cycles / 4); profiled launch, not this run"}
        except (KeyError, ZeroDivisionError):
            pass
    achieved = enc_bytes / (enc_ms * 1e-3) / 1e9 if enc_ms > 0 else 0.0
    # one encode = gc_encode_persistent_kernel (time pieces from a queue, seams closed inside; below 512 channels on 256 CUs:
    # gc_encode_kernel<false> + gc_encode_seam_kernel) + gc_encode_chain_kernel + gc_encode_kernel<true> (chain and repair
    # return at once unless a seam stayed open); launch_ms spans them, the rocprofv3 kernel stats under profiles/ list them
    # separately (their averages add up to it)
    persistent = nch >= 512                 # plan_encode_pieces: one channel group per 32 workgroups (gc_encode_kernel.hip)
    roofline = {"bound": "hbm", "kernel": "gc_encode_persistent_kernel" if persistent else "gc_encode_kernel", "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": traffic, "traffic_source": pmc_note, "algorithmic_bytes_per_launch": enc_bytes, "launch_ms": round(enc_ms, 3),
                # every timed step's launch, in order (launch_ms is their mean): back-to-back steps run a few ms slower than a
                # lone one on a box whose clock settles under sustained load
                "launch_ms_each": [round(e[1].elapsed_time(e[2]), 2) for e in evs],
                "launch_parts": (["gc_encode_persistent_kernel (pieces from a queue, seams inside)"] if persistent else
                                 ["gc_encode_kernel<false>", "gc_encode_seam_kernel"]) + ["gc_encode_chain_kernel", "gc_encode_kernel<true>"],
                "other_kernels": {"gc_coefs_kernel": {
                    "launch_ms": round(coef_ms, 3),
                    "achieved": round(COEF_BYTES_PER_SAMPLE * nch * n / (coef_ms * 1e-3) / 1e9, 2) if coef_ms > 0 else 0.0},
                    "gc_decode_direct_kernel (+fixup, tail; not part of the step)": {
                        "traffic": (pmc.get("gc_decode_direct_kernel") or {}).get("traffic_bytes_per_launch") if pmc and full else None,
                        "launch_ms": round(dec_ms, 3),
                        "achieved": round(ENC_BYTES_PER_SAMPLE * nch * n / (dec_ms * 1e-3) / 1e9, 2) if dec_ms > 0 else 0.0}},
                "pipeline_achieved": round(PIPE_BYTES_PER_SAMPLE * nch * n / ((coef_ms + enc_ms) * 1e-3) / 1e9, 2)
                if coef_ms + enc_ms > 0 else 0.0,
                "issue": issue, "issue_source": sq_note}

    cpu = None
    if not args.no_cpu_baseline and cx.world == 1:          # the CPU leg runs at N=1 only (rank 0's host cores)
        threads, cpu_note = usable_cpus()
        cch = args.cpu_channels or min(nch, 24 * threads)
        host = pcm[:cch, :n].cpu().numpy()
        from oracle import pyoracle as po                 # the oracle appears in this leg only
        po.lib()
        t1 = time.perf_counter()
        ref_coefs, ref_adpcm = po.gc_encode_batch(host, threads=threads)
        dt = time.perf_counter() - t1
        # the baseline's output doubles as the checker of this run's: every sampled channel, bit for bit
        nb = vdev.gc_byte_count(n)
        if not (np.array_equal(coefs[:cch].cpu().numpy().reshape(cch, 16), np.asarray(ref_coefs).reshape(cch, 16)) and
                np.array_equal(adpcm[:cch, :nb].cpu().numpy(), np.asarray(ref_adpcm)[:, :nb])):
            raise SystemExit("PARITY FAILURE: GPU output differs from the CPU restatement")
        verified = cch
        cpu = {"value": round(cch * n / dt / 1e6, 3), "unit": "Msamples/s", "cores": threads, "kind": "port",
               "sample": f"{cch} of the same channels x {n} samples, one task per channel on {threads} threads "
                         f"({cpu_note}); C restatement of GcAdpcmFormat.EncodeFromPcm16 (the C# reference cannot "
                         f"be built here), {dt:.1f} s wall",
               "encode_benchmarks_single_thread": encode_benchmarks_single_thread(po, np)}
        del host, ref_adpcm

    out = base_line(args, cx, "Msamples/s encoded (GC-ADPCM, 4096 ch) at 1/2/4/8 GPUs; % HBM roofline", value, ms_per_step,
                    "int32", f"BASELINE configs[1]: {nch} mono channels x 48 kHz x {args.seconds:g} s GC-ADPCM coefficient "
                             f"search + encode per GPU",
                    {"channels_per_gpu": nch, "samples_per_channel": n, "bit_exact_channels_checked": verified}, roofline, cpu)
    if cx.world > 1:
        # the line so far is complete without what follows: if the extras fail, it is printed with the failure named; if
        # they hang (a collective that never returns), a watchdog prints it and ends this rank
        res = guarded(cx, out, lambda: multi_gpu_extras(adpcm, coefs))
        if isinstance(res, tuple):
            gather, scaling, adpcm, coefs = res
        else:
            gather = res
    if gather:
        out["gather"] = gather
    if scaling:
        out["weak_scaling"] = scaling
    if not args.no_e2e and cx.world == 1:
        out["e2e"] = measure_e2e(cx, args, pcm, n, coefs, adpcm)
    if not args.no_mixed and cx.world == 1:
        del pcm, adpcm, ws
        torch.cuda.empty_cache()
        try:
            out["mixed_lengths"] = measure_mixed_lengths(cx, args, value)
        except SystemExit:
            raise
        except Exception as e:                          # noqa: BLE001 -- the line is worth more than this block
            out["mixed_lengths"] = {"error": f"{type(e).__name__}: {e}"}
    if not args.no_signals and cx.world == 1:
        torch.cuda.empty_cache()
        try:
            out["signal_sensitivity"] = measure_signal_sensitivity(cx, args)
        except SystemExit:
            raise
        except Exception as e:                          # noqa: BLE001 -- the line is worth more than this block
            out["signal_sensitivity"] = {"error": f"{type(e).__name__}: {e}"}
    if not args.no_e2e and cx.world > 1:
        # one process, N GPUs: the same 4096-channel call as the N = 1 line's e2e block, its channels spread over all
        # GPUs of the job by the library (the other ranks are idle at the final barrier meanwhile)
        devs = [0] * cx.world if args.share_gpu else list(range(cx.world))
        try:
            out["e2e_multi"] = measure_e2e(cx, args, pcm, n, coefs, adpcm, devices=devs)
        except Exception as e:                          # noqa: BLE001 -- the line is worth more than this block
            out["e2e_multi"] = {"error": f"{type(e).__name__}: {e}"}
    return out


# ====================================================================================================== CRI ADX
def run_adx(args, cx):
    import numpy as np
    torch, vdev, L, lib = cx.torch, cx.vdev, cx.L, cx.lib
    nch = args.channels
    n = int(round(args.seconds * 48000))
    pcm = vdev.synth_pcm(nch, n, cx.dev, first_channel=cx.rank * nch)
    p = lib.AdxParams()
    L.vga_adx_default_params(C.byref(p))
    nb = L.vga_adx_encoded_byte_count(n, C.byref(p))
    pitch = (nb + 15) // 16 * 16
    adx = torch.zeros((nch, pitch), dtype=torch.uint8, device=cx.dev)
    hist = torch.zeros(nch, dtype=torch.int16, device=cx.dev)
    status = torch.zeros(1, dtype=torch.int32, device=cx.dev)
    dec = vdev.alloc_pcm(nch, n, cx.dev)
    torch.cuda.synchronize()

    def step(events):
        if events is not None:
            events[0].record()
        lib.check(L.vga_adx_encode_device(pcm.data_ptr(), pcm.stride(0), nch, n, C.byref(p), adx.data_ptr(), pitch,
                                          hist.data_ptr(), cx.st()))
        if events is not None:
            events[1].record()
        lib.check(L.vga_adx_decode_device(adx.data_ptr(), pitch, nb, nch, n, C.byref(p), dec.data_ptr(), dec.stride(0),
                                          status.data_ptr(), cx.st()))
        if events is not None:
            events[2].record()

    elapsed, evs = timed_steps(cx, args, step, 3)
    if int(status.item()) != 0:
        raise SystemExit("ADX decode reported a bad frame")
    enc_ms, dec_ms = mean_ms(evs, 0, 1), mean_ms(evs, 1, 2)
    ms_per_step = elapsed / max(args.steps, 1) * 1e3
    value = nch * n * cx.world / (ms_per_step * 1e-3) / 1e6
    if cx.rank != 0:
        return None
    bytes_launch = ADX_BYTES_PER_SAMPLE * nch * n
    achieved = bytes_launch / (enc_ms * 1e-3) / 1e9 if enc_ms > 0 else 0.0
    pmc, pmc_note = load_profile_json("adx", "r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_b_pmc_traffic.json")
    traffic = None
    if pmc and nch == 4096 and n == 2880000:
        traffic = (pmc.get("adx_encode_fs18_direct_kernel") or {}).get("traffic_bytes_per_launch")
    roofline = {"bound": "hbm", "kernel": "adx_encode_fs18_direct_kernel", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": pmc_note,
                "algorithmic_bytes_per_launch": bytes_launch, "launch_ms": round(enc_ms, 3),
                "launch_parts": ["adx_encode_fs18_direct_kernel", "adx_encode_fs18_fixup_kernel", "adx_encode_fs18_tail_kernel",
                                 "adx_encode_fs18_direct_kernel<.., true> (REPAIR: returns at once unless many seams stayed open)"],
                "other_kernels": {"adx_decode_fs18_direct_kernel (+fixup, tail)": {
                    "launch_ms": round(dec_ms, 3),
                    "achieved": round(bytes_launch / (dec_ms * 1e-3) / 1e9, 2) if dec_ms > 0 else 0.0,
                    "traffic": (pmc.get("adx_decode_fs18_direct_kernel") or {}).get("traffic_bytes_per_launch")
                    if pmc and nch == 4096 and n == 2880000 else None}},
                "pipeline_achieved": round(2 * bytes_launch / ((enc_ms + dec_ms) * 1e-3) / 1e9, 2) if enc_ms + dec_ms > 0 else 0.0}
    cpu, verified = None, 0
    if not args.no_cpu_baseline and cx.world == 1:
        threads, cpu_note = usable_cpus()
        cch = args.cpu_channels or min(nch, 64 * threads)
        host = pcm[:cch, :n].cpu().numpy()
        from oracle import pyoracle as po
        po.lib()
        t1 = time.perf_counter()
        want, whist = po.adx_encode_batch(host, po.adx_params(), threads=threads)
        wdec = po.adx_decode_batch(want, n, po.adx_params(), threads=threads)
        dt = time.perf_counter() - t1
        if not (np.array_equal(adx[:cch, :nb].cpu().numpy(), want) and np.array_equal(hist[:cch].cpu().numpy(), whist) and
                np.array_equal(dec[:cch, :n].cpu().numpy(), wdec)):
            raise SystemExit("PARITY FAILURE: GPU ADX output differs from the CPU restatement")
        verified = cch
        cpu = {"value": round(cch * n / dt / 1e6, 3), "unit": "Msamples/s", "cores": threads, "kind": "port",
               "sample": f"{cch} of the same channels x {n} samples, encode + decode, one task per channel on {threads} threads "
                         f"({cpu_note}); C restatement of CriAdxFormat.EncodeFromPcm16 / ToPcm16, {dt:.1f} s wall"}
    # Looping files: CriAdxFormat.cs:59-62 pads every looping stream whose loop start is not a multiple
    # of the alignment -- LoopStart = 1000 in a mono file: Padding = 1024 - 1000 = 24 -- and until round 6 a padded stream took
    # the lane-per-channel kernels (decode of this shape: 1.1 s).  The same channels with that padding, outside the timed region.
    looping = None
    if cx.world == 1:
        lp = lib.AdxParams()
        L.vga_adx_default_params(C.byref(lp))
        lp.padding = 24
        lnb = L.vga_adx_encoded_byte_count(n, C.byref(lp))
        lpitch = (lnb + 15) // 16 * 16
        ladx = torch.zeros((nch, lpitch), dtype=torch.uint8, device=cx.dev)
        lhist = torch.zeros(nch, dtype=torch.int16, device=cx.dev)
        ldec = vdev.alloc_pcm(nch, n, cx.dev)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        best = None
        for _ in range(3):
            ev[0].record()
            lib.check(L.vga_adx_encode_device(pcm.data_ptr(), pcm.stride(0), nch, n, C.byref(lp), ladx.data_ptr(), lpitch, lhist.data_ptr(), cx.st()))
            ev[1].record()
            lib.check(L.vga_adx_decode_device(ladx.data_ptr(), lpitch, lnb, nch, n, C.byref(lp), ldec.data_ptr(), ldec.stride(0), status.data_ptr(), cx.st()))
            ev[2].record()
            torch.cuda.synchronize()
            t = (ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]))
            best = t if best is None or sum(t) < sum(best) else best
        looping = {"loop_start": 1000, "padding": 24, "encode_ms": round(best[0], 2), "decode_ms": round(best[1], 2),
                   "vs_unpadded": {"encode": round(best[0] / enc_ms, 3) if enc_ms > 0 else None,
                                   "decode": round(best[1] / dec_ms, 3) if dec_ms > 0 else None}}
        if not args.no_cpu_baseline:
            from oracle import pyoracle as po
            idx = [0, 1, nch // 2, nch - 1]
            hostl = pcm[idx, :n].cpu().numpy()
            wl, whl = po.adx_encode_batch(hostl, po.adx_params(padding=24), threads=4)
            wdl = po.adx_decode_batch(wl, n, po.adx_params(padding=24), threads=4)
            if not (np.array_equal(ladx[idx, :lnb].cpu().numpy(), wl) and np.array_equal(lhist[idx].cpu().numpy(), whl) and
                    np.array_equal(ldec[idx, :n].cpu().numpy(), wdl)):
                raise SystemExit("PARITY FAILURE: padded (looping) ADX output differs from the CPU restatement")
            looping["bit_exact_channels_checked"] = len(idx)
        del ladx, ldec
    e2e = None
    if not args.no_e2e and cx.world == 1:
        avail = host_memory_available()
        want_ch = nch
        if avail is not None and want_ch * (2 * n + nb) > 0.7 * avail:
            want_ch = max(64, int(0.7 * avail / (2 * n + nb)) // 64 * 64)
        want_ch = min(want_ch, nch)
        host = np.empty((want_ch, n), dtype=np.int16)
        for c0 in range(0, want_ch, 256):
            host[c0:c0 + 256] = pcm[c0:c0 + 256, :n].cpu().numpy()
        outs = np.zeros((want_ch, nb), dtype=np.uint8)
        hh = np.zeros(want_ch, dtype=np.int16)
        pp = (lib.i16p * want_ch)(*[host[c].ctypes.data_as(lib.i16p) for c in range(want_ch)])
        op = (lib.u8p * want_ch)(*[outs[c].ctypes.data_as(lib.u8p) for c in range(want_ch)])
        lib.check(L.vga_adx_encode_batch(pp, min(want_ch, 64), n, C.byref(p), op, hh.ctypes.data_as(lib.i16p)))     # warm-up
        e2e = host_call_e2e(cx, "vga_adx_encode_batch",
                            lambda: lib.check(L.vga_adx_encode_batch(pp, want_ch, n, C.byref(p), op, hh.ctypes.data_as(lib.i16p))),
                            want_ch * 2 * n, want_ch * nb, want_ch, want_ch * n,
                            lambda: np.array_equal(outs, adx[:want_ch, :nb].cpu().numpy()) and
                            np.array_equal(hh, hist[:want_ch].cpu().numpy()))
        del host, outs
    out = base_line(args, cx, "Msamples/s (CRI ADX encode+decode round trip, 4096 ch)", value, ms_per_step, "int32",
                     f"BASELINE configs[2]: {nch} mono channels x 48 kHz x {args.seconds:g} s CRI ADX (18-byte frames, v4, "
                     f"linear) encode + decode round trip per GPU",
                     {"channels_per_gpu": nch, "samples_per_channel": n, "bit_exact_channels_checked": verified}, roofline, cpu)
    if e2e:
        out["e2e"] = e2e
    if looping:
        out["looping"] = looping
    return out


# ====================================================================================================== CRI HCA
def run_hca(args, cx):
    import numpy as np
    torch, vdev, L, lib = cx.torch, cx.vdev, cx.L, cx.lib
    ns = args.streams
    n = int(round(args.seconds * 48000))
    hp = lib.HcaParamsC(2, 0, 0, 2, 48000, n, 0, 0, 0)            # quality High, stereo
    info = lib.HcaInfoC()
    lib.check(L.vga_hca_encoder_initialize(C.byref(hp), C.byref(info)))
    spcm = vdev.synth_pcm(ns * 2, n, cx.dev, first_channel=cx.rank * ns * 2)     # [ns*2, pitch]: stream-major planar
    ch_pitch = spcm.stride(0)
    fbytes = info.frame_count * info.frame_size
    fpitch = (fbytes + 8 + 15) // 16 * 16
    frames = torch.zeros((ns, fpitch), dtype=torch.uint8, device=cx.dev)
    status = torch.zeros(1, dtype=torch.int32, device=cx.dev)
    wsb = L.vga_hca_decode_workspace_bytes(C.byref(info), ns)
    ws = torch.empty(wsb, dtype=torch.uint8, device=cx.dev)
    dec = torch.zeros_like(spcm)
    torch.cuda.synchronize()

    def step(events):
        if events is not None:
            events[0].record()
        lib.check(L.vga_hca_encode_device(spcm.data_ptr(), 2 * ch_pitch, ch_pitch, ns, n, C.byref(info), frames.data_ptr(), fpitch,
                                          status.data_ptr(), cx.st()))
        if events is not None:
            events[1].record()

    elapsed, evs = timed_steps(cx, args, step, 2)
    enc_ms = mean_ms(evs, 0, 1)
    # decode beside it (not part of the step)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for _ in range(2):
        e[0].record()
        lib.check(L.vga_hca_decode_device(C.byref(info), frames.data_ptr(), fpitch, ns, dec.data_ptr(), 2 * ch_pitch, ch_pitch,
                                          ws.data_ptr(), wsb, status.data_ptr(), cx.st()))
        e[1].record()
        torch.cuda.synchronize()
    dec_ms = e[0].elapsed_time(e[1])
    if int(status.item()) != 0:
        raise SystemExit("HCA kernels reported an error status")
    ms_per_step = elapsed / max(args.steps, 1) * 1e3
    chs = ns * 2 * n
    value = chs * cx.world / (ms_per_step * 1e-3) / 1e6
    if cx.rank != 0:
        return None
    bytes_launch = (2.0 + info.frame_size * info.frame_count / (2.0 * n)) * chs if n else 0.0
    achieved = bytes_launch / (enc_ms * 1e-3) / 1e9 if enc_ms > 0 else 0.0
    pmc, pmc_note = load_profile_json("hca", "r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_b_pmc_traffic.json")
    traffic = None
    if pmc and ns == 1024 and n == 2880000:
        traffic = (pmc.get("hca_encode_wave_kernel") or pmc.get("hca_encode_kernel") or {}).get("traffic_bytes_per_launch")
    roofline = {"bound": "hbm", "kernel": "hca_encode_wave_kernel", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": pmc_note,
                "algorithmic_bytes_per_launch": bytes_launch, "launch_ms": round(enc_ms, 3),
                "other_kernels": {"hca_scan_kernel + hca_frames_kernel (decode)": {
                    "launch_ms": round(dec_ms, 3),
                    "achieved": round(bytes_launch / (dec_ms * 1e-3) / 1e9, 2) if dec_ms > 0 else 0.0}},
                "mfma": "the exact path uses no MFMA: a dense 128x128 DCT-IV reassociates the f64 sums (LABNOTES.md 4.4; "
                        "measured variant: tools/bench_hca_mfma.py, profiles/)"}
    cpu, verified = None, 0
    if not args.no_cpu_baseline and cx.world == 1:
        threads, cpu_note = usable_cpus()
        workers = max(1, threads - 1)                          # Cli/Batch.cs:25: ProcessorCount - 1 workers, one file each
        cs = args.cpu_channels or min(ns, 16 * workers)
        host = np.stack([spcm[2 * s:2 * s + 2, :n].cpu().numpy() for s in range(cs)])
        from oracle import pyoracle as po
        po.lib()
        t1 = time.perf_counter()
        rc, oinfo, want = po.hca_encode_batch(host, po.hca_params(2, n), threads=workers)
        dt = time.perf_counter() - t1
        if rc != 0 or not np.array_equal(frames[:cs, :fbytes].cpu().numpy(), want):
            raise SystemExit("PARITY FAILURE: GPU HCA frames differ from the CPU restatement")
        rc, wdec = po.hca_decode_batch(oinfo, want, threads=workers)
        got = dec[:2 * cs, :n].cpu().numpy().reshape(cs, 2, n)
        if rc != 0 or not np.array_equal(got, wdec):
            raise SystemExit("PARITY FAILURE: GPU HCA decode differs from the CPU restatement")
        verified = cs
        cpu = {"value": round(cs * 2 * n / dt / 1e6, 3), "unit": "Msamples/s", "cores": workers, "kind": "port",
               "sample": f"{cs} of the same stereo streams x {n} samples, encode only, one task per STREAM on {workers} workers "
                         f"(the reference has no intra-file parallelism, CriHcaFormat.cs:53-81; Cli/Batch.cs:24-25 runs "
                         f"ProcessorCount-1 files at a time; {cpu_note}); C restatement of CriHcaFormat.EncodeFromPcm16, {dt:.1f} s wall"}
    e2e = None
    if not args.no_e2e and cx.world == 1:
        avail = host_memory_available()
        want_s = ns
        per_stream = 2 * 2 * n + fbytes
        if avail is not None and want_s * per_stream > 0.7 * avail:
            want_s = max(16, int(0.7 * avail / per_stream) // 16 * 16)
        want_s = min(want_s, ns)
        host = np.empty((want_s * 2, n), dtype=np.int16)
        for c0 in range(0, want_s * 2, 256):
            host[c0:c0 + 256] = spcm[c0:c0 + 256, :n].cpu().numpy()
        outs = np.zeros((want_s, fbytes), dtype=np.uint8)
        pp = (lib.i16p * (want_s * 2))(*[host[c].ctypes.data_as(lib.i16p) for c in range(want_s * 2)])
        op = (lib.u8p * want_s)(*[outs[k].ctypes.data_as(lib.u8p) for k in range(want_s)])
        info2 = lib.HcaInfoC()
        lib.check(L.vga_hca_encode_batch(pp, min(want_s, 16), C.byref(hp), C.byref(info2), op))                     # warm-up
        e2e = host_call_e2e(cx, "vga_hca_encode_batch",
                            lambda: lib.check(L.vga_hca_encode_batch(pp, want_s, C.byref(hp), C.byref(info2), op)),
                            want_s * 2 * 2 * n, want_s * fbytes, want_s, want_s * 2 * n,
                            lambda: np.array_equal(outs, frames[:want_s, :fbytes].cpu().numpy()))
        del host, outs
    out = base_line(args, cx, "Msamples/s encoded (CRI HCA, 1024 stereo streams, quality High; channel-samples)", value, ms_per_step,
                     "f64", f"BASELINE configs[3]: {ns} stereo streams x 48 kHz x {args.seconds:g} s CRI HCA encode, quality High "
                            f"({info.frame_size}-byte frames, {info.frame_count} frames per stream) per GPU",
                     {"streams_per_gpu": ns, "channels_per_stream": 2, "samples_per_channel": n, "bit_exact_streams_checked": verified},
                     roofline, cpu)
    if e2e:
        out["e2e"] = e2e
    return out


def other_configs(args, cx):
    """BASELINE configs[2] and configs[3] on the same clock as the headline: three steps each of the ADX round trip and the
    HCA encode at their full shapes, with the roofline of the dominant kernel and a CPU leg on a small sample."""
    import copy
    import torch
    out = {}
    for codec, fn in (("adx", run_adx), ("hca", run_hca)):
        a = copy.copy(args)
        a.codec, a.steps, a.warmup, a.no_e2e = codec, 3, 1, True
        a.channels, a.streams, a.seconds = 4096, 1024, 60.0
        threads, _ = usable_cpus()
        a.cpu_channels = 8 * threads if codec == "adx" else 2 * max(1, threads - 1)
        t0 = time.perf_counter()
        r = fn(a, cx)
        torch.cuda.empty_cache()
        rf = r["roofline"]
        out["configs[2] adx" if codec == "adx" else "configs[3] hca"] = {
            "metric": r["metric"], "value": r["value"], "unit": r["unit"], "steps": r["steps"], "warmup": r["warmup"],
            "ms_per_step": r["ms_per_step"], "dtype": r["dtype"], "workload": r["config"]["workload"],
            "roofline": {k: rf[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source",
                                            "algorithmic_bytes_per_launch", "launch_ms", "other_kernels") if k in rf},
            "cpu_baseline": r["cpu_baseline"], "speedup_vs_cpu_baseline": r.get("speedup_vs_cpu_baseline"),
            "bit_exact_units_checked": r["config"].get("bit_exact_channels_checked", r["config"].get("bit_exact_streams_checked")),
            "wall_s": round(time.perf_counter() - t0, 1)}
        if "looping" in r:
            out["configs[2] adx"]["looping"] = r["looping"]
    return out


def main():
    args = parse()
    cx = setup(args)
    out = {"gc": run_gc, "adx": run_adx, "hca": run_hca}[args.codec](args, cx)
    if cx.world > 1 and cx.rank != 0:
        # this rank's tensors are gone with run_*'s frame; give the blocks back too: rank 0 may still be measuring
        # (e2e_multi spreads a call over every GPU of the job) while this rank only waits at the final barrier
        import gc
        gc.collect()
        cx.torch.cuda.empty_cache()
    if args.codec == "gc" and cx.world == 1 and out is not None and not args.no_other_configs and args.channels == 4096 and args.seconds == 60.0:
        import torch
        torch.cuda.empty_cache()
        out["other_configs"] = other_configs(args, cx)
    if cx.rank == 0 and out is not None:
        print(json.dumps(out), flush=True)
    if cx.world > 1:
        try:
            cx.dist.barrier()
            cx.dist.destroy_process_group()
        except Exception:                               # noqa: BLE001 -- the line is out; a rank that failed is gone
            pass


if __name__ == "__main__":
    main()
