#!/usr/bin/env python3
"""bench.py -- headline benchmark: GC-ADPCM batch encode (coefficient search + encode),
BASELINE.json configs[1]: 4096 independent mono channels x 48 kHz x 60 s per GPU.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path over the batch, inputs resident in HBM:
  gc_coefs_kernel  (GcAdpcmCoefficients.CalculateCoefficients for every channel)
  gc_encode_kernel (GcAdpcmEncoder.Encode for every channel)
Channels shard across GPUs with no data-path collective (weak scaling: every rank owns
its own 4096 channels; BASELINE configs[4] = 8 x configs[1]).  Rank 0 prints ONE JSON line.

roofline  : dominant kernel gc_encode_kernel, algorithmic bytes = 2 B/sample read +
            8/14 B/sample written (SURVEY.md 8d "encode-only"), / mean HIP-event duration.
cpu_baseline: the oracle (C restatement of the reference, "port") run with the reference's
            scheduling (one task per channel on all host cores) on a bounded channel subset; the
            same leg compares its output with this run's GPU output for those channels, bit for bit
            (config.bit_exact_channels_checked; 0 when the leg does not run).  The oracle is touched
            nowhere else.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
ENC_BYTES_PER_SAMPLE = 2.0 + 8.0 / 14.0
COEF_BYTES_PER_SAMPLE = 2.0
PIPE_BYTES_PER_SAMPLE = 4.0 + 8.0 / 14.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--channels", type=int, default=4096, help="channels per GPU")
    ap.add_argument("--seconds", type=float, default=60.0, help="audio seconds per channel @48 kHz")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-channels", type=int, default=0, help="channels in the CPU baseline sample (0 = auto)")
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


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback exists)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from vgaudio_amd import _lib, device as vdev, distributed as vdist
    _lib.check(_lib.lib().vga_set_device(local_rank))
    if world > 1:
        vdist.init("nccl", dev)

    nch = args.channels
    n = int(round(args.seconds * 48000))
    first_channel = rank * nch                     # contiguous channel block per GPU

    # ---- inputs resident in HBM before the timed region
    pcm = vdev.synth_pcm(nch, n, dev, first_channel=first_channel)
    adpcm = vdev.alloc_adpcm(nch, n, dev)
    L = _lib.lib()
    ws = torch.empty(max(L.vga_gcadpcm_coefs_workspace_bytes(nch, n), 16), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    def step(events=None):
        if events is not None:
            events[0].record()
        coefs = vdev.gc_coefs(pcm, n, workspace=ws)
        if events is not None:
            events[1].record()
        vdev.gc_encode(pcm, n, coefs, out=adpcm)
        if events is not None:
            events[2].record()
        return coefs

    # device wake-up (part of set-up, not of the W warm-up steps the contract asks for): the first two
    # launches after allocation run ~45 % slow (clock ramp + first-touch TLB fills of the 24 GB input,
    # profiles/r01_c_kernel_trace.csv: 339, 341, then 234 ms), whatever W the caller picks
    for _ in range(2):
        step()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()

    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    t0 = time.perf_counter()
    coefs = None
    for k in range(args.steps):
        coefs = step(evs[k])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    coef_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in evs])) if args.steps else 0.0
    enc_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in evs])) if args.steps else 0.0

    if world > 1:
        elapsed = vdist.max_over_ranks(elapsed, dev)
        # the only exchange: 32 B/channel of coefficients gathered for the caller (RCCL over xGMI)
        all_coefs = vdist.gather_channel_metadata(coefs, [nch] * world)
        assert all_coefs.shape[0] == nch * world

    samples_per_step = nch * n * world
    ms_per_step = elapsed / max(args.steps, 1) * 1e3
    value = samples_per_step / (ms_per_step * 1e-3) / 1e6

    out = None
    if rank == 0:
        verified = 0

        enc_bytes = ENC_BYTES_PER_SAMPLE * nch * n
        # HBM traffic per launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 per the gfx950
        # correction, + WRITE_SIZE; profiles/r01_pmc_traffic.json).  Only valid for the profiled shape.
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            if nch == 4096 and n == 2880000:
                traffic = round(pmc["gc_encode_kernel"]["traffic_bytes_per_launch"])
        except (OSError, KeyError, ValueError):
            pass
        # What actually binds the kernel: wave-instruction issue.  From the committed SQ
        # counter pass (profiles/r01_l_sq_counters.json): VALU-active quad-cycles / (SIMDs x kernel quad-cycles).
        issue = None
        try:
            sq = json.load(open(os.path.join(ROOT, "profiles", "r01_l_sq_counters.json")))["gc_encode_kernel"]
            if nch == 4096 and n == 2880000:
                clk_quads = sq["SQ_WAVE_CYCLES"] / sq["SQ_WAVES"]          # every wave lives the whole launch
                issue = {"valu_wave_instructions_per_launch": round(sq["SQ_INSTS_VALU"]),
                         "valu_issue_frac": round(sq["SQ_ACTIVE_INST_VALU"] / (1024 * clk_quads), 3),
                         "profiled_launch_ms": round(sq["_dur_ms"], 1),
                         "note": "1024 SIMDs x one wave-instruction per 4 cycles; profiled launch, not this run"}
        except (OSError, KeyError, ValueError, ZeroDivisionError):
            pass
        achieved = enc_bytes / (enc_ms * 1e-3) / 1e9 if enc_ms > 0 else 0.0
        # one encode = gc_encode_kernel<false> (all time pieces at once) + gc_encode_seam_kernel + gc_encode_kernel<true>
        # (the repair launch: returns at once unless a seam stayed open); launch_ms spans the three, the rocprofv3
        # kernel stats under profiles/ list them separately (their averages add up to it)
        roofline = {"bound": "hbm", "kernel": "gc_encode_kernel", "achieved": round(achieved, 2),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                    "traffic": traffic, "algorithmic_bytes_per_launch": enc_bytes, "launch_ms": round(enc_ms, 3),
                    "launch_parts": ["gc_encode_kernel<false>", "gc_encode_seam_kernel", "gc_encode_kernel<true>"],
                    "other_kernels": {"gc_coefs_kernel": {
                        "launch_ms": round(coef_ms, 3),
                        "achieved": round(COEF_BYTES_PER_SAMPLE * nch * n / (coef_ms * 1e-3) / 1e9, 2) if coef_ms > 0 else 0.0}},
                    "pipeline_achieved": round(PIPE_BYTES_PER_SAMPLE * nch * n / ((coef_ms + enc_ms) * 1e-3) / 1e9, 2)
                    if coef_ms + enc_ms > 0 else 0.0,
                    "issue": issue}

        cpu = None
        if not args.no_cpu_baseline and world == 1:          # the CPU leg runs at N=1 only (rank 0's host cores)
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
                             f"be built here), {dt:.1f} s wall"}

        out = {"metric": "Msamples/s encoded (GC-ADPCM, 4096 ch) at 1/2/4/8 GPUs; % HBM roofline",
               "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
               "config": {"workload": f"BASELINE configs[1]: {nch} mono channels x 48 kHz x {args.seconds:g} s "
                                      f"GC-ADPCM coefficient search + encode per GPU",
                          "channels_per_gpu": nch, "samples_per_channel": n, "parallelism": f"channels sharded x{world}",
                          "bit_exact_channels_checked": verified},
               "roofline": roofline, "cpu_baseline": cpu}
        if cpu:
            out["speedup_vs_cpu_baseline"] = round(value / cpu["value"], 2)
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
