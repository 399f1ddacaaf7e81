"""One ragged call of the ADX and the HCA encoder on bench.py's mixed-lengths file set (10 008 mono files, log-uniform 1-120 s,
pageable host rows), with the chunks of plan_buckets (vgaudio_amd/csrc/host_batch.hpp) shortest first and longest first, in one process on one box: wall time, where the pipeline's threads spent it
(vga_testing_last_pipeline_stats), and that both orders write the same bytes.

    python tools/time_ragged_host.py [--files N] [--codecs adx hca]"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NAMES = ["total", "setup", "feeders_memcpy", "feeders_wait_slot", "feeders_issue", "slowest_feeder", "caller_wait_upload",
         "caller_launch", "caller_tail_sync", "drainers_wait_compute", "drainers_wait_download", "drainers_memcpy",
         "slowest_drainer", "feeders", "drainers", "chunks", "units_per_chunk", "device_alloc", "entry_point",
         "feeders_chunk_boundary", "feeders_final_sync", "drainers_register"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=0, help="use only the first N files (0: all)")
    ap.add_argument("--codecs", nargs="+", default=["adx", "hca"])
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--orders", type=int, nargs="+", default=[1, 0], help="1: shortest chunks first, 0: longest first")
    ap.add_argument("--feeders", type=int, nargs="+", default=[0], help="feeder threads (0: the pipeline's one; negative: a stream each)")
    ap.add_argument("--chunk-units", type=int, nargs="+", default=[0], help="largest number of units in a chunk (0: the entry point's 1024)")
    ap.add_argument("--transfer", type=int, nargs="+", default=[0], help="page-locked rows: 0 = by transfer kernels (round 6), 1 = one copy per row")
    ap.add_argument("--lanes", type=int, nargs="+", default=[0], help="compute streams of the pipeline (0: the entry point's choice)")
    args = ap.parse_args()
    import torch
    from vgaudio_amd import _lib as lib, device as vdev
    L = lib.lib()
    dev = torch.device("cuda", 0)
    lib.check(L.vga_set_device(0))
    rng = np.random.default_rng(0xBA7C4)                     # bench.py measure_mixed_lengths
    target = 4096 * 2_880_000
    lens, total = [], 0
    while total < target:
        n = int(np.exp(rng.uniform(np.log(48000.0), np.log(120 * 48000.0))))
        lens.append(n)
        total += n
    if args.files:
        lens = lens[:args.files]
    # host memory: the PCM rows, (adxdec) as many bytes again for the decoded rows, the encoded rows -- keep to a third of what is free
    try:
        avail = [int(ln.split()[1]) * 1024 for ln in open("/proc/meminfo") if ln.startswith("MemAvailable")][0]
    except (OSError, IndexError):
        avail = 64 << 30
    need = 2 * np.cumsum(np.asarray(lens, dtype=np.int64)) * (2.4 if "adxdec" in args.codecs else 1.4)
    if need[-1] > avail / 2:
        lens = lens[:max(1, int(np.searchsorted(need, avail / 2)))]
        print(f"host memory ({avail / 2**30:.0f} GiB available): the first {len(lens)} files only")
    use = len(lens)
    rb = vdev.GcRaggedBatch(lens, dev)
    pcm = rb.synth(first_channel=1 << 20)
    host = [pcm[int(rb.pcm_offsets[i]):int(rb.pcm_offsets[i]) + lens[i]].cpu().numpy() for i in range(use)]
    del pcm
    rb.close()
    torch.cuda.empty_cache()
    samples = int(np.sum(np.asarray(lens, dtype=np.int64)))
    counts = np.asarray(lens, dtype=np.int32)
    pp = (lib.i16p * use)(*[a.ctypes.data_as(lib.i16p) for a in host])
    cp = counts.ctypes.data_as(C.POINTER(C.c_int))
    st = (C.c_double * 32)()
    print(f"{use} files, {samples} samples, {2 * samples / 1e9:.2f} GB of PCM")

    def run(name, call, outs):
        keep = {}
        for shortest_first, units, feeders, transfer, lanes in [(o, u, f, t, cl) for cl in args.lanes for t in args.transfer for f in args.feeders for u in args.chunk_units for o in args.orders] * 2:
            L.vga_testing_host_transfer_this_thread(transfer)
            L.vga_testing_host_compute_lanes_this_thread(lanes)
            L.vga_testing_buckets_order_this_thread(1 if shortest_first else 2)
            L.vga_testing_host_pipeline_this_thread(feeders, 0, units, 0)
            try:
                L.vga_release_cached_memory()
                call()                                        # warm-up: fills the library's cache of device blocks
                best, every = None, []
                for _ in range(args.reps):
                    t0 = time.perf_counter()
                    call()
                    dt = time.perf_counter() - t0
                    best = dt if best is None else min(best, dt)
                    L.vga_testing_last_pipeline_stats(st, 32)
                    every.append("%.0f (upload %.0f)" % (dt * 1e3, st[5] * 1e3))
                nf = L.vga_testing_last_pipeline_stats(st, 32)
            finally:
                L.vga_testing_buckets_order_this_thread(0)
                L.vga_testing_host_pipeline_this_thread(0, 0, 0, 0)
                L.vga_testing_host_transfer_this_thread(0)
                L.vga_testing_host_compute_lanes_this_thread(0)
            b = {k: (int(st[i]) if 13 <= i <= 16 else round(st[i] * 1e3, 1)) for i, k in enumerate(NAMES[:nf])}
            digest = [hash(o.tobytes()) for o in outs[::97]] + [int(sum(int(o[:64].sum()) for o in outs))]
            tag = "shortest first" if shortest_first else "longest first "
            tag += " transfer kernels" if transfer == 0 else " a copy per row  "
            tag += f" lanes {lanes}"
            print(f"{name} {tag} units/chunk<={units or 1024:5d} feeders {feeders:2d} {best * 1e3:7.1f} ms   upload (slowest feeder) {b['slowest_feeder']:6.1f}  after it "
                  f"{b['total'] - b['slowest_feeder']:6.1f}  chunks {b['chunks']}  drainers' memcpy {b['drainers_memcpy']:6.1f}  "
                  f"feeders' issue {b['feeders_issue']:6.1f}" + ("   every call: " + ", ".join(every) if args.reps > 2 else ""), flush=True)
            if keep.setdefault("digest", digest) != digest:
                raise SystemExit(f"{name}: the two chunk orders wrote different bytes")
            for o in outs:
                o[:] = 0

    if "gc" in args.codecs:
        nb = [L.vga_gcadpcm_sample_count_to_byte_count(int(n_)) for n_ in counts]
        outs = [np.zeros(max(n_, 1), dtype=np.uint8) for n_ in nb]
        op = (lib.u8p * use)(*[a.ctypes.data_as(lib.u8p) for a in outs])
        coefs = np.zeros((use, 16), dtype=np.int16)
        run("gc ", lambda: lib.check(L.vga_gcadpcm_encode_batch_v(pp, cp, use, None, None, coefs.ctypes.data_as(lib.i16p), op)), outs)
        del outs, op
    if "adx" in args.codecs:
        params = (lib.AdxParams * use)()
        for i in range(use):
            L.vga_adx_default_params(C.byref(params[i]))
        sizes = [L.vga_adx_encoded_byte_count(int(counts[i]), C.byref(params[i])) for i in range(use)]
        outs = [np.zeros(max(n_, 1), dtype=np.uint8) for n_ in sizes]
        op = (lib.u8p * use)(*[a.ctypes.data_as(lib.u8p) for a in outs])
        hist = np.zeros(use, dtype=np.int16)
        run("adx", lambda: lib.check(L.vga_adx_encode_batch_v(pp, cp, use, params, op, hist.ctypes.data_as(lib.i16p))), outs)
        if "adxdec" in args.codecs:                       # the same files back: vga_adx_decode_batch_v on what the call above wrote
            lib.check(L.vga_adx_encode_batch_v(pp, cp, use, params, op, hist.ctypes.data_as(lib.i16p)))
            enc = outs
            alens = np.asarray(sizes, dtype=np.int32)
            back = [np.zeros(max(int(n_), 1), dtype=np.int16) for n_ in counts]
            bp = (lib.i16p * use)(*[a.ctypes.data_as(lib.i16p) for a in back])
            ep = (lib.u8p * use)(*[a.ctypes.data_as(lib.u8p) for a in enc])
            run("adxdec", lambda: lib.check(L.vga_adx_decode_batch_v(ep, alens.ctypes.data_as(C.POINTER(C.c_int)), use, cp, params, bp)), back)
            del back, bp
        del outs, op
    if "hca" in args.codecs:
        cps = (lib.HcaParamsC * use)()
        infos = (lib.HcaInfoC * use)()
        for i in range(use):
            cps[i] = lib.HcaParamsC(2, 0, 0, 1, 48000, int(counts[i]), 0, 0, 0)
            lib.check(L.vga_hca_encoder_initialize(C.byref(cps[i]), C.byref(infos[i])))
        fsz = [infos[i].frame_count * infos[i].frame_size for i in range(use)]
        outs = [np.zeros(max(n_, 1), dtype=np.uint8) for n_ in fsz]
        op = (lib.u8p * use)(*[a.ctypes.data_as(lib.u8p) for a in outs])
        run("hca", lambda: lib.check(L.vga_hca_encode_batch_v(pp, use, cps, infos, op)), outs)
    L.vga_release_cached_memory()


if __name__ == "__main__":
    main()
