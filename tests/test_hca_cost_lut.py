"""The HCA encoder's bit-cost look-up (CostLut, vgaudio_amd/csrc/hca_encode_kernel.hip; round 5) restated in numpy with the
reference's constants (tests/golden/hca_tables.json): for every resolution the cost of a coefficient steps up once, at a
threshold found with the quantiser's own arithmetic (CriHcaEncoder.cs:589-591, CriHcaTables.cs:68-78); the kernel finds a
coefficient's rank among the fifteen sorted thresholds of its sign from a bucket of the magnitude's bits plus ONE exact
compare.  That is only right if no bucket holds two thresholds and every magnitude ScaleSpectra can produce has a bucket --
properties of the constants, checked here -- and then the rank must equal the count of thresholds the magnitude reaches, for
values on, just below and just above every threshold and every bucket edge.  CalculateUsedBits' own table look-up
(:554-597) is the reference for the cost itself.  No GPU needed."""
import json
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
T = json.load(open(os.path.join(HERE, "golden", "hca_tables.json")))
BITS = np.array(T["packed"]["QuantizeSpectrumBits"], dtype=np.int64)          # [8][16]
MAXBITS = np.array(T["packed"]["QuantizedSpectrumMaxBits"], dtype=np.int64)
G = T["generated_tables_test"]
COST_BUCKETS = 208


def f64(v):
    """the fixture holds IEEE bit patterns as hex strings"""
    return np.float64(struct.unpack(">d", bytes.fromhex(v))[0]) if isinstance(v, str) else np.float64(v)


def tables():
    step = np.array([f64(x) for x in G["QuantizerStepSize"]])
    inv = np.array([f64(x) for x in G["QuantizerInverseStepSize"]])
    maxv = np.array(G["ResolutionMaxValue"], dtype=np.int64)
    return step, inv, maxv


def used_bits_reference(x, r, step, inv, maxv):
    """CalculateUsedBits' per-coefficient cost (CriHcaEncoder.cs:570-597)"""
    if r == 0:
        return 0
    if r < 8:
        q = int(np.float64(x) * inv[r] + (inv[r] + 1)) - int(inv[r] + 0.5 - 8)
        return int(BITS[r][q])
    dead = struct.unpack("<d", struct.pack("<q", struct.unpack("<q", struct.pack("<d", step[r] / 2))[0] - int(maxv[r] + 1)))[0]
    return int(MAXBITS[r] - 1 + (1 if abs(x) >= dead else 0))


def threshold_of(r, neg, step, inv, maxv):
    if r >= 8:
        return struct.unpack("<d", struct.pack("<q", struct.unpack("<q", struct.pack("<d", step[r] / 2))[0] - int(maxv[r] + 1)))[0]
    up = inv[r] + 1
    down = int(inv[r] + 0.5 - 8)
    b0 = BITS[r][8]
    k = 1
    while k < 8 and BITS[r][8 + k] == b0:
        k += 1
    index_of = lambda x: int(np.float64(x) * inv[r] + up) - down
    as_bits = lambda d: struct.unpack("<q", struct.pack("<d", d))[0]
    as_f64 = lambda b: struct.unpack("<d", struct.pack("<q", b))[0]
    lo, hi = 0, as_bits(0.999999999999)
    while hi - lo > 1:
        mid = (lo + hi) // 2
        m = as_f64(mid)
        beyond = index_of(-m) <= 8 - k if neg else index_of(m) >= 8 + k
        if beyond:
            hi = mid
        else:
            lo = mid
    return as_f64(hi)


def hi_dword(d):
    return struct.unpack("<q", struct.pack("<d", abs(d)))[0] >> 32


def build():
    step, inv, maxv = tables()
    thr = [[threshold_of(r, sg, step, inv, maxv) for r in range(1, 16)] for sg in range(2)]
    order = [sorted(range(15), key=lambda i: (thr[sg][i], i)) for sg in range(2)]
    srt = [[thr[sg][i] for i in order[sg]] + [float("inf")] for sg in range(2)]
    key_base = (hi_dword(min(srt[0][0], srt[1][0])) >> 16) - 1
    rank_base = np.zeros((2, COST_BUCKETS), dtype=np.int64)
    for sg in range(2):
        for b in range(COST_BUCKETS):
            lower = 0.0 if b == 0 else struct.unpack("<d", struct.pack("<q", ((key_base + b) << 16) << 32))[0]
            upper = struct.unpack("<d", struct.pack("<q", ((key_base + b + 1) << 16) << 32))[0]
            inside = sum(1 for t in srt[sg][:15] if lower < t < upper)
            below = sum(1 for t in srt[sg][:15] if t <= lower)
            assert inside <= 1, (sg, b, "two thresholds share a bucket")
            assert not (b == 0 and (inside or below))
            rank_base[sg][b] = below
    return thr, order, srt, key_base, rank_base


def rank_by_lut(x, srt, key_base, rank_base):
    sg = 1 if (struct.unpack("<q", struct.pack("<d", x))[0] < 0) else 0        # the sign BIT (-0.0 counts as negative)
    b = min(max((hi_dword(x) >> 16) - key_base, 0), COST_BUCKETS - 1)
    rb = int(rank_base[sg][b])
    return sg, rb + (1 if abs(x) >= srt[sg][rb] else 0)


def test_the_constants_fit_the_buckets():
    thr, order, srt, key_base, rank_base = build()
    assert (hi_dword(0.999999999999) >> 16) - key_base < COST_BUCKETS
    for sg in range(2):
        gaps = [srt[sg][i + 1] / srt[sg][i] for i in range(14)]
        assert min(gaps) > 1.25, gaps                      # a bucket is at most 6.25 % wide


def test_rank_and_cost_agree_with_the_reference_around_every_threshold_and_edge():
    step, inv, maxv = tables()
    thr, order, srt, key_base, rank_base = build()
    rng = np.random.default_rng(5)
    probes = [0.0, -0.0, 1e-300, -1e-300, 0.999999999999, -0.999999999999]
    for sg in range(2):
        for t in srt[sg][:15]:
            for k in (-2, -1, 0, 1, 2):
                v = struct.unpack("<d", struct.pack("<q", struct.unpack("<q", struct.pack("<d", t))[0] + k))[0]
                probes += [v, -v]
    for b in range(COST_BUCKETS + 1):
        edge = struct.unpack("<d", struct.pack("<q", ((key_base + b) << 16) << 32))[0]
        for k in (-1, 0, 1):
            v = struct.unpack("<d", struct.pack("<q", struct.unpack("<q", struct.pack("<d", edge))[0] + k))[0]
            if v <= 0.999999999999:
                probes += [v, -v]
    probes += list(np.clip(rng.standard_normal(3000) * 0.3, -0.999999999999, 0.999999999999))
    probes += list(np.exp(rng.uniform(np.log(1e-6), 0, 3000)) * rng.choice([-1.0, 1.0], 3000) * 0.999999999999)
    for x in probes:
        x = float(x)
        sg, rank = rank_by_lut(x, srt, key_base, rank_base)
        want_rank = sum(1 for t in srt[sg][:15] if abs(x) >= t)
        assert rank == want_rank, (x, sg, rank, want_rank)
        reached = set(order[sg][:rank])                    # resolutions (index r - 1) whose threshold the coefficient reaches
        for r in range(1, 16):
            base = int(BITS[r][8]) if r < 8 else int(MAXBITS[r] - 1)
            got = base + (1 if (r - 1) in reached else 0)
            assert got == used_bits_reference(x, r, step, inv, maxv), (x, r, got)
