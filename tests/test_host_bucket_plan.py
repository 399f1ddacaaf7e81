"""The ragged ADX / HCA entry points cut a batch into chunks of one parameter group and similar length
(plan_buckets, vgaudio_amd/csrc/host_batch.hpp; the reference runs a worker per FILE, VGAudio.Cli/Batch.cs:24-25, and
has no such plan).  Host arithmetic, no GPU: every unit in exactly one chunk, a chunk's units within a quarter of each
other, never two groups in a chunk, the bounds respected -- and the chunks in either order (HCA runs them longest first, so
that what is left to compute and download when the upload ends is the batch's smallest chunk; ADX shortest first:
host_batch.hpp has the measurement)."""
import ctypes as C

import numpy as np
import pytest

from vgaudio_amd import _lib


def plan(group, length, max_units=1024, max_volume=1024 * 2880000, shortest_first=False):
    L = _lib.lib()
    n = len(group)
    g = np.asarray(group, dtype=np.int32)
    ln = np.asarray(length, dtype=np.int32)
    order = np.zeros(max(n, 1), dtype=np.int32)
    cap = n + 2
    begin = np.zeros(cap + 1, dtype=np.int32)
    clen = np.zeros(cap, dtype=np.int32)
    cgrp = np.zeros(cap, dtype=np.int32)
    ip = C.POINTER(C.c_int)
    k = L.vga_testing_plan_buckets(g.ctypes.data_as(ip), ln.ctypes.data_as(ip), n, max_units, max_volume, int(not shortest_first),
                                   order.ctypes.data_as(ip), begin.ctypes.data_as(ip), clen.ctypes.data_as(ip), cgrp.ctypes.data_as(ip), cap)
    assert k >= 0
    return order[:n], begin[:k + 1], clen[:k], cgrp[:k]


def check(group, length, order, begin, clen, cgrp, max_units, max_volume):
    n = len(group)
    assert sorted(order.tolist()) == list(range(n))                      # every unit once
    assert begin[0] == 0 and begin[-1] == n and np.all(np.diff(begin) > 0)
    for k in range(len(clen)):
        units = order[begin[k]:begin[k + 1]]
        ls = np.asarray(length)[units]
        assert set(np.asarray(group)[units].tolist()) == {int(cgrp[k])}  # one parameter group
        assert int(ls.max()) == int(clen[k]) == int(ls[-1])              # the chunk's longest is its last unit (capi_hca.hip reads it there)
        assert np.all(np.diff(ls) >= 0)
        assert int(ls.max()) <= int(ls.min()) + int(ls.min()) // 4 + 1024   # within a quarter of each other
        assert len(units) <= max_units
        assert len(units) == 1 or len(units) * max(int(ls.max()), 1) <= max_volume


def log_uniform_lengths(n, seed):
    rng = np.random.default_rng(seed)
    return np.exp(rng.uniform(np.log(48000.0), np.log(120 * 48000.0), n)).astype(np.int32)


@pytest.mark.parametrize("n,groups,max_units,seed", [(10008, 1, 1024, 1), (3000, 3, 256, 2), (17, 2, 4, 3), (1, 1, 1024, 4)])
def test_every_unit_in_one_chunk_of_its_group_and_length(n, groups, max_units, seed):
    rng = np.random.default_rng(seed)
    length = log_uniform_lengths(n, seed)
    group = rng.integers(0, groups, n).astype(np.int32)
    vol = 1024 * 2880000
    check(group, length, *plan(group, length, max_units, vol), max_units, vol)
    check(group, length, *plan(group, length, max_units, vol, shortest_first=True), max_units, vol)


def test_both_orders_hold_the_same_chunks():
    length = log_uniform_lengths(10008, 5)
    group = np.zeros(len(length), dtype=np.int32)
    order, begin, clen, cgrp = plan(group, length)
    assert np.all(np.diff(clen) <= 0)
    assert int(clen[0]) == int(length.max())
    # the last chunk is what remains to be computed and downloaded after the upload: the shortest files
    assert int(clen[-1]) < 2 * 48000
    o2, b2, c2, g2 = plan(group, length, shortest_first=True)
    assert np.all(np.diff(c2) >= 0)
    # the same chunks, in the opposite order
    a = [tuple(order[begin[k]:begin[k + 1]].tolist()) for k in range(len(clen))]
    b = [tuple(o2[b2[k]:b2[k + 1]].tolist()) for k in range(len(c2))]
    assert a == b[::-1]


def test_volume_bound_and_empty_units():
    # 3000 two-minute files: the volume bound (1024 channels x 60 s of padded samples) cuts before the unit bound does
    length = np.full(3000, 120 * 48000, dtype=np.int32)
    group = np.zeros(3000, dtype=np.int32)
    order, begin, clen, cgrp = plan(group, length)
    assert np.all(np.diff(begin) <= 512)
    check(group, length, order, begin, clen, cgrp, 1024, 1024 * 2880000)
    # empty units (an ADX channel with no samples keeps a group of its own, capi_adx.hip) and an empty batch
    length = np.array([0, 5, 0, 48000, 0], dtype=np.int32)
    group = np.array([1, 0, 1, 0, 1], dtype=np.int32)
    order, begin, clen, cgrp = plan(group, length)
    check(group, length, order, begin, clen, cgrp, 1024, 1024 * 2880000)
    assert sorted(cgrp.tolist()) == [0, 0, 1]
    order, begin, clen, cgrp = plan([], [])
    assert len(order) == 0 and len(clen) == 0


def test_the_order_hook_wins_over_the_entry_points_choice():
    L = _lib.lib()
    length = log_uniform_lengths(2000, 6)
    group = np.zeros(len(length), dtype=np.int32)
    try:
        L.vga_testing_buckets_order_this_thread(1)
        assert np.all(np.diff(plan(group, length)[2]) >= 0)                        # asked for longest first, forced shortest first
        L.vga_testing_buckets_order_this_thread(2)
        assert np.all(np.diff(plan(group, length, shortest_first=True)[2]) <= 0)
    finally:
        L.vga_testing_buckets_order_this_thread(0)
