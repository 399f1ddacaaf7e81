"""The GC-ADPCM encoder's piece schedule (gc::plan_encode_pieces_on, vgaudio_amd/csrc/gc_encode_kernel.hip) is host
arithmetic: whatever the batch, the pieces must COVER the longest channel (nobody encodes what lies past the last piece) and
stay within what the scratch arrays and the ragged item list are sized for.  Needs no GPU.

Round 4's advisor found the case the first test pins: a ragged batch of few channel groups with one long channel made the
two-size schedule ask for 1507 pieces of 4096 frames, the clamp to 1024 cut the plan off at 4 194 304 frames, and the
`*_v` entry points returned VGA_OK with the tail of a 30-minute channel never encoded."""
import ctypes as C

import numpy as np
import pytest

from vgaudio_amd import _lib

MAX_PIECES = 1024
MIN_PIECE_FRAMES = 3584


def plan(cus, groups, frames, group_frames, ragged):
    out = (C.c_int * 5)()
    assert _lib.lib().vga_testing_gc_plan_pieces(cus, groups, frames, group_frames, int(ragged), out) == 0
    return dict(pieces=out[0], big=out[1], nb=out[2], small=out[3], persistent=out[4])


def first(p, k):
    return min(k, p["nb"]) * p["big"] + max(k - p["nb"], 0) * p["small"]


def check(p, frames):
    assert 1 <= p["pieces"] <= MAX_PIECES
    assert first(p, p["pieces"]) >= frames, (p, frames)            # the pieces cover the channel
    if p["pieces"] > 1:
        assert first(p, p["pieces"] - 1) < frames, (p, frames)     # ... and the last one is not empty
        assert min(p["big"], p["small"]) >= MIN_PIECE_FRAMES or p["pieces"] == MAX_PIECES or p["big"] == p["small"]


def test_a_long_channel_in_a_small_ragged_batch_is_covered():
    # a 30-minute 48 kHz channel next to a short one: one group of 16 slots, 256 compute units
    frames = (30 * 60 * 48000 + 13) // 14
    p = plan(256, 1, frames, frames, True)
    check(p, frames)
    assert p["pieces"] == MAX_PIECES and p["big"] == p["small"] == -(-frames // MAX_PIECES)


@pytest.mark.parametrize("ragged", [False, True])
def test_every_plan_covers_its_longest_channel(ragged):
    rng = np.random.default_rng(20260927)
    shapes = [(256, 256, 205715), (256, 1, 205715), (256, 1, 14), (256, 1, 1), (304, 19, 10 ** 7), (256, 626, 411429),
              (64, 3, 2 ** 27), (256, 2, 6171429), (256, 128, 4 * MIN_PIECE_FRAMES), (256, 128, 4 * MIN_PIECE_FRAMES - 1)]
    for _ in range(3000):
        cus = int(rng.choice([8, 64, 104, 228, 256, 304]))
        groups = int(rng.integers(1, 3000)) if rng.random() < 0.5 else int(rng.integers(1, 12))
        frames = int(np.exp(rng.uniform(0, np.log(2 ** 27))))       # up to 2^27 frames (~11 h at 48 kHz)
        shapes.append((cus, groups, max(frames, 1)))
    for cus, groups, frames in shapes:
        # ragged batches: the groups' longest channels hold anything between one channel's frames and groups x frames
        gf = frames * groups if not ragged else max(frames, int(frames * groups * rng.uniform(0.02, 1.0)))
        check(plan(cus, groups, frames, gf, ragged), frames)
