#!/usr/bin/env python3
"""Harvests the GC-ADPCM loop-context test vectors of the reference's own tests into a data fixture
(tests/golden/gc_loop_context.json).  Run in the build container only (/root/reference is not
present on the GPU box):

    python tests/golden/make_gc_fixtures.py

Source (data, not code): the two literal arrays `Adpcm` (80 bytes) / `Pcm` (140 samples) and the
InlineData rows of Tests/Formats/GcAdpcm/GcAdpcmLoopContextTests.cs:20-43 (expected pred/scale byte
and history at five loop starts), and the alignment geometry rows of GcAdpcmAlignmentTests.cs:13-62.
"""
import json
import os
import re

REF = "/root/reference/src/VGAudio.Tests/Formats/GcAdpcm"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gc_loop_context.json")


def array_after(text, marker):
    body = text[text.index(marker):]
    body = body[body.index("{", body.index("=")) + 1:]
    body = body[:body.index("};")]
    return [int(t, 16) for t in re.findall(r"0x([0-9A-Fa-f]+)", body)]


def rows(text, method):
    head = text[:text.index("public void " + method)]
    block = head[head.rindex("[Theory]"):]
    out = []
    for args in re.findall(r"InlineData\(([^)]*)\)", block):
        out.append([int(a.strip(), 0) for a in args.split(",")])
    return out


def main():
    t = open(os.path.join(REF, "GcAdpcmLoopContextTests.cs"), encoding="utf-8-sig").read()
    a = open(os.path.join(REF, "GcAdpcmAlignmentTests.cs"), encoding="utf-8-sig").read()
    fx = {
        "source": "VGAudio.Tests/Formats/GcAdpcm/GcAdpcmLoopContextTests.cs, GcAdpcmAlignmentTests.cs",
        "adpcm": array_after(t, "byte[] Adpcm"),
        "pcm": array_after(t, "short[] Pcm"),
        "pred_scale": rows(t, "CreatingLoopContextPredScale"),      # [loopStart, expected]
        "history": rows(t, "CreatingLoopContextHistory"),           # [loopStart, hist1, hist2]
        "alignment_not_needed": rows(a, "AlignmentNotNeeded"),      # [multiple, loopStart, loopEnd, adpcmLength]
        "alignment_needed": rows(a, "AlignmentNeeded"),
        "aligned_loop_points": rows(a, "AlignedLoopPointsAreCorrect"),   # [..., expectedLoopStart, expectedLoopEnd]
        "aligned_sine": rows(a, "AlignedAdpcmIsCorrect"),           # [multiple, loopStart, sineCycles, tolerance]
    }
    assert len(fx["adpcm"]) == 80 and len(fx["pcm"]) == 140
    json.dump(fx, open(OUT, "w"), indent=1)
    print(OUT, {k: len(v) for k, v in fx.items() if isinstance(v, list)})


if __name__ == "__main__":
    main()
