#!/usr/bin/env python3
"""Writes tests/golden/codec_vectors.bin + codec_vectors_index.json: the arrays of codec_vectors.npz as one flat
little-endian file with a plain JSON index (name -> dtype, shape, byte offset), so that a program without an .npz reader
-- tools/dotnet_check/CheckVectors.cs, which runs the fixtures through the real VGAudio -- can load them.  Pure
re-packaging: no codec runs here; tests/test_oracle_fixtures_flat.py holds the two files to the .npz.

    python tests/golden/export_flat.py
"""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    z = np.load(os.path.join(HERE, "codec_vectors.npz"))
    index, blob, at = {}, bytearray(), 0
    for name in sorted(z.files):
        a = np.ascontiguousarray(z[name])
        a = a.astype(a.dtype.newbyteorder("<"), copy=False)
        raw = a.tobytes()
        index[name] = {"dtype": a.dtype.name, "shape": list(a.shape), "offset": at, "bytes": len(raw)}
        blob += raw
        pad = (-len(blob)) % 8
        blob += bytes(pad)
        at = len(blob)
    with open(os.path.join(HERE, "codec_vectors.bin"), "wb") as f:
        f.write(blob)
    with open(os.path.join(HERE, "codec_vectors_index.json"), "w") as f:
        json.dump({"_comment": "written by tests/golden/export_flat.py from codec_vectors.npz: little-endian arrays, C order, "
                               "each starting on an 8-byte boundary of codec_vectors.bin", "arrays": index}, f, indent=1, sort_keys=True)
    print("wrote %d arrays, %d bytes" % (len(index), len(blob)))


if __name__ == "__main__":
    main()
