#!/usr/bin/env python3
"""Harvests the HCA / MDCT constant tables the reference's own tests pin, and writes them as a
data fixture (tests/golden/hca_tables.json).  Run in the build container only
(/root/reference is not present on the GPU box):

    python tests/golden/make_hca_fixtures.py

Sources (data, not code):
  * the deflate-packed table blob in Codecs/CriHca/CriHcaTables.cs:80-146 (Huffman-style
    quantised-spectrum tables, resolution curve, ATH curve, MDCT window, channel mappings),
    unpacked with our own reader of the ArrayUnpacker container format;
  * the golden literals of the reference's tests: Tests/Formats/CriHca/GeneratedTables.cs
    (exact f64), Tests/Formats/CriHca/UnpackedTables.cs, Tests/Utilities/PreBuiltMdctTables.cs.
Doubles are stored as 16-hex-digit IEEE bit patterns so the fixture is bit-exact.
"""
import json
import os
import re
import struct
import zlib

REF = "/root/reference/src"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hca_tables.json")


def hex64(x):
    return struct.pack(">d", float(x)).hex()


# ---------------------------------------------------------------- packed blob
def read_packed_blob():
    text = open(f"{REF}/VGAudio/Codecs/CriHca/CriHcaTables.cs", encoding="utf-8-sig").read()
    body = text[text.index("PackedTables ="):]
    body = body[body.index("{") + 1:body.index("};")]
    return bytes(int(tok, 16) for tok in re.findall(r"0x([0-9A-Fa-f]{2})", body))


TYPES = [("B", 1), ("b", 1), ("H", 2), ("h", 2), ("H", 2), ("i", 4), ("I", 4), ("q", 8), ("Q", 8), ("f", 4), ("d", 8)]


class Reader:
    def __init__(self, data):
        self.d, self.p = data, 0

    def u8(self):
        v = self.d[self.p]
        self.p += 1
        return v

    def u16(self):
        v = struct.unpack_from("<H", self.d, self.p)[0]
        self.p += 2
        return v

    def array(self, stored, length):
        if length == 0xFFFF:
            return None
        fmt, size = TYPES[stored]
        vals = list(struct.unpack_from("<" + fmt * length, self.d, self.p))
        self.p += size * length
        return vals


def unpack_array(r, rank):
    mode_type = r.u8()
    if mode_type == 0xFF:
        return None
    mode, stored = mode_type >> 4, mode_type & 0xF
    if mode == 0:
        length = r.u16()
        if rank == 1:
            return r.array(stored, length)
        return [unpack_array(r, rank - 1) for _ in range(length)]
    if mode == 1:
        dims = [r.u16() for _ in range(rank)]

        def rec(depth):
            if depth == len(dims) - 1:
                return r.array(stored, dims[depth])
            return [rec(depth + 1) for _ in range(dims[depth])]
        return rec(0)
    if mode == 2:
        length = r.u16()
        lengths = [r.u16() for _ in range(length)]
        return [r.array(stored, n) for n in lengths]
    raise ValueError("bad mode")


def unpack_tables(blob):
    if blob[0] == 1:
        size = struct.unpack_from("<i", blob, 1)[0]
        blob = zlib.decompressobj(-15).decompress(blob[5:])[:size]
    r = Reader(blob)
    assert r.u8() == 0 and r.u8() == 0
    count = r.u16()
    arrays = [None] * count
    for _ in range(count):
        ident, typ = r.u8(), r.u8()
        arrays[ident] = unpack_array(r, typ & 0xF)
    names = ["QuantizeSpectrumBits", "QuantizeSpectrumValue", "QuantizedSpectrumBits", "QuantizedSpectrumMaxBits",
             "QuantizedSpectrumValue", "ScaleToResolutionCurve", "AthCurve", "MdctWindow", "DefaultChannelMapping",
             "ValidChannelMappings"]
    return dict(zip(names, arrays))


# ---------------------------------------------------------------- C# literal tables of the tests
def parse_cs_tables(path):
    text = open(path, encoding="utf-8-sig").read()
    text = re.sub(r"//.*", "", text)
    out = {}
    for m in re.finditer(r"public static (\w+)((?:\[\])+) (\w+) \{ get; \} =", text):
        name = m.group(3)
        i = text.index("{", m.end())
        depth, j = 0, i
        while True:
            if text[j] == "{":
                depth += 1
            elif text[j] == "}":
                depth -= 1
                if depth == 0:
                    break
            j += 1
        lit = text[i:j + 1]
        float_suffixed = re.search(r"[0-9.]f\b", lit) is not None    # `double[] = { 1.5f, ... }`: f32 widened
        py = lit.replace("{", "[").replace("}", "]")
        py = re.sub(r"0[xX]([0-9A-Fa-f]+)", lambda h: str(int(h.group(1), 16)), py)   # hex before suffix stripping
        py = re.sub(r"(?<=[0-9.])[fFdD]\b", "", py)
        py = re.sub(r"new\s*\w*\s*\[\s*\]", "", py)
        py = re.sub(r",\s*\]", "]", py)
        py = re.sub(r"(?<![\w.])0+(?=\d)", "", py)      # C# allows leading zeros in decimal literals
        vals = eval(py)  # numeric literals only

        def widen(v):
            if isinstance(v, list):
                return [widen(x) for x in v]
            return struct.unpack("<f", struct.pack("<f", v))[0]
        if float_suffixed and m.group(1) == "double":
            vals = widen(vals)
        out[name] = (m.group(1), vals)
    return out


def main():
    packed = unpack_tables(read_packed_blob())
    gen = parse_cs_tables(f"{REF}/VGAudio.Tests/Formats/CriHca/GeneratedTables.cs")
    unp = parse_cs_tables(f"{REF}/VGAudio.Tests/Formats/CriHca/UnpackedTables.cs")
    mdct = parse_cs_tables(f"{REF}/VGAudio.Tests/Utilities/PreBuiltMdctTables.cs")

    def conv(typ, v):
        if isinstance(v, list):
            return [conv(typ, x) for x in v]
        return hex64(v) if typ == "double" else (struct.pack(">f", v).hex() if typ == "float" else int(v))

    fixture = {
        "_comment": "generated by tests/golden/make_hca_fixtures.py from the reference's packed blob and test "
                    "literals; doubles are big-endian IEEE-754 hex, floats likewise (8 hex digits)",
        "packed": {k: (conv("float", v) if k == "MdctWindow" else v) for k, v in packed.items()},
        "generated_tables_test": {k: conv(t, v) for k, (t, v) in gen.items()},
        "unpacked_tables_test": {k: conv(t, v) for k, (t, v) in unp.items()},
        "prebuilt_mdct_tables_test": {k: conv(t, v) for k, (t, v) in mdct.items()},
    }
    with open(OUT, "w") as f:
        json.dump(fixture, f, separators=(",", ":"))
    print("wrote", OUT, os.path.getsize(OUT), "bytes")
    for k, v in packed.items():
        print(" ", k, len(v))


if __name__ == "__main__":
    main()
