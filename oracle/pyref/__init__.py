"""oracle/pyref -- a SECOND, independently structured restatement of the three codecs, in
plain Python, written straight from the reference's C# (not from oracle/*.c).

TEST INFRASTRUCTURE ONLY.  Like the rest of oracle/, nothing here is imported by the product
(vgaudio_amd/) or measured by bench.py.  Its one job is to pin oracle/liboracle.so a second
way: the reference holds no result vectors for GC-ADPCM coefficients/bitstreams on general
audio, for any of CRI ADX, or for the CRI HCA encoder/decoder, and the reference itself (C#)
cannot be built in this image.  Two restatements that were written separately from the same
source and agree bit for bit on seeded and edge inputs (tests/test_pyref_crosscheck.py) are
the strongest pin available; the outputs they agree on are committed as golden fixtures
(tests/golden/make_codec_fixtures.py -> tests/golden/codec_vectors.json + .npz) and the GPU
tests compare the HIP path with those fixtures as well as with the C oracle.

Numeric model (RyuJIT x64): int = wrapping int32, `/` truncates toward zero, `>>` is
arithmetic, double = IEEE binary64 without FMA contraction (Python float), float = binary32
(numpy.float32), Math.Round = ties-to-even (Python round), (int)double = truncation with
0x80000000 for NaN / out-of-range values (cvttsd2si).

Modules: gcadpcm (GcAdpcmCoefficients.cs, GcAdpcmEncoder.cs, GcAdpcmDecoder.cs, GcAdpcmMath.cs),
criadx (CriAdxCodec.cs), crihca (CriHcaEncoder.cs, CriHcaDecoder.cs, CriHcaPacking.cs,
CriHcaFrame.cs, CriHcaTables.cs, HcaInfo.cs, Utilities/Mdct.cs, BitWriter.cs, BitReader.cs,
Crc16.cs), crypt (CriAdxKey.cs, CriAdxEncryption.cs, CriHcaKey.cs, CriHcaEncryption.cs, Helpers.GetPrimes; and
VGAudio.Tools/CrackAdx/GuessAdx.cs for one file), containers (Containers/Adx/AdxWriter.cs, Containers/Hca/HcaWriter.cs,
Utilities/Interleave.cs).  Paths are relative to /root/reference/src/VGAudio/.
"""
