// CheckVectors.cs -- holds the golden fixtures of vgaudio_amd (tests/golden/) to the REAL VGAudio.
//
// The fixtures were written where two independent restatements of the C# sources agreed bit for bit (DESIGN.md section 2);
// no .NET toolchain exists in the image they were made in, so this program is shipped as source: a maintainer with
// `dotnet` runs it once and every array is compared with what VGAudio itself produces.
//
//   dotnet run -c Release -p:VGAudioSrc=<VGAudio>/src/VGAudio -- <this repo>/tests/golden
//
// Inputs: codec_vectors.bin + codec_vectors_index.json (tests/golden/export_flat.py: little-endian arrays, C order),
// codec_vectors.json (the manifest: parameters of every case), full_length_digests.json (SHA-256 of 60 s outputs; the
// input is regenerated here with the integer synthesiser of vgaudio_amd/synth.py).
// Entry points exercised: GcAdpcmCoefficients.CalculateCoefficients (Codecs/GcAdpcm/GcAdpcmCoefficients.cs:9),
// GcAdpcmEncoder.Encode (GcAdpcmEncoder.cs:14), GcAdpcmDecoder.Decode (GcAdpcmDecoder.cs:10), CriAdxCodec.Encode / Decode
// (Codecs/CriAdx/CriAdxCodec.cs:56, :9), CriHcaFormat.EncodeFromPcm16 -> CriHcaEncoder (Formats/CriHca/CriHcaFormat.cs:34,
// Codecs/CriHca/CriHcaEncoder.cs:49,126), CriHcaDecoder.Decode (CriHcaDecoder.cs:11).
using System;
using System.Collections.Generic;
using System.IO;
using System.Linq;
using System.Security.Cryptography;
using System.Text.Json;
using VGAudio.Codecs.CriAdx;
using VGAudio.Codecs.CriHca;
using VGAudio.Codecs.GcAdpcm;
using VGAudio.Formats.CriHca;
using VGAudio.Formats.Pcm16;

static class CheckVectors
{
    static byte[] Blob;
    static JsonElement Index;
    static int Failures, Checks;

    static JsonElement Entry(string name) => Index.GetProperty("arrays").GetProperty(name);
    static int[] Shape(string name) => Entry(name).GetProperty("shape").EnumerateArray().Select(e => e.GetInt32()).ToArray();

    static byte[] Bytes(string name)
    {
        JsonElement e = Entry(name);
        var a = new byte[e.GetProperty("bytes").GetInt32()];
        Buffer.BlockCopy(Blob, e.GetProperty("offset").GetInt32(), a, 0, a.Length);
        return a;
    }
    static short[] Shorts(string name)
    {
        byte[] raw = Bytes(name);
        var a = new short[raw.Length / 2];
        Buffer.BlockCopy(raw, 0, a, 0, raw.Length);          // the file is little-endian, like every platform .NET runs on
        return a;
    }
    static short[][] Rows(string name)                         // a 2-D int16 array as jagged rows
    {
        int[] s = Shape(name);
        short[] flat = Shorts(name);
        return Enumerable.Range(0, s[0]).Select(r => flat.Skip(r * s[1]).Take(s[1]).ToArray()).ToArray();
    }

    static void Check(string what, bool ok)
    {
        Checks++;
        if (!ok) Failures++;
        Console.WriteLine((ok ? "PASS  " : "FAIL  ") + what);
    }
    static bool Same(short[] a, short[] b) => a.Length == b.Length && a.AsSpan().SequenceEqual(b);
    static bool Same(byte[] a, byte[] b) => a.Length == b.Length && a.AsSpan().SequenceEqual(b);
    static string Sha(byte[] a) => Convert.ToHexString(SHA256.HashData(a)).ToLowerInvariant();
    static byte[] Raw(short[] a)
    {
        var r = new byte[a.Length * 2];
        Buffer.BlockCopy(a, 0, r, 0, r.Length);
        return r;
    }

    // ---- vgaudio_amd/synth.py, restated: integer-only, so every platform produces the same samples
    const ulong Seed = 0x5EED;
    static ulong SplitMix(ulong x)
    {
        x += 0x9E3779B97F4A7C15UL;
        ulong z = x;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9UL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBUL;
        return z ^ (z >> 31);
    }
    static long Tri(uint phase)
    {
        long q = phase >> 15;
        return q < 65536 ? q - 32768 : 98303 - q;
    }
    static short[] Synth(int channel, int n)
    {
        var inc = new uint[96];
        inc[0] = 4921183;
        for (int i = 1; i < 96; i++) inc[i] = (uint)(((ulong)inc[i - 1] * 69433) >> 16);
        ulong h = SplitMix((Seed << 32) ^ (ulong)channel);
        uint fInc = inc[channel % 96];
        uint phi = (uint)(h >> 32);
        long amp = 4000 + (long)((h & 0xFFFFFFFFUL) % 20001);
        uint lfo = (uint)(2000 + ((h >> 20) & 0x3FFF));
        ulong b = unchecked(((Seed << 32) ^ (ulong)channel) * 0x100000001B3UL);
        var x = new short[n];
        for (int i = 0; i < n; i++)
        {
            uint i32 = (uint)i;
            long s = ((Tri(unchecked(i32 * fInc)) * amp) >> 15) + ((Tri(unchecked(i32 * (uint)(3 * (ulong)fInc) + phi)) * (amp / 3)) >> 15);
            long nz = 0;
            for (int d = 0; d < 4; d++)
                nz += (long)(SplitMix(b ^ unchecked((ulong)i - (ulong)d)) & 4095) - 2048;
            s += nz >> 2;
            long env = 20480 + ((Tri(unchecked(i32 * lfo)) * 12287) >> 15);
            s = (s * env) >> 15;
            x[i] = (short)Math.Clamp(s, -32768, 32767);
        }
        return x;
    }

    static int Main(string[] args)
    {
        string dir = args.Length > 0 ? args[0] : Path.Combine("..", "..", "tests", "golden");
        Blob = File.ReadAllBytes(Path.Combine(dir, "codec_vectors.bin"));
        Index = JsonDocument.Parse(File.ReadAllText(Path.Combine(dir, "codec_vectors_index.json"))).RootElement;
        JsonElement m = JsonDocument.Parse(File.ReadAllText(Path.Combine(dir, "codec_vectors.json"))).RootElement;

        // ---------------------------------------------------------------- GC-ADPCM
        int gcN = m.GetProperty("gc").GetProperty("sample_count").GetInt32();
        foreach (JsonElement sig in m.GetProperty("gc").GetProperty("signals").EnumerateArray())
        {
            string name = sig.GetString();
            short[] pcm = Shorts($"gc_{name}_pcm");
            short[] coefs = GcAdpcmCoefficients.CalculateCoefficients(pcm);
            Check($"gc {name}: coefficients", Same(coefs, Shorts($"gc_{name}_coefs")));
            byte[] adpcm = GcAdpcmEncoder.Encode(pcm, coefs);
            Check($"gc {name}: bitstream", Same(adpcm, Bytes($"gc_{name}_adpcm")));
            short[] dec = GcAdpcmDecoder.Decode(adpcm, coefs, new GcAdpcmParameters { SampleCount = gcN });
            Check($"gc {name}: decoded", Same(dec, Shorts($"gc_{name}_decoded")));
        }
        Check("gc hostile coefficients (int32 wrap in the predictor): bitstream",
              Same(GcAdpcmEncoder.Encode(Shorts("gc_noise_fs_pcm"), Shorts("gc_hostile_coefs")), Bytes("gc_hostile_adpcm")));
        {
            JsonElement c0 = m.GetProperty("gc_config0");     // BASELINE configs[0]: 10 s of synth channel 0
            short[] pcm = Synth(0, c0.GetProperty("sample_count").GetInt32());
            Check("gc configs[0]: synthesised input", Sha(Raw(pcm)) == c0.GetProperty("input_sha256").GetString());
            short[] coefs = GcAdpcmCoefficients.CalculateCoefficients(pcm);
            Check("gc configs[0]: coefficients", coefs.Select(v => (int)v).SequenceEqual(c0.GetProperty("coefs").EnumerateArray().Select(e => e.GetInt32())));
            byte[] adpcm = GcAdpcmEncoder.Encode(pcm, coefs);
            Check("gc configs[0]: bitstream digest", Sha(adpcm) == c0.GetProperty("adpcm_sha256").GetString());
            Check("gc configs[0]: decoded digest",
                  Sha(Raw(GcAdpcmDecoder.Decode(adpcm, coefs, new GcAdpcmParameters { SampleCount = pcm.Length }))) == c0.GetProperty("decoded_sha256").GetString());
        }

        // ---------------------------------------------------------------- CRI ADX
        JsonElement adx = m.GetProperty("adx");
        int adxN = adx.GetProperty("sample_count").GetInt32();
        string[] adxSignals = adx.GetProperty("signals").EnumerateArray().Select(e => e.GetString()).ToArray();
        int k = 0;
        foreach (JsonElement c in adx.GetProperty("cases").EnumerateArray())
        {
            JsonElement p = c.GetProperty("params");
            CriAdxParameters Make(bool withFilter)
            {
                var cfg = new CriAdxParameters();
                if (p.TryGetProperty("type", out JsonElement t)) cfg.Type = (CriAdxType)t.GetInt32();
                if (withFilter && p.TryGetProperty("filter", out JsonElement f)) cfg.Filter = f.GetInt32();
                if (p.TryGetProperty("version", out JsonElement v)) cfg.Version = v.GetInt32();
                if (p.TryGetProperty("frame_size", out JsonElement fs)) cfg.FrameSize = fs.GetInt32();
                if (p.TryGetProperty("padding", out JsonElement pad)) cfg.Padding = pad.GetInt32();
                if (p.TryGetProperty("sample_rate", out JsonElement sr)) cfg.SampleRate = sr.GetInt32();
                return cfg;
            }
            foreach (string name in adxSignals)
            {
                CriAdxParameters cfg = Make(true);
                byte[] bytes = CriAdxCodec.Encode(Shorts($"adx_{name}_pcm"), cfg);
                Check($"adx case {k} {name}: bytes", Same(bytes, Bytes($"adx_{k}_{name}_bytes")));
                Check($"adx case {k} {name}: history", cfg.History == c.GetProperty("history").GetProperty(name).GetInt32());
                // CriAdxFormat.ToPcm16 decodes with History 0 and no filter setting
                Check($"adx case {k} {name}: decoded", Same(CriAdxCodec.Decode(bytes, adxN, Make(false)), Shorts($"adx_{k}_{name}_decoded")));
            }
            k++;
        }

        // ---------------------------------------------------------------- CRI HCA
        foreach (JsonElement c in m.GetProperty("hca").GetProperty("cases").EnumerateArray())
        {
            string name = c.GetProperty("name").GetString();
            short[][] pcm = Rows($"hca_{name}_pcm");
            bool looping = c.TryGetProperty("looping", out JsonElement lp) && lp.GetBoolean();
            var builder = new Pcm16FormatBuilder(pcm, 48000);
            if (looping) builder = builder.WithLoop(true, c.GetProperty("loop_start").GetInt32(), c.GetProperty("loop_end").GetInt32());
            var cfg = new CriHcaParameters { Quality = Enum.Parse<CriHcaQuality>(c.GetProperty("quality").GetString()) };
            CriHcaFormat fmt = new CriHcaFormat().EncodeFromPcm16(builder.Build(), cfg);
            int[] fs = Shape($"hca_{name}_frames");
            byte[] want = Bytes($"hca_{name}_frames");
            bool same = fmt.AudioData.Length == fs[0];
            for (int f = 0; same && f < fs[0]; f++) same = fmt.AudioData[f].AsSpan().SequenceEqual(want.AsSpan(f * fs[1], fs[1]));
            Check($"hca {name}: frames", same);
            short[][] dec = CriHcaDecoder.Decode(fmt.Hca, fmt.AudioData);
            short[][] wdec = Rows($"hca_{name}_decoded");
            Check($"hca {name}: decoded", dec.Length == wdec.Length && dec.Zip(wdec, Same).All(v => v));
        }

        // ---------------------------------------------------------------- 60 s of audio: digests
        string full = Path.Combine(dir, "full_length_digests.json");
        if (File.Exists(full))
        {
            JsonElement d = JsonDocument.Parse(File.ReadAllText(full)).RootElement;
            int n = d.GetProperty("sample_count").GetInt32();
            short[] x0 = Synth(0, n), x1 = Synth(1, n);
            JsonElement g = d.GetProperty("gc");
            Check("full length: synthesised input", Sha(Raw(x0)) == g.GetProperty("input_sha256").GetString());
            short[] coefs = GcAdpcmCoefficients.CalculateCoefficients(x0);
            Check("full length gc: coefficients", coefs.Select(v => (int)v).SequenceEqual(g.GetProperty("coefs").EnumerateArray().Select(e => e.GetInt32())));
            byte[] adpcm = GcAdpcmEncoder.Encode(x0, coefs);
            Check("full length gc: bitstream digest", Sha(adpcm) == g.GetProperty("adpcm_sha256").GetString());
            Check("full length gc: decoded digest",
                  Sha(Raw(GcAdpcmDecoder.Decode(adpcm, coefs, new GcAdpcmParameters { SampleCount = n }))) == g.GetProperty("decoded_sha256").GetString());
            JsonElement a = d.GetProperty("adx");
            var acfg = new CriAdxParameters();
            byte[] ab = CriAdxCodec.Encode(x0, acfg);
            Check("full length adx: bytes digest", Sha(ab) == a.GetProperty("bytes_sha256").GetString() && acfg.History == a.GetProperty("history").GetInt32());
            Check("full length adx: decoded digest", Sha(Raw(CriAdxCodec.Decode(ab, n, new CriAdxParameters()))) == a.GetProperty("decoded_sha256").GetString());
            JsonElement hc = d.GetProperty("hca");
            CriHcaFormat hf = new CriHcaFormat().EncodeFromPcm16(new Pcm16Format(new[] { x0, x1 }, 48000), new CriHcaParameters());
            Check("full length hca: frames digest", Sha(hf.AudioData.SelectMany(f => f).ToArray()) == hc.GetProperty("frames_sha256").GetString());
            short[][] hd = CriHcaDecoder.Decode(hf.Hca, hf.AudioData);
            Check("full length hca: decoded digest", Sha(hd.SelectMany(Raw).ToArray()) == hc.GetProperty("decoded_sha256").GetString());
        }

        Console.WriteLine($"{Checks - Failures} of {Checks} checks passed");
        return Failures == 0 ? 0 : 1;
    }
}
