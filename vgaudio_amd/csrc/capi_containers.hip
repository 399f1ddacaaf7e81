// capi_containers.hip -- C ABI of the ADX and HCA container writers (DSP lives next to the
// GC-ADPCM entry points).  Images are assembled in HBM; the host-pointer forms stage through the device.
#include "common.hpp"
#include "container_kernels.hpp"

#include <algorithm>
#include <cstring>

using namespace vga;

namespace {

int get_next_multiple(int value, int multiple)              // Utilities/Helpers.cs:71-80
{
    if (multiple <= 0) return value;
    if (value % multiple == 0) return value;
    return value + multiple - value % multiple;
}
int div_round_up(int v, int d) { return (v + d - 1) / d; }

// Utilities/Crc16.cs:12-18 with polynomial 0x8005 (HcaWriter.cs:18), MSB first, initial value 0
uint16_t crc16(const uint8_t *data, int size)
{
    uint16_t crc = 0;
    for (int i = 0; i < size; i++) {
        crc ^= (uint16_t)(data[i] << 8);
        for (int k = 0; k < 8; k++) crc = (crc & 0x8000) ? (uint16_t)((crc << 1) ^ 0x8005) : (uint16_t)(crc << 1);
    }
    return crc;
}

struct HostCursor {
    uint8_t *buf;
    int size, pos;
    bool overflow = false;
    void put8(int v) { if (pos < size) buf[pos] = (uint8_t)v; else overflow = true; pos++; }
    void put16(int v) { put8(v >> 8); put8(v); }
    void put32(int v) { put16(v >> 16); put16(v); }
    void putn(const void *p, int n) { for (int i = 0; i < n; i++) put8(((const uint8_t *)p)[i]); }
};

int adx_args(const vga_adx_file_params *p, int nch, int audio_len, vga_adx_file_layout *L, container::AdxHeaderArgs *a)
{
    if (int rc = vga_adx_file_layout_for(p, nch, L)) return rc;
    a->header_size = L->header_size; a->type = p->type; a->frame_size = p->frame_size; a->nch = nch;
    a->sample_rate = p->sample_rate; a->sample_count = L->sample_count; a->highpass_frequency = p->highpass_frequency;
    a->version = p->version; a->encryption_type = p->encryption_type; a->alignment_samples = p->alignment_samples;
    a->looping = p->looping ? 1 : 0; a->loop_start = p->loop_start; a->loop_start_offset = L->loop_start_offset;
    a->loop_end = p->loop_end; a->loop_end_offset = L->loop_end_offset;
    a->footer_size = L->footer_size; a->file_size = L->file_size;
    // the footer goes where the interleaver leaves the stream (Interleave.cs:58-77): after the blocks it copied,
    // which is short of FooterOffset when the (trimmed) frame count exceeds the frames the channels hold
    const int in_blocks = div_round_up(audio_len, p->frame_size);
    a->footer_pos = L->audio_offset + std::min(in_blocks, L->frame_count) * p->frame_size * nch;
    // everything the header writes must fit the image (MemoryStream over byte[FileSize] is not expandable)
    const int header_end = 20 + (p->version == 4 ? 4 + 4 * nch + (nch == 1 ? 4 : 0) : 0) + 24;
    if (header_end > L->file_size || L->header_size + 4 > L->file_size || L->header_size < 2) {
        set_error("ADX header (%d bytes written) does not fit the %d-byte file", header_end, L->file_size);
        return VGA_ERR_INVALID_OP;                          // NotSupportedException
    }
    return VGA_OK;
}

}  // namespace

extern "C" {

// ---------------------------------------------------------------- ADX (Containers/Adx/AdxWriter.cs:14-139)
int vga_adx_file_layout_for(const vga_adx_file_params *p, int nch, vga_adx_file_layout *L)
{
    if (!p || !L) { set_error("null argument"); return VGA_ERR_ARGUMENT; }
    if (nch < 1 || nch > 255) { set_error("ADX channel count %d does not fit the header byte", nch); return VGA_ERR_ARGUMENT; }
    if (p->frame_size < 3 || p->frame_size > 255) { set_error("ADX frame size %d out of range", p->frame_size); return VGA_ERR_ARGUMENT; }
    if (p->sample_count < 0 || p->loop_start < 0 || p->loop_end < 0) { set_error("negative sample count / loop point"); return VGA_ERR_OUT_OF_RANGE; }
    std::memset(L, 0, sizeof *L);
    const int spf = (p->frame_size - 2) * 2;                                                            // :27
    L->sample_count = (p->trim_file && p->looping) ? p->loop_end + spf * 3 : p->sample_count;           // :21
    L->frame_count = div_round_up(L->sample_count, spf);                                                // :28
    L->base_header_size = p->looping ? (p->version == 4 ? 60 : 52) : (p->version == 4 ? 36 : 32);        // :30
    if (p->looping) {                                                                                   // :58-69
        const int start = vga_adx_sample_count_to_byte_count(p->loop_start, p->frame_size) * nch + L->base_header_size + 4;
        L->alignment_bytes = get_next_multiple(start, 0x800) - start;
        if (p->version == 3) L->alignment_bytes += p->alignment_samples / spf * 0x800;
    }
    L->header_size = L->base_header_size + L->alignment_bytes;
    L->audio_offset = L->header_size + 4;
    const int64_t audio_size = (int64_t)p->frame_size * L->frame_count * nch;
    if (audio_size + L->audio_offset + 0x1000 > 0x7FFFFFFF) { set_error("ADX file would exceed 2 GiB (FileSize is an int)"); return VGA_ERR_OUT_OF_RANGE; }
    L->audio_size = (int)audio_size;
    L->footer_offset = L->audio_offset + L->audio_size;
    L->footer_size = p->looping ? get_next_multiple(L->footer_offset + p->frame_size, 0x800) - L->footer_offset : p->frame_size;
    L->loop_start_offset = L->audio_offset + vga_adx_sample_count_to_byte_count(p->loop_start, p->frame_size) * nch;
    L->loop_end_offset = L->audio_offset + get_next_multiple(vga_adx_sample_count_to_byte_count(p->loop_end, p->frame_size), p->frame_size) * nch;
    L->file_size = L->audio_offset + L->audio_size + L->footer_size;                                    // :18
    return VGA_OK;
}

int vga_adx_write_device(const uint8_t *d_audio, int64_t audio_pitch, int audio_len, const int16_t *d_history, int nch,
                         const vga_adx_file_params *p, uint8_t *d_file, void *stream)
{
    vga_adx_file_layout L;
    container::AdxHeaderArgs a;
    if (audio_len < 0 || !d_file || (audio_len > 0 && !d_audio)) { set_error("null / negative argument"); return VGA_ERR_ARGUMENT; }
    if (int rc = adx_args(p, nch, audio_len, &L, &a)) return rc;
    if (p->version == 4 && !d_history) { set_error("version 4 headers carry the channel histories"); return VGA_ERR_ARGUMENT; }
    if (audio_len > 0 && audio_pitch < audio_len) { set_error("audio pitch %lld < length %d", (long long)audio_pitch, audio_len); return VGA_ERR_ARGUMENT; }
    hipStream_t s = (hipStream_t)stream;
    VGA_HIP_TRY(hipMemsetAsync(d_file, 0, (size_t)L.file_size, s));
    if (int rc = container::launch_adx_header(a, d_history, d_file, s)) return rc;
    // WriteData (:119-131): frames of all channels in turn, FrameCount frames each
    if (int rc = container::launch_interleave(d_audio, audio_pitch, audio_len, nch, p->frame_size, L.frame_count * p->frame_size,
                                              d_file + L.audio_offset, s))
        return rc;
    return container::launch_adx_footer(a, d_file, s);
}

int vga_adx_write(const uint8_t *const *audio, int audio_len, const int16_t *history, int nch, const vga_adx_file_params *p,
                  uint8_t *file_out)
{
    vga_adx_file_layout L;
    container::AdxHeaderArgs a;
    if (audio_len < 0 || !file_out || !audio) { set_error("null / negative argument"); return VGA_ERR_ARGUMENT; }
    if (int rc = adx_args(p, nch, audio_len, &L, &a)) return rc;
    for (int c = 0; c < nch; c++)
        if (audio_len > 0 && !audio[c]) { set_error("audio[%d] is null", c); return VGA_ERR_ARGUMENT; }
    if (p->version == 4 && !history) { set_error("version 4 headers carry the channel histories"); return VGA_ERR_ARGUMENT; }
    if (int rc = require_device()) return rc;
    Stream st;
    VGA_HIP_TRY(st.create());
    DevBuf d_audio, d_hist, d_file;
    const int64_t pitch = round_up(audio_len > 0 ? audio_len : 1, 16);
    VGA_HIP_TRY(d_audio.alloc((size_t)nch * pitch));
    VGA_HIP_TRY(d_hist.alloc((size_t)nch * 2));
    VGA_HIP_TRY(d_file.alloc((size_t)L.file_size));
    for (int c = 0; c < nch && audio_len > 0; c++)
        VGA_HIP_TRY(hipMemcpyAsync(d_audio.as<uint8_t>() + c * pitch, audio[c], (size_t)audio_len, hipMemcpyHostToDevice, st.s));
    if (history) VGA_HIP_TRY(hipMemcpyAsync(d_hist.p, history, (size_t)nch * 2, hipMemcpyHostToDevice, st.s));
    if (int rc = vga_adx_write_device(d_audio.as<uint8_t>(), pitch, audio_len, history ? d_hist.as<int16_t>() : nullptr, nch, p,
                                      d_file.as<uint8_t>(), st.s))
        return rc;
    VGA_HIP_TRY(hipMemcpyAsync(file_out, d_file.p, (size_t)L.file_size, hipMemcpyDeviceToHost, st.s));
    VGA_HIP_TRY(hipStreamSynchronize(st.s));
    return VGA_OK;
}

// ---------------------------------------------------------------- HCA (Containers/Hca/HcaWriter.cs:12-185)
int vga_hca_file_size(const vga_hca_info *h)
{
    if (!h) { set_error("null HcaInfo"); return VGA_ERR_ARGUMENT; }
    const int64_t n = (int64_t)h->header_size + (int64_t)h->frame_size * h->frame_count;                // :22
    if (h->header_size < 0 || h->frame_size < 0 || h->frame_count < 0 || n > 0x7FFFFFFF) {
        set_error("HCA file size out of range");
        return VGA_ERR_OUT_OF_RANGE;
    }
    return (int)n;
}

// WriteHeader (:57-82): the chunks, zero padding, CRC-16 of everything before it.  header_out: h->header_size bytes.
int vga_hca_file_header(const vga_hca_info *h, const char *comment, float volume, int encryption_type, int encrypted_ids,
                        uint8_t *header_out)
{
    if (!h || !header_out) { set_error("null argument"); return VGA_ERR_ARGUMENT; }
    if (h->header_size < 8 || h->header_size > 0x7FFF) { set_error("HCA header size %d out of range", h->header_size); return VGA_ERR_OUT_OF_RANGE; }
    std::memset(header_out, 0, (size_t)h->header_size);
    HostCursor c{header_out, h->header_size - 2, 0};
    // WriteChunkId (:158-171): with an encryption key every non-zero id byte gets its top bit set
    auto chunk = [&](const char *id, int n) { for (int i = 0; i < n; i++) c.put8(id[i] && encrypted_ids ? (id[i] | 0x80) : id[i]); };
    chunk("HCA\0", 4);                                      // :84-89
    c.put16(0x0200);
    c.put16(h->header_size);
    chunk("fmt\0", 4);                                     // :91-103
    c.put8(h->channel_count);
    c.put8(h->sample_rate >> 16);
    c.put16(h->sample_rate);
    c.put32(h->frame_count);
    c.put16(h->inserted_samples);
    c.put16(h->appended_samples);
    chunk("comp", 4);                                      // :105-118
    c.put16(h->frame_size);
    c.put8(h->min_resolution);
    c.put8(h->max_resolution);
    c.put8(h->track_count);
    c.put8(h->channel_config);
    c.put8(h->total_band_count);
    c.put8(h->base_band_count);
    c.put8(h->stereo_band_count);
    c.put8(h->bands_per_hfr_group);
    c.put16(0);
    if (h->looping) {                                       // :120-129
        chunk("loop", 4);
        c.put32(h->loop_start_frame);
        c.put32(h->loop_end_frame);
        c.put16(h->pre_loop_samples);
        c.put16(h->post_loop_samples);
    }
    chunk("ciph", 4);                                      // :131-135
    c.put16(encryption_type);
    if (volume != 1.0f) {                                   // :137-146
        uint32_t bits;
        std::memcpy(&bits, &volume, 4);
        chunk("rva\0", 4);
        c.put32((int)bits);
    }
    bool blank = true;                                      // string.IsNullOrWhiteSpace (:66)
    if (comment)
        for (const char *s = comment; *s; s++)
            if (!(*s == ' ' || (*s >= 9 && *s <= 13))) blank = false;
    if (blank) {
        chunk("pad", 3);                                   // :154-157: three bytes
    } else {
        chunk("comm\0", 5);                                // :148-152
        c.putn(comment, (int)std::strlen(comment) + 1);
    }
    if (c.overflow) {
        set_error("HCA header chunks (%d bytes) do not fit HeaderSize %d", c.pos, h->header_size);
        return VGA_ERR_INVALID_OP;
    }
    const uint16_t crc = crc16(header_out, h->header_size - 2);   // :75-79
    header_out[h->header_size - 2] = (uint8_t)(crc >> 8);
    header_out[h->header_size - 1] = (uint8_t)crc;
    return VGA_OK;
}

// nstreams equally shaped streams (one HcaInfo): image s = header + stream s's frames, file_pitch bytes apart
int vga_hca_write_device(const vga_hca_info *h, const uint8_t *d_frames, int64_t frames_pitch, int nstreams, const char *comment,
                         float volume, int encryption_type, int encrypted_ids, uint8_t *d_files, int64_t file_pitch, void *stream)
{
    const int size = vga_hca_file_size(h);
    if (size < 0) return size;
    if (nstreams < 0 || !d_files || (!d_frames && h->frame_count > 0)) { set_error("null / negative argument"); return VGA_ERR_ARGUMENT; }
    const int64_t audio = (int64_t)h->frame_size * h->frame_count;
    if (file_pitch < size || frames_pitch < audio) { set_error("pitch smaller than the data"); return VGA_ERR_ARGUMENT; }
    if (nstreams == 0) return VGA_OK;
    std::vector<uint8_t> header((size_t)h->header_size);
    if (int rc = vga_hca_file_header(h, comment, volume, encryption_type, encrypted_ids, header.data())) return rc;
    hipStream_t s = (hipStream_t)stream;
    // the header goes into image 0 straight from the host, the other images copy it on the device
    VGA_HIP_TRY(hipMemcpyAsync(d_files, header.data(), header.size(), hipMemcpyHostToDevice, s));
    VGA_HIP_TRY(hipStreamSynchronize(s));                   // `header` is pageable and dies with this frame
    if (nstreams > 1)
        if (int rc = container::launch_replicate(d_files, h->header_size, d_files + file_pitch, file_pitch, nstreams - 1, s)) return rc;
    if (audio > 0)                                          // WriteData (:173-179): the frames, back to back
        VGA_HIP_TRY(hipMemcpy2DAsync(d_files + h->header_size, (size_t)file_pitch, d_frames, (size_t)frames_pitch, (size_t)audio,
                                     (size_t)nstreams, hipMemcpyDeviceToDevice, s));
    return VGA_OK;
}

// One stream held in host memory (CriHcaFormat.AudioData flattened): header + frames.  No device work is needed
// for a 96-byte header and a copy, so this form stays on the host.
int vga_hca_write(const vga_hca_info *h, const uint8_t *frames, const char *comment, float volume, int encryption_type,
                  int encrypted_ids, uint8_t *file_out)
{
    const int size = vga_hca_file_size(h);
    if (size < 0) return size;
    if (!file_out || (!frames && h->frame_count > 0)) { set_error("null argument"); return VGA_ERR_ARGUMENT; }
    if (int rc = vga_hca_file_header(h, comment, volume, encryption_type, encrypted_ids, file_out)) return rc;
    std::memcpy(file_out + h->header_size, frames, (size_t)h->frame_size * h->frame_count);
    return VGA_OK;
}

}  // extern "C"

// ---------------------------------------------------------------- WAVE, 16-bit PCM
namespace {

// The reference's little-endian BinaryReader over the file: reads past the end throw (EndOfStreamException),
// which the boundary reports as VGA_ERR_INVALID_DATA.
struct RiffReader {
    const uint8_t *p;
    int64_t len, pos = 0;
    bool eof = false;
    bool has(int64_t n) { if (pos + n > len) { eof = true; return false; } return true; }
    int u16() { if (!has(2)) return 0; const int v = p[pos] | (p[pos + 1] << 8); pos += 2; return v; }
    int i16() { return (int16_t)u16(); }
    int i32() { if (!has(4)) return 0; const uint32_t v = (uint32_t)p[pos] | ((uint32_t)p[pos + 1] << 8) | ((uint32_t)p[pos + 2] << 16) | ((uint32_t)p[pos + 3] << 24); pos += 4; return (int)v; }
    bool tag(char out[4]) { if (!has(4)) return false; std::memcpy(out, p + pos, 4); pos += 4; return true; }
    void skip_to(int64_t target) { if (target > pos) pos = std::min(target, len); }   // ReadBytes(remaining) stops at the end
};

const uint8_t kSubtypePcm[16] = {0x01, 0x00, 0x00, 0x00, 0x00, 0x00, 0x10, 0x00, 0x80, 0x00, 0x00, 0xAA, 0x00, 0x38, 0x9B, 0x71};

int invalid(const char *msg) { set_error("%s", msg); return VGA_ERR_INVALID_DATA; }

int wave_channel_mask(int n)                                // WaveWriter.cs:147-164
{
    switch (n) {
    case 4: return 0x0033;
    case 5: return 0x0133;
    case 6: return 0x0633;
    case 7: return 0x01f3;
    case 8: return 0x06f3;
    default: return (int)((1u << (n & 31)) - 1);
    }
}

struct LeWriter {
    uint8_t *c;
    void tag(const char *t) { std::memcpy(c, t, 4); c += 4; }
    void u16(int v) { c[0] = (uint8_t)v; c[1] = (uint8_t)(v >> 8); c += 2; }
    void u32(int v) { u16(v); u16(v >> 16); }
};

int wave_header_size(const vga_wave_params *p, int nch) { return 12 + 8 + (nch > 2 ? 40 : 16) + (p->looping ? 8 + 0x3c : 0) + 8; }

// WriteRiffHeader / WriteFmtChunk / WriteSmplChunk / the data chunk's header (WaveWriter.cs:56-129); every chunk
// size here is even, so the reference's 2-byte alignment steps never move the position
void wave_header(const vga_wave_params *p, int nch, int64_t file_size, uint8_t *out)
{
    LeWriter w{out};
    w.tag("RIFF");
    w.u32((int)(file_size - 8));
    w.tag("WAVE");
    w.tag("fmt ");
    w.u32(nch > 2 ? 40 : 16);
    w.u16(nch > 2 ? 0xFFFE : 1);
    w.u16(nch);
    w.u32(p->sample_rate);
    w.u32(p->sample_rate * 2 * nch);
    w.u16(2 * nch);
    w.u16(16);
    if (nch > 2) {
        w.u16(22);
        w.u16(16);
        w.u32(wave_channel_mask(nch));
        std::memcpy(w.c, kSubtypePcm, 16);
        w.c += 16;
    }
    if (p->looping) {
        w.tag("smpl");
        w.u32(0x3c);
        for (int i = 0; i < 7; i++) w.u32(0);
        w.u32(1);
        for (int i = 0; i < 3; i++) w.u32(0);
        w.u32(p->loop_start);
        w.u32(p->loop_end);
        w.u32(0);
        w.u32(0);
    }
    w.tag("data");
    w.u32(nch * p->sample_count * 2);
}

}  // namespace

extern "C" {

// WaveReader.ReadFile + ToAudioStream up to the audio itself (WaveReader.cs:13-68, RiffParser.cs:36-78): host only.
int vga_wave_parse(const uint8_t *file, int64_t file_len, vga_wave_info *w)
{
    if (!file || !w || file_len < 0) { set_error("null argument"); return VGA_ERR_ARGUMENT; }
    std::memset(w, 0, sizeof *w);
    RiffReader r{file, file_len};
    char id[4], type[4];
    if (!r.tag(id)) return invalid("file ends inside the RIFF header");
    const int64_t riff_size = r.i32();
    if (!r.tag(type) || r.eof) return invalid("file ends inside the RIFF header");
    if (std::memcmp(id, "RIFF", 4) != 0) return invalid("Not a valid RIFF file");                    // RiffChunk.cs:21-24
    const int64_t riff_end = 8 + riff_size;                                                            // RiffParser.cs:44-45
    bool have_fmt = false, have_data = false, have_smpl = false, have_ext = false, ext_pcm = false;
    int format_tag = 0, block_align = 0, smpl_loops = 0, smpl_start = 0, smpl_end = 0;
    while (r.pos + 8 < riff_end) {
        if (!r.tag(id)) return invalid("RIFF size runs past the end of the file");
        const int64_t size = r.i32();
        if (r.eof) return invalid("file ends inside a chunk header");
        if (size < 0) return invalid("negative chunk size");
        const int64_t body = r.pos;
        if (!std::memcmp(id, "fmt ", 4)) {                                                             // WaveFmtChunk.cs:16-34
            format_tag = r.u16();
            w->channel_count = r.i16();
            w->sample_rate = r.i32();
            r.i32();
            block_align = r.i16();
            w->bits_per_sample = r.i16();
            have_fmt = true;
            have_ext = false;
            if (format_tag == 0xFFFE) {                                                                // WaveFormatExtensible.cs:20-27
                const int64_t ext_body = r.pos + 2;
                const int ext_size = r.i16();
                r.i16();
                r.i32();
                if (r.has(16)) { ext_pcm = std::memcmp(file + r.pos, kSubtypePcm, 16) == 0; r.pos += 16; }
                have_ext = true;
                r.skip_to(ext_body + ext_size);
            }
        } else if (!std::memcmp(id, "data", 4)) {                                                      // WaveDataChunk.cs:9-15
            have_data = true;
            w->data_offset = r.pos;
            w->data_size_declared = (int)size;
            w->data_size = (int)std::min<int64_t>(size, file_len - r.pos);
            r.pos += w->data_size;
        } else if (!std::memcmp(id, "smpl", 4)) {                                                      // WaveSmplChunk.cs:19-45
            for (int i = 0; i < 7; i++) r.i32();
            smpl_loops = r.i32();
            r.i32();
            if (smpl_loops < 0) return invalid("negative loop count in the smpl chunk");
            for (int i = 0; i < smpl_loops && !r.eof; i++) {
                r.i32(); r.i32();
                const int start = r.i32(), end = r.i32();
                r.i32(); r.i32();
                if (i == 0) { smpl_start = start; smpl_end = end; }
            }
            have_smpl = true;
        } else if (!std::memcmp(id, "fact", 4)) {
            r.i32();
        }
        if (r.eof) return invalid("file ends inside a chunk");
        r.skip_to(body + size);
        r.pos = body + size + ((body + size) & 1);                                                     // chunks are 2-byte aligned
    }
    // ValidateWaveFile (WaveReader.cs:70-98)
    if (std::memcmp(type, "WAVE", 4) != 0) return invalid("Not a valid WAVE file");
    if (!have_fmt) return invalid("File must have a valid fmt chunk");
    if (!have_data) return invalid("File must have a valid data chunk");
    const int bytes_per_sample = (w->bits_per_sample + 7) / 8;
    if (format_tag != 1 && format_tag != 0xFFFE) return invalid("Must contain PCM data. Has unsupported format");
    if (w->bits_per_sample != 16 && w->bits_per_sample != 8) return invalid("Must have 8 or 16 bits per sample");
    if (w->channel_count == 0) return invalid("Channel count must not be zero");
    if (block_align != bytes_per_sample * w->channel_count) return invalid("File has invalid block alignment");
    if (have_ext && !ext_pcm) return invalid("Must contain PCM data. Has unsupported SubFormat");
    if (w->channel_count < 0) return invalid("negative channel count");
    w->sample_count_declared = w->data_size_declared / bytes_per_sample / w->channel_count;            // :27
    w->sample_count = w->data_size / bytes_per_sample / w->channel_count;                              // Interleave.cs:190
    if (have_smpl && smpl_loops > 0) {                                                                 // :32-37
        w->loop_start = smpl_start;
        w->loop_end = smpl_end;
        w->looping = smpl_end > smpl_start;
    }
    if (w->looping) {                                                                                  // AudioFormatBaseBuilder.cs:30-43
        if (w->loop_start < 0 || w->loop_start > w->sample_count || w->loop_end < 0 || w->loop_end > w->sample_count) {
            set_error("Loop points must be less than the number of samples and non-negative.");
            return VGA_ERR_OUT_OF_RANGE;
        }
    } else {
        w->loop_start = w->loop_end = 0;
    }
    return VGA_OK;
}

// InterleavedByteToShort (Interleave.cs:188-207) on the device: d_data = the data chunk's bytes (any alignment)
int vga_wave_deinterleave_pcm16_device(const uint8_t *d_data, int sample_count, int nch, int16_t *d_pcm, int64_t pcm_pitch,
                                       void *stream)
{
    if (sample_count < 0 || nch < 0) { set_error("negative size"); return VGA_ERR_ARGUMENT; }
    if (sample_count == 0 || nch == 0) return VGA_OK;
    if (!d_data || !d_pcm || pcm_pitch < sample_count) { set_error("null pointer / pitch < sample count"); return VGA_ERR_ARGUMENT; }
    return container::launch_pcm16_deinterleave(d_data, sample_count, nch, d_pcm, pcm_pitch, (hipStream_t)stream);
}

// WaveReader for a 16-bit file in host memory: pcm_out[c] = info->sample_count shorts
int vga_wave_read_pcm16(const uint8_t *file, int64_t file_len, const vga_wave_info *w, int16_t *const *pcm_out)
{
    if (!file || !w || !pcm_out) { set_error("null argument"); return VGA_ERR_ARGUMENT; }
    if (w->bits_per_sample != 16) { set_error("only 16-bit PCM is converted here (Pcm8 is outside this path)"); return VGA_ERR_ARGUMENT; }
    const int64_t bytes = (int64_t)w->sample_count * w->channel_count * 2;
    if (w->data_offset < 0 || w->data_offset + bytes > file_len) { set_error("info does not describe this file"); return VGA_ERR_ARGUMENT; }
    if (bytes == 0) return VGA_OK;
    if (int rc = require_device()) return rc;
    Stream st;
    VGA_HIP_TRY(st.create());
    DevBuf d_in, d_out;
    const int64_t pitch = round_up(w->sample_count, 8);
    VGA_HIP_TRY(d_in.alloc((size_t)bytes));
    VGA_HIP_TRY(d_out.alloc((size_t)w->channel_count * pitch * 2));
    VGA_HIP_TRY(hipMemcpyAsync(d_in.p, file + w->data_offset, (size_t)bytes, hipMemcpyHostToDevice, st.s));
    if (int rc = container::launch_pcm16_deinterleave(d_in.as<uint8_t>(), w->sample_count, w->channel_count, d_out.as<int16_t>(), pitch, st.s))
        return rc;
    for (int c = 0; c < w->channel_count; c++) {
        if (!pcm_out[c]) { set_error("pcm_out[%d] is null", c); return VGA_ERR_ARGUMENT; }
        VGA_HIP_TRY(hipMemcpyAsync(pcm_out[c], d_out.as<int16_t>() + c * pitch, (size_t)w->sample_count * 2, hipMemcpyDeviceToHost, st.s));
    }
    VGA_HIP_TRY(hipStreamSynchronize(st.s));
    return VGA_OK;
}

// WaveWriter.FileSize (WaveWriter.cs:25-30), 16-bit
int64_t vga_wave_file_size(const vga_wave_params *p, int nch)
{
    if (!p || nch < 1 || nch > 0x7FFF || p->sample_count < 0) { set_error("bad WAVE parameters"); return VGA_ERR_ARGUMENT; }
    const int64_t size = wave_header_size(p, nch) + (int64_t)nch * p->sample_count * 2;
    if (size > 0x7FFFFFFF) { set_error("WAVE file would exceed 2 GiB (FileSize is an int)"); return VGA_ERR_OUT_OF_RANGE; }
    return size;
}

int vga_wave_write_pcm16_device(const int16_t *d_pcm, int64_t pcm_pitch, int nch, const vga_wave_params *p, uint8_t *d_file,
                                void *stream)
{
    const int64_t size = vga_wave_file_size(p, nch);
    if (size < 0) return (int)size;
    if (!d_file || (p->sample_count > 0 && (!d_pcm || pcm_pitch < p->sample_count))) { set_error("null pointer / pitch < sample count"); return VGA_ERR_ARGUMENT; }
    uint8_t header[160];
    const int hs = wave_header_size(p, nch);
    wave_header(p, nch, size, header);
    hipStream_t s = (hipStream_t)stream;
    VGA_HIP_TRY(hipMemcpyAsync(d_file, header, (size_t)hs, hipMemcpyHostToDevice, s));
    VGA_HIP_TRY(hipStreamSynchronize(s));                   // `header` lives on this stack frame
    return container::launch_pcm16_interleave(d_pcm, pcm_pitch, p->sample_count, nch, d_file + hs, s);
}

int vga_wave_write_pcm16(const int16_t *const *pcm, int nch, const vga_wave_params *p, uint8_t *file_out)
{
    const int64_t size = vga_wave_file_size(p, nch);
    if (size < 0) return (int)size;
    if (!pcm || !file_out) { set_error("null argument"); return VGA_ERR_ARGUMENT; }
    const int hs = wave_header_size(p, nch);
    if (p->sample_count == 0) { wave_header(p, nch, size, file_out); return VGA_OK; }
    if (int rc = require_device()) return rc;
    Stream st;
    VGA_HIP_TRY(st.create());
    DevBuf d_in, d_file;
    const int64_t pitch = round_up(p->sample_count, 8);
    VGA_HIP_TRY(d_in.alloc((size_t)nch * pitch * 2));
    VGA_HIP_TRY(d_file.alloc((size_t)size));
    for (int c = 0; c < nch; c++) {
        if (!pcm[c]) { set_error("pcm[%d] is null", c); return VGA_ERR_ARGUMENT; }
        VGA_HIP_TRY(hipMemcpyAsync(d_in.as<int16_t>() + c * pitch, pcm[c], (size_t)p->sample_count * 2, hipMemcpyHostToDevice, st.s));
    }
    if (int rc = vga_wave_write_pcm16_device(d_in.as<int16_t>(), pitch, nch, p, d_file.as<uint8_t>(), st.s)) return rc;
    VGA_HIP_TRY(hipMemcpyAsync(file_out, d_file.p, (size_t)size, hipMemcpyDeviceToHost, st.s));
    VGA_HIP_TRY(hipStreamSynchronize(st.s));
    (void)hs;
    return VGA_OK;
}

}  // extern "C"
