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
int vga_hca_file_header(const vga_hca_info *h, const char *comment, float volume, int encryption_type, uint8_t *header_out)
{
    if (!h || !header_out) { set_error("null argument"); return VGA_ERR_ARGUMENT; }
    if (h->header_size < 8 || h->header_size > 0x7FFF) { set_error("HCA header size %d out of range", h->header_size); return VGA_ERR_OUT_OF_RANGE; }
    std::memset(header_out, 0, (size_t)h->header_size);
    HostCursor c{header_out, h->header_size - 2, 0};
    c.putn("HCA\0", 4);                                     // :84-89
    c.put16(0x0200);
    c.put16(h->header_size);
    c.putn("fmt\0", 4);                                     // :91-103
    c.put8(h->channel_count);
    c.put8(h->sample_rate >> 16);
    c.put16(h->sample_rate);
    c.put32(h->frame_count);
    c.put16(h->inserted_samples);
    c.put16(h->appended_samples);
    c.putn("comp", 4);                                      // :105-118
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
        c.putn("loop", 4);
        c.put32(h->loop_start_frame);
        c.put32(h->loop_end_frame);
        c.put16(h->pre_loop_samples);
        c.put16(h->post_loop_samples);
    }
    c.putn("ciph", 4);                                      // :131-135
    c.put16(encryption_type);
    if (volume != 1.0f) {                                   // :137-146
        uint32_t bits;
        std::memcpy(&bits, &volume, 4);
        c.putn("rva\0", 4);
        c.put32((int)bits);
    }
    bool blank = true;                                      // string.IsNullOrWhiteSpace (:66)
    if (comment)
        for (const char *s = comment; *s; s++)
            if (!(*s == ' ' || (*s >= 9 && *s <= 13))) blank = false;
    if (blank) {
        c.putn("pad", 3);                                   // :154-157: three bytes
    } else {
        c.putn("comm\0", 5);                                // :148-152
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
                         float volume, int encryption_type, uint8_t *d_files, int64_t file_pitch, void *stream)
{
    const int size = vga_hca_file_size(h);
    if (size < 0) return size;
    if (nstreams < 0 || !d_files || (!d_frames && h->frame_count > 0)) { set_error("null / negative argument"); return VGA_ERR_ARGUMENT; }
    const int64_t audio = (int64_t)h->frame_size * h->frame_count;
    if (file_pitch < size || frames_pitch < audio) { set_error("pitch smaller than the data"); return VGA_ERR_ARGUMENT; }
    if (nstreams == 0) return VGA_OK;
    std::vector<uint8_t> header((size_t)h->header_size);
    if (int rc = vga_hca_file_header(h, comment, volume, encryption_type, header.data())) return rc;
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
                  uint8_t *file_out)
{
    const int size = vga_hca_file_size(h);
    if (size < 0) return size;
    if (!file_out || (!frames && h->frame_count > 0)) { set_error("null argument"); return VGA_ERR_ARGUMENT; }
    if (int rc = vga_hca_file_header(h, comment, volume, encryption_type, file_out)) return rc;
    std::memcpy(file_out + h->header_size, frames, (size_t)h->frame_size * h->frame_count);
    return VGA_OK;
}

}  // extern "C"
