// capi_gcadpcm.hip -- C-ABI entry points for GC-ADPCM (see include/vgaudio_hip.h).
#include "common.hpp"
#include "gcadpcm_kernels.hpp"

#include <cmath>

namespace vga {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

int require_device()
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no HIP device available (%s); libvgaudio_hip has no CPU fallback",
                  e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return VGA_ERR_DEVICE;
    }
    return VGA_OK;
}

// ---- GcAdpcmMath.cs:11-47 (host) ----
static int divide_by2_round_up(int v) { return (v / 2) + (v & 1); }

}  // namespace vga

using namespace vga;

extern "C" {

const char *vga_last_error(void) { return g_err; }
const char *vga_version(void) { return "vgaudio_hip 0.1 (gfx950)"; }

int vga_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int vga_set_device(int device)
{
    VGA_HIP_TRY(hipSetDevice(device));
    return VGA_OK;
}

int vga_gcadpcm_nibble_count_to_sample_count(int nibble_count)
{
    int frames = nibble_count / 16;
    int extra_nibbles = nibble_count % 16;
    int extra_samples = extra_nibbles < 2 ? 0 : extra_nibbles - 2;
    return 14 * frames + extra_samples;
}
int vga_gcadpcm_sample_count_to_nibble_count(int sample_count)
{
    int frames = sample_count / 14;
    int extra_samples = sample_count % 14;
    int extra_nibbles = extra_samples == 0 ? 0 : extra_samples + 2;
    return 16 * frames + extra_nibbles;
}
int vga_gcadpcm_nibble_to_sample(int nibble)
{
    int frames = nibble / 16;
    int extra_nibbles = nibble % 16;
    return 14 * frames + extra_nibbles - 2;
}
int vga_gcadpcm_sample_to_nibble(int sample)
{
    int frames = sample / 14;
    int extra_samples = sample % 14;
    return 16 * frames + extra_samples + 2;
}
int vga_gcadpcm_sample_count_to_byte_count(int sample_count)
{
    return divide_by2_round_up(vga_gcadpcm_sample_count_to_nibble_count(sample_count));
}
int vga_gcadpcm_byte_count_to_sample_count(int byte_count)
{
    return vga_gcadpcm_nibble_count_to_sample_count(byte_count * 2);
}

// ---------------------------------------------------------------- device-resident
size_t vga_gcadpcm_coefs_workspace_bytes(int nch, int length)
{
    if (nch <= 0 || length < 0) return 0;
    const size_t frames = ((size_t)length + 13) / 14;
    return (size_t)nch * (frames ? frames : 1) * 16;
}

static int check_pcm_layout(const void *p, int64_t pitch, int n, const char *what)
{
    if (((uintptr_t)p & 3) != 0 || (pitch & 1) != 0 || pitch < n) {
        set_error("%s: base must be 4-byte aligned and pitch even and >= length (pitch=%lld, n=%d)", what,
                  (long long)pitch, n);
        return VGA_ERR_ARGUMENT;
    }
    return VGA_OK;
}
static int check_adpcm_layout(const void *p, int64_t pitch, int nbytes, const char *what)
{
    if (((uintptr_t)p & 7) != 0 || (pitch & 7) != 0 || pitch < nbytes) {
        set_error("%s: base must be 8-byte aligned and pitch a multiple of 8 and >= byte count (pitch=%lld, bytes=%d)",
                  what, (long long)pitch, nbytes);
        return VGA_ERR_ARGUMENT;
    }
    return VGA_OK;
}

int vga_gcadpcm_coefs_device(const int16_t *d_pcm, int64_t pcm_pitch, int nch, int length, int16_t *d_coefs,
                             void *d_workspace, size_t workspace_bytes, void *stream)
{
    if (nch < 0 || length < 0) { set_error("negative size"); return VGA_ERR_ARGUMENT; }
    if (nch == 0) return VGA_OK;
    if (int rc = check_pcm_layout(d_pcm, pcm_pitch, length, "vga_gcadpcm_coefs_device")) return rc;
    if (workspace_bytes < vga_gcadpcm_coefs_workspace_bytes(nch, length) || !d_workspace) {
        set_error("workspace too small: need %zu bytes", vga_gcadpcm_coefs_workspace_bytes(nch, length));
        return VGA_ERR_ARGUMENT;
    }
    return gc::launch_coefs(d_pcm, pcm_pitch, nch, length, d_coefs, d_workspace, (hipStream_t)stream);
}

int vga_gcadpcm_encode_device(const int16_t *d_pcm, int64_t pcm_pitch, int nch, int sample_count,
                              const int16_t *d_coefs, const int16_t *d_hist1, const int16_t *d_hist2,
                              uint8_t *d_adpcm, int64_t adpcm_pitch, void *stream)
{
    if (nch < 0 || sample_count < 0) { set_error("negative size"); return VGA_ERR_ARGUMENT; }
    if (nch == 0 || sample_count == 0) return VGA_OK;
    if (int rc = check_pcm_layout(d_pcm, pcm_pitch, sample_count, "vga_gcadpcm_encode_device")) return rc;
    if (int rc = check_adpcm_layout(d_adpcm, adpcm_pitch, vga_gcadpcm_sample_count_to_byte_count(sample_count),
                                    "vga_gcadpcm_encode_device"))
        return rc;
    return gc::launch_encode(d_pcm, pcm_pitch, nch, sample_count, d_coefs, d_hist1, d_hist2, d_adpcm, adpcm_pitch,
                             (hipStream_t)stream);
}

int vga_gcadpcm_decode_device(const uint8_t *d_adpcm, int64_t adpcm_pitch, const int16_t *d_coefs, int nch,
                              int sample_count, const int16_t *d_hist1, const int16_t *d_hist2, int16_t *d_pcm,
                              int64_t pcm_pitch, int *d_status, void *stream)
{
    if (nch < 0 || sample_count < 0) { set_error("negative size"); return VGA_ERR_ARGUMENT; }
    if (nch == 0 || sample_count == 0) return VGA_OK;
    if (int rc = check_pcm_layout(d_pcm, pcm_pitch, sample_count, "vga_gcadpcm_decode_device")) return rc;
    if (int rc = check_adpcm_layout(d_adpcm, adpcm_pitch, vga_gcadpcm_sample_count_to_byte_count(sample_count),
                                    "vga_gcadpcm_decode_device"))
        return rc;
    return gc::launch_decode(d_adpcm, adpcm_pitch, d_coefs, nch, sample_count, d_hist1, d_hist2, d_pcm, pcm_pitch,
                             d_status, (hipStream_t)stream);
}

int vga_synth_pcm16_device(int16_t *d_pcm, int64_t pcm_pitch, int nch, int length, int first_channel,
                           const uint32_t *d_params, void *stream)
{
    if (nch < 0 || length < 0 || pcm_pitch < length) { set_error("bad synth arguments"); return VGA_ERR_ARGUMENT; }
    return gc::launch_synth(d_pcm, pcm_pitch, nch, length, first_channel, d_params, (hipStream_t)stream);
}

// ---------------------------------------------------------------- host-buffer batch API
namespace {

struct GcBatch {
    Stream st;
    DevBuf pcm, coefs, adpcm, h1, h2, ws, status;
    int64_t pcm_pitch = 0, adpcm_pitch = 0;
};

int upload_pcm(GcBatch &b, const int16_t *const *pcm, int nch, int n)
{
    b.pcm_pitch = round_up(n > 0 ? n : 1, 8);
    VGA_HIP_TRY(b.pcm.alloc((size_t)nch * b.pcm_pitch * sizeof(int16_t)));
    for (int c = 0; c < nch; c++) {
        if (n > 0)
            VGA_HIP_TRY(hipMemcpyAsync(b.pcm.as<int16_t>() + (int64_t)c * b.pcm_pitch, pcm[c],
                                       (size_t)n * sizeof(int16_t), hipMemcpyHostToDevice, b.st.s));
    }
    return VGA_OK;
}

int upload_hist(GcBatch &b, int nch, const int16_t *h1, const int16_t *h2)
{
    if (h1) {
        VGA_HIP_TRY(b.h1.alloc((size_t)nch * 2));
        VGA_HIP_TRY(hipMemcpyAsync(b.h1.p, h1, (size_t)nch * 2, hipMemcpyHostToDevice, b.st.s));
    }
    if (h2) {
        VGA_HIP_TRY(b.h2.alloc((size_t)nch * 2));
        VGA_HIP_TRY(hipMemcpyAsync(b.h2.p, h2, (size_t)nch * 2, hipMemcpyHostToDevice, b.st.s));
    }
    return VGA_OK;
}

int check_ptrs(const void *const *pp, int nch, const char *what)
{
    if (nch < 0) { set_error("%s: negative channel count", what); return VGA_ERR_ARGUMENT; }
    if (nch > 0 && !pp) { set_error("%s: null channel array", what); return VGA_ERR_ARGUMENT; }
    for (int c = 0; c < nch; c++)
        if (!pp[c]) { set_error("%s: channel %d is null", what, c); return VGA_ERR_ARGUMENT; }
    return VGA_OK;
}

int download_adpcm(GcBatch &b, uint8_t *const *adpcm_out, int nch, int nbytes)
{
    for (int c = 0; c < nch; c++)
        if (nbytes > 0)
            VGA_HIP_TRY(hipMemcpyAsync(adpcm_out[c], b.adpcm.as<uint8_t>() + (int64_t)c * b.adpcm_pitch, (size_t)nbytes,
                                       hipMemcpyDeviceToHost, b.st.s));
    return VGA_OK;
}

}  // namespace

int vga_gcadpcm_calculate_coefficients_batch(const int16_t *const *pcm, int nch, int length, int16_t *coefs_out)
{
    if (length < 0) { set_error("negative length"); return VGA_ERR_ARGUMENT; }
    if (int rc = check_ptrs((const void *const *)pcm, length > 0 ? nch : 0, "pcm")) return rc;
    if (nch < 0 || (nch > 0 && !coefs_out)) { set_error("bad coefs_out/nch"); return VGA_ERR_ARGUMENT; }
    if (nch == 0) return VGA_OK;
    if (int rc = require_device()) return rc;
    GcBatch b;
    VGA_HIP_TRY(b.st.create());
    if (int rc = upload_pcm(b, pcm, nch, length)) return rc;
    VGA_HIP_TRY(b.coefs.alloc((size_t)nch * 32));
    VGA_HIP_TRY(b.ws.alloc(vga_gcadpcm_coefs_workspace_bytes(nch, length)));
    if (int rc = gc::launch_coefs(b.pcm.as<int16_t>(), b.pcm_pitch, nch, length, b.coefs.as<int16_t>(), b.ws.p, b.st.s))
        return rc;
    VGA_HIP_TRY(hipMemcpyAsync(coefs_out, b.coefs.p, (size_t)nch * 32, hipMemcpyDeviceToHost, b.st.s));
    VGA_HIP_TRY(hipStreamSynchronize(b.st.s));
    return VGA_OK;
}

int vga_gcadpcm_encode_with_coefs_batch(const int16_t *const *pcm, int nch, int pcm_length, int sample_count,
                                        const int16_t *coefs, const int16_t *hist1, const int16_t *hist2,
                                        uint8_t *const *adpcm_out)
{
    if (sample_count == -1) sample_count = pcm_length;
    if (pcm_length < 0 || sample_count < 0) { set_error("negative length"); return VGA_ERR_ARGUMENT; }
    if (sample_count > pcm_length) {
        set_error("SampleCount (%d) exceeds pcm length (%d)", sample_count, pcm_length);
        return VGA_ERR_ARGUMENT;
    }
    if (int rc = check_ptrs((const void *const *)pcm, sample_count > 0 ? nch : 0, "pcm")) return rc;
    if (int rc = check_ptrs((const void *const *)adpcm_out, sample_count > 0 ? nch : 0, "adpcm_out")) return rc;
    if (nch > 0 && !coefs) { set_error("null coefs"); return VGA_ERR_ARGUMENT; }
    if (nch <= 0 || sample_count == 0) return nch < 0 ? VGA_ERR_ARGUMENT : VGA_OK;
    if (int rc = require_device()) return rc;
    GcBatch b;
    VGA_HIP_TRY(b.st.create());
    if (int rc = upload_pcm(b, pcm, nch, sample_count)) return rc;
    if (int rc = upload_hist(b, nch, hist1, hist2)) return rc;
    VGA_HIP_TRY(b.coefs.alloc((size_t)nch * 32));
    VGA_HIP_TRY(hipMemcpyAsync(b.coefs.p, coefs, (size_t)nch * 32, hipMemcpyHostToDevice, b.st.s));
    const int nbytes = vga_gcadpcm_sample_count_to_byte_count(sample_count);
    b.adpcm_pitch = round_up(nbytes, 16);
    VGA_HIP_TRY(b.adpcm.alloc((size_t)nch * b.adpcm_pitch));
    if (int rc = gc::launch_encode(b.pcm.as<int16_t>(), b.pcm_pitch, nch, sample_count, b.coefs.as<int16_t>(),
                                   b.h1.as<int16_t>(), b.h2.as<int16_t>(), b.adpcm.as<uint8_t>(), b.adpcm_pitch, b.st.s))
        return rc;
    if (int rc = download_adpcm(b, adpcm_out, nch, nbytes)) return rc;
    VGA_HIP_TRY(hipStreamSynchronize(b.st.s));
    return VGA_OK;
}

int vga_gcadpcm_encode_batch(const int16_t *const *pcm, int nch, int sample_count, int16_t hist1, int16_t hist2,
                             int16_t *coefs_out, uint8_t *const *adpcm_out)
{
    if (sample_count < 0) { set_error("negative sample count"); return VGA_ERR_ARGUMENT; }
    if (int rc = check_ptrs((const void *const *)pcm, sample_count > 0 ? nch : 0, "pcm")) return rc;
    if (int rc = check_ptrs((const void *const *)adpcm_out, sample_count > 0 ? nch : 0, "adpcm_out")) return rc;
    if (nch < 0 || (nch > 0 && !coefs_out)) { set_error("bad coefs_out/nch"); return VGA_ERR_ARGUMENT; }
    if (nch == 0) return VGA_OK;
    if (int rc = require_device()) return rc;
    GcBatch b;
    VGA_HIP_TRY(b.st.create());
    if (int rc = upload_pcm(b, pcm, nch, sample_count)) return rc;
    std::vector<int16_t> h1v((size_t)nch, hist1), h2v((size_t)nch, hist2);
    const bool use_hist = hist1 != 0 || hist2 != 0;
    if (use_hist)
        if (int rc = upload_hist(b, nch, h1v.data(), h2v.data())) return rc;
    VGA_HIP_TRY(b.coefs.alloc((size_t)nch * 32));
    VGA_HIP_TRY(b.ws.alloc(vga_gcadpcm_coefs_workspace_bytes(nch, sample_count)));
    const int nbytes = vga_gcadpcm_sample_count_to_byte_count(sample_count);
    b.adpcm_pitch = round_up(nbytes > 0 ? nbytes : 1, 16);
    VGA_HIP_TRY(b.adpcm.alloc((size_t)nch * b.adpcm_pitch));
    // EncodeChannel (GcAdpcmFormat.cs:129-135): coefficients, then encode
    if (int rc = gc::launch_coefs(b.pcm.as<int16_t>(), b.pcm_pitch, nch, sample_count, b.coefs.as<int16_t>(), b.ws.p,
                                  b.st.s))
        return rc;
    if (int rc = gc::launch_encode(b.pcm.as<int16_t>(), b.pcm_pitch, nch, sample_count, b.coefs.as<int16_t>(),
                                   b.h1.as<int16_t>(), b.h2.as<int16_t>(), b.adpcm.as<uint8_t>(), b.adpcm_pitch, b.st.s))
        return rc;
    VGA_HIP_TRY(hipMemcpyAsync(coefs_out, b.coefs.p, (size_t)nch * 32, hipMemcpyDeviceToHost, b.st.s));
    if (int rc = download_adpcm(b, adpcm_out, nch, nbytes)) return rc;
    VGA_HIP_TRY(hipStreamSynchronize(b.st.s));
    return VGA_OK;
}

int vga_gcadpcm_decode_batch(const uint8_t *const *adpcm, const int16_t *coefs, int nch, int sample_count,
                             const int16_t *hist1, const int16_t *hist2, int16_t *const *pcm_out)
{
    if (sample_count < 0) { set_error("negative sample count"); return VGA_ERR_ARGUMENT; }
    if (int rc = check_ptrs((const void *const *)adpcm, sample_count > 0 ? nch : 0, "adpcm")) return rc;
    if (int rc = check_ptrs((const void *const *)pcm_out, sample_count > 0 ? nch : 0, "pcm_out")) return rc;
    if (nch > 0 && !coefs) { set_error("null coefs"); return VGA_ERR_ARGUMENT; }
    if (nch <= 0 || sample_count == 0) return nch < 0 ? VGA_ERR_ARGUMENT : VGA_OK;
    if (int rc = require_device()) return rc;
    GcBatch b;
    VGA_HIP_TRY(b.st.create());
    const int nbytes = vga_gcadpcm_sample_count_to_byte_count(sample_count);
    b.adpcm_pitch = round_up(nbytes, 16);
    b.pcm_pitch = round_up(sample_count, 8);
    VGA_HIP_TRY(b.adpcm.alloc((size_t)nch * b.adpcm_pitch));
    VGA_HIP_TRY(b.pcm.alloc((size_t)nch * b.pcm_pitch * 2));
    VGA_HIP_TRY(b.coefs.alloc((size_t)nch * 32));
    VGA_HIP_TRY(b.status.alloc(sizeof(int)));
    VGA_HIP_TRY(hipMemsetAsync(b.status.p, 0, sizeof(int), b.st.s));
    for (int c = 0; c < nch; c++)
        VGA_HIP_TRY(hipMemcpyAsync(b.adpcm.as<uint8_t>() + (int64_t)c * b.adpcm_pitch, adpcm[c], (size_t)nbytes,
                                   hipMemcpyHostToDevice, b.st.s));
    VGA_HIP_TRY(hipMemcpyAsync(b.coefs.p, coefs, (size_t)nch * 32, hipMemcpyHostToDevice, b.st.s));
    if (int rc = upload_hist(b, nch, hist1, hist2)) return rc;
    if (int rc = gc::launch_decode(b.adpcm.as<uint8_t>(), b.adpcm_pitch, b.coefs.as<int16_t>(), nch, sample_count,
                                   b.h1.as<int16_t>(), b.h2.as<int16_t>(), b.pcm.as<int16_t>(), b.pcm_pitch,
                                   b.status.as<int>(), b.st.s))
        return rc;
    int status = 0;
    VGA_HIP_TRY(hipMemcpyAsync(&status, b.status.p, sizeof(int), hipMemcpyDeviceToHost, b.st.s));
    for (int c = 0; c < nch; c++)
        VGA_HIP_TRY(hipMemcpyAsync(pcm_out[c], b.pcm.as<int16_t>() + (int64_t)c * b.pcm_pitch, (size_t)sample_count * 2,
                                   hipMemcpyDeviceToHost, b.st.s));
    VGA_HIP_TRY(hipStreamSynchronize(b.st.s));
    if (status != 0) {
        set_error("a frame header names predictor > 7 (the reference throws IndexOutOfRangeException)");
        return VGA_ERR_ARGUMENT;
    }
    return VGA_OK;
}

// ---------------------------------------------------------------- dsptool-compatible exports
// VGAudio.Tools/GcAdpcm/DspToolDll.cs:16-29,94-108.  void-returning like the DLLs:
// failures leave outputs untouched and are reported through vga_last_error().
void correlateCoefs(int16_t *src, uint32_t samples, int16_t *coefsOut)
{
    const int16_t *chans[1] = {src};
    (void)vga_gcadpcm_calculate_coefficients_batch(chans, 1, (int)samples, coefsOut);
}

void encode(int16_t *src, uint8_t *dst, ADPCMINFO *cxt, uint32_t samples)
{
    const int16_t *chans[1] = {src};
    uint8_t *outs[1] = {dst};
    int16_t coefs[16];
    if (vga_gcadpcm_encode_batch(chans, 1, (int)samples, 0, 0, coefs, outs) != VGA_OK) return;
    if (cxt) {
        memset(cxt, 0, sizeof *cxt);
        memcpy(cxt->coef, coefs, sizeof coefs);
        cxt->pred_scale = samples ? dst[0] : 0;
    }
}

void decode(uint8_t *src, int16_t *dst, ADPCMINFO *cxt, uint32_t samples)
{
    if (!cxt) return;
    const uint8_t *ins[1] = {src};
    int16_t *outs[1] = {dst};
    int16_t coefs[16];
    memcpy(coefs, cxt->coef, sizeof coefs);
    const int16_t h1 = cxt->yn1, h2 = cxt->yn2;
    (void)vga_gcadpcm_decode_batch(ins, coefs, 1, (int)samples, &h1, &h2, outs);
}

void encodeFrame(int16_t *src, uint8_t *dst, int16_t *coefs, uint8_t one)
{
    (void)one;
    // DspEncodeFrame (GcAdpcmEncoder.cs:48-94): src[0..1] history, src[2..15] in/out
    const int16_t *chans[1] = {src + 2};
    uint8_t *outs[1] = {dst};
    const int16_t h2 = src[0], h1 = src[1];
    if (vga_gcadpcm_encode_with_coefs_batch(chans, 1, 14, 14, coefs, &h1, &h2, outs) != VGA_OK) return;
    // the encoder's reconstruction equals the decoder's output (:156-160 vs GcAdpcmDecoder.cs:40-44)
    const uint8_t *ins[1] = {dst};
    int16_t *rec[1] = {src + 2};
    (void)vga_gcadpcm_decode_batch(ins, coefs, 1, 14, &h1, &h2, rec);
}

}  // extern "C"
