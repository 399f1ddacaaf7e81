// capi_gcadpcm.hip -- C-ABI entry points for GC-ADPCM (see include/vgaudio_hip.h).
#include "common.hpp"
#include "host_batch.hpp"
#include "../../include/vgaudio_hip_testing.h"

#include <algorithm>
#include <mutex>
#include <vector>
#include "gcadpcm_kernels.hpp"

#include <cmath>
#include <dirent.h>
#include <unistd.h>

namespace vga {

static thread_local char g_err[512] = "";
static thread_local bool g_err_pending = false;

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    g_err_pending = true;
}

bool take_error_pending()
{
    const bool was = g_err_pending;
    g_err_pending = false;
    return was;
}

// test hook (include/vgaudio_hip_testing.h): per calling thread, so that no call on another thread is affected
static thread_local int g_force_open_seams = 0;
int force_open_seams() { return g_force_open_seams; }

static thread_local int g_encoder_layout = 0;      // 0: the launcher's choice by batch size
int encoder_layout() { return g_encoder_layout; }
static thread_local int g_coefs_variant = 0;
int coefs_kernel_variant() { return g_coefs_variant; }
static thread_local int g_encoder_segments = 0;
int encoder_segments_override() { return g_encoder_segments; }
static thread_local int g_encoder_persistent = 0;
int encoder_persistent_mode() { return g_encoder_persistent; }
static thread_local int g_hca_frames_per_group = 0;
int hca_frames_per_group_override() { return g_hca_frames_per_group; }

// The host pipeline runs an upload stream, a download stream and two lanes of kernels next to whatever streams the host
// has; the HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (4 unless told otherwise), and a copy
// stream that shares a queue with a kernel stream waits behind that stream's kernels (measured: every download of a
// 4096-channel encode ended only when the last kernel had, 660 ms instead of 533 ms).  The variable is read when the
// runtime initialises, so it is set when this library is loaded -- only if the host has not set it; a host that
// initialises HIP before loading the library sets it itself.
// Whether the runtime had already been brought up when this library was loaded: its KFD device is then open.
static bool runtime_is_up()
{
    DIR *d = opendir("/proc/self/fd");
    if (!d) return false;
    bool up = false;
    char path[64], target[64];
    while (dirent *e = readdir(d)) {
        std::snprintf(path, sizeof path, "/proc/self/fd/%s", e->d_name);
        const ssize_t n = readlink(path, target, sizeof target - 1);
        if (n > 0) {
            target[n] = 0;
            if (std::strcmp(target, "/dev/kfd") == 0) { up = true; break; }
        }
    }
    closedir(d);
    return up;
}

static int g_queues_seen_by_runtime = 4;             // what the HIP runtime reads (or read) from GPU_MAX_HW_QUEUES
__attribute__((constructor)) static void ask_for_hardware_queues()
{
    const char *host = std::getenv("GPU_MAX_HW_QUEUES");
    const char *opt_out = std::getenv("VGA_HIP_NO_ENV");
    if (host) {
        g_queues_seen_by_runtime = std::atoi(host);  // the host's choice (set before it brought the runtime up, one assumes)
    } else if (opt_out && opt_out[0] && opt_out[0] != '0') {
        g_queues_seen_by_runtime = 4;                // the runtime's default
    } else if (runtime_is_up()) {
        g_queues_seen_by_runtime = 4;                // too late: the runtime initialised without the variable
    } else {
        setenv("GPU_MAX_HW_QUEUES", "16", 0);
        g_queues_seen_by_runtime = 16;
    }
}

int hardware_queues_requested() { return g_queues_seen_by_runtime; }
static thread_local PipeOverride g_pipe_override;
PipeOverride &pipe_override() { return g_pipe_override; }

static std::mutex g_devices_mutex;
static std::vector<int> g_devices;                   // vga_set_devices(); empty = the caller's current device
std::vector<int> batch_devices()
{
    std::lock_guard<std::mutex> g(g_devices_mutex);
    return g_devices;
}
ThreadHooks capture_thread_hooks()
{
    return ThreadHooks{g_force_open_seams, g_encoder_layout, g_coefs_variant, g_encoder_segments, g_hca_frames_per_group, g_pipe_override, g_encoder_persistent};
}
void apply_thread_hooks(const ThreadHooks &h)
{
    g_force_open_seams = h.force_open_seams;
    g_encoder_layout = h.encoder_layout;
    g_coefs_variant = h.coefs_variant;
    g_encoder_segments = h.encoder_segments;
    g_encoder_persistent = h.encoder_persistent;
    g_hca_frames_per_group = h.hca_frames_per_group;
    g_pipe_override = h.pipe;
}
static thread_local PipeReport g_pipe_report;
PipeReport &pipe_report() { return g_pipe_report; }
static thread_local ProgressCallback g_progress_callback;
ProgressCallback progress_callback() { return g_progress_callback; }
static thread_local ProgressSink *g_progress_sink = nullptr;
ProgressSink *&current_progress_sink() { return g_progress_sink; }

int require_device()
{
    // every entry point that is going to touch the device passes here first: a failure of an EARLIER call on this thread
    // (an argument check, say) must not make this call's buffers synchronise the device when they are released
    g_err_pending = false;
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
static int divide_by_round_up(int v, int d) { return v / d + (v % d != 0 ? 1 : 0); }   // Utilities/Extensions.cs:145

}  // namespace vga

using namespace vga;

extern "C" {

const char *vga_last_error(void) { return g_err; }

int vga_testing_force_open_seams_this_thread(int mode)
{
    const int old = g_force_open_seams;
    g_force_open_seams = mode;
    return old;
}
int vga_testing_gc_encoder_layout_this_thread(int channels_per_wave)
{
    const int old = g_encoder_layout;
    if (channels_per_wave == 0 || channels_per_wave == 4 || channels_per_wave == 8) g_encoder_layout = channels_per_wave;
    return old;
}
int vga_testing_gc_coefs_variant_this_thread(int variant)
{
    const int old = g_coefs_variant;
    if (variant >= 0 && variant <= 3) g_coefs_variant = variant;
    return old;
}
int vga_testing_gc_encoder_segments_this_thread(int segments)
{
    const int old = g_encoder_segments;
    g_encoder_segments = segments > 0 ? segments : 0;
    return old;
}
int vga_testing_gc_encoder_persistent_this_thread(int mode)
{
    const int before = g_encoder_persistent;
    g_encoder_persistent = mode >= 0 && mode <= 2 ? mode : 0;
    return before;
}
int vga_testing_gc_plan_pieces(int cus, int groups, int frames, long long group_frames, int ragged, int *out5)
{
    if (!out5 || cus <= 0 || groups <= 0 || frames <= 0) return -1;
    bool persistent = false;
    gc::Pieces seg;
    const int segments = gc::plan_encode_pieces_on(cus, groups, frames, group_frames, ragged != 0, &persistent, &seg);
    out5[0] = segments;
    out5[1] = seg.big;
    out5[2] = seg.nb;
    out5[3] = seg.small;
    out5[4] = persistent ? 1 : 0;
    return 0;
}
int vga_testing_hca_frames_per_group_this_thread(int frames)
{
    const int old = g_hca_frames_per_group;
    g_hca_frames_per_group = frames > 0 ? frames : 0;
    return old;
}
void vga_testing_host_pipeline_this_thread(int feeders, int drainers, int chunk_units, int slot_bytes)
{
    g_pipe_override.feeders = feeders;
    g_pipe_override.drainers = drainers;
    g_pipe_override.chunk_units = chunk_units;
    g_pipe_override.slot_bytes = slot_bytes;
}
void vga_testing_host_pipeline_tail_this_thread(int tail_units) { g_pipe_override.tail_units = tail_units > 0 ? tail_units : 0; }
void vga_testing_buckets_order_this_thread(int order) { g_pipe_override.buckets_order = order == 1 || order == 2 ? order : 0; }
void vga_testing_host_transfer_this_thread(int mode) { g_pipe_override.transfer = mode == 1 ? 1 : 0; }
void vga_testing_host_compute_lanes_this_thread(int lanes) { g_pipe_override.compute_lanes = lanes > 0 ? lanes : 0; }
int vga_testing_plan_buckets(const int *group, const int *length, int n, int max_units, long long max_volume, int longest_first,
                             int *order_out, int *chunk_begin_out, int *chunk_length_out, int *chunk_group_out, int max_chunks)
{
    if (n < 0 || (n > 0 && (!group || !length))) return -1;
    const BucketPlan plan = plan_buckets(std::vector<int>(group, group + n), std::vector<int>(length, length + n), max_units, max_volume, longest_first != 0);
    const int chunks = (int)plan.chunk_begin.size() - 1;
    if (chunks > max_chunks) return -1;
    for (int i = 0; i < n && order_out; i++) order_out[i] = plan.order[i];
    for (int k = 0; k <= chunks && chunk_begin_out; k++) chunk_begin_out[k] = plan.chunk_begin[k];
    for (int k = 0; k < chunks; k++) {
        if (chunk_length_out) chunk_length_out[k] = plan.chunk_length[k];
        if (chunk_group_out) chunk_group_out[k] = plan.chunk_group[k];
    }
    return chunks;
}
int vga_testing_last_pipeline_stats(double *out, int n)
{
    const PipeReport &r = g_pipe_report;
    const double v[] = {r.stats.total, r.stats.setup, r.stats.feed_copy, r.stats.feed_wait_slot, r.stats.feed_issue, r.stats.feed_max,
                        r.stats.main_wait_upload, r.stats.main_launch, r.stats.main_tail_sync, r.stats.drain_wait_compute,
                        r.stats.drain_wait_copy, r.stats.drain_copy, r.stats.drain_max, (double)r.stats.feeders, (double)r.stats.drainers,
                        (double)r.stats.chunks, (double)r.stats.chunk_units, r.t_alloc, r.t_entry, r.stats.feed_boundary, r.stats.feed_final, r.stats.drain_register};
    const int m = (int)(sizeof v / sizeof v[0]);
    for (int i = 0; i < n && i < m; i++) out[i] = v[i];
    return m;
}
void vga_release_cached_memory(void)
{
    DevicePool::get().trim();
    pipe::PinnedPool::get().trim();
    pipe::MaskedStreamPool::get().trim();
}
const char *vga_version(void) { return "vgaudio_hip 0.2 (gfx950)"; }

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

int vga_set_devices(const int *devices, int count)
{
    if (count < 0 || (count > 0 && !devices) || count > 64) { set_error("vga_set_devices: bad device list"); return VGA_ERR_ARGUMENT; }
    int n = 0;
    if (count > 0 && (hipGetDeviceCount(&n) != hipSuccess || n <= 0)) {
        set_error("no HIP device available; libvgaudio_hip has no CPU fallback");
        return VGA_ERR_DEVICE;
    }
    for (int i = 0; i < count; i++)
        if (devices[i] < 0 || devices[i] >= n) { set_error("vga_set_devices: device %d of %d does not exist", devices[i], n); return VGA_ERR_ARGUMENT; }
    std::lock_guard<std::mutex> g(g_devices_mutex);
    g_devices.assign(devices, devices + count);
    return VGA_OK;
}

int vga_set_progress_callback(vga_progress_fn fn, void *user)
{
    vga::g_progress_callback.fn = fn;
    vga::g_progress_callback.user = fn ? user : nullptr;
    return VGA_OK;
}

int vga_get_devices(int *devices, int capacity)
{
    std::lock_guard<std::mutex> g(g_devices_mutex);
    for (int i = 0; i < (int)g_devices.size() && i < capacity && devices; i++) devices[i] = g_devices[i];
    return (int)g_devices.size();
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
    return (size_t)nch * (size_t)vga::gc::coef_record_pitch((int64_t)frames) * 16;
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

// ---------------------------------------------------------------- channel metadata
static int get_next_multiple(int value, int multiple)          // Utilities/Helpers.cs:71-80
{
    if (multiple <= 0) return value;
    if (value % multiple == 0) return value;
    return value + multiple - value % multiple;
}

int vga_gcadpcm_channel_layout_for(const vga_gcadpcm_channel_params *p, vga_gcadpcm_channel_layout *out)
{
    if (!p || !out) { set_error("null argument"); return VGA_ERR_ARGUMENT; }
    if (p->sample_count < 0 || p->loop_start < 0 || p->loop_end < p->loop_start || p->loop_alignment_multiple < 0 ||
        p->samples_per_seek_table_entry < 0) {
        set_error("channel parameters out of range (samples %d, loop %d..%d, alignment %d, seek entry %d)", p->sample_count,
                  p->loop_start, p->loop_end, p->loop_alignment_multiple, p->samples_per_seek_table_entry);
        return VGA_ERR_OUT_OF_RANGE;
    }
    const int multiple = p->loop_alignment_multiple;
    out->alignment_needed = (multiple != 0 && p->loop_start % multiple != 0) ? 1 : 0;    // Helpers.cs:82-83
    out->loop_start_aligned = p->loop_start;
    out->sample_count_aligned = p->sample_count;
    if (out->alignment_needed) {                                                         // GcAdpcmAlignment.cs:29-31
        const int64_t aligned = (int64_t)p->loop_start + multiple - p->loop_start % multiple;
        const int64_t count = (int64_t)p->loop_end + (aligned - p->loop_start);
        if (count > 0x7FFFFFFF - 16) { set_error("aligned sample count overflows"); return VGA_ERR_OUT_OF_RANGE; }
        out->loop_start_aligned = get_next_multiple(p->loop_start, multiple);
        out->sample_count_aligned = (int)count;
    }
    out->seek_table_entries = p->samples_per_seek_table_entry != 0                       // GcAdpcmSeekTable.cs:27
        ? divide_by_round_up(out->sample_count_aligned, p->samples_per_seek_table_entry) : 0;
    return VGA_OK;
}

size_t vga_gcadpcm_build_channels_workspace_bytes(int nch, const vga_gcadpcm_channel_params *p)
{
    vga_gcadpcm_channel_layout L;
    if (nch <= 0 || vga_gcadpcm_channel_layout_for(p, &L) != VGA_OK) return 0;
    // decoded PCM (caller may not want it) + the re-encode input + two history arrays + a status word
    size_t bytes = (size_t)nch * (size_t)round_up(L.sample_count_aligned > 0 ? L.sample_count_aligned : 1, 8) * 2;
    if (L.alignment_needed) {
        const int keep = p->loop_end / 14 * 14;
        bytes += (size_t)nch * (size_t)round_up(L.sample_count_aligned - keep + 1, 8) * 2;
    }
    return bytes + (size_t)round_up(nch * 2, 16) * 2 + 64;
}

int vga_gcadpcm_build_channels_device(const uint8_t *d_adpcm, int64_t adpcm_pitch, const int16_t *d_coefs, int nch,
                                      const vga_gcadpcm_channel_params *p, uint8_t *d_adpcm_out, int64_t out_pitch,
                                      int16_t *d_pcm_out, int64_t pcm_pitch, int16_t *d_seek_out, int64_t seek_pitch,
                                      int16_t *d_loop_context_out, void *d_workspace, size_t workspace_bytes, void *stream)
{
    vga_gcadpcm_channel_layout L;
    if (int rc = vga_gcadpcm_channel_layout_for(p, &L)) return rc;
    if (nch < 0) { set_error("negative channel count"); return VGA_ERR_ARGUMENT; }
    if (nch == 0) return VGA_OK;
    hipStream_t st = (hipStream_t)stream;
    const int n_al = L.sample_count_aligned;
    const int bytes_in = vga_gcadpcm_sample_count_to_byte_count(L.alignment_needed ? p->loop_end : p->sample_count);
    const int bytes_al = vga_gcadpcm_sample_count_to_byte_count(n_al);
    if (int rc = check_adpcm_layout(d_adpcm, adpcm_pitch, bytes_in, "vga_gcadpcm_build_channels_device (input)")) return rc;
    if (L.alignment_needed && !d_adpcm_out) {
        set_error("the loop needs alignment: adpcm_out is required");
        return VGA_ERR_ARGUMENT;
    }
    if (d_adpcm_out)
        if (int rc = check_adpcm_layout(d_adpcm_out, out_pitch, bytes_al, "vga_gcadpcm_build_channels_device (output)")) return rc;
    if (d_pcm_out)
        if (int rc = check_pcm_layout(d_pcm_out, pcm_pitch, n_al, "vga_gcadpcm_build_channels_device (pcm)")) return rc;
    if (d_seek_out && seek_pitch < 2 * (int64_t)L.seek_table_entries) { set_error("seek table pitch too small"); return VGA_ERR_ARGUMENT; }
    if (workspace_bytes < vga_gcadpcm_build_channels_workspace_bytes(nch, p) || !d_workspace || ((uintptr_t)d_workspace & 15)) {
        set_error("workspace too small or not 16-byte aligned: need %zu bytes", vga_gcadpcm_build_channels_workspace_bytes(nch, p));
        return VGA_ERR_ARGUMENT;
    }
    // the loop context reads the pred/scale byte from the ORIGINAL stream (GcAdpcmChannelBuilder.cs:179)
    const bool want_ctx = d_loop_context_out != nullptr;
    if (want_ctx && L.loop_start_aligned != 0 &&
        L.loop_start_aligned / 14 * 8 >= vga_gcadpcm_sample_count_to_byte_count(p->sample_count)) {
        set_error("loop context: the aligned loop start (%d) lies past the original ADPCM data (the reference reads "
                  "Adpcm, not AlignedAdpcm: IndexOutOfRangeException)", L.loop_start_aligned);
        return VGA_ERR_OUT_OF_RANGE;
    }

    // workspace carve-up
    uint8_t *w = static_cast<uint8_t *>(d_workspace);
    const int64_t ws_pcm_pitch = round_up(n_al > 0 ? n_al : 1, 8);
    int16_t *pcm = d_pcm_out ? d_pcm_out : reinterpret_cast<int16_t *>(w);
    const int64_t ppitch = d_pcm_out ? pcm_pitch : ws_pcm_pitch;
    w += (size_t)nch * ws_pcm_pitch * 2;

    const bool want_seek = d_seek_out && L.seek_table_entries > 0;
    const bool ctx_needs_pcm = want_ctx && L.loop_start_aligned != 0;
    if (L.alignment_needed) {                                   // GcAdpcmAlignment.cs:33-62
        const int loop_start = p->loop_start, loop_end = p->loop_end;
        const int frames_to_keep = loop_end / 14;
        const int bytes_to_keep = frames_to_keep * 8, samples_to_keep = frames_to_keep * 14;
        const int samples_to_encode = n_al - samples_to_keep;
        if (loop_end - loop_start <= 0 && loop_end - samples_to_keep < samples_to_encode) {
            set_error("a zero-length loop cannot be aligned (the reference's fill loop never ends, GcAdpcmAlignment.cs:48)");
            return VGA_ERR_INVALID_OP;
        }
        const int64_t new_pitch = round_up(samples_to_encode + 1, 8);
        int16_t *new_pcm = reinterpret_cast<int16_t *>(w);
        w += (size_t)nch * new_pitch * 2;
        int16_t *h1 = reinterpret_cast<int16_t *>(w);
        int16_t *h2 = h1 + round_up(nch, 8);
        // :41-43 oldPcm = Decode(adpcm, SampleCount = loopEnd) -> PcmAligned[0, loopEnd)
        if (int rc = gc::launch_decode(d_adpcm, adpcm_pitch, d_coefs, nch, loop_end, nullptr, nullptr, pcm, ppitch, nullptr, st))
            return rc;
        // :44-55 the tail to encode: rest of the last kept-from frame, then the loop, wrapped
        if (int rc = gc::launch_align_gather(pcm, ppitch, nch, loop_start, loop_end, samples_to_keep, samples_to_encode, new_pcm,
                                             new_pitch, h1, h2, st))
            return rc;
        // :57-59 AdpcmAligned = kept frames + Encode(newPcm, history of the last kept sample)
        if (bytes_to_keep > 0)
            VGA_HIP_TRY(hipMemcpy2DAsync(d_adpcm_out, (size_t)out_pitch, d_adpcm, (size_t)adpcm_pitch, (size_t)bytes_to_keep,
                                         (size_t)nch, hipMemcpyDeviceToDevice, st));
        if (int rc = gc::launch_encode(new_pcm, new_pitch, nch, samples_to_encode, d_coefs, h1, h2, d_adpcm_out + bytes_to_keep,
                                       out_pitch, st))
            return rc;
        // :61-62 PcmAligned[samplesToKeep..] = Decode(newAdpcm)
        if (int rc = gc::launch_decode(d_adpcm_out + bytes_to_keep, out_pitch, d_coefs, nch, samples_to_encode, h1, h2,
                                       pcm + samples_to_keep, ppitch, nullptr, st))
            return rc;
    } else {
        if (d_adpcm_out && bytes_al > 0)
            VGA_HIP_TRY(hipMemcpy2DAsync(d_adpcm_out, (size_t)out_pitch, d_adpcm, (size_t)adpcm_pitch, (size_t)bytes_al,
                                         (size_t)nch, hipMemcpyDeviceToDevice, st));
        if (d_pcm_out || want_seek || ctx_needs_pcm)            // EnsurePcmDecoded (GcAdpcmChannelBuilder.cs:202)
            if (int rc = gc::launch_decode(d_adpcm, adpcm_pitch, d_coefs, nch, n_al, nullptr, nullptr, pcm, ppitch, nullptr, st))
                return rc;
    }
    return gc::launch_channel_meta(d_adpcm, adpcm_pitch, pcm, ppitch, nch, L.loop_start_aligned, p->samples_per_seek_table_entry,
                                   want_seek ? L.seek_table_entries : 0, want_seek ? d_seek_out : nullptr, seek_pitch,
                                   d_loop_context_out, st);
}

// ---------------------------------------------------------------- DSP container
int vga_dsp_layout_for(const vga_dsp_params *p, int nch, vga_dsp_layout *out)
{
    if (!p || !out) { set_error("null argument"); return VGA_ERR_ARGUMENT; }
    if (nch < 1) { set_error("a DSP file needs at least one channel"); return VGA_ERR_ARGUMENT; }
    if (p->samples_per_interleave < 1) {                         // DspConfiguration.cs:31-40
        set_error("Number of samples per interleave must be positive");
        return VGA_ERR_OUT_OF_RANGE;
    }
    if (p->samples_per_interleave % 14 != 0) {
        set_error("Number of samples per interleave must be divisible by 14");
        return VGA_ERR_OUT_OF_RANGE;
    }
    if (p->sample_count < 0 || p->loop_start < 0 || p->loop_end < 0) { set_error("negative sample count / loop point"); return VGA_ERR_OUT_OF_RANGE; }
    // DspWriter.cs:22-36
    const int alignment_samples = get_next_multiple(p->loop_start, p->loop_point_alignment) - p->loop_start;
    out->loop_start = p->loop_start + alignment_samples;
    out->loop_end = p->loop_end + alignment_samples;
    out->sample_count = (p->trim_file && p->looping) ? out->loop_end : std::max(p->sample_count, out->loop_end);
    out->bytes_per_interleave = vga_gcadpcm_sample_count_to_byte_count(p->samples_per_interleave);
    out->frames_per_interleave = out->bytes_per_interleave / 8;
    out->start_addr = vga_gcadpcm_sample_to_nibble(p->looping ? out->loop_start : 0);
    out->end_addr = vga_gcadpcm_sample_to_nibble(p->looping ? out->loop_end : out->sample_count - 1);
    out->cur_addr = vga_gcadpcm_sample_to_nibble(0);
    out->audio_data_size = get_next_multiple(vga_gcadpcm_sample_count_to_byte_count(out->sample_count), nch == 1 ? 1 : 8);   // :99-100
    const int64_t fs = ((int64_t)0x60 + out->audio_data_size) * nch;                                                          // :18
    if (fs > 0x7FFFFFFF) { set_error("DSP file would exceed 2 GiB (the reference's FileSize is an int)"); return VGA_ERR_OUT_OF_RANGE; }
    out->file_size = (int)fs;
    return VGA_OK;
}

int vga_dsp_write_device(const uint8_t *d_adpcm, int64_t adpcm_pitch, int adpcm_len, const int16_t *d_coefs,
                         const int16_t *d_gain, const int16_t *d_start_context, const int16_t *d_loop_context, int nch,
                         const vga_dsp_params *p, uint8_t *d_file, void *stream)
{
    vga_dsp_layout L;
    if (int rc = vga_dsp_layout_for(p, nch, &L)) return rc;
    if (adpcm_len < 0 || !d_coefs || !d_file || (adpcm_len > 0 && !d_adpcm)) { set_error("null / negative argument"); return VGA_ERR_ARGUMENT; }
    if (adpcm_len > 0)
        if (int rc = check_adpcm_layout(d_adpcm, adpcm_pitch, adpcm_len, "vga_dsp_write_device")) return rc;
    if (((uintptr_t)d_file & 7) != 0) { set_error("file image must be 8-byte aligned"); return VGA_ERR_ARGUMENT; }
    const int mono_bytes = vga_gcadpcm_sample_count_to_byte_count(L.sample_count);
    if (nch == 1 && mono_bytes > adpcm_len) {                   // Stream.Write(buffer, 0, count) past the array
        set_error("channel audio (%d bytes) is shorter than the %d bytes the header's sample count needs", adpcm_len, mono_bytes);
        return VGA_ERR_ARGUMENT;
    }
    return gc::launch_dsp_image(d_adpcm, adpcm_pitch, adpcm_len, d_coefs, d_gain, d_start_context, d_loop_context, nch,
                                L.sample_count, vga_gcadpcm_sample_count_to_nibble_count(L.sample_count), p->sample_rate,
                                p->looping ? 1 : 0, L.start_addr, L.end_addr, L.cur_addr, L.bytes_per_interleave,
                                L.frames_per_interleave, L.audio_data_size, mono_bytes, d_file, (size_t)L.file_size,
                                (hipStream_t)stream);
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

// Channels per pipeline chunk (LABNOTES.md 5, host path): small enough that the first kernels start after a quarter
// of configs[1] has arrived, large enough that the coefficient kernel (one wave per channel) still has a wave per SIMD.
static constexpr int GC_CHUNK_CHANNELS = 1024;

// channels per share when a call is spread over several GPUs (vga_set_devices): below this one GPU's pipeline is faster
static constexpr int GC_MIN_SHARE_CHANNELS = 128;

static int calculate_coefficients_batch_one(const int16_t *const *pcm, int nch, int length, int16_t *coefs_out);
int vga_gcadpcm_calculate_coefficients_batch(const int16_t *const *pcm, int nch, int length, int16_t *coefs_out)
{
    if (nch <= 0 || !pcm || !coefs_out) return calculate_coefficients_batch_one(pcm, nch, length, coefs_out);
    return for_each_device_share(nch, GC_MIN_SHARE_CHANNELS, [&](int first, int count) {
        return calculate_coefficients_batch_one(pcm + first, count, length, coefs_out + (size_t)first * 16);
    });
}
static int calculate_coefficients_batch_one(const int16_t *const *pcm, int nch, int length, int16_t *coefs_out)
{
    if (length < 0) { set_error("negative length"); return VGA_ERR_ARGUMENT; }
    if (int rc = check_ptrs((const void *const *)pcm, length > 0 ? nch : 0, "pcm")) return rc;
    if (nch < 0 || (nch > 0 && !coefs_out)) { set_error("bad coefs_out/nch"); return VGA_ERR_ARGUMENT; }
    if (nch == 0) return VGA_OK;
    if (int rc = require_device()) return rc;
    GcBatch b;
    b.pcm_pitch = round_up(length > 0 ? length : 1, 8);
    VGA_HIP_TRY(b.pcm.alloc((size_t)nch * b.pcm_pitch * sizeof(int16_t)));
    VGA_HIP_TRY(b.coefs.alloc((size_t)nch * 32));
    pipe::Job job;
    job.units = nch;
    if (length > 0) {
        job.in_rows = (const void *const *)pcm;
        job.in_row_bytes = (size_t)length * sizeof(int16_t);
        job.d_in = b.pcm.as<char>();
        job.d_in_pitch = (size_t)b.pcm_pitch * sizeof(int16_t);
    }
    job.compute = [&](int first, int count, hipStream_t s, std::string &why) -> int {
        const int rc = gc::launch_coefs(b.pcm.as<int16_t>() + (int64_t)first * b.pcm_pitch, b.pcm_pitch, count, length,
                                        b.coefs.as<int16_t>() + (int64_t)first * 16, b.ws.p, s);
        if (rc) why = vga_last_error();
        return rc;
    };
    // one chunk's workspace: the chunks' kernels run one after the other on the compute stream
    VGA_HIP_TRY(b.ws.alloc(vga_gcadpcm_coefs_workspace_bytes(planned_chunk_units(job, GC_CHUNK_CHANNELS), length)));
    if (int rc = run_batch_pipeline(job, GC_CHUNK_CHANNELS)) return rc;
    VGA_HIP_TRY(hipMemcpy(coefs_out, b.coefs.p, (size_t)nch * 32, hipMemcpyDeviceToHost));
    return VGA_OK;
}

static int encode_with_coefs_batch_one(const int16_t *const *pcm, int nch, int pcm_length, int sample_count, const int16_t *coefs,
                                      const int16_t *hist1, const int16_t *hist2, uint8_t *const *adpcm_out);
int vga_gcadpcm_encode_with_coefs_batch(const int16_t *const *pcm, int nch, int pcm_length, int sample_count,
                                        const int16_t *coefs, const int16_t *hist1, const int16_t *hist2,
                                        uint8_t *const *adpcm_out)
{
    if (nch <= 0 || !pcm || !coefs || !adpcm_out)
        return encode_with_coefs_batch_one(pcm, nch, pcm_length, sample_count, coefs, hist1, hist2, adpcm_out);
    return for_each_device_share(nch, GC_MIN_SHARE_CHANNELS, [&](int first, int count) {
        return encode_with_coefs_batch_one(pcm + first, count, pcm_length, sample_count, coefs + (size_t)first * 16,
                                          hist1 ? hist1 + first : nullptr, hist2 ? hist2 + first : nullptr, adpcm_out + first);
    });
}
static int encode_with_coefs_batch_one(const int16_t *const *pcm, int nch, int pcm_length, int sample_count, const int16_t *coefs,
                                      const int16_t *hist1, const int16_t *hist2, uint8_t *const *adpcm_out)
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
    b.pcm_pitch = round_up(sample_count, 8);
    VGA_HIP_TRY(b.pcm.alloc((size_t)nch * b.pcm_pitch * sizeof(int16_t)));
    if (int rc = upload_hist(b, nch, hist1, hist2)) return rc;
    VGA_HIP_TRY(b.coefs.alloc((size_t)nch * 32));
    VGA_HIP_TRY(hipMemcpyAsync(b.coefs.p, coefs, (size_t)nch * 32, hipMemcpyHostToDevice, b.st.s));
    VGA_HIP_TRY(hipStreamSynchronize(b.st.s));
    const int nbytes = vga_gcadpcm_sample_count_to_byte_count(sample_count);
    b.adpcm_pitch = round_up(nbytes, 16);
    VGA_HIP_TRY(b.adpcm.alloc((size_t)nch * b.adpcm_pitch));
    DevBuf scratch;                                       // the encoder's piece states: one chunk at a time uses it
    pipe::Job job;
    job.units = nch;
    job.in_rows = (const void *const *)pcm;
    job.in_row_bytes = (size_t)sample_count * sizeof(int16_t);
    job.d_in = b.pcm.as<char>();
    job.d_in_pitch = (size_t)b.pcm_pitch * sizeof(int16_t);
    job.out_rows = (void *const *)adpcm_out;
    job.out_row_bytes = (size_t)nbytes;
    job.d_out = b.adpcm.as<char>();
    job.d_out_pitch = (size_t)b.adpcm_pitch;
    job.compute = [&](int first, int count, hipStream_t s, std::string &why) -> int {
        const int rc = gc::launch_encode(b.pcm.as<int16_t>() + (int64_t)first * b.pcm_pitch, b.pcm_pitch, count, sample_count,
                                         b.coefs.as<int16_t>() + (int64_t)first * 16,
                                         b.h1.p ? b.h1.as<int16_t>() + first : nullptr, b.h2.p ? b.h2.as<int16_t>() + first : nullptr,
                                         b.adpcm.as<uint8_t>() + (int64_t)first * b.adpcm_pitch, b.adpcm_pitch, s, scratch.p, scratch.bytes);
        if (rc) why = vga_last_error();
        return rc;
    };
    VGA_HIP_TRY(scratch.alloc(gc::encode_scratch_bytes(planned_chunk_units(job, GC_CHUNK_CHANNELS))));
    return run_batch_pipeline(job, GC_CHUNK_CHANNELS);
}

static int encode_batch_one(const int16_t *const *pcm, int nch, int sample_count, int16_t hist1, int16_t hist2, int16_t *coefs_out,
                           uint8_t *const *adpcm_out);
int vga_gcadpcm_encode_batch(const int16_t *const *pcm, int nch, int sample_count, int16_t hist1, int16_t hist2,
                             int16_t *coefs_out, uint8_t *const *adpcm_out)
{
    if (nch <= 0 || !pcm || !coefs_out || !adpcm_out) return encode_batch_one(pcm, nch, sample_count, hist1, hist2, coefs_out, adpcm_out);
    return for_each_device_share(nch, GC_MIN_SHARE_CHANNELS, [&](int first, int count) {
        return encode_batch_one(pcm + first, count, sample_count, hist1, hist2, coefs_out + (size_t)first * 16, adpcm_out + first);
    });
}
static int encode_batch_one(const int16_t *const *pcm, int nch, int sample_count, int16_t hist1, int16_t hist2, int16_t *coefs_out,
                           uint8_t *const *adpcm_out)
{
    if (sample_count < 0) { set_error("negative sample count"); return VGA_ERR_ARGUMENT; }
    if (int rc = check_ptrs((const void *const *)pcm, sample_count > 0 ? nch : 0, "pcm")) return rc;
    if (int rc = check_ptrs((const void *const *)adpcm_out, sample_count > 0 ? nch : 0, "adpcm_out")) return rc;
    if (nch < 0 || (nch > 0 && !coefs_out)) { set_error("bad coefs_out/nch"); return VGA_ERR_ARGUMENT; }
    if (nch == 0) return VGA_OK;
    if (int rc = require_device()) return rc;
    const double t_entry = pipe::detail::now();
    GcBatch b;
    VGA_HIP_TRY(b.st.create());
    b.pcm_pitch = round_up(sample_count > 0 ? sample_count : 1, 8);
    VGA_HIP_TRY(b.pcm.alloc((size_t)nch * b.pcm_pitch * sizeof(int16_t)));
    std::vector<int16_t> h1v((size_t)nch, hist1), h2v((size_t)nch, hist2);
    const bool use_hist = hist1 != 0 || hist2 != 0;
    if (use_hist)
        if (int rc = upload_hist(b, nch, h1v.data(), h2v.data())) return rc;
    VGA_HIP_TRY(hipStreamSynchronize(b.st.s));
    VGA_HIP_TRY(b.coefs.alloc((size_t)nch * 32));
    const int nbytes = vga_gcadpcm_sample_count_to_byte_count(sample_count);
    b.adpcm_pitch = round_up(nbytes > 0 ? nbytes : 1, 16);
    VGA_HIP_TRY(b.adpcm.alloc((size_t)nch * b.adpcm_pitch));
    // two compute lanes: a chunk's kernels need not wait for the chunk before (the short chunks at the end of the upload
    // are bound by the latency of one channel's coefficient search, ~25 ms, not by the chip); each lane has its own
    // piece states and workspace
    constexpr int LANES = 2;
    DevBuf scratch[LANES], ws[LANES];
    pipe::Job job;
    job.units = nch;
    job.compute_lanes = hardware_queues_requested() >= 6 ? LANES : 1;   // a lane more than the queues hold would stall the copies
    if (sample_count > 0) {
        job.in_rows = (const void *const *)pcm;
        job.in_row_bytes = (size_t)sample_count * sizeof(int16_t);
        job.d_in = b.pcm.as<char>();
        job.d_in_pitch = (size_t)b.pcm_pitch * sizeof(int16_t);
        job.out_rows = (void *const *)adpcm_out;
        job.out_row_bytes = (size_t)nbytes;
        job.d_out = b.adpcm.as<char>();
        job.d_out_pitch = (size_t)b.adpcm_pitch;
    }
    // EncodeChannel (GcAdpcmFormat.cs:129-135): coefficients, then encode -- per chunk of channels, so that the next
    // chunk's upload and the previous chunk's download overlap these kernels
    job.compute = [&](int first, int count, hipStream_t s, std::string &why) -> int {
        const int lane = pipe::compute_lane();
        int rc = gc::launch_coefs(b.pcm.as<int16_t>() + (int64_t)first * b.pcm_pitch, b.pcm_pitch, count, sample_count,
                                  b.coefs.as<int16_t>() + (int64_t)first * 16, ws[lane].p, s);
        if (!rc)
            rc = gc::launch_encode(b.pcm.as<int16_t>() + (int64_t)first * b.pcm_pitch, b.pcm_pitch, count, sample_count,
                                   b.coefs.as<int16_t>() + (int64_t)first * 16,
                                   b.h1.p ? b.h1.as<int16_t>() + first : nullptr, b.h2.p ? b.h2.as<int16_t>() + first : nullptr,
                                   b.adpcm.as<uint8_t>() + (int64_t)first * b.adpcm_pitch, b.adpcm_pitch, s, scratch[lane].p,
                                   scratch[lane].bytes);
        if (rc) why = vga_last_error();
        return rc;
    };
    const int chunk = planned_chunk_units(job, GC_CHUNK_CHANNELS);
    const int lanes_used = nch > 1 ? job.compute_lanes : 1;           // (a single chunk is split in two as well)
    for (int l = 0; l < lanes_used; l++) {
        VGA_HIP_TRY(scratch[l].alloc(gc::encode_scratch_bytes(chunk)));
        VGA_HIP_TRY(ws[l].alloc(vga_gcadpcm_coefs_workspace_bytes(chunk, sample_count)));
    }
    pipe_report().t_alloc = pipe::detail::now() - t_entry;
    if (int rc = run_batch_pipeline(job, GC_CHUNK_CHANNELS)) return rc;
    VGA_HIP_TRY(hipMemcpy(coefs_out, b.coefs.p, (size_t)nch * 32, hipMemcpyDeviceToHost));
    pipe_report().t_entry = pipe::detail::now() - t_entry;      // without the buffers' release (the destructors below)
    return VGA_OK;
}

static int decode_batch_one(const uint8_t *const *adpcm, const int16_t *coefs, int nch, int sample_count, const int16_t *hist1,
                           const int16_t *hist2, int16_t *const *pcm_out);
int vga_gcadpcm_decode_batch(const uint8_t *const *adpcm, const int16_t *coefs, int nch, int sample_count,
                             const int16_t *hist1, const int16_t *hist2, int16_t *const *pcm_out)
{
    if (nch <= 0 || !adpcm || !coefs || !pcm_out) return decode_batch_one(adpcm, coefs, nch, sample_count, hist1, hist2, pcm_out);
    return for_each_device_share(nch, GC_MIN_SHARE_CHANNELS, [&](int first, int count) {
        return decode_batch_one(adpcm + first, coefs + (size_t)first * 16, count, sample_count, hist1 ? hist1 + first : nullptr,
                               hist2 ? hist2 + first : nullptr, pcm_out + first);
    });
}
static int decode_batch_one(const uint8_t *const *adpcm, const int16_t *coefs, int nch, int sample_count, const int16_t *hist1,
                           const int16_t *hist2, int16_t *const *pcm_out)
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
    VGA_HIP_TRY(hipMemcpyAsync(b.coefs.p, coefs, (size_t)nch * 32, hipMemcpyHostToDevice, b.st.s));
    if (int rc = upload_hist(b, nch, hist1, hist2)) return rc;
    VGA_HIP_TRY(hipStreamSynchronize(b.st.s));
    pipe::Job job;
    job.units = nch;
    job.in_rows = (const void *const *)adpcm;
    job.in_row_bytes = (size_t)nbytes;
    job.d_in = b.adpcm.as<char>();
    job.d_in_pitch = (size_t)b.adpcm_pitch;
    job.out_rows = (void *const *)pcm_out;
    job.out_row_bytes = (size_t)sample_count * 2;
    job.d_out = b.pcm.as<char>();
    job.d_out_pitch = (size_t)b.pcm_pitch * 2;
    job.compute = [&](int first, int count, hipStream_t s, std::string &why) -> int {
        const int rc = gc::launch_decode(b.adpcm.as<uint8_t>() + (int64_t)first * b.adpcm_pitch, b.adpcm_pitch,
                                         b.coefs.as<int16_t>() + (int64_t)first * 16, count, sample_count,
                                         b.h1.p ? b.h1.as<int16_t>() + first : nullptr, b.h2.p ? b.h2.as<int16_t>() + first : nullptr,
                                         b.pcm.as<int16_t>() + (int64_t)first * b.pcm_pitch, b.pcm_pitch, b.status.as<int>(), s);
        if (rc) why = vga_last_error();
        return rc;
    };
    if (int rc = run_batch_pipeline(job, 2 * GC_CHUNK_CHANNELS)) return rc;
    int status = 0;
    VGA_HIP_TRY(hipMemcpy(&status, b.status.p, sizeof(int), hipMemcpyDeviceToHost));
    if (status != 0) {
        set_error("a frame header names predictor > 7 (the reference throws IndexOutOfRangeException)");
        return VGA_ERR_ARGUMENT;
    }
    return VGA_OK;
}

// DspWriter.GetFile for channels held in host memory: the image is assembled on the device and copied back once.
int vga_dsp_write(const uint8_t *const *adpcm, int adpcm_len, const int16_t *coefs, const int16_t *gain,
                  const int16_t *start_context, const int16_t *loop_context, int nch, const vga_dsp_params *p,
                  uint8_t *file_out)
{
    vga_dsp_layout L;
    if (int rc = vga_dsp_layout_for(p, nch, &L)) return rc;
    if (adpcm_len < 0) { set_error("negative length"); return VGA_ERR_ARGUMENT; }
    if (int rc = check_ptrs((const void *const *)adpcm, adpcm_len > 0 ? nch : 0, "adpcm")) return rc;
    if (!coefs || !file_out) { set_error("null coefs / output"); return VGA_ERR_ARGUMENT; }
    if (int rc = require_device()) return rc;
    GcBatch b;
    VGA_HIP_TRY(b.st.create());
    b.adpcm_pitch = round_up(adpcm_len > 0 ? adpcm_len : 1, 16);
    DevBuf file, d_gain, d_sc, d_lc;
    VGA_HIP_TRY(b.adpcm.alloc((size_t)nch * b.adpcm_pitch));
    VGA_HIP_TRY(b.coefs.alloc((size_t)nch * 32));
    VGA_HIP_TRY(file.alloc((size_t)L.file_size));
    for (int c = 0; c < nch; c++)
        if (adpcm_len > 0)
            VGA_HIP_TRY(hipMemcpyAsync(b.adpcm.as<uint8_t>() + (int64_t)c * b.adpcm_pitch, adpcm[c], (size_t)adpcm_len,
                                       hipMemcpyHostToDevice, b.st.s));
    VGA_HIP_TRY(hipMemcpyAsync(b.coefs.p, coefs, (size_t)nch * 32, hipMemcpyHostToDevice, b.st.s));
    auto upload = [&](DevBuf &d, const int16_t *src, size_t shorts) -> int {
        if (!src) return VGA_OK;
        VGA_HIP_TRY(d.alloc(shorts * 2));
        VGA_HIP_TRY(hipMemcpyAsync(d.p, src, shorts * 2, hipMemcpyHostToDevice, b.st.s));
        return VGA_OK;
    };
    if (int rc = upload(d_gain, gain, (size_t)nch)) return rc;
    if (int rc = upload(d_sc, start_context, (size_t)nch * 3)) return rc;
    if (int rc = upload(d_lc, loop_context, (size_t)nch * 3)) return rc;
    if (int rc = vga_dsp_write_device(b.adpcm.as<uint8_t>(), b.adpcm_pitch, adpcm_len, b.coefs.as<int16_t>(),
                                      gain ? d_gain.as<int16_t>() : nullptr, start_context ? d_sc.as<int16_t>() : nullptr,
                                      loop_context ? d_lc.as<int16_t>() : nullptr, nch, p, file.as<uint8_t>(), b.st.s))
        return rc;
    VGA_HIP_TRY(hipMemcpyAsync(file_out, file.p, (size_t)L.file_size, hipMemcpyDeviceToHost, b.st.s));
    VGA_HIP_TRY(hipStreamSynchronize(b.st.s));
    return VGA_OK;
}

// GcAdpcmChannel(GcAdpcmChannelBuilder) for a batch of freshly encoded channels that share one loop
// (GcAdpcmFormat.cs:27-40): alignment re-encode, loop context, seek table.  Outputs may be null.
int vga_gcadpcm_build_channels_batch(const uint8_t *const *adpcm, const int16_t *coefs, int nch,
                                     const vga_gcadpcm_channel_params *p, uint8_t *const *adpcm_out,
                                     int16_t *const *pcm_out, int16_t *const *seek_table_out, int16_t *loop_context_out)
{
    vga_gcadpcm_channel_layout L;
    if (int rc = vga_gcadpcm_channel_layout_for(p, &L)) return rc;
    if (int rc = check_ptrs((const void *const *)adpcm, nch, "adpcm")) return rc;
    if (nch > 0 && !coefs) { set_error("null coefs"); return VGA_ERR_ARGUMENT; }
    if (nch == 0) return VGA_OK;
    if (L.alignment_needed && !adpcm_out) { set_error("the loop needs alignment: adpcm_out is required"); return VGA_ERR_ARGUMENT; }
    if (adpcm_out) if (int rc = check_ptrs((const void *const *)adpcm_out, nch, "adpcm_out")) return rc;
    if (pcm_out) if (int rc = check_ptrs((const void *const *)pcm_out, L.sample_count_aligned > 0 ? nch : 0, "pcm_out")) return rc;
    if (seek_table_out && L.seek_table_entries > 0)
        if (int rc = check_ptrs((const void *const *)seek_table_out, nch, "seek_table_out")) return rc;
    if (int rc = require_device()) return rc;
    GcBatch b;
    VGA_HIP_TRY(b.st.create());
    const int bytes_in = vga_gcadpcm_sample_count_to_byte_count(p->sample_count);
    const int bytes_al = vga_gcadpcm_sample_count_to_byte_count(L.sample_count_aligned);
    const int64_t in_pitch = round_up(bytes_in > 0 ? bytes_in : 1, 16);
    b.adpcm_pitch = round_up(bytes_al > 0 ? bytes_al : 1, 16);
    b.pcm_pitch = round_up(L.sample_count_aligned > 0 ? L.sample_count_aligned : 1, 8);
    const int64_t seek_pitch = round_up(2 * (L.seek_table_entries > 0 ? L.seek_table_entries : 1), 8);
    DevBuf in, seek, ctx;
    VGA_HIP_TRY(in.alloc((size_t)nch * in_pitch));
    VGA_HIP_TRY(b.adpcm.alloc((size_t)nch * b.adpcm_pitch));
    VGA_HIP_TRY(b.pcm.alloc((size_t)nch * b.pcm_pitch * 2));
    VGA_HIP_TRY(b.coefs.alloc((size_t)nch * 32));
    VGA_HIP_TRY(seek.alloc((size_t)nch * seek_pitch * 2));
    VGA_HIP_TRY(ctx.alloc((size_t)nch * 6));
    const size_t wsb = vga_gcadpcm_build_channels_workspace_bytes(nch, p);
    VGA_HIP_TRY(b.ws.alloc(wsb));
    for (int c = 0; c < nch; c++)
        if (bytes_in > 0)
            VGA_HIP_TRY(hipMemcpyAsync(in.as<uint8_t>() + (int64_t)c * in_pitch, adpcm[c], (size_t)bytes_in,
                                       hipMemcpyHostToDevice, b.st.s));
    VGA_HIP_TRY(hipMemcpyAsync(b.coefs.p, coefs, (size_t)nch * 32, hipMemcpyHostToDevice, b.st.s));
    if (int rc = vga_gcadpcm_build_channels_device(in.as<uint8_t>(), in_pitch, b.coefs.as<int16_t>(), nch, p,
                                                   (adpcm_out || L.alignment_needed) ? b.adpcm.as<uint8_t>() : nullptr,
                                                   b.adpcm_pitch, pcm_out ? b.pcm.as<int16_t>() : nullptr, b.pcm_pitch,
                                                   (seek_table_out && L.seek_table_entries > 0) ? seek.as<int16_t>() : nullptr,
                                                   seek_pitch, loop_context_out ? ctx.as<int16_t>() : nullptr, b.ws.p, wsb, b.st.s))
        return rc;
    if (adpcm_out)
        if (int rc = download_adpcm(b, adpcm_out, nch, bytes_al)) return rc;
    for (int c = 0; c < nch; c++) {
        if (pcm_out && L.sample_count_aligned > 0)
            VGA_HIP_TRY(hipMemcpyAsync(pcm_out[c], b.pcm.as<int16_t>() + (int64_t)c * b.pcm_pitch,
                                       (size_t)L.sample_count_aligned * 2, hipMemcpyDeviceToHost, b.st.s));
        if (seek_table_out && L.seek_table_entries > 0)
            VGA_HIP_TRY(hipMemcpyAsync(seek_table_out[c], seek.as<int16_t>() + (int64_t)c * seek_pitch,
                                       (size_t)L.seek_table_entries * 4, hipMemcpyDeviceToHost, b.st.s));
    }
    if (loop_context_out)
        VGA_HIP_TRY(hipMemcpyAsync(loop_context_out, ctx.p, (size_t)nch * 6, hipMemcpyDeviceToHost, b.st.s));
    VGA_HIP_TRY(hipStreamSynchronize(b.st.s));
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
