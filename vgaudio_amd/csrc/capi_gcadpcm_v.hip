// capi_gcadpcm_v.hip -- GC-ADPCM over RAGGED batches: channels of different lengths in one call.
//
// The reference's batch path is a Parallel.ForEach over FILES (VGAudio.Cli/Batch.cs:24-25 -> Convert.cs:19), every file
// with its own length and channel count, each ending up in GcAdpcmFormat.EncodeFromPcm16 (Formats/GcAdpcm/GcAdpcmFormat.cs:
// 58-74) one channel at a time.  One call per file leaves the chip idle (a lone 60 s channel: 0.4 % of the batch rate);
// here the channels of many files travel in ONE call.  The kernels are the ones of the equal-length entry points with
// per-channel shapes read from device tables (gc::Ragged, gcadpcm_kernels.hpp): channels are cut into time pieces of one
// common length, work slots are handed out longest channel first.
#include "common.hpp"
#include "gcadpcm_kernels.hpp"
#include "host_batch.hpp"

#include <algorithm>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

namespace vga {
namespace {

constexpr int64_t GUARD_BYTES = 256;      // after the last row of a packed buffer: clamped loads of short rows stay inside

// Shapes of one group of channels (a whole ragged batch, or one pipeline chunk of it), as the kernels index them:
// LOCAL channel i = 0 .. count-1; offsets count from the device buffers' bases.
struct RaggedShape {
    int count = 0;
    std::vector<int> length, order;
    std::vector<int64_t> pcm_off, adpcm_off, rec_off;
    int max_length = 0;
    int solo_channels = 0, solo_usable = 0;   // gc::ragged_solo_count: the coefficient search's five-wave channels
    int64_t total_frames = 0, records = 0;     // records: slots of the coefficient workspace (an empty channel owns one)
    bool uniform = false;                      // every channel the same length: the equal-length kernels apply
    int64_t pcm_pitch = 0, adpcm_pitch = 0;    // ... with these pitches
    // the encoder's plan (gc::plan_encode_pieces) and, for persistent workgroups, its items biggest first: the queue then
    // ends with the short ones (a channel's partial last piece, the pieces of short files) and little is left to wait for
    gc::Pieces seg;
    int segments = 1;
    bool persistent = false;
    std::vector<uint32_t> items;

    // lengths[0 .. count); the rows start at pcm_base (samples) / adpcm_base (bytes) and follow each other, every row
    // rounded up to 8 samples / 16 bytes
    void build(const int *lengths, int n, int64_t pcm_base, int64_t adpcm_base, int64_t *pcm_end = nullptr, int64_t *adpcm_end = nullptr)
    {
        count = n;
        length.assign(lengths, lengths + n);
        order.resize(n);
        pcm_off.resize(n);
        adpcm_off.resize(n);
        rec_off.resize(n);
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return length[a] > length[b]; });
        max_length = 0;
        total_frames = records = 0;
        uniform = n > 0;
        for (int c = 0; c < n; c++) {
            pcm_off[c] = pcm_base;
            adpcm_off[c] = adpcm_base;
            rec_off[c] = records;
            const int64_t frames = ((int64_t)length[c] + 13) / 14;
            pcm_base += round_up(length[c], 8);
            adpcm_base += round_up(vga_gcadpcm_sample_count_to_byte_count(length[c]), 16);
            records += vga::gc::coef_record_pitch(frames);
            total_frames += frames;
            max_length = std::max(max_length, length[c]);
            uniform = uniform && length[c] == length[0];
        }
        if (uniform) {
            pcm_pitch = round_up(length[0], 8);
            adpcm_pitch = round_up(vga_gcadpcm_sample_count_to_byte_count(length[0]), 16);
        }
        if (pcm_end) *pcm_end = pcm_base;
        if (adpcm_end) *adpcm_end = adpcm_base;
        items.clear();
        solo_channels = solo_usable = 0;
        if (n > 0 && !uniform && max_length > 0) {
            std::vector<int> by_length(n);
            for (int i = 0; i < n; i++) by_length[i] = length[order[i]];
            solo_channels = gc::ragged_solo_count(by_length.data(), n, total_frames, device_cu_count(), &solo_usable);
            const int groups = (n + 15) / 16;
            std::vector<int> gframes(groups);
            int64_t group_frames = 0;
            for (int g = 0; g < groups; g++) {
                gframes[g] = (length[order[g * 16]] + 13) / 14;                   // slot 0 of a group holds its longest channel
                group_frames += gframes[g];
            }
            segments = gc::plan_encode_pieces(groups, (max_length + 13) / 14, group_frames, true, &persistent, &seg);
            if (persistent && groups < (1 << 20) && segments <= 4096) {
                struct Item { int size, y, g; };
                std::vector<Item> list;
                for (int y = 0; y < segments; y++)
                    for (int g = 0; g < groups; g++)
                        if (seg.first(y) < gframes[g])
                            list.push_back({(int)std::min<int64_t>(seg.frames(y), gframes[g] - seg.first(y)), y, g});
                std::stable_sort(list.begin(), list.end(), [](const Item &a, const Item &b) { return a.size > b.size; });
                items.reserve(list.size());
                for (const Item &it : list) items.push_back(((uint32_t)it.y << 20) | (uint32_t)it.g);
            } else
                persistent = false;
        }
    }
    // bytes of the device image of the tables: order, length (int32), then pcm_off, adpcm_off, rec_off (int64)
    size_t table_bytes() const { return (size_t)round_up((int64_t)count * 8, 16) + (size_t)count * 24 + items.size() * 4; }
    void write_tables(unsigned char *host) const
    {
        int *o = reinterpret_cast<int *>(host);
        int *l = o + count;
        int64_t *p = reinterpret_cast<int64_t *>(host + round_up((int64_t)count * 8, 16));
        for (int c = 0; c < count; c++) {
            o[c] = order[c];
            l[c] = length[c];
            p[c] = pcm_off[c];
            p[count + c] = adpcm_off[c];
            p[2 * count + c] = rec_off[c];
        }
        if (!items.empty()) memcpy(p + 3 * (size_t)count, items.data(), items.size() * 4);
    }
    gc::Ragged device_view(const unsigned char *dev) const
    {
        gc::Ragged r;
        r.order = reinterpret_cast<const int *>(dev);
        r.length = r.order + count;
        r.pcm_off = reinterpret_cast<const int64_t *>(dev + round_up((int64_t)count * 8, 16));
        r.adpcm_off = r.pcm_off + count;
        r.rec_off = r.pcm_off + 2 * count;
        r.max_length = max_length;
        r.total_frames = total_frames;
        r.solo_channels = solo_channels;
        r.solo_usable = solo_usable;
        if (!items.empty()) {
            r.items = reinterpret_cast<const uint32_t *>(r.pcm_off + 3 * (size_t)count);
            r.n_items = (int)items.size();
            r.segments = segments;
            r.persistent = persistent ? 1 : 0;
            r.seg = seg;
        }
        return r;
    }
};

int check_counts(const int *counts, int n, const char *what)
{
    if (n < 0) { set_error("%s: negative channel count", what); return VGA_ERR_ARGUMENT; }
    if (n > 0 && !counts) { set_error("%s: null sample counts", what); return VGA_ERR_ARGUMENT; }
    for (int c = 0; c < n; c++)
        if (counts[c] < 0) { set_error("%s: channel %d has a negative sample count", what, c); return VGA_ERR_ARGUMENT; }
    return VGA_OK;
}

int check_rows(const void *const *pp, const int *counts, int n, const char *what)
{
    if (n > 0 && !pp) { set_error("%s: null channel array", what); return VGA_ERR_ARGUMENT; }
    for (int c = 0; c < n; c++)
        if (counts[c] > 0 && !pp[c]) { set_error("%s: channel %d is null", what, c); return VGA_ERR_ARGUMENT; }
    return VGA_OK;
}

// the three launches on one group of channels, uniform groups through the equal-length kernels
int launch_coefs_group(const RaggedShape &sh, const gc::Ragged &rg, const int16_t *d_pcm, int16_t *d_coefs, void *ws, hipStream_t s)
{
    if (sh.count <= 0) return VGA_OK;
    if (sh.uniform)
        return gc::launch_coefs(d_pcm + sh.pcm_off[0], sh.pcm_pitch, sh.count, sh.length[0], d_coefs, ws, s);
    return gc::launch_coefs(d_pcm, 0, sh.count, 0, d_coefs, ws, s, &rg);
}
int launch_encode_group(const RaggedShape &sh, const gc::Ragged &rg, const int16_t *d_pcm, const int16_t *d_coefs, const int16_t *h1,
                        const int16_t *h2, uint8_t *d_adpcm, hipStream_t s, void *scratch, size_t scratch_bytes)
{
    if (sh.count <= 0) return VGA_OK;
    if (sh.uniform)
        return gc::launch_encode(d_pcm + sh.pcm_off[0], sh.pcm_pitch, sh.count, sh.length[0], d_coefs, h1, h2, d_adpcm + sh.adpcm_off[0],
                                 sh.adpcm_pitch, s, scratch, scratch_bytes);
    return gc::launch_encode(d_pcm, 0, sh.count, 0, d_coefs, h1, h2, d_adpcm, 0, s, scratch, scratch_bytes, &rg);
}
int launch_decode_group(const RaggedShape &sh, const gc::Ragged &rg, const uint8_t *d_adpcm, const int16_t *d_coefs, const int16_t *h1,
                        const int16_t *h2, int16_t *d_pcm, int *d_status, hipStream_t s)
{
    if (sh.count <= 0) return VGA_OK;
    if (sh.uniform)
        return gc::launch_decode(d_adpcm + sh.adpcm_off[0], sh.adpcm_pitch, d_coefs, sh.count, sh.length[0], h1, h2, d_pcm + sh.pcm_off[0],
                                 sh.pcm_pitch, d_status, s);
    return gc::launch_decode(d_adpcm, 0, d_coefs, sh.count, 0, h1, h2, d_pcm, 0, d_status, s, &rg);
}

}  // namespace
}  // namespace vga

using namespace vga;

// ---------------------------------------------------------------- device-resident ragged batches
struct vga_gcadpcm_ragged {
    RaggedShape shape;
    gc::Ragged view;
    void *d_tables = nullptr;
    int device = 0;
    int64_t pcm_samples = 0, adpcm_bytes = 0;
};

extern "C" {

int vga_gcadpcm_ragged_create(const int *sample_counts, int nch, vga_gcadpcm_ragged **out)
{
    if (!out) { set_error("null output"); return VGA_ERR_ARGUMENT; }
    *out = nullptr;
    if (int rc = check_counts(sample_counts, nch, "vga_gcadpcm_ragged_create")) return rc;
    if (int rc = require_device()) return rc;
    vga_gcadpcm_ragged *r = new vga_gcadpcm_ragged;
    int64_t pcm_end = 0, adpcm_end = 0;
    r->shape.build(sample_counts, nch, 0, 0, &pcm_end, &adpcm_end);
    r->pcm_samples = pcm_end + GUARD_BYTES / 2;
    r->adpcm_bytes = adpcm_end + GUARD_BYTES;
    (void)hipGetDevice(&r->device);
    const size_t tb = r->shape.table_bytes();
    std::vector<unsigned char> host(tb ? tb : 16);
    r->shape.write_tables(host.data());
    hipError_t e = hipMalloc(&r->d_tables, host.size());
    if (e == hipSuccess) e = hipMemcpy(r->d_tables, host.data(), host.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (r->d_tables) (void)hipFree(r->d_tables);
        delete r;
        set_error("ragged tables: %s", hipGetErrorString(e));
        return VGA_ERR_DEVICE;
    }
    r->view = r->shape.device_view(static_cast<const unsigned char *>(r->d_tables));
    *out = r;
    return VGA_OK;
}

void vga_gcadpcm_ragged_destroy(vga_gcadpcm_ragged *r)
{
    if (!r) return;
    if (r->d_tables) (void)hipFree(r->d_tables);
    delete r;
}

int vga_gcadpcm_ragged_channels(const vga_gcadpcm_ragged *r) { return r ? r->shape.count : 0; }
int64_t vga_gcadpcm_ragged_pcm_samples(const vga_gcadpcm_ragged *r) { return r ? r->pcm_samples : 0; }
int64_t vga_gcadpcm_ragged_adpcm_bytes(const vga_gcadpcm_ragged *r) { return r ? r->adpcm_bytes : 0; }
size_t vga_gcadpcm_ragged_coefs_workspace_bytes(const vga_gcadpcm_ragged *r) { return r ? (size_t)std::max<int64_t>(r->shape.records, 1) * 16 : 0; }

int vga_gcadpcm_ragged_offsets(const vga_gcadpcm_ragged *r, int64_t *pcm_offsets_out, int64_t *adpcm_offsets_out)
{
    if (!r) { set_error("null ragged batch"); return VGA_ERR_ARGUMENT; }
    for (int c = 0; c < r->shape.count; c++) {
        if (pcm_offsets_out) pcm_offsets_out[c] = r->shape.pcm_off[c];
        if (adpcm_offsets_out) adpcm_offsets_out[c] = r->shape.adpcm_off[c];
    }
    return VGA_OK;
}

static int check_ragged_call(const vga_gcadpcm_ragged *r, const void *a, const void *b, const char *what)
{
    if (!r) { set_error("%s: null ragged batch", what); return VGA_ERR_ARGUMENT; }
    if (r->shape.count > 0 && (!a || !b)) { set_error("%s: null device buffer", what); return VGA_ERR_ARGUMENT; }
    if (((uintptr_t)a & 15) || ((uintptr_t)b & 15)) { set_error("%s: device buffers must be 16-byte aligned", what); return VGA_ERR_ARGUMENT; }
    int device = -1;
    (void)hipGetDevice(&device);
    if (device != r->device) { set_error("%s: the ragged batch was created on device %d, the current one is %d", what, r->device, device); return VGA_ERR_ARGUMENT; }
    return VGA_OK;
}

int vga_gcadpcm_coefs_device_v(const vga_gcadpcm_ragged *r, const int16_t *d_pcm, int16_t *d_coefs, void *d_workspace,
                               size_t workspace_bytes, void *stream)
{
    if (int rc = check_ragged_call(r, d_pcm, d_coefs, "vga_gcadpcm_coefs_device_v")) return rc;
    if (r->shape.count == 0) return VGA_OK;
    if (!d_workspace || workspace_bytes < vga_gcadpcm_ragged_coefs_workspace_bytes(r)) {
        set_error("workspace too small: need %zu bytes", vga_gcadpcm_ragged_coefs_workspace_bytes(r));
        return VGA_ERR_ARGUMENT;
    }
    return launch_coefs_group(r->shape, r->view, d_pcm, d_coefs, d_workspace, (hipStream_t)stream);
}

int vga_gcadpcm_encode_device_v(const vga_gcadpcm_ragged *r, const int16_t *d_pcm, const int16_t *d_coefs, const int16_t *d_hist1,
                                const int16_t *d_hist2, uint8_t *d_adpcm, void *stream)
{
    if (int rc = check_ragged_call(r, d_pcm, d_adpcm, "vga_gcadpcm_encode_device_v")) return rc;
    if (r->shape.count > 0 && !d_coefs) { set_error("null coefficients"); return VGA_ERR_ARGUMENT; }
    return launch_encode_group(r->shape, r->view, d_pcm, d_coefs, d_hist1, d_hist2, d_adpcm, (hipStream_t)stream, nullptr, 0);
}

int vga_gcadpcm_decode_device_v(const vga_gcadpcm_ragged *r, const uint8_t *d_adpcm, const int16_t *d_coefs, const int16_t *d_hist1,
                                const int16_t *d_hist2, int16_t *d_pcm, int *d_status, void *stream)
{
    if (int rc = check_ragged_call(r, d_adpcm, d_pcm, "vga_gcadpcm_decode_device_v")) return rc;
    if (r->shape.count > 0 && !d_coefs) { set_error("null coefficients"); return VGA_ERR_ARGUMENT; }
    return launch_decode_group(r->shape, r->view, d_adpcm, d_coefs, d_hist1, d_hist2, d_pcm, d_status, (hipStream_t)stream);
}

}  // extern "C"

// ---------------------------------------------------------------- host rows: the pipelined calls
namespace {

// channels per pipeline chunk, by volume: what 1024 channels of BASELINE configs[1] hold (the equal-length entry points'
// chunk); a call below 256 MB of rows is one chunk
constexpr int64_t CHUNK_SAMPLES = (int64_t)1024 * 2880000;
constexpr int GC_MIN_SHARE_CHANNELS = 128;

// One ragged call: the whole batch's rows packed on the device, chunks of channels by volume, every chunk with its own
// shape tables (local channel indices, offsets from the call's buffers).
struct RaggedCall {
    int nch = 0;
    std::vector<size_t> pcm_bytes, pcm_off_bytes, adpcm_bytes, adpcm_off_bytes;   // per channel (caller order)
    std::vector<int> chunk_begin;
    std::vector<RaggedShape> chunks;
    std::vector<gc::Ragged> views;
    DevBuf pcm, adpcm, coefs, h1, h2, tables, status;
    size_t max_pcm_row = 0, max_adpcm_row = 0;
    int64_t max_chunk_records = 1;
    int max_chunk_channels = 1;

    int build(const int *counts, int n)
    {
        nch = n;
        int64_t total = 0;
        for (int c = 0; c < n; c++) total += counts[c];
        const bool small = (size_t)total * 2 < ((size_t)256 << 20);
        const PipeOverride &o = pipe_override();
        chunk_begin.assign(1, 0);
        int64_t acc = 0;
        for (int c = 0; c < n; c++) {
            acc += counts[c];
            const bool cut = o.chunk_units > 0 ? (c + 1 - chunk_begin.back()) >= o.chunk_units : (!small && acc >= CHUNK_SAMPLES);
            if (cut && c + 1 < n) {
                chunk_begin.push_back(c + 1);
                acc = 0;
            }
        }
        // the last chunk once more, into (5/8, 3/8) of its samples: what runs after the last upload is a short chunk's kernels
        // and download (the equal-length entry points do the same: host_batch.hpp, tail_units)
        if (!small && o.chunk_units <= 0 && n - chunk_begin.back() >= 2) {
            const int first = chunk_begin.back();
            int64_t rest = 0, head = 0;
            for (int c = first; c < n; c++) rest += counts[c];
            int cut = first;
            while (cut + 1 < n && head + counts[cut] <= rest * 5 / 8) head += counts[cut++];
            if (cut > first && cut < n) chunk_begin.push_back(cut);
        }
        chunk_begin.push_back(n);
        chunks.resize(chunk_begin.size() - 1);
        int64_t pcm_base = 0, adpcm_base = 0;
        size_t table_total = 0;
        for (size_t k = 0; k + 1 < chunk_begin.size(); k++) {
            chunks[k].build(counts + chunk_begin[k], chunk_begin[k + 1] - chunk_begin[k], pcm_base, adpcm_base, &pcm_base, &adpcm_base);
            table_total += (size_t)round_up((int64_t)chunks[k].table_bytes(), 16);
            max_chunk_records = std::max(max_chunk_records, chunks[k].records);
            max_chunk_channels = std::max(max_chunk_channels, chunks[k].count);
        }
        pcm_bytes.resize(n); pcm_off_bytes.resize(n); adpcm_bytes.resize(n); adpcm_off_bytes.resize(n);
        for (size_t k = 0; k < chunks.size(); k++)
            for (int i = 0; i < chunks[k].count; i++) {
                const int c = chunk_begin[k] + i;
                pcm_bytes[c] = (size_t)counts[c] * 2;
                pcm_off_bytes[c] = (size_t)chunks[k].pcm_off[i] * 2;
                adpcm_bytes[c] = (size_t)vga_gcadpcm_sample_count_to_byte_count(counts[c]);
                adpcm_off_bytes[c] = (size_t)chunks[k].adpcm_off[i];
                max_pcm_row = std::max<size_t>(max_pcm_row, (size_t)round_up((int64_t)pcm_bytes[c], 16));
                max_adpcm_row = std::max<size_t>(max_adpcm_row, (size_t)round_up((int64_t)adpcm_bytes[c], 16));
            }
        VGA_HIP_TRY(pcm.alloc((size_t)pcm_base * 2 + GUARD_BYTES));
        VGA_HIP_TRY(adpcm.alloc((size_t)adpcm_base + GUARD_BYTES));
        VGA_HIP_TRY(coefs.alloc((size_t)std::max(n, 1) * 32));
        VGA_HIP_TRY(tables.alloc(table_total ? table_total : 16));
        std::vector<unsigned char> host(table_total ? table_total : 16);
        size_t at = 0;
        views.resize(chunks.size());
        for (size_t k = 0; k < chunks.size(); k++) {
            chunks[k].write_tables(host.data() + at);
            views[k] = chunks[k].device_view(tables.as<unsigned char>() + at);
            at += (size_t)round_up((int64_t)chunks[k].table_bytes(), 16);
        }
        VGA_HIP_TRY(hipMemcpy(tables.p, host.data(), host.size(), hipMemcpyHostToDevice));
        return VGA_OK;
    }
    int chunk_of(int first) const
    {
        return (int)(std::upper_bound(chunk_begin.begin(), chunk_begin.end(), first) - chunk_begin.begin()) - 1;
    }
    int upload_hist(const int16_t *hist1, const int16_t *hist2)
    {
        if (hist1) {
            VGA_HIP_TRY(h1.alloc((size_t)nch * 2));
            VGA_HIP_TRY(hipMemcpy(h1.p, hist1, (size_t)nch * 2, hipMemcpyHostToDevice));
        }
        if (hist2) {
            VGA_HIP_TRY(h2.alloc((size_t)nch * 2));
            VGA_HIP_TRY(hipMemcpy(h2.p, hist2, (size_t)nch * 2, hipMemcpyHostToDevice));
        }
        return VGA_OK;
    }
};

int encode_batch_v_rows(const int16_t *const *pcm, const int *counts, int nch, const int16_t *hist1, const int16_t *hist2,
                        int16_t *coefs_out, uint8_t *const *adpcm_out, bool with_coefs, const int16_t *coefs_in)
{
    if (int rc = check_counts(counts, nch, "sample_counts")) return rc;
    if (int rc = check_rows((const void *const *)pcm, counts, nch, "pcm")) return rc;
    if (adpcm_out || !with_coefs)
        if (int rc = check_rows((const void *const *)adpcm_out, counts, nch, "adpcm_out")) return rc;
    if (nch > 0 && with_coefs && !coefs_out) { set_error("null coefs_out"); return VGA_ERR_ARGUMENT; }
    if (nch > 0 && !with_coefs && !coefs_in) { set_error("null coefs"); return VGA_ERR_ARGUMENT; }
    if (nch == 0) return VGA_OK;
    if (int rc = require_device()) return rc;
    RaggedCall call;
    if (int rc = call.build(counts, nch)) return rc;
    if (int rc = call.upload_hist(hist1, hist2)) return rc;
    if (!with_coefs) VGA_HIP_TRY(hipMemcpy(call.coefs.p, coefs_in, (size_t)nch * 32, hipMemcpyHostToDevice));
    const bool encode = adpcm_out != nullptr;
    constexpr int LANES = 2;
    DevBuf scratch[LANES], ws[LANES];
    pipe::Job job;
    job.units = nch;
    job.compute_lanes = hardware_queues_requested() >= 6 ? LANES : 1;
    job.chunk_begin = call.chunk_begin;
    job.in_rows = (const void *const *)pcm;
    job.in_row_sizes = call.pcm_bytes.data();
    job.d_in_offsets = call.pcm_off_bytes.data();
    job.in_row_bytes = std::max<size_t>(call.max_pcm_row, 16);
    job.d_in = call.pcm.as<char>();
    job.d_in_pitch = job.in_row_bytes;
    if (encode) {
        job.out_rows = (void *const *)adpcm_out;
        job.out_row_sizes = call.adpcm_bytes.data();
        job.d_out_offsets = call.adpcm_off_bytes.data();
        job.out_row_bytes = std::max<size_t>(call.max_adpcm_row, 16);
        job.d_out = call.adpcm.as<char>();
        job.d_out_pitch = job.out_row_bytes;
    }
    // EncodeChannel (GcAdpcmFormat.cs:129-135) for the chunk's channels: coefficients, then encode
    job.compute = [&](int first, int count, hipStream_t s, std::string &why) -> int {
        const int lane = pipe::compute_lane();
        const int k = call.chunk_of(first);
        const RaggedShape &sh = call.chunks[k];
        int rc = VGA_OK;
        if (sh.count != count) { set_error("internal: chunk %d has %d channels, asked for %d", k, sh.count, count); rc = VGA_ERR_DEVICE; }
        int16_t *d_coefs = call.coefs.as<int16_t>() + (int64_t)first * 16;
        if (!rc && with_coefs) rc = launch_coefs_group(sh, call.views[k], call.pcm.as<int16_t>(), d_coefs, ws[lane].p, s);
        if (!rc && encode)
            rc = launch_encode_group(sh, call.views[k], call.pcm.as<int16_t>(), d_coefs, call.h1.p ? call.h1.as<int16_t>() + first : nullptr,
                                     call.h2.p ? call.h2.as<int16_t>() + first : nullptr, call.adpcm.as<uint8_t>(), s, scratch[lane].p,
                                     scratch[lane].bytes);
        if (rc) why = vga_last_error();
        return rc;
    };
    const int lanes_used = job.compute_lanes;
    for (int l = 0; l < lanes_used; l++) {
        if (encode) VGA_HIP_TRY(scratch[l].alloc(gc::encode_scratch_bytes(call.max_chunk_channels)));
        if (with_coefs) VGA_HIP_TRY(ws[l].alloc((size_t)call.max_chunk_records * 16));
    }
    if (int rc = run_batch_pipeline(job, nch)) return rc;
    if (with_coefs) VGA_HIP_TRY(hipMemcpy(coefs_out, call.coefs.p, (size_t)nch * 32, hipMemcpyDeviceToHost));
    return VGA_OK;
}

int decode_batch_v_rows(const uint8_t *const *adpcm, const int16_t *coefs, const int *counts, int nch, const int16_t *hist1,
                        const int16_t *hist2, int16_t *const *pcm_out)
{
    if (int rc = check_counts(counts, nch, "sample_counts")) return rc;
    if (int rc = check_rows((const void *const *)adpcm, counts, nch, "adpcm")) return rc;
    if (int rc = check_rows((const void *const *)pcm_out, counts, nch, "pcm_out")) return rc;
    if (nch > 0 && !coefs) { set_error("null coefs"); return VGA_ERR_ARGUMENT; }
    if (nch == 0) return VGA_OK;
    if (int rc = require_device()) return rc;
    RaggedCall call;
    if (int rc = call.build(counts, nch)) return rc;
    if (int rc = call.upload_hist(hist1, hist2)) return rc;
    VGA_HIP_TRY(hipMemcpy(call.coefs.p, coefs, (size_t)nch * 32, hipMemcpyHostToDevice));
    VGA_HIP_TRY(call.status.alloc(sizeof(int)));
    VGA_HIP_TRY(hipMemset(call.status.p, 0, sizeof(int)));
    pipe::Job job;
    job.units = nch;
    job.chunk_begin = call.chunk_begin;
    job.in_rows = (const void *const *)adpcm;
    job.in_row_sizes = call.adpcm_bytes.data();
    job.d_in_offsets = call.adpcm_off_bytes.data();
    job.in_row_bytes = std::max<size_t>(call.max_adpcm_row, 16);
    job.d_in = call.adpcm.as<char>();
    job.d_in_pitch = job.in_row_bytes;
    job.out_rows = (void *const *)pcm_out;
    job.out_row_sizes = call.pcm_bytes.data();
    job.d_out_offsets = call.pcm_off_bytes.data();
    job.out_row_bytes = std::max<size_t>(call.max_pcm_row, 16);
    job.d_out = call.pcm.as<char>();
    job.d_out_pitch = job.out_row_bytes;
    job.compute = [&](int first, int count, hipStream_t s, std::string &why) -> int {
        const int k = call.chunk_of(first);
        const RaggedShape &sh = call.chunks[k];
        int rc = VGA_OK;
        if (sh.count != count) { set_error("internal: chunk %d has %d channels, asked for %d", k, sh.count, count); rc = VGA_ERR_DEVICE; }
        if (!rc)
            rc = launch_decode_group(sh, call.views[k], call.adpcm.as<uint8_t>(), call.coefs.as<int16_t>() + (int64_t)first * 16,
                                     call.h1.p ? call.h1.as<int16_t>() + first : nullptr, call.h2.p ? call.h2.as<int16_t>() + first : nullptr,
                                     call.pcm.as<int16_t>(), call.status.as<int>(), s);
        if (rc) why = vga_last_error();
        return rc;
    };
    if (int rc = run_batch_pipeline(job, nch)) return rc;
    int status = 0;
    VGA_HIP_TRY(hipMemcpy(&status, call.status.p, sizeof(int), hipMemcpyDeviceToHost));
    if (status != 0) {
        set_error("a frame header names predictor > 7 (the reference throws IndexOutOfRangeException)");
        return VGA_ERR_ARGUMENT;
    }
    return VGA_OK;
}

// The pipeline works through the rows in order, and what runs after the last upload -- the last chunk's kernels and its
// download -- is the call's tail.  With the caller's (any) order that chunk holds files of every length, and the
// coefficient search of a few hundred channels lasts as long as its LONGEST one (one wave per channel: 31 ms for 120 s):
// the mixed-lengths set of bench.py ended 100 ms after its upload.  The rows are therefore taken longest first (a stable
// sort of pointers; results go back to the caller's rows, coefficients and histories are gathered / scattered here): the
// long files' kernels run under the uploads that follow them and the tail is a chunk of short files.
struct LongestFirst {
    std::vector<int> order;                           // position -> the caller's index
    bool identity = true;
    LongestFirst(const int *counts, int n)
    {
        order.resize(n > 0 ? n : 0);
        for (int i = 0; i < n; i++) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return counts[a] > counts[b]; });
        for (int i = 0; i < n && identity; i++) identity = order[i] == i;
    }
    template <class T> std::vector<T> gather(const T *v) const
    {
        std::vector<T> out(order.size());
        for (size_t i = 0; i < order.size(); i++) out[i] = v[order[i]];
        return out;
    }
    // rows of `width` elements
    template <class T> std::vector<T> gather_rows(const T *v, int width) const
    {
        std::vector<T> out(order.size() * (size_t)width);
        for (size_t i = 0; i < order.size(); i++) std::copy(v + (size_t)order[i] * width, v + (size_t)(order[i] + 1) * width, out.begin() + i * width);
        return out;
    }
};

int encode_batch_v_one(const int16_t *const *pcm, const int *counts, int nch, const int16_t *hist1, const int16_t *hist2,
                       int16_t *coefs_out, uint8_t *const *adpcm_out, bool with_coefs, const int16_t *coefs_in)
{
    // (the argument checks of encode_batch_v_rows, before anything is read through the arrays)
    if (int rc = check_counts(counts, nch, "sample_counts")) return rc;
    if (int rc = check_rows((const void *const *)pcm, counts, nch, "pcm")) return rc;
    if (adpcm_out || !with_coefs)
        if (int rc = check_rows((const void *const *)adpcm_out, counts, nch, "adpcm_out")) return rc;
    if (nch > 0 && with_coefs && !coefs_out) { set_error("null coefs_out"); return VGA_ERR_ARGUMENT; }
    if (nch > 0 && !with_coefs && !coefs_in) { set_error("null coefs"); return VGA_ERR_ARGUMENT; }
    const LongestFirst lf(counts, nch);
    if (nch < 2 || lf.identity) return encode_batch_v_rows(pcm, counts, nch, hist1, hist2, coefs_out, adpcm_out, with_coefs, coefs_in);
    const std::vector<int> n2 = lf.gather(counts);
    const std::vector<const int16_t *> in2 = lf.gather(pcm);
    std::vector<uint8_t *> out2;
    if (adpcm_out) out2 = lf.gather(adpcm_out);
    std::vector<int16_t> h1, h2, cin, cout;
    if (hist1) h1 = lf.gather(hist1);
    if (hist2) h2 = lf.gather(hist2);
    if (!with_coefs) cin = lf.gather_rows(coefs_in, 16);
    if (with_coefs) cout.resize((size_t)nch * 16);
    const int rc = encode_batch_v_rows(in2.data(), n2.data(), nch, hist1 ? h1.data() : nullptr, hist2 ? h2.data() : nullptr,
                                       with_coefs ? cout.data() : nullptr, adpcm_out ? out2.data() : nullptr, with_coefs,
                                       with_coefs ? nullptr : cin.data());
    if (rc == VGA_OK && with_coefs)
        for (int i = 0; i < nch; i++) std::copy(cout.begin() + (size_t)i * 16, cout.begin() + (size_t)(i + 1) * 16, coefs_out + (size_t)lf.order[i] * 16);
    return rc;
}

int decode_batch_v_one(const uint8_t *const *adpcm, const int16_t *coefs, const int *counts, int nch, const int16_t *hist1,
                       const int16_t *hist2, int16_t *const *pcm_out)
{
    if (int rc = check_counts(counts, nch, "sample_counts")) return rc;
    if (int rc = check_rows((const void *const *)adpcm, counts, nch, "adpcm")) return rc;
    if (int rc = check_rows((const void *const *)pcm_out, counts, nch, "pcm_out")) return rc;
    if (nch > 0 && !coefs) { set_error("null coefs"); return VGA_ERR_ARGUMENT; }
    const LongestFirst lf(counts, nch);
    if (nch < 2 || lf.identity) return decode_batch_v_rows(adpcm, coefs, counts, nch, hist1, hist2, pcm_out);
    const std::vector<int> n2 = lf.gather(counts);
    const std::vector<const uint8_t *> in2 = lf.gather(adpcm);
    const std::vector<int16_t *> out2 = lf.gather(pcm_out);
    const std::vector<int16_t> c2 = lf.gather_rows(coefs, 16);
    std::vector<int16_t> h1, h2;
    if (hist1) h1 = lf.gather(hist1);
    if (hist2) h2 = lf.gather(hist2);
    return decode_batch_v_rows(in2.data(), c2.data(), n2.data(), nch, hist1 ? h1.data() : nullptr, hist2 ? h2.data() : nullptr, out2.data());
}

}  // namespace

extern "C" {

int vga_gcadpcm_encode_batch_v(const int16_t *const *pcm, const int *sample_counts, int nch, const int16_t *hist1, const int16_t *hist2,
                               int16_t *coefs_out, uint8_t *const *adpcm_out)
{
    if (nch > 0 && !adpcm_out) { set_error("null adpcm_out"); return VGA_ERR_ARGUMENT; }
    if (nch <= 0 || !pcm || !sample_counts || !coefs_out)
        return encode_batch_v_one(pcm, sample_counts, nch, hist1, hist2, coefs_out, adpcm_out, true, nullptr);
    return for_each_device_share(nch, GC_MIN_SHARE_CHANNELS, [&](int first, int count) {
        return encode_batch_v_one(pcm + first, sample_counts + first, count, hist1 ? hist1 + first : nullptr, hist2 ? hist2 + first : nullptr,
                                  coefs_out + (size_t)first * 16, adpcm_out + first, true, nullptr);
    });
}

int vga_gcadpcm_calculate_coefficients_batch_v(const int16_t *const *pcm, const int *lengths, int nch, int16_t *coefs_out)
{
    if (nch <= 0 || !pcm || !lengths || !coefs_out) return encode_batch_v_one(pcm, lengths, nch, nullptr, nullptr, coefs_out, nullptr, true, nullptr);
    return for_each_device_share(nch, GC_MIN_SHARE_CHANNELS, [&](int first, int count) {
        return encode_batch_v_one(pcm + first, lengths + first, count, nullptr, nullptr, coefs_out + (size_t)first * 16, nullptr, true, nullptr);
    });
}

int vga_gcadpcm_encode_with_coefs_batch_v(const int16_t *const *pcm, const int *sample_counts, int nch, const int16_t *coefs,
                                          const int16_t *hist1, const int16_t *hist2, uint8_t *const *adpcm_out)
{
    if (nch > 0 && !adpcm_out) { set_error("null adpcm_out"); return VGA_ERR_ARGUMENT; }
    if (nch <= 0 || !pcm || !sample_counts || !coefs)
        return encode_batch_v_one(pcm, sample_counts, nch, hist1, hist2, nullptr, adpcm_out, false, coefs);
    return for_each_device_share(nch, GC_MIN_SHARE_CHANNELS, [&](int first, int count) {
        return encode_batch_v_one(pcm + first, sample_counts + first, count, hist1 ? hist1 + first : nullptr, hist2 ? hist2 + first : nullptr,
                                  nullptr, adpcm_out + first, false, coefs + (size_t)first * 16);
    });
}

int vga_gcadpcm_decode_batch_v(const uint8_t *const *adpcm, const int16_t *coefs, const int *sample_counts, int nch, const int16_t *hist1,
                               const int16_t *hist2, int16_t *const *pcm_out)
{
    if (nch <= 0 || !adpcm || !coefs || !sample_counts || !pcm_out) return decode_batch_v_one(adpcm, coefs, sample_counts, nch, hist1, hist2, pcm_out);
    return for_each_device_share(nch, GC_MIN_SHARE_CHANNELS, [&](int first, int count) {
        return decode_batch_v_one(adpcm + first, coefs + (size_t)first * 16, sample_counts + first, count, hist1 ? hist1 + first : nullptr,
                                  hist2 ? hist2 + first : nullptr, pcm_out + first);
    });
}

}  // extern "C"
