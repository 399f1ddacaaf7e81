// Drives vgaudio_amd/csrc/host_pipeline.hpp against tests/host/mockhip (asynchronous mock streams) on the CPU:
// data integrity over ragged shapes, error propagation without hangs, pinned-pool reuse.  Built with
// -fsanitize=thread by tests/test_host_pipeline.py.
#include "../../vgaudio_amd/csrc/host_pipeline.hpp"

#include <atomic>
#include <cstdio>
#include <random>

using namespace vga::pipe;

static int g_taper = 0, g_tail = 0, g_head = 0;
static bool g_direct = false, g_direct_out = false, g_shared = false;
static int g_lanes = 1;
// round 6: direct transfers by a "kernel" (Job::transfer: a closure on the mock stream that copies the pieces) instead of one
// copy per row, with and without CU masks; g_transfer_fails: the n-th transfer launch from now refuses
static bool g_transfer = false;
static int g_transfer_cus = 0;
static std::atomic<int> g_transfer_launches{0}, g_transfer_pieces{0}, g_transfer_fails{-1};
static void use_transfer(Job &job, int feeders)
{
    if (!g_transfer) return;
    job.gather_in = true;
    job.scatter_out = true;
    job.piece_bytes = 4096;                                  // (the floor: rows of a few KB still come in several pieces)
    job.transfer_cus = g_transfer_cus;
    job.total_cus = 64;
    if (job.direct) job.feeders = 1;                         // (the gather is the one feeder's; more feeders: plain copies)
    (void)feeders;
    job.transfer = [](const Job::TransferPiece *pieces, int n, hipStream_t s, std::string &why) -> int {
        if (g_transfer_fails.load() == 0) {
            g_transfer_fails = -1;
            why = "transfer refused";
            return -9;
        }
        if (g_transfer_fails.load() > 0) g_transfer_fails--;
        g_transfer_launches++;
        g_transfer_pieces += n;
        mockLaunch(s, [pieces, n] {
            for (int i = 0; i < n; i++) std::memcpy(pieces[i].dst, pieces[i].src, pieces[i].bytes);
        });
        return 0;
    };
}
static int run_case(int units, int in_rpu, int out_rpu, size_t in_bytes, size_t out_bytes, int chunk, int feeders, int drainers,
                    size_t slot_bytes, int delay_us, int fail_after, bool compute_fails)
{
    const size_t in_pitch = (in_bytes + 15) / 16 * 16, out_pitch = (out_bytes + 15) / 16 * 16;
    const int in_rows = units * in_rpu, out_rows = units * out_rpu;
    std::vector<std::vector<unsigned char>> in(in_rows, std::vector<unsigned char>(in_bytes)), out(out_rows, std::vector<unsigned char>(out_bytes, 0xEE));
    std::mt19937 rng(units * 131 + chunk);
    for (auto &r : in) for (auto &b : r) b = (unsigned char)rng();
    std::vector<const void *> in_ptrs(in_rows);
    std::vector<void *> out_ptrs(out_rows);
    for (int r = 0; r < in_rows; r++) in_ptrs[r] = in[r].data();
    for (int r = 0; r < out_rows; r++) out_ptrs[r] = out[r].data();
    std::vector<char> d_in((size_t)in_rows * in_pitch + 64, 0x11), d_out((size_t)out_rows * out_pitch + 64, 0x22);

    Job job;
    job.units = units;
    job.chunk_units = chunk;
    job.in_rows_per_unit = in_rpu;
    job.in_rows = in_ptrs.data();
    job.in_row_bytes = in_bytes;
    job.d_in = d_in.data();
    job.d_in_pitch = in_pitch;
    job.out_rows_per_unit = out_rpu;
    job.out_rows = out_ptrs.data();
    job.out_row_bytes = out_bytes;
    job.d_out = d_out.data();
    job.d_out_pitch = out_pitch;
    job.feeders = feeders;
    job.drainers = drainers;
    job.slot_bytes = slot_bytes;
    job.taper_min_units = g_taper;
    job.tail_units = g_tail;
    job.head_units = g_head;
    job.direct = g_direct;
    job.direct_out = g_direct_out;
    job.shared_streams = g_shared;
    job.compute_lanes = g_lanes;
    use_transfer(job, feeders);
    if (in_rows == 0) job.in_rows = nullptr;                 // a job with nothing to upload / nothing to download
    if (out_rows == 0) job.out_rows = nullptr;
    int launches = 0;
    std::vector<int> seen(units, 0);
    // "kernel": output row (u, j) byte i = sum over the unit's input rows of byte (i mod in_bytes), plus j and i
    job.compute = [&](int first, int count, hipStream_t s, std::string &why) -> int {
        launches++;
        for (int u = first; u < first + count; u++) seen[u]++;
        if (compute_fails && first > 0) {
            why = "compute refused";
            return -3;
        }
        mockLaunch(s, [&, first, count] {
            for (int u = first; u < first + count; u++)
                for (int j = 0; j < out_rpu; j++)
                    for (size_t i = 0; i < out_bytes; i++) {
                        unsigned v = (unsigned)j + (unsigned)i;
                        for (int k = 0; k < in_rpu; k++) v += (unsigned char)d_in[(size_t)(u * in_rpu + k) * in_pitch + i % in_bytes];
                        d_out[(size_t)(u * out_rpu + j) * out_pitch + i] = (char)v;
                    }
        });
        return 0;
    };
    // progress: chunks are reported once each, in order, one at a time, and when a chunk is reported its output rows are
    // already complete in the caller's memory
    auto expected = [&](int u, int j, size_t i) {
        unsigned v = (unsigned)j + (unsigned)i;
        for (int k = 0; k < in_rpu; k++) v += in[u * in_rpu + k][i % in_bytes];
        return (unsigned char)v;
    };
    int reported = 0, report_calls = 0, report_bad = 0;
    std::atomic<int> in_report{0};
    job.chunk_done = [&](int first, int count) {
        if (in_report.fetch_add(1) != 0) report_bad++;
        if (first != reported || count <= 0 || first + count > units) report_bad++;
        for (int u = first; u < first + count && u < units; u++)
            for (int j = 0; j < out_rpu; j++)
                for (size_t i = 0; i < out_bytes; i++)
                    if (out[u * out_rpu + j][i] != expected(u, j, i)) { report_bad++; break; }
        reported = first + count;
        report_calls++;
        in_report.fetch_sub(1);
    };
    mock_copy_delay_us() = delay_us;
    mock_fail_memcpy_after() = fail_after;
    const Result r = run(job);
    mock_fail_memcpy_after() = -1;
    mock_copy_delay_us() = 0;
    if (fail_after >= 0 || compute_fails) {
        if (r.code == 0) { std::printf("expected a failure (units %d)\n", units); return 1; }
        if (report_bad || reported > units) { std::printf("progress after a failure: %d bad reports\n", report_bad); return 1; }
        return 0;
    }
    if (r.code != 0) { std::printf("unexpected failure %d: %s\n", r.code, r.why.c_str()); return 1; }
    if (report_bad || reported != units || report_calls != r.stats.chunks) {
        std::printf("progress: %d bad reports, %d of %d units in %d calls for %d chunks\n", report_bad, reported, units, report_calls, r.stats.chunks);
        return 1;
    }
    for (int u = 0; u < units; u++)
        if (seen[u] != 1) { std::printf("unit %d computed %d times\n", u, seen[u]); return 1; }
    for (int u = 0; u < units; u++)
        for (int j = 0; j < out_rpu; j++)
            for (size_t i = 0; i < out_bytes; i++) {
                unsigned v = (unsigned)j + (unsigned)i;
                for (int k = 0; k < in_rpu; k++) v += in[u * in_rpu + k][i % in_bytes];
                if (out[u * out_rpu + j][i] != (unsigned char)v) {
                    std::printf("mismatch: units %d chunk %d unit %d row %d byte %zu\n", units, chunk, u, j, i);
                    return 1;
                }
            }
    return 0;
}

// Ragged jobs (the `*_v` entry points): every row its own size (some empty), packed on the "device" with 16-byte alignment,
// chunks cut by explicit unit boundaries.  Output row u: byte i = input byte (i mod size) of row u + i.
static int run_ragged_case(int units, int seed, int chunks_wanted, int feeders, int drainers, size_t slot_bytes, int delay_us, int fail_after)
{
    std::mt19937 rng(seed);
    std::vector<size_t> in_size(units), out_size(units), in_off(units), out_off(units);
    size_t ia = 0, oa = 0, in_max = 16, out_max = 16;
    for (int u = 0; u < units; u++) {
        const unsigned pick = rng() % 10;
        in_size[u] = pick == 0 ? 0 : (pick < 7 ? 1 + rng() % 300 : 500 + rng() % 4000);
        out_size[u] = in_size[u] == 0 ? 0 : (in_size[u] * 9 + 31) / 32;
        in_off[u] = ia;
        out_off[u] = oa;
        ia += (in_size[u] + 15) / 16 * 16;
        oa += (out_size[u] + 15) / 16 * 16;
        in_max = std::max(in_max, (in_size[u] + 15) / 16 * 16);
        out_max = std::max(out_max, (out_size[u] + 15) / 16 * 16);
    }
    std::vector<std::vector<unsigned char>> in(units), out(units);
    for (int u = 0; u < units; u++) {
        in[u].resize(in_size[u] + 1);                          // (+1: a valid pointer for empty rows)
        out[u].assign(out_size[u] + 1, 0xEE);
        for (auto &b : in[u]) b = (unsigned char)rng();
    }
    std::vector<const void *> in_ptrs(units);
    std::vector<void *> out_ptrs(units);
    for (int u = 0; u < units; u++) { in_ptrs[u] = in[u].data(); out_ptrs[u] = out[u].data(); }
    std::vector<char> d_in(ia + 64, 0x11), d_out(oa + 64, 0x22);
    Job job;
    job.units = units;
    job.in_rows = in_ptrs.data();
    job.in_row_bytes = in_max;
    job.in_row_sizes = in_size.data();
    job.d_in = d_in.data();
    job.d_in_pitch = in_max;
    job.d_in_offsets = in_off.data();
    job.out_rows = out_ptrs.data();
    job.out_row_bytes = out_max;
    job.out_row_sizes = out_size.data();
    job.d_out = d_out.data();
    job.d_out_pitch = out_max;
    job.d_out_offsets = out_off.data();
    job.feeders = feeders;
    job.drainers = drainers;
    job.slot_bytes = slot_bytes;
    job.direct = g_direct;
    job.direct_out = g_direct_out;
    job.shared_streams = g_shared;
    job.compute_lanes = g_lanes;
    use_transfer(job, feeders);
    // chunks by bytes: a boundary whenever a chunk holds its share of the input
    job.chunk_begin.push_back(0);
    size_t acc = 0;
    for (int u = 0; u < units; u++) {
        acc += in_size[u];
        if (acc >= ia / std::max(chunks_wanted, 1) + 1 && u + 1 < units) { job.chunk_begin.push_back(u + 1); acc = 0; }
    }
    job.chunk_begin.push_back(units);
    std::vector<int> seen(units, 0);
    job.compute = [&](int first, int count, hipStream_t s, std::string &) -> int {
        for (int u = first; u < first + count; u++) seen[u]++;
        mockLaunch(s, [&, first, count] {
            for (int u = first; u < first + count; u++)
                for (size_t i = 0; i < out_size[u]; i++)
                    d_out[out_off[u] + i] = (char)((unsigned char)d_in[in_off[u] + i % in_size[u]] + (unsigned)i);
        });
        return 0;
    };
    int reported = 0, report_bad = 0;
    job.chunk_done = [&](int first, int count) {
        if (first != reported) report_bad++;
        for (int u = first; u < first + count; u++)
            for (size_t i = 0; i < out_size[u]; i++)
                if (out[u][i] != (unsigned char)(in[u][i % in_size[u]] + (unsigned)i)) { report_bad++; break; }
        reported = first + count;
    };
    mock_copy_delay_us() = delay_us;
    mock_fail_memcpy_after() = fail_after;
    const Result r = run(job);
    mock_fail_memcpy_after() = -1;
    mock_copy_delay_us() = 0;
    if (fail_after >= 0) {
        if (r.code == 0) { std::printf("ragged: expected a failure\n"); return 1; }
        return report_bad ? 1 : 0;
    }
    if (r.code != 0) { std::printf("ragged: unexpected failure %d: %s\n", r.code, r.why.c_str()); return 1; }
    if (report_bad || reported != units || r.stats.chunks != (int)job.chunk_begin.size() - 1) {
        std::printf("ragged progress: %d bad, %d of %d units, %d chunks\n", report_bad, reported, units, r.stats.chunks);
        return 1;
    }
    for (int u = 0; u < units; u++) {
        if (seen[u] != 1) { std::printf("ragged: unit %d computed %d times\n", u, seen[u]); return 1; }
        for (size_t i = 0; i < out_size[u]; i++)
            if (out[u][i] != (unsigned char)(in[u][i % in_size[u]] + (unsigned)i)) {
                std::printf("ragged mismatch: units %d seed %d unit %d byte %zu\n", units, seed, u, i);
                return 1;
            }
        if (out[u][out_size[u]] != 0xEE) { std::printf("ragged: wrote past the end of row %d\n", u); return 1; }
    }
    return 0;
}

static int all_cases();

// run_on_devices: the units of one call cut into shares, every share a whole pipeline on its own thread with its own
// device current; here two or three shares run at once (the mock has four devices), one of them may fail.
static int share_cases()
{
    int bad = 0;
    {
        const auto plan = plan_shares({0, 1, 2, 3}, 10, 4);       // shares of at least 4 units: two of them
        if (plan.size() != 2 || plan[0].count != 5 || plan[1].first != 5 || plan[1].device != 1) { std::printf("plan_shares (min units)\n"); bad++; }
        const auto all = plan_shares({2, 2, 0}, 7, 0);
        if (all.size() != 3 || all[0].count != 3 || all[1].count != 2 || all[2].first != 5 || all[2].device != 0) { std::printf("plan_shares (split)\n"); bad++; }
        if (!plan_shares({}, 5, 0).empty() || !plan_shares({0}, 0, 0).empty()) { std::printf("plan_shares (empty)\n"); bad++; }
    }
    for (int fail_share : {-1, 1}) {
        std::mutex m;
        std::vector<int> devices_seen;
        const auto shares = plan_shares({3, 1, 2}, 96, 8);
        (void)hipSetDevice(0);
        const Result r = run_on_devices(shares, [&](const Share &sh, std::string &why) -> int {
            int dev = -1;
            (void)hipGetDevice(&dev);
            {
                std::lock_guard<std::mutex> g(m);
                devices_seen.push_back(dev * 1000 + sh.device);
            }
            if (sh.index == fail_share) {
                why = "share refused";
                return -7;
            }
            // every share is a whole pipelined call over its own units
            return run_case(sh.count, 1, 1, 700 + sh.index, 90, 8, 2, 2, 2048, 10, -1, false) ? -1 : 0;
        });
        int dev = -1;
        (void)hipGetDevice(&dev);
        if (dev != 0) { std::printf("the caller's device changed to %d\n", dev); bad++; }
        if (fail_share < 0 && r.code != 0) { std::printf("shares failed: %d %s\n", r.code, r.why.c_str()); bad++; }
        if (fail_share >= 0 && (r.code != -7 || r.why != "share refused")) { std::printf("failing share not reported: %d\n", r.code); bad++; }
        std::sort(devices_seen.begin(), devices_seen.end());
        // share 0 runs on the calling thread, whose device is not touched; the others run with their device current
        if (devices_seen != std::vector<int>{3, 1001, 2002} && devices_seen != std::vector<int>{1001, 2002, 3003}) {
            std::printf("shares saw the wrong devices:");
            for (int d : devices_seen) std::printf(" %d", d);
            std::printf("\n");
            bad++;
        }
    }
    return bad;
}

int main()
{
    int bad = 0;
    bad += share_cases();
    // direct uploads, direct downloads, shared copy streams, compute lanes
    const int modes[][4] = {{0, 0, 0, 1}, {1, 1, 0, 1}, {0, 0, 1, 1}, {1, 1, 1, 3}, {1, 0, 1, 2}, {0, 1, 0, 3}};
    for (const auto &m : modes) {
        g_direct = m[0];
        g_direct_out = m[1];
        g_shared = m[2];
        g_lanes = m[3];
        bad += all_cases();
    }
    // the same shapes with the rows moved by transfer launches (direct modes only use them), CU masks on and off, and a
    // transfer launch that refuses
    g_transfer = true;
    for (int cus : {0, 8}) {
        g_transfer_cus = cus;
        for (const auto &m : modes) {
            g_direct = m[0];
            g_direct_out = m[1];
            g_shared = m[2];
            g_lanes = m[3];
            bad += all_cases();
        }
    }
    if (g_transfer_launches.load() == 0 || g_transfer_pieces.load() <= g_transfer_launches.load()) { std::printf("no transfer launches\n"); bad++; }
    if (mock_masked_streams() == 0) { std::printf("no CU-masked streams\n"); bad++; }
    g_direct = g_direct_out = g_shared = true;
    for (int nth : {0, 2}) {
        g_transfer_fails = nth;
        Job probe;                                           // (run_case reports "expected a failure" through fail_after / compute_fails only)
        std::vector<std::vector<unsigned char>> in(12, std::vector<unsigned char>(9000, 7)), out(12, std::vector<unsigned char>(9000, 0));
        std::vector<const void *> ip(12);
        std::vector<void *> op(12);
        for (int r = 0; r < 12; r++) { ip[r] = in[r].data(); op[r] = out[r].data(); }
        std::vector<char> d_in(12 * 9008 + 64), d_out(12 * 9008 + 64);
        probe.units = 12; probe.chunk_units = 4; probe.in_rows = ip.data(); probe.in_row_bytes = 9000; probe.d_in = d_in.data(); probe.d_in_pitch = 9008;
        probe.out_rows = op.data(); probe.out_row_bytes = 9000; probe.d_out = d_out.data(); probe.d_out_pitch = 9008;
        probe.direct = probe.direct_out = probe.shared_streams = true;
        probe.feeders = 1; probe.drainers = 2;
        probe.compute = [&](int first, int count, hipStream_t s, std::string &) -> int {
            mockLaunch(s, [&, first, count] { std::memcpy(d_out.data() + (size_t)first * 9008, d_in.data() + (size_t)first * 9008, (size_t)count * 9008); });
            return 0;
        };
        use_transfer(probe, 1);
        const Result r = run(probe);
        if (r.code != -9 || r.why != "transfer refused") { std::printf("a refused transfer launch was not reported: %d %s\n", r.code, r.why.c_str()); bad++; }
    }
    g_transfer_fails = -1;
    g_transfer = false;
    MaskedStreamPool::get().trim();
    PinnedPool::get().trim();
    if (mock_registered() != 0) { std::printf("%d rows left registered\n", mock_registered()); bad++; }
    if (mock_unregister_violations() != 0) { std::printf("%d rows unregistered with a copy still pending\n", mock_unregister_violations()); bad++; }
    std::printf(bad ? "FAILED %d\n" : "ok\n", bad);
    return bad ? 1 : 0;
}

static int all_cases()
{
    int bad = 0;
    // units, in_rpu, out_rpu, in_bytes, out_bytes, chunk, feeders, drainers, slot_bytes, delay, fail_after, compute_fails
    bad += run_case(1, 1, 1, 100, 40, 0, 8, 4, 1 << 20, 0, -1, false);
    bad += run_case(37, 1, 1, 1000, 300, 8, 8, 4, 2048, 20, -1, false);           // ragged last chunk, several rows per slot
    bad += run_case(64, 2, 1, 513, 77, 16, 3, 2, 512, 50, -1, false);             // stream-like units, one row per slot
    bad += run_case(5, 1, 3, 64, 64, 1, 8, 8, 1 << 20, 10, -1, false);            // more workers than rows
    bad += run_case(200, 1, 1, 4096, 1200, 50, 8, 4, 16384, 5, -1, false);
    bad += run_case(33, 1, 1, 256, 64, 4, 4, 2, 1024, 30, 7, false);              // a copy fails mid-way: error, no hang
    bad += run_case(33, 1, 1, 256, 64, 4, 4, 2, 1024, 30, 0, false);
    bad += run_case(40, 1, 1, 256, 64, 8, 4, 2, 1024, 10, -1, true);              // the compute callback refuses the 2nd chunk
    bad += run_case(21, 1, 0, 300, 16, 4, 3, 2, 1024, 10, -1, false);             // upload only (coefficient search)
    bad += run_case(21, 0, 2, 16, 300, 4, 3, 2, 1024, 10, -1, false);             // download only
    g_taper = 3;                                                                  // tapered last chunk
    bad += run_case(37, 1, 1, 1000, 300, 8, 8, 4, 2048, 20, -1, false);
    bad += run_case(64, 2, 1, 513, 77, 32, 3, 2, 512, 10, -1, false);
    bad += run_case(100, 1, 1, 128, 32, 40, 4, 2, 4096, 5, -1, false);
    bad += run_case(9, 1, 1, 128, 32, 100, 4, 2, 4096, 5, -1, false);             // one chunk: nothing to taper
    g_taper = 0;
    g_tail = 3;                                                                   // last chunk split once: (rest, tail)
    bad += run_case(37, 1, 1, 1000, 300, 8, 8, 4, 2048, 20, -1, false);
    bad += run_case(64, 2, 1, 513, 77, 32, 3, 2, 512, 10, -1, false);
    bad += run_case(9, 1, 1, 128, 32, 100, 4, 2, 4096, 5, -1, false);             // one chunk: nothing to split
    bad += run_case(10, 1, 1, 128, 32, 8, 4, 2, 4096, 5, -1, false);              // last chunk shorter than the tail
    g_tail = 0;
    g_head = 2;                                                                   // short first chunk
    bad += run_case(37, 1, 1, 1000, 300, 8, 8, 4, 2048, 20, -1, false);
    bad += run_case(10, 1, 1, 128, 32, 8, 4, 2, 4096, 5, -1, false);              // too few units: no head chunk
    g_tail = 3;
    bad += run_case(64, 2, 1, 513, 77, 16, 3, 2, 512, 10, -1, false);             // head and tail together
    g_tail = 0;
    g_head = 0;
    // ragged rows, chunks by bytes
    bad += run_ragged_case(1, 5, 1, 2, 2, 4096, 0, -1);
    bad += run_ragged_case(57, 6, 4, 3, 2, 2048, 10, -1);                         // several rows per slot, rows larger than a slot
    bad += run_ragged_case(200, 7, 7, 4, 3, 8192, 5, -1);
    bad += run_ragged_case(33, 8, 1, 1, 1, 512, 5, -1);                           // one chunk
    bad += run_ragged_case(64, 9, 5, 3, 2, 2048, 10, 9);                          // a copy fails mid-way
    return bad;
}
