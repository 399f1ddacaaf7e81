// hca_info.hpp -- plain (no HIP) descriptions of an HCA stream shared by the kernels, the C ABI and the host-side lane
// emulators of the CPU test-suite.
#pragma once
#include <cstdint>

namespace vga {
namespace hca {

constexpr int SUBFRAMES = 8;
constexpr int SPSF = 128;     // SamplesPerSubFrame
constexpr int SPF = 1024;     // SamplesPerFrame

enum { CH_DISCRETE = 0, CH_STEREO_PRIMARY = 1, CH_STEREO_SECONDARY = 2 };

// HcaInfo subset + derived per-channel layout, passed by value to the kernels
struct DeviceInfo {
    int nch, frame_size, frame_count, sample_count, inserted_samples;
    int total_band_count, base_band_count, stereo_band_count, hfr_band_count, bands_per_hfr_group, hfr_group_count;
    int channel_type[8];
    int coded_count[8];
    uint8_t ath_curve[128];
};

}  // namespace hca
}  // namespace vga
