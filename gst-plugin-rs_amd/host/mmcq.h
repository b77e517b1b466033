// Host half of colordetect: modified-median-cut quantisation over the 32768-bin histogram that
// the HIP kernel produced, and the nearest-CSS-colour lookup.
//
// The reference calls color_thief::get_palette (crate color-thief 0.2.2, not under
// /root/reference; call site video/videofx/src/colordetect/imp.rs:68-74) and
// color_name::css::Color::similar (color-name 1.2.0; call site :77-79).  What is implemented here
// is the published MMCQ algorithm (Leptonica -> quantize.js -> color-thief) as described in
// SURVEY.md Appendix A.1.  Parity with the crates is pinned only by the reference's own test
// (solid red => "red", tests/colordetect.rs:21-68); everything else is "parity unpinned".
#pragma once

#include <cstdint>
#include <vector>

namespace mvfx {

constexpr int kHistBins = 32768; // 5 bits per channel

struct Rgb8 {
    uint8_t r, g, b;
};

// hist: 32768 counts indexed (r5<<10)|(g5<<5)|b5; minmax = {rmin,rmax,gmin,gmax,bmin,bmax} of the
// 5-bit channel values of the counted samples.  Returns the palette, most dominant first, at
// most max_colors entries; empty on invalid arguments (max_colors < 2 or > 255).
std::vector<Rgb8> mmcq_palette(const uint32_t *hist, const uint32_t minmax[6], uint32_t max_colors);

// Lower-case CSS colour name nearest to (r,g,b) (squared Euclidean distance in RGB).
const char *css_color_similar(uint8_t r, uint8_t g, uint8_t b);

} // namespace mvfx
