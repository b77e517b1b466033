// roundedcorners mask rendering through the system libcairo (see cairo_mask.cpp).
#pragma once

#include <cstdint>

namespace mvfx {

// Renders the A8 mask of border/imp.rs:108-180 into `mask` (stride x round_up_2(height) bytes of
// host memory).  Returns an mvfx_status; MVFX_ERR_IO when libcairo cannot be loaded.
int cairo_render_rounded_mask(uint8_t *mask, uint32_t width, uint32_t height, uint32_t stride, uint32_t border_radius_px);

// cairo_version_string() of the library in use, or nullptr when it cannot be loaded.
const char *cairo_mask_library_version();

} // namespace mvfx
