// Adobe .cube parser of the colorlut element (host side, no GPU needed).
// Mirrors video/colorlut/src/parser.rs: CubeLut { domain_scale, domain_offset, kind }.
#pragma once

#include <cstdint>
#include <string>
#include <string_view>
#include <vector>

namespace mvfx {

struct CubeLut {
    float domain_scale[3] = {1.0f, 1.0f, 1.0f};  // parser.rs:264-268
    float domain_offset[3] = {0.0f, 0.0f, 0.0f}; // parser.rs:270-274
    float domain_min[3] = {0.0f, 0.0f, 0.0f};    // as parsed (DOMAIN_MIN / DOMAIN_MAX), kept for write_cube
    float domain_max[3] = {1.0f, 1.0f, 1.0f};
    bool is_3d = false;
    uint32_t size = 0;
    std::vector<float> rgba;     // 3-D: size^3 x [r,g,b,1.0], R fastest (parser.rs:43-53, 253-256)
    std::vector<float> table[3]; // 1-D: r, g, b (parser.rs:226-236)
};

// parser.rs:110-282. Returns true and fills `out`, or false with a message in `error`
// (CubeParseError::InvalidLut).
bool parse_cube(std::string_view text, CubeLut &out, std::string &error);

// parser.rs:105-108 (fs::read_to_string + parse). io_error is set when the file cannot be
// read or is not valid UTF-8 (CubeParseError::Io).
bool parse_cube_file(const char *path, CubeLut &out, std::string &error, bool &io_error);

// Adobe .cube text of `lut`; parse_cube(write_cube(lut)) reproduces every float bit for bit (9 significant digits).
std::string write_cube(const CubeLut &lut);

} // namespace mvfx
