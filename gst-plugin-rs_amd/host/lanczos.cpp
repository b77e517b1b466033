// Lanczos3 tap tables (see lanczos.h).  Every expression is f32 and evaluated in the order image 0.25 evaluates
// it (centre = (o + 0.5) * ratio; window = centre -/+ 3 * max(ratio, 1); weights normalised by their running f32
// sum), because the resized u8 image -- and with it the hash bits of videocompare's Mean / Gradient / VertGradient /
// DoubleGradient algorithms (hashed_image.rs:89-107) -- depends on each rounding.  Built with -ffp-contract=off.
#include "lanczos.h"

#include <algorithm>
#include <cmath>

namespace mvfx {

namespace {

inline float sinc_pi(float t)
{
    const float a = t * 3.14159274101257324f; // f32::consts::PI
    return t == 0.0f ? 1.0f : std::sin(a) / a;
}

inline float lanczos3(float x) { return std::fabs(x) < 3.0f ? sinc_pi(x) * sinc_pi(x / 3.0f) : 0.0f; }

} // namespace

LanczosAxis lanczos3_axis(uint32_t in_size, uint32_t out_size)
{
    LanczosAxis ax;
    ax.left.resize(out_size);
    ax.count.resize(out_size);
    ax.offset.resize(out_size);
    const float ratio = static_cast<float>(in_size) / static_cast<float>(out_size);
    const float sratio = ratio < 1.0f ? 1.0f : ratio;
    const float support = 3.0f * sratio;
    for (uint32_t o = 0; o < out_size; o++) {
        float centre = (static_cast<float>(o) + 0.5f) * ratio;
        const int64_t lo = std::clamp<int64_t>(static_cast<int64_t>(std::floor(centre - support)), 0, static_cast<int64_t>(in_size) - 1);
        const int64_t hi = std::clamp<int64_t>(static_cast<int64_t>(std::ceil(centre + support)), lo + 1, static_cast<int64_t>(in_size));
        centre -= 0.5f;
        ax.left[o] = static_cast<uint32_t>(lo);
        ax.count[o] = static_cast<uint32_t>(hi - lo);
        ax.offset[o] = static_cast<uint32_t>(ax.weights.size());
        float total = 0.0f;
        for (int64_t i = lo; i < hi; i++) {
            const float w = lanczos3((static_cast<float>(i) - centre) / sratio);
            ax.weights.push_back(w);
            total += w;
        }
        for (size_t k = ax.offset[o]; k < ax.weights.size(); k++)
            ax.weights[k] /= total;
    }
    return ax;
}

} // namespace mvfx
