// Modified median cut quantisation (MMCQ) on a 5-5-5 histogram + CSS colour naming.
// See mmcq.h for provenance and pinning status.
#include "mmcq.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace mvfx {
namespace {

constexpr int kShift = 3;            // 8 - SIGNAL_BITS
constexpr int kMult = 1 << kShift;   // 8
constexpr double kFractionByPopulation = 0.75;
constexpr int kMaxIterations = 1000;

inline int bin_of(int r, int g, int b) { return (r << 10) | (g << 5) | b; }

// One occupied bin of the histogram.  The boxes of the median cut only ever ask for sums over their bins, and an empty bin adds
// nothing to any of them: every box owns a slice [begin, end) of ONE array of the occupied bins (partitioned in place when a box is
// cut, like a quicksort), so a sum costs the bins that hold pixels, not the volume of the box -- a 4K frame of flat bars occupies 38
// of the 32768 bins, a natural-like one ~4000: host time per 4K frame (quality 10, 5 colours) 281 -> 10 us and 178 -> 66 us; a
// uniform-random frame, every bin occupied, 399 -> 437 us (round 3; tests/test_mmcq_host_cpu.py pins the palettes to the oracle's).
struct Entry {
    uint8_t c[3]; // r, g, b bin coordinates (0..31)
    uint32_t h;   // the bin's count
};

struct Box {
    int lo[3];
    int hi[3]; // inclusive; hi < lo encodes an empty box
    int count = 0;
    int volume = 0;
    Rgb8 average{0, 0, 0};
    int begin = 0, end = 0; // this box's occupied bins: entries[begin .. end)

    // Recomputes the cached population, volume and population-weighted mean colour.
    void refresh(const Entry *entries)
    {
        // i32 accumulators as in the crate; a release build of the reference wraps on overflow (first possible at
        // ~8.5 M samples in one box: 8K frames at quality <= 3), so they are carried as u32 and read back as i32
        uint32_t total_u = 0, n_u = 0, sum_u[3] = {0, 0, 0};
        for (int i = begin; i < end; i++) {
            const Entry &e = entries[i];
            if (e.h <= 8000000u) { // h (8 c + 4) < 2^31: the crate's f64 product and its cast to i32 are exact, plain integers say the same
                n_u += e.h;
                sum_u[0] += e.h * (8u * e.c[0] + 4u);
                sum_u[1] += e.h * (8u * e.c[1] + 4u);
                sum_u[2] += e.h * (8u * e.c[2] + 4u);
            } else {
                const double h = static_cast<double>(static_cast<int32_t>(e.h));
                n_u += static_cast<uint32_t>(static_cast<int32_t>(h));
                sum_u[0] += static_cast<uint32_t>(static_cast<int32_t>(h * (e.c[0] + 0.5) * kMult));
                sum_u[1] += static_cast<uint32_t>(static_cast<int32_t>(h * (e.c[1] + 0.5) * kMult));
                sum_u[2] += static_cast<uint32_t>(static_cast<int32_t>(h * (e.c[2] + 0.5) * kMult));
            }
            total_u += e.h;
        }
        const int n_i32 = static_cast<int32_t>(n_u);
        const int sum[3] = {static_cast<int32_t>(sum_u[0]), static_cast<int32_t>(sum_u[1]), static_cast<int32_t>(sum_u[2])};
        count = static_cast<int32_t>(total_u);
        volume = (hi[0] - lo[0] + 1) * (hi[1] - lo[1] + 1) * (hi[2] - lo[2] + 1);
        if (n_i32 > 0) {
            average = Rgb8{static_cast<uint8_t>(sum[0] / n_i32), static_cast<uint8_t>(sum[1] / n_i32),
                           static_cast<uint8_t>(sum[2] / n_i32)};
        } else { // centre of the (possibly empty) box, clamped to a byte
            int c[3];
            for (int k = 0; k < 3; k++)
                c[k] = std::min(kMult * (lo[k] + hi[k] + 1) / 2, 255);
            average = Rgb8{static_cast<uint8_t>(c[0]), static_cast<uint8_t>(c[1]), static_cast<uint8_t>(c[2])};
        }
    }

    int widest_axis() const
    {
        const int w0 = hi[0] - lo[0], w1 = hi[1] - lo[1], w2 = hi[2] - lo[2];
        const int m = std::max(w0, std::max(w1, w2));
        return m == w0 ? 0 : (m == w1 ? 1 : 2);
    }
};

bool by_count(const Box &a, const Box &b) { return a.count < b.count; }

bool by_count_times_volume(const Box &a, const Box &b)
{
    if (a.count == b.count)
        return a.volume < b.volume;
    return static_cast<long long>(a.count) * a.volume < static_cast<long long>(b.count) * b.volume;
}

// Splits `box` at the population median along its widest axis.  Returns false when the box has
// no pixels / no cut plane; `second_valid` is false when the box holds a single pixel.
bool split(Entry *entries, const Box &box, Box &first, Box &second, bool &second_valid)
{
    second_valid = false;
    if (box.count == 0)
        return false;
    if (box.count == 1) {
        first = box;
        return true;
    }
    const int axis = box.widest_axis();
    int cumulative[32], remaining[32];
    std::fill(cumulative, cumulative + 32, -1);
    std::fill(remaining, remaining + 32, -1);
    uint32_t plane[32] = {}; // population of every plane across the axis (u32: the sums of i32 terms wrap like the crate's)
    for (int i = box.begin; i < box.end; i++)
        plane[entries[i].c[axis]] += entries[i].h;
    int total = 0;
    for (int i = box.lo[axis]; i <= box.hi[axis]; i++) {
        total = static_cast<int32_t>(static_cast<uint32_t>(total) + plane[i]);
        cumulative[i] = total;
    }
    for (int i = 0; i < 32; i++)
        if (cumulative[i] != -1)
            remaining[i] = total - cumulative[i];

    const int vmin = box.lo[axis], vmax = box.hi[axis];
    for (int i = vmin; i <= vmax; i++) {
        if (cumulative[i] <= total / 2)
            continue;
        const int left = i - vmin, right = vmax - i;
        int cut = left <= right ? std::min(vmax - 1, i + right / 2)
                                : std::max(vmin, static_cast<int>((i - 1) - left / 2.0));
        while (cut < 0 || cumulative[cut] <= 0) // avoid an empty first half
            cut++;
        int after = remaining[cut];
        while (after == 0 && cut > 0 && cumulative[cut - 1] > 0) {
            cut--;
            after = remaining[cut];
        }
        first = box;
        second = box;
        first.hi[axis] = cut;
        second.lo[axis] = cut + 1;
        // the occupied bins of the two halves: partition the box's slice at the cut plane
        Entry *mid = std::partition(entries + box.begin, entries + box.end, [axis, cut](const Entry &e) { return e.c[axis] <= cut; });
        first.end = second.begin = static_cast<int>(mid - entries);
        first.refresh(entries);
        second.refresh(entries);
        second_valid = true;
        return true;
    }
    return false;
}

template <typename Less>
bool refine(std::vector<Box> &queue, Less less, int target, Entry *entries)
{
    int colors = 1;
    for (int it = 0; it < kMaxIterations; it++) {
        if (queue.empty())
            break;
        Box top = queue.back();
        if (top.count == 0) {
            std::stable_sort(queue.begin(), queue.end(), less);
            continue;
        }
        queue.pop_back();
        Box a, b;
        bool have_b = false;
        if (!split(entries, top, a, b, have_b))
            return false;
        queue.push_back(a);
        if (have_b) {
            queue.push_back(b);
            colors++;
        }
        std::stable_sort(queue.begin(), queue.end(), less);
        if (colors >= target)
            break;
    }
    return true;
}

struct NamedColor {
    const char *name;
    uint8_t r, g, b;
};

// CSS Color Module Level 4 named colours in alphabetical order.
const NamedColor kCss[] = {
    {"aliceblue", 240, 248, 255}, {"antiquewhite", 250, 235, 215}, {"aqua", 0, 255, 255},
    {"aquamarine", 127, 255, 212}, {"azure", 240, 255, 255}, {"beige", 245, 245, 220},
    {"bisque", 255, 228, 196}, {"black", 0, 0, 0}, {"blanchedalmond", 255, 235, 205},
    {"blue", 0, 0, 255}, {"blueviolet", 138, 43, 226}, {"brown", 165, 42, 42},
    {"burlywood", 222, 184, 135}, {"cadetblue", 95, 158, 160}, {"chartreuse", 127, 255, 0},
    {"chocolate", 210, 105, 30}, {"coral", 255, 127, 80}, {"cornflowerblue", 100, 149, 237},
    {"cornsilk", 255, 248, 220}, {"crimson", 220, 20, 60}, {"cyan", 0, 255, 255},
    {"darkblue", 0, 0, 139}, {"darkcyan", 0, 139, 139}, {"darkgoldenrod", 184, 134, 11},
    {"darkgray", 169, 169, 169}, {"darkgreen", 0, 100, 0}, {"darkgrey", 169, 169, 169},
    {"darkkhaki", 189, 183, 107}, {"darkmagenta", 139, 0, 139}, {"darkolivegreen", 85, 107, 47},
    {"darkorange", 255, 140, 0}, {"darkorchid", 153, 50, 204}, {"darkred", 139, 0, 0},
    {"darksalmon", 233, 150, 122}, {"darkseagreen", 143, 188, 143}, {"darkslateblue", 72, 61, 139},
    {"darkslategray", 47, 79, 79}, {"darkslategrey", 47, 79, 79}, {"darkturquoise", 0, 206, 209},
    {"darkviolet", 148, 0, 211}, {"deeppink", 255, 20, 147}, {"deepskyblue", 0, 191, 255},
    {"dimgray", 105, 105, 105}, {"dimgrey", 105, 105, 105}, {"dodgerblue", 30, 144, 255},
    {"firebrick", 178, 34, 34}, {"floralwhite", 255, 250, 240}, {"forestgreen", 34, 139, 34},
    {"fuchsia", 255, 0, 255}, {"gainsboro", 220, 220, 220}, {"ghostwhite", 248, 248, 255},
    {"gold", 255, 215, 0}, {"goldenrod", 218, 165, 32}, {"gray", 128, 128, 128},
    {"green", 0, 128, 0}, {"greenyellow", 173, 255, 47}, {"grey", 128, 128, 128},
    {"honeydew", 240, 255, 240}, {"hotpink", 255, 105, 180}, {"indianred", 205, 92, 92},
    {"indigo", 75, 0, 130}, {"ivory", 255, 255, 240}, {"khaki", 240, 230, 140},
    {"lavender", 230, 230, 250}, {"lavenderblush", 255, 240, 245}, {"lawngreen", 124, 252, 0},
    {"lemonchiffon", 255, 250, 205}, {"lightblue", 173, 216, 230}, {"lightcoral", 240, 128, 128},
    {"lightcyan", 224, 255, 255}, {"lightgoldenrodyellow", 250, 250, 210}, {"lightgray", 211, 211, 211},
    {"lightgreen", 144, 238, 144}, {"lightgrey", 211, 211, 211}, {"lightpink", 255, 182, 193},
    {"lightsalmon", 255, 160, 122}, {"lightseagreen", 32, 178, 170}, {"lightskyblue", 135, 206, 250},
    {"lightslategray", 119, 136, 153}, {"lightslategrey", 119, 136, 153}, {"lightsteelblue", 176, 196, 222},
    {"lightyellow", 255, 255, 224}, {"lime", 0, 255, 0}, {"limegreen", 50, 205, 50},
    {"linen", 250, 240, 230}, {"magenta", 255, 0, 255}, {"maroon", 128, 0, 0},
    {"mediumaquamarine", 102, 205, 170}, {"mediumblue", 0, 0, 205}, {"mediumorchid", 186, 85, 211},
    {"mediumpurple", 147, 112, 219}, {"mediumseagreen", 60, 179, 113}, {"mediumslateblue", 123, 104, 238},
    {"mediumspringgreen", 0, 250, 154}, {"mediumturquoise", 72, 209, 204}, {"mediumvioletred", 199, 21, 133},
    {"midnightblue", 25, 25, 112}, {"mintcream", 245, 255, 250}, {"mistyrose", 255, 228, 225},
    {"moccasin", 255, 228, 181}, {"navajowhite", 255, 222, 173}, {"navy", 0, 0, 128},
    {"oldlace", 253, 245, 230}, {"olive", 128, 128, 0}, {"olivedrab", 107, 142, 35},
    {"orange", 255, 165, 0}, {"orangered", 255, 69, 0}, {"orchid", 218, 112, 214},
    {"palegoldenrod", 238, 232, 170}, {"palegreen", 152, 251, 152}, {"paleturquoise", 175, 238, 238},
    {"palevioletred", 219, 112, 147}, {"papayawhip", 255, 239, 213}, {"peachpuff", 255, 218, 185},
    {"peru", 205, 133, 63}, {"pink", 255, 192, 203}, {"plum", 221, 160, 221},
    {"powderblue", 176, 224, 230}, {"purple", 128, 0, 128}, {"rebeccapurple", 102, 51, 153},
    {"red", 255, 0, 0}, {"rosybrown", 188, 143, 143}, {"royalblue", 65, 105, 225},
    {"saddlebrown", 139, 69, 19}, {"salmon", 250, 128, 114}, {"sandybrown", 244, 164, 96},
    {"seagreen", 46, 139, 87}, {"seashell", 255, 245, 238}, {"sienna", 160, 82, 45},
    {"silver", 192, 192, 192}, {"skyblue", 135, 206, 235}, {"slateblue", 106, 90, 205},
    {"slategray", 112, 128, 144}, {"slategrey", 112, 128, 144}, {"snow", 255, 250, 250},
    {"springgreen", 0, 255, 127}, {"steelblue", 70, 130, 180}, {"tan", 210, 180, 140},
    {"teal", 0, 128, 128}, {"thistle", 216, 191, 216}, {"tomato", 255, 99, 71},
    {"turquoise", 64, 224, 208}, {"violet", 238, 130, 238}, {"wheat", 245, 222, 179},
    {"white", 255, 255, 255}, {"whitesmoke", 245, 245, 245}, {"yellow", 255, 255, 0},
    {"yellowgreen", 154, 205, 50},
};

} // namespace

std::vector<Rgb8> mmcq_palette(const uint32_t *hist, const uint32_t minmax[6], uint32_t max_colors)
{
    std::vector<Rgb8> palette;
    if (!hist || !minmax || max_colors < 2 || max_colors > 255)
        return palette;
    Box root;
    for (int k = 0; k < 3; k++) {
        root.lo[k] = static_cast<int>(minmax[2 * k]);
        root.hi[k] = static_cast<int>(minmax[2 * k + 1]);
    }
    // the occupied bins inside the root box, once
    std::vector<Entry> occupied;
    occupied.reserve(4096);
    for (int r = std::max(root.lo[0], 0); r <= std::min(root.hi[0], 31); r++)
        for (int g = std::max(root.lo[1], 0); g <= std::min(root.hi[1], 31); g++) {
            const uint32_t *row = hist + bin_of(r, g, 0);
            uint32_t any = 0; // most (r, g) rows of a picture's histogram are empty: one vectorised pass decides
            for (int b = 0; b < 32; b++) any |= row[b];
            if (!any) continue;
            for (int b = std::max(root.lo[2], 0); b <= std::min(root.hi[2], 31); b++)
                if (row[b]) occupied.push_back(Entry{{static_cast<uint8_t>(r), static_cast<uint8_t>(g), static_cast<uint8_t>(b)}, row[b]});
        }
    Entry *entries = occupied.data();
    root.begin = 0;
    root.end = static_cast<int>(occupied.size());
    root.refresh(entries);
    std::vector<Box> queue{root};

    // phase 1: split the most populous boxes until ceil(0.75 * max_colors) colours
    const int first_target = static_cast<int>(std::ceil(kFractionByPopulation * max_colors));
    if (!refine(queue, by_count, first_target, entries))
        return palette;
    // phase 2: split by population x volume for the remainder
    std::stable_sort(queue.begin(), queue.end(), by_count_times_volume);
    if (!refine(queue, by_count_times_volume, static_cast<int>(max_colors) - static_cast<int>(queue.size()), entries))
        return palette;

    for (auto it = queue.rbegin(); it != queue.rend() && palette.size() < max_colors; ++it)
        palette.push_back(it->average);
    return palette;
}

const char *css_color_similar(uint8_t r, uint8_t g, uint8_t b)
{
    const NamedColor *best = &kCss[0];
    long best_d = -1;
    for (const NamedColor &c : kCss) {
        const long dr = static_cast<long>(r) - c.r, dg = static_cast<long>(g) - c.g, db = static_cast<long>(b) - c.b;
        const long d = dr * dr + dg * dg + db * db;
        if (best_d < 0 || d < best_d) {
            best_d = d;
            best = &c;
        }
    }
    return best->name;
}

} // namespace mvfx
