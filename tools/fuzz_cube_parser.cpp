// tools/fuzz_cube_parser.cpp -- sanitizer fuzz of the host-side .cube parser and the MMCQ median cut
// (the two pieces of host logic that eat untrusted input).  CPU only:
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -Igst-plugin-rs_amd/host \
//       tools/fuzz_cube_parser.cpp gst-plugin-rs_amd/host/cube_parser.cpp gst-plugin-rs_amd/host/mmcq.cpp -o /tmp/fz/fuzz && /tmp/fz/fuzz tests/golden/*.cube
#include "cube_parser.h"
#include "mmcq.h"

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <random>
#include <sstream>

int main(int argc, char **argv)
{
    std::vector<std::string> seeds;
    for (int i = 1; i < argc; i++) {
        std::ifstream f(argv[i], std::ios::binary);
        std::stringstream ss;
        ss << f.rdbuf();
        seeds.push_back(ss.str());
    }
    seeds.push_back("LUT_3D_SIZE 2\n0 0 0\n1 0 0\n0 1 0\n1 1 0\n0 0 1\n1 0 1\n0 1 1\n1 1 1\n");
    seeds.push_back("TITLE \"x\"\nLUT_1D_SIZE 2\nDOMAIN_MIN 0 0 0\nDOMAIN_MAX 1 1 1\n0 0 0\n1 1 1\n");
    std::mt19937_64 rng(12345);
    const char alphabet[] = "0123456789.-+eE \t\n#LUT_13DSIZEDOMAINMXTLnaif\"\r\xc2\xa0\xff";
    size_t accepted = 0, rejected = 0;
    const int iters = argc > 1 && getenv("FUZZ_ITERS") ? atoi(getenv("FUZZ_ITERS")) : 200000;
    for (int it = 0; it < iters; it++) {
        std::string s = seeds[rng() % seeds.size()];
        const int edits = 1 + (int)(rng() % 8);
        for (int e = 0; e < edits && !s.empty(); e++) {
            const size_t pos = rng() % s.size();
            switch (rng() % 5) {
            case 0: s[pos] = alphabet[rng() % (sizeof(alphabet) - 1)]; break;
            case 1: s.insert(pos, 1, alphabet[rng() % (sizeof(alphabet) - 1)]); break;
            case 2: s.erase(pos, 1 + rng() % 4); break;
            case 3: s.insert(pos, s.substr(rng() % s.size(), rng() % 32)); break;
            default: s.resize(pos); break;
            }
        }
        mvfx::CubeLut lut;
        std::string err;
        if (mvfx::parse_cube(s, lut, err)) {
            accepted++;
            // an accepted LUT must be self-consistent
            const size_t n = lut.size;
            if (lut.is_3d ? lut.rgba.size() != n * n * n * 4 : (lut.table[0].size() != n || lut.table[1].size() != n || lut.table[2].size() != n)) {
                fprintf(stderr, "inconsistent LUT accepted\n");
                return 1;
            }
        } else {
            rejected++;
            if (err.empty()) { fprintf(stderr, "rejection without message\n"); return 1; }
        }
    }
    // MMCQ on random sparse histograms, incl. degenerate ones
    for (int it = 0; it < 3000; it++) {
        std::vector<uint32_t> hist(32768, 0);
        const int bins = (int)(rng() % 200);
        uint32_t mm[6] = {31, 0, 31, 0, 31, 0};
        for (int b = 0; b < bins; b++) {
            const uint32_t idx = (uint32_t)(rng() % 32768);
            hist[idx] += 1 + (uint32_t)(rng() % (it % 7 == 0 ? 1000000 : 50));
            const uint32_t r = idx >> 10, g = (idx >> 5) & 31, bb = idx & 31;
            mm[0] = std::min(mm[0], r); mm[1] = std::max(mm[1], r);
            mm[2] = std::min(mm[2], g); mm[3] = std::max(mm[3], g);
            mm[4] = std::min(mm[4], bb); mm[5] = std::max(mm[5], bb);
        }
        if (bins == 0) { mm[0] = mm[2] = mm[4] = 255; mm[1] = mm[3] = mm[5] = 0; }
        const uint32_t max_colors = 2 + (uint32_t)(rng() % 254);
        (void)mvfx::mmcq_palette(hist.data(), mm, max_colors);
    }
    printf("fuzz ok: %zu accepted, %zu rejected, 3000 MMCQ runs\n", accepted, rejected);
    return 0;
}
