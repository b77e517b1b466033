// Adobe .cube parser (host). Behaviour follows video/colorlut/src/parser.rs line by line in
// WHAT it accepts and rejects; the implementation is a small hand-written scanner.
//
// Reference rules reproduced (SURVEY.md 8a note a7):
//  - lines are trimmed; empty lines and lines starting with '#' are skipped        (:118-123)
//  - tokens are separated by Unicode white space                                     (:125)
//  - keywords TITLE, DOMAIN_MIN, DOMAIN_MAX, LUT_1D_SIZE, LUT_3D_SIZE; anything else is a
//    data row of exactly three floats (so unknown keywords fail as "Invalid float")  (:132-201)
//  - no header keyword after the first data row                                      (:284-303)
//  - exactly one size keyword; 1-D size 2..=65536, 3-D size 2..=256                 (:12-16,:146-176)
//  - numbers use Rust's str::parse grammar (stricter than strtof)
//  - domain check min >= max after all lines                                         (:205-212)
//  - value count must equal size (1-D) or size^3 (3-D)                               (:219-251)
#include "cube_parser.h"

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <locale.h>

namespace mvfx {
namespace {

// Decodes one UTF-8 scalar starting at s[i]; returns its length (1 on malformed input).
size_t decode_utf8(std::string_view s, size_t i, uint32_t &cp)
{
    const auto b = [&](size_t k) { return static_cast<unsigned char>(s[k]); };
    const unsigned char c = b(i);
    if (c < 0x80) { cp = c; return 1; }
    if ((c & 0xE0) == 0xC0 && i + 1 < s.size()) { cp = ((c & 0x1Fu) << 6) | (b(i + 1) & 0x3Fu); return 2; }
    if ((c & 0xF0) == 0xE0 && i + 2 < s.size()) {
        cp = ((c & 0x0Fu) << 12) | ((b(i + 1) & 0x3Fu) << 6) | (b(i + 2) & 0x3Fu);
        return 3;
    }
    if ((c & 0xF8) == 0xF0 && i + 3 < s.size()) {
        cp = ((c & 0x07u) << 18) | ((b(i + 1) & 0x3Fu) << 12) | ((b(i + 2) & 0x3Fu) << 6) | (b(i + 3) & 0x3Fu);
        return 4;
    }
    cp = 0xFFFD;
    return 1;
}

// char::is_whitespace (Unicode White_Space property)
bool is_unicode_space(uint32_t cp)
{
    return (cp >= 0x09 && cp <= 0x0D) || cp == 0x20 || cp == 0x85 || cp == 0xA0 || cp == 0x1680 ||
           (cp >= 0x2000 && cp <= 0x200A) || cp == 0x2028 || cp == 0x2029 || cp == 0x202F ||
           cp == 0x205F || cp == 0x3000;
}

// split_whitespace(); trimming falls out of it (a trimmed-empty line has no tokens).
std::vector<std::string_view> tokens_of(std::string_view line)
{
    std::vector<std::string_view> out;
    size_t i = 0, start = std::string_view::npos;
    while (i < line.size()) {
        uint32_t cp;
        const size_t n = decode_utf8(line, i, cp);
        if (is_unicode_space(cp)) {
            if (start != std::string_view::npos) {
                out.push_back(line.substr(start, i - start));
                start = std::string_view::npos;
            }
        } else if (start == std::string_view::npos) {
            start = i;
        }
        i += n;
    }
    if (start != std::string_view::npos)
        out.push_back(line.substr(start));
    return out;
}

bool all_digits(std::string_view s, size_t from, size_t &to)
{
    size_t i = from;
    while (i < s.size() && s[i] >= '0' && s[i] <= '9')
        i++;
    to = i;
    return i > from;
}

bool iequals(std::string_view a, const char *b)
{
    const size_t n = std::strlen(b);
    if (a.size() != n)
        return false;
    for (size_t i = 0; i < n; i++) {
        char c = a[i];
        if (c >= 'A' && c <= 'Z')
            c = static_cast<char>(c - 'A' + 'a');
        if (c != b[i])
            return false;
    }
    return true;
}

// Rust `str::parse::<f32>()`: [+-]? (inf|infinity|nan | digits* [. digits*] ([eE][+-]?digits+)?)
// with at least one mantissa digit; correctly rounded, locale-independent.
bool rust_f32(std::string_view tok, float &out)
{
    if (tok.empty() || tok.size() > 4096)
        return false;
    size_t i = (tok[0] == '+' || tok[0] == '-') ? 1 : 0;
    const std::string_view body = tok.substr(i);
    if (body.empty())
        return false;
    if (!(iequals(body, "inf") || iequals(body, "infinity") || iequals(body, "nan"))) {
        size_t j = i, k;
        const bool int_digits = all_digits(tok, j, k);
        j = k;
        bool frac_digits = false;
        if (j < tok.size() && tok[j] == '.') {
            frac_digits = all_digits(tok, j + 1, k);
            j = k > j + 1 ? k : j + 1;
        }
        if (!int_digits && !frac_digits)
            return false;
        if (j < tok.size() && (tok[j] == 'e' || tok[j] == 'E')) {
            j++;
            if (j < tok.size() && (tok[j] == '+' || tok[j] == '-'))
                j++;
            if (!all_digits(tok, j, k))
                return false;
            j = k;
        }
        if (j != tok.size())
            return false;
    }
    // strtof_l in the "C" locale: correctly rounded like Rust's dec2flt (overflow -> inf, underflow -> 0/denormal)
    // and -- unlike plain strtof -- independent of the process locale (gst-launch and GTK applications call
    // setlocale(LC_ALL, ""); under a comma-decimal LC_NUMERIC strtof stops at the '.').
    static const locale_t c_locale = newlocale(LC_ALL_MASK, "C", static_cast<locale_t>(nullptr));
    const std::string z(tok);
    char *end = nullptr;
    out = c_locale ? strtof_l(z.c_str(), &end, c_locale) : std::strtof(z.c_str(), &end);
    return end == z.c_str() + z.size();
}

// Rust `str::parse::<usize>()`: optional '+', decimal digits, overflow is an error.
bool rust_usize(std::string_view tok, uint64_t &out)
{
    size_t i = (!tok.empty() && tok[0] == '+') ? 1 : 0;
    if (i >= tok.size())
        return false;
    uint64_t v = 0;
    for (; i < tok.size(); i++) {
        if (tok[i] < '0' || tok[i] > '9')
            return false;
        const uint64_t d = static_cast<uint64_t>(tok[i] - '0');
        if (v > (UINT64_MAX - d) / 10)
            return false;
        v = v * 10 + d;
    }
    out = v;
    return true;
}

std::string at_line(const char *what, size_t line_no, std::string_view line)
{
    std::string s(what);
    s += " at line " + std::to_string(line_no) + ": ";
    s.append(line.data(), line.size());
    return s;
}

enum class State { Header, Lut1D, Lut3D }; // parser.rs:96-101

} // namespace

bool parse_cube(std::string_view text, CubeLut &out, std::string &error)
{
    float dmin[3] = {0.0f, 0.0f, 0.0f}, dmax[3] = {1.0f, 1.0f, 1.0f};
    State state = State::Header;
    bool have_data = false;
    uint64_t size = 0;
    std::vector<float> values; // flattened [r,g,b] rows

    size_t pos = 0, line_no = 0;
    while (pos < text.size()) { // str::lines()
        size_t nl = text.find('\n', pos);
        std::string_view raw = text.substr(pos, nl == std::string_view::npos ? std::string_view::npos : nl - pos);
        pos = nl == std::string_view::npos ? text.size() : nl + 1;
        line_no++;
        const std::vector<std::string_view> tok = tokens_of(raw);
        if (tok.empty() || tok[0][0] == '#')
            continue;
        // the reference's messages quote the trimmed line
        const std::string_view line(tok.front().data(), (tok.back().data() + tok.back().size()) - tok.front().data());
        const std::string_view kw = tok[0];

        const auto header_only = [&]() { // ensure_header parser.rs:284-303
            if (have_data) {
                error = at_line("Header found after LUT data", line_no, line);
                return false;
            }
            return true;
        };
        const auto vec3 = [&](float dst[3]) { // parse_vec3 :317-334
            if (tok.size() < 4) { error = at_line("Invalid line", line_no, line); return false; }
            float v[3];
            for (int c = 0; c < 3; c++)
                if (!rust_f32(tok[1 + c], v[c])) { error = at_line("Invalid float", line_no, line); return false; }
            if (tok.size() > 4) { error = at_line("Invalid line", line_no, line); return false; }
            dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2];
            return true;
        };

        if (kw == "TITLE") {
            if (!header_only()) return false;
        } else if (kw == "DOMAIN_MIN") {
            if (!header_only() || !vec3(dmin)) return false;
        } else if (kw == "DOMAIN_MAX") {
            if (!header_only() || !vec3(dmax)) return false;
        } else if (kw == "LUT_1D_SIZE" || kw == "LUT_3D_SIZE") {
            const bool one_d = kw == "LUT_1D_SIZE";
            if (!header_only()) return false;
            if (state != State::Header) { // :149-153 / :166-170
                error = at_line(one_d ? "Invalid LUT_1D_SIZE" : "Invalid LUT_3D_SIZE", line_no, line);
                return false;
            }
            if (tok.size() < 2) { error = at_line("Invalid line", line_no, line); return false; }
            uint64_t v;
            if (!rust_usize(tok[1], v)) { error = at_line("Invalid integer", line_no, line); return false; }
            if (tok.size() > 2) { error = at_line("Invalid line", line_no, line); return false; }
            const uint64_t lo = 2, hi = one_d ? 65536 : 256; // parser.rs:12-16
            if (v < lo || v > hi) {
                error = "Invalid LUT size " + std::to_string(v) + " at line " + std::to_string(line_no) +
                        ", expected " + std::to_string(lo) + "..=" + std::to_string(hi);
                return false;
            }
            size = v;
            state = one_d ? State::Lut1D : State::Lut3D;
        } else { // data row :178-201
            if (state == State::Header) {
                error = at_line("LUT data found before LUT size", line_no, line);
                return false;
            }
            have_data = true;
            float v[3];
            for (size_t c = 0; c < 3; c++) {
                if (c >= tok.size()) { error = at_line("Invalid line", line_no, line); return false; }
                if (!rust_f32(tok[c], v[c])) { error = at_line("Invalid float", line_no, line); return false; }
            }
            if (tok.size() > 3) { error = at_line("Invalid line", line_no, line); return false; }
            values.insert(values.end(), v, v + 3);
        }
    }

    // :205-212 -- plain comparisons, so NaN bounds pass exactly as in the reference
    if (dmin[0] >= dmax[0] || dmin[1] >= dmax[1] || dmin[2] >= dmax[2]) {
        error = "Invalid domain min/max";
        return false;
    }
    if (state == State::Header) { // :215-217
        error = "Missing LUT size";
        return false;
    }
    const uint64_t rows = values.size() / 3;
    CubeLut lut;
    lut.size = static_cast<uint32_t>(size);
    if (state == State::Lut1D) {
        if (rows != size) {
            error = "Invalid 1D LUT value count, expected " + std::to_string(size) + ", got " + std::to_string(rows);
            return false;
        }
        lut.is_3d = false;
        for (int c = 0; c < 3; c++) {
            lut.table[c].resize(size);
            for (uint64_t i = 0; i < size; i++)
                lut.table[c][i] = values[3 * i + c];
        }
    } else {
        const uint64_t expected = size * size * size;
        if (rows != expected) {
            error = "Invalid 3D LUT value count, expected " + std::to_string(expected) + ", got " + std::to_string(rows);
            return false;
        }
        lut.is_3d = true;
        lut.rgba.resize(expected * 4);
        for (uint64_t i = 0; i < expected; i++) {
            lut.rgba[4 * i + 0] = values[3 * i + 0];
            lut.rgba[4 * i + 1] = values[3 * i + 1];
            lut.rgba[4 * i + 2] = values[3 * i + 2];
            lut.rgba[4 * i + 3] = 1.0f;
        }
    }
    for (int c = 0; c < 3; c++) { // :264-274
        lut.domain_min[c] = dmin[c];
        lut.domain_max[c] = dmax[c];
        lut.domain_scale[c] = 1.0f / (dmax[c] - dmin[c]);
        lut.domain_offset[c] = -dmin[c] * lut.domain_scale[c];
    }
    out = std::move(lut);
    return true;
}

namespace {
bool valid_utf8(const std::string &s)
{
    size_t i = 0;
    while (i < s.size()) {
        const unsigned char c = static_cast<unsigned char>(s[i]);
        size_t n = c < 0x80 ? 1 : (c & 0xE0) == 0xC0 ? 2 : (c & 0xF0) == 0xE0 ? 3 : (c & 0xF8) == 0xF0 ? 4 : 0;
        if (n == 0 || i + n > s.size())
            return false;
        for (size_t k = 1; k < n; k++)
            if ((static_cast<unsigned char>(s[i + k]) & 0xC0) != 0x80)
                return false;
        if (n == 2 && c < 0xC2) return false;                                        // overlong
        if (n == 3 && c == 0xE0 && static_cast<unsigned char>(s[i + 1]) < 0xA0) return false;
        if (n == 3 && c == 0xED && static_cast<unsigned char>(s[i + 1]) > 0x9F) return false; // surrogates
        if (n == 4 && (c > 0xF4 || (c == 0xF0 && static_cast<unsigned char>(s[i + 1]) < 0x90) ||
                       (c == 0xF4 && static_cast<unsigned char>(s[i + 1]) > 0x8F)))
            return false;
        i += n;
    }
    return true;
}
} // namespace

std::string write_cube(const CubeLut &lut)
{
    static const locale_t c_locale = newlocale(LC_ALL_MASK, "C", static_cast<locale_t>(nullptr));
    const locale_t old = c_locale ? uselocale(c_locale) : static_cast<locale_t>(nullptr); // '.' decimal point whatever LC_NUMERIC says
    std::string out;
    char line[128];
    const auto f3 = [&](const char *key, float a, float b, float c) {
        snprintf(line, sizeof(line), "%s%.9g %.9g %.9g\n", key, (double)a, (double)b, (double)c);
        out += line;
    };
    snprintf(line, sizeof(line), "%s %u\n", lut.is_3d ? "LUT_3D_SIZE" : "LUT_1D_SIZE", lut.size);
    out += line;
    const bool default_domain = lut.domain_min[0] == 0.0f && lut.domain_min[1] == 0.0f && lut.domain_min[2] == 0.0f &&
                                lut.domain_max[0] == 1.0f && lut.domain_max[1] == 1.0f && lut.domain_max[2] == 1.0f;
    if (!default_domain) {
        f3("DOMAIN_MIN ", lut.domain_min[0], lut.domain_min[1], lut.domain_min[2]);
        f3("DOMAIN_MAX ", lut.domain_max[0], lut.domain_max[1], lut.domain_max[2]);
    }
    if (lut.is_3d) {
        const size_t n = (size_t)lut.size * lut.size * lut.size;
        out.reserve(out.size() + n * 36);
        for (size_t i = 0; i < n; i++) f3("", lut.rgba[4 * i], lut.rgba[4 * i + 1], lut.rgba[4 * i + 2]); // R fastest (parser.rs:43-53)
    } else {
        for (size_t i = 0; i < lut.size; i++) f3("", lut.table[0][i], lut.table[1][i], lut.table[2][i]);
    }
    if (old) uselocale(old);
    return out;
}

bool parse_cube_file(const char *path, CubeLut &out, std::string &error, bool &io_error)
{
    io_error = false;
    std::FILE *f = path ? std::fopen(path, "rb") : nullptr;
    if (!f) {
        io_error = true;
        error = std::string("IO error: ") + (path ? std::strerror(errno) : "no path");
        return false;
    }
    std::string text;
    char buf[1 << 16];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0)
        text.append(buf, n);
    const bool read_failed = std::ferror(f) != 0;
    std::fclose(f);
    if (read_failed || !valid_utf8(text)) { // fs::read_to_string fails on invalid UTF-8
        io_error = true;
        error = read_failed ? "IO error: read failed" : "IO error: stream did not contain valid UTF-8";
        return false;
    }
    if (!parse_cube(text, out, error)) {
        error = "Invalid LUT: " + error;
        return false;
    }
    return true;
}

} // namespace mvfx
