// roundedcorners alpha mask, rendered the way the reference renders it: by libcairo.
//
// video/videofx/src/border/imp.rs:57-106 (draw_rounded_corners) and :108-180 (generate_alpha_mask)
// draw the rounded rectangle with cairo (through cairo-rs) into the A8 plane ONCE per caps /
// border-radius change; the per-frame work of the element is zero pixel arithmetic
// (border/imp.rs:561-563).  The bytes of the mask are therefore *defined* by the system's libcairo
// (anti-aliasing scan converter, fill-then-stroke OVER compositing); an analytic kernel can only
// approximate them (round 1: up to 30/255 off on arc pixels).  This file replays the exact call
// sequence through the same C library, resolved at run time with dlopen (libcairo is a dependency
// of the reference itself, not of this repository's oracle); the HIP side keeps the per-frame
// part: the I420 -> A420 compose (csrc/videofx_kernels.hip).
//
// No link-time dependency: a box without libcairo gets MVFX_ERR_IO with a message naming the
// library, never an approximated mask.
#include "cairo_mask.h"

#include <dlfcn.h>

#include <cstring>
#include <mutex>

#include "mvfx_internal.h"

namespace mvfx {
namespace {

// The subset of the cairo C API the reference's draw_rounded_corners / generate_alpha_mask use.
struct CairoApi {
    void *handle = nullptr;
    void *(*image_surface_create_for_data)(unsigned char *, int, int, int, int) = nullptr;
    int (*surface_status)(void *) = nullptr;
    void (*surface_flush)(void *) = nullptr;
    void (*surface_destroy)(void *) = nullptr;
    void *(*create)(void *) = nullptr;
    int (*status)(void *) = nullptr;
    void (*destroy)(void *) = nullptr;
    void (*new_sub_path)(void *) = nullptr;
    void (*arc)(void *, double, double, double, double, double) = nullptr;
    void (*close_path)(void *) = nullptr;
    void (*set_source_rgb)(void *, double, double, double) = nullptr;
    void (*set_source_rgba)(void *, double, double, double, double) = nullptr;
    void (*fill_preserve)(void *) = nullptr;
    void (*set_line_width)(void *, double) = nullptr;
    void (*stroke)(void *) = nullptr;
    const char *(*status_to_string)(int) = nullptr;
    const char *(*version_string)(void) = nullptr;
    char error[256] = "";
};

CairoApi g_cairo;
std::once_flag g_cairo_once;

template <typename F> bool resolve(void *h, const char *name, F &fn)
{
    fn = reinterpret_cast<F>(dlsym(h, name));
    return fn != nullptr;
}

void load_cairo()
{
    // MVFX_CAIRO_LIBRARY overrides the search (a maintainer pinning a specific cairo build)
    const char *env = getenv("MVFX_CAIRO_LIBRARY");
    const char *candidates[] = {env, "libcairo.so.2", "/usr/lib/x86_64-linux-gnu/libcairo.so.2",
                                "/lib/x86_64-linux-gnu/libcairo.so.2", "/opt/conda/lib/libcairo.so.2"};
    void *h = nullptr;
    for (const char *c : candidates) {
        if (!c || !*c) continue;
        h = dlopen(c, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
    }
    if (!h) {
        snprintf(g_cairo.error, sizeof(g_cairo.error), "libcairo.so.2 not found (%s)", dlerror());
        return;
    }
    CairoApi &a = g_cairo;
    const bool ok = resolve(h, "cairo_image_surface_create_for_data", a.image_surface_create_for_data) &&
                    resolve(h, "cairo_surface_status", a.surface_status) && resolve(h, "cairo_surface_flush", a.surface_flush) &&
                    resolve(h, "cairo_surface_destroy", a.surface_destroy) && resolve(h, "cairo_create", a.create) &&
                    resolve(h, "cairo_status", a.status) && resolve(h, "cairo_destroy", a.destroy) &&
                    resolve(h, "cairo_new_sub_path", a.new_sub_path) && resolve(h, "cairo_arc", a.arc) &&
                    resolve(h, "cairo_close_path", a.close_path) && resolve(h, "cairo_set_source_rgb", a.set_source_rgb) &&
                    resolve(h, "cairo_set_source_rgba", a.set_source_rgba) && resolve(h, "cairo_fill_preserve", a.fill_preserve) &&
                    resolve(h, "cairo_set_line_width", a.set_line_width) && resolve(h, "cairo_stroke", a.stroke) &&
                    resolve(h, "cairo_status_to_string", a.status_to_string) &&
                    resolve(h, "cairo_version_string", a.version_string);
    if (!ok) {
        snprintf(a.error, sizeof(a.error), "libcairo is missing a required symbol (%s)", dlerror());
        dlclose(h);
        return;
    }
    a.handle = h;
}

constexpr int kCairoFormatA8 = 2;  // cairo_format_t CAIRO_FORMAT_A8
constexpr int kCairoStatusSuccess = 0;

} // namespace

const char *cairo_mask_library_version()
{
    std::call_once(g_cairo_once, load_cairo);
    return g_cairo.handle ? g_cairo.version_string() : nullptr;
}

int cairo_render_rounded_mask(uint8_t *mask, uint32_t width, uint32_t height, uint32_t stride, uint32_t border_radius_px)
{
    const uint32_t rows = (height + 1) & ~1u; // border/imp.rs:469-470: stride[3] * round_up_2(height)
    const size_t bytes = (size_t)stride * rows;
    if (border_radius_px == 0) { // border/imp.rs:123-128: opaque plane, cairo never runs
        memset(mask, 0xff, bytes);
        return MVFX_OK;
    }
    std::call_once(g_cairo_once, load_cairo);
    const CairoApi &c = g_cairo;
    if (!c.handle)
        return fail(MVFX_ERR_IO, "roundedcorners: %s; the reference renders this mask with libcairo and so does this library", c.error);
    if (width > 32767u || height > 32767u)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "roundedcorners: %ux%u exceeds cairo's image surface limit", width, height);
    memset(mask, 0, bytes); // border/imp.rs:130 alpha_mem.fill(0)
    void *surface = c.image_surface_create_for_data(mask, kCairoFormatA8, (int)width, (int)height, (int)stride);
    if (int st = c.surface_status(surface); st != kCairoStatusSuccess) {
        c.surface_destroy(surface);
        return fail(MVFX_ERR_INVALID_ARGUMENT, "roundedcorners: Failed to create cairo image surface: %s", c.status_to_string(st));
    }
    void *cr = c.create(surface);
    if (int st = c.status(cr); st != kCairoStatusSuccess) {
        c.destroy(cr);
        c.surface_destroy(surface);
        return fail(MVFX_ERR_DEVICE, "roundedcorners: Failed to create cairo context: %s", c.status_to_string(st));
    }
    // border/imp.rs:64-103, statement for statement (f64 arithmetic as in the Rust source)
    const double border_radius = (double)border_radius_px;
    const double degrees = 3.14159265358979323846264338327950288 / 180.0;
    const double w = (double)(int)width, h = (double)(int)height;
    c.new_sub_path(cr);
    c.arc(cr, w - border_radius, border_radius, border_radius, -90.0 * degrees, 0.0 * degrees);
    c.arc(cr, w - border_radius, h - border_radius, border_radius, 0.0 * degrees, 90.0 * degrees);
    c.arc(cr, border_radius, h - border_radius, border_radius, 90.0 * degrees, 180.0 * degrees);
    c.arc(cr, border_radius, border_radius, border_radius, 180.0 * degrees, 270.0 * degrees);
    c.close_path(cr);
    c.set_source_rgb(cr, 0.0, 0.0, 0.0);
    c.fill_preserve(cr);
    c.set_source_rgba(cr, 0.0, 0.0, 0.0, 1.0);
    c.set_line_width(cr, 1.0);
    c.stroke(cr);
    const int st = c.status(cr);
    c.destroy(cr);
    c.surface_flush(surface);
    c.surface_destroy(surface);
    if (st != kCairoStatusSuccess)
        return fail(MVFX_ERR_DEVICE, "roundedcorners: Failed to draw rounded corners: %s", c.status_to_string(st));
    return MVFX_OK;
}

} // namespace mvfx
