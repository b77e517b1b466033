// imagersoverlay's per-frame work for gfx950: `composition.blend(frame)` (video/image/src/overlay/imp.rs:703-727), i.e.
// libgstvideo's gst_video_overlay_composition_blend of ONE unscaled BGRA rectangle onto a packed RGB frame, in place.
// Integer arithmetic of libgstvideo 1.14.0 (see include/mi355vfx.h and oracle/videofx_oracle.c; pinned by vectors made with
// the image's own library).  One lane per overlay pixel of the clipped rectangle; HBM traffic = the rectangle only
// (4 B read of the overlay + 3-4 B read + 3-4 B written of the frame per covered pixel), the rest of the frame is not touched.
#include "mvfx_internal.h"

#include <algorithm>

namespace mvfx {
namespace {

constexpr int kBlock = 256;

struct BlendLayout {
    int bpp, ir, ig, ib, ia; // byte index of R, G, B, alpha (-1: none) inside a destination pixel
};

bool blend_layout(int format, BlendLayout *l)
{
    switch (format) {
    case MVFX_FORMAT_RGBA: case MVFX_FORMAT_RGBX: *l = {4, 0, 1, 2, 3}; return true;
    case MVFX_FORMAT_BGRA: case MVFX_FORMAT_BGRX: *l = {4, 2, 1, 0, 3}; return true;
    case MVFX_FORMAT_ARGB: case MVFX_FORMAT_XRGB: *l = {4, 1, 2, 3, 0}; return true;
    case MVFX_FORMAT_ABGR: case MVFX_FORMAT_XBGR: *l = {4, 3, 2, 1, 0}; return true;
    case MVFX_FORMAT_RGB: *l = {3, 0, 1, 2, -1}; return true;
    case MVFX_FORMAT_BGR: *l = {3, 2, 1, 0, -1}; return true;
    default: return false;
    }
}

// rectangle [0,cw) x [0,ch) of overlay pixels (ox0 + i, oy0 + j) onto frame pixels (dx0 + i, dy0 + j)
__global__ __launch_bounds__(kBlock) void overlay_blend_kernel(uint8_t *frame, uint32_t frame_stride, const uint8_t *overlay,
                                                               uint32_t overlay_stride, uint32_t cw, uint32_t ch, uint32_t dx0,
                                                               uint32_t dy0, uint32_t ox0, uint32_t oy0, BlendLayout l, uint32_t g,
                                                               bool have_g)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x, j = blockIdx.y;
    if (i >= cw || j >= ch) return;
    const uint32_t s = *reinterpret_cast<const uint32_t *>(overlay + (uint64_t)(oy0 + j) * overlay_stride + (uint64_t)(ox0 + i) * 4); // B G R A
    uint32_t a_s = s >> 24;
    if (have_g) a_s = a_s * g / 255u;
    if (a_s == 0) return; // the library skips the pixel: destination untouched
    uint8_t *d = frame + (uint64_t)(dy0 + j) * frame_stride + (uint64_t)(dx0 + i) * l.bpp;
    const uint32_t a_d = l.ia >= 0 ? d[l.ia] : 255u;
    const uint32_t w = a_d * (255u - a_s);          // weight of the destination colour, x 255
    const uint32_t a_o = a_s + w / 255u;
    const uint32_t div = a_o ? a_o : 1u;
    const uint32_t cr = (s >> 16) & 0xffu, cg = (s >> 8) & 0xffu, cb = s & 0xffu;
    const uint32_t r = (cr * a_s + d[l.ir] * w / 255u) / div, gg = (cg * a_s + d[l.ig] * w / 255u) / div, b = (cb * a_s + d[l.ib] * w / 255u) / div;
    d[l.ir] = (uint8_t)min(r, 255u);
    d[l.ig] = (uint8_t)min(gg, 255u);
    d[l.ib] = (uint8_t)min(b, 255u);
    if (l.ia >= 0) d[l.ia] = (uint8_t)a_o;
}

struct Clip {
    uint32_t cw, ch, dx0, dy0, ox0, oy0;
};

// intersection of the rectangle at (x, y) with the frame; false when empty
bool clip_rect(const mvfx_frame *frame, const mvfx_frame *overlay, int32_t x, int32_t y, Clip *c)
{
    const int64_t x0 = std::max<int64_t>(x, 0), y0 = std::max<int64_t>(y, 0);
    const int64_t x1 = std::min<int64_t>((int64_t)x + overlay->width, frame->width), y1 = std::min<int64_t>((int64_t)y + overlay->height, frame->height);
    if (x1 <= x0 || y1 <= y0) return false;
    *c = Clip{(uint32_t)(x1 - x0), (uint32_t)(y1 - y0), (uint32_t)x0, (uint32_t)y0, (uint32_t)(x0 - x), (uint32_t)(y0 - y)};
    return true;
}

int check_blend_args(const mvfx_frame *frame, const mvfx_frame *overlay, float global_alpha, BlendLayout *l)
{
    if (!frame || !overlay)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "overlay_blend: NULL frame");
    if (!blend_layout(frame->format, l))
        return fail(MVFX_ERR_UNSUPPORTED_FORMAT, "overlay_blend: destination format %d is not a packed RGB format", frame->format);
    if (overlay->format != MVFX_FORMAT_BGRA)
        return fail(MVFX_ERR_UNSUPPORTED_FORMAT, "overlay_blend: the overlay rectangle must be BGRA (overlay/imp.rs:270-283)");
    if (int rc = check_packed_frame(frame, "overlay_blend frame"); rc != MVFX_OK) return rc;
    if (int rc = check_packed_frame(overlay, "overlay_blend overlay"); rc != MVFX_OK) return rc;
    if (!(global_alpha >= 0.0f && global_alpha <= 1.0f))
        return fail(MVFX_ERR_INVALID_ARGUMENT, "overlay_blend: global alpha %g is not in [0, 1] (gst_video_overlay_rectangle_set_global_alpha)", (double)global_alpha);
    if (((uintptr_t)overlay->data | overlay->stride) & 3)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "overlay_blend: the BGRA overlay must be 4-byte aligned");
    return MVFX_OK;
}

void launch_blend(uint8_t *frame, uint32_t frame_stride, const uint8_t *overlay, uint32_t overlay_stride, const Clip &c, const BlendLayout &l,
                  float global_alpha, hipStream_t st)
{
    const dim3 grid((c.cw + kBlock - 1) / kBlock, c.ch);
    MVFX_LAUNCH(overlay_blend_kernel, grid, dim3(kBlock), 0, st, frame, frame_stride, overlay, overlay_stride, c.cw, c.ch, c.dx0, c.dy0,
                       c.ox0, c.oy0, l, (uint32_t)(int)(global_alpha * 255.0f), global_alpha != 1.0f);
}

} // namespace
} // namespace mvfx

using namespace mvfx;

extern "C" {

int mvfx_overlay_blend(const mvfx_frame *frame, const mvfx_frame *overlay, int32_t x, int32_t y, float global_alpha, mvfx_stream stream)
{
    BlendLayout l;
    if (int rc = check_blend_args(frame, overlay, global_alpha, &l); rc != MVFX_OK) return rc;
    if (int rc = require_device(); rc != MVFX_OK) return rc;
    Clip c;
    if (!clip_rect(frame, overlay, x, y, &c)) return MVFX_OK; // nothing of the rectangle is inside the frame
    if (c.ch > 65535u) return fail(MVFX_ERR_INVALID_ARGUMENT, "overlay_blend: rectangle taller than 65535 rows");
    launch_blend(static_cast<uint8_t *>(frame->data), frame->stride, static_cast<const uint8_t *>(overlay->data), overlay->stride, c, l, global_alpha,
                 as_stream(stream));
    MVFX_HIP_TRY(hipGetLastError());
    return MVFX_OK;
}

int mvfx_overlay_blend_host(const mvfx_frame *frame, const mvfx_frame *overlay, int32_t x, int32_t y, float global_alpha)
{
    BlendLayout l;
    if (int rc = check_blend_args(frame, overlay, global_alpha, &l); rc != MVFX_OK) return rc;
    if (int rc = require_device(); rc != MVFX_OK) return rc;
    Clip c;
    if (!clip_rect(frame, overlay, x, y, &c)) return MVFX_OK;
    if (c.ch > 65535u) return fail(MVFX_ERR_INVALID_ARGUMENT, "overlay_blend: rectangle taller than 65535 rows");
    // only the clipped rectangle of either image travels: tight copies in two scratch blocks
    const size_t frow = (size_t)c.cw * l.bpp, orow = (size_t)c.cw * 4;
    void *fdev = nullptr, *odev = nullptr;
    if (int rc = host_scratch(frow * c.ch + 16, 0, &fdev); rc != MVFX_OK) return rc;
    if (int rc = host_scratch(orow * c.ch + 16, 1, &odev); rc != MVFX_OK) return rc;
    hipStream_t st = host_stream();
    uint8_t *fhost = static_cast<uint8_t *>(frame->data) + (size_t)c.dy0 * frame->stride + (size_t)c.dx0 * l.bpp;
    const uint8_t *ohost = static_cast<const uint8_t *>(overlay->data) + (size_t)c.oy0 * overlay->stride + (size_t)c.ox0 * 4;
    MVFX_HIP_TRY(hipMemcpy2DAsync(fdev, frow, fhost, frame->stride, frow, c.ch, hipMemcpyHostToDevice, st));
    MVFX_HIP_TRY(hipMemcpy2DAsync(odev, orow, ohost, overlay->stride, orow, c.ch, hipMemcpyHostToDevice, st));
    const Clip tight{c.cw, c.ch, 0, 0, 0, 0};
    launch_blend(static_cast<uint8_t *>(fdev), (uint32_t)frow, static_cast<const uint8_t *>(odev), (uint32_t)orow, tight, l, global_alpha, st);
    MVFX_HIP_TRY(hipGetLastError());
    MVFX_HIP_TRY(hipMemcpy2DAsync(fhost, frame->stride, fdev, frow, frow, c.ch, hipMemcpyDeviceToHost, st));
    MVFX_HIP_TRY(hipStreamSynchronize(st));
    return MVFX_OK;
}

} // extern "C"
