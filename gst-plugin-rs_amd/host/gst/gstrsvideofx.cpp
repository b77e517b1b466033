// libgstrsvideofx.so -- plugin `rsvideofx` with elements `roundedcorners`, `colordetect` and
// `videocompare`.  Same surface as video/videofx/src/{lib.rs,border,colordetect,videocompare}.
// `videocompare` is a GstVideoAggregator subclass in the reference; the GStreamer 1.14 of the
// build image only has GstAggregator, so here it derives from GstAggregator directly
// and keeps per-pad GstVideoInfo itself; names, request pads, properties, reference-pad rule,
// output buffer and bus message are the reference's.
#include "mvfx_gst_common.h"

#include <gst/base/gstaggregator.h>

#include <gst/video/gstvideometa.h>

#include <mutex>
#include <string>
#include <vector>

GST_DEBUG_CATEGORY_STATIC(colordetect_debug);
GST_DEBUG_CATEGORY_STATIC(roundedcorners_debug);

// ------------------------------------------------------------------------- colordetect

struct GstColorDetect {
    GstVideoFilter parent;
    std::mutex *lock;
    guint quality, max_colors;       // Settings (colordetect/imp.rs:30-43)
    gboolean have_state;             // State { color_format, current_color } (:45-48)
    std::string *current_color;
    gboolean have_color;
};
struct GstColorDetectClass {
    GstVideoFilterClass parent_class;
};
G_DEFINE_TYPE(GstColorDetect, gst_color_detect, GST_TYPE_VIDEO_FILTER)

enum { PROP_C_0, PROP_QUALITY, PROP_MAX_COLORS };

static void gst_color_detect_set_property(GObject *obj, guint id, const GValue *value, GParamSpec *pspec)
{
    GstColorDetect *self = reinterpret_cast<GstColorDetect *>(obj);
    std::lock_guard<std::mutex> g(*self->lock);
    switch (id) {
    case PROP_QUALITY:
        GST_CAT_INFO_OBJECT(colordetect_debug, obj, "Changing quality from %u to %u", self->quality, g_value_get_uint(value));
        self->quality = g_value_get_uint(value);
        break;
    case PROP_MAX_COLORS:
        GST_CAT_INFO_OBJECT(colordetect_debug, obj, "Changing max_colors from %u to %u", self->max_colors, g_value_get_uint(value));
        self->max_colors = g_value_get_uint(value);
        break;
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID(obj, id, pspec); break;
    }
}

static void gst_color_detect_get_property(GObject *obj, guint id, GValue *value, GParamSpec *pspec)
{
    GstColorDetect *self = reinterpret_cast<GstColorDetect *>(obj);
    std::lock_guard<std::mutex> g(*self->lock);
    switch (id) {
    case PROP_QUALITY: g_value_set_uint(value, self->quality); break;
    case PROP_MAX_COLORS: g_value_set_uint(value, self->max_colors); break;
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID(obj, id, pspec); break;
    }
}

// VideoFilterImpl::set_info (colordetect/imp.rs:260-294): keeps the previous colour
static gboolean gst_color_detect_set_info(GstVideoFilter *filter, GstCaps *incaps, GstVideoInfo *, GstCaps *outcaps, GstVideoInfo *)
{
    GstColorDetect *self = reinterpret_cast<GstColorDetect *>(filter);
    GST_CAT_DEBUG_OBJECT(colordetect_debug, self, "Configured for caps %" GST_PTR_FORMAT " to %" GST_PTR_FORMAT, incaps, outcaps);
    self->have_state = TRUE;
    return TRUE;
}

static gboolean gst_color_detect_stop(GstBaseTransform *trans) // :252-256
{
    GstColorDetect *self = reinterpret_cast<GstColorDetect *>(trans);
    self->have_state = FALSE;
    self->have_color = FALSE;
    GST_CAT_INFO_OBJECT(colordetect_debug, self, "Stopped");
    return TRUE;
}

// color_changed (:88-112) preceded by the name-change test of detect_color (:76-85)
static void color_detect_report(GstColorDetect *self, const uint32_t *palette, uint32_t n)
{
    const char *name = mvfx_css_color_similar((palette[0] >> 16) & 0xff, (palette[0] >> 8) & 0xff, palette[0] & 0xff);
    if (self->have_color && *self->current_color == name)
        return;
    *self->current_color = name;
    self->have_color = TRUE;
    GST_CAT_DEBUG_OBJECT(colordetect_debug, self, "Dominant color changed to %s", name);
    GValue list = G_VALUE_INIT;
    g_value_init(&list, GST_TYPE_LIST);
    for (uint32_t i = 0; i < n; i++) {
        GValue v = G_VALUE_INIT;
        g_value_init(&v, G_TYPE_UINT);
        g_value_set_uint(&v, palette[i]);
        gst_value_list_append_and_take_value(&list, &v);
    }
    GstStructure *s = gst_structure_new("colordetect", "dominant-color", G_TYPE_STRING, name, NULL);
    gst_structure_take_value(s, "palette", &list);
    gst_element_post_message(GST_ELEMENT(self), gst_message_new_element(GST_OBJECT(self), s));
}

// transform_frame_ip_passthrough (:296-305) -> detect_color (:57-86)
static GstFlowReturn gst_color_detect_transform_frame_ip(GstVideoFilter *filter, GstVideoFrame *frame)
{
    GstColorDetect *self = reinterpret_cast<GstColorDetect *>(filter);
    if (!self->have_state) {
        GST_ELEMENT_ERROR(self, CORE, NEGOTIATION, ("Have no state yet"), (NULL));
        return GST_FLOW_NOT_NEGOTIATED;
    }
    guint quality, max_colors;
    {
        std::lock_guard<std::mutex> g(*self->lock);
        quality = self->quality;
        max_colors = self->max_colors;
    }
    const mvfx_frame f = mvfx_frame_from_gst(frame);
    uint32_t palette[256];
    uint32_t n = 0;
    const int rc = mvfx_colordetect_palette_host(&f, quality, max_colors, palette, &n);
    if (rc != MVFX_OK) {
        GST_CAT_ERROR_OBJECT(colordetect_debug, self, "%s", mvfx_last_error());
        return GST_FLOW_ERROR; // get_palette(..).map_err(|_| FlowError::Error) (:74)
    }
    color_detect_report(self, palette, n);
    return GST_FLOW_OK;
}

// Device-resident path: histogram straight from HBM, only 128 KB come back for the median cut
static GstFlowReturn gst_color_detect_bt_transform_ip(GstBaseTransform *bt, GstBuffer *buf)
{
    if (!mvfx_buffer_is_hip(buf))
        return GST_BASE_TRANSFORM_CLASS(gst_color_detect_parent_class)->transform_ip(bt, buf);
    GstColorDetect *self = reinterpret_cast<GstColorDetect *>(bt);
    GstVideoFilter *vf = GST_VIDEO_FILTER(bt);
    if (!vf->negotiated || !self->have_state)
        return GST_FLOW_NOT_NEGOTIATED;
    guint quality, max_colors;
    {
        std::lock_guard<std::mutex> g(*self->lock);
        quality = self->quality;
        max_colors = self->max_colors;
    }
    GstMapInfo map;
    mvfx_frame f;
    if (!mvfx_hip_map_frame(buf, &vf->in_info, GST_MAP_READ, &map, &f))
        return GST_FLOW_ERROR;
    uint32_t palette[256];
    uint32_t n = 0;
    mvfx_hip_buffer_acquire(buf, mvfx_thread_stream()); // the producer's fence; the call itself returns the palette (sync)
    const int rc = mvfx_colordetect_palette(&f, quality, max_colors, palette, &n, mvfx_thread_stream());
    gst_buffer_unmap(buf, &map);
    if (rc != MVFX_OK) {
        GST_CAT_ERROR_OBJECT(colordetect_debug, self, "%s", mvfx_last_error());
        return GST_FLOW_ERROR;
    }
    color_detect_report(self, palette, n);
    return GST_FLOW_OK;
}

static void gst_color_detect_finalize(GObject *obj)
{
    GstColorDetect *self = reinterpret_cast<GstColorDetect *>(obj);
    delete self->current_color;
    delete self->lock;
    G_OBJECT_CLASS(gst_color_detect_parent_class)->finalize(obj);
}

static void gst_color_detect_class_init(GstColorDetectClass *klass)
{
    GObjectClass *gobject = G_OBJECT_CLASS(klass);
    GstElementClass *element = GST_ELEMENT_CLASS(klass);
    GstBaseTransformClass *bt = GST_BASE_TRANSFORM_CLASS(klass);
    gobject->set_property = gst_color_detect_set_property;
    gobject->get_property = gst_color_detect_get_property;
    gobject->finalize = gst_color_detect_finalize;
    const GParamFlags flags = (GParamFlags)(G_PARAM_READWRITE | GST_PARAM_MUTABLE_PLAYING | G_PARAM_STATIC_STRINGS);
    g_object_class_install_property(gobject, PROP_QUALITY, // colordetect/imp.rs:126-133
        g_param_spec_uint("quality", "Quality of an output colors", "A step in pixels to improve performance", 0, 10, 10, flags));
    g_object_class_install_property(gobject, PROP_MAX_COLORS, // :134-141
        g_param_spec_uint("max-colors", "Number of colors in the output palette",
                          "Actual colors count can be lower depending on the image", 2, 255, 2, flags));
    gst_element_class_set_static_metadata(element, "Dominant color detection", "Filter/Video",
                                          "Detects the dominant color of a video", "Philippe Normand <philn@igalia.com>");
    static const gchar *const formats[] = {"RGB", "RGBA", "ARGB", "BGR", "BGRA", NULL}; // :212-240
    mvfx_add_pad_templates(element, mvfx_caps_plus_hip(mvfx_video_caps(formats)), mvfx_caps_plus_hip(mvfx_video_caps(formats)));
    bt->transform_ip = gst_color_detect_bt_transform_ip;
    bt->passthrough_on_same_caps = TRUE;   // :246-250
    bt->transform_ip_on_passthrough = TRUE;
    bt->stop = gst_color_detect_stop;
    GST_VIDEO_FILTER_CLASS(klass)->set_info = gst_color_detect_set_info;
    GST_VIDEO_FILTER_CLASS(klass)->transform_frame_ip = gst_color_detect_transform_frame_ip;
}

static void gst_color_detect_init(GstColorDetect *self)
{
    self->lock = new std::mutex();
    self->current_color = new std::string();
    self->quality = 10;
    self->max_colors = 2;
    self->have_state = FALSE;
    self->have_color = FALSE;
}

// ------------------------------------------------------------------------- roundedcorners

struct GstRoundedCorners {
    GstBaseTransform parent;
    std::mutex *lock;
    guint border_radius_px; // Settings { border_radius_px, changed } (border/imp.rs:27-43)
    gboolean changed;
    GstMemory *alpha_mem;   // State { alpha_mem, out_info } (:45-48); a memory:HIPMemory block on the device path
    GstVideoInfo out_info;
    GstVideoInfo in_info;
    gboolean hip;           // negotiated on memory:HIPMemory: the A420 frame is composed in HBM
    gboolean have_state;
};
struct GstRoundedCornersClass {
    GstBaseTransformClass parent_class;
};
G_DEFINE_TYPE(GstRoundedCorners, gst_rounded_corners, GST_TYPE_BASE_TRANSFORM)

enum { PROP_R_0, PROP_BORDER_RADIUS };

static void gst_rounded_corners_set_property(GObject *obj, guint id, const GValue *value, GParamSpec *pspec)
{
    GstRoundedCorners *self = reinterpret_cast<GstRoundedCorners *>(obj);
    if (id != PROP_BORDER_RADIUS) { G_OBJECT_WARN_INVALID_PROPERTY_ID(obj, id, pspec); return; }
    gboolean reconfigure = FALSE;
    {
        std::lock_guard<std::mutex> g(*self->lock);
        const guint r = g_value_get_uint(value);
        if (self->border_radius_px != r) { // border/imp.rs:299-310
            self->changed = TRUE;
            GST_CAT_INFO_OBJECT(roundedcorners_debug, obj, "Changing border radius from %u to %u", self->border_radius_px, r);
            self->border_radius_px = r;
            reconfigure = TRUE;
        }
    }
    if (reconfigure)
        gst_base_transform_reconfigure_src(GST_BASE_TRANSFORM(self));
}

static void gst_rounded_corners_get_property(GObject *obj, guint id, GValue *value, GParamSpec *pspec)
{
    GstRoundedCorners *self = reinterpret_cast<GstRoundedCorners *>(obj);
    if (id != PROP_BORDER_RADIUS) { G_OBJECT_WARN_INVALID_PROPERTY_ID(obj, id, pspec); return; }
    std::lock_guard<std::mutex> g(*self->lock);
    g_value_set_uint(value, self->border_radius_px);
}

static gboolean gst_rounded_corners_stop(GstBaseTransform *trans) // :380-386
{
    GstRoundedCorners *self = reinterpret_cast<GstRoundedCorners *>(trans);
    std::lock_guard<std::mutex> g(*self->lock);
    if (self->alpha_mem) gst_memory_unref(self->alpha_mem);
    self->alpha_mem = nullptr;
    self->have_state = FALSE;
    GST_CAT_INFO_OBJECT(roundedcorners_debug, self, "Stopped");
    return TRUE;
}

// transform_caps (:388-442)
static GstCaps *gst_rounded_corners_transform_caps(GstBaseTransform *trans, GstPadDirection direction, GstCaps *caps, GstCaps *filter)
{
    GstRoundedCorners *self = reinterpret_cast<GstRoundedCorners *>(trans);
    GstCaps *other = gst_caps_copy(caps);
    if (direction == GST_PAD_SRC) {
        for (guint i = 0; i < gst_caps_get_size(other); i++)
            gst_structure_set(gst_caps_get_structure(other, i), "format", G_TYPE_STRING, "I420", NULL);
    } else {
        guint radius;
        {
            std::lock_guard<std::mutex> g(*self->lock);
            radius = self->border_radius_px;
        }
        for (guint i = 0; i < gst_caps_get_size(other); i++) {
            GstStructure *s = gst_caps_get_structure(other, i);
            if (radius == 0) {
                GValue list = G_VALUE_INIT;
                g_value_init(&list, GST_TYPE_LIST);
                for (const char *f : {"I420", "A420"}) {
                    GValue v = G_VALUE_INIT;
                    g_value_init(&v, G_TYPE_STRING);
                    g_value_set_string(&v, f);
                    gst_value_list_append_and_take_value(&list, &v);
                }
                gst_structure_take_value(s, "format", &list);
            } else {
                gst_structure_set(s, "format", G_TYPE_STRING, "A420", NULL);
            }
        }
    }
    GST_CAT_DEBUG_OBJECT(roundedcorners_debug, self, "Transformed caps from %" GST_PTR_FORMAT " to %" GST_PTR_FORMAT " in direction %d",
                         caps, other, (int)direction);
    if (filter) {
        GstCaps *r = gst_caps_intersect_full(filter, other, GST_CAPS_INTERSECT_FIRST);
        gst_caps_unref(other);
        return r;
    }
    return other;
}

// set_caps (:444-480)
static gboolean gst_rounded_corners_set_caps(GstBaseTransform *trans, GstCaps *incaps, GstCaps *outcaps)
{
    GstRoundedCorners *self = reinterpret_cast<GstRoundedCorners *>(trans);
    GstVideoInfo info;
    if (!gst_video_info_from_caps(&info, outcaps)) {
        GST_CAT_ERROR_OBJECT(roundedcorners_debug, self, "Failed to parse output caps");
        return FALSE;
    }
    GST_CAT_DEBUG_OBJECT(roundedcorners_debug, self, "Configured for caps %" GST_PTR_FORMAT " to %" GST_PTR_FORMAT, incaps, outcaps);
    if (GST_VIDEO_INFO_FORMAT(&info) == GST_VIDEO_FORMAT_I420) {
        gst_base_transform_set_passthrough(trans, TRUE);
        return TRUE;
    }
    gst_base_transform_set_passthrough(trans, FALSE);
    const guint ru2_height = (GST_VIDEO_INFO_HEIGHT(&info) + 1) & ~1u;
    const gsize alpha_size = (gsize)GST_VIDEO_INFO_PLANE_STRIDE(&info, 3) * ru2_height;
    std::lock_guard<std::mutex> g(*self->lock);
    if (self->alpha_mem) gst_memory_unref(self->alpha_mem);
    self->hip = mvfx_caps_has_hip_feature(outcaps);
    if (self->hip) {
        GstAllocator *alloc = mvfx_hip_allocator_get();
        self->alpha_mem = gst_allocator_alloc(alloc, alpha_size, NULL);
        gst_object_unref(alloc);
        if (!self->alpha_mem || !gst_video_info_from_caps(&self->in_info, incaps)) {
            GST_CAT_ERROR_OBJECT(roundedcorners_debug, self, "Failed to set up the device alpha plane: %s", mvfx_last_error());
            return FALSE;
        }
    } else {
        self->alpha_mem = gst_allocator_alloc(NULL, alpha_size, NULL);
    }
    self->out_info = info;
    self->have_state = TRUE;
    self->changed = TRUE;
    return TRUE;
}

// generate_alpha_mask (:108-180): the mask is rendered by libcairo with the reference's call sequence
// (mvfx_roundedcorners_mask*, host/cairo_mask.cpp) into the shared alpha GstMemory -- system memory on the
// reference's path, HBM on the memory:HIPMemory path
static gboolean rounded_corners_generate_mask(GstRoundedCorners *self, guint radius)
{
    // make_mut(): the memory may be shared with in-flight buffers -> copy on write
    if (!gst_mini_object_is_writable(GST_MINI_OBJECT_CAST(self->alpha_mem))) {
        GstMemory *copy = gst_memory_copy(self->alpha_mem, 0, -1);
        gst_memory_unref(self->alpha_mem);
        self->alpha_mem = copy;
    }
    GstMapInfo map;
    if (!gst_memory_map(self->alpha_mem, &map, (GstMapFlags)(GST_MAP_WRITE | (self->hip ? MVFX_MAP_HIP : 0)))) {
        GST_CAT_ERROR_OBJECT(roundedcorners_debug, self, "Failed to map alpha memory as writable");
        return FALSE;
    }
    const uint32_t mw = (uint32_t)GST_VIDEO_INFO_WIDTH(&self->out_info), mh = (uint32_t)GST_VIDEO_INFO_HEIGHT(&self->out_info),
                   ms = (uint32_t)GST_VIDEO_INFO_PLANE_STRIDE(&self->out_info, 3);
    int rc;
    if (self->hip) { // uploaded once to where the compose kernel reads it (synchronous)
        rc = mvfx_roundedcorners_mask(map.data, mw, mh, ms, radius, mvfx_thread_stream());
    } else {
        rc = mvfx_roundedcorners_mask_host(map.data, mw, mh, ms, radius);
    }
    gst_memory_unmap(self->alpha_mem, &map);
    if (rc != MVFX_OK) {
        GST_CAT_ERROR_OBJECT(roundedcorners_debug, self, "Failed to draw rounded corners: %s", mvfx_last_error());
        return FALSE;
    }
    return TRUE;
}

// prepare_output_buffer (:482-559) + add_video_meta (:182-268)
static GstFlowReturn gst_rounded_corners_prepare_output_buffer(GstBaseTransform *trans, GstBuffer *inbuf, GstBuffer **outbuf)
{
    GstRoundedCorners *self = reinterpret_cast<GstRoundedCorners *>(trans);
    if (gst_base_transform_is_passthrough(trans)) {
        *outbuf = inbuf;
        return GST_FLOW_OK;
    }
    std::lock_guard<std::mutex> g(*self->lock);
    if (!self->have_state) {
        GST_ELEMENT_ERROR(self, CORE, NEGOTIATION, ("Have no state yet"), (NULL));
        return GST_FLOW_NOT_NEGOTIATED;
    }
    if (self->hip && mvfx_buffer_is_hip(inbuf) && GST_VIDEO_INFO_FORMAT(&self->out_info) != GST_VIDEO_FORMAT_I420) {
        // the mask lives where the frames live (round 6): the streaming thread adopts the device of the incoming memory, and a mask that was
        // made on another device -- set_caps ran before the first buffer, or the stream changed device -- is made again there
        // (d3d12colorlut/imp.rs:494-542 rebuilds its context the same way)
        if (!mvfx_hip_follow_device(inbuf, GST_OBJECT(self))) return GST_FLOW_ERROR;
        if (mvfx_hip_memory_device(self->alpha_mem) != mvfx_hip_buffer_device(inbuf)) {
            GstAllocator *alloc = mvfx_hip_allocator_get();
            GstMemory *fresh = gst_allocator_alloc(alloc, self->alpha_mem->maxsize, NULL);
            gst_object_unref(alloc);
            if (!fresh) {
                GST_ELEMENT_ERROR(self, RESOURCE, NO_SPACE_LEFT, ("%s", mvfx_last_error()), (NULL));
                return GST_FLOW_ERROR;
            }
            gst_memory_unref(self->alpha_mem);
            self->alpha_mem = fresh;
            self->changed = TRUE;
        }
    }
    if (self->changed) {
        self->changed = FALSE;
        GST_CAT_DEBUG_OBJECT(roundedcorners_debug, self, "Caps or border radius changed, generating alpha mask");
        if (GST_VIDEO_INFO_FORMAT(&self->out_info) == GST_VIDEO_FORMAT_I420) {
            *outbuf = inbuf;
            return GST_FLOW_OK;
        }
        if (!rounded_corners_generate_mask(self, self->border_radius_px)) {
            GST_ELEMENT_ERROR(self, CORE, NEGOTIATION, ("Failed to generate alpha mask"), (NULL));
            return GST_FLOW_NOT_NEGOTIATED;
        }
    }
    if (self->hip) {
        // device path: one A420 buffer in HBM, Y/U/V + the mask copied by ONE kernel launch (the zero-copy
        // "append the shared alpha memory" trick of the system-memory path would leave a two-memory buffer that
        // no device kernel downstream could address as one frame)
        if (!mvfx_buffer_is_hip(inbuf)) {
            GST_ELEMENT_ERROR(self, CORE, NEGOTIATION, ("negotiated memory:HIPMemory but got a system-memory buffer"), (NULL));
            return GST_FLOW_NOT_NEGOTIATED;
        }
        GstBuffer *out = NULL;
        GstFlowReturn fr = mvfx_hip_new_output(trans, inbuf, GST_VIDEO_INFO_SIZE(&self->out_info), &out);
        if (fr != GST_FLOW_OK)
            return fr;
        GstMapInfo imap, omap, amap;
        if (!gst_buffer_map(inbuf, &imap, (GstMapFlags)(GST_MAP_READ | MVFX_MAP_HIP))) {
            gst_buffer_unref(out);
            return GST_FLOW_ERROR;
        }
        if (!gst_buffer_map(out, &omap, (GstMapFlags)(GST_MAP_WRITE | MVFX_MAP_HIP))) {
            gst_buffer_unmap(inbuf, &imap);
            gst_buffer_unref(out);
            return GST_FLOW_ERROR;
        }
        if (!gst_memory_map(self->alpha_mem, &amap, (GstMapFlags)(GST_MAP_READ | MVFX_MAP_HIP))) {
            gst_buffer_unmap(out, &omap);
            gst_buffer_unmap(inbuf, &imap);
            gst_buffer_unref(out);
            return GST_FLOW_ERROR;
        }
        mvfx_planar_frame pi, po;
        memset(&pi, 0, sizeof(pi));
        memset(&po, 0, sizeof(po));
        for (guint p = 0; p < 3; p++) {
            pi.data[p] = imap.data + GST_VIDEO_INFO_PLANE_OFFSET(&self->in_info, p);
            pi.stride[p] = (uint32_t)GST_VIDEO_INFO_PLANE_STRIDE(&self->in_info, p);
        }
        for (guint p = 0; p < 4; p++) {
            po.data[p] = omap.data + GST_VIDEO_INFO_PLANE_OFFSET(&self->out_info, p);
            po.stride[p] = (uint32_t)GST_VIDEO_INFO_PLANE_STRIDE(&self->out_info, p);
        }
        pi.width = po.width = (uint32_t)GST_VIDEO_INFO_WIDTH(&self->out_info);
        pi.height = po.height = (uint32_t)GST_VIDEO_INFO_HEIGHT(&self->out_info);
        pi.format = MVFX_FORMAT_I420;
        po.format = MVFX_FORMAT_A420;
        // per frame, as the hsv / colorlut elements pick it (the mask was uploaded synchronously).  It pays only with a few output blocks in
        // rotation (mvfx_hip_decide_allocation's pool minimum): 4K, one stream 39-45 k fps, alternating 50-64 k; with a pool that recycled
        // one block the alternation LOST a third
        const mvfx_stream st = mvfx_element_stream(inbuf);
        mvfx_hip_buffer_acquire(inbuf, st);
        mvfx_hip_buffer_acquire(out, st);
        MvfxFenceScope fs; // one fence for both buffers (the reader's too: the input block may be recycled and overwritten next), on the kernel
        mvfx_hip_fence_begin_buffers(&fs, inbuf, out, st);
        int rc = mvfx_roundedcorners_compose_a420(&pi, amap.data, (uint32_t)GST_VIDEO_INFO_PLANE_STRIDE(&self->out_info, 3), &po, st);
        mvfx_hip_fence_end(&fs, st, NULL, GST_OBJECT(self));
        gst_memory_unmap(self->alpha_mem, &amap);
        gst_buffer_unmap(out, &omap);
        gst_buffer_unmap(inbuf, &imap);
        if (rc != MVFX_OK) {
            gst_buffer_unref(out);
            return MVFX_GST_FLOW(self, rc);
        }
        *outbuf = out;
        return GST_FLOW_OK;
    }
    GstBuffer *buf;
    if (gst_buffer_is_writable(inbuf)) {
        buf = inbuf;
    } else {
        buf = gst_buffer_copy(inbuf);
    }
    const gsize alpha_plane_offset = gst_buffer_get_size(buf);
    gst_buffer_append_memory(buf, gst_memory_ref(self->alpha_mem));

    gint strides[GST_VIDEO_MAX_PLANES] = {0, 0, 0, 0};
    gsize offsets[GST_VIDEO_MAX_PLANES] = {0, 0, 0, 0};
    GstVideoFrameFlags vflags = GST_VIDEO_FRAME_FLAG_NONE;
    GstVideoMeta *meta = gst_buffer_get_video_meta(buf);
    if (meta) {
        vflags = meta->flags;
        for (guint p = 0; p < meta->n_planes && p < 3; p++) {
            offsets[p] = meta->offset[p];
            strides[p] = meta->stride[p];
        }
        if (GST_META_FLAG_IS_SET(GST_META_CAST(meta), GST_META_FLAG_LOCKED)) {
            // border/imp.rs:202-224: a locked (pool-owned) meta cannot be removed -> new buffer with the memories,
            // flags and timestamps of this one and none of its metas
            GstBuffer *fresh = gst_buffer_copy_region(
                buf, (GstBufferCopyFlags)(GST_BUFFER_COPY_FLAGS | GST_BUFFER_COPY_TIMESTAMPS | GST_BUFFER_COPY_MEMORY), 0, -1);
            if (buf != inbuf) gst_buffer_unref(buf);
            buf = fresh;
        } else {
            gst_buffer_remove_meta(buf, GST_META_CAST(meta));
        }
    } else {
        for (guint p = 0; p < 3; p++) {
            offsets[p] = GST_VIDEO_INFO_PLANE_OFFSET(&self->out_info, p);
            strides[p] = GST_VIDEO_INFO_PLANE_STRIDE(&self->out_info, p);
        }
    }
    offsets[3] = alpha_plane_offset;
    strides[3] = GST_VIDEO_INFO_PLANE_STRIDE(&self->out_info, 3);
    gst_buffer_add_video_meta_full(buf, vflags, GST_VIDEO_INFO_FORMAT(&self->out_info), GST_VIDEO_INFO_WIDTH(&self->out_info),
                                   GST_VIDEO_INFO_HEIGHT(&self->out_info), 4, offsets, strides);
    *outbuf = buf;
    return GST_FLOW_OK;
}

static GstFlowReturn gst_rounded_corners_transform_ip(GstBaseTransform *, GstBuffer *) { return GST_FLOW_OK; } // :561-563

static gboolean gst_rounded_corners_propose_allocation(GstBaseTransform *trans, GstQuery *decide_query, GstQuery *query) // :565-572
{
    gst_query_add_allocation_meta(query, GST_VIDEO_META_API_TYPE, NULL);
    return GST_BASE_TRANSFORM_CLASS(gst_rounded_corners_parent_class)->propose_allocation(trans, decide_query, query);
}

static void gst_rounded_corners_finalize(GObject *obj)
{
    GstRoundedCorners *self = reinterpret_cast<GstRoundedCorners *>(obj);
    if (self->alpha_mem) gst_memory_unref(self->alpha_mem);
    delete self->lock;
    G_OBJECT_CLASS(gst_rounded_corners_parent_class)->finalize(obj);
}

static void gst_rounded_corners_class_init(GstRoundedCornersClass *klass)
{
    GObjectClass *gobject = G_OBJECT_CLASS(klass);
    GstElementClass *element = GST_ELEMENT_CLASS(klass);
    GstBaseTransformClass *bt = GST_BASE_TRANSFORM_CLASS(klass);
    gobject->set_property = gst_rounded_corners_set_property;
    gobject->get_property = gst_rounded_corners_get_property;
    gobject->finalize = gst_rounded_corners_finalize;
    g_object_class_install_property(gobject, PROP_BORDER_RADIUS, // border/imp.rs:279-294
        g_param_spec_uint("border-radius-px", "Border radius in pixels", "Draw rounded corners with given border radius",
                          0, G_MAXUINT, 0, (GParamFlags)(G_PARAM_READWRITE | GST_PARAM_MUTABLE_PLAYING | G_PARAM_STATIC_STRINGS)));
    gst_element_class_set_static_metadata(element, "Rounded Corners", "Filter/Effect/Converter/Video",
                                          "Adds rounded corners to video", "Sanchayan Maity <sanchayan@asymptotic.io>");
    static const gchar *const sink_formats[] = {"I420", NULL};
    static const gchar *const src_formats[] = {"I420", "A420", NULL};
    mvfx_add_pad_templates(element, mvfx_caps_plus_hip(mvfx_video_caps(sink_formats)), mvfx_caps_plus_hip(mvfx_video_caps(src_formats))); // :343-370 (+ HIP twin)
    bt->stop = gst_rounded_corners_stop;
    bt->transform_caps = gst_rounded_corners_transform_caps;
    bt->set_caps = gst_rounded_corners_set_caps;
    bt->prepare_output_buffer = gst_rounded_corners_prepare_output_buffer;
    bt->transform_ip = gst_rounded_corners_transform_ip; // AlwaysInPlace (:374-378)
    bt->propose_allocation = gst_rounded_corners_propose_allocation;
}

static void gst_rounded_corners_init(GstRoundedCorners *self)
{
    self->lock = new std::mutex();
    self->border_radius_px = 0;
    self->changed = FALSE;
    self->alpha_mem = nullptr;
    self->hip = FALSE;
    self->have_state = FALSE;
    gst_base_transform_set_in_place(GST_BASE_TRANSFORM(self), TRUE);
}


// ------------------------------------------------------------------------- videocompare

GST_DEBUG_CATEGORY_STATIC(videocompare_debug);

// enum HashAlgorithm (videocompare/mod.rs:57-92) = mvfx_hash_algo; `dssim` is a non-default cargo feature there

static GType gst_video_compare_hash_algorithm_get_type(void)
{
    static gsize type = 0;
    if (g_once_init_enter(&type)) {
        static const GEnumValue values[] = {
            {MVFX_HASH_MEAN, "Mean: The Mean hashing algorithm.", "mean"},
            {MVFX_HASH_GRADIENT, "Gradient: The Gradient hashing algorithm.", "gradient"},
            {MVFX_HASH_VERTGRADIENT, "VertGradient: The Vertical-Gradient hashing algorithm.", "vertgradient"},
            {MVFX_HASH_DOUBLEGRADIENT, "DoubleGradient: The Double-Gradient hashing algorithm.", "doublegradient"},
            {MVFX_HASH_BLOCKHASH, "Blockhash: The Blockhash (block median value perceptual hash) algorithm.", "blockhash"},
            {MVFX_HASH_DSSIM, "Dssim: Image similarity comparison simulating human perception.", "dssim"},
            {0, NULL, NULL}};
        g_once_init_leave(&type, g_enum_register_static("GstVideoCompareHashAlgorithm", values));
    }
    return (GType)type;
}

struct GstVideoCompare {
    GstAggregator parent;
    std::mutex *lock;
    gint hash_algo;                  // Settings (videocompare/imp.rs:27-43)
    gdouble max_distance_threshold;
    GstPad *reference_pad;           // first requested sink pad (imp.rs:210-233), not ref-counted
};
struct GstVideoCompareClass {
    GstAggregatorClass parent_class;
};
G_DEFINE_TYPE(GstVideoCompare, gst_video_compare, GST_TYPE_AGGREGATOR)

enum { PROP_V_0, PROP_HASH_ALGO, PROP_MAX_DIST };

static void gst_video_compare_set_property(GObject *obj, guint id, const GValue *value, GParamSpec *pspec)
{
    GstVideoCompare *self = reinterpret_cast<GstVideoCompare *>(obj);
    std::lock_guard<std::mutex> g(*self->lock);
    switch (id) {
    case PROP_HASH_ALGO:
        GST_CAT_INFO_OBJECT(videocompare_debug, obj, "Changing hash-algo from %d to %d", self->hash_algo, g_value_get_enum(value));
        self->hash_algo = g_value_get_enum(value);
        break;
    case PROP_MAX_DIST:
        GST_CAT_INFO_OBJECT(videocompare_debug, obj, "Changing max-dist-threshold from %f to %f", self->max_distance_threshold, g_value_get_double(value));
        self->max_distance_threshold = g_value_get_double(value);
        break;
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID(obj, id, pspec); break;
    }
}

static void gst_video_compare_get_property(GObject *obj, guint id, GValue *value, GParamSpec *pspec)
{
    GstVideoCompare *self = reinterpret_cast<GstVideoCompare *>(obj);
    std::lock_guard<std::mutex> g(*self->lock);
    switch (id) {
    case PROP_HASH_ALGO: g_value_set_enum(value, self->hash_algo); break;
    case PROP_MAX_DIST: g_value_set_double(value, self->max_distance_threshold); break;
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID(obj, id, pspec); break;
    }
}

// AggregatorImpl::create_new_pad (imp.rs:210-233): the first sink pad is the reference
static GstAggregatorPad *gst_video_compare_create_new_pad(GstAggregator *agg, GstPadTemplate *templ, const gchar *req_name, const GstCaps *caps)
{
    GstVideoCompare *self = reinterpret_cast<GstVideoCompare *>(agg);
    GstAggregatorPad *pad = GST_AGGREGATOR_CLASS(gst_video_compare_parent_class)->create_new_pad(agg, templ, req_name, caps);
    if (pad) {
        std::lock_guard<std::mutex> g(*self->lock);
        if (!self->reference_pad && GST_PAD_DIRECTION(pad) == GST_PAD_SINK) {
            GST_CAT_INFO_OBJECT(videocompare_debug, self, "Reference sink pad selected: %s", GST_PAD_NAME(pad));
            self->reference_pad = GST_PAD(pad);
        }
    }
    return pad;
}

// ElementImpl::release_pad (imp.rs:188-206): re-pick a reference when it goes away
static void gst_video_compare_release_pad(GstElement *element, GstPad *pad)
{
    GstVideoCompare *self = reinterpret_cast<GstVideoCompare *>(element);
    {
        std::lock_guard<std::mutex> g(*self->lock);
        if (self->reference_pad == pad) {
            self->reference_pad = nullptr;
            for (GList *l = element->sinkpads; l; l = l->next)
                if (l->data != pad)
                    self->reference_pad = GST_PAD(l->data);
        }
    }
    GST_ELEMENT_CLASS(gst_video_compare_parent_class)->release_pad(element, pad);
}

// update_src_caps (imp.rs:235-255): the src caps are the reference pad's caps
static GstFlowReturn gst_video_compare_update_src_caps(GstAggregator *agg, GstCaps *caps, GstCaps **ret)
{
    GstVideoCompare *self = reinterpret_cast<GstVideoCompare *>(agg);
    GstCaps *sink_caps = nullptr;
    {
        std::lock_guard<std::mutex> g(*self->lock);
        if (self->reference_pad)
            sink_caps = gst_pad_get_current_caps(self->reference_pad);
    }
    if (!sink_caps)
        sink_caps = gst_caps_ref(caps); // allow any caps for now
    if (!gst_caps_can_intersect(sink_caps, caps)) {
        GST_CAT_ERROR_OBJECT(videocompare_debug, self, "Proposed src caps (%" GST_PTR_FORMAT ") not supported, needs to intersect with the reference sink caps (%" GST_PTR_FORMAT ")", caps, sink_caps);
        gst_caps_unref(sink_caps);
        return GST_FLOW_NOT_NEGOTIATED;
    }
    GST_CAT_INFO_OBJECT(videocompare_debug, self, "Caps for src pad: %" GST_PTR_FORMAT, sink_caps);
    *ret = sink_caps;
    return GST_FLOW_OK;
}

// One pad's buffer as the C ABI's frame view.  memory:HIPMemory buffers are mapped to their DEVICE pointer (no copy,
// `device` set); system-memory buffers through GstVideoFrame (strides from GstVideoMeta honoured).
struct CompareView {
    mvfx_frame f;
    gboolean device, mapped_frame, mapped_hip;
    GstVideoFrame frame;
    GstMapInfo map;
    GstBuffer *buf;
};

static int compare_view_open(GstPad *pad, GstBuffer *buf, CompareView *v)
{
    v->device = v->mapped_frame = v->mapped_hip = FALSE;
    v->buf = buf;
    GstCaps *caps = gst_pad_get_current_caps(pad);
    GstVideoInfo info;
    if (!caps || !gst_video_info_from_caps(&info, caps)) {
        if (caps) gst_caps_unref(caps);
        return MVFX_ERR_NOT_NEGOTIATED;
    }
    gst_caps_unref(caps);
    if (mvfx_buffer_is_hip(buf)) {
        if (!mvfx_hip_map_frame(buf, &info, GST_MAP_READ, &v->map, &v->f))
            return MVFX_ERR_INVALID_ARGUMENT;
        v->device = v->mapped_hip = TRUE;
        mvfx_hip_buffer_acquire(buf, mvfx_thread_stream()); // the producer's fence; the hash / distance calls below synchronise
        GST_CAT_LOG_OBJECT(videocompare_debug, pad, "frame of %" GST_PTR_FORMAT " stays in device memory", pad);
        return MVFX_OK;
    }
    if (!gst_video_frame_map(&v->frame, &info, buf, GST_MAP_READ))
        return MVFX_ERR_INVALID_ARGUMENT;
    v->mapped_frame = TRUE;
    v->f = mvfx_frame_from_gst(&v->frame);
    return MVFX_OK;
}

static void compare_view_close(CompareView *v)
{
    if (v->mapped_hip) gst_buffer_unmap(v->buf, &v->map);
    if (v->mapped_frame) gst_video_frame_unmap(&v->frame);
    v->mapped_hip = v->mapped_frame = FALSE;
}

// HashedImage::Dssim + compare (hashed_image.rs:49-59,72-75): both frames mapped, one distance
static int video_compare_ssim(GstPad *ref_pad, GstBuffer *ref_buf, GstPad *pad, GstBuffer *buf, double *distance,
                              gboolean *size_mismatch)
{
    CompareView a, b;
    int rc = compare_view_open(ref_pad, ref_buf, &a);
    if (rc != MVFX_OK) return rc;
    rc = compare_view_open(pad, buf, &b);
    if (rc != MVFX_OK) { compare_view_close(&a); return rc; }
    if (a.f.width != b.f.width || a.f.height != b.f.height)
        *size_mismatch = TRUE;
    else if (a.device && b.device)
        rc = mvfx_ssim_distance(&a.f, &b.f, distance, mvfx_thread_stream()); // both frames already in HBM
    else if (!a.device && !b.device)
        rc = mvfx_ssim_distance_host(&a.f, &b.f, distance);
    else
        rc = MVFX_ERR_NOT_NEGOTIATED; // one pad on device memory, the other on system memory
    compare_view_close(&b);
    compare_view_close(&a);
    return rc;
}

// HasherEngine::hash_image (hashed_image.rs:24-64) on one pad's buffer with the ImageHasher algorithms
static int video_compare_hash(GstPad *pad, GstBuffer *buf, gint algo, uint64_t *hash, guint *w, guint *h)
{
    CompareView v;
    int rc = compare_view_open(pad, buf, &v);
    if (rc != MVFX_OK) return rc;
    *w = v.f.width;
    *h = v.f.height;
    if (algo == MVFX_HASH_BLOCKHASH)
        rc = v.device ? mvfx_blockhash(&v.f, hash, mvfx_thread_stream()) : mvfx_blockhash_host(&v.f, hash);
    else // mean / gradient / vertgradient / doublegradient
        rc = v.device ? mvfx_image_hash(&v.f, algo, hash, NULL, mvfx_thread_stream()) : mvfx_image_hash_host(&v.f, algo, hash, NULL);
    compare_view_close(&v);
    return rc;
}

// VideoAggregatorImpl::aggregate_frames (imp.rs:259-389)
static GstFlowReturn gst_video_compare_aggregate(GstAggregator *agg, gboolean timeout)
{
    GstVideoCompare *self = reinterpret_cast<GstVideoCompare *>(agg);
    GstPad *reference_pad;
    gint algo;
    gdouble threshold;
    {
        std::lock_guard<std::mutex> g(*self->lock);
        reference_pad = self->reference_pad;
        algo = self->hash_algo;
        threshold = self->max_distance_threshold;
    }
    if (!reference_pad) {
        GST_CAT_WARNING_OBJECT(videocompare_debug, self, "No reference sink pad exists");
        return GST_FLOW_EOS;
    }
    if (algo < MVFX_HASH_MEAN || algo > MVFX_HASH_DSSIM) {
        GST_ELEMENT_ERROR(self, LIBRARY, SETTINGS, ("unknown hash-algo %d", algo), (NULL));
        return GST_FLOW_ERROR;
    }
    GstAggregatorPad *ref_apad = GST_AGGREGATOR_PAD(reference_pad);
    GstBuffer *ref_buf = gst_aggregator_pad_pop_buffer(ref_apad);
    if (!ref_buf) {
        if (gst_aggregator_pad_is_eos(ref_apad))
            return GST_FLOW_EOS;
        GST_CAT_WARNING_OBJECT(videocompare_debug, self, "The reference sink pad '%s' has not produced a buffer, image comparison not possible", GST_PAD_NAME(reference_pad));
        return GST_FLOW_OK;
    }
    // running time of the reference buffer (imp.rs:300-306)
    guint64 running_time = GST_CLOCK_TIME_NONE;
    if (GST_BUFFER_PTS_IS_VALID(ref_buf) && ref_apad->segment.format == GST_FORMAT_TIME)
        running_time = gst_segment_to_running_time(&ref_apad->segment, GST_FORMAT_TIME, GST_BUFFER_PTS(ref_buf));

    uint64_t ref_hash = 0;
    guint rw = 0, rh = 0;
    int rc = algo == MVFX_HASH_DSSIM ? MVFX_OK : video_compare_hash(reference_pad, ref_buf, algo, &ref_hash, &rw, &rh);
    if (rc != MVFX_OK) {
        gst_buffer_unref(ref_buf);
        return MVFX_GST_FLOW(self, rc);
    }

    GValue distances = G_VALUE_INIT;
    g_value_init(&distances, GST_TYPE_ARRAY);
    gboolean any_below = FALSE;
    GstFlowReturn ret = GST_FLOW_OK;
    GList *pads = nullptr;
    GST_OBJECT_LOCK(self);
    for (GList *l = GST_ELEMENT(self)->sinkpads; l; l = l->next)
        pads = g_list_prepend(pads, gst_object_ref(l->data));
    GST_OBJECT_UNLOCK(self);
    pads = g_list_reverse(pads);
    for (GList *l = pads; l && ret == GST_FLOW_OK; l = l->next) {
        GstPad *pad = GST_PAD(l->data);
        if (pad == reference_pad)
            continue; // do not compare the reference pad with itself
        GstBuffer *buf = gst_aggregator_pad_pop_buffer(GST_AGGREGATOR_PAD(pad));
        if (!buf)
            break; // imp.rs:331-334: no frame on this pad yet
        uint64_t hash = 0;
        guint w = 0, h = 0;
        gdouble ssim_distance = 0.0;
        gboolean size_mismatch = FALSE;
        if (algo == MVFX_HASH_DSSIM)
            rc = video_compare_ssim(reference_pad, ref_buf, pad, buf, &ssim_distance, &size_mismatch);
        else
            rc = video_compare_hash(pad, buf, algo, &hash, &w, &h);
        gst_buffer_unref(buf);
        if (rc == MVFX_OK && (size_mismatch || w != rw || h != rh)) { // imp.rs:337-346
            GST_CAT_ERROR_OBJECT(videocompare_debug, self, "Video streams do not have the same sizes (add videoscale and force the sizes to be equal on all sink pads)");
            ret = GST_FLOW_NOT_NEGOTIATED;
            break;
        }
        if (rc != MVFX_OK) {
            ret = MVFX_GST_FLOW(self, rc);
            break;
        }
        const gdouble distance = algo == MVFX_HASH_DSSIM ? ssim_distance // hashed_image.rs:72-75
                                                         : (gdouble)mvfx_hash_distance(ref_hash, hash); // hashed_image.rs:70
        if (distance <= threshold)
            any_below = TRUE;
        GstStructure *pd = gst_structure_new("pad-distance", "pad", GST_TYPE_PAD, pad, "distance", G_TYPE_DOUBLE, distance, NULL); // mod.rs:162-169
        GValue v = G_VALUE_INIT;
        g_value_init(&v, GST_TYPE_STRUCTURE);
        g_value_take_boxed(&v, pd);
        gst_value_array_append_and_take_value(&distances, &v);
    }
    g_list_free_full(pads, gst_object_unref);

    if (ret == GST_FLOW_OK && any_below) { // imp.rs:356-377, message layout mod.rs:110-123
        GstStructure *s = gst_structure_new("videocompare", "running-time", G_TYPE_UINT64, running_time, NULL);
        gst_structure_take_value(s, "pad-distances", &distances);
        GST_CAT_DEBUG_OBJECT(videocompare_debug, self, "Image detected %" GST_TIME_FORMAT, GST_TIME_ARGS(running_time));
        gst_element_post_message(GST_ELEMENT(self), gst_message_new_element(GST_OBJECT(self), s));
    } else {
        g_value_unset(&distances);
    }
    if (ret != GST_FLOW_OK) {
        gst_buffer_unref(ref_buf);
        return ret;
    }
    // output the reference buffer (imp.rs:308-313)
    return gst_aggregator_finish_buffer(agg, ref_buf);
}

static void gst_video_compare_finalize(GObject *obj)
{
    delete reinterpret_cast<GstVideoCompare *>(obj)->lock;
    G_OBJECT_CLASS(gst_video_compare_parent_class)->finalize(obj);
}

static void gst_video_compare_class_init(GstVideoCompareClass *klass)
{
    GObjectClass *gobject = G_OBJECT_CLASS(klass);
    GstElementClass *element = GST_ELEMENT_CLASS(klass);
    GstAggregatorClass *agg = GST_AGGREGATOR_CLASS(klass);
    gobject->set_property = gst_video_compare_set_property;
    gobject->get_property = gst_video_compare_get_property;
    gobject->finalize = gst_video_compare_finalize;
    const GParamFlags flags = (GParamFlags)(G_PARAM_READWRITE | GST_PARAM_MUTABLE_READY | G_PARAM_STATIC_STRINGS);
    g_object_class_install_property(gobject, PROP_HASH_ALGO, // imp.rs:78-83
        g_param_spec_enum("hash-algo", "Hashing Algorithm", "Which hashing algorithm to use for image comparisons",
                          gst_video_compare_hash_algorithm_get_type(), MVFX_HASH_BLOCKHASH, flags));
    g_object_class_install_property(gobject, PROP_MAX_DIST, // imp.rs:84-89
        g_param_spec_double("max-dist-threshold", "Maximum Distance Threshold",
                            "Maximum distance threshold to emit messages when an image is detected, by default emits only on exact match",
                            0.0, G_MAXDOUBLE, 0.0, flags));
    gst_element_class_set_static_metadata(element, "Image comparison", "Filter/Video", "Compare similarity of video frames",
                                          "Rafael Caricio <rafael@caricio.com>"); // imp.rs:145-156
    static const gchar *const formats[] = {"RGB", "RGBA", NULL}; // imp.rs:158-186
    GstCaps *caps = mvfx_caps_plus_hip(mvfx_video_caps(formats)); // + the memory:HIPMemory twin
    gst_element_class_add_pad_template(element, gst_pad_template_new_with_gtype("sink_%u", GST_PAD_SINK, GST_PAD_REQUEST, caps, GST_TYPE_AGGREGATOR_PAD));
    gst_element_class_add_pad_template(element, gst_pad_template_new_with_gtype("src", GST_PAD_SRC, GST_PAD_ALWAYS, caps, GST_TYPE_AGGREGATOR_PAD));
    gst_caps_unref(caps);
    element->release_pad = gst_video_compare_release_pad;
    agg->create_new_pad = gst_video_compare_create_new_pad;
    agg->update_src_caps = gst_video_compare_update_src_caps;
    agg->aggregate = gst_video_compare_aggregate;
}

static void gst_video_compare_init(GstVideoCompare *self)
{
    self->lock = new std::mutex();
    self->hash_algo = MVFX_HASH_BLOCKHASH;
    self->max_distance_threshold = 0.0;
    self->reference_pad = nullptr;
}

// ------------------------------------------------------------------------- plugin (videofx/src/lib.rs:25-48)

static gboolean plugin_init(GstPlugin *plugin)
{
    GST_DEBUG_CATEGORY_INIT(colordetect_debug, "colordetect", 0, "Dominant color detection");
    GST_DEBUG_CATEGORY_INIT(roundedcorners_debug, "roundedcorners", 0, "Rounded corners");
    GST_DEBUG_CATEGORY_INIT(videocompare_debug, "videocompare", 0, "Video frames comparison");
    return gst_element_register(plugin, "roundedcorners", GST_RANK_NONE, gst_rounded_corners_get_type()) &&
           gst_element_register(plugin, "colordetect", GST_RANK_NONE, gst_color_detect_get_type()) &&
           gst_element_register(plugin, "videocompare", GST_RANK_NONE, gst_video_compare_get_type());
}

GST_PLUGIN_DEFINE(GST_VERSION_MAJOR, GST_VERSION_MINOR, rsvideofx, "GStreamer Rust Video Effects Plugin", plugin_init,
                  MVFX_GST_VERSION, "MPL", "gst-plugin-videofx", MVFX_GST_ORIGIN)
