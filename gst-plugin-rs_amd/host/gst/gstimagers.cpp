// libgstimagers.so -- plugin `imagers` with element `imagersoverlay`.
// Same surface as video/image/src/{lib.rs,overlay/imp.rs}: factory, GType, klass, the twelve properties, the
// positioning rules of update_composition (imp.rs:84-191), passthrough logic of start / before_transform
// (imp.rs:527-597) and the two per-frame behaviours of transform_frame_ip (imp.rs:703-727): attach a
// GstVideoOverlayCompositionMeta when downstream takes it, else blend -- the blend runs in the HIP kernel behind
// mvfx_overlay_blend{,_host} (csrc/overlay_kernels.hip, arithmetic pinned against libgstvideo).
//
// Differences, stated: the reference decodes `location` with the `image` crate (any format it knows); this build decodes
// PNG through libpng's simplified API (BGRA8 straight out of png_image_finish_read).  Pad templates list the ten packed RGB
// formats the blend kernel handles instead of every raw format.  A rectangle whose render size differs from the image
// (overlay-width / overlay-height) is scaled ONCE per composition change by libgstvideo's own
// gst_video_blend_scale_linear_RGBA (what gst_video_overlay_composition_blend would do per frame), then blended unscaled.
#include "mvfx_gst_common.h"

#include <gst/video/video-overlay-composition.h>

#include <mutex>
#include <string>

#if __has_include(<png.h>)
#include <png.h>
#define MVFX_HAVE_PNG 1
#else
#define MVFX_HAVE_PNG 0
#endif

GST_DEBUG_CATEGORY_STATIC(imagersoverlay_debug); // overlay/imp.rs:18-24

enum PositioningMode { POS_RELATIVE_TO_EDGES = 0, POS_ABSOLUTE = 1 }; // overlay/imp.rs:35-42

static GType positioning_mode_get_type(void)
{
    static gsize once = 0;
    static GType type = 0;
    if (g_once_init_enter(&once)) {
        static const GEnumValue values[] = {{POS_RELATIVE_TO_EDGES, "PixelsRelativeToEdges", "pixels-relative-to-edges"},
                                            {POS_ABSOLUTE, "PixelsAbsolute", "pixels-absolute"},
                                            {0, NULL, NULL}};
        type = g_enum_register_static("GstImageRsOverlayPositioningMode", values);
        g_once_init_leave(&once, 1);
    }
    return type;
}

struct Settings { // overlay/imp.rs:44-76
    std::string location;
    bool has_location = false;
    gint offset_x = 0, offset_y = 0;
    gdouble relative_x = 0, relative_y = 0, coef_x = 0, coef_y = 0;
    gint positioning_mode = POS_RELATIVE_TO_EDGES;
    guint overlay_width = 0, overlay_height = 0;
    gfloat alpha = 1.0f;
    guint64 max_alloc = 0;
};

struct GstImageRsOverlay {
    GstVideoFilter parent;
    std::mutex *lock; // state + settings (the reference locks state, then settings)
    Settings *settings;
    // State (overlay/imp.rs:26-33)
    std::string *loaded_location;
    bool has_loaded;
    GstBuffer *image;            // decoded BGRA pixels + GstVideoMeta (load_image :193-283)
    GstVideoOverlayComposition *composition;
    GstBuffer *render;           // pixels at render size (== image unless overlay-width/height scale it)
    gint comp_x, comp_y;
    guint render_w, render_h;
    void *render_dev;            // device copy of `render` for memory:HIPMemory frames
    gsize render_dev_size;
    bool update_composition, allow_attaching;
};
struct GstImageRsOverlayClass {
    GstVideoFilterClass parent_class;
};
G_DEFINE_TYPE(GstImageRsOverlay, gst_image_rs_overlay, GST_TYPE_VIDEO_FILTER)

enum { PROP_0, PROP_LOCATION, PROP_OFFSET_X, PROP_OFFSET_Y, PROP_RELATIVE_X, PROP_RELATIVE_Y, PROP_OVERLAY_WIDTH, PROP_OVERLAY_HEIGHT,
       PROP_ALPHA, PROP_MAX_ALLOC, PROP_POSITIONING_MODE, PROP_COEF_X, PROP_COEF_Y };

static void gst_image_rs_overlay_set_property(GObject *obj, guint id, const GValue *value, GParamSpec *pspec)
{
    GstImageRsOverlay *self = reinterpret_cast<GstImageRsOverlay *>(obj);
    std::lock_guard<std::mutex> g(*self->lock);
    Settings &s = *self->settings;
    switch (id) { // overlay/imp.rs:398-466: everything but max-alloc-bytes marks the composition stale
    case PROP_LOCATION: {
        const gchar *v = g_value_get_string(value);
        s.has_location = v != NULL;
        s.location = v ? v : "";
        break;
    }
    case PROP_OFFSET_X: s.offset_x = g_value_get_int(value); break;
    case PROP_OFFSET_Y: s.offset_y = g_value_get_int(value); break;
    case PROP_RELATIVE_X: s.relative_x = g_value_get_double(value); break;
    case PROP_RELATIVE_Y: s.relative_y = g_value_get_double(value); break;
    case PROP_COEF_X: s.coef_x = g_value_get_double(value); break;
    case PROP_COEF_Y: s.coef_y = g_value_get_double(value); break;
    case PROP_POSITIONING_MODE: s.positioning_mode = g_value_get_enum(value); break;
    case PROP_OVERLAY_WIDTH: s.overlay_width = g_value_get_uint(value); break;
    case PROP_OVERLAY_HEIGHT: s.overlay_height = g_value_get_uint(value); break;
    case PROP_ALPHA: s.alpha = g_value_get_float(value); break;
    case PROP_MAX_ALLOC: s.max_alloc = g_value_get_uint64(value); return;
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID(obj, id, pspec); return;
    }
    self->update_composition = true;
}

static void gst_image_rs_overlay_get_property(GObject *obj, guint id, GValue *value, GParamSpec *pspec)
{
    GstImageRsOverlay *self = reinterpret_cast<GstImageRsOverlay *>(obj);
    std::lock_guard<std::mutex> g(*self->lock);
    const Settings &s = *self->settings;
    switch (id) {
    case PROP_LOCATION: g_value_set_string(value, s.has_location ? s.location.c_str() : NULL); break;
    case PROP_OFFSET_X: g_value_set_int(value, s.offset_x); break;
    case PROP_OFFSET_Y: g_value_set_int(value, s.offset_y); break;
    case PROP_RELATIVE_X: g_value_set_double(value, s.relative_x); break;
    case PROP_RELATIVE_Y: g_value_set_double(value, s.relative_y); break;
    case PROP_COEF_X: g_value_set_double(value, s.coef_x); break;
    case PROP_COEF_Y: g_value_set_double(value, s.coef_y); break;
    case PROP_POSITIONING_MODE: g_value_set_enum(value, s.positioning_mode); break;
    case PROP_OVERLAY_WIDTH: g_value_set_uint(value, s.overlay_width); break;
    case PROP_OVERLAY_HEIGHT: g_value_set_uint(value, s.overlay_height); break;
    case PROP_ALPHA: g_value_set_float(value, s.alpha); break;
    case PROP_MAX_ALLOC: g_value_set_uint64(value, s.max_alloc); break;
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID(obj, id, pspec); break;
    }
}

static void drop_composition(GstImageRsOverlay *self)
{
    if (self->composition) gst_video_overlay_composition_unref(self->composition);
    self->composition = NULL;
    if (self->render) gst_buffer_unref(self->render);
    self->render = NULL;
}

// load_image (overlay/imp.rs:193-283); lock held.  Returns FALSE after posting an element error.
static gboolean load_image(GstImageRsOverlay *self)
{
    const Settings &s = *self->settings;
    if (self->has_loaded == s.has_location && (!s.has_location || *self->loaded_location == s.location))
        return TRUE; // "No need to update"
    self->has_loaded = false;
    if (self->image) gst_buffer_unref(self->image);
    self->image = NULL;
    if (!s.has_location) {
        GST_CAT_DEBUG_OBJECT(imagersoverlay_debug, self, "No location set");
        return TRUE;
    }
#if MVFX_HAVE_PNG
    png_image img;
    memset(&img, 0, sizeof(img));
    img.version = PNG_IMAGE_VERSION;
    if (!png_image_begin_read_from_file(&img, s.location.c_str())) {
        // io error vs decode error as the reference distinguishes them (ResourceError::OpenRead / StreamError::Decode)
        if (g_file_test(s.location.c_str(), G_FILE_TEST_EXISTS))
            GST_ELEMENT_ERROR(self, STREAM, DECODE, ("Could not decode overlay image container: %s", img.message), (NULL));
        else
            GST_ELEMENT_ERROR(self, RESOURCE, OPEN_READ, ("Could not load overlay image: %s", img.message), (NULL));
        return FALSE;
    }
    img.format = PNG_FORMAT_BGRA; // the reference swaps RGBA -> BGRA after decoding (imp.rs:257-262)
    const gsize stride = (gsize)img.width * 4, size = stride * img.height;
    if (s.max_alloc != 0 && size > s.max_alloc) {
        png_image_free(&img);
        GST_ELEMENT_ERROR(self, STREAM, DECODE, ("Could not decode overlay image container: memory limit of %" G_GUINT64_FORMAT " bytes exceeded",
                                                 s.max_alloc), (NULL));
        return FALSE;
    }
    guint8 *pixels = (guint8 *)g_malloc(size ? size : 1);
    if (!png_image_finish_read(&img, NULL, pixels, (png_int_32)stride, NULL)) {
        g_free(pixels);
        GST_ELEMENT_ERROR(self, STREAM, DECODE, ("Could not decode overlay image container: %s", img.message), (NULL));
        return FALSE;
    }
    GstBuffer *buf = gst_buffer_new_wrapped(pixels, size);
    gsize offsets[GST_VIDEO_MAX_PLANES] = {0, 0, 0, 0};
    gint strides[GST_VIDEO_MAX_PLANES] = {(gint)stride, 0, 0, 0};
    gst_buffer_add_video_meta_full(buf, GST_VIDEO_FRAME_FLAG_NONE, GST_VIDEO_FORMAT_BGRA, img.width, img.height, 1, offsets, strides);
    self->image = buf;
    *self->loaded_location = s.location;
    self->has_loaded = true;
    self->update_composition = true;
    GST_CAT_INFO_OBJECT(imagersoverlay_debug, self, "Updated pixbuf, %u x %u", img.width, img.height);
    return TRUE;
#else
    GST_ELEMENT_ERROR(self, STREAM, DECODE, ("Could not decode overlay image container: this build has no PNG decoder (png.h missing)"), (NULL));
    return FALSE;
#endif
}

// update_composition (overlay/imp.rs:84-191); lock held
static void update_composition(GstImageRsOverlay *self)
{
    if (!self->update_composition) return;
    GstVideoFilter *vf = GST_VIDEO_FILTER(self);
    const gint64 video_width = GST_VIDEO_INFO_WIDTH(&vf->in_info), video_height = GST_VIDEO_INFO_HEIGHT(&vf->in_info);
    drop_composition(self);
    self->update_composition = false;
    const Settings &s = *self->settings;
    if (s.alpha == 0.0f || !self->image) return;
    GstVideoMeta *meta = gst_buffer_get_video_meta(self->image);
    const gint64 width = s.overlay_width == 0 ? meta->width : s.overlay_width;
    const gint64 height = s.overlay_height == 0 ? meta->height : s.overlay_height;
    // `(f64) as i64` saturates in Rust; the products are bounded by +-2^31 here
    const gint64 rx = (gint64)(s.relative_x * (gdouble)video_width), ry = (gint64)(s.relative_y * (gdouble)video_height);
    gint64 x, y;
    if (s.positioning_mode == POS_ABSOLUTE) {
        x = (gint64)s.offset_x + rx + (gint64)(s.coef_x * (gdouble)video_width);
        y = (gint64)s.offset_y + ry + (gint64)(s.coef_y * (gdouble)video_height);
    } else {
        x = s.offset_x < 0 ? video_width + (gint64)s.offset_x - width + rx : (gint64)s.offset_x + rx;
        y = s.offset_y < 0 ? video_height + (gint64)s.offset_y - height + ry : (gint64)s.offset_y + ry;
    }
    x = CLAMP(x, (gint64)G_MININT32, (gint64)G_MAXINT32);
    y = CLAMP(y, (gint64)G_MININT32, (gint64)G_MAXINT32);
    GST_CAT_DEBUG_OBJECT(imagersoverlay_debug, self, "overlay rendered: %" G_GINT64_FORMAT " x %" G_GINT64_FORMAT " @ %" G_GINT64_FORMAT
                         ",%" G_GINT64_FORMAT " (onto %" G_GINT64_FORMAT " x %" G_GINT64_FORMAT ")", width, height, x, y, video_width, video_height);
    GstVideoOverlayRectangle *rect = gst_video_overlay_rectangle_new_raw(self->image, (gint)x, (gint)y, (guint)width, (guint)height,
                                                                         GST_VIDEO_OVERLAY_FORMAT_FLAG_NONE);
    if (s.alpha != 1.0f) gst_video_overlay_rectangle_set_global_alpha(rect, s.alpha);
    self->composition = gst_video_overlay_composition_new(rect);
    // the pixels the blend kernel reads: the image itself, or -- render size != image size -- what libgstvideo's blend would
    // scale to on every frame (gst_video_overlay_rectangle_get_pixels_unscaled_raw vs the scaled getter), produced once here
    // FLAG_GLOBAL_ALPHA: "I handle the global alpha myself" -- the pixels come back unmodified and the kernel applies alpha,
    // exactly what gst_video_overlay_composition_blend does before it calls gst_video_blend(.., global_alpha)
    GstBuffer *px = gst_video_overlay_rectangle_get_pixels_raw(rect, GST_VIDEO_OVERLAY_FORMAT_FLAG_GLOBAL_ALPHA);
    self->render = gst_buffer_ref(px);
    GstVideoMeta *rmeta = gst_buffer_get_video_meta(px);
    self->render_w = rmeta ? rmeta->width : (guint)width;
    self->render_h = rmeta ? rmeta->height : (guint)height;
    self->comp_x = (gint)x;
    self->comp_y = (gint)y;
    gst_video_overlay_rectangle_unref(rect);
    self->render_dev_size = 0; // device copy is stale
    GST_CAT_DEBUG_OBJECT(imagersoverlay_debug, self, "Composition updated");
}

static gboolean gst_image_rs_overlay_start(GstBaseTransform *trans) // overlay/imp.rs:527-553
{
    GstImageRsOverlay *self = reinterpret_cast<GstImageRsOverlay *>(trans);
    bool empty;
    {
        std::lock_guard<std::mutex> g(*self->lock);
        empty = !self->settings->has_location;
    }
    if (empty) GST_CAT_INFO_OBJECT(imagersoverlay_debug, self, "no image location set, doing nothing");
    gst_base_transform_set_passthrough(trans, empty);
    return TRUE;
}

static gboolean gst_image_rs_overlay_stop(GstBaseTransform *trans) // overlay/imp.rs:555-563
{
    GstImageRsOverlay *self = reinterpret_cast<GstImageRsOverlay *>(trans);
    std::lock_guard<std::mutex> g(*self->lock);
    drop_composition(self);
    if (self->image) gst_buffer_unref(self->image);
    self->image = NULL;
    self->has_loaded = false;
    self->loaded_location->clear();
    mvfx_device_free(self->render_dev);
    self->render_dev = NULL;
    self->render_dev_size = 0;
    GST_CAT_DEBUG_OBJECT(imagersoverlay_debug, self, "Image removed");
    return TRUE;
}

static void gst_image_rs_overlay_before_transform(GstBaseTransform *trans, GstBuffer *inbuf) // overlay/imp.rs:565-597
{
    GstImageRsOverlay *self = reinterpret_cast<GstImageRsOverlay *>(trans);
    const GstClockTime stream_time = gst_segment_to_stream_time(&trans->segment, GST_FORMAT_TIME, GST_BUFFER_PTS(inbuf));
    if (GST_CLOCK_TIME_IS_VALID(stream_time)) gst_object_sync_values(GST_OBJECT(self), stream_time);
    bool passthrough;
    {
        std::lock_guard<std::mutex> g(*self->lock);
        if (!load_image(self)) return;
        if (!self->update_composition) return;
        update_composition(self);
        passthrough = self->composition == NULL; // so that the buffer is writable when it reaches transform_ip
    }
    gst_base_transform_set_passthrough(trans, passthrough);
}

static GstCaps *gst_image_rs_overlay_transform_caps(GstBaseTransform *trans, GstPadDirection direction, GstCaps *caps, GstCaps *filter)
{ // overlay/imp.rs:599-688
    const gchar *const feat = GST_CAPS_FEATURE_META_GST_VIDEO_OVERLAY_COMPOSITION;
    GstCaps *with = gst_caps_new_empty(), *kept = gst_caps_new_empty();
    for (guint i = 0; i < gst_caps_get_size(caps); i++) {
        GstStructure *st = gst_caps_get_structure(caps, i);
        GstCapsFeatures *f = gst_caps_get_features(caps, i);
        GstCapsFeatures *nf = gst_caps_features_copy(f);
        if (direction == GST_PAD_SINK) {
            if (!gst_caps_features_is_any(nf) && !gst_caps_features_contains(nf, feat)) gst_caps_features_add(nf, feat);
        } else {
            gst_caps_features_remove(nf, feat);
        }
        gst_caps_append_structure_full(with, gst_structure_copy(st), nf);
        if (gst_caps_features_is_any(f) || gst_caps_features_contains(f, GST_CAPS_FEATURE_MEMORY_SYSTEM_MEMORY) ||
            gst_caps_features_contains(f, MVFX_CAPS_FEATURE_MEMORY_HIP) || gst_caps_features_contains(f, feat))
            gst_caps_append_structure_full(kept, gst_structure_copy(st), gst_caps_features_copy(f));
    }
    GstCaps *tmp = direction == GST_PAD_SINK ? gst_caps_merge(with, kept) : gst_caps_merge(kept, with);
    GST_CAT_DEBUG_OBJECT(imagersoverlay_debug, trans, "filter %" GST_PTR_FORMAT ", expanded caps %" GST_PTR_FORMAT, filter, tmp);
    if (filter) {
        GstCaps *r = gst_caps_intersect_full(filter, tmp, GST_CAPS_INTERSECT_FIRST);
        gst_caps_unref(tmp);
        return r;
    }
    return tmp;
}

static gboolean gst_image_rs_overlay_set_caps(GstBaseTransform *trans, GstCaps *incaps, GstCaps *outcaps) // overlay/imp.rs:690-699
{
    if (!GST_BASE_TRANSFORM_CLASS(gst_image_rs_overlay_parent_class)->set_caps(trans, incaps, outcaps)) return FALSE;
    GstImageRsOverlay *self = reinterpret_cast<GstImageRsOverlay *>(trans);
    GstCapsFeatures *f = gst_caps_get_features(outcaps, 0);
    std::lock_guard<std::mutex> g(*self->lock);
    self->allow_attaching = f == NULL || gst_caps_features_contains(f, GST_CAPS_FEATURE_META_GST_VIDEO_OVERLAY_COMPOSITION);
    self->update_composition = true; // the positions depend on the video size
    return TRUE;
}

// transform_frame_ip (overlay/imp.rs:703-727) is reached through BaseTransform::transform_ip here so that device buffers
// (memory:HIPMemory) never get CPU-mapped by GstVideoFilter
static GstFlowReturn gst_image_rs_overlay_transform_ip(GstBaseTransform *trans, GstBuffer *buf)
{
    GstImageRsOverlay *self = reinterpret_cast<GstImageRsOverlay *>(trans);
    GstVideoFilter *vf = GST_VIDEO_FILTER(trans);
    if (!vf->negotiated) return GST_FLOW_NOT_NEGOTIATED;
    std::lock_guard<std::mutex> g(*self->lock);
    if (!self->composition) return GST_FLOW_OK;
    if (self->allow_attaching) {
        gst_buffer_add_video_overlay_composition_meta(buf, self->composition);
        return GST_FLOW_OK;
    }
    GstMapInfo omap;
    if (!gst_buffer_map(self->render, &omap, GST_MAP_READ)) return GST_FLOW_ERROR;
    mvfx_frame overlay = {omap.data, self->render_w, self->render_h, self->render_w * 4, MVFX_FORMAT_BGRA};
    const gfloat alpha = self->settings->alpha;
    int rc;
    if (mvfx_buffer_is_hip(buf)) {
        GstMapInfo map;
        mvfx_frame f;
        if (!mvfx_hip_map_frame(buf, &vf->in_info, GST_MAP_READWRITE, &map, &f)) { gst_buffer_unmap(self->render, &omap); return GST_FLOW_ERROR; }
        rc = MVFX_OK;
        if (self->render_dev_size != omap.size) { // the rectangle's pixels go to HBM once per composition
            mvfx_device_free(self->render_dev);
            self->render_dev = NULL;
            rc = mvfx_device_alloc(&self->render_dev, omap.size);
            if (rc == MVFX_OK) rc = mvfx_copy_to_device(self->render_dev, omap.data, omap.size, NULL);
            self->render_dev_size = rc == MVFX_OK ? omap.size : 0;
        }
        if (rc == MVFX_OK) {
            overlay.data = self->render_dev;
            mvfx_stream st = mvfx_thread_stream();
            mvfx_hip_buffer_acquire(buf, st);
            rc = mvfx_overlay_blend(&f, &overlay, self->comp_x, self->comp_y, alpha, st);
            mvfx_hip_buffer_release(buf, st);
        }
        gst_buffer_unmap(buf, &map);
    } else {
        GstVideoFrame frame;
        if (!gst_video_frame_map(&frame, &vf->in_info, buf, GST_MAP_READWRITE)) { gst_buffer_unmap(self->render, &omap); return GST_FLOW_ERROR; }
        mvfx_frame f = mvfx_frame_from_gst(&frame);
        rc = mvfx_overlay_blend_host(&f, &overlay, self->comp_x, self->comp_y, alpha);
        gst_video_frame_unmap(&frame);
    }
    gst_buffer_unmap(self->render, &omap);
    if (rc != MVFX_OK) {
        GST_CAT_ERROR_OBJECT(imagersoverlay_debug, self, "Blending failed: %s", mvfx_last_error());
        return GST_FLOW_ERROR;
    }
    return GST_FLOW_OK;
}

MVFX_DEFINE_HIP_ALLOCATION_VFUNCS(gst_image_rs_overlay, gst_image_rs_overlay_parent_class)

static void gst_image_rs_overlay_finalize(GObject *obj)
{
    GstImageRsOverlay *self = reinterpret_cast<GstImageRsOverlay *>(obj);
    drop_composition(self);
    if (self->image) gst_buffer_unref(self->image);
    mvfx_device_free(self->render_dev);
    delete self->loaded_location;
    delete self->settings;
    delete self->lock;
    G_OBJECT_CLASS(gst_image_rs_overlay_parent_class)->finalize(obj);
}

static void gst_image_rs_overlay_class_init(GstImageRsOverlayClass *klass)
{
    GObjectClass *gobject = G_OBJECT_CLASS(klass);
    GstElementClass *element = GST_ELEMENT_CLASS(klass);
    GstBaseTransformClass *bt = GST_BASE_TRANSFORM_CLASS(klass);
    gobject->set_property = gst_image_rs_overlay_set_property;
    gobject->get_property = gst_image_rs_overlay_get_property;
    gobject->finalize = gst_image_rs_overlay_finalize;
    const GParamFlags playing = (GParamFlags)(G_PARAM_READWRITE | GST_PARAM_CONTROLLABLE | GST_PARAM_MUTABLE_PLAYING);
    const GParamFlags ready = (GParamFlags)(G_PARAM_READWRITE | GST_PARAM_MUTABLE_READY);
    g_object_class_install_property(gobject, PROP_LOCATION, g_param_spec_string("location", "location", "Location of image file to overlay", NULL, playing));
    g_object_class_install_property(gobject, PROP_OFFSET_X, g_param_spec_int("offset-x", "X Offset",
        "For positive value, horizontal offset of overlay image in pixels from left of video image. For negative value, horizontal offset of overlay image in pixels from right of video image",
        G_MININT, G_MAXINT, 0, playing));
    g_object_class_install_property(gobject, PROP_OFFSET_Y, g_param_spec_int("offset-y", "Y Offset",
        "For positive value, vertical offset of overlay image in pixels from top of video image. For negative value, vertical offset of overlay image in pixels from bottom of video image",
        G_MININT, G_MAXINT, 0, playing));
    g_object_class_install_property(gobject, PROP_RELATIVE_X, g_param_spec_double("relative-x", "Relative X Offset",
        "Horizontal offset of overlay image in fractions of video image width, from top-left corner of video image (in relative positioning)", -1.0, 1.0, 0.0, playing));
    g_object_class_install_property(gobject, PROP_RELATIVE_Y, g_param_spec_double("relative-y", "Relative Y Offset",
        "Vertical offset of overlay image in fractions of video image width, from top-left corner of video image (in relative positioning)", -1.0, 1.0, 0.0, playing));
    g_object_class_install_property(gobject, PROP_OVERLAY_WIDTH, g_param_spec_uint("overlay-width", "Overlay Width",
        "Width of overlay image in pixels (0 = same as overlay image)", 0, G_MAXUINT, 0, playing));
    g_object_class_install_property(gobject, PROP_OVERLAY_HEIGHT, g_param_spec_uint("overlay-height", "Overlay Height",
        "Height of overlay image in pixels (0 = same as overlay image", 0, G_MAXUINT, 0, playing));
    g_object_class_install_property(gobject, PROP_ALPHA, g_param_spec_float("alpha", "Alpha", "Global alpha of overlay image", 0.0f, 1.0f, 1.0f, playing));
    g_object_class_install_property(gobject, PROP_MAX_ALLOC, g_param_spec_uint64("max-alloc-bytes", "Memory allocation limits",
        "Max. amount of data to allocate for decoding (bytes, 0=disable)", 0, G_MAXUINT64, 0, ready));
    g_object_class_install_property(gobject, PROP_POSITIONING_MODE, g_param_spec_enum("positioning-mode", "Positioning mode",
        "Positioning mode of offset-x and offset-y properties", positioning_mode_get_type(), POS_RELATIVE_TO_EDGES, ready));
    g_object_class_install_property(gobject, PROP_COEF_X, g_param_spec_double("coef-x", "Relative X Offset",
        "Horizontal offset of overlay image in fractions of video image width, from top-left corner of video image (absolute positioning)", -1.0, 1.0, 0.0, playing));
    g_object_class_install_property(gobject, PROP_COEF_Y, g_param_spec_double("coef-y", "Relative Y Offset",
        "Vertical offset of overlay image in fractions of video image height, from top-left corner of video image (absolute positioning)", -1.0, 1.0, 0.0, playing));

    gst_element_class_set_static_metadata(element, "image-rs overlay", "Video/Overlay",
                                          "Renders images decoded with image-rs over raw video frames", "Amyspark <amy@centricular.com>");
    // the ten packed RGB formats the blend kernel handles, each with ANY features (overlay-composition meta, HIP memory) and plain
    static const gchar *const formats[] = {"RGBx", "xRGB", "BGRx", "xBGR", "RGBA", "ARGB", "BGRA", "ABGR", "RGB", "BGR", NULL};
    GstCaps *plain = mvfx_video_caps(formats);
    GstCaps *any = gst_caps_copy(plain);
    gst_caps_set_features(any, 0, gst_caps_features_new_any());
    gst_caps_append(any, plain);
    mvfx_add_pad_templates(element, gst_caps_ref(any), any);

    bt->start = gst_image_rs_overlay_start;
    bt->stop = gst_image_rs_overlay_stop;
    bt->before_transform = gst_image_rs_overlay_before_transform;
    bt->transform_caps = gst_image_rs_overlay_transform_caps;
    bt->set_caps = gst_image_rs_overlay_set_caps;
    bt->transform_ip = gst_image_rs_overlay_transform_ip;
    bt->propose_allocation = gst_image_rs_overlay_propose_allocation;
    bt->decide_allocation = gst_image_rs_overlay_decide_allocation;
    bt->passthrough_on_same_caps = FALSE;    // BaseTransformImpl consts (imp.rs:521-525)
    bt->transform_ip_on_passthrough = FALSE;
}

static void gst_image_rs_overlay_init(GstImageRsOverlay *self)
{
    self->lock = new std::mutex();
    self->settings = new Settings();
    self->loaded_location = new std::string();
    self->has_loaded = false;
    self->image = NULL;
    self->composition = NULL;
    self->render = NULL;
    self->render_dev = NULL;
    self->render_dev_size = 0;
    self->comp_x = self->comp_y = 0;
    self->render_w = self->render_h = 0;
    self->update_composition = false;
    self->allow_attaching = false;
}

static gboolean plugin_init(GstPlugin *plugin)
{
    GST_DEBUG_CATEGORY_INIT(imagersoverlay_debug, "imagersoverlay", 0, "image-rs overlay");
    return gst_element_register(plugin, "imagersoverlay", GST_RANK_NONE, gst_image_rs_overlay_get_type());
}

GST_PLUGIN_DEFINE(GST_VERSION_MAJOR, GST_VERSION_MINOR, imagers, "GStreamer plugin based on image-rs", plugin_init, MVFX_GST_VERSION, "MPL",
                  "gst-plugin-image", MVFX_GST_ORIGIN)
