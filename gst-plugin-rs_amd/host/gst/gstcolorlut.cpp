// libgstcolorlut.so -- plugin `colorlut` with element `colorlut`.
// Same surface as video/colorlut/src/{lib.rs,colorlut/imp.rs}; LUT application runs in the HIP
// kernels behind include/mi355vfx.h, .cube parsing in host/cube_parser.cpp.
#include "mvfx_gst_common.h"
#include "mvfx_pair_hold.h"

#include <mutex>
#include <string>

GST_DEBUG_CATEGORY_STATIC(colorlut_debug); // colorlut/imp.rs:41-43

struct GstColorLut {
    GstVideoFilter parent;
    std::mutex *lock;
    std::string *location; // Settings { location: Option<String> } (:45-48); empty + !has_location = None
    gboolean has_location;
    mvfx_cube_lut *lut;    // State { lut: Option<CubeLut> } (:50-53)
    MvfxPairHold *hold;    // pair launches on device buffers (mvfx_pair_hold.h); the LUT lives from start() to stop(), which flushes first
};
struct GstColorLutClass {
    GstVideoFilterClass parent_class;
};
G_DEFINE_TYPE(GstColorLut, gst_color_lut, GST_TYPE_VIDEO_FILTER)

enum { PROP_0, PROP_LOCATION };

static void gst_color_lut_set_property(GObject *obj, guint id, const GValue *value, GParamSpec *pspec)
{
    GstColorLut *self = reinterpret_cast<GstColorLut *>(obj);
    if (id != PROP_LOCATION) { G_OBJECT_WARN_INVALID_PROPERTY_ID(obj, id, pspec); return; }
    std::lock_guard<std::mutex> g(*self->lock);
    const gchar *s = g_value_get_string(value);
    self->has_location = s != NULL;
    *self->location = s ? s : "";
}

static void gst_color_lut_get_property(GObject *obj, guint id, GValue *value, GParamSpec *pspec)
{
    GstColorLut *self = reinterpret_cast<GstColorLut *>(obj);
    if (id != PROP_LOCATION) { G_OBJECT_WARN_INVALID_PROPERTY_ID(obj, id, pspec); return; }
    std::lock_guard<std::mutex> g(*self->lock);
    g_value_set_string(value, self->has_location ? self->location->c_str() : NULL);
}

// BaseTransformImpl::start (colorlut/imp.rs:168-194)
static gboolean gst_color_lut_start(GstBaseTransform *trans)
{
    GstColorLut *self = reinterpret_cast<GstColorLut *>(trans);
    std::string location;
    {
        std::lock_guard<std::mutex> g(*self->lock);
        if (!self->has_location) {
            GST_ELEMENT_ERROR(self, RESOURCE, SETTINGS, ("LUT file location is not configured"), (NULL));
            return FALSE;
        }
        location = *self->location;
    }
    mvfx_cube_lut *lut = nullptr;
    if (mvfx_cube_lut_parse_file(location.c_str(), &lut) != MVFX_OK) {
        GST_ELEMENT_ERROR(self, RESOURCE, READ, ("%s", mvfx_last_error()), (NULL));
        return FALSE;
    }
    GST_CAT_TRACE_OBJECT(colorlut_debug, self, "Parsed LUT: %s size %u", mvfx_cube_lut_is_3d(lut) ? "3D" : "1D",
                         mvfx_cube_lut_size(lut));
    std::lock_guard<std::mutex> g(*self->lock);
    if (self->lut) mvfx_cube_lut_free(self->lut);
    self->lut = lut;
    return TRUE;
}

static int gst_color_lut_pair_launch(GstObject *element, const mvfx_frame *in, const mvfx_frame *out, uint32_t n, mvfx_stream st)
{
    GstColorLut *self = reinterpret_cast<GstColorLut *>(element);
    return n == 1 ? mvfx_colorlut_transform_frame(self->lut, in, out, st) : mvfx_colorlut_transform_frames(self->lut, in, out, n, st);
}

MVFX_PAIR_DEFINE_OPS(gst_color_lut, GstColorLut, gst_color_lut_pair_launch)

// EOS, flush-start: nothing stays held back across them
static gboolean gst_color_lut_sink_event(GstBaseTransform *bt, GstEvent *event)
{
    if (GST_EVENT_TYPE(event) == GST_EVENT_EOS || GST_EVENT_TYPE(event) == GST_EVENT_FLUSH_START)
        gst_color_lut_flush_cb(GST_OBJECT(bt));
    return GST_BASE_TRANSFORM_CLASS(gst_color_lut_parent_class)->sink_event(bt, event);
}

// BaseTransformImpl::stop (:196-199)
static gboolean gst_color_lut_stop(GstBaseTransform *trans)
{
    GstColorLut *self = reinterpret_cast<GstColorLut *>(trans);
    mvfx_pair_stop(self->hold, GST_OBJECT(trans), &gst_color_lut_pair_ops); // the held-back frame still needs the LUT
    mvfx_pair_print_stats(self->hold, GST_OBJECT(trans), "colorlut");
    mvfx_direct_reset(self);
    std::lock_guard<std::mutex> g(*self->lock);
    if (self->lut) mvfx_cube_lut_free(self->lut);
    self->lut = nullptr;
    return TRUE;
}

// VideoFilterImpl::transform_frame (:203-223)
static GstFlowReturn gst_color_lut_transform_frame(GstVideoFilter *filter, GstVideoFrame *in, GstVideoFrame *out)
{
    GstColorLut *self = reinterpret_cast<GstColorLut *>(filter);
    std::lock_guard<std::mutex> g(*self->lock); // state lock held for the frame, like the reference
    if (!self->lut) {
        GST_CAT_ERROR_OBJECT(colorlut_debug, self, "No LUT configured");
        return GST_FLOW_ERROR;
    }
    const mvfx_frame fi = mvfx_frame_from_gst(in), fo = mvfx_frame_from_gst(out);
    const int rc = mvfx_colorlut_transform_frame_host(self->lut, &fi, &fo);
    return MVFX_GST_FLOW(self, rc);
}

MVFX_DEFINE_HIP_ALLOCATION_VFUNCS(gst_color_lut, gst_color_lut_parent_class)

static GstFlowReturn gst_color_lut_prepare_output_buffer(GstBaseTransform *bt, GstBuffer *inbuf, GstBuffer **outbuf)
{
    if (!mvfx_buffer_is_hip(inbuf))
        return GST_BASE_TRANSFORM_CLASS(gst_color_lut_parent_class)->prepare_output_buffer(bt, inbuf, outbuf);
    return mvfx_hip_new_output(bt, inbuf, GST_VIDEO_INFO_SIZE(&GST_VIDEO_FILTER(bt)->out_info), outbuf);
}

// Device-resident path (cf. d3d12colorlut/imp.rs:544-719): LUT applied HBM -> HBM
static GstFlowReturn gst_color_lut_bt_transform(GstBaseTransform *bt, GstBuffer *inbuf, GstBuffer *outbuf)
{
    if (!mvfx_buffer_is_hip(inbuf) || !mvfx_buffer_is_hip(outbuf))
        return GST_BASE_TRANSFORM_CLASS(gst_color_lut_parent_class)->transform(bt, inbuf, outbuf);
    GstColorLut *self = reinterpret_cast<GstColorLut *>(bt);
    GstVideoFilter *vf = GST_VIDEO_FILTER(bt);
    if (!vf->negotiated)
        return GST_FLOW_NOT_NEGOTIATED;
    std::lock_guard<std::mutex> g(*self->lock);
    if (!self->lut) {
        GST_CAT_ERROR_OBJECT(colorlut_debug, self, "No LUT configured");
        return GST_FLOW_ERROR;
    }
    GstMapInfo imap, omap;
    if (GST_VIDEO_INFO_FORMAT(&vf->in_info) == GST_VIDEO_FORMAT_I420) {
        // memory:HIPMemory I420 (decoder output): the `videoconvert ! colorlut ! videoconvert` of the reference's example
        // pipeline (colorlut/imp.rs:17-19) in one kernel, no RGBA frame in HBM
        mvfx_planar_frame pi, po;
        if (!mvfx_hip_map_i420(inbuf, &vf->in_info, GST_MAP_READ, &imap, &pi))
            return GST_FLOW_ERROR;
        if (!mvfx_hip_map_i420(outbuf, &vf->out_info, GST_MAP_WRITE, &omap, &po)) {
            gst_buffer_unmap(inbuf, &imap);
            return GST_FLOW_ERROR;
        }
        mvfx_stream st = mvfx_element_stream(inbuf);
        mvfx_hip_buffer_acquire(inbuf, st);
        mvfx_hip_buffer_acquire(outbuf, st);
        MvfxFenceScope fs;
        mvfx_hip_fence_begin_buffers(&fs, inbuf, outbuf, st);
        int rc = mvfx_colorlut_transform_i420(self->lut, &pi, &po, 0, st);
        mvfx_hip_fence_end(&fs, st, NULL, GST_OBJECT(self)); // one fence for both buffers, carried by the kernel
        gst_buffer_unmap(outbuf, &omap);
        gst_buffer_unmap(inbuf, &imap);
        return MVFX_GST_FLOW(self, rc);
    }
    mvfx_frame fi, fo;
    if (!mvfx_hip_map_frame(inbuf, &vf->in_info, GST_MAP_READ, &imap, &fi))
        return GST_FLOW_ERROR;
    if (!mvfx_hip_map_frame(outbuf, &vf->out_info, GST_MAP_WRITE, &omap, &fo)) {
        gst_buffer_unmap(inbuf, &imap);
        return GST_FLOW_ERROR;
    }
    // fences instead of a host wait per buffer (d3d12colorlut/imp.rs:695-714)
    mvfx_stream st = mvfx_element_stream(inbuf);
    if (mvfx_pair_enabled() && gst_buffer_n_memory(inbuf) == 1 && gst_buffer_n_memory(outbuf) == 1) {
        gst_buffer_unmap(outbuf, &omap); // (a MVFX_MAP_HIP map is the device pointer: it stays valid while the memory lives)
        gst_buffer_unmap(inbuf, &imap);
        const int prc = mvfx_pair_submit(self->hold, GST_OBJECT(self), &gst_color_lut_pair_ops, inbuf, outbuf, fi, fo,
                                         st, TRUE, [] {});
        if (prc == MVFX_PAIR_FAILED_EARLIER) return mvfx_pair_flow_error(self->hold, GST_OBJECT(self));
        if (prc != MVFX_PAIR_NOT_TAKEN) return MVFX_GST_FLOW(self, prc);
        mvfx_hip_buffer_acquire(inbuf, st);
        mvfx_hip_buffer_acquire(outbuf, st);
        GstMemory *const both[2] = {gst_buffer_peek_memory(inbuf, 0), gst_buffer_peek_memory(outbuf, 0)};
        MvfxFenceScope dfs;
        mvfx_hip_fence_begin(&dfs, both, 2, st, TRUE);
        const int drc = mvfx_colorlut_transform_frame(self->lut, &fi, &fo, st);
        mvfx_hip_fence_end(&dfs, st, NULL, GST_OBJECT(self));
        return MVFX_GST_FLOW(self, drc);
    }
    MvfxFenceScope fs;
    // The direct-dispatch lane (csrc/direct_dispatch.h), as in hsvfilter / hsvdetector (gsthsv.cpp): the frame pair goes out as a packet of the library's
    // own on the lane queue its stream maps to, its completion signal is the fence of both buffers.  RGBA8 through a 3-D LUT only; anything else
    // comes back MVFX_ERR_DIRECT_UNAVAILABLE with nothing done and takes the stream.  In queue order the lane buys this kernel nothing; what does
    // (+8-13 %) is the packet WITHOUT the barrier bit, which lets the next frame's workgroups start in the tail of this one's, as the frames of a batched
    // launch do -- allowed when neither buffer's acquire rested on the queue's order (the usual case: the input's producer has finished or is a source).
    const int lane_queue = mvfx_direct_queue_of_stream(st);
    gboolean in_order = FALSE;
    if (!mvfx_direct_discouraged(self) && mvfx_hip_buffer_acquire_direct_ordered(inbuf, st, lane_queue, &in_order) &&
        mvfx_hip_buffer_acquire_direct_ordered(outbuf, st, lane_queue, &in_order)) {
        mvfx_hip_fence_begin_buffers(&fs, inbuf, outbuf, st);
        mvfx_thread_set_options(MVFX_OPT_DIRECT_DISPATCH | MVFX_OPT_DIRECT_ONLY | (in_order ? 0u : MVFX_OPT_DIRECT_UNORDERED));
        const int drc = mvfx_colorlut_transform_frame(self->lut, &fi, &fo, st);
        mvfx_thread_set_options(0);
        if (drc != MVFX_ERR_DIRECT_UNAVAILABLE) {
            mvfx_hip_fence_end(&fs, st, NULL, GST_OBJECT(self));
            gst_buffer_unmap(outbuf, &omap);
            gst_buffer_unmap(inbuf, &imap);
            return MVFX_GST_FLOW(self, drc);
        }
        mvfx_hip_fence_cancel(&fs);
    }
    mvfx_hip_buffer_acquire(inbuf, st);
    mvfx_hip_buffer_acquire(outbuf, st);
    mvfx_hip_fence_begin_buffers(&fs, inbuf, outbuf, st);
    int rc = mvfx_colorlut_transform_frame(self->lut, &fi, &fo, st);
    mvfx_hip_fence_end(&fs, st, NULL, GST_OBJECT(self)); // one fence for both buffers, carried by the kernel
    gst_buffer_unmap(outbuf, &omap);
    gst_buffer_unmap(inbuf, &imap);
    return MVFX_GST_FLOW(self, rc);
}

static gboolean gst_color_lut_set_info(GstVideoFilter *vf, GstCaps *, GstVideoInfo *in_info, GstCaps *, GstVideoInfo *)
{
    mvfx_pair_set_interval(reinterpret_cast<GstColorLut *>(vf)->hold, in_info); // a held-back frame waits one frame interval at most
    return TRUE;
}

static void gst_color_lut_finalize(GObject *obj)
{
    GstColorLut *self = reinterpret_cast<GstColorLut *>(obj);
    if (self->lut) mvfx_cube_lut_free(self->lut);
    delete self->location;
    delete self->hold;
    delete self->lock;
    G_OBJECT_CLASS(gst_color_lut_parent_class)->finalize(obj);
}

static void gst_color_lut_class_init(GstColorLutClass *klass)
{
    GObjectClass *gobject = G_OBJECT_CLASS(klass);
    GstElementClass *element = GST_ELEMENT_CLASS(klass);
    gobject->set_property = gst_color_lut_set_property;
    gobject->get_property = gst_color_lut_get_property;
    gobject->finalize = gst_color_lut_finalize;
    g_object_class_install_property(gobject, PROP_LOCATION, // colorlut/imp.rs:68-81
        g_param_spec_string("location", "Location", "Location of the LUT file to read from", NULL,
                            (GParamFlags)(G_PARAM_READWRITE | GST_PARAM_MUTABLE_READY | G_PARAM_STATIC_STRINGS)));
    gst_element_class_set_static_metadata(element, "Color LUT", "Filter/Effect/Video", "Apply color lookup table",
                                          "Seungha Yang <seungha@centricular.com>"); // :107-118
    // :120-158.  RGBA64_LE/BE exist only in newer GStreamer: advertise them when the
    // running library knows the names, RGBA always.
    const gboolean has64 = gst_video_format_from_string("RGBA64_LE") != GST_VIDEO_FORMAT_UNKNOWN;
    static const gchar *const all[] = {"RGBA64_LE", "RGBA64_BE", "RGBA", NULL};
    static const gchar *const only8[] = {"RGBA", NULL};
    // the reference's system-memory caps first, then their memory:HIPMemory twin, then HIP-only I420 (fused converters)
    static const gchar *const i420[] = {"I420", NULL};
    GstCaps *tmpl[2];
    for (GstCaps *&c : tmpl) {
        c = mvfx_caps_plus_hip(mvfx_video_caps(has64 ? all : only8));
        GstCaps *sys420 = mvfx_video_caps(i420);
        gst_caps_append(c, mvfx_caps_with_hip_feature(sys420));
        gst_caps_unref(sys420);
    }
    mvfx_add_pad_templates(element, tmpl[0], tmpl[1]);
    GST_BASE_TRANSFORM_CLASS(klass)->prepare_output_buffer = gst_color_lut_prepare_output_buffer;
    GST_BASE_TRANSFORM_CLASS(klass)->propose_allocation = gst_color_lut_propose_allocation; // d3d12colorlut/imp.rs:385-492
    GST_BASE_TRANSFORM_CLASS(klass)->decide_allocation = gst_color_lut_decide_allocation;
    GST_BASE_TRANSFORM_CLASS(klass)->transform = gst_color_lut_bt_transform;
    GST_BASE_TRANSFORM_CLASS(klass)->sink_event = gst_color_lut_sink_event;
    GST_BASE_TRANSFORM_CLASS(klass)->start = gst_color_lut_start;
    GST_BASE_TRANSFORM_CLASS(klass)->stop = gst_color_lut_stop;
    GST_VIDEO_FILTER_CLASS(klass)->transform_frame = gst_color_lut_transform_frame; // NeverInPlace (:162-166)
    GST_VIDEO_FILTER_CLASS(klass)->set_info = gst_color_lut_set_info;
}

static void gst_color_lut_init(GstColorLut *self)
{
    self->lock = new std::mutex();
    self->hold = new MvfxPairHold();
    self->location = new std::string();
    self->has_location = FALSE;
    self->lut = nullptr;
}

static gboolean plugin_init(GstPlugin *plugin) // colorlut/src/lib.rs:22-31
{
    GST_DEBUG_CATEGORY_INIT(colorlut_debug, "colorlut", 0, "Color LUT");
    return gst_element_register(plugin, "colorlut", GST_RANK_NONE, gst_color_lut_get_type());
}

// "MPL-2.0" is only a known licence string since GStreamer 1.20 (videofx/src/lib.rs:41 FIXME)
#if GST_CHECK_VERSION(1, 20, 0)
#define MVFX_COLORLUT_LICENSE "MPL-2.0"
#else
#define MVFX_COLORLUT_LICENSE "MPL"
#endif
GST_PLUGIN_DEFINE(GST_VERSION_MAJOR, GST_VERSION_MINOR, colorlut, "GStreamer Color LUT Plugin", plugin_init,
                  MVFX_GST_VERSION, MVFX_COLORLUT_LICENSE, "gst-plugin-colorlut", MVFX_GST_ORIGIN)
