// libgsthsv.so -- plugin `hsv` with elements `hsvfilter` and `hsvdetector`.
// Same surface as video/hsv/src/{lib.rs,hsvfilter,hsvdetector} of the reference; the per-pixel
// work is done by the HIP kernels behind include/mi355vfx.h.
#include "mvfx_gst_common.h"
#include "mvfx_pair_hold.h"

#include <mutex>

GST_DEBUG_CATEGORY_STATIC(hsvfilter_debug);   // hsvfilter/imp.rs:59-65
GST_DEBUG_CATEGORY_STATIC(hsvdetector_debug); // hsvdetector/imp.rs:64-70

// ------------------------------------------------------------------------- hsvfilter

struct GstHsvFilter {
    GstVideoFilter parent;
    std::mutex *lock;                 // settings: Mutex<Settings> (hsvfilter/imp.rs:55-57)
    mvfx_hsvfilter_settings settings;
    void *i420_scratch;               // device frame for the fused I420 path (streaming thread only)
    gsize i420_scratch_size;
    // pair launches on device buffers, opt-in (mvfx_pair_hold.h): the held-back frame's settings (the ones in force when ITS buffer
    // came) and those of the buffer being submitted; both written by the streaming thread only, pend_settings under the hold's lock
    MvfxPairHold *hold;
    mvfx_hsvfilter_settings pend_settings, cur_settings;
};
struct GstHsvFilterClass {
    GstVideoFilterClass parent_class;
};
G_DEFINE_TYPE(GstHsvFilter, gst_hsv_filter, GST_TYPE_VIDEO_FILTER)

enum { PROP_F_0, PROP_HUE_SHIFT, PROP_SAT_MUL, PROP_SAT_OFF, PROP_VAL_MUL, PROP_VAL_OFF };

static void gst_hsv_filter_set_property(GObject *obj, guint id, const GValue *value, GParamSpec *pspec)
{
    GstHsvFilter *self = reinterpret_cast<GstHsvFilter *>(obj);
    std::lock_guard<std::mutex> g(*self->lock);
    float *dst = nullptr;
    switch (id) {
    case PROP_HUE_SHIFT: dst = &self->settings.hue_shift; break;
    case PROP_SAT_MUL: dst = &self->settings.saturation_mul; break;
    case PROP_SAT_OFF: dst = &self->settings.saturation_off; break;
    case PROP_VAL_MUL: dst = &self->settings.value_mul; break;
    case PROP_VAL_OFF: dst = &self->settings.value_off; break;
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID(obj, id, pspec); return;
    }
    GST_CAT_INFO_OBJECT(hsvfilter_debug, obj, "Changing %s from %f to %f", pspec->name, *dst, g_value_get_float(value));
    *dst = g_value_get_float(value);
}

static void gst_hsv_filter_get_property(GObject *obj, guint id, GValue *value, GParamSpec *pspec)
{
    GstHsvFilter *self = reinterpret_cast<GstHsvFilter *>(obj);
    std::lock_guard<std::mutex> g(*self->lock);
    switch (id) {
    case PROP_HUE_SHIFT: g_value_set_float(value, self->settings.hue_shift); break;
    case PROP_SAT_MUL: g_value_set_float(value, self->settings.saturation_mul); break;
    case PROP_SAT_OFF: g_value_set_float(value, self->settings.saturation_off); break;
    case PROP_VAL_MUL: g_value_set_float(value, self->settings.value_mul); break;
    case PROP_VAL_OFF: g_value_set_float(value, self->settings.value_off); break;
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID(obj, id, pspec); break;
    }
}

// VideoFilterImpl::transform_frame_ip (hsvfilter/imp.rs:322-377)
static GstFlowReturn gst_hsv_filter_transform_frame_ip(GstVideoFilter *filter, GstVideoFrame *frame)
{
    GstHsvFilter *self = reinterpret_cast<GstHsvFilter *>(filter);
    mvfx_hsvfilter_settings s;
    {
        std::lock_guard<std::mutex> g(*self->lock); // snapshot once per frame (hsvfilter/imp.rs:85)
        s = self->settings;
    }
    const mvfx_frame f = mvfx_frame_from_gst(frame);
    const int rc = mvfx_hsvfilter_transform_frame_ip_host(&f, &s);
    return MVFX_GST_FLOW(self, rc);
}

// Device-resident path: a `video/x-raw(memory:HIPMemory)` buffer is filtered in HBM, no PCIe copy.
MVFX_DEFINE_HIP_ALLOCATION_VFUNCS(gst_hsv_filter, gst_hsv_filter_parent_class)

// in place: in == out (mvfx_pair_hold.h); two frames carry their own settings
static int gst_hsv_filter_pair_launch(GstObject *element, const mvfx_frame *, const mvfx_frame *out, uint32_t n, mvfx_stream st)
{
    GstHsvFilter *self = reinterpret_cast<GstHsvFilter *>(element);
    if (n == 1) return mvfx_hsvfilter_transform_frame_ip(out, &self->pend_settings, st);
    const mvfx_hsvfilter_settings settings[2] = {self->pend_settings, self->cur_settings};
    return mvfx_hsvfilter_transform_frames_ip_settings(out, n, settings, st);
}
MVFX_PAIR_DEFINE_OPS(gst_hsv_filter, GstHsvFilter, gst_hsv_filter_pair_launch)

static GstFlowReturn gst_hsv_filter_bt_transform_ip(GstBaseTransform *bt, GstBuffer *buf)
{
    if (!mvfx_buffer_is_hip(buf))
        return GST_BASE_TRANSFORM_CLASS(gst_hsv_filter_parent_class)->transform_ip(bt, buf); // GstVideoFilter: map + transform_frame_ip
    GstHsvFilter *self = reinterpret_cast<GstHsvFilter *>(bt);
    GstVideoFilter *vf = GST_VIDEO_FILTER(bt);
    if (!vf->negotiated)
        return GST_FLOW_NOT_NEGOTIATED;
    mvfx_hsvfilter_settings s;
    {
        std::lock_guard<std::mutex> g(*self->lock);
        s = self->settings;
    }
    GstMapInfo map;
    if (GST_VIDEO_INFO_FORMAT(&vf->in_info) == GST_VIDEO_FORMAT_I420) {
        // memory:HIPMemory I420: `videoconvert ! hsvfilter ! videoconvert` in one kernel (out of place: the co-sited chroma
        // filter reads neighbour input pixels), then the result replaces the buffer's planes (the element is AlwaysInPlace)
        mvfx_planar_frame pin, pout;
        if (!mvfx_hip_map_i420(buf, &vf->in_info, GST_MAP_READWRITE, &map, &pin))
            return GST_FLOW_ERROR;
        const gsize size = GST_VIDEO_INFO_SIZE(&vf->in_info);
        int rc = MVFX_OK;
        if (self->i420_scratch_size < size) { // kept across frames; sized by the first frame of each caps
            mvfx_device_free(self->i420_scratch);
            self->i420_scratch = NULL;
            self->i420_scratch_size = 0;
            rc = mvfx_device_alloc(&self->i420_scratch, size);
            if (rc == MVFX_OK) self->i420_scratch_size = size;
        }
        void *scratch = self->i420_scratch;
        if (rc == MVFX_OK) {
            pout = pin;
            for (guint p = 0; p < 3; p++)
                pout.data[p] = (guint8 *)scratch + GST_VIDEO_INFO_PLANE_OFFSET(&vf->in_info, p);
            mvfx_stream st = mvfx_thread_stream();
            mvfx_hip_buffer_acquire(buf, st);
            rc = mvfx_hsvfilter_transform_i420(&pin, &pout, &s, 0, st);
            if (rc == MVFX_OK)
                rc = mvfx_copy_device_to_device(map.data, scratch, size, st); // synchronises the stream
        }
        gst_buffer_unmap(buf, &map);
        return MVFX_GST_FLOW(self, rc);
    }
    mvfx_frame f;
    if (!mvfx_hip_map_frame(buf, &vf->in_info, GST_MAP_READWRITE, &map, &f))
        return GST_FLOW_ERROR;
    // no host wait per buffer: the stream first waits (on the device) for whoever touched the block last, and the block's
    // fence is recorded behind the kernel -- the next element, possibly on another streaming thread and stream, waits for
    // it the same way, a CPU map waits on the host (d3d12colorlut/imp.rs:695-714 does this with an ID3D12Fence)
    // default cache policy (thread option word 0): the next element reads this frame on the GPU.
    // MVFX_COMBINE (environment, read once): the launch combiner -- still one call per buffer, but the frames the hsvfilter elements
    // of this process hand in at about the same time leave as one batched launch with per-frame settings.  1: ordered through this
    // thread's stream; 2: the fenced entry -- the buffer's fence goes in as an event, the batch's event comes back as its new fence,
    // nothing is enqueued on a stream of this thread ("the launch combiner")
    static const int combine = g_getenv("MVFX_COMBINE") ? atoi(g_getenv("MVFX_COMBINE")) : 0;
    int rc;
    if (combine == 2 && gst_buffer_n_memory(buf) == 1) {
        GstMemory *mem = gst_buffer_peek_memory(buf, 0);
        mvfx_event done = NULL;
        rc = mvfx_hsvfilter_transform_frame_ip_fenced(&f, &s, mvfx_hip_memory_pending_fence(mem), &done);
        if (rc == MVFX_OK && done) mvfx_hip_memory_set_borrowed_fence(mem, done);
        gst_buffer_unmap(buf, &map);
        return MVFX_GST_FLOW(self, rc);
    }
    mvfx_stream st = mvfx_element_stream(buf);
    // Pair launches, opt-in (MVFX_ELEMENT_PAIR=1|2; mvfx_pair_hold.h).  One 4K frame per launch fills and drains the chip for 33 MB:
    // 0.67 of the HBM peak from one streaming thread however the streams are rotated; TWO frames per launch on two alternating streams
    // reach 0.71 (profiles/r4/element_path.txt).  The default is the reference's contract to the letter: one launch per call, the
    // call's flow return is that frame's (hsvfilter/imp.rs:322-326).
    if (mvfx_pair_enabled() && combine == 0 && gst_buffer_n_memory(buf) == 1) {
        gst_buffer_unmap(buf, &map); // (a MVFX_MAP_HIP map is the device pointer: it stays valid while the memory lives)
        self->cur_settings = s;
        const int prc = mvfx_pair_submit(self->hold, GST_OBJECT(self), &gst_hsv_filter_pair_ops, NULL, buf, f, f, st, TRUE,
                                         [&] { self->pend_settings = s; });
        if (prc == MVFX_PAIR_FAILED_EARLIER) return mvfx_pair_flow_error(self->hold, GST_OBJECT(self));
        if (prc != MVFX_PAIR_NOT_TAKEN) return MVFX_GST_FLOW(self, prc);
        GstMemory *mem = gst_buffer_peek_memory(buf, 0);
        mvfx_hip_buffer_acquire(buf, st);
        MvfxFenceScope dfs;
        mvfx_hip_fence_begin(&dfs, &mem, 1, st, TRUE);
        rc = mvfx_hsvfilter_transform_frame_ip(&f, &s, st);
        mvfx_hip_fence_end(&dfs, st, NULL, GST_OBJECT(self));
        return MVFX_GST_FLOW(self, rc);
    }
    // The direct-dispatch lane (round 6, MVFX_OPT_DIRECT_DISPATCH): when nothing is pending on the block -- the usual state of a recycled block --
    // the library may send the frame's kernel out on its own queue, without the release fence a stream's kernel packets carry: one call per buffer
    // faster than a batched launch on a stream (csrc/direct_dispatch.h).  Not when a consumer close behind had to wait for such a fence on its thread
    // (mvfx_direct_discouraged), not with the launch combiner.
    MvfxFenceScope fs; // the fence rides on the kernel; recorded behind it where no kernel of this thread took it (the combiner's launches)
    if (combine == 0 && !mvfx_direct_discouraged(self) && mvfx_hip_buffer_acquire_direct(buf, st, mvfx_direct_queue_of_stream(st))) {
        mvfx_hip_fence_begin_buffers(&fs, buf, NULL, st);
        mvfx_thread_set_options(MVFX_OPT_DIRECT_DISPATCH | MVFX_OPT_DIRECT_ONLY);
        rc = mvfx_hsvfilter_transform_frame_ip(&f, &s, st);
        mvfx_thread_set_options(0);
        if (rc != MVFX_ERR_DIRECT_UNAVAILABLE) {
            mvfx_hip_fence_end(&fs, st, NULL, GST_OBJECT(self));
            gst_buffer_unmap(buf, &map);
            return MVFX_GST_FLOW(self, rc);
        }
        mvfx_hip_fence_cancel(&fs); // not a frame for the lane (row padding, RGB / BGR, literal-kernel settings, no HSA queue): the stream it is
    }
    mvfx_hip_buffer_acquire(buf, st);
    mvfx_hip_fence_begin_buffers(&fs, buf, NULL, st);
    rc = combine == 1 ? mvfx_hsvfilter_transform_frame_ip_combined(&f, &s, st) : mvfx_hsvfilter_transform_frame_ip(&f, &s, st);
    mvfx_hip_fence_end(&fs, st, NULL, GST_OBJECT(self));
    gst_buffer_unmap(buf, &map);
    return MVFX_GST_FLOW(self, rc);
}

// EOS, flush-start: nothing stays held back across them
static gboolean gst_hsv_filter_sink_event(GstBaseTransform *bt, GstEvent *event)
{
    if (GST_EVENT_TYPE(event) == GST_EVENT_EOS || GST_EVENT_TYPE(event) == GST_EVENT_FLUSH_START)
        gst_hsv_filter_flush_cb(GST_OBJECT(bt));
    return GST_BASE_TRANSFORM_CLASS(gst_hsv_filter_parent_class)->sink_event(bt, event);
}

static gboolean gst_hsv_filter_stop(GstBaseTransform *bt)
{
    GstHsvFilter *self = reinterpret_cast<GstHsvFilter *>(bt);
    mvfx_pair_stop(self->hold, GST_OBJECT(bt), &gst_hsv_filter_pair_ops);
    mvfx_pair_print_stats(self->hold, GST_OBJECT(bt), "hsvfilter");
    mvfx_direct_reset(self); // the next run finds out again whether somebody waits close behind its frames
    return TRUE;
}

static gboolean gst_hsv_filter_set_info(GstVideoFilter *vf, GstCaps *, GstVideoInfo *in_info, GstCaps *, GstVideoInfo *)
{
    mvfx_pair_set_interval(reinterpret_cast<GstHsvFilter *>(vf)->hold, in_info); // a held-back frame waits one frame interval at most
    return TRUE;
}

static void gst_hsv_filter_finalize(GObject *obj)
{
    mvfx_device_free(reinterpret_cast<GstHsvFilter *>(obj)->i420_scratch);
    delete reinterpret_cast<GstHsvFilter *>(obj)->lock;
    delete reinterpret_cast<GstHsvFilter *>(obj)->hold;
    G_OBJECT_CLASS(gst_hsv_filter_parent_class)->finalize(obj);
}

static void gst_hsv_filter_class_init(GstHsvFilterClass *klass)
{
    GObjectClass *gobject = G_OBJECT_CLASS(klass);
    GstElementClass *element = GST_ELEMENT_CLASS(klass);
    GstVideoFilterClass *vfilter = GST_VIDEO_FILTER_CLASS(klass);
    gobject->set_property = gst_hsv_filter_set_property;
    gobject->get_property = gst_hsv_filter_get_property;
    gobject->finalize = gst_hsv_filter_finalize;
    const GParamFlags flags = (GParamFlags)(G_PARAM_READWRITE | GST_PARAM_MUTABLE_PLAYING | G_PARAM_STATIC_STRINGS);
    // hsvfilter/imp.rs:124-161 (defaults :25-29)
    g_object_class_install_property(gobject, PROP_HUE_SHIFT,
        g_param_spec_float("hue-shift", "Hue shift", "Hue shifting in degrees", -G_MAXFLOAT, G_MAXFLOAT, 0.0f, flags));
    g_object_class_install_property(gobject, PROP_SAT_MUL,
        g_param_spec_float("saturation-mul", "Saturation multiplier",
                           "Saturation multiplier to apply to the saturation value (before offset)", -G_MAXFLOAT, G_MAXFLOAT, 1.0f, flags));
    g_object_class_install_property(gobject, PROP_SAT_OFF,
        g_param_spec_float("saturation-off", "Saturation offset",
                           "Saturation offset to add to the saturation value (after multiplier)", -G_MAXFLOAT, G_MAXFLOAT, 0.0f, flags));
    g_object_class_install_property(gobject, PROP_VAL_MUL,
        g_param_spec_float("value-mul", "Value multiplier", "Value multiplier to apply to the value (before offset)",
                           -G_MAXFLOAT, G_MAXFLOAT, 1.0f, flags));
    g_object_class_install_property(gobject, PROP_VAL_OFF,
        g_param_spec_float("value-off", "Value offset", "Value offset to add to the value (after multiplier)",
                           -G_MAXFLOAT, G_MAXFLOAT, 0.0f, flags));
    gst_element_class_set_static_metadata(element, "HSV filter", "Filter/Effect/Converter/Video",
        "Works within the HSV colorspace to apply transformations to incoming frames",
        "Julien Bardagi <julien.bardagi@gmail.com>"); // hsvfilter/imp.rs:261-272
    static const gchar *const formats[] = {"RGBx", "xRGB", "BGRx", "xBGR", "RGBA", "ARGB", "BGRA", "ABGR", "RGB", "BGR", NULL};
    // :274-312, then the memory:HIPMemory twin, then HIP-only I420 (fused converters)
    static const gchar *const i420[] = {"I420", NULL};
    GstCaps *tmpl[2];
    for (GstCaps *&c : tmpl) {
        c = mvfx_caps_plus_hip(mvfx_video_caps(formats));
        GstCaps *sys420 = mvfx_video_caps(i420);
        gst_caps_append(c, mvfx_caps_with_hip_feature(sys420));
        gst_caps_unref(sys420);
    }
    mvfx_add_pad_templates(element, tmpl[0], tmpl[1]);
    vfilter->transform_frame_ip = gst_hsv_filter_transform_frame_ip; // AlwaysInPlace (:315-320)
    vfilter->set_info = gst_hsv_filter_set_info;
    GST_BASE_TRANSFORM_CLASS(klass)->transform_ip = gst_hsv_filter_bt_transform_ip;
    GST_BASE_TRANSFORM_CLASS(klass)->sink_event = gst_hsv_filter_sink_event;
    GST_BASE_TRANSFORM_CLASS(klass)->stop = gst_hsv_filter_stop;
    GST_BASE_TRANSFORM_CLASS(klass)->propose_allocation = gst_hsv_filter_propose_allocation; // d3d12colorlut/imp.rs:385-492
    GST_BASE_TRANSFORM_CLASS(klass)->decide_allocation = gst_hsv_filter_decide_allocation;
}

static void gst_hsv_filter_init(GstHsvFilter *self)
{
    self->lock = new std::mutex();
    self->hold = new MvfxPairHold();
    self->settings = self->pend_settings = self->cur_settings = mvfx_hsvfilter_settings{0.0f, 1.0f, 0.0f, 1.0f, 0.0f};
    self->i420_scratch = NULL;
    self->i420_scratch_size = 0;
}

// ------------------------------------------------------------------------- hsvdetector

struct GstHsvDetector {
    GstVideoFilter parent;
    std::mutex *lock;
    mvfx_hsvdetector_settings settings;
    // pair launches on device buffers (mvfx_pair_hold.h): the held-back frame and the settings in force when its buffer came
    MvfxPairHold *hold;
    mvfx_hsvdetector_settings pend_settings;
};
struct GstHsvDetectorClass {
    GstVideoFilterClass parent_class;
};
G_DEFINE_TYPE(GstHsvDetector, gst_hsv_detector, GST_TYPE_VIDEO_FILTER)

enum { PROP_D_0, PROP_HUE_REF, PROP_HUE_VAR, PROP_SAT_REF, PROP_SAT_VAR, PROP_VAL_REF, PROP_VAL_VAR };

static float *hsv_detector_field(GstHsvDetector *self, guint id)
{
    switch (id) {
    case PROP_HUE_REF: return &self->settings.hue_ref;
    case PROP_HUE_VAR: return &self->settings.hue_var;
    case PROP_SAT_REF: return &self->settings.saturation_ref;
    case PROP_SAT_VAR: return &self->settings.saturation_var;
    case PROP_VAL_REF: return &self->settings.value_ref;
    case PROP_VAL_VAR: return &self->settings.value_var;
    default: return nullptr;
    }
}

static void gst_hsv_detector_set_property(GObject *obj, guint id, const GValue *value, GParamSpec *pspec)
{
    GstHsvDetector *self = reinterpret_cast<GstHsvDetector *>(obj);
    std::lock_guard<std::mutex> g(*self->lock);
    float *dst = hsv_detector_field(self, id);
    if (!dst) { G_OBJECT_WARN_INVALID_PROPERTY_ID(obj, id, pspec); return; }
    GST_CAT_INFO_OBJECT(hsvdetector_debug, obj, "Changing %s from %f to %f", pspec->name, *dst, g_value_get_float(value));
    *dst = g_value_get_float(value);
}

static void gst_hsv_detector_get_property(GObject *obj, guint id, GValue *value, GParamSpec *pspec)
{
    GstHsvDetector *self = reinterpret_cast<GstHsvDetector *>(obj);
    std::lock_guard<std::mutex> g(*self->lock);
    float *src = hsv_detector_field(self, id);
    if (!src) { G_OBJECT_WARN_INVALID_PROPERTY_ID(obj, id, pspec); return; }
    g_value_set_float(value, *src);
}

static const gchar *const kDetectorIn[] = {"RGBx", "xRGB", "BGRx", "xBGR", "RGB", "BGR", NULL};  // hsvdetector/imp.rs:78-87
static const gchar *const kDetectorOut[] = {"RGBA", "ARGB", "BGRA", "ABGR", NULL};              // :89-96
// memory:HIPMemory sink caps additionally take I420 (decoder output): `videoconvert ! hsvdetector` as one kernel
static const gchar *const kDetectorInHip[] = {"RGBx", "xRGB", "BGRx", "xBGR", "RGB", "BGR", "I420", NULL};

// BaseTransformImpl::transform_caps (hsvdetector/imp.rs:386-419): replace the `format` field of
// every structure with the list of the other side, then intersect with the filter (First mode)
static GstCaps *gst_hsv_detector_transform_caps(GstBaseTransform *trans, GstPadDirection direction, GstCaps *caps,
                                                GstCaps *filter)
{
    GstCaps *other = gst_caps_copy(caps);
    GstCaps *tmpl = mvfx_video_caps(direction == GST_PAD_SRC ? kDetectorIn : kDetectorOut);
    GstCaps *tmpl_hip = mvfx_video_caps(direction == GST_PAD_SRC ? kDetectorInHip : kDetectorOut);
    const GValue *formats = gst_structure_get_value(gst_caps_get_structure(tmpl, 0), "format");
    const GValue *formats_hip = gst_structure_get_value(gst_caps_get_structure(tmpl_hip, 0), "format");
    for (guint i = 0; i < gst_caps_get_size(other); i++) {
        GstCapsFeatures *f = gst_caps_get_features(other, i);
        const gboolean hip = f && gst_caps_features_contains(f, MVFX_CAPS_FEATURE_MEMORY_HIP);
        gst_structure_set_value(gst_caps_get_structure(other, i), "format", hip ? formats_hip : formats);
    }
    gst_caps_unref(tmpl);
    gst_caps_unref(tmpl_hip);
    GST_CAT_DEBUG_OBJECT(hsvdetector_debug, trans, "Transformed caps from %" GST_PTR_FORMAT " to %" GST_PTR_FORMAT " in direction %d",
                         caps, other, (int)direction);
    if (filter) {
        GstCaps *r = gst_caps_intersect_full(filter, other, GST_CAPS_INTERSECT_FIRST);
        gst_caps_unref(other);
        return r;
    }
    return other;
}

// VideoFilterImpl::transform_frame (hsvdetector/imp.rs:422-707)
static GstFlowReturn gst_hsv_detector_transform_frame(GstVideoFilter *filter, GstVideoFrame *in, GstVideoFrame *out)
{
    GstHsvDetector *self = reinterpret_cast<GstHsvDetector *>(filter);
    mvfx_hsvdetector_settings s;
    {
        std::lock_guard<std::mutex> g(*self->lock);
        s = self->settings;
    }
    const mvfx_frame fi = mvfx_frame_from_gst(in), fo = mvfx_frame_from_gst(out);
    const int rc = mvfx_hsvdetector_transform_frame_host(&fi, &fo, &s);
    return MVFX_GST_FLOW(self, rc);
}

MVFX_DEFINE_HIP_ALLOCATION_VFUNCS(gst_hsv_detector, gst_hsv_detector_parent_class)

static GstFlowReturn gst_hsv_detector_prepare_output_buffer(GstBaseTransform *bt, GstBuffer *inbuf, GstBuffer **outbuf)
{
    if (!mvfx_buffer_is_hip(inbuf))
        return GST_BASE_TRANSFORM_CLASS(gst_hsv_detector_parent_class)->prepare_output_buffer(bt, inbuf, outbuf);
    return mvfx_hip_new_output(bt, inbuf, GST_VIDEO_INFO_SIZE(&GST_VIDEO_FILTER(bt)->out_info), outbuf);
}

static int gst_hsv_detector_pair_launch(GstObject *element, const mvfx_frame *in, const mvfx_frame *out, uint32_t n, mvfx_stream st)
{
    GstHsvDetector *self = reinterpret_cast<GstHsvDetector *>(element);
    return n == 1 ? mvfx_hsvdetector_transform_frame(in, out, &self->pend_settings, st)
                  : mvfx_hsvdetector_transform_frames(in, out, n, &self->pend_settings, st);
}

MVFX_PAIR_DEFINE_OPS(gst_hsv_detector, GstHsvDetector, gst_hsv_detector_pair_launch)

static GstFlowReturn gst_hsv_detector_bt_transform(GstBaseTransform *bt, GstBuffer *inbuf, GstBuffer *outbuf)
{
    if (!mvfx_buffer_is_hip(inbuf) || !mvfx_buffer_is_hip(outbuf))
        return GST_BASE_TRANSFORM_CLASS(gst_hsv_detector_parent_class)->transform(bt, inbuf, outbuf);
    GstHsvDetector *self = reinterpret_cast<GstHsvDetector *>(bt);
    GstVideoFilter *vf = GST_VIDEO_FILTER(bt);
    if (!vf->negotiated)
        return GST_FLOW_NOT_NEGOTIATED;
    mvfx_hsvdetector_settings s;
    {
        std::lock_guard<std::mutex> g(*self->lock);
        s = self->settings;
    }
    GstMapInfo imap, omap;
    mvfx_frame fi, fo;
    mvfx_planar_frame pi;
    const gboolean i420 = GST_VIDEO_INFO_FORMAT(&vf->in_info) == GST_VIDEO_FORMAT_I420;
    if (i420 ? !mvfx_hip_map_i420(inbuf, &vf->in_info, GST_MAP_READ, &imap, &pi) : !mvfx_hip_map_frame(inbuf, &vf->in_info, GST_MAP_READ, &imap, &fi))
        return GST_FLOW_ERROR;
    if (!mvfx_hip_map_frame(outbuf, &vf->out_info, GST_MAP_WRITE, &omap, &fo)) {
        gst_buffer_unmap(inbuf, &imap);
        return GST_FLOW_ERROR;
    }
    mvfx_stream st = mvfx_element_stream(inbuf);
    if (!i420 && mvfx_pair_enabled() && gst_buffer_n_memory(inbuf) == 1 && gst_buffer_n_memory(outbuf) == 1) {
        gst_buffer_unmap(outbuf, &omap); // (a MVFX_MAP_HIP map is the device pointer: it stays valid while the memory lives)
        gst_buffer_unmap(inbuf, &imap);
        // pend_settings is written by this thread only (under the hold's lock, for the flush on another thread to read)
        const gboolean same = memcmp(&self->pend_settings, &s, sizeof s) == 0;
        const int prc = mvfx_pair_submit(self->hold, GST_OBJECT(self), &gst_hsv_detector_pair_ops, inbuf, outbuf,
                                         fi, fo, st, same, [&] { self->pend_settings = s; });
        if (prc == MVFX_PAIR_FAILED_EARLIER) return mvfx_pair_flow_error(self->hold, GST_OBJECT(self));
        if (prc != MVFX_PAIR_NOT_TAKEN) return MVFX_GST_FLOW(self, prc);
        mvfx_hip_buffer_acquire(inbuf, st);
        mvfx_hip_buffer_acquire(outbuf, st);
        GstMemory *const both[2] = {gst_buffer_peek_memory(inbuf, 0), gst_buffer_peek_memory(outbuf, 0)};
        MvfxFenceScope dfs;
        mvfx_hip_fence_begin(&dfs, both, 2, st, TRUE);
        const int drc = mvfx_hsvdetector_transform_frame(&fi, &fo, &s, st);
        mvfx_hip_fence_end(&dfs, st, NULL, GST_OBJECT(self));
        return MVFX_GST_FLOW(self, drc);
    }
    MvfxFenceScope fs; // one fence for both buffers (the reader's too: the input block may be recycled and overwritten next), on the kernel
    // The direct-dispatch lane, as in hsvfilter: the frame pair goes out on the lane queue its stream maps to -- the queue the filter in front of
    // this element used for the same frame (both pick the stream from the buffer's frame number), so the filter's kernel sits in front of ours in
    // an in-order queue and nobody waits for anybody on a host thread.
    const int lane_queue = mvfx_direct_queue_of_stream(st);
    if (!i420 && !mvfx_direct_discouraged(self) && mvfx_hip_buffer_acquire_direct(inbuf, st, lane_queue) && mvfx_hip_buffer_acquire_direct(outbuf, st, lane_queue)) {
        mvfx_hip_fence_begin_buffers(&fs, inbuf, outbuf, st);
        mvfx_thread_set_options(MVFX_OPT_DIRECT_DISPATCH | MVFX_OPT_DIRECT_ONLY);
        const int drc = mvfx_hsvdetector_transform_frame(&fi, &fo, &s, st);
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
    int rc = i420 ? mvfx_hsvdetector_transform_i420(&pi, &fo, &s, 0, st) : mvfx_hsvdetector_transform_frame(&fi, &fo, &s, st);
    mvfx_hip_fence_end(&fs, st, NULL, GST_OBJECT(self));
    gst_buffer_unmap(outbuf, &omap);
    gst_buffer_unmap(inbuf, &imap);
    return MVFX_GST_FLOW(self, rc);
}

// EOS, flush-start, stop: nothing stays held back across them
static gboolean gst_hsv_detector_sink_event(GstBaseTransform *bt, GstEvent *event)
{
    if (GST_EVENT_TYPE(event) == GST_EVENT_EOS || GST_EVENT_TYPE(event) == GST_EVENT_FLUSH_START)
        gst_hsv_detector_flush_cb(GST_OBJECT(bt));
    return GST_BASE_TRANSFORM_CLASS(gst_hsv_detector_parent_class)->sink_event(bt, event);
}

static gboolean gst_hsv_detector_stop(GstBaseTransform *bt)
{
    mvfx_pair_stop(reinterpret_cast<GstHsvDetector *>(bt)->hold, GST_OBJECT(bt), &gst_hsv_detector_pair_ops);
    mvfx_pair_print_stats(reinterpret_cast<GstHsvDetector *>(bt)->hold, GST_OBJECT(bt), "hsvdetector");
    mvfx_direct_reset(bt);
    return TRUE;
}

static gboolean gst_hsv_detector_set_info(GstVideoFilter *vf, GstCaps *, GstVideoInfo *in_info, GstCaps *, GstVideoInfo *)
{
    mvfx_pair_set_interval(reinterpret_cast<GstHsvDetector *>(vf)->hold, in_info); // a held-back frame waits one frame interval at most
    return TRUE;
}

static void gst_hsv_detector_finalize(GObject *obj)
{
    delete reinterpret_cast<GstHsvDetector *>(obj)->hold;
    delete reinterpret_cast<GstHsvDetector *>(obj)->lock;
    G_OBJECT_CLASS(gst_hsv_detector_parent_class)->finalize(obj);
}

static void gst_hsv_detector_class_init(GstHsvDetectorClass *klass)
{
    GObjectClass *gobject = G_OBJECT_CLASS(klass);
    GstElementClass *element = GST_ELEMENT_CLASS(klass);
    gobject->set_property = gst_hsv_detector_set_property;
    gobject->get_property = gst_hsv_detector_get_property;
    gobject->finalize = gst_hsv_detector_finalize;
    const GParamFlags flags = (GParamFlags)(G_PARAM_READWRITE | GST_PARAM_MUTABLE_PLAYING | G_PARAM_STATIC_STRINGS);
    // hsvdetector/imp.rs:164-212 (defaults :26-31)
    g_object_class_install_property(gobject, PROP_HUE_REF,
        g_param_spec_float("hue-ref", "Hue reference", "Hue reference in degrees", -G_MAXFLOAT, G_MAXFLOAT, 0.0f, flags));
    g_object_class_install_property(gobject, PROP_HUE_VAR,
        g_param_spec_float("hue-var", "Hue variation", "Allowed hue variation from the reference hue angle, in degrees",
                           0.0f, 180.0f, 10.0f, flags));
    g_object_class_install_property(gobject, PROP_SAT_REF,
        g_param_spec_float("saturation-ref", "Saturation reference", "Reference saturation value", 0.0f, 1.0f, 0.0f, flags));
    g_object_class_install_property(gobject, PROP_SAT_VAR,
        g_param_spec_float("saturation-var", "Saturation variation", "Allowed saturation variation from the reference value",
                           0.0f, 1.0f, 0.15f, flags));
    g_object_class_install_property(gobject, PROP_VAL_REF,
        g_param_spec_float("value-ref", "Value reference", "Reference value value", 0.0f, 1.0f, 0.0f, flags));
    g_object_class_install_property(gobject, PROP_VAL_VAR,
        g_param_spec_float("value-var", "Value variation", "Allowed value variation from the reference value",
                           0.0f, 1.0f, 0.3f, flags));
    gst_element_class_set_static_metadata(element, "HSV detector", "Filter/Effect/Converter/Video",
        "Works within the HSV colorspace to mark positive pixels", "Julien Bardagi <julien.bardagi@gmail.com>");
    GstCaps *sink_tmpl = mvfx_video_caps(kDetectorIn); // the reference's caps, then the HIP twin incl. I420
    {
        GstCaps *hip_in = mvfx_video_caps(kDetectorInHip);
        gst_caps_append(sink_tmpl, mvfx_caps_with_hip_feature(hip_in));
        gst_caps_unref(hip_in);
    }
    mvfx_add_pad_templates(element, sink_tmpl, mvfx_caps_plus_hip(mvfx_video_caps(kDetectorOut)));
    GST_BASE_TRANSFORM_CLASS(klass)->transform_caps = gst_hsv_detector_transform_caps;
    GST_BASE_TRANSFORM_CLASS(klass)->prepare_output_buffer = gst_hsv_detector_prepare_output_buffer;
    GST_BASE_TRANSFORM_CLASS(klass)->propose_allocation = gst_hsv_detector_propose_allocation;
    GST_BASE_TRANSFORM_CLASS(klass)->decide_allocation = gst_hsv_detector_decide_allocation;
    GST_BASE_TRANSFORM_CLASS(klass)->transform = gst_hsv_detector_bt_transform;
    GST_BASE_TRANSFORM_CLASS(klass)->sink_event = gst_hsv_detector_sink_event;
    GST_BASE_TRANSFORM_CLASS(klass)->stop = gst_hsv_detector_stop;
    GST_VIDEO_FILTER_CLASS(klass)->transform_frame = gst_hsv_detector_transform_frame; // NeverInPlace (:380-384)
    GST_VIDEO_FILTER_CLASS(klass)->set_info = gst_hsv_detector_set_info;
}

static void gst_hsv_detector_init(GstHsvDetector *self)
{
    self->lock = new std::mutex();
    self->hold = new MvfxPairHold();
    self->settings = mvfx_hsvdetector_settings{0.0f, 10.0f, 0.0f, 0.15f, 0.0f, 0.3f};
    self->pend_settings = self->settings;
}

// ------------------------------------------------------------------------- plugin (hsv/src/lib.rs:23-42)

static gboolean plugin_init(GstPlugin *plugin)
{
    GST_DEBUG_CATEGORY_INIT(hsvfilter_debug, "hsvfilter", 0, "Rust HSV transformation filter");
    GST_DEBUG_CATEGORY_INIT(hsvdetector_debug, "hsvdetector", 0, "Rust HSV-based detection filter");
    return gst_element_register(plugin, "hsvfilter", GST_RANK_NONE, gst_hsv_filter_get_type()) &&
           gst_element_register(plugin, "hsvdetector", GST_RANK_NONE, gst_hsv_detector_get_type());
}

GST_PLUGIN_DEFINE(GST_VERSION_MAJOR, GST_VERSION_MINOR, hsv, "GStreamer plugin with HSV manipulation elements", plugin_init,
                  MVFX_GST_VERSION, "MIT/X11", "gst-plugin-hsv", MVFX_GST_ORIGIN)
