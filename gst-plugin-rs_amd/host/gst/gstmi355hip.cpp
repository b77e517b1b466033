// libgstmi355hip.so -- plugin `mi355hip`: `hipupload` / `hipdownload`, the bridges between system
// memory and `video/x-raw(memory:HIPMemory)`.  With them a chain such as
//   videotestsrc ! hipupload ! hsvfilter ! hsvdetector ! colorlut ! hipdownload ! ...
// crosses PCIe once in and once out instead of twice per element.
#include "mvfx_gst_common.h"
#include "mvfxhipmemory.h"

#include <gst/base/gstpushsrc.h>

GST_DEBUG_CATEGORY_STATIC(mi355hip_debug);

struct GstMi355HipCopy {
    GstBaseTransform parent;
    GstVideoInfo info;
    gboolean have_info;
    GstBufferPool *pinned; // hipdownload: page-locked system-memory output buffers (D2H is then a plain DMA)
    gint device_id;        // hipupload, property: the device its blocks live on (-1: the streaming thread's current device)
};
struct GstMi355HipCopyClass {
    GstBaseTransformClass parent_class;
    gboolean upload;
};

static gpointer hipcopy_parent_class = NULL;

static GstCaps *hipcopy_transform_caps(GstBaseTransform *trans, GstPadDirection direction, GstCaps *caps, GstCaps *filter)
{
    const gboolean upload = ((GstMi355HipCopyClass *)G_OBJECT_GET_CLASS(trans))->upload;
    // upload: sink = system, src = HIP; download: the reverse
    const gboolean to_hip = upload ? direction == GST_PAD_SINK : direction == GST_PAD_SRC;
    GstCaps *other = mvfx_caps_set_hip_feature(caps, to_hip);
    if (filter) {
        GstCaps *r = gst_caps_intersect_full(filter, other, GST_CAPS_INTERSECT_FIRST);
        gst_caps_unref(other);
        return r;
    }
    return other;
}

static gboolean hipcopy_set_caps(GstBaseTransform *trans, GstCaps *incaps, GstCaps *)
{
    GstMi355HipCopy *self = (GstMi355HipCopy *)trans;
    self->have_info = gst_video_info_from_caps(&self->info, incaps);
    if (self->pinned) {
        gst_buffer_pool_set_active(self->pinned, FALSE);
        gst_object_unref(self->pinned);
        self->pinned = NULL;
    }
    return self->have_info;
}

static GstFlowReturn hipcopy_prepare_output_buffer(GstBaseTransform *trans, GstBuffer *inbuf, GstBuffer **outbuf)
{
    GstMi355HipCopy *self = (GstMi355HipCopy *)trans;
    const gboolean upload = ((GstMi355HipCopyClass *)G_OBJECT_GET_CLASS(trans))->upload;
    if (!self->have_info)
        return GST_FLOW_NOT_NEGOTIATED;
    *outbuf = NULL;
    if (upload && !mvfx_hip_select_device(self->device_id, GST_ELEMENT(trans))) // this streaming thread allocates (and copies) on `device-id`
        return GST_FLOW_ERROR;
    if (upload) { // device buffers come from the pool negotiated with downstream (hipcopy_decide_allocation)
        GstBufferPool *pool = gst_base_transform_get_buffer_pool(trans);
        if (pool) {
            if (mvfx_is_hip_buffer_pool(pool)) {
                if (!gst_buffer_pool_is_active(pool))
                    gst_buffer_pool_set_active(pool, TRUE);
                if (gst_buffer_pool_acquire_buffer(pool, outbuf, NULL) != GST_FLOW_OK)
                    *outbuf = NULL;
            }
            gst_object_unref(pool);
        }
    }
    if (!upload) { // page-locked system memory of our own pool; MVFX_HIP_PAGEABLE=1 keeps malloc'ed buffers (A/B measurements)
        if (!self->pinned && !g_getenv("MVFX_HIP_PAGEABLE")) {
            GstCaps *caps = gst_video_info_to_caps(&self->info);
            self->pinned = mvfx_pinned_buffer_pool_new_configured(caps, (guint)GST_VIDEO_INFO_SIZE(&self->info), 0);
            gst_caps_unref(caps);
            if (self->pinned && !gst_buffer_pool_set_active(self->pinned, TRUE)) {
                gst_object_unref(self->pinned);
                self->pinned = NULL;
            }
        }
        if (self->pinned && gst_buffer_pool_acquire_buffer(self->pinned, outbuf, NULL) != GST_FLOW_OK)
            *outbuf = NULL;
    }
    if (!*outbuf) {
        GstAllocator *alloc = upload ? mvfx_hip_allocator_get() : NULL;
        *outbuf = gst_buffer_new_allocate(alloc, GST_VIDEO_INFO_SIZE(&self->info), NULL);
        if (alloc) gst_object_unref(alloc);
    }
    if (!*outbuf) {
        GST_ELEMENT_ERROR(trans, RESOURCE, NO_SPACE_LEFT, ("%s", mvfx_last_error()), (NULL));
        return GST_FLOW_ERROR;
    }
    gst_buffer_copy_into(*outbuf, inbuf, (GstBufferCopyFlags)(GST_BUFFER_COPY_FLAGS | GST_BUFFER_COPY_TIMESTAMPS), 0, -1);
    return GST_FLOW_OK;
}

// hipupload: SRC caps are HIP -> HIP pool for the output; hipdownload: SINK caps are HIP -> offer one upstream
static gboolean hipcopy_decide_allocation(GstBaseTransform *trans, GstQuery *query)
{
    mvfx_hip_decide_allocation(query); // no-op unless the src caps carry memory:HIPMemory
    return GST_BASE_TRANSFORM_CLASS(hipcopy_parent_class)->decide_allocation(trans, query);
}

static gboolean hipcopy_propose_allocation(GstBaseTransform *trans, GstQuery *decide_query, GstQuery *query)
{
    if (!GST_BASE_TRANSFORM_CLASS(hipcopy_parent_class)->propose_allocation(trans, decide_query, query))
        return FALSE;
    if (mvfx_hip_propose_allocation(query)) // hipdownload: the sink caps carry memory:HIPMemory
        return TRUE;
    // hipupload: offer upstream a pool of page-locked system memory -- a source that fills our buffers makes the H2D copy a
    // plain DMA at PCIe speed instead of the runtime's chunked staging of pageable memory
    GstCaps *caps = NULL;
    gboolean need_pool = FALSE;
    gst_query_parse_allocation(query, &caps, &need_pool);
    GstVideoInfo info;
    if (caps && gst_video_info_from_caps(&info, caps) && !g_getenv("MVFX_HIP_PAGEABLE")) {
        const guint size = (guint)GST_VIDEO_INFO_SIZE(&info);
        GstBufferPool *pool = need_pool ? mvfx_pinned_buffer_pool_new_configured(caps, size, 2) : NULL;
        gst_query_add_allocation_pool(query, pool, size, 2, 0);
        if (pool) gst_object_unref(pool);
        gst_query_add_allocation_meta(query, GST_VIDEO_META_API_TYPE, NULL);
    }
    return TRUE;
}

static GstFlowReturn hipcopy_transform(GstBaseTransform *trans, GstBuffer *inbuf, GstBuffer *outbuf)
{
    GstMi355HipCopy *self = (GstMi355HipCopy *)trans;
    const gboolean upload = ((GstMi355HipCopyClass *)G_OBJECT_GET_CLASS(trans))->upload;
    GstBuffer *sys = upload ? inbuf : outbuf, *dev = upload ? outbuf : inbuf;
    GstVideoFrame frame; // the system-memory side, plane strides honoured (GstVideoMeta aware)
    if (!gst_video_frame_map(&frame, &self->info, sys, upload ? GST_MAP_READ : GST_MAP_WRITE))
        return GST_FLOW_ERROR;
    GstMapInfo dmap;
    if (!mvfx_buffer_is_hip(dev) || !gst_buffer_map(dev, &dmap, (GstMapFlags)(MVFX_MAP_HIP | (upload ? GST_MAP_WRITE : GST_MAP_READ)))) {
        gst_video_frame_unmap(&frame);
        GST_ELEMENT_ERROR(trans, CORE, NEGOTIATION, ("buffer is not HIP memory"), (NULL));
        return GST_FLOW_NOT_NEGOTIATED;
    }
    int rc = MVFX_OK;
    // the copies run on this thread's stream behind the fence of the device block (its producer, or the last reader of a
    // recycled block); the system-memory side is only borrowed, so the call returns when the copy has landed
    if (!mvfx_hip_follow_device(dev, GST_OBJECT(trans))) { // hipdownload: the device of the incoming memory; hipupload: of the block just allocated
        gst_buffer_unmap(dev, &dmap);
        gst_video_frame_unmap(&frame);
        return GST_FLOW_ERROR;
    }
    mvfx_stream st = mvfx_thread_stream();
    mvfx_hip_buffer_acquire(dev, st);
    // device buffers always use the default GstVideoInfo layout (offsets / strides of `info`)
    for (guint p = 0; p < GST_VIDEO_INFO_N_PLANES(&self->info) && rc == MVFX_OK; p++) {
        guint8 *d = dmap.data + GST_VIDEO_INFO_PLANE_OFFSET(&self->info, p);
        guint8 *s = (guint8 *)GST_VIDEO_FRAME_PLANE_DATA(&frame, p);
        const gint dstride = GST_VIDEO_INFO_PLANE_STRIDE(&self->info, p), sstride = GST_VIDEO_FRAME_PLANE_STRIDE(&frame, p);
        const guint rows = GST_VIDEO_FRAME_COMP_HEIGHT(&frame, p == 3 ? 3 : p);
        if (dstride == sstride) {
            rc = upload ? mvfx_copy_to_device_async(d, s, (size_t)dstride * rows, st) : mvfx_copy_to_host_async(s, d, (size_t)dstride * rows, st);
        } else {
            const size_t row = (size_t)MIN(dstride, sstride);
            for (guint y = 0; y < rows && rc == MVFX_OK; y++)
                rc = upload ? mvfx_copy_to_device_async(d + (size_t)y * dstride, s + (size_t)y * sstride, row, st)
                            : mvfx_copy_to_host_async(s + (size_t)y * sstride, d + (size_t)y * dstride, row, st);
        }
    }
    if (rc == MVFX_OK) rc = mvfx_stream_synchronize(st);
    gst_buffer_unmap(dev, &dmap);
    gst_video_frame_unmap(&frame);
    return MVFX_GST_FLOW(trans, rc);
}

enum { PROP_COPY_0, PROP_COPY_DEVICE_ID };

static void hipcopy_set_property(GObject *obj, guint id, const GValue *value, GParamSpec *pspec)
{
    if (id == PROP_COPY_DEVICE_ID) ((GstMi355HipCopy *)obj)->device_id = g_value_get_int(value);
    else G_OBJECT_WARN_INVALID_PROPERTY_ID(obj, id, pspec);
}

static void hipcopy_get_property(GObject *obj, guint id, GValue *value, GParamSpec *pspec)
{
    if (id == PROP_COPY_DEVICE_ID) g_value_set_int(value, ((GstMi355HipCopy *)obj)->device_id);
    else G_OBJECT_WARN_INVALID_PROPERTY_ID(obj, id, pspec);
}

// start(): a `device-id` that does not exist fails HERE, as a RESOURCE error of the element, not in the first buffer
static gboolean hipcopy_start(GstBaseTransform *trans)
{
    GstMi355HipCopy *self = (GstMi355HipCopy *)trans;
    const gboolean upload = ((GstMi355HipCopyClass *)G_OBJECT_GET_CLASS(trans))->upload;
    return !upload || mvfx_hip_select_device(self->device_id, GST_ELEMENT(trans));
}

static gboolean hipcopy_stop(GstBaseTransform *trans)
{
    GstMi355HipCopy *self = (GstMi355HipCopy *)trans;
    if (self->pinned) {
        gst_buffer_pool_set_active(self->pinned, FALSE);
        gst_object_unref(self->pinned);
        self->pinned = NULL;
    }
    return TRUE;
}

static void hipcopy_class_init_common(GstMi355HipCopyClass *klass, gboolean upload)
{
    GstElementClass *element = GST_ELEMENT_CLASS(klass);
    GstBaseTransformClass *bt = GST_BASE_TRANSFORM_CLASS(klass);
    hipcopy_parent_class = g_type_class_peek_parent(klass);
    klass->upload = upload;
    GstCaps *sys = gst_caps_new_empty_simple("video/x-raw");
    GstCaps *hip = mvfx_caps_with_hip_feature(sys);
    gst_element_class_add_pad_template(element, gst_pad_template_new("sink", GST_PAD_SINK, GST_PAD_ALWAYS, upload ? sys : hip));
    gst_element_class_add_pad_template(element, gst_pad_template_new("src", GST_PAD_SRC, GST_PAD_ALWAYS, upload ? hip : sys));
    gst_caps_unref(sys);
    gst_caps_unref(hip);
    gst_element_class_set_static_metadata(element, upload ? "HIP uploader" : "HIP downloader", "Filter/Video",
                                          upload ? "Copies system-memory video frames into MI355X HBM (memory:HIPMemory)"
                                                 : "Copies memory:HIPMemory video frames back to system memory",
                                          "mi355-vfx");
    bt->transform_caps = hipcopy_transform_caps;
    bt->set_caps = hipcopy_set_caps;
    bt->prepare_output_buffer = hipcopy_prepare_output_buffer;
    bt->decide_allocation = hipcopy_decide_allocation;
    bt->propose_allocation = hipcopy_propose_allocation;
    bt->transform = hipcopy_transform;
    bt->start = hipcopy_start;
    bt->stop = hipcopy_stop;
    bt->passthrough_on_same_caps = FALSE;
    if (upload) {
        G_OBJECT_CLASS(klass)->set_property = hipcopy_set_property;
        G_OBJECT_CLASS(klass)->get_property = hipcopy_get_property;
        g_object_class_install_property(G_OBJECT_CLASS(klass), PROP_COPY_DEVICE_ID,
            g_param_spec_int("device-id", "Device ID", "HIP device the uploaded frames live on (-1 = the streaming thread's current device); the "
                             "elements downstream follow the device of their input memory", -1, G_MAXINT, -1,
                             (GParamFlags)(G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS)));
    }
}

static void hipupload_class_init(gpointer klass, gpointer) { hipcopy_class_init_common((GstMi355HipCopyClass *)klass, TRUE); }
static void hipdownload_class_init(gpointer klass, gpointer) { hipcopy_class_init_common((GstMi355HipCopyClass *)klass, FALSE); }
static void hipcopy_init(GTypeInstance *inst, gpointer)
{
    ((GstMi355HipCopy *)inst)->have_info = FALSE;
    ((GstMi355HipCopy *)inst)->pinned = NULL;
    ((GstMi355HipCopy *)inst)->device_id = -1;
}

static GType hipcopy_register(const gchar *name, GClassInitFunc class_init)
{
    GTypeInfo info;
    memset(&info, 0, sizeof(info));
    info.class_size = sizeof(GstMi355HipCopyClass);
    info.class_init = class_init;
    info.instance_size = sizeof(GstMi355HipCopy);
    info.instance_init = hipcopy_init;
    return g_type_register_static(GST_TYPE_BASE_TRANSFORM, name, &info, (GTypeFlags)0);
}


// ------------------------------------------------------------------------------------ hiptestsrc
// A generator-free frame source for throughput measurements (tools/bench_gst_pipeline.py): videotestsrc spends more time
// painting a 4K frame than the whole filter chain needs.  The frame is painted ONCE in set_caps -- for the packed RGB formats
// what `videotestsrc pattern=smpte` paints (bars, -I / white / +Q, super-black / black / grey, LCG snow: tests/frames.py has the
// same restatement, checked against the element), pseudo-random bytes for every other format.
//   system memory: every pool buffer is filled once (a 33 MB memcpy per 4K frame would cap the source at ~350 fps) and handed out
//                  again as it comes back -- fine for elements that leave system buffers untouched (hipupload, out-of-place
//                  filters); an in-place filter on system memory sees its own output again next time round;
//   HIP memory:    every frame is refreshed from a device-resident master with one asynchronous device-to-device copy on the
//                  streaming thread's stream, ordered by the block's fence (13 us of GPU time per 4K frame): in-place filters
//                  downstream (hsvfilter) always get the same input.
struct GstMi355HipTestSrc {
    GstPushSrc parent;
    GstVideoInfo info;
    gboolean have_info, hip;
    gboolean refresh;    // property: re-copy the master into every recycled device block (default); FALSE: fill each block once
    gint device_id;      // property: the device the frames are born on (-1: the streaming thread's current device)
    guint8 *pattern;
    gsize pattern_size;
    void *master;        // device copy of `pattern` (memory:HIPMemory caps)
    guint64 n;
    gint64 rate_t0;      // MVFX_TESTSRC_RATE=N: monotonic time when the block of buffer N had come back from downstream, work done
};
struct GstMi355HipTestSrcClass { GstPushSrcClass parent_class; };
G_DEFINE_TYPE(GstMi355HipTestSrc, gst_mi355_hip_test_src, GST_TYPE_PUSH_SRC)

static GQuark hiptestsrc_filled_quark(void) { return g_quark_from_static_string("mvfx-hiptestsrc-filled"); }

static GstCaps *hiptestsrc_fixate(GstBaseSrc *src, GstCaps *caps)
{
    caps = gst_caps_make_writable(caps);
    GstStructure *s = gst_caps_get_structure(caps, 0);
    // fields a downstream capsfilter left out are set, the others fixated to the nearest value (videotestsrc's defaults)
    if (gst_structure_has_field(s, "format")) gst_structure_fixate_field_string(s, "format", "RGBA");
    else gst_structure_set(s, "format", G_TYPE_STRING, "RGBA", NULL);
    if (gst_structure_has_field(s, "width")) gst_structure_fixate_field_nearest_int(s, "width", 320);
    else gst_structure_set(s, "width", G_TYPE_INT, 320, NULL);
    if (gst_structure_has_field(s, "height")) gst_structure_fixate_field_nearest_int(s, "height", 240);
    else gst_structure_set(s, "height", G_TYPE_INT, 240, NULL);
    if (gst_structure_has_field(s, "framerate")) gst_structure_fixate_field_nearest_fraction(s, "framerate", 30, 1);
    else gst_structure_set(s, "framerate", GST_TYPE_FRACTION, 30, 1, NULL);
    return GST_BASE_SRC_CLASS(gst_mi355_hip_test_src_parent_class)->fixate(src, caps);
}

// videotestsrc pattern=smpte, frame 0, for packed 8-bit RGB formats (3 or 4 bytes per pixel); FALSE for anything else
static gboolean hiptestsrc_paint_smpte(const GstVideoInfo *info, guint8 *dst)
{
    const GstVideoFormatInfo *f = info->finfo;
    if (!GST_VIDEO_FORMAT_INFO_IS_RGB(f) || GST_VIDEO_INFO_N_PLANES(info) != 1 || GST_VIDEO_FORMAT_INFO_DEPTH(f, 0) != 8)
        return FALSE;
    const gint bpp = GST_VIDEO_FORMAT_INFO_PSTRIDE(f, 0);
    if (bpp != 3 && bpp != 4) return FALSE;
    const gint w = GST_VIDEO_INFO_WIDTH(info), h = GST_VIDEO_INFO_HEIGHT(info), stride = GST_VIDEO_INFO_PLANE_STRIDE(info, 0);
    const gint ro = GST_VIDEO_FORMAT_INFO_POFFSET(f, 0), go = GST_VIDEO_FORMAT_INFO_POFFSET(f, 1), bo = GST_VIDEO_FORMAT_INFO_POFFSET(f, 2);
    static const guint8 bars[7][3] = {{255, 255, 255}, {255, 255, 0}, {0, 255, 255}, {0, 255, 0}, {255, 0, 255}, {255, 0, 0}, {0, 0, 255}};
    static const guint8 low[3][3] = {{0, 0, 128}, {255, 255, 255}, {0, 128, 255}};
    static const guint8 greys[3] = {0, 0, 19};
    const gint y1 = 2 * h / 3, y2 = 3 * h / 4, x_snow = w * 3 / 4;
    guint32 lcg = 0;
    memset(dst, 255, GST_VIDEO_INFO_SIZE(info)); // alpha / padding byte of the 4-byte formats
    for (gint y = 0; y < h; y++) {
        guint8 *row = dst + (gsize)y * stride;
        for (gint x = 0; x < w; x++) {
            guint8 r, g, b;
            if (y < y2) {
                gint i = 0;
                while (i < 6 && x >= (i + 1) * w / 7) i++;
                const guint8 *c = bars[y < y1 ? i : 6 - i];
                if (y >= y1 && (i & 1)) r = g = b = 0;
                else { r = c[0]; g = c[1]; b = c[2]; }
            } else if (x < w / 2) {
                gint i = 0;
                while (i < 2 && x >= (i + 1) * w / 6) i++;
                r = low[i][0]; g = low[i][1]; b = low[i][2];
            } else if (x < x_snow) {
                gint i = 0;
                while (i < 2 && x >= w / 2 + (i + 1) * w / 12) i++;
                r = g = b = greys[i];
            } else {
                lcg = lcg * 1103515245u + 12345u;
                r = g = b = (guint8)(lcg >> 16);
            }
            guint8 *px = row + (gsize)x * bpp;
            px[ro] = r; px[go] = g; px[bo] = b;
        }
    }
    return TRUE;
}

static gboolean hiptestsrc_set_caps(GstBaseSrc *src, GstCaps *caps)
{
    GstMi355HipTestSrc *self = (GstMi355HipTestSrc *)src;
    self->have_info = gst_video_info_from_caps(&self->info, caps);
    if (!self->have_info) return FALSE;
    self->hip = mvfx_caps_has_hip_feature(caps);
    g_free(self->pattern);
    self->pattern_size = GST_VIDEO_INFO_SIZE(&self->info);
    self->pattern = (guint8 *)g_malloc(self->pattern_size);
    if (!hiptestsrc_paint_smpte(&self->info, self->pattern)) {
        guint64 x = 0x5EED0001ull; // splitmix64 bytes
        for (gsize i = 0; i < self->pattern_size; i += 8) {
            x += 0x9E3779B97F4A7C15ull;
            guint64 z = x;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            z ^= z >> 31;
            memcpy(self->pattern + i, &z, MIN((gsize)8, self->pattern_size - i));
        }
    }
    if (self->master) {
        mvfx_device_free(self->master);
        self->master = NULL;
    }
    if (self->hip) {
        if (!mvfx_hip_select_device(self->device_id, GST_ELEMENT(self))) return FALSE; // the master and (below) the pool's blocks on `device-id`
        if (mvfx_device_alloc(&self->master, self->pattern_size) != MVFX_OK ||
            mvfx_copy_to_device(self->master, self->pattern, self->pattern_size, NULL) != MVFX_OK) {
            GST_ELEMENT_ERROR(self, RESOURCE, FAILED, ("hiptestsrc: %s", mvfx_last_error()), (NULL));
            return FALSE;
        }
    }
    gst_base_src_set_blocksize(src, (guint)self->pattern_size);
    return TRUE;
}

static gboolean hiptestsrc_decide_allocation(GstBaseSrc *src, GstQuery *query)
{
    // (the base class allocates the first buffer on this thread right after: a HIP thread keeps its current device)
    if (((GstMi355HipTestSrc *)src)->hip && !mvfx_hip_select_device(((GstMi355HipTestSrc *)src)->device_id, GST_ELEMENT(src))) return FALSE;
    // HIP pool when the caps carry memory:HIPMemory; otherwise downstream's pool (hipupload offers page-locked buffers) or,
    // when nobody proposed one, a plain video buffer pool of our own -- buffers must be RECYCLED for the fill-once scheme
    if (!mvfx_hip_decide_allocation(query) && gst_query_get_n_allocation_pools(query) == 0) {
        GstCaps *caps = NULL;
        gst_query_parse_allocation(query, &caps, NULL);
        GstVideoInfo info;
        if (caps && gst_video_info_from_caps(&info, caps)) {
            GstBufferPool *pool = gst_video_buffer_pool_new();
            GstStructure *config = gst_buffer_pool_get_config(pool);
            gst_buffer_pool_config_set_params(config, caps, (guint)GST_VIDEO_INFO_SIZE(&info), 2, 0);
            if (gst_buffer_pool_set_config(pool, config))
                gst_query_add_allocation_pool(query, pool, (guint)GST_VIDEO_INFO_SIZE(&info), 2, 0);
            gst_object_unref(pool);
        }
    }
    return GST_BASE_SRC_CLASS(gst_mi355_hip_test_src_parent_class)->decide_allocation(src, query);
}

static GstFlowReturn hiptestsrc_fill(GstPushSrc *psrc, GstBuffer *buf)
{
    GstMi355HipTestSrc *self = (GstMi355HipTestSrc *)psrc;
    if (!self->have_info) return GST_FLOW_NOT_NEGOTIATED;
    // timestamps first: the stream the device copy below runs on is picked from the buffer's frame number (mvfx_element_stream)
    const GstClockTime dur = GST_VIDEO_INFO_FPS_N(&self->info) > 0
        ? gst_util_uint64_scale_int(GST_SECOND, GST_VIDEO_INFO_FPS_D(&self->info), GST_VIDEO_INFO_FPS_N(&self->info)) : GST_CLOCK_TIME_NONE;
    GST_BUFFER_PTS(buf) = GST_CLOCK_TIME_IS_VALID(dur) ? self->n * dur : GST_CLOCK_TIME_NONE;
    GST_BUFFER_DTS(buf) = GST_CLOCK_TIME_NONE;
    GST_BUFFER_DURATION(buf) = dur;
    GST_BUFFER_OFFSET(buf) = self->n++;
    GstMemory *mem = gst_buffer_peek_memory(buf, 0);
    // MVFX_TESTSRC_RATE=N (measurement aid, memory:HIPMemory): the buffers per second of the whole pipeline between buffer N and the last
    // one of num-buffers, taken INSIDE the process at two like instants -- a recycled block is back AND everything downstream did on it
    // has finished (a host wait on its fence) -- so neither process start-up nor the tear-down is in the figure
    // (tools/bench_gst_pipeline.py had to difference two whole gst-launch runs for that)
    static const gint64 rate_mark = g_getenv("MVFX_TESTSRC_RATE") ? g_ascii_strtoll(g_getenv("MVFX_TESTSRC_RATE"), NULL, 10) : 0;
    if (rate_mark > 0 && self->hip && mvfx_buffer_is_hip(buf)) {
        const guint64 idx = self->n - 1;
        const gboolean last = GST_BASE_SRC(self)->num_buffers > 0 && idx + 1 == (guint64)GST_BASE_SRC(self)->num_buffers;
        if (idx == (guint64)rate_mark || (last && self->rate_t0 && idx > (guint64)rate_mark)) {
            mvfx_hip_memory_wait(mem);
            const gint64 now = g_get_monotonic_time();
            if (idx == (guint64)rate_mark)
                self->rate_t0 = now;
            else
                g_printerr("hiptestsrc %s: %.1f buffers/s between buffer %" G_GINT64_FORMAT " and buffer %" G_GUINT64_FORMAT "\n", GST_OBJECT_NAME(self),
                           (double)(idx - rate_mark) * 1e6 / (double)(now - self->rate_t0), rate_mark, idx);
        }
    }
    if (self->hip && self->master && mvfx_buffer_is_hip(buf)) {
        GstMapInfo map;
        if (!gst_buffer_map(buf, &map, (GstMapFlags)(MVFX_MAP_HIP | GST_MAP_WRITE))) return GST_FLOW_ERROR;
        int rc = MVFX_OK;
        // refresh=false (throughput measurements of a filter alone): a block is filled on its first trip only -- an in-place filter
        // downstream then works on its own output from the second trip on, and the source costs no HBM traffic
        if (self->refresh || !gst_mini_object_get_qdata(GST_MINI_OBJECT_CAST(mem), hiptestsrc_filled_quark())) {
            mvfx_stream st = mvfx_element_stream(buf); // the stream the filters downstream will pick for this frame
            mvfx_hip_buffer_acquire(buf, st);
            rc = mvfx_copy_device_to_device_async(map.data, self->master, MIN(self->pattern_size, map.size), st);
            mvfx_hip_buffer_release(buf, st);
            gst_mini_object_set_qdata(GST_MINI_OBJECT_CAST(mem), hiptestsrc_filled_quark(), GINT_TO_POINTER(1), NULL);
        }
        gst_buffer_unmap(buf, &map);
        if (rc != MVFX_OK) return MVFX_GST_FLOW(self, rc);
    } else if (!gst_mini_object_get_qdata(GST_MINI_OBJECT_CAST(mem), hiptestsrc_filled_quark())) {
        GstMapInfo map;
        if (!gst_buffer_map(buf, &map, GST_MAP_WRITE)) return GST_FLOW_ERROR;
        memcpy(map.data, self->pattern, MIN(self->pattern_size, map.size));
        gst_buffer_unmap(buf, &map);
        gst_mini_object_set_qdata(GST_MINI_OBJECT_CAST(mem), hiptestsrc_filled_quark(), GINT_TO_POINTER(1), NULL);
    }
    return GST_FLOW_OK;
}

// videotestsrc's: a live source waits on the clock for each buffer's timestamp (the base class does not by itself)
static void hiptestsrc_get_times(GstBaseSrc *src, GstBuffer *buffer, GstClockTime *start, GstClockTime *end)
{
    *start = *end = GST_CLOCK_TIME_NONE;
    if (!gst_base_src_is_live(src)) return;
    const GstClockTime ts = GST_BUFFER_PTS(buffer);
    if (!GST_CLOCK_TIME_IS_VALID(ts)) return;
    *start = ts;
    if (GST_CLOCK_TIME_IS_VALID(GST_BUFFER_DURATION(buffer))) *end = ts + GST_BUFFER_DURATION(buffer);
}

static gboolean hiptestsrc_start(GstBaseSrc *src)
{
    // a `device-id` that does not exist fails here, as a RESOURCE error of the element
    if (!mvfx_hip_select_device(((GstMi355HipTestSrc *)src)->device_id, GST_ELEMENT(src))) return FALSE;
    ((GstMi355HipTestSrc *)src)->n = 0;
    ((GstMi355HipTestSrc *)src)->rate_t0 = 0;
    return TRUE;
}

static void gst_mi355_hip_test_src_finalize(GObject *obj)
{
    GstMi355HipTestSrc *self = (GstMi355HipTestSrc *)obj;
    g_free(self->pattern);
    if (self->master) mvfx_device_free(self->master);
    G_OBJECT_CLASS(gst_mi355_hip_test_src_parent_class)->finalize(obj);
}

enum { PROP_TS_0, PROP_TS_REFRESH, PROP_TS_IS_LIVE, PROP_TS_DEVICE_ID };

static void hiptestsrc_set_property(GObject *obj, guint id, const GValue *value, GParamSpec *pspec)
{
    if (id == PROP_TS_REFRESH) ((GstMi355HipTestSrc *)obj)->refresh = g_value_get_boolean(value);
    else if (id == PROP_TS_IS_LIVE) gst_base_src_set_live(GST_BASE_SRC(obj), g_value_get_boolean(value));
    else if (id == PROP_TS_DEVICE_ID) ((GstMi355HipTestSrc *)obj)->device_id = g_value_get_int(value);
    else G_OBJECT_WARN_INVALID_PROPERTY_ID(obj, id, pspec);
}

static void hiptestsrc_get_property(GObject *obj, guint id, GValue *value, GParamSpec *pspec)
{
    if (id == PROP_TS_REFRESH) g_value_set_boolean(value, ((GstMi355HipTestSrc *)obj)->refresh);
    else if (id == PROP_TS_IS_LIVE) g_value_set_boolean(value, gst_base_src_is_live(GST_BASE_SRC(obj)));
    else if (id == PROP_TS_DEVICE_ID) g_value_set_int(value, ((GstMi355HipTestSrc *)obj)->device_id);
    else G_OBJECT_WARN_INVALID_PROPERTY_ID(obj, id, pspec);
}

static void gst_mi355_hip_test_src_class_init(GstMi355HipTestSrcClass *klass)
{
    GstElementClass *element = GST_ELEMENT_CLASS(klass);
    GstBaseSrcClass *bs = GST_BASE_SRC_CLASS(klass);
    G_OBJECT_CLASS(klass)->finalize = gst_mi355_hip_test_src_finalize;
    G_OBJECT_CLASS(klass)->set_property = hiptestsrc_set_property;
    G_OBJECT_CLASS(klass)->get_property = hiptestsrc_get_property;
    g_object_class_install_property(G_OBJECT_CLASS(klass), PROP_TS_REFRESH,
        g_param_spec_boolean("refresh", "Refresh", "memory:HIPMemory: copy the pattern into every recycled buffer (FALSE: fill each buffer once; "
                             "in-place filters downstream then see their own output again)", TRUE,
                             (GParamFlags)(G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS)));
    // like videotestsrc's: a live source hands out each frame at its timestamp (the base class waits on the clock), e.g. a camera
    g_object_class_install_property(G_OBJECT_CLASS(klass), PROP_TS_IS_LIVE,
        g_param_spec_boolean("is-live", "Is Live", "Whether to act as a live source (one buffer per frame interval of the caps)", FALSE,
                             (GParamFlags)(G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS)));
    g_object_class_install_property(G_OBJECT_CLASS(klass), PROP_TS_DEVICE_ID,
        g_param_spec_int("device-id", "Device ID", "memory:HIPMemory: HIP device the frames are born on (-1 = the streaming thread's current device)",
                         -1, G_MAXINT, -1, (GParamFlags)(G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS)));
    GstCaps *sys = gst_caps_new_empty_simple("video/x-raw");
    GstCaps *both = mvfx_caps_plus_hip(sys);
    gst_element_class_add_pad_template(element, gst_pad_template_new("src", GST_PAD_SRC, GST_PAD_ALWAYS, both));
    gst_caps_unref(both);
    gst_element_class_set_static_metadata(element, "HIP test source", "Source/Video",
                                          "Pre-painted videotestsrc-smpte frames in system or memory:HIPMemory buffers (no per-frame generator cost)",
                                          "mi355-vfx");
    bs->fixate = hiptestsrc_fixate;
    bs->set_caps = hiptestsrc_set_caps;
    bs->decide_allocation = hiptestsrc_decide_allocation;
    bs->start = hiptestsrc_start;
    bs->get_times = hiptestsrc_get_times;
    GST_PUSH_SRC_CLASS(klass)->fill = hiptestsrc_fill;
}

static void gst_mi355_hip_test_src_init(GstMi355HipTestSrc *self)
{
    self->have_info = self->hip = FALSE;
    self->refresh = TRUE;
    self->device_id = -1;
    self->pattern = NULL;
    self->pattern_size = 0;
    self->master = NULL;
    self->n = 0;
    self->rate_t0 = 0;
    gst_base_src_set_format(GST_BASE_SRC(self), GST_FORMAT_TIME);
}

static gboolean plugin_init(GstPlugin *plugin)
{
    GST_DEBUG_CATEGORY_INIT(mi355hip_debug, "mi355hip", 0, "MI355X HIP memory upload/download");
    return gst_element_register(plugin, "hipupload", GST_RANK_NONE, hipcopy_register("GstMi355HipUpload", hipupload_class_init)) &&
           gst_element_register(plugin, "hipdownload", GST_RANK_NONE, hipcopy_register("GstMi355HipDownload", hipdownload_class_init)) &&
           gst_element_register(plugin, "hiptestsrc", GST_RANK_NONE, gst_mi355_hip_test_src_get_type());
}

GST_PLUGIN_DEFINE(GST_VERSION_MAJOR, GST_VERSION_MINOR, mi355hip, "MI355X HIP device-memory bridge elements", plugin_init,
                  MVFX_GST_VERSION, "MIT/X11", "mi355-vfx", MVFX_GST_ORIGIN)
