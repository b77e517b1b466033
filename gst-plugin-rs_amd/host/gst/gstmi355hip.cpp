// libgstmi355hip.so -- plugin `mi355hip`: `hipupload` / `hipdownload`, the bridges between system
// memory and `video/x-raw(memory:HIPMemory)`.  With them a chain such as
//   videotestsrc ! hipupload ! hsvfilter ! hsvdetector ! colorlut ! hipdownload ! ...
// crosses PCIe once in and once out instead of twice per element.
#include "mvfx_gst_common.h"
#include "mvfxhipmemory.h"

GST_DEBUG_CATEGORY_STATIC(mi355hip_debug);

struct GstMi355HipCopy {
    GstBaseTransform parent;
    GstVideoInfo info;
    gboolean have_info;
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
    return self->have_info;
}

static GstFlowReturn hipcopy_prepare_output_buffer(GstBaseTransform *trans, GstBuffer *inbuf, GstBuffer **outbuf)
{
    GstMi355HipCopy *self = (GstMi355HipCopy *)trans;
    const gboolean upload = ((GstMi355HipCopyClass *)G_OBJECT_GET_CLASS(trans))->upload;
    if (!self->have_info)
        return GST_FLOW_NOT_NEGOTIATED;
    *outbuf = NULL;
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
    mvfx_hip_propose_allocation(query); // no-op unless the sink caps carry memory:HIPMemory
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
    // device buffers always use the default GstVideoInfo layout (offsets / strides of `info`)
    for (guint p = 0; p < GST_VIDEO_INFO_N_PLANES(&self->info) && rc == MVFX_OK; p++) {
        guint8 *d = dmap.data + GST_VIDEO_INFO_PLANE_OFFSET(&self->info, p);
        guint8 *s = (guint8 *)GST_VIDEO_FRAME_PLANE_DATA(&frame, p);
        const gint dstride = GST_VIDEO_INFO_PLANE_STRIDE(&self->info, p), sstride = GST_VIDEO_FRAME_PLANE_STRIDE(&frame, p);
        const guint rows = GST_VIDEO_FRAME_COMP_HEIGHT(&frame, p == 3 ? 3 : p);
        if (dstride == sstride) {
            rc = upload ? mvfx_copy_to_device(d, s, (size_t)dstride * rows, NULL) : mvfx_copy_to_host(s, d, (size_t)dstride * rows, NULL);
        } else {
            const size_t row = (size_t)MIN(dstride, sstride);
            for (guint y = 0; y < rows && rc == MVFX_OK; y++)
                rc = upload ? mvfx_copy_to_device(d + (size_t)y * dstride, s + (size_t)y * sstride, row, NULL)
                            : mvfx_copy_to_host(s + (size_t)y * sstride, d + (size_t)y * dstride, row, NULL);
        }
    }
    gst_buffer_unmap(dev, &dmap);
    gst_video_frame_unmap(&frame);
    return MVFX_GST_FLOW(trans, rc);
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
    bt->passthrough_on_same_caps = FALSE;
}

static void hipupload_class_init(gpointer klass, gpointer) { hipcopy_class_init_common((GstMi355HipCopyClass *)klass, TRUE); }
static void hipdownload_class_init(gpointer klass, gpointer) { hipcopy_class_init_common((GstMi355HipCopyClass *)klass, FALSE); }
static void hipcopy_init(GTypeInstance *inst, gpointer) { ((GstMi355HipCopy *)inst)->have_info = FALSE; }

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

static gboolean plugin_init(GstPlugin *plugin)
{
    GST_DEBUG_CATEGORY_INIT(mi355hip_debug, "mi355hip", 0, "MI355X HIP memory upload/download");
    return gst_element_register(plugin, "hipupload", GST_RANK_NONE, hipcopy_register("GstMi355HipUpload", hipupload_class_init)) &&
           gst_element_register(plugin, "hipdownload", GST_RANK_NONE, hipcopy_register("GstMi355HipDownload", hipdownload_class_init));
}

GST_PLUGIN_DEFINE(GST_VERSION_MAJOR, GST_VERSION_MINOR, mi355hip, "MI355X HIP device-memory bridge elements", plugin_init,
                  MVFX_GST_VERSION, "MIT/X11", "mi355-vfx", MVFX_GST_ORIGIN)
