// Shared glue for the GStreamer elements that sit on top of the C ABI (include/mi355vfx.h).
// The elements keep the reference's factory names, GType names, klass strings, pad caps and
// GObject properties and forward the mapped GstVideoFrame to the HIP kernels
// through the *_host entry points (system-memory buffers: H2D -> kernel -> D2H inside the call,
// because a GstVideoFilter vfunc only borrows the frame).
#pragma once

#include <gst/base/gstbasetransform.h>
#include <gst/gst.h>
#include <gst/video/gstvideofilter.h>
#include <gst/video/video.h>

#include "mi355vfx.h"
#include "mvfxhipmemory.h"

#ifndef PACKAGE
#define PACKAGE "mi355vfx"
#endif
#ifndef MVFX_GST_VERSION
#define MVFX_GST_VERSION "0.16.0-mi355vfx"
#endif
#define MVFX_GST_ORIGIN "https://gitlab.freedesktop.org/gstreamer/gst-plugins-rs"

// GstVideoFormat -> mvfx_format; -1 for formats outside the path
static inline int mvfx_format_from_gst(GstVideoFormat f)
{
    switch (f) {
    case GST_VIDEO_FORMAT_RGBx: return MVFX_FORMAT_RGBX;
    case GST_VIDEO_FORMAT_xRGB: return MVFX_FORMAT_XRGB;
    case GST_VIDEO_FORMAT_BGRx: return MVFX_FORMAT_BGRX;
    case GST_VIDEO_FORMAT_xBGR: return MVFX_FORMAT_XBGR;
    case GST_VIDEO_FORMAT_RGBA: return MVFX_FORMAT_RGBA;
    case GST_VIDEO_FORMAT_ARGB: return MVFX_FORMAT_ARGB;
    case GST_VIDEO_FORMAT_BGRA: return MVFX_FORMAT_BGRA;
    case GST_VIDEO_FORMAT_ABGR: return MVFX_FORMAT_ABGR;
    case GST_VIDEO_FORMAT_RGB: return MVFX_FORMAT_RGB;
    case GST_VIDEO_FORMAT_BGR: return MVFX_FORMAT_BGR;
    case GST_VIDEO_FORMAT_I420: return MVFX_FORMAT_I420;
    case GST_VIDEO_FORMAT_A420: return MVFX_FORMAT_A420;
    default: break;
    }
    // RGBA64_LE / RGBA64_BE only exist in newer GStreamer; resolve them by name at run time
    const gchar *name = gst_video_format_to_string(f);
    if (name && g_strcmp0(name, "RGBA64_LE") == 0) return MVFX_FORMAT_RGBA64_LE;
    if (name && g_strcmp0(name, "RGBA64_BE") == 0) return MVFX_FORMAT_RGBA64_BE;
    return -1;
}

// plane 0 of a mapped frame as the C ABI's view (plane_data(0), plane_stride()[0], ...)
static inline mvfx_frame mvfx_frame_from_gst(GstVideoFrame *frame)
{
    mvfx_frame f;
    f.data = GST_VIDEO_FRAME_PLANE_DATA(frame, 0);
    f.width = (uint32_t)GST_VIDEO_FRAME_WIDTH(frame);
    f.height = (uint32_t)GST_VIDEO_FRAME_HEIGHT(frame);
    f.stride = (uint32_t)GST_VIDEO_FRAME_PLANE_STRIDE(frame, 0);
    f.format = mvfx_format_from_gst(GST_VIDEO_FRAME_FORMAT(frame));
    return f;
}

// mvfx status -> GstFlowReturn, posting an element error like the reference's
// element_imp_error! / panic-to-error conversion does (SURVEY.md 8b "Errors")
static inline GstFlowReturn mvfx_gst_flow(GstElement *element, int rc)
{
    if (rc == MVFX_OK)
        return GST_FLOW_OK;
    GST_ELEMENT_ERROR(element, LIBRARY, FAILED, ("%s", mvfx_last_error()),
                      ("mvfx status %d (%s)", rc, mvfx_status_string(rc)));
    return rc == MVFX_ERR_NOT_NEGOTIATED ? GST_FLOW_NOT_NEGOTIATED : GST_FLOW_ERROR;
}
#define MVFX_GST_FLOW(element, rc) mvfx_gst_flow(GST_ELEMENT(element), (rc))

// "video/x-raw, format={...}, width=[1,max], height=[1,max], framerate=[0/1,max]" -- what
// gst_video::VideoCapsBuilder::new().format_list(..).build() produces
static inline GstCaps *mvfx_video_caps(const gchar *const *formats)
{
    GValue list = G_VALUE_INIT;
    g_value_init(&list, GST_TYPE_LIST);
    guint n = 0;
    for (const gchar *const *f = formats; *f; f++) {
        GValue v = G_VALUE_INIT;
        g_value_init(&v, G_TYPE_STRING);
        g_value_set_string(&v, *f);
        gst_value_list_append_and_take_value(&list, &v);
        n++;
    }
    GstCaps *caps = gst_caps_new_simple("video/x-raw", "width", GST_TYPE_INT_RANGE, 1, G_MAXINT, "height",
                                        GST_TYPE_INT_RANGE, 1, G_MAXINT, "framerate", GST_TYPE_FRACTION_RANGE, 0, 1,
                                        G_MAXINT, 1, NULL);
    GstStructure *s = gst_caps_get_structure(caps, 0);
    if (n == 1) {
        gst_structure_set(s, "format", G_TYPE_STRING, formats[0], NULL);
        g_value_unset(&list);
    } else {
        gst_structure_take_value(s, "format", &list);
    }
    return caps;
}

// ---- video/x-raw(memory:HIPMemory) support shared by the elements ----

// appends the memory:HIPMemory twin of every structure
static inline GstCaps *mvfx_caps_plus_hip(GstCaps *caps)
{
    GstCaps *hip = mvfx_caps_with_hip_feature(caps);
    gst_caps_append(caps, hip);
    return caps;
}

// plane 0 of a HIP buffer (default GstVideoInfo layout) as the C ABI's view; data = DEVICE pointer
static inline gboolean mvfx_hip_map_frame(GstBuffer *buf, const GstVideoInfo *info, GstMapFlags rw, GstMapInfo *map, mvfx_frame *f)
{
    if (!gst_buffer_map(buf, map, (GstMapFlags)(MVFX_MAP_HIP | rw)))
        return FALSE;
    f->data = map->data + GST_VIDEO_INFO_PLANE_OFFSET(info, 0);
    f->width = (uint32_t)GST_VIDEO_INFO_WIDTH(info);
    f->height = (uint32_t)GST_VIDEO_INFO_HEIGHT(info);
    f->stride = (uint32_t)GST_VIDEO_INFO_PLANE_STRIDE(info, 0);
    f->format = mvfx_format_from_gst(GST_VIDEO_INFO_FORMAT(info));
    return TRUE;
}

// all planes of a HIP buffer holding an I420 frame (default GstVideoInfo layout); data = DEVICE pointers
static inline gboolean mvfx_hip_map_i420(GstBuffer *buf, const GstVideoInfo *info, GstMapFlags rw, GstMapInfo *map, mvfx_planar_frame *f)
{
    if (GST_VIDEO_INFO_FORMAT(info) != GST_VIDEO_FORMAT_I420 || !gst_buffer_map(buf, map, (GstMapFlags)(MVFX_MAP_HIP | rw)))
        return FALSE;
    memset(f, 0, sizeof(*f));
    for (guint p = 0; p < 3; p++) {
        f->data[p] = map->data + GST_VIDEO_INFO_PLANE_OFFSET(info, p);
        f->stride[p] = (uint32_t)GST_VIDEO_INFO_PLANE_STRIDE(info, p);
    }
    f->width = (uint32_t)GST_VIDEO_INFO_WIDTH(info);
    f->height = (uint32_t)GST_VIDEO_INFO_HEIGHT(info);
    f->format = MVFX_FORMAT_I420;
    return TRUE;
}

// prepare_output_buffer for elements whose negotiated OUTPUT is HIP memory: a device buffer of
// the output frame size with the input's flags and timestamps
static inline GstFlowReturn mvfx_hip_new_output(GstBaseTransform *trans, GstBuffer *inbuf, gsize size, GstBuffer **outbuf)
{
    *outbuf = NULL;
    if (!mvfx_hip_follow_device(inbuf, GST_OBJECT(trans))) // the output block is allocated on the input's device
        return GST_FLOW_ERROR;
    GstBufferPool *pool = gst_base_transform_get_buffer_pool(trans); // negotiated by decide_allocation
    if (pool) {
        if (mvfx_is_hip_buffer_pool(pool)) {
            if (!gst_buffer_pool_is_active(pool))
                gst_buffer_pool_set_active(pool, TRUE);
            if (gst_buffer_pool_acquire_buffer(pool, outbuf, NULL) != GST_FLOW_OK)
                *outbuf = NULL;
        }
        gst_object_unref(pool);
    }
    if (*outbuf && mvfx_buffer_is_hip(inbuf) && mvfx_hip_buffer_device(*outbuf) != mvfx_hip_buffer_device(inbuf)) {
        // the pool's blocks are from the device the stream was on before the incoming memory changed device: a one-off block on the new one
        gst_buffer_unref(*outbuf);
        *outbuf = NULL;
    }
    if (!*outbuf) { // no ALLOCATION query answered yet / foreign pool: a one-off device buffer
        GstAllocator *alloc = mvfx_hip_allocator_get();
        *outbuf = gst_buffer_new_allocate(alloc, size, NULL);
        gst_object_unref(alloc);
    }
    if (!*outbuf) {
        GST_ELEMENT_ERROR(trans, RESOURCE, NO_SPACE_LEFT, ("%s", mvfx_last_error()), (NULL));
        return GST_FLOW_ERROR;
    }
    gst_buffer_copy_into(*outbuf, inbuf, (GstBufferCopyFlags)(GST_BUFFER_COPY_FLAGS | GST_BUFFER_COPY_TIMESTAMPS), 0, -1);
    return GST_FLOW_OK;
}

// propose_allocation / decide_allocation shared by the elements: the base class first, then the HIP pool
// when the negotiated caps carry memory:HIPMemory (no-ops on system-memory caps)
#define MVFX_DEFINE_HIP_ALLOCATION_VFUNCS(prefix, parent_class_ptr)                                                   \
    static gboolean prefix##_propose_allocation(GstBaseTransform *trans, GstQuery *decide_query, GstQuery *query)     \
    {                                                                                                                 \
        if (!GST_BASE_TRANSFORM_CLASS(parent_class_ptr)->propose_allocation(trans, decide_query, query))              \
            return FALSE;                                                                                             \
        mvfx_hip_propose_allocation(query);                                                                           \
        return TRUE;                                                                                                  \
    }                                                                                                                 \
    static gboolean prefix##_decide_allocation(GstBaseTransform *trans, GstQuery *query)                              \
    {                                                                                                                 \
        mvfx_hip_decide_allocation(query);                                                                            \
        return GST_BASE_TRANSFORM_CLASS(parent_class_ptr)->decide_allocation(trans, query);                           \
    }

// The HIP stream a device-memory element enqueues THIS buffer on.  Consecutive buffers of a video stream are independent frames and
// every buffer carries its own fence, so a streaming thread may alternate between private streams (mvfx_thread_stream_n): the tail
// of one frame's kernel then overlaps the head of the next one's instead of running back to back on one stream (4K hsvfilter, one
// streaming thread: 0.52 -> 0.67 of the HBM peak, profiles/r3/stream_rotation.txt).  The index comes from the BUFFER -- its frame
// number pts / duration, else its offset -- not from a counter of the element: every element of a chain then puts the same frame on the
// same stream (no cross-stream fence wait inside a frame's chain; a per-element counter cost the three-filter chain 20 %), and
// consecutive frames alternate.  MVFX_ELEMENT_STREAMS = 1..4 (environment, read once).
#ifndef MVFX_ELEMENT_STREAMS_DEFAULT
#define MVFX_ELEMENT_STREAMS_DEFAULT 2
#endif
static inline mvfx_stream mvfx_element_stream(GstBuffer *buf)
{
    // the streaming thread adopts the device of the incoming memory first (d3d12colorlut/imp.rs:494-542): the streams below, the LUT replica, the
    // output blocks are then that device's.  One compare per buffer when nothing changes.
    mvfx_hip_follow_device(buf, NULL);
    static const guint n = [] {
        const gchar *e = g_getenv("MVFX_ELEMENT_STREAMS");
        const int v = e ? atoi(e) : MVFX_ELEMENT_STREAMS_DEFAULT;
        return (guint)CLAMP(v, 1, 4);
    }();
    if (n == 1) return mvfx_thread_stream();
    guint64 frame;
    if (buf && GST_BUFFER_PTS_IS_VALID(buf) && GST_BUFFER_DURATION_IS_VALID(buf) && GST_BUFFER_DURATION(buf) > 0)
        frame = (GST_BUFFER_PTS(buf) + GST_BUFFER_DURATION(buf) / 2) / GST_BUFFER_DURATION(buf);
    else if (buf && GST_BUFFER_OFFSET_IS_VALID(buf))
        frame = GST_BUFFER_OFFSET(buf);
    else {
        static thread_local guint64 counter = 0;
        frame = counter++;
    }
    return mvfx_thread_stream_n((guint)(frame % n));
}

static inline void mvfx_add_pad_templates(GstElementClass *klass, GstCaps *sink_caps, GstCaps *src_caps)
{
    gst_element_class_add_pad_template(klass, gst_pad_template_new("sink", GST_PAD_SINK, GST_PAD_ALWAYS, sink_caps));
    gst_element_class_add_pad_template(klass, gst_pad_template_new("src", GST_PAD_SRC, GST_PAD_ALWAYS, src_caps));
    gst_caps_unref(sink_caps);
    gst_caps_unref(src_caps);
}
