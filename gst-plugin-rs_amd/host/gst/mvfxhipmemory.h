// HIP device memory for GStreamer buffers: `video/x-raw(memory:HIPMemory)` (modelled on
// the reference's d3d12colorlut: video/colorlut/src/d3d12colorlut/imp.rs:385-492 allocation
// queries, :544-719 transform on device memory).  Lives in libmvfxgst.so so that the GType is
// registered once per process and shared by the three element plugins and the hipupload /
// hipdownload plugin.
#pragma once

#include <gst/gst.h>
#include <gst/video/video.h>

G_BEGIN_DECLS

#define MVFX_CAPS_FEATURE_MEMORY_HIP "memory:HIPMemory"
#define MVFX_HIP_MEMORY_TYPE "HIPMemory"
// map flag: GstMapInfo.data is the DEVICE pointer (no copy).  A plain READ/WRITE map of HIP
// memory still works for CPU elements: it goes through a host shadow copy (D2H on map for READ,
// H2D on unmap for WRITE).
#define MVFX_MAP_HIP ((GstMapFlags)(GST_MAP_FLAG_LAST << 1))

GstAllocator *mvfx_hip_allocator_get(void);           // singleton, new reference
gboolean mvfx_is_hip_memory(GstMemory *mem);
gboolean mvfx_buffer_is_hip(GstBuffer *buf);          // single HIP memory holding the whole frame
gboolean mvfx_caps_has_hip_feature(const GstCaps *caps);
// "video/x-raw(memory:HIPMemory), format={...}, ..." twin of a system-memory caps
GstCaps *mvfx_caps_with_hip_feature(const GstCaps *system_caps);
// copy of `caps` with every structure's features replaced by memory:HIPMemory / system memory
GstCaps *mvfx_caps_set_hip_feature(const GstCaps *caps, gboolean hip);

G_END_DECLS
