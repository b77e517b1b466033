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
void mvfx_hip_allocator_trim(void);                   // returns the cached device blocks of the free list to HIP

// ---- fences instead of per-buffer stream synchronisation (d3d12colorlut/imp.rs:695-714 sets a fence on the output
// memory and returns).  An element brackets the kernels it enqueues on `stream` for a device buffer:
//     mvfx_hip_buffer_acquire(buf, stream);   // stream waits (on the device) for whoever used the block last
//     ... launch ...
//     mvfx_hip_buffer_release(buf, stream);   // record the block's fence on the stream; no host wait
// A CPU map of the memory (and hipdownload) waits for the fence on the host.
void mvfx_hip_memory_acquire(GstMemory *mem, void *stream);
void mvfx_hip_memory_release(GstMemory *mem, void *stream);
void mvfx_hip_memory_wait(GstMemory *mem);
// Fences as plain events, for work that is not enqueued on a stream of the element's (the launch combiner's fenced entry):
// the event a new user of the block has to wait for (NULL: none pending), and a borrowed event -- owned by somebody else, valid for
// the life of the process -- as the block's new fence (it stands in front of the block's own event until the next release).
// Deferred work (round 4): an element may hold a buffer's kernel back (hsvfilter launches two consecutive frames together) and say so on
// the block: `flush(owner)` is called -- with no lock of the memory held -- by the NEXT user of the block before it looks at the fence
// (mvfx_hip_memory_acquire / _wait / _pending_fence / _release, a CPU map); it must launch the work and mvfx_hip_memory_release the
// block (after mvfx_hip_memory_clear_deferred).  The mark holds a reference on `owner` (a GstObject); the owner keeps one on the memory.
typedef void (*MvfxDeferredFlush)(GstObject *owner);
void mvfx_hip_memory_set_deferred(GstMemory *mem, MvfxDeferredFlush flush, GstObject *owner);
void mvfx_hip_memory_clear_deferred(GstMemory *mem, GstObject *owner); // no-op unless the mark is this owner's
// Idle flush: `cb(owner)` runs on a process-wide timer thread `after_us` from now unless re-armed (the deadline moves) or cancelled
// first; one entry per owner, which is referenced while armed.  A held-back frame never waits longer than one frame interval.
void mvfx_idle_arm(GstObject *owner, MvfxDeferredFlush cb, guint64 after_us);
void mvfx_idle_cancel(GstObject *owner);
// The owner launching its held-back work: acquire WITHOUT flushing its own mark, release that records the fence and drops the mark in one
// critical section (a consumer on another thread must never see "no mark, no fence yet").
void mvfx_hip_memory_flush_foreign(GstMemory *mem, GstObject *owner);
gboolean mvfx_hip_memory_busy(GstMemory *mem, GstObject *owner);
void mvfx_hip_memory_release_tagged(GstMemory *mem, void *stream, GstObject *tag);
void mvfx_hip_buffers_release(GstBuffer *a, GstBuffer *b, void *stream);
// one launch's fence, split around the launch (mvfxhipmemory.cpp): declare on the stack, _begin, launch through the C ABI, _end
typedef struct { GstMemory *mems[8]; guint64 seen[8]; guint n; gpointer fence; } MvfxFenceScope;
void mvfx_hip_fence_begin(MvfxFenceScope *scope, GstMemory *const *mems, guint n, void *stream, gboolean plain);
void mvfx_hip_fence_begin_buffers(MvfxFenceScope *scope, GstBuffer *a, GstBuffer *b, void *stream);
void mvfx_hip_fence_end(MvfxFenceScope *scope, void *stream, GstObject *owner, GstObject *tag);
void mvfx_hip_memories_release_tagged(GstMemory *const *mems, guint n, void *stream, GstObject *tag);
void mvfx_hip_memories_release_as_owner(GstMemory *const *mems, guint n, void *stream, GstObject *owner);
void mvfx_hip_memory_acquire_as_owner(GstMemory *mem, void *stream, GstObject *owner);
void mvfx_hip_memory_release_as_owner(GstMemory *mem, void *stream, GstObject *owner);
void *mvfx_hip_memory_pending_fence(GstMemory *mem);
// ---- the direct-dispatch lane (round 6; include/mi355vfx.h MVFX_OPT_DIRECT_DISPATCH) as the elements use it ----
// The acquire of a frame that is to go out on lane queue `queue` (mvfx_direct_queue_of_stream(stream)).  TRUE when every block of the buffer is
// free to be worked on by such a dispatch: no held-back work, no borrowed fence still running, and the block's own fence either fired, or a direct
// dispatch IN FRONT of ours on the same lane queue (the queues are in order: a filter's frame and the detector that reads it), or a direct dispatch
// on the other queue -- which our queue then waits for ON THE DEVICE (a barrier packet with its completion signal: mvfx_direct_queue_wait_event; the
// fence stays referenced by ours until our dispatch has finished).  FALSE: an ordinary stream fence is still pending (a source's copy, a stream
// kernel) and the pipeline's start-up budget of waits for such fences ("seeding") is used up, or the lane is parked because its acquires were mostly
// refused (mvfxhipmemory.cpp, "PARKING"): acquire and launch on the stream as ever.
// After TRUE the element opens its fence scope and calls the library with MVFX_OPT_DIRECT_DISPATCH | MVFX_OPT_DIRECT_ONLY; on
// MVFX_ERR_DIRECT_UNAVAILABLE it cancels the scope (mvfx_hip_fence_cancel) and takes the ordinary path.
gboolean mvfx_hip_buffer_acquire_direct(GstBuffer *buf, void *stream, int queue);
// The same; *relied_on_order is SET (never cleared) when a block's fence was found pending on the same lane queue, i.e. when the TRUE rests on that
// queue being in order.  An element whose buffers all came back without it may send its packet without the barrier bit (MVFX_OPT_DIRECT_UNORDERED).
gboolean mvfx_hip_buffer_acquire_direct_ordered(GstBuffer *buf, void *stream, int queue, gboolean *relied_on_order);
void mvfx_hip_fence_cancel(MvfxFenceScope *scope);
// A consumer that found a direct fence still PENDING had to wait on its own thread (a HIP stream cannot wait for it on the device): the producer
// named by the fence's tag is told, and goes back to launching on its streams -- a direct dispatch pays where nobody is close behind the frame
// (hsvfilter ! fakesink, ! queue ! encoder ...), inside a tight device chain (hsvfilter ! hsvdetector on one thread) the stream order is the
// better fence.  With hysteresis: after eight such waits, and for 2048 frames, then the producer tries the lane again.  _reset at stop().
gboolean mvfx_direct_discouraged(const void *producer_tag);
void mvfx_direct_reset(const void *producer_tag);
void mvfx_hip_memory_set_borrowed_fence(GstMemory *mem, void *event);
void mvfx_hip_buffer_acquire(GstBuffer *buf, void *stream);
void mvfx_hip_buffer_release(GstBuffer *buf, void *stream);
gboolean mvfx_is_hip_memory(GstMemory *mem);
// ---- devices (round 6).  A block lives on the device that was the allocating thread's current one; `hipupload` / `hiptestsrc` choose it with
// their `device-id` property.  Every element that works on device buffers adopts the device of its INPUT memory before it acquires the block
// (the reference's d3d12colorlut: "Device updated from ... to ...", video/colorlut/src/d3d12colorlut/imp.rs:494-542): the streaming thread's
// current device is set, so its private streams, scratch, output blocks and LUT replica are that device's.
int mvfx_hip_memory_device(GstMemory *mem);            // ordinal, -1 for memory that is not ours
int mvfx_hip_buffer_device(GstBuffer *buf);            // of the buffer's first memory
// Makes the device of `buf`'s memory the calling thread's current device.  TRUE when it already was or the switch succeeded (`owner`, may be NULL,
// logs the switch); FALSE + GST_ELEMENT_ERROR(RESOURCE) on `owner` when the device cannot be selected.  Buffers that are not HIP memory: TRUE.
gboolean mvfx_hip_follow_device(GstBuffer *buf, GstObject *owner);
// `device-id` of an element that CREATES device buffers: -1 keeps the calling thread's current device; otherwise validates the ordinal against
// the visible devices and makes it current.  FALSE + GST_ELEMENT_ERROR(RESOURCE, NOT_FOUND) when there is no such device.
gboolean mvfx_hip_select_device(gint device_id, GstElement *owner);
gboolean mvfx_buffer_is_hip(GstBuffer *buf);          // single HIP memory holding the whole frame
gboolean mvfx_caps_has_hip_feature(const GstCaps *caps);
// "video/x-raw(memory:HIPMemory), format={...}, ..." twin of a system-memory caps
GstCaps *mvfx_caps_with_hip_feature(const GstCaps *system_caps);
// copy of `caps` with every structure's features replaced by memory:HIPMemory / system memory
GstCaps *mvfx_caps_set_hip_feature(const GstCaps *caps, gboolean hip);


// ---- buffer pool + ALLOCATION query helpers (the reference's d3d12colorlut does the same with
// GstD3D12BufferPool: video/colorlut/src/d3d12colorlut/imp.rs:385-492) ----
GstBufferPool *mvfx_hip_buffer_pool_new(void);        // buffers = ONE HIP memory of the configured size + GstVideoMeta
gboolean mvfx_is_hip_buffer_pool(GstBufferPool *pool);
// propose_allocation of an element whose SINK caps carry memory:HIPMemory: offers a HIP pool of the frame
// size, the HIP allocator and GstVideoMeta support.  Returns FALSE (nothing added) for system-memory caps.
gboolean mvfx_hip_propose_allocation(GstQuery *query);
// decide_allocation of an element whose SRC caps carry memory:HIPMemory: makes sure pool 0 of the query is a
// HIP pool (the one downstream proposed, or a new one).  Call before chaining up to the base class.
gboolean mvfx_hip_decide_allocation(GstQuery *query);
// System-memory buffers backed by page-locked host memory (hipHostMalloc): what hipupload offers upstream and what
// hipdownload hands downstream, so that the PCIe copies are plain DMA instead of the runtime's pageable staging.
GstBufferPool *mvfx_pinned_buffer_pool_new(void);
GstBufferPool *mvfx_pinned_buffer_pool_new_configured(GstCaps *caps, guint size, guint min_buffers);
guint64 mvfx_hip_pool_buffers_allocated(void);        // process-wide counters for tests / debugging
guint64 mvfx_hip_pool_buffers_acquired(void);

G_END_DECLS
