// Pair launches for the OUT-OF-PLACE elements (hsvdetector, colorlut) on memory:HIPMemory buffers.
//
// One 4K frame per launch fills and drains the chip for 33-66 MB; two consecutive frames in one launch on two alternating streams
// close most of the distance to the batched entries (profiles/r4/element_path.txt; hsvfilter's in-place original of this is
// gst_hsv_filter_bt_transform_ip).  The element contract stays one transform() call per buffer (hsvdetector/imp.rs:422,
// colorlut/imp.rs:203): the call returns at once, the frame's kernel is HELD BACK, both of its blocks -- the input, which the
// upstream pool may hand out again, and the output, which the next element reads -- carry the element's deferred mark
// (mvfx_hip_memory_set_deferred), and the kernel leaves together with the next buffer's frame: or alone, the moment anybody looks at
// either block's fence (the next element's acquire, a CPU map, hipdownload, the source refilling a recycled block), when the
// geometry or the settings change, at EOS, on flush-start and in stop().
//
// Holding back only pays when nobody looks: behind another device element every held-back frame is flushed alone by that element's
// first look, and the chain hsvfilter ! hsvdetector ! colorlut ran at 13.3 k instead of 17.3 k fps with all three holding
// (profiles/r4/element_pairs.txt).  So the hold watches itself: kMvfxPairStreak held-back frames in a row that somebody else's look
// flushed switch the element to a plain launch per buffer for the next kMvfxPairDirect buffers, then it tries again.  And a buffer
// whose INPUT block is still being written when it arrives (mvfx_hip_memory_busy: an upstream kernel or the source's copy in flight,
// or held back) is launched the plain way at once: the LAST element of that chain paired cost 22.8 k -> 14.8 k fps, because a pair
// waits for the second frame's whole upstream chain where two single frames pipeline on two streams.
#pragma once

#include "mvfx_gst_common.h"

#include <mutex>

struct MvfxPairHold {
    std::mutex lock;
    GstMemory *in_mem = NULL, *out_mem = NULL; // referenced while set
    mvfx_frame fi, fo;
    mvfx_stream st = NULL;                      // the stream a lone launch of the held-back frame goes on
    guint pair_no = 0;
    guint foreign_streak = 0, direct_left = 0;  // see above
    guint64 n_buffers = 0, n_pairs = 0, n_singles = 0, n_direct = 0;
};
constexpr guint kMvfxPairStreak = 4, kMvfxPairDirect = 1024;
constexpr int MVFX_PAIR_NOT_TAKEN = 0x7fff0001; // mvfx_pair_submit: the caller launches this buffer itself, the plain way

// Memory references dropped AFTER the element's lock is released (declare it before the lock guard): the last reference frees the
// block, and freeing runs a stale mark's callback -- no callback of another element ever runs under this element's lock.
struct MvfxUnrefLater {
    GstMemory *m[4];
    int n = 0;
    void add(GstMemory *mem) { m[n++] = mem; }
    ~MvfxUnrefLater() { for (int i = 0; i < n; i++) gst_memory_unref(m[i]); }
};

// Launches n (1 or 2) frames of the element on `st`; called with the hold's lock held, so it may read what the element stored next
// to the held-back frame (its settings).
typedef int (*MvfxPairLaunch)(GstObject *element, const mvfx_frame *in, const mvfx_frame *out, uint32_t n, mvfx_stream st);

// MVFX_ELEMENT_PAIR: 0 = a launch per buffer, 1 (default) = hold back unless the element finds itself inside a chain (see above),
// 2 = always hold back (tests: the cross-thread flushes of neighbouring elements, all the time)
static inline int mvfx_pair_mode(void)
{
    static const int mode = g_getenv("MVFX_ELEMENT_PAIR") ? atoi(g_getenv("MVFX_ELEMENT_PAIR")) : 1;
    return mode;
}
static inline gboolean mvfx_pair_enabled(void) { return mvfx_pair_mode() != 0; }

// The held-back frame leaves alone (lock held).  A failure cannot be the flow return of its buffer any more: it is posted.
static inline void mvfx_pair_flush_locked(MvfxPairHold *h, GstObject *element, MvfxPairLaunch launch, MvfxUnrefLater *later)
{
    if (!h->out_mem) return;
    GstMemory *in = h->in_mem, *out = h->out_mem;
    h->in_mem = h->out_mem = NULL;
    // the marks stay until the fences are recorded (release_as_owner): a user on another thread runs into the hold's lock meanwhile
    mvfx_hip_memory_acquire_as_owner(in, h->st, element);
    mvfx_hip_memory_acquire_as_owner(out, h->st, element);
    // one fence for the launch, not one per block, and carried by the kernel itself (mvfx_hip_fence_begin / _end)
    GstMemory *const both[2] = {in, out};
    MvfxFenceScope fs;
    mvfx_hip_fence_begin(&fs, both, 2, h->st, FALSE);
    const int rc = launch(element, &h->fi, &h->fo, 1, h->st);
    mvfx_hip_fence_end(&fs, h->st, element, element);
    later->add(in);
    later->add(out);
    h->n_singles++;
    if (rc != MVFX_OK)
        GST_ELEMENT_ERROR(GST_ELEMENT(element), LIBRARY, FAILED, ("%s", mvfx_last_error()), ("held-back frame"));
}

static inline void mvfx_pair_flush(MvfxPairHold *h, GstObject *element, MvfxPairLaunch launch)
{
    MvfxUnrefLater later;
    std::lock_guard<std::mutex> g(h->lock);
    mvfx_pair_flush_locked(h, element, launch, &later);
}

// The flush registered on the blocks: somebody looked at a held-back frame
static inline void mvfx_pair_flush_foreign(MvfxPairHold *h, GstObject *element, MvfxPairLaunch launch)
{
    MvfxUnrefLater later;
    std::lock_guard<std::mutex> g(h->lock);
    if (h->out_mem && mvfx_pair_mode() == 1 && ++h->foreign_streak >= kMvfxPairStreak) {
        h->foreign_streak = 0;
        h->direct_left = kMvfxPairDirect;
    }
    mvfx_pair_flush_locked(h, element, launch, &later);
}

// One buffer of the element.  `compatible`: the held-back frame (if any) may share a launch with this one as far as the element's
// own state goes (same settings); geometry, formats and distinct blocks are checked here.  `store`: called under the lock when this
// frame becomes the held-back one (the element copies its settings next to it).  Returns the MVFX_* code of a launch, MVFX_OK when
// the frame was held back.
template <typename Store>
static inline int mvfx_pair_submit(MvfxPairHold *h, GstObject *element, MvfxPairLaunch launch, MvfxDeferredFlush flush_cb, GstBuffer *inbuf,
                                   GstBuffer *outbuf, const mvfx_frame &fi, const mvfx_frame &fo, mvfx_stream st, gboolean compatible, Store store)
{
    GstMemory *in = gst_buffer_peek_memory(inbuf, 0), *out = gst_buffer_peek_memory(outbuf, 0);
    // input still being written (an upstream kernel or the source's copy in flight, or held back): this element is a stage of a
    // dependent chain -- no holding back behind that (asked BEFORE the foreign flush below turns a held-back kernel into a fence)
    const gboolean chained = mvfx_pair_mode() == 1 && mvfx_hip_memory_busy(in, element);
    // what OTHER elements hold back on the two blocks (an upstream in-place filter's kernel on our input, a former user's on a
    // recycled output block) leaves now, before our lock is taken: under the lock no foreign flush ever runs (mvfxhipmemory.cpp)
    mvfx_hip_memory_flush_foreign(in, element);
    mvfx_hip_memory_flush_foreign(out, element);
    MvfxUnrefLater later;
    std::unique_lock<std::mutex> g(h->lock);
    h->n_buffers++;
    if (chained || h->direct_left) { // ... or every held-back frame was flushed by somebody's look lately: a plain launch per buffer for a while
        if (h->direct_left) h->direct_left--;
        h->n_direct++;
        mvfx_pair_flush_locked(h, element, launch, &later);
        return MVFX_PAIR_NOT_TAKEN;
    }
    const auto same_shape = [](const mvfx_frame &a, const mvfx_frame &b) {
        return a.width == b.width && a.height == b.height && a.stride == b.stride && a.format == b.format;
    };
    if (h->out_mem && (!compatible || !same_shape(h->fi, fi) || !same_shape(h->fo, fo) || h->in_mem == in || h->in_mem == out ||
                       h->out_mem == in || h->out_mem == out))
        mvfx_pair_flush_locked(h, element, launch, &later); // the held-back frame goes first, alone
    if (!h->out_mem) {
        h->in_mem = gst_memory_ref(in);
        h->out_mem = gst_memory_ref(out);
        h->fi = fi;
        h->fo = fo;
        h->st = st;
        store();
        g.unlock();
        // (set_deferred runs whatever somebody else holds back on the block first -- an upstream element's kernel on our input --
        // and must not be called under the lock: that flush may be ours on another block)
        mvfx_hip_memory_set_deferred(in, flush_cb, element);
        mvfx_hip_memory_set_deferred(out, flush_cb, element);
        return MVFX_OK;
    }
    GstMemory *first_in = h->in_mem, *first_out = h->out_mem;
    const mvfx_frame ins[2] = {h->fi, fi}, outs[2] = {h->fo, fo};
    h->in_mem = h->out_mem = NULL;
    static const int pair_stream_mode = g_getenv("MVFX_PAIR_STREAM") ? atoi(g_getenv("MVFX_PAIR_STREAM")) : 1;
    // consecutive PAIRS alternate between two streams of this thread (0: the second frame's own stream, experiment)
    const mvfx_stream pst = pair_stream_mode ? mvfx_thread_stream_n(h->pair_no++ & 1u) : st;
    mvfx_hip_memory_acquire_as_owner(first_in, pst, element);
    mvfx_hip_memory_acquire_as_owner(first_out, pst, element);
    mvfx_hip_memory_acquire_as_owner(in, pst, element);
    mvfx_hip_memory_acquire_as_owner(out, pst, element);
    // ONE fence for the four blocks, carried by the kernel itself as its stop event: an event record costs the device a bubble
    GstMemory *const all[4] = {first_in, first_out, in, out};
    MvfxFenceScope fs;
    mvfx_hip_fence_begin(&fs, all, 4, pst, FALSE);
    const int rc = launch(element, ins, outs, 2, pst);
    mvfx_hip_fence_end(&fs, pst, element, element);
    h->n_pairs++;
    h->foreign_streak = 0;
    later.add(first_in);
    later.add(first_out);
    return rc;
}

static inline void mvfx_pair_print_stats(MvfxPairHold *h, GstObject *element, const char *what)
{
    if (g_getenv("MVFX_ELEMENT_PAIR_STATS") && h->n_buffers)
        g_printerr("%s %s: %" G_GUINT64_FORMAT " device buffers = 2 x %" G_GUINT64_FORMAT " pair launches + %" G_GUINT64_FORMAT
                   " single launches + %" G_GUINT64_FORMAT " direct launches\n", what, GST_OBJECT_NAME(element), h->n_buffers, h->n_pairs,
                   h->n_singles, h->n_direct);
}
