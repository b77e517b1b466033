// GstAllocator for HIP device memory (see mvfxhipmemory.h).
#include "mvfxhipmemory.h"

#include "mi355vfx.h"

#include <string.h>

#include <initializer_list>
#include <vector>

// A fence = ONE recorded event, shared by every block the launch behind it touched (an out-of-place pair launch touches four).  An
// event record is not free on the device -- a barrier packet with a completion signal between two kernels of the stream: the
// hsvdetector element with one event per block ran at 71 k fps, with two records per pair launch instead of four at 89 k, with one at
// 95 k (profiles/r4/element_pairs.txt) -- so blocks released together share one.  Fences are reference-counted and pooled: an event is
// re-recorded only when no block points at it any more (a stream that already waits for it captured the earlier record).
// deps (round 6): fences of OTHER dispatches that a lane dispatch under this fence waits for on the device (a barrier packet across the lane's queues,
// mvfx_direct_queue_wait_event).  They stay referenced -- so that nobody arms their signals again under the barrier packet -- until this fence's own
// dispatch is known to have finished: when the fence is taken out of the pool again, or destroyed.
#define MVFX_FENCE_MAX_DEPS 8
typedef struct MvfxFence_ { gint refs; mvfx_event ev; struct MvfxFence_ *deps[MVFX_FENCE_MAX_DEPS]; int ndeps; } MvfxFence;
#define MVFX_FENCE_POOL_MAX 256
static GMutex fence_pool_lock;
static MvfxFence *fence_pool[MVFX_FENCE_POOL_MAX];
static int fence_pool_n;
static void fence_unref(MvfxFence *f);

// the dispatch under `f` (if any) has finished -> the fences it waited for on the device may go
static void fence_drop_deps(MvfxFence *f)
{
    if (f->ndeps == 0) return;
    mvfx_event_synchronize(f->ev); // (long fired in practice: the fence comes out of the pool a pool's worth of frames later)
    for (int i = 0; i < f->ndeps; i++) fence_unref(f->deps[i]);
    f->ndeps = 0;
}

static MvfxFence *fence_get(void) // one reference, event not recorded yet; NULL when no event can be created
{
    MvfxFence *f = NULL;
    g_mutex_lock(&fence_pool_lock);
    if (fence_pool_n > 0) f = fence_pool[--fence_pool_n];
    g_mutex_unlock(&fence_pool_lock);
    if (f) fence_drop_deps(f);
    if (!f) {
        mvfx_event ev = NULL;
        if (mvfx_event_create(&ev) != MVFX_OK || !ev) return NULL;
        f = g_new0(MvfxFence, 1);
        f->ev = ev;
    }
    f->refs = 1;
    return f;
}

static MvfxFence *fence_ref(MvfxFence *f)
{
    if (f) g_atomic_int_inc(&f->refs);
    return f;
}

static void fence_unref(MvfxFence *f)
{
    if (!f || !g_atomic_int_dec_and_test(&f->refs)) return;
    g_mutex_lock(&fence_pool_lock);
    const gboolean kept = fence_pool_n < MVFX_FENCE_POOL_MAX;
    if (kept) fence_pool[fence_pool_n++] = f;
    g_mutex_unlock(&fence_pool_lock);
    if (!kept) {
        fence_drop_deps(f);
        mvfx_event_destroy(f->ev);
        g_free(f);
    }
}

// MVFX_LANE_STATS=1: what the lane's acquires did, printed when the process ends (gst-launch runs)
static gint lane_stat[8]; // 0 taken, 1 refused: ordinary fence pending, 2 refused: held-back work / borrowed fence, 3 device-side waits armed,
                          // 4 host waits in the acquire, 5 host waits of a stream consumer for a direct fence, 6 relied on queue order, 7 times parked
static void lane_stats_print(void)
{
    g_printerr("mvfx lane: %d acquires taken (%d rested on queue order, %d device-side waits, %d host waits), refused: %d ordinary fence pending, "
               "%d held-back work or borrowed fence; parked %d times; %d host waits of stream consumers for direct fences\n", lane_stat[0], lane_stat[6],
               lane_stat[3], lane_stat[4], lane_stat[1], lane_stat[2], lane_stat[7], lane_stat[5]);
}
static inline void lane_count(int what)
{
    static const gboolean on = [] {
        const gchar *e = g_getenv("MVFX_LANE_STATS");
        const gboolean v = e && atoi(e) != 0;
        if (v) atexit(lane_stats_print);
        return v;
    }();
    if (on) g_atomic_int_inc(&lane_stat[what]);
}

// PARKING (csrc/direct_dispatch.h): when the lane's acquires are mostly refused -- the frames keep meeting stream fences: tee siblings, an element without
// lane kernels in the chain, a source that copies every frame -- the lane's two hardware queues only slow the streams down (hardware queues are few).
// Over windows of kLaneWindow acquires: three quarters refused -> the queues are destroyed (mvfx_direct_lane_park) and nobody asks for the lane for the
// next kLaneParkedFor acquires; then it is tried again.
static const gint kLaneWindow = 2048, kLaneParkedFor = 65536;
static gint lane_win_taken, lane_win_refused, lane_parked_left;

static inline gboolean lane_parked(void)
{
    for (gint v = g_atomic_int_get(&lane_parked_left); v > 0; v = g_atomic_int_get(&lane_parked_left))
        if (g_atomic_int_compare_and_exchange(&lane_parked_left, v, v - 1)) return TRUE;
    return FALSE;
}

static void lane_window(gboolean taken)
{
    const gint t = taken ? g_atomic_int_add(&lane_win_taken, 1) + 1 : g_atomic_int_get(&lane_win_taken);
    const gint r = taken ? g_atomic_int_get(&lane_win_refused) : g_atomic_int_add(&lane_win_refused, 1) + 1;
    if (t + r < kLaneWindow) return;
    g_atomic_int_set(&lane_win_taken, 0);
    g_atomic_int_set(&lane_win_refused, 0);
    if (r >= 3 * t && g_atomic_int_compare_and_exchange(&lane_parked_left, 0, kLaneParkedFor)) {
        lane_count(7);
        mvfx_direct_lane_park();
    }
}

static const gint kLaneSeed = 512;
static gint lane_seed_budget = kLaneSeed; // (see "SEEDING" in mvfx_hip_buffer_acquire_direct_ordered; refilled by mvfx_direct_reset)

// Dependencies the calling thread's next lane dispatch waits for on the device (stashed by mvfx_hip_buffer_acquire_direct*, moved onto the dispatch's
// fence by mvfx_hip_fence_begin): each entry holds a reference.
static thread_local MvfxFence *tls_lane_deps[MVFX_FENCE_MAX_DEPS];
static thread_local int tls_lane_ndeps = 0;

// the lane dispatch they were stashed for is not going to happen: wait for them on this thread (the barrier packets already in the queue then pass),
// and let them go
static void lane_deps_abandon(void)
{
    for (int i = 0; i < tls_lane_ndeps; i++) {
        mvfx_event_synchronize(tls_lane_deps[i]->ev);
        fence_unref(tls_lane_deps[i]);
    }
    tls_lane_ndeps = 0;
}

typedef struct {
    GstMemory mem;
    void *dptr;       // hipMalloc'ed
    int device;       // ordinal the block lives on
    guint8 *shadow;   // host copy while CPU-mapped
    GstMapFlags shadow_flags;
    gint cpu_maps;
    GMutex lock;
    // the fence of the buffer (d3d12colorlut/imp.rs:695-714 keeps an ID3D12Fence value on the output memory): recorded by
    // the last stream that enqueued work on the block; the next user makes its stream wait for it, a CPU map waits on the host.
    // Releases CHAIN: a stream that records while another stream's record is still pending first waits for that one (device
    // side), so that the single event always covers every user so far -- two readers of one buffer behind a `tee` on two
    // streaming threads must both have finished before a recycled block is written again.
    MvfxFence *fence;   // referenced; NULL: nothing was ever recorded on the block
    gboolean pending;
    // a fence recorded by somebody else on a stream of theirs (the launch combiner's batch event): not owned, never re-recorded
    // here; while set it is what the next user waits for, and the next release chains onto it like onto a pending own record
    mvfx_event borrowed;
    // work an element holds back on this block (mvfx_hip_memory_set_deferred): flushed by the next user before it looks at the fence
    MvfxDeferredFlush deferred_flush;
    GstObject *deferred_owner; // referenced while set
    const void *fence_owner;   // who recorded the pending fence (compared only, never dereferenced): mvfx_hip_memory_busy
    // which record the pending fence is and on which stream it was made; which record the last acquire saw, on which stream: a stream
    // does not wait for its own earlier work, and a release that finds the fence its acquire already waited for does not wait again.
    // (Three HIP calls per block and buffer -- wait, wait, record -- were most of the 14 us of host time a streaming thread spent per
    // buffer in an out-of-place element: profiles/r4/element_pairs.txt)
    guint64 fence_seq, acq_seq;
    mvfx_stream fence_stream, acq_stream;
} MvfxHipMemory;

typedef struct { GstAllocator parent; } MvfxHipAllocator;
typedef struct { GstAllocatorClass parent_class; } MvfxHipAllocatorClass;

G_DEFINE_TYPE(MvfxHipAllocator, mvfx_hip_allocator, GST_TYPE_ALLOCATOR)

// Small free list keyed by (device, size): video buffers of one stream all have the same size, so a freed
// block is handed to the next allocation instead of paying hipFree + hipMalloc per frame.  The block keeps its
// fence: work still in flight on it when the buffer was dropped orders before the next owner's first kernel.
#define MVFX_FREELIST_MAX 16
static GMutex freelist_lock;
static struct { void *dptr; gsize size; int device; MvfxFence *fence; gboolean pending; } freelist[MVFX_FREELIST_MAX];

static int current_device(void)
{
    return mvfx_current_device();
}

static gboolean freelist_take(gsize size, int device, void **dptr, MvfxFence **ev, gboolean *pending)
{
    gboolean found = FALSE;
    g_mutex_lock(&freelist_lock);
    for (int i = 0; i < MVFX_FREELIST_MAX && !found; i++)
        if (freelist[i].dptr && freelist[i].size == size && freelist[i].device == device) {
            *dptr = freelist[i].dptr;
            *ev = freelist[i].fence;
            *pending = freelist[i].pending;
            freelist[i].dptr = NULL;
            found = TRUE;
        }
    g_mutex_unlock(&freelist_lock);
    return found;
}

static void release_block(void *dptr, MvfxFence *fence, gboolean pending)
{
    if (pending && fence) mvfx_event_synchronize(fence->ev);
    mvfx_device_free(dptr);
    fence_unref(fence);
}

// Always keeps the block: when the list is full the OLDEST entry is evicted and freed (sizes of a previous negotiation would
// otherwise occupy the slots for the rest of the process and send every later free down the device-synchronising hipFree path).
static void freelist_give(void *dptr, gsize size, int device, MvfxFence *ev, gboolean pending)
{
    static guint next_victim = 0;
    void *old_dptr = NULL;
    MvfxFence *old_ev = NULL;
    gboolean old_pending = FALSE;
    g_mutex_lock(&freelist_lock);
    int slot = -1;
    for (int i = 0; i < MVFX_FREELIST_MAX && slot < 0; i++)
        if (!freelist[i].dptr) slot = i;
    if (slot < 0) {
        slot = (int)(next_victim++ % MVFX_FREELIST_MAX);
        old_dptr = freelist[slot].dptr; old_ev = freelist[slot].fence; old_pending = freelist[slot].pending;
    }
    freelist[slot].dptr = dptr; freelist[slot].size = size; freelist[slot].device = device;
    freelist[slot].fence = ev; freelist[slot].pending = pending;
    g_mutex_unlock(&freelist_lock);
    if (old_dptr) release_block(old_dptr, old_ev, old_pending);
}

// Gives cached blocks back to the device: those of `size` bytes (0: every size) on the calling thread's device.  The victims are
// collected under the lock and released outside it (release_block waits for the block's fence and hipFree synchronises the device:
// another pipeline's alloc / free must not queue behind that on freelist_lock).
static void freelist_trim(gsize size)
{
    struct { void *dptr; MvfxFence *ev; gboolean pending; } victims[MVFX_FREELIST_MAX];
    int n = 0;
    const int device = current_device();
    g_mutex_lock(&freelist_lock);
    for (int i = 0; i < MVFX_FREELIST_MAX; i++)
        if (freelist[i].dptr && (size == 0 || (freelist[i].size == size && freelist[i].device == device))) {
            victims[n].dptr = freelist[i].dptr; victims[n].ev = freelist[i].fence; victims[n].pending = freelist[i].pending;
            n++;
            freelist[i].dptr = NULL;
        }
    g_mutex_unlock(&freelist_lock);
    for (int i = 0; i < n; i++) release_block(victims[i].dptr, victims[i].ev, victims[i].pending);
}

void mvfx_hip_allocator_trim(void)
{
    freelist_trim(0);
}

static GstMemory *mvfx_hip_alloc(GstAllocator *allocator, gsize size, GstAllocationParams *params)
{
    void *dptr = NULL;
    MvfxFence *ev = NULL;
    gboolean pending = FALSE;
    const int device = current_device();
    if (!freelist_take(size, device, &dptr, &ev, &pending)) {
        if (mvfx_device_alloc(&dptr, size) != MVFX_OK) {
            GST_ERROR("HIP allocation of %" G_GSIZE_FORMAT " bytes failed: %s", size, mvfx_last_error());
            return NULL;
        }
    }
    MvfxHipMemory *m = g_new0(MvfxHipMemory, 1);
    // NO_SHARE: there is no mem_share (sub-range views of device blocks are not needed by these elements);
    // gst_buffer_copy_region / gst_buffer_resize then copy instead of calling a NULL vfunc
    gst_memory_init(GST_MEMORY_CAST(m), GST_MEMORY_FLAG_NO_SHARE, allocator, NULL, size, 255, 0, size);
    m->dptr = dptr;
    m->device = device;
    m->fence = ev;
    m->pending = pending;
    m->fence_stream = m->acq_stream = (mvfx_stream)(gintptr)-1; // a fence that came with a recycled block: made on no stream anybody has
    g_mutex_init(&m->lock);
    return GST_MEMORY_CAST(m);
}

static void run_deferred(MvfxHipMemory *m);

static void mvfx_hip_free(GstAllocator *, GstMemory *mem)
{
    MvfxHipMemory *m = (MvfxHipMemory *)mem;
    // a mark can only be stale here (an owner with work held back on the block keeps a reference on it): this drops the mark's
    // reference on its owner
    run_deferred(m);
    if (m->borrowed) { // the free list carries owned events only: finish the borrowed one here (rare: a buffer dropped right behind the combiner)
        mvfx_event_synchronize(m->borrowed);
        m->borrowed = NULL;
    }
    freelist_give(m->dptr, mem->maxsize, m->device, m->fence, m->pending);
    g_free(m->shadow);
    g_mutex_clear(&m->lock);
    g_free(m);
}

// Runs the block's deferred work, if any: the owner is referenced across the call, the memory's lock is NOT held (the flush releases
// the block, which takes it).  The flush clears the mark (mvfx_hip_memory_clear_deferred); a flush that forgets to is unmarked here.
static void run_deferred(MvfxHipMemory *m)
{
    g_mutex_lock(&m->lock);
    MvfxDeferredFlush fn = m->deferred_flush;
    GstObject *owner = fn ? GST_OBJECT(gst_object_ref(m->deferred_owner)) : NULL;
    g_mutex_unlock(&m->lock);
    if (!fn) return;
    fn(owner);
    mvfx_hip_memory_clear_deferred(GST_MEMORY_CAST(m), owner);
    gst_object_unref(owner);
}

void mvfx_hip_memory_set_deferred(GstMemory *mem, MvfxDeferredFlush flush, GstObject *owner)
{
    if (!mvfx_is_hip_memory(mem) || !flush || !owner) return;
    MvfxHipMemory *m = (MvfxHipMemory *)mem;
    // Somebody else's held-back work on this block comes first -- and a foreign mark is NEVER overwritten: two holding readers of one
    // input block (tee ! queue ! hsvdetector on one branch, tee ! queue ! colorlut on the other) could both find the block unmarked
    // and both store, the second store dropping the first owner's mark (and leaking its reference), so that the source's refill of
    // the recycled block flushed only one of the two held-back kernels (advisor r4).  Check and set under one critical section; a
    // foreign mark found there is run with the lock released, then the check repeats.
    for (;;) {
        g_mutex_lock(&m->lock);
        if (!m->deferred_flush) {
            m->deferred_flush = flush;
            m->deferred_owner = GST_OBJECT(gst_object_ref(owner));
            g_mutex_unlock(&m->lock);
            return;
        }
        if (m->deferred_owner == owner) { // ours already (the hold re-marks a block it still holds)
            m->deferred_flush = flush;
            g_mutex_unlock(&m->lock);
            return;
        }
        g_mutex_unlock(&m->lock);
        run_deferred(m);
    }
}

void mvfx_hip_memory_clear_deferred(GstMemory *mem, GstObject *owner)
{
    if (!mvfx_is_hip_memory(mem)) return;
    MvfxHipMemory *m = (MvfxHipMemory *)mem;
    GstObject *drop = NULL;
    g_mutex_lock(&m->lock);
    if (m->deferred_flush && m->deferred_owner == owner) {
        drop = m->deferred_owner;
        m->deferred_flush = NULL;
        m->deferred_owner = NULL;
    }
    g_mutex_unlock(&m->lock);
    if (drop) gst_object_unref(drop);
}

static gboolean deferred_is_owners(MvfxHipMemory *m, GstObject *owner)
{
    if (!owner) return FALSE;
    g_mutex_lock(&m->lock);
    const gboolean mine = m->deferred_flush != NULL && m->deferred_owner == owner;
    g_mutex_unlock(&m->lock);
    return mine;
}

// ---- producers whose direct fences a consumer had to wait for on its own thread (mvfxhipmemory.h) ----
// With hysteresis since round 6's chain soak: a producer goes back to its streams only when consumers had to wait kDiscourageAfter times (the first
// frames of a pipeline always produce a wait or two -- the source's blocks carry stream fences, an element takes the stream once, the next one waits for
// the one before it -- and ONE such wait used to switch the lane off for the rest of the run: thread-separated chains then ran at 10 k fps in a run
// and 19 k in the next), and it tries the lane again after kDiscourageFrames frames.
#define MVFX_DISCOURAGED_MAX 64
static const guint kDiscourageAfter = 8, kDiscourageFrames = 2048;
static GMutex discouraged_lock;
static struct { const void *tag; guint waits, skipped; } discouraged[MVFX_DISCOURAGED_MAX];
static gint discouraged_n; // read without the lock on the fast path

gboolean mvfx_direct_discouraged(const void *tag)
{
    if (!tag || g_atomic_int_get(&discouraged_n) == 0) return FALSE;
    gboolean hit = FALSE;
    g_mutex_lock(&discouraged_lock);
    for (gint i = 0; i < discouraged_n; i++)
        if (discouraged[i].tag == tag) {
            hit = discouraged[i].waits >= kDiscourageAfter;
            if (hit && ++discouraged[i].skipped >= kDiscourageFrames) discouraged[i].waits = discouraged[i].skipped = 0; // the next frame tries again
            break;
        }
    g_mutex_unlock(&discouraged_lock);
    return hit;
}

static void direct_discourage(const void *tag)
{
    if (!tag) return;
    g_mutex_lock(&discouraged_lock);
    gint at = -1;
    for (gint i = 0; i < discouraged_n && at < 0; i++)
        if (discouraged[i].tag == tag) at = i;
    if (at < 0 && discouraged_n < MVFX_DISCOURAGED_MAX) {
        at = discouraged_n;
        discouraged[at].tag = tag;
        discouraged[at].waits = discouraged[at].skipped = 0;
        g_atomic_int_set(&discouraged_n, discouraged_n + 1);
    }
    if (at >= 0 && discouraged[at].waits < kDiscourageAfter) discouraged[at].waits++;
    g_mutex_unlock(&discouraged_lock);
}

void mvfx_direct_reset(const void *tag)
{
    g_atomic_int_set(&lane_seed_budget, kLaneSeed);
    g_atomic_int_set(&lane_parked_left, 0); // (a new run finds out for itself)
    g_atomic_int_set(&lane_win_taken, 0);
    g_atomic_int_set(&lane_win_refused, 0);
    g_mutex_lock(&discouraged_lock);
    for (gint i = 0; i < discouraged_n; i++)
        if (discouraged[i].tag == tag) {
            discouraged[i] = discouraged[discouraged_n - 1];
            g_atomic_int_set(&discouraged_n, discouraged_n - 1);
            break;
        }
    g_mutex_unlock(&discouraged_lock);
}

// a consumer's stream "waits" for a block's fence: a direct fence that has not fired makes the calling thread wait -- its producer is told
static void wait_for_fence(mvfx_stream stream, mvfx_event ev, const void *producer_tag)
{
    if (mvfx_event_is_direct(ev) && mvfx_event_query(ev) != 1) {
        lane_count(5);
        if (producer_tag) direct_discourage(producer_tag);
    }
    mvfx_stream_wait_event(stream, ev);
}

gboolean mvfx_hip_buffer_acquire_direct(GstBuffer *buf, mvfx_stream stream, int queue) { return mvfx_hip_buffer_acquire_direct_ordered(buf, stream, queue, NULL); }

gboolean mvfx_hip_buffer_acquire_direct_ordered(GstBuffer *buf, mvfx_stream stream, int queue, gboolean *relied_on_order)
{
    if (lane_parked()) {
        lane_deps_abandon();
        return FALSE;
    }
    for (guint i = 0; buf && i < gst_buffer_n_memory(buf); i++) {
        GstMemory *mem = gst_buffer_peek_memory(buf, i);
        if (!mvfx_is_hip_memory(mem)) { lane_deps_abandon(); return FALSE; }
        MvfxHipMemory *m = (MvfxHipMemory *)mem;
        MvfxFence *other = NULL;
        gboolean seeding = FALSE;
        g_mutex_lock(&m->lock);
        gboolean ok = m->deferred_flush == NULL && (!m->borrowed || mvfx_event_query(m->borrowed) == 1);
        if (!ok) lane_count(2);
        if (ok && m->pending && m->fence) {
            if (mvfx_event_query(m->fence->ev) == 1)
                m->pending = FALSE; // seen finished: nobody has to wait for it any more
            else if (!mvfx_event_is_direct(m->fence->ev)) {
                if (g_atomic_int_get(&lane_seed_budget) > 0 && g_atomic_int_add(&lane_seed_budget, -1) > 0) {
                    other = fence_ref(m->fence); // seeding (below): waited for on this thread, once
                    seeding = TRUE;
                } else {
                    ok = FALSE;     // an ordinary fence still pending: this frame's place is behind it on a stream
                    lane_count(1);
                    lane_window(FALSE);
                }
            } else if (mvfx_event_direct_queue(m->fence->ev) != queue)
                other = fence_ref(m->fence); // a direct dispatch on the lane's OTHER queue: waited for below
            else {
                lane_count(6);
                if (relied_on_order) *relied_on_order = TRUE; // a direct dispatch in front of ours on the same lane queue: fine as long as our packet keeps its place
                // (without the out parameter: the caller's packets always carry the barrier bit -- the queue is in order, nothing to do)
            }
        }
        g_mutex_unlock(&m->lock);
        if (other) {
            // A dispatch on the lane's OTHER queue (this element's own on the block a pool's worth of frames ago -- an odd pool --, or another
            // element's on another streaming thread, whose blocks come back in no particular order): waited for ON THE DEVICE, by a barrier packet in
            // our queue (mvfx_direct_queue_wait_event), as a HIP stream waits for an event.  Until round 6's chain soak the streaming thread waited
            // here itself: three lane elements with queues between them ran at 10-12 k fps where HIP streams gave 17-22 k
            // (profiles/r6/lane_chain_soak.txt).  The fence stays referenced until our own dispatch has finished (MvfxFence::deps).
            // SEEDING.  Whether a block is worked on by lane dispatches or by stream kernels perpetuates itself: an element that finds a stream fence
            // pending on a block takes the stream, so the next element finds a stream fence on it ... and a thread-separated chain on device-born
            // frames settled wherever its first frames had put each block -- five of twelve blocks on streams for good, 14 k fps where the all-lane
            // state gives 19 k.  The first kLaneSeed stream fences a pipeline meets are therefore waited for on the streaming thread (they are the
            // source's fills and the first frames' kernels: microseconds), which starts every block on the lane.  Frames whose producers stay on
            // streams (hipupload's copies, an element without lane kernels) use the budget up in their first seconds and are refused as ever.
            const int armed = seeding ? -1 : tls_lane_ndeps < MVFX_FENCE_MAX_DEPS ? mvfx_direct_queue_wait_event(queue, other->ev) : -1;
            lane_count(armed == 1 ? 3 : armed < 0 ? 4 : 0 /* fired meanwhile */);
            if (armed < 0) mvfx_event_synchronize(other->ev); // (no lane, no room: the old way)
            g_mutex_lock(&m->lock);
            ok = m->fence == other && m->deferred_flush == NULL && !m->borrowed; // (nobody else came in between)
            if (ok && armed <= 0) m->pending = FALSE;                             // seen finished
            g_mutex_unlock(&m->lock);
            if (armed == 1) {
                tls_lane_deps[tls_lane_ndeps++] = other; // (keeps the reference)
                if (relied_on_order) *relied_on_order = TRUE;
            } else {
                fence_unref(other);
            }
        }
        if (!ok) {
            lane_deps_abandon();
            return FALSE;
        }
        lane_count(0);
        lane_window(TRUE);
        g_mutex_lock(&m->lock);
        m->acq_seq = m->fence_seq; // (the fence scope's chaining then has nothing to wait for on `stream`)
        m->acq_stream = stream;
        g_mutex_unlock(&m->lock);
    }
    return buf != NULL;
}

void mvfx_hip_fence_cancel(MvfxFenceScope *sc)
{
    lane_deps_abandon(); // (a scope with no device blocks took none of them)
    if (sc->n == 0) return;
    mvfx_thread_clear_completion_event();
    if (MvfxFence *f = (MvfxFence *)sc->fence) {
        // the dispatch the barrier packets were for is not coming: wait for the dependencies here, the packets then pass and the fences may go
        for (int i = 0; i < f->ndeps; i++) {
            mvfx_event_synchronize(f->deps[i]->ev);
            fence_unref(f->deps[i]);
        }
        f->ndeps = 0;
    }
    fence_unref((MvfxFence *)sc->fence);
    sc->fence = NULL;
    sc->n = 0;
}

void mvfx_hip_memory_acquire(GstMemory *mem, mvfx_stream stream) { mvfx_hip_memory_acquire_as_owner(mem, stream, NULL); }

// Somebody else's held-back work on the block, run now.  An element that launches under its own lock (mvfx_pair_hold.h, hsvfilter)
// calls this BEFORE taking that lock: a foreign flush takes the other element's lock, and two elements that wait for each other's
// flush under their own locks never come back (hsvfilter ! queue ! hsvdetector on recycled blocks did exactly that).
void mvfx_hip_memory_flush_foreign(GstMemory *mem, GstObject *owner)
{
    if (!mvfx_is_hip_memory(mem)) return;
    MvfxHipMemory *m = (MvfxHipMemory *)mem;
    if (!deferred_is_owners(m, owner)) run_deferred(m);
}

// owner != NULL: the caller holds its own lock and has called mvfx_hip_memory_flush_foreign() on the block before taking it: NO
// callback runs here.  (A mark that appears on the block after that call can only be a sibling READER's behind a tee -- a writer
// cannot get hold of a block this element owns or reads -- and a reader's held-back work does not have to come first.)
void mvfx_hip_memory_acquire_as_owner(GstMemory *mem, mvfx_stream stream, GstObject *owner)
{
    if (!mvfx_is_hip_memory(mem)) return;
    MvfxHipMemory *m = (MvfxHipMemory *)mem;
    if (!owner) run_deferred(m);
    g_mutex_lock(&m->lock);
    if (m->borrowed)
        mvfx_stream_wait_event(stream, m->borrowed);
    if (m->pending && m->fence && m->fence_stream != stream)
        wait_for_fence(stream, m->fence->ev, m->fence_owner); // device-side wait; the host goes on (same stream: in order anyway)
    m->acq_seq = m->fence_seq;
    m->acq_stream = stream;
    g_mutex_unlock(&m->lock);
}

// Is device work still running on the block (its fence recorded and not yet reached), or held back on it by somebody else?  An
// element asks this of its INPUT before it holds its own kernel back: behind work that is still in flight the element is one stage
// of a dependent chain, and holding back there only lengthens the chain (profiles/r4/element_pairs.txt).  Never blocks, runs no
// callback.
gboolean mvfx_hip_memory_busy(GstMemory *mem, GstObject *owner)
{
    if (!mvfx_is_hip_memory(mem)) return FALSE;
    MvfxHipMemory *m = (MvfxHipMemory *)mem;
    g_mutex_lock(&m->lock);
    gboolean busy = m->deferred_flush != NULL && m->deferred_owner != owner;
    if (!busy && m->borrowed) busy = mvfx_event_query(m->borrowed) != 1;
    // (the asking element's OWN last kernel on a recycled block does not count: that is the launch rate it is trying to raise)
    if (!busy && m->pending && m->fence && m->fence_owner != (const void *)owner) busy = mvfx_event_query(m->fence->ev) != 1;
    g_mutex_unlock(&m->lock);
    return busy;
}

void *mvfx_hip_memory_pending_fence(GstMemory *mem)
{
    if (!mvfx_is_hip_memory(mem)) return NULL;
    MvfxHipMemory *m = (MvfxHipMemory *)mem;
    run_deferred(m);
    g_mutex_lock(&m->lock);
    // the borrowed fence is always younger than the own record it was set behind (the combiner's launch waited for that one)
    void *ev = m->borrowed ? m->borrowed : (m->pending && m->fence ? m->fence->ev : NULL);
    g_mutex_unlock(&m->lock);
    return ev;
}

void mvfx_hip_memory_set_borrowed_fence(GstMemory *mem, void *event)
{
    if (!mvfx_is_hip_memory(mem)) return;
    MvfxHipMemory *m = (MvfxHipMemory *)mem;
    g_mutex_lock(&m->lock);
    m->borrowed = event;
    g_mutex_unlock(&m->lock);
}

// ---- release: the stream's work on one or several blocks gets ONE fence ---------------------------------------------------------
//
// Phase A, per block under its lock: chain onto what other users left since our acquire (a second reader behind a tee; a borrowed
// fence): the stream waits for it on the device, so the record covers them too.  Then the one record.  Phase B, per block under its
// lock again: the block points at the fence, its held-back mark (if it is `owner`'s) goes away in the same critical section -- a
// consumer on another thread either still sees the mark (its flush then waits for the owner's lock and finds the work launched) or
// already sees the fence, never neither.  Somebody who recorded on a block between the two phases (the tee sibling again) is chained
// onto there, and that block gets a fence of its own.
static void group_phase_a(MvfxFenceScope *sc, GstMemory *const *mems, guint n, mvfx_stream stream)
{
    sc->n = 0;
    sc->fence = NULL;
    for (guint i = 0; i < n && sc->n < 8; i++)
        if (mvfx_is_hip_memory(mems[i])) sc->mems[sc->n++] = mems[i];
    for (guint i = 0; i < sc->n; i++) {
        MvfxHipMemory *m = (MvfxHipMemory *)sc->mems[i];
        g_mutex_lock(&m->lock);
        if (m->fence && m->pending && m->fence_stream != stream && !(m->acq_seq == m->fence_seq && m->acq_stream == stream))
            wait_for_fence(stream, m->fence->ev, m->fence_owner);
        if (m->borrowed) { // same chaining for a fence somebody else recorded; the record then covers it
            mvfx_stream_wait_event(stream, m->borrowed);
            m->borrowed = NULL;
        }
        sc->seen[i] = m->fence_seq;
        g_mutex_unlock(&m->lock);
    }
}

static void group_phase_b(MvfxFenceScope *sc, mvfx_stream stream, GstObject *owner, const void *tag, MvfxFence *f)
{
    // a DIRECT fence (the launch went out on the library's own queue): no stream of anybody's is ordered behind the work it stands for
    const gboolean direct = f && mvfx_event_is_direct(f->ev);
    if (!f) mvfx_stream_synchronize(stream); // no event: fall back to a blocking hand-off
    for (guint i = 0; i < sc->n; i++) {
        MvfxHipMemory *m = (MvfxHipMemory *)sc->mems[i];
        MvfxFence *old = NULL, *mine = f ? fence_ref(f) : NULL;
        GstObject *drop = NULL;
        g_mutex_lock(&m->lock);
        if (m->fence_seq != sc->seen[i] && m->fence && m->pending) {
            // recorded on in between: wait for that too, and a later fence of this block's own covers both
            mvfx_stream_wait_event(stream, m->fence->ev);
            if (direct) mvfx_stream_wait_event(stream, f->ev); // (the new record must cover the lane's kernel as well: the calling thread waits for it)
            fence_unref(mine);
            mine = fence_get();
            if (mine && mvfx_event_record(mine->ev, stream) != MVFX_OK) {
                fence_unref(mine);
                mine = NULL;
            }
            if (!mine) mvfx_stream_synchronize(stream);
        }
        old = m->fence;
        m->fence = mine;
        m->pending = mine != NULL;
        m->fence_seq++;
        m->fence_stream = direct && mine == f ? (mvfx_stream)(gintptr)-1 : stream;
        m->fence_owner = tag;
        if (owner && m->deferred_flush && m->deferred_owner == owner) {
            drop = m->deferred_owner;
            m->deferred_flush = NULL;
            m->deferred_owner = NULL;
        }
        g_mutex_unlock(&m->lock);
        fence_unref(old);
        if (drop) gst_object_unref(drop);
    }
    fence_unref(f);
    sc->n = 0;
}

static void release_group(GstMemory *const *mems, guint n, mvfx_stream stream, GstObject *owner, const void *tag)
{
    MvfxFenceScope sc;
    group_phase_a(&sc, mems, n, stream);
    if (sc.n == 0) return;
    MvfxFence *f = fence_get();
    if (f && mvfx_event_record(f->ev, stream) != MVFX_OK) {
        fence_unref(f);
        f = NULL;
    }
    group_phase_b(&sc, stream, owner, tag, f);
}

// The same release split around the launch, so that the fence costs the device nothing: _begin does the chaining waits and sets the
// fence's event as the calling thread's completion event (include/mi355vfx.h: the kernels of the next library call carry it as
// their stop event), the caller launches, _end publishes the fence on the blocks -- or records it the ordinary way when no kernel
// took it.  plain: the caller is an ordinary user of the blocks (somebody's held-back work on them comes first); otherwise it holds
// its own lock and has flushed foreign work before (mvfx_hip_memory_flush_foreign).
void mvfx_hip_fence_begin(MvfxFenceScope *sc, GstMemory *const *mems, guint n, mvfx_stream stream, gboolean plain)
{
    if (plain)
        for (guint i = 0; i < n; i++)
            if (mvfx_is_hip_memory(mems[i])) run_deferred((MvfxHipMemory *)mems[i]);
    group_phase_a(sc, mems, n, stream);
    if (sc->n == 0) { lane_deps_abandon(); return; }
    MvfxFence *f = fence_get();
    sc->fence = f;
    if (f) {
        for (int i = 0; i < tls_lane_ndeps; i++) f->deps[f->ndeps++] = tls_lane_deps[i]; // (fence_get left ndeps == 0; the references move)
        tls_lane_ndeps = 0;
        mvfx_thread_set_completion_event(f->ev);
    } else {
        lane_deps_abandon();
    }
}

// the blocks of one or two buffers as an ordinary user (input and output of an out-of-place launch)
void mvfx_hip_fence_begin_buffers(MvfxFenceScope *sc, GstBuffer *a, GstBuffer *b, mvfx_stream stream)
{
    GstMemory *mems[8];
    guint n = 0;
    for (GstBuffer *buf : {a, b})
        for (guint i = 0; buf && i < gst_buffer_n_memory(buf) && n < 8; i++) mems[n++] = gst_buffer_peek_memory(buf, i);
    mvfx_hip_fence_begin(sc, mems, n, stream, TRUE);
}

void mvfx_hip_fence_end(MvfxFenceScope *sc, mvfx_stream stream, GstObject *owner, GstObject *tag)
{
    if (sc->n == 0) return;
    MvfxFence *f = (MvfxFence *)sc->fence;
    // CONTRACT of every C-ABI entry called between _begin and _end: its LAST device operation is an MVFX_LAUNCH kernel on `stream`
    // (the kernel carries the fence's event as the stop event of its dispatch packet).  An entry that ends with a hipMemcpyAsync /
    // hipMemsetAsync, a launch on a partner stream or a raw hipLaunchKernelGGL would publish a premature fence (advisor r4).  Nothing in
    // the type system enforces that, so MVFX_FENCE_CHECK=1 (tests/test_gst_pipelines_gpu.py runs the device chains under it) checks it
    // at run time: once the stream has drained, a carried fence must have fired.
    const int carried = f ? mvfx_thread_clear_completion_event() : 0;
    if (f && carried <= 0 && mvfx_event_record(f->ev, stream) != MVFX_OK) {
        fence_unref(f);
        f = NULL;
    }
    static const gboolean check = g_getenv("MVFX_FENCE_CHECK") != NULL && atoi(g_getenv("MVFX_FENCE_CHECK")) != 0;
    if (check && f && carried > 0 && !mvfx_event_is_direct(f->ev)) { // (a direct fence is not on the stream: its draining says nothing)
        // the stream drained, so the fence must have fired (the other direction -- a fence that fires BEFORE a trailing copy of the call
        // has finished -- cannot be seen from here; the contract above is kept by review: every entry the elements call under a fence
        // scope ends in MVFX_LAUNCH; `grep -n hipLaunchKernelGGL csrc/` finds the macro itself and colorlut's content probe, which is launched
        // BEFORE the frame's kernel and touches no output)
        mvfx_stream_synchronize(stream);
        if (mvfx_event_query(f->ev) != 1)
            g_critical("MVFX_FENCE_CHECK: the fence carried by the last library call has not fired although its stream has drained: "
                       "no kernel of the call took the completion event");
    }
    group_phase_b(sc, stream, owner, tag, f);
}

void mvfx_hip_memory_release(GstMemory *mem, mvfx_stream stream) { mvfx_hip_memory_release_as_owner(mem, stream, NULL); }

// `owner` != NULL: the owner of the block's held-back work has just enqueued it on `stream` (no callback runs: see
// mvfx_hip_memory_acquire_as_owner); NULL: a plain user -- somebody's held-back work on the block still comes first.
void mvfx_hip_memory_release_as_owner(GstMemory *mem, mvfx_stream stream, GstObject *owner)
{
    if (!mvfx_is_hip_memory(mem)) return;
    if (!owner) run_deferred((MvfxHipMemory *)mem);
    release_group(&mem, 1, stream, owner, owner);
}

// Several blocks one launch touched (the two frames of a pair launch, input and output of an out-of-place element): one fence.
void mvfx_hip_memories_release_as_owner(GstMemory *const *mems, guint n, mvfx_stream stream, GstObject *owner)
{
    if (!owner)
        for (guint i = 0; i < n; i++)
            if (mvfx_is_hip_memory(mems[i])) run_deferred((MvfxHipMemory *)mems[i]);
    release_group(mems, n, stream, owner, owner);
}

// Plain releases that also say who recorded the fence (an element's launch per buffer: mvfx_hip_memory_busy must not take the element's
// own previous kernel on a recycled block for somebody else's work)
void mvfx_hip_memory_release_tagged(GstMemory *mem, mvfx_stream stream, GstObject *tag)
{
    mvfx_hip_memories_release_tagged(&mem, 1, stream, tag);
}

void mvfx_hip_memories_release_tagged(GstMemory *const *mems, guint n, mvfx_stream stream, GstObject *tag)
{
    for (guint i = 0; i < n; i++)
        if (mvfx_is_hip_memory(mems[i])) run_deferred((MvfxHipMemory *)mems[i]);
    release_group(mems, n, stream, NULL, tag);
}

void mvfx_hip_memory_wait(GstMemory *mem)
{
    if (!mvfx_is_hip_memory(mem)) return;
    MvfxHipMemory *m = (MvfxHipMemory *)mem;
    run_deferred(m);
    g_mutex_lock(&m->lock);
    if (m->borrowed) {
        mvfx_event_synchronize(m->borrowed);
        m->borrowed = NULL;
    }
    if (m->pending && m->fence) {
        mvfx_event_synchronize(m->fence->ev);
        m->pending = FALSE;
    }
    g_mutex_unlock(&m->lock);
}

void mvfx_hip_buffer_acquire(GstBuffer *buf, mvfx_stream stream)
{
    for (guint i = 0; buf && i < gst_buffer_n_memory(buf); i++)
        mvfx_hip_memory_acquire(gst_buffer_peek_memory(buf, i), stream);
}

void mvfx_hip_buffer_release(GstBuffer *buf, mvfx_stream stream) { mvfx_hip_buffers_release(buf, NULL, stream); }

// input and output buffer of one launch: one fence for all their blocks
void mvfx_hip_buffers_release(GstBuffer *a, GstBuffer *b, mvfx_stream stream)
{
    GstMemory *mems[8];
    guint n = 0;
    for (GstBuffer *buf : {a, b})
        for (guint i = 0; buf && i < gst_buffer_n_memory(buf); i++) {
            if (n == 8) { // (never with video buffers: flush what there is and go on)
                mvfx_hip_memories_release_as_owner(mems, n, stream, NULL);
                n = 0;
            }
            mems[n++] = gst_buffer_peek_memory(buf, i);
        }
    if (n) mvfx_hip_memories_release_as_owner(mems, n, stream, NULL);
}

static gpointer mvfx_hip_map_full(GstMemory *mem, GstMapInfo *info, gsize maxsize)
{
    MvfxHipMemory *m = (MvfxHipMemory *)mem;
    if (info->flags & MVFX_MAP_HIP)
        return m->dptr; // device pointer, zero copy; the caller orders its work with mvfx_hip_memory_acquire / _release
    mvfx_hip_memory_wait(mem); // a CPU reader / writer needs the enqueued work finished
    g_mutex_lock(&m->lock);
    if (m->cpu_maps == 0) {
        m->shadow = (guint8 *)g_malloc(mem->maxsize);
        m->shadow_flags = (GstMapFlags)0;
        // always fetch: a partial WRITE map must not lose the bytes it does not touch
        if (mvfx_copy_to_host(m->shadow, m->dptr, mem->maxsize, NULL) != MVFX_OK) {
            g_free(m->shadow);
            m->shadow = NULL;
            g_mutex_unlock(&m->lock);
            return NULL;
        }
    }
    m->cpu_maps++;
    m->shadow_flags = (GstMapFlags)(m->shadow_flags | info->flags);
    g_mutex_unlock(&m->lock);
    return m->shadow;
}

static void mvfx_hip_unmap_full(GstMemory *mem, GstMapInfo *info)
{
    MvfxHipMemory *m = (MvfxHipMemory *)mem;
    if (info->flags & MVFX_MAP_HIP)
        return;
    g_mutex_lock(&m->lock);
    if (--m->cpu_maps == 0) {
        if (m->shadow_flags & GST_MAP_WRITE)
            mvfx_copy_to_device(m->dptr, m->shadow, mem->maxsize, NULL);
        g_free(m->shadow);
        m->shadow = NULL;
    }
    g_mutex_unlock(&m->lock);
}

static GstMemory *mvfx_hip_copy(GstMemory *mem, gssize offset, gssize size)
{
    MvfxHipMemory *m = (MvfxHipMemory *)mem;
    if (size == -1)
        size = mem->size > (gsize)offset ? mem->size - offset : 0;
    mvfx_hip_memory_wait(mem);
    GstMemory *copy = mvfx_hip_alloc(mem->allocator, size, NULL);
    if (!copy)
        return NULL;
    mvfx_hip_memory_wait(copy); // a recycled block may still have a reader in flight
    if (mvfx_copy_device_to_device(((MvfxHipMemory *)copy)->dptr, (guint8 *)m->dptr + mem->offset + offset, size, NULL) != MVFX_OK) {
        gst_memory_unref(copy);
        return NULL;
    }
    return copy;
}

static void mvfx_hip_allocator_class_init(MvfxHipAllocatorClass *klass)
{
    GstAllocatorClass *ac = GST_ALLOCATOR_CLASS(klass);
    ac->alloc = mvfx_hip_alloc;
    ac->free = mvfx_hip_free;
}

static void mvfx_hip_allocator_init(MvfxHipAllocator *self)
{
    GstAllocator *a = GST_ALLOCATOR_CAST(self);
    a->mem_type = MVFX_HIP_MEMORY_TYPE;
    a->mem_map_full = mvfx_hip_map_full;
    a->mem_unmap_full = mvfx_hip_unmap_full;
    a->mem_copy = mvfx_hip_copy;
    // no mem_share: memories are created GST_MEMORY_FLAG_NO_SHARE
    GST_OBJECT_FLAG_SET(a, GST_ALLOCATOR_FLAG_CUSTOM_ALLOC);
}

GstAllocator *mvfx_hip_allocator_get(void)
{
    static gsize once = 0;
    static GstAllocator *singleton = NULL;
    if (g_once_init_enter(&once)) {
        singleton = (GstAllocator *)g_object_new(mvfx_hip_allocator_get_type(), NULL);
        gst_object_ref_sink(singleton);
        // one allocator for the life of the process, like GStreamer's own system-memory allocator: tell the leaks tracer
        GST_OBJECT_FLAG_SET(singleton, GST_OBJECT_FLAG_MAY_BE_LEAKED);
        g_once_init_leave(&once, 1);
    }
    return (GstAllocator *)gst_object_ref(singleton);
}

gboolean mvfx_is_hip_memory(GstMemory *mem)
{
    return mem && mem->allocator && g_strcmp0(mem->allocator->mem_type, MVFX_HIP_MEMORY_TYPE) == 0;
}

int mvfx_hip_memory_device(GstMemory *mem)
{
    return mvfx_is_hip_memory(mem) ? ((MvfxHipMemory *)mem)->device : -1;
}

int mvfx_hip_buffer_device(GstBuffer *buf)
{
    return buf && gst_buffer_n_memory(buf) > 0 ? mvfx_hip_memory_device(gst_buffer_peek_memory(buf, 0)) : -1;
}

gboolean mvfx_hip_follow_device(GstBuffer *buf, GstObject *owner)
{
    const int want = mvfx_hip_buffer_device(buf);
    if (want < 0) return TRUE;
    const int have = mvfx_current_device(); // (hipGetDevice: a thread-local read inside the runtime)
    if (have == want) return TRUE;
    if (mvfx_set_device(want) != MVFX_OK) {
        if (owner && GST_IS_ELEMENT(owner))
            GST_ELEMENT_ERROR(GST_ELEMENT(owner), RESOURCE, FAILED, ("cannot select device %d of the incoming memory: %s", want, mvfx_last_error()), (NULL));
        return FALSE;
    }
    if (owner) GST_INFO_OBJECT(owner, "Device updated from %d to %d", have, want);
    return TRUE;
}

gboolean mvfx_hip_select_device(gint device_id, GstElement *owner)
{
    if (device_id < 0) return TRUE;
    const int n = mvfx_device_count();
    if (device_id >= n) {
        if (owner) GST_ELEMENT_ERROR(owner, RESOURCE, NOT_FOUND, ("device-id %d: only %d HIP device(s) visible", device_id, n), (NULL));
        return FALSE;
    }
    if (mvfx_current_device() != device_id && mvfx_set_device(device_id) != MVFX_OK) {
        if (owner) GST_ELEMENT_ERROR(owner, RESOURCE, FAILED, ("device-id %d: %s", device_id, mvfx_last_error()), (NULL));
        return FALSE;
    }
    return TRUE;
}

gboolean mvfx_buffer_is_hip(GstBuffer *buf)
{
    return buf && gst_buffer_n_memory(buf) == 1 && mvfx_is_hip_memory(gst_buffer_peek_memory(buf, 0));
}

gboolean mvfx_caps_has_hip_feature(const GstCaps *caps)
{
    if (!caps || gst_caps_is_empty(caps) || gst_caps_is_any(caps))
        return FALSE;
    GstCapsFeatures *f = gst_caps_get_features(caps, 0);
    return f && gst_caps_features_contains(f, MVFX_CAPS_FEATURE_MEMORY_HIP);
}

GstCaps *mvfx_caps_set_hip_feature(const GstCaps *caps, gboolean hip)
{
    GstCaps *out = gst_caps_copy(caps);
    for (guint i = 0; i < gst_caps_get_size(out); i++)
        gst_caps_set_features(out, i, hip ? gst_caps_features_new(MVFX_CAPS_FEATURE_MEMORY_HIP, NULL)
                                          : gst_caps_features_new_empty());
    return out;
}

GstCaps *mvfx_caps_with_hip_feature(const GstCaps *system_caps)
{
    return mvfx_caps_set_hip_feature(system_caps, TRUE);
}

// ------------------------------------------------------------------------------------ buffer pool

typedef struct {
    GstBufferPool parent;
    GstVideoInfo info;
    gboolean have_info;
    gsize size;
} MvfxHipBufferPool;
typedef struct { GstBufferPoolClass parent_class; } MvfxHipBufferPoolClass;

G_DEFINE_TYPE(MvfxHipBufferPool, mvfx_hip_buffer_pool, GST_TYPE_BUFFER_POOL)

GST_DEBUG_CATEGORY_STATIC(mvfx_hip_pool_debug);
static volatile gint64 pool_allocated = 0, pool_acquired = 0;

static const gchar **mvfx_hip_pool_get_options(GstBufferPool *)
{
    static const gchar *options[] = {GST_BUFFER_POOL_OPTION_VIDEO_META, NULL};
    return options;
}

static gboolean mvfx_hip_pool_set_config(GstBufferPool *pool, GstStructure *config)
{
    MvfxHipBufferPool *self = (MvfxHipBufferPool *)pool;
    GstCaps *caps = NULL;
    guint size = 0, min = 0, max = 0;
    if (!gst_buffer_pool_config_get_params(config, &caps, &size, &min, &max) || !caps)
        return FALSE;
    self->have_info = gst_video_info_from_caps(&self->info, caps);
    if (self->have_info && GST_VIDEO_INFO_SIZE(&self->info) > size) {
        size = (guint)GST_VIDEO_INFO_SIZE(&self->info);
        gst_buffer_pool_config_set_params(config, caps, size, min, max);
    }
    self->size = size;
    GST_CAT_DEBUG_OBJECT(mvfx_hip_pool_debug, pool, "configured: %" GST_PTR_FORMAT " size %u min %u max %u", caps, size, min, max);
    return GST_BUFFER_POOL_CLASS(mvfx_hip_buffer_pool_parent_class)->set_config(pool, config);
}

static GstFlowReturn mvfx_hip_pool_alloc_buffer(GstBufferPool *pool, GstBuffer **buffer, GstBufferPoolAcquireParams *)
{
    MvfxHipBufferPool *self = (MvfxHipBufferPool *)pool;
    GstAllocator *alloc = mvfx_hip_allocator_get();
    GstBuffer *buf = gst_buffer_new_allocate(alloc, self->size, NULL);
    gst_object_unref(alloc);
    if (!buf)
        return GST_FLOW_ERROR;
    if (self->have_info) // device buffers always use the default GstVideoInfo layout
        gst_buffer_add_video_meta_full(buf, GST_VIDEO_FRAME_FLAG_NONE, GST_VIDEO_INFO_FORMAT(&self->info),
                                       GST_VIDEO_INFO_WIDTH(&self->info), GST_VIDEO_INFO_HEIGHT(&self->info),
                                       GST_VIDEO_INFO_N_PLANES(&self->info), self->info.offset, self->info.stride);
    __atomic_add_fetch(&pool_allocated, 1, __ATOMIC_RELAXED);
    GST_CAT_LOG_OBJECT(mvfx_hip_pool_debug, pool, "allocated device buffer %p (%" G_GSIZE_FORMAT " bytes)", (void *)buf, self->size);
    *buffer = buf;
    return GST_FLOW_OK;
}

static GstFlowReturn mvfx_hip_pool_acquire_buffer(GstBufferPool *pool, GstBuffer **buffer, GstBufferPoolAcquireParams *params)
{
    const GstFlowReturn ret = GST_BUFFER_POOL_CLASS(mvfx_hip_buffer_pool_parent_class)->acquire_buffer(pool, buffer, params);
    if (ret == GST_FLOW_OK)
        __atomic_add_fetch(&pool_acquired, 1, __ATOMIC_RELAXED);
    return ret;
}

// The pool's buffers go back to the allocator (its free list) when the pool stops; nothing of that size may be asked for again
// (caps change, READY -> NULL): give THIS pool's blocks back to the device -- the cached blocks of other sizes belong to other
// pipelines of the process, which keep them.
static gboolean mvfx_hip_pool_stop(GstBufferPool *pool)
{
    MvfxHipBufferPool *self = (MvfxHipBufferPool *)pool;
    const gboolean ok = GST_BUFFER_POOL_CLASS(mvfx_hip_buffer_pool_parent_class)->stop(pool);
    if (self->size > 0) freelist_trim(self->size);
    return ok;
}

static void mvfx_hip_buffer_pool_class_init(MvfxHipBufferPoolClass *klass)
{
    GstBufferPoolClass *pc = GST_BUFFER_POOL_CLASS(klass);
    pc->stop = mvfx_hip_pool_stop;
    pc->get_options = mvfx_hip_pool_get_options;
    pc->set_config = mvfx_hip_pool_set_config;
    pc->alloc_buffer = mvfx_hip_pool_alloc_buffer;
    pc->acquire_buffer = mvfx_hip_pool_acquire_buffer;
    GST_DEBUG_CATEGORY_INIT(mvfx_hip_pool_debug, "mvfxhippool", 0, "MI355X HIP device-memory buffer pool");
}

static void mvfx_hip_buffer_pool_init(MvfxHipBufferPool *self)
{
    self->have_info = FALSE;
    self->size = 0;
}

GstBufferPool *mvfx_hip_buffer_pool_new(void)
{
    GstBufferPool *pool = (GstBufferPool *)g_object_new(mvfx_hip_buffer_pool_get_type(), NULL);
    gst_object_ref_sink(pool);
    return pool;
}

gboolean mvfx_is_hip_buffer_pool(GstBufferPool *pool)
{
    return pool && G_TYPE_CHECK_INSTANCE_TYPE(pool, mvfx_hip_buffer_pool_get_type());
}

guint64 mvfx_hip_pool_buffers_allocated(void) { return (guint64)__atomic_load_n(&pool_allocated, __ATOMIC_RELAXED); }
guint64 mvfx_hip_pool_buffers_acquired(void) { return (guint64)__atomic_load_n(&pool_acquired, __ATOMIC_RELAXED); }

static GstBufferPool *new_configured_pool(GstCaps *caps, guint size)
{
    GstBufferPool *pool = mvfx_hip_buffer_pool_new();
    GstStructure *config = gst_buffer_pool_get_config(pool);
    gst_buffer_pool_config_set_params(config, caps, size, 0, 0);
    gst_buffer_pool_config_add_option(config, GST_BUFFER_POOL_OPTION_VIDEO_META);
    if (!gst_buffer_pool_set_config(pool, config)) {
        gst_object_unref(pool);
        return NULL;
    }
    return pool;
}

// ------------------------------------------------------------------------------------ pinned host pool

typedef struct {
    GstBufferPool parent;
    GstVideoInfo info;
    gboolean have_info;
    gsize size;
} MvfxPinnedBufferPool;
typedef struct { GstBufferPoolClass parent_class; } MvfxPinnedBufferPoolClass;

G_DEFINE_TYPE(MvfxPinnedBufferPool, mvfx_pinned_buffer_pool, GST_TYPE_BUFFER_POOL)

static const gchar **mvfx_pinned_pool_get_options(GstBufferPool *)
{
    static const gchar *options[] = {GST_BUFFER_POOL_OPTION_VIDEO_META, NULL};
    return options;
}

static gboolean mvfx_pinned_pool_set_config(GstBufferPool *pool, GstStructure *config)
{
    MvfxPinnedBufferPool *self = (MvfxPinnedBufferPool *)pool;
    GstCaps *caps = NULL;
    guint size = 0, min = 0, max = 0;
    if (!gst_buffer_pool_config_get_params(config, &caps, &size, &min, &max) || !caps)
        return FALSE;
    self->have_info = gst_video_info_from_caps(&self->info, caps);
    if (self->have_info && GST_VIDEO_INFO_SIZE(&self->info) > size) {
        size = (guint)GST_VIDEO_INFO_SIZE(&self->info);
        gst_buffer_pool_config_set_params(config, caps, size, min, max);
    }
    self->size = size;
    return GST_BUFFER_POOL_CLASS(mvfx_pinned_buffer_pool_parent_class)->set_config(pool, config);
}

static void pinned_block_free(gpointer block) { mvfx_host_free(block); }

static GstFlowReturn mvfx_pinned_pool_alloc_buffer(GstBufferPool *pool, GstBuffer **buffer, GstBufferPoolAcquireParams *)
{
    MvfxPinnedBufferPool *self = (MvfxPinnedBufferPool *)pool;
    void *block = NULL;
    GstBuffer *buf;
    if (mvfx_host_alloc(&block, self->size) == MVFX_OK) {
        buf = gst_buffer_new_wrapped_full((GstMemoryFlags)0, block, self->size, 0, self->size, block, pinned_block_free);
    } else { // still system memory, only slower to copy from / to
        GST_WARNING("pinned host allocation of %" G_GSIZE_FORMAT " bytes failed (%s): pageable buffer instead", self->size, mvfx_last_error());
        buf = gst_buffer_new_allocate(NULL, self->size, NULL);
        if (!buf) return GST_FLOW_ERROR;
    }
    if (self->have_info)
        gst_buffer_add_video_meta_full(buf, GST_VIDEO_FRAME_FLAG_NONE, GST_VIDEO_INFO_FORMAT(&self->info),
                                       GST_VIDEO_INFO_WIDTH(&self->info), GST_VIDEO_INFO_HEIGHT(&self->info),
                                       GST_VIDEO_INFO_N_PLANES(&self->info), self->info.offset, self->info.stride);
    *buffer = buf;
    return GST_FLOW_OK;
}

static void mvfx_pinned_buffer_pool_class_init(MvfxPinnedBufferPoolClass *klass)
{
    GstBufferPoolClass *pc = GST_BUFFER_POOL_CLASS(klass);
    pc->get_options = mvfx_pinned_pool_get_options;
    pc->set_config = mvfx_pinned_pool_set_config;
    pc->alloc_buffer = mvfx_pinned_pool_alloc_buffer;
}

static void mvfx_pinned_buffer_pool_init(MvfxPinnedBufferPool *self)
{
    self->have_info = FALSE;
    self->size = 0;
}

GstBufferPool *mvfx_pinned_buffer_pool_new(void)
{
    GstBufferPool *pool = (GstBufferPool *)g_object_new(mvfx_pinned_buffer_pool_get_type(), NULL);
    gst_object_ref_sink(pool);
    return pool;
}

GstBufferPool *mvfx_pinned_buffer_pool_new_configured(GstCaps *caps, guint size, guint min_buffers)
{
    GstBufferPool *pool = mvfx_pinned_buffer_pool_new();
    GstStructure *config = gst_buffer_pool_get_config(pool);
    gst_buffer_pool_config_set_params(config, caps, size, min_buffers, 0);
    gst_buffer_pool_config_add_option(config, GST_BUFFER_POOL_OPTION_VIDEO_META);
    if (!gst_buffer_pool_set_config(pool, config)) {
        gst_object_unref(pool);
        return NULL;
    }
    return pool;
}

gboolean mvfx_hip_propose_allocation(GstQuery *query)
{
    GstCaps *caps = NULL;
    gboolean need_pool = FALSE;
    gst_query_parse_allocation(query, &caps, &need_pool);
    GstVideoInfo info;
    if (!caps || !mvfx_caps_has_hip_feature(caps) || !gst_video_info_from_caps(&info, caps))
        return FALSE;
    const guint size = (guint)GST_VIDEO_INFO_SIZE(&info);
    GstBufferPool *pool = need_pool ? new_configured_pool(caps, size) : NULL;
    gst_query_add_allocation_pool(query, pool, size, 0, 0);
    if (pool)
        gst_object_unref(pool);
    GstAllocator *alloc = mvfx_hip_allocator_get();
    gst_query_add_allocation_param(query, alloc, NULL);
    gst_object_unref(alloc);
    gst_query_add_allocation_meta(query, GST_VIDEO_META_API_TYPE, NULL);
    return TRUE;
}

gboolean mvfx_hip_decide_allocation(GstQuery *query)
{
    GstCaps *caps = NULL;
    gst_query_parse_allocation(query, &caps, NULL);
    GstVideoInfo info;
    if (!caps || !mvfx_caps_has_hip_feature(caps) || !gst_video_info_from_caps(&info, caps))
        return FALSE;
    guint size = (guint)GST_VIDEO_INFO_SIZE(&info), min = 0, max = 0;
    GstBufferPool *pool = NULL;
    const gboolean have = gst_query_get_n_allocation_pools(query) > 0;
    if (have) {
        guint psize = 0;
        gst_query_parse_nth_allocation_pool(query, 0, &pool, &psize, &min, &max);
        if (psize > size)
            size = psize;
        if (pool && !mvfx_is_hip_buffer_pool(pool)) { // a system-memory pool cannot back memory:HIPMemory caps
            gst_object_unref(pool);
            pool = NULL;
        }
    }
    if (!pool)
        pool = new_configured_pool(caps, size);
    if (!pool)
        return FALSE;
    // A few blocks in rotation (the pool's queue is first in, first out): behind a sink that drops its buffers at once a pool without a
    // minimum recycles ONE block, every frame then writes where the previous frame's kernel may still be writing, and consecutive
    // frames can never overlap -- with the elements alternating between two streams that wait even crosses streams
    // (MVFX_HIP_POOL_MIN, default 4; profiles/r3/gst_pool_min_buffers.txt).
    static const guint pool_min = [] { const gchar *e = g_getenv("MVFX_HIP_POOL_MIN"); return (guint)CLAMP(e ? atoi(e) : 4, 0, 64); }();
    if (min < pool_min) min = pool_min;
    if (max != 0 && max < min) max = min;
    if (have)
        gst_query_set_nth_allocation_pool(query, 0, pool, size, min, max);
    else
        gst_query_add_allocation_pool(query, pool, size, min, max);
    gst_object_unref(pool);
    // the base class would otherwise hand the pool a system-memory allocator from the query's params
    while (gst_query_get_n_allocation_params(query) > 0)
        gst_query_remove_nth_allocation_param(query, 0);
    GstAllocator *alloc = mvfx_hip_allocator_get();
    gst_query_add_allocation_param(query, alloc, NULL);
    gst_object_unref(alloc);
    return TRUE;
}


// ---- idle flush: a process-wide timer for work that elements hold back (mvfx_pair_hold.h) ---------------------------------------
//
// A held-back frame leaves with the next buffer, or when somebody looks at its blocks.  A live source that stalls would otherwise
// leave its last frame unprocessed until EOS: the hold arms this timer for one frame interval whenever a frame becomes the held-back
// one; re-arming moves the deadline.  One entry per owner, referenced while armed; the callback runs on the timer thread with no
// lock of this file held.
namespace {
struct IdleEntry {
    GstObject *owner;
    MvfxDeferredFlush cb;
    gint64 deadline; // g_get_monotonic_time()
};
GMutex idle_lock;
GCond idle_cond;
std::vector<IdleEntry> *idle_entries; // never freed: the thread lives as long as the process
gboolean idle_thread_started;

gpointer idle_thread(gpointer)
{
    g_mutex_lock(&idle_lock);
    for (;;) {
        gint64 first = G_MAXINT64;
        for (const IdleEntry &e : *idle_entries) first = MIN(first, e.deadline);
        if (first == G_MAXINT64) {
            g_cond_wait(&idle_cond, &idle_lock);
            continue;
        }
        const gint64 now = g_get_monotonic_time();
        if (first > now) {
            g_cond_wait_until(&idle_cond, &idle_lock, first);
            continue;
        }
        std::vector<IdleEntry> due;
        for (size_t i = 0; i < idle_entries->size();) {
            if ((*idle_entries)[i].deadline <= now) {
                due.push_back((*idle_entries)[i]);
                (*idle_entries)[i] = idle_entries->back();
                idle_entries->pop_back();
            } else
                i++;
        }
        g_mutex_unlock(&idle_lock);
        for (const IdleEntry &e : due) {
            e.cb(e.owner);
            gst_object_unref(e.owner);
        }
        g_mutex_lock(&idle_lock);
    }
    return NULL;
}
} // namespace

void mvfx_idle_arm(GstObject *owner, MvfxDeferredFlush cb, guint64 after_us)
{
    if (!owner || !cb) return;
    const gint64 deadline = g_get_monotonic_time() + (gint64)after_us;
    g_mutex_lock(&idle_lock);
    if (!idle_entries) idle_entries = new std::vector<IdleEntry>();
    gboolean found = FALSE, wake = FALSE;
    for (IdleEntry &e : *idle_entries)
        if (e.owner == owner) { // later than before: the thread wakes at the old deadline, finds nothing due and sleeps again
            e.deadline = deadline;
            e.cb = cb;
            found = TRUE;
        }
    if (!found) {
        idle_entries->push_back(IdleEntry{GST_OBJECT(gst_object_ref(owner)), cb, deadline});
        wake = TRUE;
    }
    if (!idle_thread_started) {
        idle_thread_started = TRUE;
        g_thread_unref(g_thread_new("mvfx-idle-flush", idle_thread, NULL));
    } else if (wake)
        g_cond_signal(&idle_cond);
    g_mutex_unlock(&idle_lock);
}

void mvfx_idle_cancel(GstObject *owner)
{
    GstObject *drop = NULL;
    g_mutex_lock(&idle_lock);
    if (idle_entries)
        for (size_t i = 0; i < idle_entries->size(); i++)
            if ((*idle_entries)[i].owner == owner) {
                drop = owner;
                (*idle_entries)[i] = idle_entries->back();
                idle_entries->pop_back();
                break;
            }
    g_mutex_unlock(&idle_lock);
    if (drop) gst_object_unref(drop);
}
