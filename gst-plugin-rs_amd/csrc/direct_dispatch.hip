// Host side of the direct-dispatch lane (direct_dispatch.h has the why): per device two HSA queues of the library's own, the lane's kernels loaded
// from the code objects embedded in this library (build/direct_blob.o: csrc/direct/hsv_direct_kernels.hip and colorlut_direct_kernels.hip as bare
// gfx950 ELFs), a ring of kernel argument blocks, and the table that turns an mvfx_event into a direct fence (an HSA completion signal).
#include "direct_dispatch.h"
#include "direct_dispatch_colorlut.h"
#include "mvfx_internal.h"

#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <shared_mutex>
#include <unordered_map>
#include <vector>

extern "C" const char mvfx_direct_hsaco[];
extern "C" const char mvfx_direct_hsaco_end[];
extern "C" const char mvfx_direct_lut_hsaco[];
extern "C" const char mvfx_direct_lut_hsaco_end[];

namespace mvfx {

const char *const kDirectKernelNames[kDirectKernels] = {"mvfx_direct_hsvfilter4_pos.kd", "mvfx_direct_hsvfilter4_pos_nt.kd",
                                                        "mvfx_direct_hsvfilter4_neg.kd", "mvfx_direct_hsvfilter4_neg_nt.kd", "mvfx_direct_hsvdetector4.kd",
                                                        "mvfx_direct_colorlut_xtile.kd", "mvfx_direct_colorlut_xwg.kd"};

namespace {

constexpr uint32_t kQueuePackets = 1024, kArgSlots = 1024, kArgSlotBytes = 192, kQueues = 2;
static_assert(sizeof(DirectHsvArgs) <= kArgSlotBytes && sizeof(DirectDetArgs) <= kArgSlotBytes && sizeof(DirectLutArgs) <= kArgSlotBytes,
              "one kernel argument block per slot");
constexpr size_t kKernelArgBytes[kDirectKernels] = {sizeof(DirectHsvArgs), sizeof(DirectHsvArgs), sizeof(DirectHsvArgs), sizeof(DirectHsvArgs), sizeof(DirectDetArgs),
                                                    sizeof(DirectLutArgs), sizeof(DirectLutArgs)};

struct Lane {
    std::atomic<bool> ok{false};
    hsa_agent_t agent{};
    hsa_queue_t *queue[kQueues] = {};  // frames alternate between them (each in order: its packets carry the barrier bit)
    std::atomic<uint64_t> next{0};     // dispatch counter: argument slot = next % kArgSlots
    char *args = nullptr;              // kArgSlots x kArgSlotBytes in DEVICE memory, written by the CPU through the BAR
    uint64_t kernel[kDirectKernels] = {};
    uint32_t kernel_lds[kDirectKernels] = {};
    hsa_executable_t exe{};
    // the completion signal of the dispatch that used an argument slot last: the slot is written again only when that kernel has finished (its
    // workgroups read the block as they start).  Signals are never destroyed (free list below), so a stale handle is still a signal.
    std::atomic<uint64_t> slot_signal[kArgSlots];
    // PARKING (direct_park): the two hardware queues exist only while the lane is in use.  Dispatchers hold `qmu` shared while they touch a queue;
    // parking and un-parking take it exclusively.
    std::shared_mutex qmu;
    bool parked = false;
};

bool create_queues(Lane &l)
{
    for (uint32_t k = 0; k < kQueues; k++)
        if (hsa_queue_create(l.agent, kQueuePackets, HSA_QUEUE_TYPE_MULTI, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &l.queue[k]) != HSA_STATUS_SUCCESS) {
            for (uint32_t j = 0; j < k; j++) { (void)hsa_queue_destroy(l.queue[j]); l.queue[j] = nullptr; }
            l.queue[k] = nullptr;
            return false;
        }
    return true;
}

// called with l.qmu held shared: the lane's queues exist, or the lock is swapped for the exclusive one while they are made again
bool ensure_unparked(Lane &l, std::shared_lock<std::shared_mutex> &lk)
{
    while (l.parked) {
        lk.unlock();
        {
            std::unique_lock<std::shared_mutex> x(l.qmu);
            if (l.parked) {
                if (!create_queues(l)) { lk.lock(); return false; }
                l.parked = false;
            }
        }
        lk.lock();
    }
    return true;
}

bool enabled()
{
    static const bool on = [] { const char *e = getenv("MVFX_DIRECT_DISPATCH"); return !(e && atoi(e) == 0); }();
    return on;
}

struct AgentPick { int want_domain, want_bus, want_dev; hsa_agent_t agent; bool found; };
hsa_status_t match_agent(hsa_agent_t a, void *data)
{
    AgentPick *p = static_cast<AgentPick *>(data);
    hsa_device_type_t t;
    if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS || t != HSA_DEVICE_TYPE_GPU) return HSA_STATUS_SUCCESS;
    uint32_t bdf = 0, domain = 0;
    if (hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
    (void)hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &domain);
    if ((int)((bdf >> 8) & 0xff) == p->want_bus && (int)((bdf >> 3) & 0x1f) == p->want_dev && (int)domain == p->want_domain && !p->found) {
        p->agent = a;
        p->found = true;
    }
    return HSA_STATUS_SUCCESS;
}

struct CpuPick { hsa_agent_t agent; bool found; };
hsa_status_t first_cpu(hsa_agent_t a, void *data)
{
    CpuPick *p = static_cast<CpuPick *>(data);
    hsa_device_type_t t;
    if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) == HSA_STATUS_SUCCESS && t == HSA_DEVICE_TYPE_CPU && !p->found) { p->agent = a; p->found = true; }
    return HSA_STATUS_SUCCESS;
}

struct PoolPick { hsa_amd_memory_pool_t pool; bool found; };
hsa_status_t device_pool(hsa_amd_memory_pool_t pool, void *data)
{
    PoolPick *p = static_cast<PoolPick *>(data);
    hsa_amd_segment_t seg;
    uint32_t flags = 0;
    bool alloc = false;
    if (hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg) != HSA_STATUS_SUCCESS || seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
    (void)hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
    (void)hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
    if ((flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED) && alloc && !p->found) { p->pool = pool; p->found = true; }
    return HSA_STATUS_SUCCESS;
}

// one attempt per device and process; a failure leaves ok == false and the callers on their HIP streams
void build_lane(Lane &l, int device)
{
    if (hsa_init() != HSA_STATUS_SUCCESS) return; // (reference counted: HIP holds the runtime open already)
    int dom = 0, bus = 0, dev = 0;
    if (hipDeviceGetAttribute(&dom, hipDeviceAttributePciDomainID, device) != hipSuccess || hipDeviceGetAttribute(&bus, hipDeviceAttributePciBusId, device) != hipSuccess ||
        hipDeviceGetAttribute(&dev, hipDeviceAttributePciDeviceId, device) != hipSuccess)
        return;
    AgentPick pick{dom, bus, dev, {}, false};
    if (hsa_iterate_agents(match_agent, &pick) != HSA_STATUS_SUCCESS || !pick.found) return;
    l.agent = pick.agent;
    // The argument blocks live in DEVICE memory and the CPU writes them through the BAR, as HIP does with its own (HIP_FORCE_DEV_KERNARG): every
    // wave's first instruction loads them, and from host memory that is a PCIe round trip per wave -- the same kernel ran 30 us per 4K frame with
    // host-memory argument blocks and 15.7 with device-memory ones (profiles/r6/aql_probe_real_kernel_*_kernarg.txt).  No BAR access, no lane.
    CpuPick cpu{{}, false};
    PoolPick pool{{}, false};
    if (hsa_iterate_agents(first_cpu, &cpu) != HSA_STATUS_SUCCESS || !cpu.found) return;
    if (hsa_amd_agent_iterate_memory_pools(l.agent, device_pool, &pool) != HSA_STATUS_SUCCESS || !pool.found) return;
    hsa_amd_memory_pool_access_t access = HSA_AMD_MEMORY_POOL_ACCESS_NEVER_ALLOWED;
    if (hsa_amd_agent_memory_pool_get_info(cpu.agent, pool.pool, HSA_AMD_AGENT_MEMORY_POOL_INFO_ACCESS, &access) != HSA_STATUS_SUCCESS ||
        access == HSA_AMD_MEMORY_POOL_ACCESS_NEVER_ALLOWED)
        return;
    hsa_code_object_reader_t reader, lut_reader;
    if (hsa_code_object_reader_create_from_memory(mvfx_direct_hsaco, (size_t)(mvfx_direct_hsaco_end - mvfx_direct_hsaco), &reader) != HSA_STATUS_SUCCESS) return;
    if (hsa_code_object_reader_create_from_memory(mvfx_direct_lut_hsaco, (size_t)(mvfx_direct_lut_hsaco_end - mvfx_direct_lut_hsaco), &lut_reader) != HSA_STATUS_SUCCESS) return;
    if (hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &l.exe) != HSA_STATUS_SUCCESS) return;
    if (hsa_executable_load_agent_code_object(l.exe, l.agent, reader, nullptr, nullptr) != HSA_STATUS_SUCCESS) return;
    if (hsa_executable_load_agent_code_object(l.exe, l.agent, lut_reader, nullptr, nullptr) != HSA_STATUS_SUCCESS) return;
    if (hsa_executable_freeze(l.exe, nullptr) != HSA_STATUS_SUCCESS) return;
    for (int k = 0; k < kDirectKernels; k++) {
        hsa_executable_symbol_t sym;
        uint32_t kasize = 0, lds = 0, priv = 0;
        if (hsa_executable_get_symbol_by_name(l.exe, kDirectKernelNames[k], &l.agent, &sym) != HSA_STATUS_SUCCESS) return;
        if (hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &l.kernel[k]) != HSA_STATUS_SUCCESS) return;
        (void)hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &kasize);
        (void)hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &lds);
        (void)hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &priv);
        if (kasize != kKernelArgBytes[k] || priv != 0) return; // the argument block is the struct and nothing else (no implicit arguments, no scratch)
        l.kernel_lds[k] = lds;
    }
    if (hsa_amd_memory_pool_allocate(pool.pool, (size_t)kArgSlots * kArgSlotBytes, 0, reinterpret_cast<void **>(&l.args)) != HSA_STATUS_SUCCESS) return;
    if (hsa_amd_agents_allow_access(1, &cpu.agent, nullptr, l.args) != HSA_STATUS_SUCCESS) return;
    std::memset(l.args, 0, (size_t)kArgSlots * kArgSlotBytes);
    for (auto &s : l.slot_signal) s.store(0, std::memory_order_relaxed);
    if (!create_queues(l)) return;
    l.ok.store(true, std::memory_order_release);
}

constexpr int kMaxDevices = 64;
Lane g_lanes[kMaxDevices];
std::once_flag g_lane_once[kMaxDevices];
Lane *lane_of(int device)
{
    if (device < 0 || device >= kMaxDevices) return nullptr;
    std::call_once(g_lane_once[device], [device] { build_lane(g_lanes[device], device); });
    return g_lanes[device].ok.load(std::memory_order_acquire) ? &g_lanes[device] : nullptr;
}

// ---- direct fences ----------------------------------------------------------------------------------------------------------------
// hipEvent_t -> the HSA signal of its last lane dispatch.  An event keeps ONE signal for its life (made on its first lane use); `direct` says whether
// the event's current meaning is that signal (true) or an ordinary HIP record (false).
struct DirectFence {
    hsa_signal_t sig{};
    std::atomic<bool> direct{false};
    std::atomic<int> queue{-1}; // the lane queue of the dispatch it stands for
};
std::shared_mutex g_fence_mu;
std::unordered_map<hipEvent_t, DirectFence *> g_fences;
std::vector<DirectFence *> g_fence_free; // of destroyed events (their signals live on: Lane::slot_signal may still name them)

DirectFence *fence_find(hipEvent_t e)
{
    std::shared_lock<std::shared_mutex> g(g_fence_mu);
    auto it = g_fences.find(e);
    return it == g_fences.end() ? nullptr : it->second;
}

DirectFence *fence_get(hipEvent_t e)
{
    if (DirectFence *f = fence_find(e)) return f;
    std::unique_lock<std::shared_mutex> g(g_fence_mu);
    auto it = g_fences.find(e);
    if (it != g_fences.end()) return it->second; // another thread was faster
    DirectFence *fresh = nullptr;
    if (!g_fence_free.empty()) {
        fresh = g_fence_free.back();
        g_fence_free.pop_back();
    } else {
        fresh = new (std::nothrow) DirectFence();
        if (!fresh) return nullptr;
        if (hsa_signal_create(0, 0, nullptr, &fresh->sig) != HSA_STATUS_SUCCESS) { delete fresh; return nullptr; }
    }
    fresh->direct.store(false, std::memory_order_relaxed);
    g_fences.emplace(e, fresh);
    return fresh;
}

void wait_signal(hsa_signal_t s)
{
    // active wait first (a 4K frame is ~12-25 us away; the timeout is in ticks of the 100 MHz system timestamp: 50 us), then blocked
    if (hsa_signal_wait_scacquire(s, HSA_SIGNAL_CONDITION_LT, 1, 5000, HSA_WAIT_STATE_ACTIVE) < 1) return;
    while (hsa_signal_wait_scacquire(s, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED) >= 1) {}
}

} // namespace

int direct_queue_wait(hipEvent_t e, int queue)
{
    DirectFence *f = fence_find(e);
    if (!f || !f->direct.load(std::memory_order_acquire) || hsa_signal_load_scacquire(f->sig) < 1) return 0;
    const uint32_t mine = (uint32_t)queue % kQueues;
    if ((uint32_t)f->queue.load(std::memory_order_relaxed) % kQueues == mine) return 1; // in front of the caller in the same in-order queue
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) { (void)hipGetLastError(); return fail(MVFX_ERR_DEVICE, "direct_queue_wait: no current device"); }
    Lane *l = lane_of(device);
    if (!l) return fail(MVFX_ERR_DIRECT_UNAVAILABLE, "direct_queue_wait: no lane on device %d", device);
    std::shared_lock<std::shared_mutex> lk(l->qmu);
    if (!ensure_unparked(*l, lk)) return fail(MVFX_ERR_DIRECT_UNAVAILABLE, "direct_queue_wait: the lane's queues could not be made again");
    hsa_queue_t *q = l->queue[mine];
    const uint64_t idx = hsa_queue_add_write_index_relaxed(q, 1);
    while (idx - hsa_queue_load_read_index_scacquire(q) >= q->size) {}
    hsa_barrier_and_packet_t *p = reinterpret_cast<hsa_barrier_and_packet_t *>(q->base_address) + (idx & (q->size - 1));
    std::memset(reinterpret_cast<char *>(p) + 4, 0, sizeof *p - 4);
    p->dep_signal[0] = f->sig; // satisfied when the signal is 0: the other queue's dispatch has finished
    // no barrier bit on the barrier packet itself (it waits for its dependency, not for what is in front of it here); the caller's kernel packet behind
    // it carries the bit and so waits for this packet
    const uint16_t header = (uint16_t)(HSA_PACKET_TYPE_BARRIER_AND << HSA_PACKET_HEADER_TYPE);
    __atomic_store_n(reinterpret_cast<uint32_t *>(p), (uint32_t)header, __ATOMIC_RELEASE);
    hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)idx);
    return 1;
}

namespace {
// a barrier packet behind each queue, waited for here; the caller holds l.qmu (shared or exclusive) and the lane is not parked
void quiesce_locked(Lane &l)
{
    hsa_signal_t done[kQueues] = {};
    uint32_t n = 0;
    for (uint32_t k = 0; k < kQueues; k++) {
        if (hsa_signal_create(1, 0, nullptr, &done[n]) != HSA_STATUS_SUCCESS) continue;
        hsa_queue_t *q = l.queue[k];
        const uint64_t idx = hsa_queue_add_write_index_relaxed(q, 1);
        while (idx - hsa_queue_load_read_index_scacquire(q) >= q->size) {}
        hsa_barrier_and_packet_t *p = reinterpret_cast<hsa_barrier_and_packet_t *>(q->base_address) + (idx & (q->size - 1));
        std::memset(reinterpret_cast<char *>(p) + 4, 0, sizeof *p - 4); // no dependency signals: the barrier BIT is what waits for everything in front
        p->completion_signal = done[n];
        const uint16_t header = (uint16_t)((HSA_PACKET_TYPE_BARRIER_AND << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER));
        __atomic_store_n(reinterpret_cast<uint32_t *>(p), (uint32_t)header, __ATOMIC_RELEASE);
        hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)idx);
        n++;
    }
    for (uint32_t k = 0; k < n; k++) {
        wait_signal(done[k]);
        (void)hsa_signal_destroy(done[k]);
    }
}
} // namespace

void direct_quiesce(int device)
{
    if (device < 0 || device >= kMaxDevices || !g_lanes[device].ok.load(std::memory_order_acquire)) return; // (never builds a lane)
    Lane &l = g_lanes[device];
    std::shared_lock<std::shared_mutex> lk(l.qmu);
    if (!l.parked) quiesce_locked(l); // (a parked lane was drained when it was parked)
}

int direct_park(int device)
{
    if (device < 0 || device >= kMaxDevices || !g_lanes[device].ok.load(std::memory_order_acquire)) return 0;
    Lane &l = g_lanes[device];
    std::unique_lock<std::shared_mutex> x(l.qmu);
    if (l.parked) return 0;
    quiesce_locked(l);
    for (uint32_t k = 0; k < kQueues; k++) {
        (void)hsa_queue_destroy(l.queue[k]);
        l.queue[k] = nullptr;
    }
    l.parked = true;
    return 1;
}

int direct_event_state(hipEvent_t e)
{
    DirectFence *f = fence_find(e);
    if (!f || !f->direct.load(std::memory_order_acquire)) return 0;
    return hsa_signal_load_scacquire(f->sig) < 1 ? 1 : 2;
}

int direct_event_queue(hipEvent_t e)
{
    DirectFence *f = fence_find(e);
    return f && f->direct.load(std::memory_order_acquire) ? f->queue.load(std::memory_order_relaxed) : -1;
}

int direct_queue_hint(hipStream_t stream)
{
    const int idx = thread_stream_index(stream);
    if (idx >= 0) return idx & 1;
    const uint64_t h = reinterpret_cast<uint64_t>(stream) * 0x9E3779B97F4A7C15ull;
    return (int)(h >> 63);
}

int direct_event_wait(hipEvent_t e)
{
    DirectFence *f = fence_find(e);
    if (f && f->direct.load(std::memory_order_acquire)) wait_signal(f->sig);
    return MVFX_OK;
}

void direct_event_forget(hipEvent_t e)
{
    DirectFence *f = fence_find(e);
    if (!f || !f->direct.load(std::memory_order_acquire)) return;
    wait_signal(f->sig); // (re-recording an event whose lane dispatch is still running: finish that first -- nobody may miss it)
    f->direct.store(false, std::memory_order_release);
}

void direct_event_destroy(hipEvent_t e)
{
    DirectFence *f = nullptr;
    {
        std::unique_lock<std::shared_mutex> g(g_fence_mu);
        auto it = g_fences.find(e);
        if (it == g_fences.end()) return;
        f = it->second;
        g_fences.erase(it);
    }
    wait_signal(f->sig);
    f->direct.store(false, std::memory_order_relaxed);
    std::unique_lock<std::shared_mutex> g(g_fence_mu);
    g_fence_free.push_back(f);
}

namespace {
// one dispatch: `bytes` of kernel arguments, kernel `k`, `wgs_x` x `wgs_y` workgroups of 256 lanes, on lane queue `queue`; in_order: with the barrier bit
int submit(const void *args, size_t bytes, int k, uint32_t wgs_x, uint32_t wgs_y, int queue, bool in_order = true)
{
    if (!enabled()) return 1;
    hipEvent_t ev = completion_event();
    if (!ev) return 1; // no fence to carry the completion: the caller orders by its stream
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) { (void)hipGetLastError(); return 1; }
    Lane *l = lane_of(device);
    if (!l) return 1;
    DirectFence *f = fence_get(ev);
    if (!f) return 1;
    if (f->direct.load(std::memory_order_acquire)) wait_signal(f->sig); // this event's previous lane dispatch (never pending in practice: fences are re-used when unreferenced)
    std::shared_lock<std::shared_mutex> lk(l->qmu); // (the queues stay until this dispatch is in one)
    if (!ensure_unparked(*l, lk)) return 1;

    const uint64_t n = l->next.fetch_add(1, std::memory_order_relaxed);
    hsa_queue_t *q = l->queue[(uint32_t)queue % kQueues];
    // the argument block: a slot of its own until the kernel that read it last has finished
    char *slot = l->args + (size_t)(n % kArgSlots) * kArgSlotBytes;
    if (const uint64_t last = l->slot_signal[n % kArgSlots].exchange(f->sig.handle, std::memory_order_acq_rel); last != 0 && last != f->sig.handle)
        wait_signal(hsa_signal_t{last}); // (1024 dispatches ago: long finished -- unless the signal has been armed again since, then for that dispatch)
    // ARM THE FENCE ONLY NOW, behind the last wait of this function: an armed signal is one whose dispatch goes out without waiting for anybody.
    // (Until round 6's soak the signal was armed in front of the slot wait.  Signals are few and re-armed all the time, so the slot's "last user"
    // is usually a signal somebody has armed again: two streaming threads -- a `queue` between two lane elements -- each armed their own and then
    // waited for the other's, which was never submitted.  tools/soak_lane_chain.py; tests/test_direct_dispatch_gpu.py::test_two_threads_few_fences.)
    hsa_signal_store_relaxed(f->sig, 1);
    f->queue.store((int)((uint32_t)queue % kQueues), std::memory_order_relaxed);
    f->direct.store(true, std::memory_order_release);
    note_completion_event_used();

    std::memcpy(slot, args, bytes);
    // device memory written through the BAR: the writes must have LANDED before the doorbell can lead a wave to them -- store fence, the last
    // byte once more, full fence, read it back (what the HIP runtime does for its own device-memory argument blocks)
    __builtin_ia32_sfence();
    reinterpret_cast<volatile char *>(slot)[bytes - 1] = static_cast<const char *>(args)[bytes - 1];
    __builtin_ia32_mfence();
    const volatile char landed = reinterpret_cast<volatile char *>(slot)[bytes - 1];
    (void)landed;
    const uint64_t idx = hsa_queue_add_write_index_relaxed(q, 1);
    while (idx - hsa_queue_load_read_index_scacquire(q) >= q->size) {} // queue full: 1024 frames in flight on it
    hsa_kernel_dispatch_packet_t *p = reinterpret_cast<hsa_kernel_dispatch_packet_t *>(q->base_address) + (idx & (q->size - 1));
    p->workgroup_size_x = 256; p->workgroup_size_y = 1; p->workgroup_size_z = 1; p->reserved0 = 0;
    p->grid_size_x = wgs_x * 256u; p->grid_size_y = wgs_y; p->grid_size_z = 1;
    p->private_segment_size = 0;
    p->group_segment_size = l->kernel_lds[k];
    p->kernel_object = l->kernel[k];
    p->kernarg_address = slot;
    p->reserved2 = 0;
    p->completion_signal = f->sig;
    // What the packet says -- and what a HIP stream's packets cannot be made to say.  Measured with this kernel, 4K frames, microseconds per frame
    // (tools/probes/aql_scope.cpp, profiles/r6/aql_probe_real_kernel_store_policies.txt):
    //                                              one queue    two queues alternating
    //   acquire agent, release agent (HIP's)         13.5-15.8    11.4-12.2      <- a HIP stream pair: 12.3
    //   acquire agent, release NONE                  13.4-13.9    11.0
    // The RELEASE fence is the cost: an L2 write-back walk on all eight XCDs at the end of every dispatch, whoever else is running.  The lane's
    // kernels store write-through and drain (csrc/direct/hsv_direct_kernels.hip), so there is nothing for it to write back: release NONE.  The
    // acquire stays at agent scope (the frame was written by somebody else's kernel or copy; cheap).  Barrier bit set: each of the two queues
    // is in order, like a stream -- without it the same kernel ran no faster (one queue) or slower (two).  colorlut's kernels are another matter (a
    // one-frame grid is 1.05 occupancy rounds of workgroups that each fill a window before their first pixel): without the bit the next frame's
    // workgroups start in the tail of this one's, as the frames of a batched launch do -- per-wave windows, natural-like 4K: 70.3 k fps in order,
    // 75-76 k without (two HIP streams: 69-70 k; profiles/r6/colorlut_lane.txt).  Only for a caller that says nothing in front matters to it.
    const uint16_t header = (uint16_t)((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | ((in_order ? 1 : 0) << HSA_PACKET_HEADER_BARRIER) |
                                       (HSA_FENCE_SCOPE_AGENT << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) |
                                       (HSA_FENCE_SCOPE_NONE << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
    const uint16_t setup = (uint16_t)((wgs_y > 1 ? 2 : 1) << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS);
    __atomic_store_n(reinterpret_cast<uint32_t *>(p), (uint32_t)header | ((uint32_t)setup << 16), __ATOMIC_RELEASE);
    hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)idx);
    return MVFX_OK;
}
} // namespace

int direct_hsvfilter_submit(const DirectHsvArgs &args, bool neg_shift, bool nontemporal, int queue)
{
    return submit(&args, sizeof args, (neg_shift ? 2 : 0) + (nontemporal ? 1 : 0), (args.groups + 511u) / 512u, 1, queue);
}

int direct_hsvdetector_submit(const DirectDetArgs &args, int queue)
{
    return submit(&args, sizeof args, 4, (args.groups + 511u) / 512u, 1, queue);
}

int direct_colorlut_submit(const DirectLutArgs &args, bool wg_window, int queue, bool in_order)
{
    if (wg_window) // the workgroup owns 128 x 40 pixels (colorlut_xwg_body)
        return submit(&args, sizeof args, 6, (args.width + 127u) / 128u, (args.height + 8u * kXRows - 1u) / (8u * kXRows), queue, in_order);
    // a wave owns 64 x 20 pixels, four waves side by side (colorlut_xtile_body)
    return submit(&args, sizeof args, 5, ((args.width + 63u) / 64u + 3u) / 4u, (args.height + 4u * kXRows - 1u) / (4u * kXRows), queue, in_order);
}

} // namespace mvfx
