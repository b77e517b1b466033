// tools/probes/aql_scope.cpp -- round 6: what does a kernel boundary of the single-frame path cost, and which part of the dispatch packet is it?
// tools/exp_boundary_cost.py showed that a packet with NO work costs 0.9-2 us of chip time even on another queue.  Here the same streaming shape
// (one 33 MB frame per dispatch, in place) goes out as hand-written AQL packets on this process's own HSA queues, with the packet header's acquire /
// release fence scopes (the cache maintenance the command processor performs around every dispatch), the barrier bit and the completion signal varied:
//     scopes  none / agent / system   x   barrier bit 0 / 1   x   1 / 2 queues   x   a completion signal on every packet or on every 64th
// Frames: 16 x 3840 x 2160 x 4 bytes from hipMalloc (HIP and HSA share the process's address space).  Each cell: microseconds per frame over 3000 frames.
//   hipcc -O2 tools/probes/aql_scope.cpp -o tools/probes/aql_scope.bin -lhsa-runtime64        (run from the repo root: it loads the .hsaco beside it)
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define HSA_OK(x) do { hsa_status_t s_ = (x); if (s_ != HSA_STATUS_SUCCESS) { const char *m_ = nullptr; hsa_status_string(s_, &m_); fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, m_ ? m_ : "?"); exit(2); } } while (0)
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(2); } } while (0)

static hsa_agent_t g_gpu;
static hsa_region_t g_kernarg;
static bool g_have_gpu = false, g_have_kernarg = false;

static hsa_status_t pick_agent(hsa_agent_t a, void *)
{
    hsa_device_type_t t;
    hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_GPU && !g_have_gpu) { g_gpu = a; g_have_gpu = true; }
    return HSA_STATUS_SUCCESS;
}
static hsa_status_t pick_region(hsa_region_t r, void *)
{
    hsa_region_segment_t seg;
    hsa_region_get_info(r, HSA_REGION_INFO_SEGMENT, &seg);
    uint32_t flags = 0;
    hsa_region_get_info(r, HSA_REGION_INFO_GLOBAL_FLAGS, &flags);
    if (seg == HSA_REGION_SEGMENT_GLOBAL && (flags & HSA_REGION_GLOBAL_FLAG_KERNARG) && !g_have_kernarg) { g_kernarg = r; g_have_kernarg = true; }
    return HSA_STATUS_SUCCESS;
}

static hsa_agent_t g_cpu;
static bool g_have_cpu = false;
static hsa_amd_memory_pool_t g_devpool;
static bool g_have_devpool = false;
static hsa_status_t pick_cpu(hsa_agent_t a, void *)
{
    hsa_device_type_t t;
    hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_CPU && !g_have_cpu) { g_cpu = a; g_have_cpu = true; }
    return HSA_STATUS_SUCCESS;
}
static hsa_status_t pick_devpool(hsa_amd_memory_pool_t p, void *)
{
    hsa_amd_segment_t seg;
    uint32_t flags = 0;
    bool alloc = false;
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
    if (seg == HSA_AMD_SEGMENT_GLOBAL && (flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED) && alloc && !g_have_devpool) { g_devpool = p; g_have_devpool = true; }
    return HSA_STATUS_SUCCESS;
}

struct Args { uint8_t *base; uint32_t n_groups; uint32_t key; };
// the argument block of the library's direct-dispatch kernels (gst-plugin-rs_amd/csrc/direct_dispatch.h), for `aql_scope.bin <hsaco> <kernel>.kd`:
// the real hsvfilter kernel through the same packets (timing only: the 20 floats are arbitrary finite values)
struct DirectArgs { uint8_t *frame; uint32_t groups, word3, frame_bytes; int32_t off, bgr; uint32_t reserved; float p[20]; };

int main(int argc, char **argv)
{
    const char *hsaco = argc > 1 ? argv[1] : "tools/probes/aql_scope_kernel.hsaco";
    const char *kname = argc > 2 ? argv[2] : "rmw_kernel.kd";
    const bool real = argc > 2;
    HIP_OK(hipSetDevice(0));
    const uint32_t W = 3840, H = 2160, NF = 16;
    const size_t FB = (size_t)W * H * 4;
    uint8_t *pool = nullptr;
    HIP_OK(hipMalloc(&pool, FB * NF));
    HIP_OK(hipMemset(pool, 0x5a, FB * NF));
    HIP_OK(hipDeviceSynchronize());
    HSA_OK(hsa_init());
    HSA_OK(hsa_iterate_agents(pick_agent, nullptr));
    if (!g_have_gpu) { fprintf(stderr, "no GPU agent\n"); return 2; }
    HSA_OK(hsa_agent_iterate_regions(g_gpu, pick_region, nullptr));
    if (!g_have_kernarg) { // kernarg regions hang off the CPU agent on some stacks: look everywhere
        struct L { static hsa_status_t each(hsa_agent_t a, void *) { hsa_agent_iterate_regions(a, pick_region, nullptr); return HSA_STATUS_SUCCESS; } };
        HSA_OK(hsa_iterate_agents(L::each, nullptr));
    }
    if (!g_have_kernarg) { fprintf(stderr, "no kernarg region\n"); return 2; }
    // code object
    FILE *f = fopen(hsaco, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", hsaco); return 2; }
    std::vector<char> blob;
    { char buf[65536]; size_t n; while ((n = fread(buf, 1, sizeof buf, f)) > 0) blob.insert(blob.end(), buf, buf + n); fclose(f); }
    hsa_code_object_reader_t reader;
    hsa_executable_t exe;
    HSA_OK(hsa_code_object_reader_create_from_memory(blob.data(), blob.size(), &reader));
    HSA_OK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe));
    HSA_OK(hsa_executable_load_agent_code_object(exe, g_gpu, reader, nullptr, nullptr));
    HSA_OK(hsa_executable_freeze(exe, nullptr));
    hsa_executable_symbol_t sym;
    HSA_OK(hsa_executable_get_symbol_by_name(exe, kname, &g_gpu, &sym));
    uint64_t kobj = 0; uint32_t kasize = 0, lds = 0, priv = 0;
    HSA_OK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &kobj));
    HSA_OK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &kasize));
    HSA_OK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &lds));
    HSA_OK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &priv));
    printf("# kernel object %#llx, kernarg %u B, LDS %u, private %u\n", (unsigned long long)kobj, kasize, lds, priv);
    const uint32_t slot = (kasize + 63u) & ~63u, nslots = 4096;
    char *kargs = nullptr;
    // AQL_DEV_KERNARG=1: the argument blocks in DEVICE memory (the GPU's coarse-grained pool, made accessible to the CPU: written through the BAR), as
    // HIP places them (HIP_FORCE_DEV_KERNARG): a wave's first s_load then reads HBM instead of crossing PCIe to host memory
    const bool devk = getenv("AQL_DEV_KERNARG") && atoi(getenv("AQL_DEV_KERNARG"));
    if (devk) {
        HSA_OK(hsa_iterate_agents(pick_cpu, nullptr));
        HSA_OK(hsa_amd_agent_iterate_memory_pools(g_gpu, pick_devpool, nullptr));
        if (!g_have_cpu || !g_have_devpool) { fprintf(stderr, "no CPU agent / device pool\n"); return 2; }
        HSA_OK(hsa_amd_memory_pool_allocate(g_devpool, (size_t)slot * nslots, 0, reinterpret_cast<void **>(&kargs)));
        HSA_OK(hsa_amd_agents_allow_access(1, &g_cpu, nullptr, kargs));
        printf("# kernel arguments in device memory %p (written by the CPU through the BAR)\n", (void *)kargs);
    } else {
        HSA_OK(hsa_memory_allocate(g_kernarg, (size_t)slot * nslots, reinterpret_cast<void **>(&kargs)));
    }
    memset(kargs, 0, (size_t)slot * nslots);
    const uint32_t n_groups = W * H / 4, wgs = (n_groups + 511) / 512;
    for (uint32_t i = 0; i < nslots; i++) {
        if (real) {
            DirectArgs a{};
            a.frame = pool + (size_t)(i % NF) * FB; a.groups = n_groups; a.word3 = 4u | (5u << 3) | (6u << 6) | (10u << 15); a.frame_bytes = (uint32_t)FB;
            for (int k = 0; k < 20; k++) a.p[k] = 0.25f + 0.01f * k;
            if (sizeof a != kasize) { fprintf(stderr, "argument block %zu != kernarg %u\n", sizeof a, kasize); return 2; }
            memcpy(kargs + (size_t)i * slot, &a, sizeof a);
        } else {
            Args a{pool + (size_t)(i % NF) * FB, n_groups, 0x01010101u * (1 + i % 3)};
            memcpy(kargs + (size_t)i * slot, &a, sizeof a);
        }
    }
    if (devk) { __builtin_ia32_sfence(); volatile char sink = kargs[(size_t)slot * nslots - 1]; (void)sink; } // the writes have landed
    hsa_queue_t *q[2];
    for (int k = 0; k < 2; k++) HSA_OK(hsa_queue_create(g_gpu, 4096, HSA_QUEUE_TYPE_MULTI, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &q[k]));
    const int kSig = 64;
    hsa_signal_t sig[kSig];
    for (int k = 0; k < kSig; k++) HSA_OK(hsa_signal_create(1, 0, nullptr, &sig[k]));

    auto run = [&](int acq, int rel, int barrier, int nq, bool sig_every, uint32_t frames) -> double {
        auto submit = [&](uint32_t i, hsa_signal_t s) {
            hsa_queue_t *Q = q[i % nq];
            const uint64_t idx = hsa_queue_add_write_index_relaxed(Q, 1);
            while (idx - hsa_queue_load_read_index_scacquire(Q) >= Q->size) {}
            hsa_kernel_dispatch_packet_t *p = reinterpret_cast<hsa_kernel_dispatch_packet_t *>(Q->base_address) + (idx & (Q->size - 1));
            p->workgroup_size_x = 256; p->workgroup_size_y = 1; p->workgroup_size_z = 1; p->reserved0 = 0;
            p->grid_size_x = wgs * 256; p->grid_size_y = 1; p->grid_size_z = 1;
            p->private_segment_size = priv; p->group_segment_size = lds;
            p->kernel_object = kobj;
            p->kernarg_address = kargs + (size_t)(i % nslots) * slot;
            p->reserved2 = 0;
            p->completion_signal = s;
            const uint16_t header = (uint16_t)((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (barrier << HSA_PACKET_HEADER_BARRIER) |
                                               (acq << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (rel << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
            const uint16_t setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
            __atomic_store_n(reinterpret_cast<uint32_t *>(p), (uint32_t)header | ((uint32_t)setup << 16), __ATOMIC_RELEASE);
            hsa_signal_store_screlease(Q->doorbell_signal, (hsa_signal_value_t)idx);
        };
        // every frame gets a signal of its own when sig_every (what a fence per buffer needs); else only every 64th packet (flow control)
        auto wait_zero = [&](hsa_signal_t s) { while (hsa_signal_wait_scacquire(s, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE) != 0) {} };
        auto pass = [&](uint32_t n) {
            uint32_t marks = 0;
            for (uint32_t i = 0; i < n; i++) {
                hsa_signal_t s{0};
                if (sig_every || (i % 64 == 63) || i + 1 == n) {
                    s = sig[marks++ % kSig];
                    wait_zero(s); // the previous use of this slot has completed (a completion decrements 1 -> 0)
                    hsa_signal_store_relaxed(s, 1);
                }
                submit(i, s);
            }
            for (int k = 0; k < kSig; k++) wait_zero(sig[k]);
        };
        for (int k = 0; k < kSig; k++) hsa_signal_store_relaxed(sig[k], 0);
        pass(600);
        double best = 1e30;
        for (int r = 0; r < 3; r++) {
            const auto t0 = std::chrono::steady_clock::now();
            pass(frames);
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (dt < best) best = dt;
        }
        return best / frames * 1e6;
    };
    const char *names[3] = {"none", "agent", "system"};
    printf("# us per 4K frame (33.2 MB read + 33.2 MB written per dispatch); 11.06 us = 6.0 TB/s\n");
    printf("# %-7s %-7s %-7s %-6s %-10s %s\n", "acquire", "release", "barrier", "queues", "signal", "us/frame");
    const int combos[][2] = {{1, 1}, {1, 0}, {0, 0}};
    for (int nq = 1; nq <= 2; nq++)
        for (int barrier = 1; barrier >= 0; barrier--)
            for (const auto &c : combos)
                for (int se = 0; se < 2; se++) {
                    const double us = run(c[0], c[1], barrier, nq, se != 0, 3000);
                    printf("  %-7s %-7s %-7d %-6d %-10s %.2f\n", names[c[0]], names[c[1]], barrier, nq, se ? "every" : "every 64th", us);
                    fflush(stdout);
                }
    // sanity: every frame was XOR-ed an even or odd number of times with one of three keys; just make sure nothing faulted
    HIP_OK(hipDeviceSynchronize());
    printf("# done\n");
    return 0;
}
