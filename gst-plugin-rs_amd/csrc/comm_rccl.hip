// The one collective on the path, inside the library: videocompare on frames whose rows are distributed over the GPUs of a node
// (rank r holds block-row band r of every pad's frame).  Per aggregate (videocompare/imp.rs:259-389):
//
//   band kernel (blockhash_sums_kernel over this rank's rows, all pads in one launch)
//     -> ncclAllReduce(sum) of n_pads x 64 u32 over xGMI, in place, on the same stream (512 B for a pair: latency-bound)
//     -> one workgroup that derives the 64 hash bits of every pad (gen_hash!: 4 bands of 16 blocks, upper median) and the
//        Hamming distances to the reference pad ON THE DEVICE
//     -> one D2H of (n_pads - 1) x 4 bytes.
//
// RCCL is loaded at run time (dlopen "librccl.so.1"; MVFX_RCCL_LIBRARY overrides the path), like libcairo: a single-GPU user
// of libmi355vfx.so needs no RCCL.  The communicator is created by the library from a 128-byte unique id that rank 0 makes and
// the host layer distributes by whatever it has (the reference's elements would use their own signalling; the bench and tests use
// torch.distributed / a file).  comm == NULL = one GPU holding whole frames: same kernels, no all-reduce.
#include "mvfx_internal.h"

#include <dlfcn.h>

#include <cstring>
#include <mutex>

namespace mvfx {

int blockhash_bands_impl(const mvfx_frame *bands, uint32_t n_pads, uint32_t full_height, uint32_t band_first_row, uint32_t *sums_device,
                         hipStream_t st); // videofx_kernels.hip

namespace {

// ---- the part of the RCCL API this file uses (rccl.h: enum values and signatures of RCCL 2.x / ROCm 7) ---------------------------
struct NcclUniqueId { char internal[128]; };
typedef struct ncclComm *NcclComm;
enum { kNcclSuccess = 0 };
enum { kNcclSum = 0, kNcclMax = 2, kNcclMin = 3 };
enum { kNcclInt32 = 2, kNcclUint32 = 3, kNcclInt64 = 4, kNcclUint64 = 5, kNcclFloat32 = 7, kNcclFloat64 = 8 };

struct Rccl {
    void *handle = nullptr;
    int (*GetUniqueId)(NcclUniqueId *) = nullptr;
    int (*CommInitRank)(NcclComm *, int, NcclUniqueId, int) = nullptr;
    int (*CommDestroy)(NcclComm) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    char error[256] = "";
};

Rccl *rccl()
{
    static Rccl lib;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {getenv("MVFX_RCCL_LIBRARY"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            if (!n || !*n) continue;
            lib.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (lib.handle) break;
            snprintf(lib.error, sizeof(lib.error), "%s", dlerror());
        }
        if (!lib.handle) return;
        lib.GetUniqueId = reinterpret_cast<decltype(lib.GetUniqueId)>(dlsym(lib.handle, "ncclGetUniqueId"));
        lib.CommInitRank = reinterpret_cast<decltype(lib.CommInitRank)>(dlsym(lib.handle, "ncclCommInitRank"));
        lib.CommDestroy = reinterpret_cast<decltype(lib.CommDestroy)>(dlsym(lib.handle, "ncclCommDestroy"));
        lib.AllReduce = reinterpret_cast<decltype(lib.AllReduce)>(dlsym(lib.handle, "ncclAllReduce"));
        lib.GetErrorString = reinterpret_cast<decltype(lib.GetErrorString)>(dlsym(lib.handle, "ncclGetErrorString"));
        if (!lib.GetUniqueId || !lib.CommInitRank || !lib.CommDestroy || !lib.AllReduce) {
            snprintf(lib.error, sizeof(lib.error), "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce");
            lib.handle = nullptr;
        }
    });
    return &lib;
}

int need_rccl(Rccl **out)
{
    Rccl *r = rccl();
    if (!r->handle)
        return fail(MVFX_ERR_IO, "RCCL cannot be loaded (%s); set MVFX_RCCL_LIBRARY", r->error[0] ? r->error : "librccl.so.1 not found");
    *out = r;
    return MVFX_OK;
}

#define MVFX_NCCL_TRY(r, expr)                                                                                                   \
    do {                                                                                                                         \
        const int mvfx_n_ = (expr);                                                                                              \
        if (mvfx_n_ != kNcclSuccess)                                                                                             \
            return fail(MVFX_ERR_DEVICE, "%s failed: %s", #expr, (r)->GetErrorString ? (r)->GetErrorString(mvfx_n_) : "RCCL error"); \
    } while (0)

// One wave per pad: lane i holds block sum i (u32, sizes that are multiples of 8).  gen_hash! of image_hasher's blockhash: the 64
// blocks form 4 bands of 16; the median is the element at index 8 of the sorted band (upper median); bit = v > median or
// (v == median and median > half the largest possible block sum).  Rank by counting: element with exactly 8 smaller-or-tied-earlier
// elements is sorted[8].  Then the Hamming distance of every pad's hash to pad 0's (hashed_image.rs:70).
__global__ __launch_bounds__(1024) void blockhash_bits_distance_kernel(const uint32_t *sums, uint32_t n_pads, uint64_t half_block_value,
                                                                       uint64_t *hashes_out, uint32_t *distances_out)
{
    __shared__ uint64_t s_hash[16];
    const uint32_t pad = threadIdx.x >> 6, lane = threadIdx.x & 63, band = lane >> 4;
    if (pad < n_pads) {
        const uint32_t v = sums[pad * 64 + lane];
        uint32_t rank = 0;
#pragma unroll
        for (uint32_t k = 0; k < 16; k++) {
            const uint32_t other_lane = band * 16 + k;
            const uint32_t o = (uint32_t)__shfl((int)v, (int)other_lane);
            rank += (o < v || (o == v && other_lane < lane)) ? 1u : 0u;
        }
        // the lane whose rank is 8 holds sorted[8] of its band; every lane of the band fetches it
        const uint64_t is_median = __ballot(rank == 8);
        const uint32_t median_lane = (uint32_t)__ffsll((long long)((is_median >> (band * 16)) & 0xffffull)) - 1 + band * 16;
        const uint32_t median = (uint32_t)__shfl((int)v, (int)median_lane);
        const bool bit = v > median || (v == median && (uint64_t)median > half_block_value);
        const uint64_t hash = __ballot(bit);
        if (lane == 0) {
            s_hash[pad] = hash;
            hashes_out[pad] = hash;
        }
    }
    __syncthreads();
    if (threadIdx.x >= 1 && threadIdx.x < n_pads)
        distances_out[threadIdx.x - 1] = (uint32_t)__popcll(s_hash[0] ^ s_hash[threadIdx.x]);
}

constexpr uint32_t kMaxShardedPads = 16;

} // namespace
} // namespace mvfx

using namespace mvfx;

struct mvfx_comm {
    NcclComm comm;
    int rank, world, device;
};

extern "C" {

int mvfx_comm_unique_id(uint8_t id_out[128])
{
    if (!id_out) return fail(MVFX_ERR_INVALID_ARGUMENT, "comm: NULL id");
    Rccl *r = nullptr;
    if (int rc = need_rccl(&r); rc != MVFX_OK) return rc;
    NcclUniqueId id;
    MVFX_NCCL_TRY(r, r->GetUniqueId(&id));
    std::memcpy(id_out, id.internal, sizeof(id.internal));
    return MVFX_OK;
}

int mvfx_comm_create(const uint8_t id[128], int rank, int world, mvfx_comm **out)
{
    if (!id || !out || world < 1 || rank < 0 || rank >= world)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "comm: bad arguments (rank %d of %d)", rank, world);
    if (int rc = require_device(); rc != MVFX_OK) return rc;
    Rccl *r = nullptr;
    if (int rc = need_rccl(&r); rc != MVFX_OK) return rc;
    NcclUniqueId nid;
    std::memcpy(nid.internal, id, sizeof(nid.internal));
    NcclComm c = nullptr;
    MVFX_NCCL_TRY(r, r->CommInitRank(&c, world, nid, rank)); // binds to the calling thread's current device
    mvfx_comm *m = new mvfx_comm;
    m->comm = c;
    m->rank = rank;
    m->world = world;
    (void)hipGetDevice(&m->device);
    *out = m;
    return MVFX_OK;
}

int mvfx_comm_destroy(mvfx_comm *comm)
{
    if (!comm) return MVFX_OK;
    Rccl *r = rccl();
    if (r->handle && comm->comm) (void)r->CommDestroy(comm->comm);
    delete comm;
    return MVFX_OK;
}

int mvfx_comm_rank(const mvfx_comm *comm) { return comm ? comm->rank : 0; }
int mvfx_comm_world(const mvfx_comm *comm) { return comm ? comm->world : 1; }

int mvfx_comm_allreduce(mvfx_comm *comm, void *buffer_device, size_t count, int32_t dtype, int32_t op, mvfx_stream stream)
{
    if (!buffer_device && count) return fail(MVFX_ERR_INVALID_ARGUMENT, "allreduce: NULL buffer");
    int nt, no;
    switch (dtype) {
    case MVFX_DTYPE_U32: nt = kNcclUint32; break;
    case MVFX_DTYPE_U64: nt = kNcclUint64; break;
    case MVFX_DTYPE_F64: nt = kNcclFloat64; break;
    default: return fail(MVFX_ERR_INVALID_ARGUMENT, "allreduce: dtype %d", dtype);
    }
    switch (op) {
    case MVFX_REDUCE_SUM: no = kNcclSum; break;
    case MVFX_REDUCE_MIN: no = kNcclMin; break;
    case MVFX_REDUCE_MAX: no = kNcclMax; break;
    default: return fail(MVFX_ERR_INVALID_ARGUMENT, "allreduce: op %d", op);
    }
    if (!comm || count == 0) return MVFX_OK; // one rank: the buffer already holds the total
    Rccl *r = nullptr;
    if (int rc = need_rccl(&r); rc != MVFX_OK) return rc;
    MVFX_NCCL_TRY(r, r->AllReduce(buffer_device, buffer_device, count, nt, no, comm->comm, as_stream(stream)));
    return MVFX_OK;
}

int mvfx_videocompare_sharded_distances(mvfx_comm *comm, const mvfx_frame *bands, uint32_t n_pads, uint32_t full_height,
                                        uint32_t band_first_row, double *distances_out, uint64_t *hashes_out, mvfx_stream stream)
{
    if (!bands || !distances_out || n_pads < 2)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "videocompare: need the reference pad and at least one other pad");
    if (n_pads > kMaxShardedPads)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "videocompare: at most %u pads per sharded aggregate", kMaxShardedPads);
    if (bands[0].width % 8 != 0 || full_height % 8 != 0)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "videocompare: %ux%u is not a multiple of 8 in both dimensions: image_hasher's f32 path "
                    "sums every block as one ordered chain and cannot be split into row bands", bands[0].width, full_height);
    hipStream_t st = as_stream(stream);
    void *scratch = nullptr;
    // [n_pads x 64 sums][n_pads hashes (u64)][n_pads - 1 distances]
    const size_t sums_bytes = (size_t)n_pads * 64 * sizeof(uint32_t), hash_bytes = (size_t)n_pads * sizeof(uint64_t);
    if (int rc = stream_scratch(st, sums_bytes + hash_bytes + (size_t)n_pads * sizeof(uint32_t), &scratch); rc != MVFX_OK) return rc;
    uint32_t *sums_dev = static_cast<uint32_t *>(scratch);
    uint64_t *hash_dev = reinterpret_cast<uint64_t *>(static_cast<uint8_t *>(scratch) + sums_bytes);
    uint32_t *dist_dev = reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(scratch) + sums_bytes + hash_bytes);
    if (int rc = blockhash_bands_impl(bands, n_pads, full_height, band_first_row, sums_dev, st); rc != MVFX_OK) return rc;
    // the totals of a block never exceed 765 x (w/8) x (h/8) < 2^32 for every frame a u32 block sum can describe at all
    if (int rc = mvfx_comm_allreduce(comm, sums_dev, (size_t)n_pads * 64, MVFX_DTYPE_U32, MVFX_REDUCE_SUM, stream); rc != MVFX_OK) return rc;
    const uint64_t half_block_value = (uint64_t)765 * (bands[0].width / 8) * (full_height / 8) / 2;
    MVFX_LAUNCH(blockhash_bits_distance_kernel, dim3(1), dim3(64 * n_pads), 0, st, sums_dev, n_pads, half_block_value, hash_dev, dist_dev);
    MVFX_HIP_TRY(hipGetLastError());
    uint32_t dist[kMaxShardedPads] = {};
    uint64_t hashes[kMaxShardedPads] = {};
    MVFX_HIP_TRY(hipMemcpyAsync(dist, dist_dev, (size_t)(n_pads - 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    if (hashes_out)
        MVFX_HIP_TRY(hipMemcpyAsync(hashes, hash_dev, hash_bytes, hipMemcpyDeviceToHost, st));
    MVFX_HIP_TRY(hipStreamSynchronize(st));
    for (uint32_t p = 0; p + 1 < n_pads; p++) distances_out[p] = (double)dist[p]; // hashed_image.rs:70 `left.dist(right) as f64`
    if (hashes_out)
        for (uint32_t p = 0; p < n_pads; p++) hashes_out[p] = hashes[p];
    return MVFX_OK;
}

int mvfx_videocompare_sharded_dssim(mvfx_comm *comm, const mvfx_frame *reference_frame, const mvfx_frame *other_frame,
                                    uint32_t row_begin, uint32_t row_end, double *distance_out, mvfx_stream stream)
{
    if (!reference_frame || !other_frame || !distance_out)
        return fail(MVFX_ERR_INVALID_ARGUMENT, "videocompare dssim: NULL frame or output");
    hipStream_t st = as_stream(stream);
    double sums[5] = {}, counts[5] = {};
    uint32_t n_scales = 0;
    if (int rc = mvfx_ssim_partial_sums(reference_frame, other_frame, row_begin, row_end, sums, counts, &n_scales, stream); rc != MVFX_OK) return rc;
    double tot[10];
    for (int i = 0; i < 5; i++) { tot[i] = sums[i]; tot[5 + i] = counts[i]; }
    void *scratch = nullptr;
    if (comm) { // 80 + 40 bytes through the collective; with one rank the band's sums are the totals
        if (int rc = stream_scratch(st, 16 * sizeof(double), &scratch); rc != MVFX_OK) return rc;
        MVFX_HIP_TRY(hipMemcpyAsync(scratch, tot, sizeof(tot), hipMemcpyHostToDevice, st));
        if (int rc = mvfx_comm_allreduce(comm, scratch, 10, MVFX_DTYPE_F64, MVFX_REDUCE_SUM, stream); rc != MVFX_OK) return rc;
        MVFX_HIP_TRY(hipMemcpyAsync(tot, scratch, sizeof(tot), hipMemcpyDeviceToHost, st));
        MVFX_HIP_TRY(hipStreamSynchronize(st));
    }
    double mean[5], dev[5] = {};
    for (int i = 0; i < 5; i++) mean[i] = tot[5 + i] != 0.0 ? tot[i] / tot[5 + i] : 0.0;
    if (int rc = mvfx_ssim_partial_deviation(mean, dev, stream); rc != MVFX_OK) return rc;
    if (comm) {
        MVFX_HIP_TRY(hipMemcpyAsync(scratch, dev, sizeof(dev), hipMemcpyHostToDevice, st));
        if (int rc = mvfx_comm_allreduce(comm, scratch, 5, MVFX_DTYPE_F64, MVFX_REDUCE_SUM, stream); rc != MVFX_OK) return rc;
        MVFX_HIP_TRY(hipMemcpyAsync(dev, scratch, sizeof(dev), hipMemcpyDeviceToHost, st));
        MVFX_HIP_TRY(hipStreamSynchronize(st));
    }
    double mad[5];
    for (int i = 0; i < 5; i++) mad[i] = tot[5 + i] != 0.0 ? dev[i] / tot[5 + i] : 0.0;
    *distance_out = mvfx_ssim_combine(mean, mad, n_scales);
    return MVFX_OK;
}

} // extern "C"
