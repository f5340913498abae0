// The per-frame schedule and the slab exchange behind the C ABI (include/tsdf_amd.h: tsdf_pipeline_*, tsdf_slab_exchange_*).
//
// tsdf_pipeline: the step of BASELINE configs[2] -- bilateral filter, integrate, ray cast + normals, frame after frame as the
// reference's kinfu loop integrates them (src/Tools/kinfu.cpp:32-56: one blocking call per frame on host buffers; its ray cast
// comes once at the end, :181) -- on frames that live in HBM, on two HIP streams: the step's own, and a second one of lower
// priority for the pieces of the NEXT frame that depend on nothing before them (its filter, and its brick culling when its pose
// is known already).  Every kernel of the step fills the chip while it is in full swing, but each ends with a ramp-down (integrate
// ~10 us, the two ray kernels ~15 and ~20 us) and the small kernels between them (reach summary, resolve + normals) are chains of
// memory round trips on a few thousand waves: a kernel of EQUAL priority beside them only takes turns with them (measured: no
// gain), one of LOWER priority gets the slots the step cannot use at that moment (0.345 -> 0.325 ms per step).
// The next frame's work is released by THIS frame's integrate (an event), so it never runs beside the memory-bound
// integrate_kernel, and the next step waits for it (another event).  Results cannot change: the same kernels run on the same
// inputs, only earlier.  (Up to round 2 this schedule lived in Python, tsdf_amd/pipeline.py on torch streams; that file is now a
// ctypes wrapper of these entry points, and tools/kinfu_stream.cpp drives them from C++.)
//
// tsdf_slab_exchange: the one collective of a sharded frame -- ncclAllGather of the ranks' 8-byte hit records -- enqueued by
// RCCL on the caller's HIP stream: one more launch between the slab ray cast and the merge kernel, no event, no second stream
// (torch.distributed's all_gather_into_tensor runs on the process group's own stream and hands over with events on either side:
// 0.2-1.2 ms between kernels of a compute stream on an MI355X box against 0.02 ms this way).  librccl is opened at
// run time (dlopen), so that a process that already holds one -- torch ships its own -- uses that one; the 128-byte unique id is
// made by rank 0 and travels to the other ranks by whatever means the caller has (torch.distributed, MPI, a file).
#include <dlfcn.h>

#include <new>

#include <cstdlib>
#include <cstring>

#include <algorithm>
#include <vector>

#include "common.hpp"

// ---- RCCL, by hand: only the five entry points used, resolved with dlsym (rccl.h: NCCL_UNIQUE_ID_BYTES = 128, ncclUint8 = 1)
namespace {
struct NcclUniqueId {
    char internal[128];
};
typedef int (*nccl_get_unique_id_fn)(NcclUniqueId *);
typedef int (*nccl_comm_init_rank_fn)(void **comm, int nranks, NcclUniqueId id, int rank);
typedef int (*nccl_all_gather_fn)(const void *send, void *recv, size_t count, int dtype, void *comm, hipStream_t stream);
typedef int (*nccl_comm_destroy_fn)(void *comm);
typedef const char *(*nccl_get_error_string_fn)(int);
typedef int (*nccl_comm_count_fn)(const void *comm, int *count);
constexpr int kNcclUint8 = 1;

struct Rccl {
    void *dl = nullptr;
    nccl_get_unique_id_fn get_unique_id = nullptr;
    nccl_comm_init_rank_fn comm_init_rank = nullptr;
    nccl_all_gather_fn all_gather = nullptr;
    nccl_comm_destroy_fn comm_destroy = nullptr;
    nccl_get_error_string_fn error_string = nullptr;
    nccl_comm_count_fn comm_count = nullptr;   // (optional: what the communicator itself says its size is)
};

// library == nullptr: the librccl this process holds already if any, else the one of the ROCm installation
int open_rccl(const char *library, Rccl &r) {
    const char *fallbacks[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    if (library && *library) {
        r.dl = dlopen(library, RTLD_NOW | RTLD_LOCAL);
    } else {
        for (const char *f : fallbacks)
            if (!r.dl) r.dl = dlopen(f, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
        for (const char *f : fallbacks)
            if (!r.dl) r.dl = dlopen(f, RTLD_NOW | RTLD_LOCAL);
    }
    if (!r.dl) {
        tsdf::set_error("slab exchange: cannot open librccl (%s)", dlerror());
        return TSDF_ERR_DEVICE;
    }
    r.get_unique_id = (nccl_get_unique_id_fn)dlsym(r.dl, "ncclGetUniqueId");
    r.comm_init_rank = (nccl_comm_init_rank_fn)dlsym(r.dl, "ncclCommInitRank");
    r.all_gather = (nccl_all_gather_fn)dlsym(r.dl, "ncclAllGather");
    r.comm_destroy = (nccl_comm_destroy_fn)dlsym(r.dl, "ncclCommDestroy");
    r.error_string = (nccl_get_error_string_fn)dlsym(r.dl, "ncclGetErrorString");
    r.comm_count = (nccl_comm_count_fn)dlsym(r.dl, "ncclCommCount");
    if (!r.get_unique_id || !r.comm_init_rank || !r.all_gather || !r.comm_destroy || !r.error_string) {
        tsdf::set_error("slab exchange: librccl lacks a symbol (ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy / ncclGetErrorString)");
        dlclose(r.dl);
        r.dl = nullptr;
        return TSDF_ERR_DEVICE;
    }
    return TSDF_OK;
}

int rccl_fail(const Rccl &r, int rc, const char *what) {
    tsdf::set_error("%s failed: %s", what, r.error_string ? r.error_string(rc) : "?");
    return TSDF_ERR_DEVICE;
}
}  // namespace

struct tsdf_slab_exchange {
    Rccl rccl;
    void *comm;
    int rank, world;
    tsdf_exchange_fn callback;   // != nullptr: the caller's own collective instead of RCCL
    void *user;
};

struct tsdf_pipeline {
    tsdf_volume *volume;
    const tsdf_bilateral *filter;
    tsdf_slab_exchange *exchange;   // nullptr: a whole volume
    uint32_t width, height;
    int flags;
    hipStream_t main, side, xstream;
    hipStream_t volume_stream_before;
    uint16_t *filtered[2], *tile_max[2];
    // events are made once and recorded again every other frame (creating one per frame makes the runtime grow its pool of
    // signals now and then: a stall of tens of milliseconds in the middle of a stream)
    hipEvent_t done[2], ready[2];   // [b]: the integrate that read buffer b has finished; buffer b has been filtered (and culled) ahead
    hipEvent_t bulk;                // the bulk ray kernel of this frame's cast has ended (TSDF_PIPE_RELEASE=1)
    uint32_t *release_word;         // signal memory: the cell-parallel cast's first kernel stores release_seq there (see tsdf_pipeline_step); may be null
    uint32_t release_seq;
    hipEvent_t cast, merged;        // third-stream exchange: the slab cast has left its records; the merge has consumed them
    bool merged_pending;
    const uint16_t *ahead_depth;    // the frame filtered ahead into buffer ahead_buf (nullptr: none)
    int ahead_buf;
    bool ahead_culled;
    uint64_t frames;
    tsdf_hit_record *hits_mine, *hits_all;
};

// tsdf_tracker: the closed loop of BASELINE configs[4], frame-to-model.  The reference has the two halves -- src/Tools/kinfu.cpp
// integrates a sequence with given poses, src/Tools/tsdf_icp.cpp (:115-198) renders a volume to a depth image and aligns one frame to it
// with ICPOdometry (third_party/ICP_CUDA/ICPOdometry.cpp:97-136) -- this composes them per frame on two streams: the new frame's
// bilateral filter and its ICP pyramid / vertex / normal maps depend on nothing in the volume and run on the lower-priority stream
// beside the ray cast that renders the model from the previous pose; the model's maps, the 19 ICP iterations and the integrate follow
// on the step's stream.  The pose is composed by the caller (the Camera class lives in the host library): align() blocks for the
// 4 x 4 result, integrate() is asynchronous.
struct tsdf_tracker {
    tsdf_volume *volume;
    const tsdf_bilateral *filter;
    tsdf_icp *icp;
    uint32_t width, height;
    float depth_cutoff;
    hipStream_t main, side;
    hipStream_t volume_stream_before, icp_stream_before;
    uint16_t *filtered[2], *tile_max[2], *model;
    hipEvent_t integrated[2];   // [b]: the integrate that read buffer b is done
    hipEvent_t ready;           // the frame in buffer `cur` has been filtered (and its ICP maps built)
    bool ready_pending;         // ... and the step's stream has not waited for that yet
    int cur;
    bool have_frame;
    uint64_t frames;            // frames integrated
};

using namespace tsdf;

static int run_filter(tsdf_pipeline *p, const uint16_t *depth, int b, hipStream_t s) {
    return tsdf_bilateral_filter_u16_device_tiles(p->filter, depth, p->filtered[b], (int)p->width, (int)p->height, p->tile_max[b], s);
}

extern "C" {

int tsdf_slab_exchange_unique_id(uint8_t id[TSDF_EXCHANGE_ID_BYTES], const char *rccl_library) {
    TSDF_REQUIRE(id, "tsdf_slab_exchange_unique_id: null argument");
    Rccl r;
    int rc = open_rccl(rccl_library, r);
    if (rc != TSDF_OK) return rc;
    NcclUniqueId u;
    memset(&u, 0, sizeof(u));
    const int e = r.get_unique_id(&u);
    if (e != 0) return rccl_fail(r, e, "ncclGetUniqueId");
    static_assert(sizeof(u) == TSDF_EXCHANGE_ID_BYTES, "unique id size");
    memcpy(id, &u, sizeof(u));
    return TSDF_OK;   // (the library stays open: the communicator that follows uses it)
}

int tsdf_slab_exchange_create(int rank, int world, const uint8_t id[TSDF_EXCHANGE_ID_BYTES], const char *rccl_library,
                              tsdf_slab_exchange **out) {
    TSDF_REQUIRE(out && id, "tsdf_slab_exchange_create: null argument");
    *out = nullptr;
    TSDF_REQUIRE(world >= 1 && rank >= 0 && rank < world, "tsdf_slab_exchange_create: rank %d outside a world of %d", rank, world);
    tsdf_slab_exchange *x = new (std::nothrow) tsdf_slab_exchange();
    if (!x) {
        set_error("out of host memory");
        return TSDF_ERR_NOMEM;
    }
    x->comm = nullptr;
    x->rank = rank;
    x->world = world;
    x->callback = nullptr;
    x->user = nullptr;
    int rc = open_rccl(rccl_library, x->rccl);
    if (rc != TSDF_OK) {
        delete x;
        return rc;
    }
    NcclUniqueId u;
    memcpy(&u, id, sizeof(u));
    const int e = x->rccl.comm_init_rank(&x->comm, world, u, rank);   // (collective: every rank of the world calls it)
    if (e != 0) {
        rc = rccl_fail(x->rccl, e, "ncclCommInitRank");
        delete x;
        return rc;
    }
    *out = x;
    return TSDF_OK;
}

int tsdf_slab_exchange_create_callback(int rank, int world, tsdf_exchange_fn all_gather, void *user, tsdf_slab_exchange **out) {
    TSDF_REQUIRE(out && all_gather, "tsdf_slab_exchange_create_callback: null argument");
    *out = nullptr;
    TSDF_REQUIRE(world >= 1 && rank >= 0 && rank < world, "tsdf_slab_exchange_create_callback: rank %d outside a world of %d", rank, world);
    tsdf_slab_exchange *x = new (std::nothrow) tsdf_slab_exchange();
    if (!x) {
        set_error("out of host memory");
        return TSDF_ERR_NOMEM;
    }
    x->comm = nullptr;
    x->rank = rank;
    x->world = world;
    x->callback = all_gather;
    x->user = user;
    *out = x;
    return TSDF_OK;
}

static int loopback_all_gather(void *user, const tsdf_hit_record *device_mine, tsdf_hit_record *device_all, uint32_t n_pixels, void *hip_stream) {
    const tsdf_slab_exchange *x = static_cast<const tsdf_slab_exchange *>(user);
    for (int r = 0; r < x->world; r++)
        if (hipMemcpyAsync(device_all + (size_t)r * n_pixels, device_mine, (size_t)n_pixels * sizeof(tsdf_hit_record), hipMemcpyDeviceToDevice,
                           (hipStream_t)hip_stream) != hipSuccess)
            return TSDF_ERR_DEVICE;
    return TSDF_OK;
}

int tsdf_slab_exchange_create_loopback(int rank, int world, tsdf_slab_exchange **out) {
    const int rc = tsdf_slab_exchange_create_callback(rank, world, loopback_all_gather, nullptr, out);
    if (rc == TSDF_OK) (*out)->user = *out;
    return rc;
}

int tsdf_slab_exchange_world(const tsdf_slab_exchange *x, int *rank, int *world) {
    TSDF_REQUIRE(x, "null exchange");
    if (rank) *rank = x->rank;
    if (world) *world = x->world;
    return TSDF_OK;
}

int tsdf_slab_exchange_ranks_seen(const tsdf_slab_exchange *x, int *ranks) {
    TSDF_REQUIRE(x && ranks, "null argument");
    *ranks = x->world;
    if (!x->callback && x->comm && x->rccl.comm_count) {
        int n = 0;
        const int e = x->rccl.comm_count(x->comm, &n);
        if (e != 0) return rccl_fail(x->rccl, e, "ncclCommCount");
        *ranks = n;
    }
    return TSDF_OK;
}

namespace {
// words of a and b that differ (a NaN equals a NaN)
__global__ __launch_bounds__(256) void count_differing_words_kernel(const float *__restrict__ a, const float *__restrict__ b, size_t n,
                                                                    unsigned long long *__restrict__ count) {
    unsigned long long c = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float x = a[i], y = b[i];
        c += (__float_as_uint(x) != __float_as_uint(y) && !(x != x && y != y)) ? 1ull : 0ull;
    }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63u) == 0 && c) atomicAdd(count, c);
}
}  // namespace

// SURVEY.md 8e, mode B: the cross-rank validator of the merge path.  Every rank gathers every rank's distance slab (its own planes,
// padded to the longest slab; the same collective as the frame's, by the byte), assembles the whole volume, casts it the ordinary
// single-volume way and counts the words in which that picture differs from the merged one it was handed.  Collective: every rank of
// the exchange calls it with its own slab.  Expensive by design (4 N bytes per rank at the node's link rate, a whole volume resident per
// rank): for validation, never inside a timed step.
int tsdf_slab_validate_merge(tsdf_volume *slab, tsdf_slab_exchange *x, uint32_t width, uint32_t height, const float pose[16],
                             const float kinv[9], const float *device_merged_vertices, const float *device_merged_normals,
                             uint64_t *differing_words) {
    TSDF_REQUIRE(slab && x && pose && kinv && device_merged_vertices && differing_words, "tsdf_slab_validate_merge: null argument");
    *differing_words = ~0ull;
    const Geom &g = slab->g;
    const size_t xy = (size_t)g.X * g.Y;
    hipStream_t s = slab->stream;
    int rc = occupancy_join(slab);
    if (rc != TSDF_OK) return rc;
    // (1) everybody's plane range
    tsdf_hit_record *ranges_dev = nullptr;
    std::vector<tsdf_hit_record> ranges((size_t)x->world);
    TSDF_HIP(hipMalloc((void **)&ranges_dev, ((size_t)x->world + 1) * sizeof(tsdf_hit_record)), "validate merge: alloc");
    const uint32_t mine[2] = {slab->z_begin, slab->z_end};
    static_assert(sizeof(tsdf_hit_record) == 8, "a record is two words");
    hipError_t e = hipMemcpyAsync(ranges_dev + x->world, mine, sizeof(mine), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) rc = tsdf_slab_exchange_all_gather(x, ranges_dev + x->world, ranges_dev, 1, s);
    if (e == hipSuccess && rc == TSDF_OK) e = hipMemcpyAsync(ranges.data(), ranges_dev, ranges.size() * sizeof(tsdf_hit_record), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && rc == TSDF_OK) e = hipStreamSynchronize(s);
    (void)hipFree(ranges_dev);
    if (rc != TSDF_OK) return rc;
    if (e != hipSuccess) return hip_fail(e, "validate merge: plane ranges");
    uint32_t longest = 0, covered = 0;
    for (int r = 0; r < x->world; r++) {
        uint32_t zb, ze;
        memcpy(&zb, reinterpret_cast<const char *>(&ranges[r]), 4);
        memcpy(&ze, reinterpret_cast<const char *>(&ranges[r]) + 4, 4);
        TSDF_REQUIRE(zb == covered && ze > zb && ze <= g.Z, "tsdf_slab_validate_merge: the ranks' slabs are not consecutive plane ranges of the grid");
        covered = ze;
        longest = std::max(longest, ze - zb);
    }
    TSDF_REQUIRE(covered == g.Z, "tsdf_slab_validate_merge: the ranks' slabs do not cover the grid");
    // the slabs travel by the byte: records of 8 bytes, rank r's at r * per_rank
    const size_t per_rank = ((size_t)longest * xy + 1) / 2;   // records per rank
    TSDF_REQUIRE(per_rank <= 0xffffffffull, "tsdf_slab_validate_merge: slab too long for one collective");
    tsdf_hit_record *send = nullptr, *recv = nullptr;
    tsdf_volume *whole = nullptr;
    float *V = nullptr, *N = nullptr;
    unsigned long long *count_dev = nullptr;
    auto cleanup = [&]() {
        if (send) (void)hipFree(send);
        if (recv) (void)hipFree(recv);
        if (V) (void)hipFree(V);
        if (N) (void)hipFree(N);
        if (count_dev) (void)hipFree(count_dev);
        if (whole) (void)tsdf_volume_destroy(whole);
    };
    // Everything that can fail on ONE rank -- the staging buffers, the whole volume, the pictures -- is allocated before the large
    // collective, and the ranks tell each other how that went (one record a rank): a rank that ran out of memory must not leave the
    // others waiting in ncclAllGather for a partner that has returned.  (The checks above are on gathered data: every rank takes the
    // same way out.  What cannot be helped: a rank that fails before the FIRST collective -- 8 (world + 1) bytes of device memory, a
    // dead stream -- leaves the others in it.)
    const size_t words = (size_t)width * height * 3;
    tsdf_hit_record *status_dev = nullptr;
    std::vector<tsdf_hit_record> status((size_t)x->world);
    e = hipMalloc((void **)&status_dev, ((size_t)x->world + 1) * sizeof(tsdf_hit_record));
    if (e != hipSuccess) return hip_fail(e, "validate merge: alloc");   // (see above: nothing to tell the others with)
    int local_rc = TSDF_OK;
    hipError_t le = hipMalloc((void **)&send, per_rank * sizeof(tsdf_hit_record));
    if (le == hipSuccess) le = hipMalloc((void **)&recv, per_rank * (size_t)x->world * sizeof(tsdf_hit_record));
    if (le == hipSuccess) le = hipMalloc((void **)&V, words * sizeof(float));
    if (le == hipSuccess) le = hipMalloc((void **)&N, words * sizeof(float));
    if (le == hipSuccess) le = hipMalloc((void **)&count_dev, sizeof(unsigned long long));
    if (le == hipSuccess) le = hipMemsetAsync(count_dev, 0, sizeof(unsigned long long), s);
    if (le == hipSuccess) le = hipMemsetAsync(send, 0, per_rank * sizeof(tsdf_hit_record), s);
    if (le == hipSuccess)
        le = hipMemcpyAsync(send, slab->dist + (size_t)(slab->z_begin - g.z_store_begin) * xy, (size_t)(slab->z_end - slab->z_begin) * xy * sizeof(float),
                            hipMemcpyDeviceToDevice, s);
    if (le != hipSuccess) local_rc = hip_fail(le, "validate merge: staging");
    // the whole volume with this slab's header
    if (local_rc == TSDF_OK) local_rc = tsdf_volume_create(g.X, g.Y, g.Z, g.phys.x, g.phys.y, g.phys.z, &whole);
    if (local_rc == TSDF_OK) {
        const float off[3] = {g.offset.x, g.offset.y, g.offset.z};
        local_rc = tsdf_volume_set_header(whole, off, g.trunc, slab->max_weight, slab->global_translation, slab->global_rotation);
    }
    if (local_rc == TSDF_OK) local_rc = tsdf_volume_set_stream(whole, s);
    {
        const uint32_t word[2] = {local_rc == TSDF_OK ? 0u : 1u, 0u};
        e = hipMemcpyAsync(status_dev + x->world, word, sizeof(word), hipMemcpyHostToDevice, s);
        if (e == hipSuccess) rc = tsdf_slab_exchange_all_gather(x, status_dev + x->world, status_dev, 1, s);
        if (e == hipSuccess && rc == TSDF_OK) e = hipMemcpyAsync(status.data(), status_dev, status.size() * sizeof(tsdf_hit_record), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess && rc == TSDF_OK) e = hipStreamSynchronize(s);
        (void)hipFree(status_dev);
        if (rc != TSDF_OK) { cleanup(); return rc; }
        if (e != hipSuccess) { cleanup(); return hip_fail(e, "validate merge: status"); }
        for (int r = 0; r < x->world; r++) {
            uint32_t failed;
            memcpy(&failed, reinterpret_cast<const char *>(&status[r]), 4);
            if (failed) {
                cleanup();
                if (local_rc != TSDF_OK) return local_rc;   // (this rank's own error, message set)
                set_error("tsdf_slab_validate_merge: rank %d could not allocate its buffers; every rank returns", r);
                return TSDF_ERR_DEVICE;
            }
        }
    }
    // (2) the slabs themselves, (3) assembled
    rc = tsdf_slab_exchange_all_gather(x, send, recv, (uint32_t)per_rank, s);
    if (rc != TSDF_OK) { cleanup(); return rc; }
    for (int r = 0; r < x->world && e == hipSuccess; r++) {
        uint32_t zb, ze;
        memcpy(&zb, reinterpret_cast<const char *>(&ranges[r]), 4);
        memcpy(&ze, reinterpret_cast<const char *>(&ranges[r]) + 4, 4);
        e = hipMemcpyAsync(whole->dist + (size_t)zb * xy, reinterpret_cast<const float *>(recv + (size_t)r * per_rank), (size_t)(ze - zb) * xy * sizeof(float),
                           hipMemcpyDeviceToDevice, s);
    }
    if (e != hipSuccess) { cleanup(); return hip_fail(e, "validate merge: assembling the volume"); }
    rc = tsdf_volume_mark_dirty(whole);   // (the distances were written through the pointer)
    if (rc == TSDF_OK) rc = tsdf_raycast_device(whole, width, height, pose, kinv, V, device_merged_normals ? N : nullptr);
    if (rc != TSDF_OK) { cleanup(); return rc; }
    hipLaunchKernelGGL(count_differing_words_kernel, dim3(256), dim3(256), 0, s, V, device_merged_vertices, words, count_dev);
    if (device_merged_normals) hipLaunchKernelGGL(count_differing_words_kernel, dim3(256), dim3(256), 0, s, N, device_merged_normals, words, count_dev);
    unsigned long long n_diff = 0;
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(&n_diff, count_dev, sizeof(n_diff), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    cleanup();
    if (e != hipSuccess) return hip_fail(e, "validate merge: comparing the pictures");
    *differing_words = n_diff;
    return TSDF_OK;
}

int tsdf_slab_exchange_all_gather(tsdf_slab_exchange *x, const tsdf_hit_record *device_mine, tsdf_hit_record *device_all,
                                  uint32_t n_pixels, void *hip_stream) {
    TSDF_REQUIRE(x && device_mine && device_all && n_pixels > 0, "tsdf_slab_exchange_all_gather: bad argument");
    if (x->callback) {
        const int rc = x->callback(x->user, device_mine, device_all, n_pixels, hip_stream);
        if (rc != TSDF_OK) {
            set_error("slab exchange: the caller's all-gather returned %d", rc);
            return TSDF_ERR_DEVICE;
        }
        return TSDF_OK;
    }
    const int e = x->rccl.all_gather(device_mine, device_all, (size_t)n_pixels * sizeof(tsdf_hit_record), kNcclUint8, x->comm,
                                     (hipStream_t)hip_stream);
    if (e != 0) return rccl_fail(x->rccl, e, "ncclAllGather");
    return TSDF_OK;
}

int tsdf_slab_exchange_destroy(tsdf_slab_exchange *x) {
    if (!x) return TSDF_OK;
    if (x->comm && x->rccl.comm_destroy) (void)x->rccl.comm_destroy(x->comm);
    // (the library handle is left open: RCCL keeps threads and device state of its own until the process ends)
    delete x;
    return TSDF_OK;
}

int tsdf_pipeline_destroy(tsdf_pipeline *p) {
    if (!p) return TSDF_OK;
    if (p->main) (void)hipStreamSynchronize(p->main);
    if (p->side) (void)hipStreamSynchronize(p->side);
    if (p->xstream) (void)hipStreamSynchronize(p->xstream);
    if (p->volume && p->volume->attached == p) {   // (this pipeline did put the volume on its stream)
        (void)tsdf_integrate_discard_prepared(p->volume);
        (void)tsdf_volume_set_stream(p->volume, p->volume_stream_before);
        p->volume->attached = nullptr;
    }
    for (int b = 0; b < 2; b++) {
        if (p->filtered[b]) (void)hipFree(p->filtered[b]);
        if (p->tile_max[b]) (void)hipFree(p->tile_max[b]);
        if (p->done[b]) (void)hipEventDestroy(p->done[b]);
        if (p->ready[b]) (void)hipEventDestroy(p->ready[b]);
    }
    if (p->bulk) (void)hipEventDestroy(p->bulk);
    if (p->cast) (void)hipEventDestroy(p->cast);
    if (p->merged) (void)hipEventDestroy(p->merged);
    if (p->hits_mine) (void)hipFree(p->hits_mine);
    if (p->hits_all) (void)hipFree(p->hits_all);
    if (p->release_word) (void)hipFree(p->release_word);
    if (p->side) (void)hipStreamDestroy(p->side);
    if (p->xstream) (void)hipStreamDestroy(p->xstream);
    if (p->main) (void)hipStreamDestroy(p->main);
    delete p;
    return TSDF_OK;
}

int tsdf_pipeline_create(tsdf_volume *volume, const tsdf_bilateral *filter, uint32_t width, uint32_t height, int flags,
                         tsdf_slab_exchange *exchange, tsdf_pipeline **out) {
    TSDF_REQUIRE(out, "tsdf_pipeline_create: null out pointer");
    *out = nullptr;
    TSDF_REQUIRE(volume && filter && width > 0 && height > 0 && width <= 65535 && height <= 65535, "tsdf_pipeline_create: bad argument");
    const bool slab = volume->z_begin != 0 || volume->z_end != volume->g.Z;
    TSDF_REQUIRE(exchange || !slab, "tsdf_pipeline_create: a Z-slab needs a slab exchange");
    // one pipeline or tracker per volume at a time: each puts the volume on its own stream and restores the previous one when it
    // goes -- two of them would restore each other's (destroyed) streams
    TSDF_REQUIRE(!volume->attached, "tsdf_pipeline_create: the volume is attached to another pipeline or tracker (destroy that one first)");
    tsdf_pipeline *p = new (std::nothrow) tsdf_pipeline();
    if (!p) {
        set_error("out of host memory");
        return TSDF_ERR_NOMEM;
    }
    memset(p, 0, sizeof(*p));
    p->volume = volume;
    p->filter = filter;
    p->exchange = exchange;
    p->width = width;
    p->height = height;
    p->flags = flags;
    p->volume_stream_before = volume->stream;
    // Priorities: the step's stream above the side stream.  With a slab exchange the two are EQUAL: ncclAllGather enqueued on a
    // stream of raised priority made the one-rank step 0.58 ms instead of 0.37 (RCCL 2.26.6), and equal priorities still give
    // 0.366 against 0.381 without the second stream.
    int least = 0, greatest = 0;
    hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);   // (numerically: greatest <= least)
    const bool overlap = (flags & TSDF_PIPELINE_OVERLAP) != 0;
    const bool equal = (flags & TSDF_PIPELINE_EQUAL_PRIORITY) != 0 || exchange != nullptr || !overlap;
    const int normal = (greatest <= 0 && 0 <= least) ? 0 : least;
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&p->main, hipStreamNonBlocking, equal ? normal : greatest);
    if (e == hipSuccess && overlap) e = hipStreamCreateWithPriority(&p->side, hipStreamNonBlocking, normal);
    if (e == hipSuccess && exchange && (flags & TSDF_PIPELINE_EXCHANGE_STREAM)) e = hipStreamCreateWithPriority(&p->xstream, hipStreamNonBlocking, normal);
    const size_t n = (size_t)width * height;
    const size_t tiles = (size_t)((width + TSDF_DEPTH_TILE - 1) / TSDF_DEPTH_TILE) * ((height + TSDF_DEPTH_TILE - 1) / TSDF_DEPTH_TILE);
    for (int b = 0; b < 2 && e == hipSuccess; b++) {
        e = hipMalloc((void **)&p->filtered[b], n * sizeof(uint16_t));
        if (e == hipSuccess) e = hipMalloc((void **)&p->tile_max[b], tiles * sizeof(uint16_t));
        if (e == hipSuccess) e = hipEventCreateWithFlags(&p->done[b], stream_order_event_flags());
        if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ready[b], stream_order_event_flags());
    }
    if (e == hipSuccess && overlap) e = hipEventCreateWithFlags(&p->bulk, stream_order_event_flags());
    if (e == hipSuccess && overlap && !exchange && tuning().pipe_word_release) {
        // (a word the second stream can wait for without an event in the step's stream; where the runtime has no such memory the events do it)
        if (hipExtMallocWithFlags((void **)&p->release_word, 2 * sizeof(uint32_t), hipMallocSignalMemory) != hipSuccess) {
            (void)hipGetLastError();
            p->release_word = nullptr;
        } else if (hipMemset(p->release_word, 0, 2 * sizeof(uint32_t)) != hipSuccess) {
            (void)hipFree(p->release_word);
            p->release_word = nullptr;
        }
    }
    if (e == hipSuccess && p->xstream) e = hipEventCreateWithFlags(&p->cast, hipEventDisableTiming);
    if (e == hipSuccess && p->xstream) e = hipEventCreateWithFlags(&p->merged, hipEventDisableTiming);
    if (e == hipSuccess && exchange) {
        e = hipMalloc((void **)&p->hits_mine, n * sizeof(tsdf_hit_record));
        if (e == hipSuccess) e = hipMalloc((void **)&p->hits_all, n * sizeof(tsdf_hit_record) * (size_t)exchange->world);
    }
    if (e != hipSuccess) {
        const int rc = hip_fail(e, "tsdf_pipeline_create");
        tsdf_pipeline_destroy(p);   // (the volume is not attached yet: its stream stays)
        return rc;
    }
    // whatever the volume has in flight on its previous stream comes before the pipeline's first launch
    (void)hipStreamSynchronize(volume->stream);
    (void)tsdf_volume_set_stream(volume, p->main);
    volume->attached = p;
    *out = p;
    return TSDF_OK;
}

int tsdf_pipeline_streams(const tsdf_pipeline *p, void **main_stream, void **side_stream) {
    TSDF_REQUIRE(p, "null pipeline");
    if (main_stream) *main_stream = p->main;
    if (side_stream) *side_stream = p->side;
    return TSDF_OK;
}

int tsdf_pipeline_hit_buffers(const tsdf_pipeline *p, tsdf_hit_record **device_mine, tsdf_hit_record **device_all) {
    TSDF_REQUIRE(p, "null pipeline");
    if (device_mine) *device_mine = p->hits_mine;
    if (device_all) *device_all = p->hits_all;
    return TSDF_OK;
}

int tsdf_pipeline_step(tsdf_pipeline *p, const uint16_t *device_depth, const tsdf_camera_matrices *cam, float *device_vertices,
                       float *device_normals, const uint16_t *next_device_depth, const tsdf_camera_matrices *next_cam) {
    TSDF_REQUIRE(p && device_depth && cam && device_vertices, "tsdf_pipeline_step: null argument");
    const uint32_t W = p->width, H = p->height;
    const int b = (int)(p->frames & 1u);
    int rc;
    // TSDF_PIPE_RELEASE=2 (experiment): the next frame's FILTER is released by the end of the previous step already and runs beside
    // this frame's integrate (its buffers were last read by the integrate before that); its culling still waits for this integrate.
    const bool early_filter = p->side && next_device_depth && !p->exchange && tuning().pipe_release == 2;
    if (early_filter) {
        TSDF_HIP(hipEventRecord(p->bulk, p->main), "pipeline: previous step done");
        TSDF_HIP(hipStreamWaitEvent(p->side, p->bulk, 0), "pipeline: release the next frame's filter");
        rc = run_filter(p, next_device_depth, 1 - b, p->side);
        if (rc != TSDF_OK) return rc;
    }
    if (p->ahead_depth && p->ahead_depth == device_depth && p->ahead_buf == b) {
        // The frame filtered (and culled) ahead on the side stream.  A wait packet in the step's stream costs it 6-12 us even when the
        // event completed long ago (the packet breaks the back-to-back dispatch of resolve -> integrate); when the event is already
        // complete at this call no packet is needed, and with TSDF_PIPE_HOST_WAIT=1 the host waits for it here instead (it completes while
        // the previous frame's tail kernel runs: the host then enqueues this step beside the rest of the previous one).
        if (hipEventQuery(p->ready[b]) != hipSuccess) {
            if (tuning().pipe_host_wait) TSDF_HIP(hipEventSynchronize(p->ready[b]), "pipeline: wait for the frame filtered ahead");
            else TSDF_HIP(hipStreamWaitEvent(p->main, p->ready[b], 0), "pipeline: wait for the frame filtered ahead");
        }
    } else {
        if (p->ahead_depth) {
            // Another frame than the one announced: the side stream may still be writing a filtered buffer, its tile maxima and
            // the volume's brick list (prepare) -- wait for it before this step filters and culls for itself, and drop the list.
            TSDF_HIP(hipStreamWaitEvent(p->main, p->ready[p->ahead_buf], 0), "pipeline: wait for the side stream");
            (void)tsdf_integrate_discard_prepared(p->volume);
        }
        rc = run_filter(p, device_depth, b, p->main);
        if (rc != TSDF_OK) return rc;
    }
    p->ahead_depth = nullptr;
    rc = tsdf_integrate_device_tiles(p->volume, p->filtered[b], W, H, cam->pose, cam->inv_pose, cam->k, cam->kinv, p->tile_max[b]);
    if (rc != TSDF_OK) return rc;
    bool late_release = false;
    bool by_word = false;   // the second stream waits for a word the cast's first kernel stores, not for an event of the step's stream
    // The word's promise, kept on every way out: once the side stream has been told to wait for this step's sequence number, a step
    // that returns early -- the filter ahead failed, the cast failed or took the march -- stores the number itself (behind whatever the
    // step's stream holds) and takes the pointer back from the volume: otherwise the side stream waits for ever (synchronize and
    // destroy with it) and a later cell-parallel cast would store through a pointer whose memory the pipeline has freed.
    struct WordGuard {
        tsdf_pipeline *p = nullptr;
        ~WordGuard() {
            if (!p || !p->volume->release_word) return;   // (taken: the cast's first kernel stores it)
            p->volume->release_word = nullptr;
            if (hipStreamWriteValue32(p->main, p->release_word, p->release_seq, 0) != hipSuccess) {
                (void)hipGetLastError();
                set_error("tsdf_pipeline_step: the release word could not be stored; the side stream is left waiting (destroy the pipeline)");
            }
        }
    } word_guard;
    auto filter_ahead = [&](hipEvent_t release) -> int {
        if (by_word) TSDF_HIP(hipStreamWaitValue32(p->side, p->release_word, p->release_seq, hipStreamWaitValueGte, 0xffffffffu), "pipeline: release the next frame's filter");
        else TSDF_HIP(hipStreamWaitEvent(p->side, release, 0), "pipeline: release the next frame's filter");
        int rc_ = early_filter ? TSDF_OK : run_filter(p, next_device_depth, 1 - b, p->side);
        if (rc_ != TSDF_OK) return rc_;
        if (next_cam) {
            rc_ = tsdf_integrate_prepare_device_tiles(p->volume, p->filtered[1 - b], W, H, next_cam->pose, next_cam->inv_pose, next_cam->k,
                                                      next_cam->kinv, p->tile_max[1 - b], p->side);
            if (rc_ != TSDF_OK) return rc_;
        }
        TSDF_HIP(hipEventRecord(p->ready[1 - b], p->side), "pipeline: next frame ready");
        p->ahead_depth = next_device_depth;
        p->ahead_buf = 1 - b;
        return TSDF_OK;
    };
    if (p->side) {
        // An event recorded behind integrate costs the step's stream 5-6 us (a packet between two kernels that would otherwise run back
        // to back; profiles/r05y_sync_ubench.txt).  When the cast that follows is the cell-parallel one its first kernel -- in front of
        // which on this stream lies exactly what the second stream has to wait for -- stores a sequence number into a word of signal
        // memory as it starts, and the second stream waits for that value: the wait's cost lands where there is slack.  Measured
        // (TSDF_PIPE_WORD_RELEASE=1, profiles/r05y_word_release_ab.txt): the gap behind integrate goes (5.6 -> 0 us), but the runtime's
        // wait is a kernel that spins on the word for the 100-150 us until it comes, and the cast's first kernel beside it takes 14 us
        // instead of 11.8: 0.2167 -> 0.2149 ms per step over three runs each.  Off by default.
        const bool tighten_ahead = p->volume->occ_tighten_due && !p->volume->occ_dirty && !(p->flags & TSDF_PIPELINE_NO_TIGHTEN_AHEAD);
        by_word = p->release_word && next_device_depth && !p->exchange && !tighten_ahead && tuning().pipe_release == 0 &&
                  raycast_takes_cells(p->volume, W, H, cam->pose, cam->kinv);
        if (by_word) {
            p->release_seq++;
            p->volume->release_word = p->release_word;
            p->volume->release_value = p->release_seq;
            word_guard.p = p;   // (from here on every way out of the step leaves the word stored: the side stream waits for it)
        } else {
            TSDF_HIP(hipEventRecord(p->done[b], p->main), "pipeline: integrate done");
        }
        if (p->volume->occ_tighten_due && !p->volume->occ_dirty && !(p->flags & TSDF_PIPELINE_NO_TIGHTEN_AHEAD)) {
            // the periodic tightening of the ray caster's flags (every 16th frame: a scan of what integrate has written since the last
            // one, 60-125 us) goes beside this frame's ray cast instead of in front of it: the flags as they are still cover the
            // distances, and the next integrate waits (occupancy_join)
            // (round 6: the rebuild -- 40 + 15 us alone, 45 + 42 beside the cast -- in front of the filter and the culling makes the side
            // stream's chain longer than the cast it runs beside: every 16th step is 1.10-1.12 x the median.  On a third stream, beside the
            // cast AND the filter, it cost the cast more than it saved the chain: 1.17-1.18 x, profiles/r06_step_jitter.txt.)
            TSDF_HIP(hipStreamWaitEvent(p->side, p->done[b], 0), "pipeline: release the occupancy rebuild");
            rc = occupancy_tighten_on(p->volume, p->side);
            if (rc != TSDF_OK) return rc;
        }
        // The next frame's filter + culling on the side stream: released by THIS frame's integrate (never beside integrate_kernel; the
        // other buffer was last read by the previous frame's integrate, which lies before it on the step's stream; the brick list, the
        // boxes and the plane constants are free once this frame's integrate_kernel is done) -- or, TSDF_PIPE_RELEASE=1, only by the end
        // of this frame's bulk ray kernel, so that it runs beside the tail kernel instead (whole-volume casts).
        late_release = next_device_depth && !p->exchange && tuning().pipe_release == 1;
        if (next_device_depth && !late_release) {
            rc = filter_ahead(p->done[b]);
            if (rc != TSDF_OK) return rc;
        }
    }
    if (!p->exchange) {
        p->volume->after_bulk = late_release ? p->bulk : nullptr;
        rc = tsdf_raycast_device(p->volume, W, H, cam->pose, cam->kinv, device_vertices, device_normals);
        p->volume->after_bulk = nullptr;
        // (a cast that did not take the kernel that stores the word -- the list's count arrived between the two looks at it, or the
        // cast failed: word_guard stores it from the step's stream, behind whatever was launched)
        if (rc != TSDF_OK) return rc;
        if (late_release) {
            rc = filter_ahead(p->bulk);
            if (rc != TSDF_OK) return rc;
        }
    } else {
        if (p->xstream && p->merged_pending)   // the previous frame's merge still reads hits_all / the collective hits_mine
            TSDF_HIP(hipStreamWaitEvent(p->main, p->merged, 0), "pipeline: wait for the previous exchange");
        rc = tsdf_raycast_slab_device(p->volume, W, H, cam->pose, cam->kinv, p->hits_mine);
        if (rc != TSDF_OK) return rc;
        hipStream_t xs = p->main;
        if (p->xstream) {
            // exchange + merge on a third stream: the next frame's integrate need not wait for the collective
            TSDF_HIP(hipEventRecord(p->cast, p->main), "pipeline: slab cast done");
            TSDF_HIP(hipStreamWaitEvent(p->xstream, p->cast, 0), "pipeline: exchange after the slab cast");
            xs = p->xstream;
        }
        rc = tsdf_slab_exchange_all_gather(p->exchange, p->hits_mine, p->hits_all, W * H, xs);
        if (rc != TSDF_OK) return rc;
        if (device_normals)
            rc = tsdf_merge_hits_normals_device(p->volume, p->hits_all, (uint32_t)p->exchange->world, W, H, cam->pose, cam->kinv, device_vertices,
                                                device_normals, xs);
        else
            rc = tsdf_merge_hits_device(p->volume, p->hits_all, (uint32_t)p->exchange->world, W, H, cam->pose, cam->kinv, device_vertices, xs);
        if (rc != TSDF_OK) return rc;
        if (p->xstream) {
            TSDF_HIP(hipEventRecord(p->merged, p->xstream), "pipeline: merge done");
            p->merged_pending = true;
        }
    }
    p->frames++;
    return TSDF_OK;
}

int tsdf_pipeline_synchronize(tsdf_pipeline *p) {
    TSDF_REQUIRE(p, "null pipeline");
    TSDF_HIP(hipStreamSynchronize(p->main), "pipeline synchronize");
    if (p->side) TSDF_HIP(hipStreamSynchronize(p->side), "pipeline synchronize");
    if (p->xstream) TSDF_HIP(hipStreamSynchronize(p->xstream), "pipeline synchronize");
    return TSDF_OK;
}

int tsdf_tracker_destroy(tsdf_tracker *t) {
    if (!t) return TSDF_OK;
    if (t->main) (void)hipStreamSynchronize(t->main);
    if (t->side) (void)hipStreamSynchronize(t->side);
    if (t->volume && t->volume->attached == t) {
        (void)tsdf_volume_set_stream(t->volume, t->volume_stream_before);
        t->volume->attached = nullptr;
        if (t->icp) (void)tsdf_icp_set_stream(t->icp, t->icp_stream_before);   // (the ICP's own previous stream, not the volume's)
    }
    for (int b = 0; b < 2; b++) {
        if (t->filtered[b]) (void)hipFree(t->filtered[b]);
        if (t->tile_max[b]) (void)hipFree(t->tile_max[b]);
        if (t->integrated[b]) (void)hipEventDestroy(t->integrated[b]);
    }
    if (t->ready) (void)hipEventDestroy(t->ready);
    if (t->model) (void)hipFree(t->model);
    if (t->side) (void)hipStreamDestroy(t->side);
    if (t->main) (void)hipStreamDestroy(t->main);
    delete t;
    return TSDF_OK;
}

int tsdf_tracker_create(tsdf_volume *volume, const tsdf_bilateral *filter, tsdf_icp *icp, uint32_t width, uint32_t height,
                        float depth_cutoff, int flags, tsdf_tracker **out) {
    TSDF_REQUIRE(out, "tsdf_tracker_create: null out pointer");
    *out = nullptr;
    TSDF_REQUIRE(volume && filter && icp && width > 0 && height > 0 && width <= 65535 && height <= 65535, "tsdf_tracker_create: bad argument");
    TSDF_REQUIRE(volume->z_begin == 0 && volume->z_end == volume->g.Z, "tsdf_tracker_create: tracking needs a whole volume");
    TSDF_REQUIRE(!volume->attached, "tsdf_tracker_create: the volume is attached to another pipeline or tracker (destroy that one first)");
    tsdf_tracker *t = new (std::nothrow) tsdf_tracker();
    if (!t) {
        set_error("out of host memory");
        return TSDF_ERR_NOMEM;
    }
    memset(t, 0, sizeof(*t));
    t->filter = filter;
    t->width = width;
    t->height = height;
    t->depth_cutoff = depth_cutoff;
    t->volume_stream_before = volume->stream;
    int least = 0, greatest = 0;
    hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
    const bool overlap = (flags & TSDF_PIPELINE_OVERLAP) != 0;
    const int normal = (greatest <= 0 && 0 <= least) ? 0 : least;
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&t->main, hipStreamNonBlocking, overlap ? greatest : normal);
    if (e == hipSuccess && overlap) e = hipStreamCreateWithPriority(&t->side, hipStreamNonBlocking, normal);
    const size_t n = (size_t)width * height;
    const size_t tiles = (size_t)((width + TSDF_DEPTH_TILE - 1) / TSDF_DEPTH_TILE) * ((height + TSDF_DEPTH_TILE - 1) / TSDF_DEPTH_TILE);
    for (int b = 0; b < 2 && e == hipSuccess; b++) {
        e = hipMalloc((void **)&t->filtered[b], n * sizeof(uint16_t));
        if (e == hipSuccess) e = hipMalloc((void **)&t->tile_max[b], tiles * sizeof(uint16_t));
        if (e == hipSuccess) e = hipEventCreateWithFlags(&t->integrated[b], stream_order_event_flags());
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&t->ready, stream_order_event_flags());
    if (e == hipSuccess) e = hipMalloc((void **)&t->model, n * sizeof(uint16_t));
    if (e != hipSuccess) {
        const int rc = hip_fail(e, "tsdf_tracker_create");
        tsdf_tracker_destroy(t);
        return rc;
    }
    (void)hipStreamSynchronize(volume->stream);   // whatever the volume has in flight comes before the tracker's first launch
    t->volume = volume;
    t->icp = icp;
    {
        void *before = nullptr;
        (void)tsdf_icp_stream(icp, &before);
        t->icp_stream_before = (hipStream_t)before;
    }
    (void)tsdf_volume_set_stream(volume, t->main);
    (void)tsdf_icp_set_stream(icp, t->main);
    volume->attached = t;
    *out = t;
    return TSDF_OK;
}

int tsdf_tracker_streams(const tsdf_tracker *t, void **main_stream, void **side_stream) {
    TSDF_REQUIRE(t, "null tracker");
    if (main_stream) *main_stream = t->main;
    if (side_stream) *side_stream = t->side;
    return TSDF_OK;
}

int tsdf_tracker_buffers(const tsdf_tracker *t, const uint16_t **device_model, const uint16_t **device_filtered) {
    TSDF_REQUIRE(t, "null tracker");
    if (device_model) *device_model = t->model;
    if (device_filtered) *device_filtered = t->filtered[t->cur];
    return TSDF_OK;
}

int tsdf_tracker_filter(tsdf_tracker *t, const uint16_t *device_depth) {
    TSDF_REQUIRE(t && device_depth, "tsdf_tracker_filter: null argument");
    const int b = (int)(t->frames & 1u);
    const hipStream_t s = t->side ? t->side : t->main;
    // buffer b was last read by the integrate two frames back
    if (t->side) TSDF_HIP(hipStreamWaitEvent(t->side, t->integrated[b], 0), "tracker: the frame buffer is free");
    int rc = tsdf_bilateral_filter_u16_device_tiles(t->filter, device_depth, t->filtered[b], (int)t->width, (int)t->height, t->tile_max[b], s);
    if (rc != TSDF_OK) return rc;
    if (t->frames > 0) {
        // ICPOdometry::initICP of the new frame (pyramid, vertex and normal maps): they wait for nothing but the filter
        (void)tsdf_icp_set_stream(t->icp, s);
        rc = tsdf_icp_init_device(t->icp, 0, t->filtered[b], t->depth_cutoff);
        (void)tsdf_icp_set_stream(t->icp, t->main);
        if (rc != TSDF_OK) return rc;
    }
    if (t->side) {
        TSDF_HIP(hipEventRecord(t->ready, t->side), "tracker: frame ready");
        t->ready_pending = true;
    }
    t->cur = b;
    t->have_frame = true;
    return TSDF_OK;
}

static int tracker_join(tsdf_tracker *t) {
    if (t->ready_pending) {
        TSDF_HIP(hipStreamWaitEvent(t->main, t->ready, 0), "tracker: wait for the filtered frame");
        t->ready_pending = false;
    }
    return TSDF_OK;
}

int tsdf_tracker_align(tsdf_tracker *t, const tsdf_camera_matrices *previous, double T_prev_curr[16], float *last_error,
                       float *last_inliers) {
    TSDF_REQUIRE(t && previous && T_prev_curr, "tsdf_tracker_align: null argument");
    TSDF_REQUIRE(t->have_frame && t->frames > 0, "tsdf_tracker_align: no frame filtered, or nothing integrated to align it to");
    // the model image: the volume rendered from the previous pose (GPURaycaster::render_to_depth_image, src/RayCaster/GPURaycaster.cu:575-579)
    int rc = tsdf_raycast_depth_device(t->volume, t->width, t->height, previous->pose, previous->inv_pose, previous->kinv, t->model, nullptr);
    if (rc != TSDF_OK) return rc;
    rc = tracker_join(t);   // (the model's maps reuse the pyramid scratch of the new frame's)
    if (rc != TSDF_OK) return rc;
    rc = tsdf_icp_init_device(t->icp, 1, t->model, t->depth_cutoff);
    if (rc != TSDF_OK) return rc;
    return tsdf_icp_get_incremental_transformation(t->icp, T_prev_curr, last_error, last_inliers);
}

int tsdf_tracker_integrate(tsdf_tracker *t, const tsdf_camera_matrices *camera) {
    TSDF_REQUIRE(t && camera, "tsdf_tracker_integrate: null argument");
    TSDF_REQUIRE(t->have_frame, "tsdf_tracker_integrate: no frame filtered");
    int rc = tracker_join(t);
    if (rc != TSDF_OK) return rc;
    const int b = t->cur;
    rc = tsdf_integrate_device_tiles(t->volume, t->filtered[b], t->width, t->height, camera->pose, camera->inv_pose, camera->k, camera->kinv,
                                     t->tile_max[b]);
    if (rc != TSDF_OK) return rc;
    if (t->side) TSDF_HIP(hipEventRecord(t->integrated[b], t->main), "tracker: integrate done");
    t->have_frame = false;
    t->frames++;
    return TSDF_OK;
}

int tsdf_tracker_synchronize(tsdf_tracker *t) {
    TSDF_REQUIRE(t, "null tracker");
    TSDF_HIP(hipStreamSynchronize(t->main), "tracker synchronize");
    if (t->side) TSDF_HIP(hipStreamSynchronize(t->side), "tracker synchronize");
    return TSDF_OK;
}

}  // extern "C"
