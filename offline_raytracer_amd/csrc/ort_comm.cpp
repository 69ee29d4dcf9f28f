/*
 * ort_comm.cpp -- the multi-GPU side of the render call, in the C++ host: block packing and the ONE collective
 * of the path, a gather of every rank's packed 8x8 blocks to rank 0 over RCCL (xGMI inside a node).
 *
 * The reference is one shared-memory process (code/macos_main.mm:565-671: eight pthreads pulling 32x32 tiles from
 * one queue and writing one framebuffer); its counterpart on N GPUs is: scene replicated, blocks dealt round-robin
 * (block_id % world == rank), every rank renders into its own PACKED buffer [local block][pixel in block][rgb]
 * (ORT_RENDER_PACKED; the CHUNK partial sums use the same layout, so a rank's workspace is 1/N of a frame's), then
 *     root : ncclGroupStart; ncclRecv x (N-1) into per-rank staging slots; ncclGroupEnd; un-permute into the frame
 *     peers: ncclSend of the packed buffer
 * -- each peer->root transfer rides its own xGMI link, W*H*12/N bytes per rank.  No reduction, nothing else.
 *
 * RCCL is bound at run time (dlopen of librccl.so) the first time a communicator is created: a one-GPU render never
 * loads it.  Two ways to form the communicator: one process per GPU (ort_comm_create: the unique id travels out of
 * band, e.g. through torch.distributed or a file) and one process driving all GPUs (ort_comm_create_local:
 * ncclCommInitAll), which is what bin/ort_render --gpus N uses.
 */
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

/* RCCL types and enum values only: the entry points are looked up with dlsym at run time, so a one-GPU build needs
   neither the library nor -- below -- its header.  Without the header the handful of ABI-stable declarations the
   gather uses are spelled out here (nccl.h: opaque communicator pointer, 128-byte unique id, result and data-type
   enums with ncclSuccess = 0 and ncclFloat = 7). */
#if defined(__has_include) && __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclFloat = 7 } ncclDataType_t;
}
#endif

#include "ort_scene.h"

namespace ort {

namespace {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl g_rccl;

int load_rccl(std::string *err) {
    if (g_rccl.lib) return ORT_OK;
    /* If the process already has an RCCL (a host program that links it; PyTorch ships its own librccl.so), use THAT
       one -- two RCCL builds in one process is asking for trouble -- otherwise load the system's.  RTLD_LOCAL: nothing
       else in the process should start resolving nccl* symbols against a library we pulled in. */
    const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void *h = nullptr;
    for (const char *n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD))) break;
    for (const char *n : names)
        if (!h && (h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) { *err = std::string("cannot load librccl.so: ") + dlerror(); return ORT_ERR_UNSUPPORTED; }
    Rccl r;
    r.lib = h;
#define ORT_SYM(field, name)                                                                      \
    *(void **)(&r.field) = dlsym(h, name);                                                        \
    if (!r.field) { *err = std::string("librccl.so lacks ") + name; dlclose(h); return ORT_ERR_UNSUPPORTED; }
    ORT_SYM(GetUniqueId, "ncclGetUniqueId")
    ORT_SYM(CommInitRank, "ncclCommInitRank")
    ORT_SYM(CommInitAll, "ncclCommInitAll")
    ORT_SYM(CommDestroy, "ncclCommDestroy")
    ORT_SYM(GroupStart, "ncclGroupStart")
    ORT_SYM(GroupEnd, "ncclGroupEnd")
    ORT_SYM(Send, "ncclSend")
    ORT_SYM(Recv, "ncclRecv")
    ORT_SYM(GetErrorString, "ncclGetErrorString")
#undef ORT_SYM
    g_rccl = r;
    return ORT_OK;
}

#define ORT_NCCL(call)                                                                            \
    do {                                                                                          \
        ncclResult_t r_ = (call);                                                                 \
        if (r_ != ncclSuccess) { *err = std::string(#call) + ": " + g_rccl.GetErrorString(r_); return ORT_ERR_HIP; } \
    } while (0)
#define ORT_HIPC(call)                                                                            \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) { *err = std::string(#call) + ": " + hipGetErrorString(e_); return ORT_ERR_HIP; } \
    } while (0)

/* blocks of shard (index, count) in a W x H image: ids index, index + count, ... below the grid size */
inline uint32_t grid_w(int32_t w) { return (uint32_t)((w + 7) / 8); }
inline uint32_t grid_total(int32_t w, int32_t h) { return grid_w(w) * (uint32_t)((h + 7) / 8); }
inline uint32_t blocks_of(int32_t w, int32_t h, uint32_t index, uint32_t count) {
    const uint32_t total = grid_total(w, h);
    return total > index ? (total - index + count - 1) / count : 0;
}

/* one thread per pixel of the shard's packed buffer: packed [local block][pixel in block] -> full frame (row 0 = bottom) */
__global__ void unpack_blocks_kernel(const float *packed, float *full, int w, int h, uint32_t index, uint32_t count, uint32_t nblocks) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nblocks * 64u) return;
    const uint32_t blk = index + (i >> 6) * count, pin = i & 63u, gw = (uint32_t)((w + 7) / 8);
    const int x = (int)((blk % gw) * 8u + (pin & 7u)), y = (int)((blk / gw) * 8u + (pin >> 3));
    if (x >= w || y >= h) return;
    const float *s = packed + 3u * (size_t)i;
    float *d = full + 3u * ((size_t)y * (size_t)w + (size_t)x);
    d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
}

int unpack_on_device(const void *d_packed, int32_t w, int32_t h, uint32_t index, uint32_t count, void *d_full, hipStream_t stream, std::string *err) {
    const uint32_t nb = blocks_of(w, h, index, count);
    if (!nb) return ORT_OK;
    const uint32_t n = nb * 64u;
    hipLaunchKernelGGL(unpack_blocks_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, (const float *)d_packed, (float *)d_full, w, h, index, count, nb);
    ORT_HIPC(hipGetLastError());
    return ORT_OK;
}

} // namespace

struct Comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    void *staging = nullptr; /* root: (world - 1) slots of max_blocks * 768 B */
    size_t staging_bytes = 0;
};

uint64_t comm_shard_blocks(int32_t w, int32_t h, uint32_t index, uint32_t count) { return blocks_of(w, h, index, count ? count : 1); }

void pack_blocks_host(const float *full, int32_t w, int32_t h, uint32_t index, uint32_t count, float *packed) {
    const uint32_t nb = blocks_of(w, h, index, count), gw = grid_w(w);
    for (uint32_t k = 0; k < nb; ++k) {
        const uint32_t blk = index + k * count;
        for (uint32_t pin = 0; pin < 64; ++pin) {
            const int x = (int)((blk % gw) * 8u + (pin & 7u)), y = (int)((blk / gw) * 8u + (pin >> 3));
            float *d = packed + 3u * ((size_t)k * 64u + pin);
            if (x < w && y < h) memcpy(d, full + 3u * ((size_t)y * (size_t)w + (size_t)x), 12);
            else d[0] = d[1] = d[2] = 0.0f;
        }
    }
}

void unpack_blocks_host(const float *packed, int32_t w, int32_t h, uint32_t index, uint32_t count, float *full) {
    const uint32_t nb = blocks_of(w, h, index, count), gw = grid_w(w);
    for (uint32_t k = 0; k < nb; ++k) {
        const uint32_t blk = index + k * count;
        for (uint32_t pin = 0; pin < 64; ++pin) {
            const int x = (int)((blk % gw) * 8u + (pin & 7u)), y = (int)((blk / gw) * 8u + (pin >> 3));
            if (x < w && y < h) memcpy(full + 3u * ((size_t)y * (size_t)w + (size_t)x), packed + 3u * ((size_t)k * 64u + pin), 12);
        }
    }
}

int unpack_blocks_device(const void *d_packed, int32_t w, int32_t h, uint32_t index, uint32_t count, void *d_full, void *stream, std::string *err) {
    return unpack_on_device(d_packed, w, h, index, count, d_full, (hipStream_t)stream, err);
}

int comm_unique_id(void *id, std::string *err) {
    int rc = load_rccl(err);
    if (rc) return rc;
    ncclUniqueId u;
    ORT_NCCL(g_rccl.GetUniqueId(&u));
    static_assert(sizeof(u) == ORT_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(id, &u, sizeof(u));
    return ORT_OK;
}

int comm_create(const void *id, int rank, int world, int device, Comm **out, std::string *err) {
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) { *err = "bad rank / world"; return ORT_ERR_INVALID; }
    Comm *c = new Comm();
    c->rank = rank; c->world = world; c->device = device;
    /* ORT_COMM_FORCE_RCCL (tests on a one-GPU box): a world of one still gets a real communicator, and its gather
       sends the packed blocks to itself through RCCL -- the same entry points, argument lists and stream semantics as
       the N-rank path, exercised on hardware */
    const bool force = world == 1 && getenv("ORT_COMM_FORCE_RCCL") != nullptr;
    if (world > 1 || force) {
        int rc = load_rccl(err);
        if (rc) { delete c; return rc; }
        if (!id && !force) { delete c; *err = "null unique id"; return ORT_ERR_INVALID; }
        hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) { delete c; *err = std::string("hipSetDevice: ") + hipGetErrorString(e); return ORT_ERR_HIP; }
        ncclUniqueId u;
        if (id) memcpy(&u, id, sizeof(u));
        else {
            ncclResult_t r0 = g_rccl.GetUniqueId(&u);
            if (r0 != ncclSuccess) { *err = std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(r0); delete c; return ORT_ERR_HIP; }
        }
        ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, u, rank);
        if (r != ncclSuccess) { *err = std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r); delete c; return ORT_ERR_HIP; }
    }
    *out = c;
    return ORT_OK;
}

int comm_create_local(int world, const int *devices, Comm **out, std::string *err) {
    for (int i = 0; i < world; ++i) out[i] = nullptr;
    if (world < 1) { *err = "bad world"; return ORT_ERR_INVALID; }
    std::vector<ncclComm_t> comms((size_t)world, nullptr);
    /* several shards on ONE device (rehearsals on a one-GPU box): no communicator -- RCCL refuses two ranks on a
       device, and nothing has to travel; the gather is then the un-permute kernels alone */
    bool one_device = devices != nullptr;
    for (int i = 1; i < world && one_device; ++i) one_device = devices[i] == devices[0];
    if (world > 1 && !one_device) {
        int rc = load_rccl(err);
        if (rc) return rc;
        ORT_NCCL(g_rccl.CommInitAll(comms.data(), world, devices));
    }
    for (int i = 0; i < world; ++i) {
        Comm *c = new Comm();
        c->rank = i; c->world = world; c->device = devices ? devices[i] : i; c->comm = comms[(size_t)i];
        out[i] = c;
    }
    return ORT_OK;
}

void comm_destroy(Comm *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->staging) (void)hipFree(c->staging);
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    delete c;
}

static int ensure_staging(Comm *c, size_t need, std::string *err) {
    if (c->staging_bytes >= need) return ORT_OK;
    if (c->staging) ORT_HIPC(hipFree(c->staging));
    c->staging = nullptr; c->staging_bytes = 0;
    ORT_HIPC(hipMalloc(&c->staging, need));
    c->staging_bytes = need;
    return ORT_OK;
}

/* this rank's part of the gather, to be called between ncclGroupStart / ncclGroupEnd */
static int post_transfers(Comm *c, const void *d_packed, int32_t w, int32_t h, hipStream_t stream, std::string *err) {
    const size_t slot = (size_t)blocks_of(w, h, 0, (uint32_t)c->world) * 768u; /* rank 0 owns the most blocks */
    if (c->rank == 0) {
        for (int r = 1; r < c->world; ++r) {
            const size_t n = (size_t)blocks_of(w, h, (uint32_t)r, (uint32_t)c->world) * 192u;
            if (n) ORT_NCCL(g_rccl.Recv((char *)c->staging + (size_t)(r - 1) * slot, n, ncclFloat, r, c->comm, stream));
        }
    } else {
        const size_t n = (size_t)blocks_of(w, h, (uint32_t)c->rank, (uint32_t)c->world) * 192u;
        if (n) ORT_NCCL(g_rccl.Send(d_packed, n, ncclFloat, 0, c->comm, stream));
    }
    return ORT_OK;
}

static int unpack_all(Comm *root, const void *d_packed_root, void *d_full, int32_t w, int32_t h, hipStream_t stream, std::string *err) {
    const size_t slot = (size_t)blocks_of(w, h, 0, (uint32_t)root->world) * 768u;
    int rc = unpack_on_device(d_packed_root, w, h, 0, (uint32_t)root->world, d_full, stream, err);
    for (int r = 1; r < root->world && rc == ORT_OK; ++r)
        rc = unpack_on_device((char *)root->staging + (size_t)(r - 1) * slot, w, h, (uint32_t)r, (uint32_t)root->world, d_full, stream, err);
    return rc;
}

int gather_framebuffer(Comm *c, const void *d_packed, void *d_full, int32_t w, int32_t h, void *stream_v, std::string *err) {
    hipStream_t stream = (hipStream_t)stream_v;
    ORT_HIPC(hipSetDevice(c->device));
    if (c->rank == 0 && !d_full) { *err = "rank 0 needs the full framebuffer"; return ORT_ERR_INVALID; }
    int rc;
    if (c->world == 1 && c->comm) { /* forced communicator of one: the blocks travel rank 0 -> rank 0 through RCCL */
        const size_t n = (size_t)blocks_of(w, h, 0, 1) * 192u;
        if ((rc = ensure_staging(c, n * 4u + 16u, err))) return rc;
        ORT_NCCL(g_rccl.GroupStart());
        ncclResult_t rs = g_rccl.Send(d_packed, n, ncclFloat, 0, c->comm, stream);
        ncclResult_t rr = g_rccl.Recv(c->staging, n, ncclFloat, 0, c->comm, stream);
        ncclResult_t ge = g_rccl.GroupEnd();
        if (rs != ncclSuccess || rr != ncclSuccess || ge != ncclSuccess) {
            *err = std::string("RCCL self transfer: ") + g_rccl.GetErrorString(rs != ncclSuccess ? rs : rr != ncclSuccess ? rr : ge);
            return ORT_ERR_HIP;
        }
        return unpack_on_device(c->staging, w, h, 0, 1, d_full, stream, err);
    }
    if (c->world == 1) return unpack_on_device(d_packed, w, h, 0, 1, d_full, stream, err);
    if (c->rank == 0) {
        const size_t slot = (size_t)blocks_of(w, h, 0, (uint32_t)c->world) * 768u;
        if ((rc = ensure_staging(c, slot * (size_t)(c->world - 1), err))) return rc;
    }
    ORT_NCCL(g_rccl.GroupStart());
    rc = post_transfers(c, d_packed, w, h, stream, err);
    ncclResult_t ge = g_rccl.GroupEnd();
    if (rc) return rc;
    if (ge != ncclSuccess) { *err = std::string("ncclGroupEnd: ") + g_rccl.GetErrorString(ge); return ORT_ERR_HIP; }
    if (c->rank == 0) return unpack_all(c, d_packed, d_full, w, h, stream, err);
    return ORT_OK;
}

/* one process, all GPUs: every rank's transfers inside one group */
int gather_framebuffer_local(Comm **cs, int world, const void *const *d_packed, void *d_full_root, int32_t w, int32_t h,
                             void *const *streams, std::string *err) {
    if (world < 1 || !cs || !cs[0]) { *err = "bad communicator list"; return ORT_ERR_INVALID; }
    if (world == 1) return gather_framebuffer(cs[0], d_packed[0], d_full_root, w, h, streams ? streams[0] : nullptr, err);
    int rc;
    ORT_HIPC(hipSetDevice(cs[0]->device));
    if (!cs[0]->comm) { /* all shards on one device (comm_create_local): un-permute each in place */
        for (int r = 0; r < world; ++r) {
            if (cs[r]->device != cs[0]->device) { *err = "communicators without RCCL must share one device"; return ORT_ERR_INVALID; }
            if ((rc = unpack_on_device(d_packed[r], w, h, (uint32_t)r, (uint32_t)world, d_full_root, (hipStream_t)(streams ? streams[0] : nullptr), err))) return rc;
        }
        return ORT_OK;
    }
    const size_t slot = (size_t)blocks_of(w, h, 0, (uint32_t)world) * 768u;
    if ((rc = ensure_staging(cs[0], slot * (size_t)(world - 1), err))) return rc;
    ORT_NCCL(g_rccl.GroupStart());
    for (int r = 0; r < world && rc == ORT_OK; ++r) {
        hipError_t e = hipSetDevice(cs[r]->device);
        if (e != hipSuccess) { *err = std::string("hipSetDevice: ") + hipGetErrorString(e); rc = ORT_ERR_HIP; break; }
        rc = post_transfers(cs[r], d_packed[r], w, h, (hipStream_t)(streams ? streams[r] : nullptr), err);
    }
    ncclResult_t ge = g_rccl.GroupEnd();
    if (rc) return rc;
    if (ge != ncclSuccess) { *err = std::string("ncclGroupEnd: ") + g_rccl.GetErrorString(ge); return ORT_ERR_HIP; }
    ORT_HIPC(hipSetDevice(cs[0]->device));
    return unpack_all(cs[0], d_packed[0], d_full_root, w, h, (hipStream_t)(streams ? streams[0] : nullptr), err);
}

} // namespace ort
