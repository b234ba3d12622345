// Collectives inside the library: RCCL over xGMI, one process per GPU (north_star:
// "MSM and NTT shard across the 8 GPUs of one node with RCCL all-reduce/all-gather over xGMI").
//
// A Rust / C caller has no torch.distributed: it passes the 128-byte unique id of rank 0 to every
// rank (any out-of-band channel: a file, an env var, MPI, the job launcher) and calls
// zk_comm_init; from then on sharded proving sessions (zk_proof_set_sharding_comm) and
// zk_ntt_sharded exchange their data themselves, stream-ordered on the context's stream:
//   * commitments of a transcript round   all-gather of 64 B per column   (latency-bound)
//   * finished quotient cosets            all-gather of n x 32 B per rank, device to device
//   * advice columns (device-gather mode) all-gather of n x 32 B per rank, device to device
//   * one NTT over W ranks                all-to-all: W - 1 grouped ncclSend / ncclRecv pairs, so every
//                                         xGMI link carries its m / W x 32 B slice at the same time
//                                         (point-to-point fabric: a ring would serialise the links)
// There is no elliptic-curve reduction in RCCL (sum / prod / min / max on numeric types only), so a
// point-sharded MSM all-gathers its 64-byte partial results and adds them on the host.
//
// librccl is loaded on first use (dlopen): the library itself has no link-time dependency on it, and
// a single-GPU deployment never touches it.
#include <dlfcn.h>

#include <string>
#include <rccl/rccl.h>

#include "ctx.hpp"

namespace zk {

struct Rccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
};

static Rccl& rccl() {
    static Rccl r = [] {
        Rccl x;
        // The RCCL that belongs to the HIP runtime this library links, by path: a bare "librccl.so.1" resolves to whatever copy
        // the process already holds under that name -- torch's, when torch is imported, which talks to torch's own (possibly
        // never initialised) HIP runtime instead of the one our streams and buffers live in.
        std::string beside;
        Dl_info info;
        if (dladdr((const void*)&hipGetDeviceCount, &info) && info.dli_fname) {
            beside = info.dli_fname;
            const size_t slash = beside.rfind('/');
            beside = slash == std::string::npos ? std::string() : beside.substr(0, slash + 1) + "librccl.so.1";
        }
        for (const std::string& name : {beside, std::string("/opt/rocm/lib/librccl.so.1"), std::string("librccl.so.1"), std::string("librccl.so")}) {
            if (name.empty()) continue;
            x.lib = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (x.lib) break;
        }
        if (!x.lib) return x;
#define ZK_SYM(field, sym) x.field = (decltype(x.field))dlsym(x.lib, sym); if (!x.field) return x;
        ZK_SYM(GetUniqueId, "ncclGetUniqueId") ZK_SYM(CommInitRank, "ncclCommInitRank") ZK_SYM(CommDestroy, "ncclCommDestroy")
        ZK_SYM(AllGather, "ncclAllGather") ZK_SYM(Send, "ncclSend") ZK_SYM(Recv, "ncclRecv") ZK_SYM(GroupStart, "ncclGroupStart")
        ZK_SYM(GroupEnd, "ncclGroupEnd") ZK_SYM(GetErrorString, "ncclGetErrorString")
#undef ZK_SYM
        // RCCL looks the HSA runtime up by name (dlopen("libhsa-runtime64.so")) for its capability queries.  A process that has
        // imported torch holds torch's bundled copy under that name, and until torch touches the GPU that copy is not
        // initialised: the queries fail with HSA_STATUS_ERROR_NOT_INITIALIZED and ncclCommInitRank reports "no ROCm-capable
        // device".  hsa_init is reference-counted: initialise whichever copies are already loaded.
        for (const char* name : {"libhsa-runtime64.so", "libhsa-runtime64.so.1"}) {
            if (void* h = dlopen(name, RTLD_NOW | RTLD_NOLOAD)) {
                if (auto init = (int (*)())dlsym(h, "hsa_init")) (void)init();
            }
        }
        x.ok = true;
        return x;
    }();
    return r;
}

#define ZK_NCCL(ctx, call)                                                                          \
    do {                                                                                           \
        ncclResult_t r__ = (call);                                                                 \
        if (r__ != ncclSuccess) return (ctx)->fail(ZK_ERR_HIP, "%s failed: %s", #call, rccl().GetErrorString(r__)); \
    } while (0)

bool comm_ready(const zk_ctx* ctx) { return ctx->comm != nullptr; }

// all-gather of DEVICE buffers on the context's stream (no host synchronisation)
int comm_allgather_dev(zk_ctx* ctx, const void* d_send, size_t bytes, void* d_recv) {
    if (!ctx->comm) return ctx->fail(ZK_ERR_INVALID_ARG, "no communicator: call zk_comm_init first");
    ZK_NCCL(ctx, rccl().AllGather(d_send, d_recv, bytes, ncclUint8, (ncclComm_t)ctx->comm, ctx->stream));
    return ZK_OK;
}
// all-gather of small HOST buffers (commitments): staged through the context's scratch
int comm_allgather_host(zk_ctx* ctx, const void* h_send, size_t bytes, void* h_recv) {
    if (!ctx->comm) return ctx->fail(ZK_ERR_INVALID_ARG, "no communicator: call zk_comm_init first");
    const size_t world = ctx->comm_world;
    char* d = (char*)ctx->get_scratch(SC_COMM, bytes * (world + 1) + 256);
    if (!d) return ZK_ERR_OOM;
    ZK_HIP(ctx, hipMemcpyAsync(d, h_send, bytes, hipMemcpyHostToDevice, ctx->stream));
    ZK_NCCL(ctx, rccl().AllGather(d, d + ((bytes + 255) & ~(size_t)255), bytes, ncclUint8, (ncclComm_t)ctx->comm, ctx->stream));
    ZK_HIP(ctx, hipMemcpyAsync(h_recv, d + ((bytes + 255) & ~(size_t)255), bytes * world, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZK_OK;
}
// all-to-all of DEVICE buffers: block p of d_send goes to rank p, block p of d_recv comes from rank p
int comm_alltoall_dev(zk_ctx* ctx, const void* d_send, size_t bytes_per_peer, void* d_recv) {
    if (!ctx->comm) return ctx->fail(ZK_ERR_INVALID_ARG, "no communicator: call zk_comm_init first");
    ZK_NCCL(ctx, rccl().GroupStart());
    // inside an open group nothing may return early: the first error is remembered, the group is always closed
    ncclResult_t first = ncclSuccess;
    for (uint32_t p = 0; p < ctx->comm_world && first == ncclSuccess; ++p) {
        first = rccl().Send((const char*)d_send + (size_t)p * bytes_per_peer, bytes_per_peer, ncclUint8, (int)p, (ncclComm_t)ctx->comm, ctx->stream);
        if (first == ncclSuccess) first = rccl().Recv((char*)d_recv + (size_t)p * bytes_per_peer, bytes_per_peer, ncclUint8, (int)p, (ncclComm_t)ctx->comm, ctx->stream);
    }
    const ncclResult_t closed = rccl().GroupEnd();
    ZK_NCCL(ctx, first);
    ZK_NCCL(ctx, closed);
    return ZK_OK;
}
void comm_release(zk_ctx* ctx) {
    if (ctx->comm && rccl().ok) (void)rccl().CommDestroy((ncclComm_t)ctx->comm);
    ctx->comm = nullptr;
    ctx->comm_world = 1;
    ctx->comm_rank = 0;
}

}  // namespace zk

using namespace zk;

extern "C" {

int zk_comm_unique_id(void* out128) {
    if (!out128) return ZK_ERR_INVALID_ARG;
    if (!rccl().ok) return ZK_ERR_UNSUPPORTED;
    ncclUniqueId id;
    if (rccl().GetUniqueId(&id) != ncclSuccess) return ZK_ERR_HIP;
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    memcpy(out128, &id, sizeof id);
    return ZK_OK;
}

int zk_comm_init(zk_ctx* ctx, const void* id128, uint32_t rank, uint32_t world) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, id128 && world >= 1 && rank < world, "need the 128-byte unique id and rank < world");
    if (!rccl().ok) return ctx->fail(ZK_ERR_UNSUPPORTED, "librccl could not be loaded");
    if (ctx->comm) comm_release(ctx);
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclComm_t comm = nullptr;
    ZK_NCCL(ctx, rccl().CommInitRank(&comm, (int)world, id, (int)rank));
    ctx->comm = comm;
    ctx->comm_rank = rank;
    ctx->comm_world = world;
    return ZK_OK;
}

int zk_comm_destroy(zk_ctx* ctx) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    (void)hipStreamSynchronize(ctx->stream);
    comm_release(ctx);
    return ZK_OK;
}

// all-gather / all-to-all over the communicator as plain entry points (device buffers; the call is
// stream-ordered and returns without waiting): what zk_allgather_fn / zk_alltoall_fn callbacks of a
// caller without its own collectives can forward to
int zk_comm_allgather(zk_ctx* ctx, const void* d_send, size_t bytes, void* d_recv) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, d_send && d_recv, "null pointer");
    return comm_allgather_dev(ctx, d_send, bytes, d_recv);
}
int zk_comm_alltoall(zk_ctx* ctx, const void* d_send, size_t bytes_per_peer, void* d_recv) {
    if (!ctx) return ZK_ERR_INVALID_ARG;
    ZK_REQUIRE(ctx, d_send && d_recv, "null pointer");
    return comm_alltoall_dev(ctx, d_send, bytes_per_peer, d_recv);
}

}  // extern "C"
