// libdotring_hip.so — C ABI, part 5 of 5: the RCCL communicator of the base-sharded MSM (second mode).
//
// One process per GPU.  A large MSM shards by bases: rank g reduces its (base, scalar) pairs to ONE G1 point, the points
// are exchanged with ncclAllGather (96 bytes per rank: latency-bound on xGMI, the per-link bandwidth is irrelevant) and
// every rank folds them with the group law — point addition is not an RCCL reduction op, so this is an all-gather plus a
// local fold, not an all-reduce.  librccl is opened with dlopen on first use (it is a 570 MB library no single-GPU
// process needs), and nothing here goes through PyTorch: the unique id travels between the ranks by whatever channel
// the launcher offers (dot_ring_amd/parallel.py uses a TCP socket on MASTER_ADDR).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "capi_internal.hpp"

using namespace dri;

namespace {

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclAllGather) all_gather = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclCommCount) comm_count = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    std::string error;
};

Rccl& rccl() {
    static Rccl r = [] {
        Rccl x;
        const char* env = std::getenv("DOTRING_RCCL_LIB");
        const bool forced = env && *env;                 // DOTRING_RCCL_LIB names THE library: no silent fallback to another one
        const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            if (!n || !*n) continue;
            if (forced && n != env) break;
            x.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (x.handle) break;
        }
        if (!x.handle) {
            const char* e = dlerror();
            x.error = std::string("librccl not found: ") + (e ? e : "dlopen failed");
            return x;
        }
        auto sym = [&](const char* name) {
            void* p = dlsym(x.handle, name);
            if (!p && x.error.empty()) x.error = std::string("librccl lacks ") + name;
            return p;
        };
        x.get_unique_id = reinterpret_cast<decltype(x.get_unique_id)>(sym("ncclGetUniqueId"));
        x.comm_init_rank = reinterpret_cast<decltype(x.comm_init_rank)>(sym("ncclCommInitRank"));
        x.all_gather = reinterpret_cast<decltype(x.all_gather)>(sym("ncclAllGather"));
        x.comm_destroy = reinterpret_cast<decltype(x.comm_destroy)>(sym("ncclCommDestroy"));
        x.comm_count = reinterpret_cast<decltype(x.comm_count)>(sym("ncclCommCount"));
        x.error_string = reinterpret_cast<decltype(x.error_string)>(sym("ncclGetErrorString"));
        return x;
    }();
    return r;
}

int rccl_ready() {
    Rccl& r = rccl();
    if (!r.error.empty()) return fail(DR_ERR_DEVICE, r.error);
    return DR_OK;
}

#define NCCL_TRY(expr)                                                                                       \
    do {                                                                                                     \
        ncclResult_t r_ = (expr);                                                                            \
        if (r_ != ncclSuccess) return fail(DR_ERR_DEVICE, std::string(#expr) + ": " + rccl().error_string(r_)); \
    } while (0)

}  // namespace

struct dr_comm {
    dr_ctx* ctx = nullptr;
    int device = 0, rank = 0, world = 1;
    ncclComm_t comm = nullptr;
    Scratch send, recv;
};

static_assert(DR_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");

int dr_comm_unique_id(uint8_t out_id[DR_COMM_ID_BYTES]) {
    if (!out_id) return fail(DR_ERR_INVALID, "null buffer");
    TRY(rccl_ready());
    ncclUniqueId id;
    NCCL_TRY(rccl().get_unique_id(&id));
    std::memcpy(out_id, id.internal, DR_COMM_ID_BYTES);
    return DR_OK;
}

int dr_comm_create(dr_ctx* ctx, const uint8_t id_bytes[DR_COMM_ID_BYTES], int rank, int world, dr_comm** out) {
    TRY(use_ctx(ctx));
    if (!out || !id_bytes) return fail(DR_ERR_INVALID, "null argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return fail(DR_ERR_INVALID, "bad rank / world size");
    TRY(rccl_ready());
    dr_comm* c = new (std::nothrow) dr_comm();
    if (!c) return fail(DR_ERR_NOMEM, "out of host memory");
    c->ctx = ctx;
    c->device = ctx->device;
    c->rank = rank;
    c->world = world;
    ncclUniqueId id;
    std::memcpy(id.internal, id_bytes, DR_COMM_ID_BYTES);
    ncclResult_t r = rccl().comm_init_rank(&c->comm, world, id, rank);       // collective: every rank of the job calls it
    if (r != ncclSuccess) {
        delete c;
        return fail(DR_ERR_DEVICE, std::string("ncclCommInitRank: ") + rccl().error_string(r));
    }
    *out = c;
    return DR_OK;
}

void dr_comm_destroy(dr_comm* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->comm) (void)rccl().comm_destroy(c->comm);
    c->send.release();
    c->recv.release();
    delete c;
}

int dr_comm_rank(const dr_comm* c) { return c ? c->rank : -1; }
int dr_comm_world(const dr_comm* c) { return c ? c->world : 0; }

// the number of ranks RCCL itself holds in this communicator (ncclCommCount): what a report should quote, not the caller's idea of it
int dr_comm_count(const dr_comm* c, int* out_ranks) {
    if (!c || !c->comm || !out_ranks) return fail(DR_ERR_INVALID, "null argument");
    TRY(rccl_ready());
    int n = 0;
    NCCL_TRY(rccl().comm_count(c->comm, &n));
    *out_ranks = n;
    return DR_OK;
}

// all-gather of `bytes` bytes per rank between host buffers: staged through HBM, ncclAllGather on the context's stream
int dr_comm_all_gather(dr_comm* c, const void* send, size_t bytes, void* recv /* world * bytes */) {
    if (!c || !c->ctx) return fail(DR_ERR_INVALID, "null communicator");
    if (bytes == 0) return DR_OK;
    if (!send || !recv) return fail(DR_ERR_INVALID, "null buffer");
    dr_ctx* ctx = c->ctx;
    TRY(use_ctx(ctx));
    TRY(c->send.reserve(bytes));
    TRY(c->recv.reserve(bytes * (size_t)c->world));
    hipStream_t st = ctx->stream;
    HIP_TRY(hipMemcpyAsync(c->send.p, send, bytes, hipMemcpyHostToDevice, st));
    NCCL_TRY(rccl().all_gather(c->send.p, c->recv.p, bytes, ncclUint8, c->comm, st));
    HIP_TRY(hipMemcpyAsync(recv, c->recv.p, bytes * (size_t)c->world, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return DR_OK;
}

// One MSM whose bases are sharded over the ranks of `c`: this rank holds `n_local` of them (srs, from `offset`) with
// their scalars; the result — identical on every rank — is the sum of all ranks' partial MSMs.
// Every rank ALWAYS enters the all-gather: a rank whose local part fails (bad arguments, out of memory, a kernel error)
// sends status 2 in byte 96 of its record instead of leaving the others blocked in the collective, and then every rank
// returns an error — the failing one its own, the others DR_ERR_DEVICE naming the rank.
int dr_g1_msm_sharded_dev(dr_ctx* ctx, dr_comm* c, const dr_srs* srs, size_t offset, const void* d_scalars, size_t n_local,
                          uint8_t out_be_xy[96], int* is_inf) {
    if (!c || c->ctx != ctx) return fail(DR_ERR_INVALID, "communicator belongs to another context");   // no communicator to enter
    uint8_t mine[97] = {0};
    int inf = 1;
    int local_rc = out_be_xy ? DR_OK : fail(DR_ERR_INVALID, "null argument");
    std::string local_msg = local_rc == DR_OK ? std::string() : std::string(dr_last_error());
    if (local_rc == DR_OK && n_local) {
        local_rc = dr_g1_msm_dev(ctx, srs, offset, d_scalars, n_local, mine, &inf);
        if (local_rc != DR_OK) local_msg = dr_last_error();
    }
    mine[96] = local_rc != DR_OK ? 2 : (inf ? 1 : 0);
    std::vector<uint8_t> all((size_t)c->world * 97);
    TRY(dr_comm_all_gather(c, mine, 97, all.data()));
    if (local_rc != DR_OK) return fail(local_rc, "local shard of the sharded MSM failed: " + local_msg);
    std::vector<uint8_t> pts;
    for (int r = 0; r < c->world; r++) {
        uint8_t st = all[(size_t)r * 97 + 96];
        if (st > 1) return fail(DR_ERR_DEVICE, "sharded MSM: the local shard of rank " + std::to_string(r) + " failed");
        if (!st) pts.insert(pts.end(), all.begin() + (size_t)r * 97, all.begin() + (size_t)r * 97 + 96);
    }
    return dr_g1_sum(pts.data(), pts.size() / 96, out_be_xy, is_inf);
}
