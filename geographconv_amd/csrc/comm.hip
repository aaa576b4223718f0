// C-ABI collectives for the partitioned step -- RCCL over xGMI, one process per GPU.
// The reference is single-process (gcnmodel.py:409-430 is the step being partitioned) and has no counterpart; these are
// the entry points a host WITHOUT torch.distributed binds (the Python host of this repository drives the same RCCL
// through torch.distributed by default and through these with GEOGCN_DIST_BACKEND=native).
// RCCL is resolved at run time (dlopen): libgeogcn.so has no link dependency on it, loads on machines without it, and
// inside a process that already carries an RCCL (PyTorch ships its own) the loaded copy is reused, never a second one.
// It also BUILDS without the RCCL headers: only a handful of opaque types and enum values are needed, declared below
// (NCCL's stable public ABI) when <rccl/rccl.h> is absent.
#include "common.h"

#include <dlfcn.h>
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>      // types and enums only; every function is looked up below
#else
extern "C" {
typedef struct ncclComm* ncclComm_t;
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
typedef enum { ncclChar = 0, ncclFloat = 7 } ncclDataType_t;
}
#endif

#include <mutex>

struct geogcn_comm {
    ncclComm_t comm = nullptr;
    int world = 0, rank = 0;
};

namespace geogcn {
namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    char why[256] = {0};
};

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // a copy already in the process first (RTLD_NOLOAD), then the system one
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names)
            if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL);
        for (const char* n : names)
            if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!r.lib) {
            snprintf(r.why, sizeof r.why, "RCCL not found (%s)", dlerror());
            return;
        }
        bool ok = true;
#define GEOGCN_SYM(field, name)                                                   \
    do {                                                                          \
        r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, name));        \
        if (!r.field) { ok = false; snprintf(r.why, sizeof r.why, "RCCL lacks %s", name); } \
    } while (0)
        GEOGCN_SYM(GetUniqueId, "ncclGetUniqueId");
        GEOGCN_SYM(CommInitRank, "ncclCommInitRank");
        GEOGCN_SYM(CommDestroy, "ncclCommDestroy");
        GEOGCN_SYM(AllReduce, "ncclAllReduce");
        GEOGCN_SYM(AllGather, "ncclAllGather");
        GEOGCN_SYM(Send, "ncclSend");
        GEOGCN_SYM(Recv, "ncclRecv");
        GEOGCN_SYM(GroupStart, "ncclGroupStart");
        GEOGCN_SYM(GroupEnd, "ncclGroupEnd");
        GEOGCN_SYM(GetErrorString, "ncclGetErrorString");
#undef GEOGCN_SYM
        if (!ok) r.lib = nullptr;
    });
    return r.lib ? &r : nullptr;
}
// RCCL results are passed through as 1000 + ncclResult_t (> 0 like a hipError_t, distinguishable from one)
#define GEOGCN_NCCL(R, call)                                                                   \
    do {                                                                                       \
        ncclResult_t e__ = (call);                                                             \
        if (e__ != ncclSuccess) {                                                              \
            geogcn::set_error("%s: %s", #call, (R)->GetErrorString(e__));                      \
            return 1000 + (int)e__;                                                            \
        }                                                                                      \
    } while (0)

}  // namespace
}  // namespace geogcn

using namespace geogcn;

extern "C" {

int geogcn_comm_available(void) { return rccl() ? 1 : 0; }

int geogcn_comm_unique_id(void* id_out, size_t id_bytes) {
    GEOGCN_REQUIRE(id_out, GEOGCN_E_NULL, "comm_unique_id: null output");
    GEOGCN_REQUIRE(id_bytes >= GEOGCN_COMM_ID_BYTES, GEOGCN_E_SIZE, "comm_unique_id: buffer of %zu bytes, need %d", id_bytes,
                   GEOGCN_COMM_ID_BYTES);
    Rccl* r = rccl();
    GEOGCN_REQUIRE(r, GEOGCN_E_ARG, "comm_unique_id: RCCL is not available in this process");
    static_assert(sizeof(ncclUniqueId) == GEOGCN_COMM_ID_BYTES, "unique id size");
    ncclUniqueId id;
    GEOGCN_NCCL(r, r->GetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return 0;
}

int geogcn_comm_init_rank(const void* id, int32_t world, int32_t rank, geogcn_comm** out) {
    GEOGCN_REQUIRE(id && out, GEOGCN_E_NULL, "comm_init_rank: null argument");
    GEOGCN_REQUIRE(world >= 1 && rank >= 0 && rank < world, GEOGCN_E_SIZE, "comm_init_rank: rank %d of %d", rank, world);
    Rccl* r = rccl();
    GEOGCN_REQUIRE(r, GEOGCN_E_ARG, "comm_init_rank: RCCL is not available in this process");
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    ncclComm_t c = nullptr;
    GEOGCN_NCCL(r, r->CommInitRank(&c, world, uid, rank));         // binds the calling thread's current HIP device
    geogcn_comm* h = new geogcn_comm;
    h->comm = c;
    h->world = world;
    h->rank = rank;
    *out = h;
    return 0;
}

void geogcn_comm_destroy(geogcn_comm* comm) {
    if (!comm) return;
    Rccl* r = rccl();
    if (r && comm->comm) r->CommDestroy(comm->comm);
    delete comm;
}

int32_t geogcn_comm_world(const geogcn_comm* comm) { return comm ? comm->world : 0; }
int32_t geogcn_comm_rank(const geogcn_comm* comm) { return comm ? comm->rank : -1; }

int geogcn_comm_allreduce_sum_f32(geogcn_comm* comm, float* buf, int64_t n, void* stream) {
    GEOGCN_REQUIRE(comm && comm->comm, GEOGCN_E_NULL, "comm_allreduce_sum_f32: null communicator");
    GEOGCN_REQUIRE(n >= 0, GEOGCN_E_SIZE, "comm_allreduce_sum_f32: n = %lld", (long long)n);
    if (n == 0) return 0;
    GEOGCN_REQUIRE(buf, GEOGCN_E_NULL, "comm_allreduce_sum_f32: null buffer");
    Rccl* r = rccl();
    GEOGCN_NCCL(r, r->AllReduce(buf, buf, (size_t)n, ncclFloat, ncclSum, comm->comm, (hipStream_t)stream));
    return 0;
}

int geogcn_comm_allgather(geogcn_comm* comm, const void* send, void* recv, int64_t bytes_per_rank, void* stream) {
    GEOGCN_REQUIRE(comm && comm->comm, GEOGCN_E_NULL, "comm_allgather: null communicator");
    GEOGCN_REQUIRE(bytes_per_rank >= 0, GEOGCN_E_SIZE, "comm_allgather: %lld bytes per rank", (long long)bytes_per_rank);
    if (bytes_per_rank == 0) return 0;
    GEOGCN_REQUIRE(send && recv, GEOGCN_E_NULL, "comm_allgather: null buffer");
    Rccl* r = rccl();
    GEOGCN_NCCL(r, r->AllGather(send, recv, (size_t)bytes_per_rank, ncclChar, comm->comm, (hipStream_t)stream));
    return 0;
}

int geogcn_comm_alltoall(geogcn_comm* comm, const void* send, void* recv, int64_t bytes_per_peer, void* stream) {
    GEOGCN_REQUIRE(comm && comm->comm, GEOGCN_E_NULL, "comm_alltoall: null communicator");
    GEOGCN_REQUIRE(bytes_per_peer >= 0, GEOGCN_E_SIZE, "comm_alltoall: %lld bytes per peer", (long long)bytes_per_peer);
    if (bytes_per_peer == 0) return 0;
    GEOGCN_REQUIRE(send && recv && send != recv, GEOGCN_E_NULL, "comm_alltoall: null or aliased buffers");
    Rccl* r = rccl();
    // xGMI is point to point: every peer pair is its own link, so the exchange is W - 1 concurrent sends and receives
    // in one group (what RCCL's own all-to-all does); the own panel goes through the same path (a device copy)
    GEOGCN_NCCL(r, r->GroupStart());
    for (int p = 0; p < comm->world; ++p) {
        const char* s = (const char*)send + (size_t)p * (size_t)bytes_per_peer;
        char* d = (char*)recv + (size_t)p * (size_t)bytes_per_peer;
        ncclResult_t e1 = r->Send(s, (size_t)bytes_per_peer, ncclChar, p, comm->comm, (hipStream_t)stream);
        ncclResult_t e2 = e1 == ncclSuccess ? r->Recv(d, (size_t)bytes_per_peer, ncclChar, p, comm->comm, (hipStream_t)stream) : e1;
        if (e2 != ncclSuccess) {
            r->GroupEnd();
            geogcn::set_error("comm_alltoall: peer %d: %s", p, r->GetErrorString(e2));
            return 1000 + (int)e2;
        }
    }
    GEOGCN_NCCL(r, r->GroupEnd());
    return 0;
}

int geogcn_comm_alltoallv(geogcn_comm* comm, const void* send, const int64_t* send_bytes, void* recv, const int64_t* recv_bytes,
                          void* stream) {
    GEOGCN_REQUIRE(comm && comm->comm, GEOGCN_E_NULL, "comm_alltoallv: null communicator");
    GEOGCN_REQUIRE(send_bytes && recv_bytes, GEOGCN_E_NULL, "comm_alltoallv: null size vectors");
    int64_t s_total = 0, r_total = 0;
    for (int p = 0; p < comm->world; ++p) {
        GEOGCN_REQUIRE(send_bytes[p] >= 0 && recv_bytes[p] >= 0, GEOGCN_E_SIZE, "comm_alltoallv: negative size for peer %d", p);
        s_total += send_bytes[p];
        r_total += recv_bytes[p];
    }
    GEOGCN_REQUIRE((send || s_total == 0) && (recv || r_total == 0), GEOGCN_E_NULL, "comm_alltoallv: null buffer");
    GEOGCN_REQUIRE(send != recv || (s_total == 0 && r_total == 0), GEOGCN_E_NULL, "comm_alltoallv: aliased buffers");
    Rccl* r = rccl();
    // the halo exchange: rank p gets the rows of mine its block of A_hat references, and nothing else; pieces are laid
    // out back to back in peer order on both sides.  Every rank calls with its own vectors (a rank with nothing to
    // send or receive still enters the group: its peers' sizes for it are zero as well, by symmetry of the lists)
    GEOGCN_NCCL(r, r->GroupStart());
    size_t so = 0, ro = 0;
    for (int p = 0; p < comm->world; ++p) {
        ncclResult_t e = ncclSuccess;
        if (send_bytes[p] > 0) e = r->Send((const char*)send + so, (size_t)send_bytes[p], ncclChar, p, comm->comm, (hipStream_t)stream);
        if (e == ncclSuccess && recv_bytes[p] > 0)
            e = r->Recv((char*)recv + ro, (size_t)recv_bytes[p], ncclChar, p, comm->comm, (hipStream_t)stream);
        if (e != ncclSuccess) {
            r->GroupEnd();
            geogcn::set_error("comm_alltoallv: peer %d: %s", p, r->GetErrorString(e));
            return 1000 + (int)e;
        }
        so += (size_t)send_bytes[p];
        ro += (size_t)recv_bytes[p];
    }
    GEOGCN_NCCL(r, r->GroupEnd());
    return 0;
}

}  // extern "C"
