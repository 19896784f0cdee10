// Native RCCL for frame-sharded solves: the ONE collective of an optimizer step,
//     ncclAllReduce(buf, buf, n, ncclDouble, ncclSum, comm, stream)
// of the packed reduced system, is issued by the library itself, stream-ordered on the context's HIP stream - no
// host language, no callback, nothing to synchronise with; a Rust (or C) host only creates the communicator.
//
// RCCL is resolved at run time, not at link time: a communicator must be driven by the RCCL instance that created it,
// and a process may already carry one that is not the system's (a PyTorch process loads its own
// torch/lib/librccl.so, soname librccl.so.1).  So: the instance already mapped into the process wins
// (dlopen RTLD_NOLOAD), otherwise librccl.so.1 from the loader's search path (/opt/rocm/lib).  CCAL_RCCL_LIB overrides.
// The declarations come from <rccl/rccl.h> (types and enum values only).
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include <rccl/rccl.h>

#include "ccal_internal.hpp"

namespace ccal {

namespace {
struct RcclApi {
    void* handle = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    bool ok = false;
};
RcclApi g_rccl;
std::once_flag g_rccl_once;

void load_rccl() {
    const char* env = std::getenv("CCAL_RCCL_LIB");
    void* h = nullptr;
    if (env && env[0]) h = dlopen(env, RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);      // the instance the process already has
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) return;
    g_rccl.handle = h;
#define SYM(field, name) g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name))
    SYM(AllReduce, "ncclAllReduce"); SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy"); SYM(CommCount, "ncclCommCount"); SYM(GetErrorString, "ncclGetErrorString");
    SYM(GetVersion, "ncclGetVersion"); SYM(CommInitAll, "ncclCommInitAll"); SYM(CommAbort, "ncclCommAbort");
#undef SYM
    g_rccl.ok = g_rccl.AllReduce && g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.GetErrorString;
}
const RcclApi* rccl() {
    std::call_once(g_rccl_once, load_rccl);
    return g_rccl.ok ? &g_rccl : nullptr;
}
}  // namespace

// In-place sum of `count` doubles over the ranks of `comm`, ordered on `st`.  Returns a ccal_status.
int rccl_allreduce_sum(ccal_ctx* ctx, void* comm, double* buf, size_t count, hipStream_t st) {
    const RcclApi* r = rccl();
    if (!r) { note_error(ctx, "RCCL is not available (librccl.so.1 not found)"); return CCAL_ERR_UNSUPPORTED; }
    const ncclResult_t e = r->AllReduce(buf, buf, count, ncclDouble, ncclSum, static_cast<ncclComm_t>(comm), st);
    if (e != ncclSuccess) {
        try { ctx->err = std::string("ncclAllReduce: ") + r->GetErrorString(e); } catch (...) { }
        return CCAL_ERR_HIP;
    }
    return CCAL_OK;
}

// One communicator per listed device, all in THIS process (single-process sharded solves, ccal_multi.hip): the n
// communicators are then driven by n host threads, one ncclAllReduce per thread and step.
int rccl_comm_init_all(const int* devices, int n, void** comms_out, std::string* err) {
    const RcclApi* r = rccl();
    if (!r || !r->CommInitAll) { if (err) *err = "RCCL is not available (librccl.so.1 / ncclCommInitAll not found)"; return CCAL_ERR_UNSUPPORTED; }
    std::vector<ncclComm_t> comms((size_t)n, nullptr);
    const ncclResult_t e = r->CommInitAll(comms.data(), n, devices);
    if (e != ncclSuccess) { if (err) *err = std::string("ncclCommInitAll: ") + r->GetErrorString(e); return CCAL_ERR_HIP; }
    for (int i = 0; i < n; ++i) comms_out[i] = comms[i];
    return CCAL_OK;
}
void rccl_comm_abort(void* comm) {
    const RcclApi* r = rccl();
    if (!r || !comm) return;
    if (r->CommAbort) (void)r->CommAbort(static_cast<ncclComm_t>(comm));
    else (void)r->CommDestroy(static_cast<ncclComm_t>(comm));
}

}  // namespace ccal

using namespace ccal;

extern "C" {

int ccal_rccl_available(void) { return rccl() ? 1 : 0; }

int ccal_rccl_version(void) {
    const RcclApi* r = rccl();
    int v = 0;
    if (!r || !r->GetVersion || r->GetVersion(&v) != ncclSuccess) return 0;
    return v;
}

int ccal_rccl_unique_id(void* id128) {
    if (!id128) return CCAL_ERR_INVALID_ARG;
    const RcclApi* r = rccl();
    if (!r) return CCAL_ERR_UNSUPPORTED;
    ncclUniqueId id;
    if (r->GetUniqueId(&id) != ncclSuccess) return CCAL_ERR_HIP;
    static_assert(sizeof(id) == CCAL_RCCL_UNIQUE_ID_BYTES, "ncclUniqueId size");
    std::memcpy(id128, &id, sizeof id);
    return CCAL_OK;
}

int ccal_rccl_comm_create(ccal_ctx* ctx, int world, int rank, const void* id128, void** comm_out) {
    if (!ctx || !id128 || !comm_out || world < 1 || rank < 0 || rank >= world) return CCAL_ERR_INVALID_ARG;
    *comm_out = nullptr;
    CCAL_API_TRY
    const RcclApi* r = rccl();
    if (!r) { ctx->err = "RCCL is not available (librccl.so.1 not found)"; return CCAL_ERR_UNSUPPORTED; }
    if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return CCAL_ERR_HIP; }
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    ncclComm_t comm = nullptr;
    const ncclResult_t e = r->CommInitRank(&comm, world, id, rank);
    if (e != ncclSuccess) { ctx->err = std::string("ncclCommInitRank: ") + r->GetErrorString(e); return CCAL_ERR_HIP; }
    *comm_out = comm;
    return CCAL_OK;
    CCAL_API_CATCH(ctx)
}

int ccal_rccl_comm_destroy(void* comm) {
    if (!comm) return CCAL_OK;
    const RcclApi* r = rccl();
    if (!r) return CCAL_ERR_UNSUPPORTED;
    return r->CommDestroy(static_cast<ncclComm_t>(comm)) == ncclSuccess ? CCAL_OK : CCAL_ERR_HIP;
}

int ccal_rccl_comm_count(void* comm) {
    const RcclApi* r = rccl();
    int n = 0;
    if (!comm || !r || !r->CommCount || r->CommCount(static_cast<ncclComm_t>(comm), &n) != ncclSuccess) return -1;
    return n;
}

int ccal_set_rccl_comm(ccal_problem* p, void* nccl_comm) {
    if (!p) return CCAL_ERR_INVALID_ARG;
    if (nccl_comm && !rccl()) { note_error(p->ctx, "RCCL is not available (librccl.so.1 not found)"); return CCAL_ERR_UNSUPPORTED; }
    // early-exit groups of the last solve (each with its collective on the old communicator) may still be queued: they
    // must have run before the caller may destroy that communicator
    if (p->rccl_comm != nccl_comm) {
        int rc = drain_pending_groups(p);
        if (rc != CCAL_OK) return rc;
    }
    p->rccl_comm = nccl_comm;
    return CCAL_OK;
}

}  // extern "C"
