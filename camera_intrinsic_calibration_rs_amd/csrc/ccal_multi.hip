// Single-process multi-GPU (include/ccal.h, "one process, several GPUs"): the reference is ONE process whose
// calib_camera is one blocking call (src/util.rs:384-390, src/bin/camera_calibration.rs:70), so the drop-in for it must
// reach all GPUs of a node from one call too.  A ccal_multi is a set of contexts - one per listed device, each with its
// own stream and, inside ccal_solve_sharded, its own host thread - plus the transport of the step's one all-reduce:
//   * devices all different and RCCL there:  one communicator per device from ncclCommInitAll, the library issues
//     ncclAllReduce on every context's stream (ccal_rccl.hip) - the production path over xGMI;
//   * otherwise (a device listed more than once - how the 1-GPU test box runs the sharded path - or no RCCL but peer
//     access): the IN-PROCESS transport below.  Stream-ordered like RCCL: nothing waits on the host for the device.
// A ccal_multi_problem is one problem description sharded by contiguous frame-slot ranges, balanced by corner count, all
// cameras' observations of a slot on one shard; its entry points mirror the single-GPU ones.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>

#include "ccal_fused.hpp"

using namespace ccal;

// ---------------------------------------------------------------------------------------------------------------------
// In-process transport (v2, round 5): the sum over the ranks is taken by the DECIDING kernel itself.  Every rank's reduce
// kernel leaves its packed sums in one of two buffers of its own (alternating by collective); the rank then POSTS the
// buffer - records an event behind the reduce on its stream and publishes pointer + sequence number - and makes its
// stream wait for the peers' events of the same collective; its k_head / k_solve receives all ranks' buffer addresses
// in its argument block and adds them in rank order (the same bits on every rank).  Per collective and rank: ONE event
// record, ONE host rendezvous (per-rank posted counters, no central barrier), n - 1 stream waits - and no launch at all
// (v1: k_sum_ranks + k_copy_sum as launches of their own, two host barriers, 2 (n - 1) waits).
// Why two buffers are enough: rank q overwrites buffer b at collective c + 2; that reduce sits behind q's deciding kernel
// of collective c + 1 in q's stream, which waited for the event rank r recorded BEHIND its reduce of c + 1 - and that
// is behind r's deciding kernel of collective c, the last reader of q's buffer b.
// The host side only orders event records against waits (a wait captures the record made before it) and never waits for
// the device, so groups are enqueued ahead exactly as with RCCL.  A rank that fails sets `abort`: peers spinning in the
// rendezvous come back with an error instead of hanging; the rendezvous has a timeout of its own.
// ---------------------------------------------------------------------------------------------------------------------
namespace ccal {

constexpr int kInprocMaxRanks = CCAL_MULTI_MAX_DEVICES;
static_assert(kInprocMaxRanks == kMaxPeers, "PeerView holds one pointer per rank");

struct InprocRank { InprocComm* c; int rank; };
struct InprocComm {
    int n = 0;
    std::vector<int> device;
    std::vector<hipEvent_t> ready;                   // [parity][rank]
    std::vector<const double*> src;                  // [parity][rank]: what the rank posted for the collective of that parity
    std::vector<size_t> cnt;                         // [parity][rank]
    std::vector<uint64_t> calls;                     // [rank]: collectives this rank has begun (its thread only)
    std::unique_ptr<std::atomic<uint64_t>[]> posted; // [rank]: collectives whose event record + pointer are published
    std::vector<InprocRank> handles;
    std::atomic<int> abort{0}, sleepers{0};
    std::mutex m; std::condition_variable cv;        // back-off of a rank that has spun for ~50 us (a peer is uploading, or slow)
    double timeout_s = 600.0;
};

static inline void relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
}
// wait until rank q has posted collective number `call`.  Spins ~50 us (the peers of a step arrive within microseconds of each
// other), then sleeps on the condition variable in 200-us slices.  false: aborted (by a peer, or by this rank after `timeout_s`).
static bool inproc_wait_posted(InprocComm* c, int q, uint64_t call) {
    std::atomic<uint64_t>& pq = c->posted[(size_t)q];
    const auto t0 = std::chrono::steady_clock::now();
    for (long spins = 1;; ++spins) {
        if (pq.load(std::memory_order_acquire) >= call) return !c->abort.load(std::memory_order_acquire);
        if (c->abort.load(std::memory_order_acquire)) return false;
        if (spins & 0xFF) { relax(); continue; }
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (c->timeout_s > 0 && el > c->timeout_s) { inproc_abort(c); return false; }
        if (el > 50e-6) {
            std::unique_lock<std::mutex> lk(c->m);
            // store(posted) -> load(sleepers) on the poster's side against add(sleepers) -> load(posted) here is a store-load pair
            // on each side: only sequentially consistent operations order it (with release / acquire the poster could read
            // sleepers == 0 while this thread still read the old `posted`, and a wake-up was lost for a 200-us slice)
            c->sleepers.fetch_add(1, std::memory_order_seq_cst);
            c->cv.wait_for(lk, std::chrono::microseconds(200), [&] { return pq.load(std::memory_order_seq_cst) >= call || c->abort.load(std::memory_order_seq_cst) != 0; });
            c->sleepers.fetch_sub(1, std::memory_order_acq_rel);
        }
    }
}

InprocComm* inproc_create(int n, const int* devices, std::string* err) {
    auto fail = [&](const std::string& m) -> InprocComm* { if (err) *err = m; return nullptr; };
    if (n < 1 || n > kInprocMaxRanks) return fail("in-process transport: 1 .. 16 ranks");
    // peer access between every pair of different devices (same device: nothing to enable)
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < n; ++k) {
            if (devices[i] == devices[k]) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devices[i], devices[k]) != hipSuccess || !can)
                return fail("in-process transport: no peer access between device " + std::to_string(devices[i]) + " and " + std::to_string(devices[k]));
            if (hipSetDevice(devices[i]) != hipSuccess) return fail("hipSetDevice failed");
            const hipError_t e = hipDeviceEnablePeerAccess(devices[k], 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return fail(std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
            (void)hipGetLastError();
        }
    std::unique_ptr<InprocComm> c(new InprocComm());
    c->n = n;
    c->device.assign(devices, devices + n);
    c->ready.assign((size_t)2 * n, nullptr);
    c->src.assign((size_t)2 * n, nullptr); c->cnt.assign((size_t)2 * n, 0); c->calls.assign((size_t)n, 0);
    c->posted.reset(new std::atomic<uint64_t>[(size_t)n]);
    for (int r = 0; r < n; ++r) c->posted[(size_t)r].store(0);
    c->handles.resize((size_t)n);
    for (int r = 0; r < n; ++r) {
        c->handles[r] = InprocRank{ c.get(), r };
        bool ok = hipSetDevice(devices[r]) == hipSuccess;
        for (int par = 0; par < 2 && ok; ++par) ok = hipEventCreateWithFlags(&c->ready[par * n + r], hipEventDisableTiming) == hipSuccess;
        if (!ok) { InprocComm* raw = c.release(); inproc_destroy(raw); return fail("in-process transport: event creation failed"); }
    }
    return c.release();
}
void inproc_destroy(InprocComm* c) {
    if (!c) return;
    for (int r = 0; r < c->n; ++r) {
        (void)hipSetDevice(c->device[r]);
        for (int par = 0; par < 2; ++par) if (c->ready[par * c->n + r]) (void)hipEventDestroy(c->ready[par * c->n + r]);
    }
    delete c;
}
void inproc_abort(InprocComm* c) {
    if (!c) return;
    c->abort.store(1, std::memory_order_seq_cst);
    if (c->sleepers.load(std::memory_order_seq_cst) > 0) { std::lock_guard<std::mutex> lk(c->m); c->cv.notify_all(); }
}
bool inproc_aborted(const InprocComm* c) { return c && c->abort.load(std::memory_order_acquire) != 0; }
void inproc_set_timeout(InprocComm* c, double seconds) { if (c) c->timeout_s = seconds > 0 ? seconds : 600.0; }
void* inproc_rank_handle(InprocComm* c, int rank) { return &c->handles[(size_t)rank]; }
int inproc_recover(InprocComm* c) {
    if (!c) return CCAL_OK;
    // no rank is inside a collective any more (their threads have returned): whatever they enqueued runs to its end - a
    // wait on an event that was never recorded is no wait - then the rendezvous counters start over
    for (int r = 0; r < c->n; ++r)
        if (hipSetDevice(c->device[r]) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return CCAL_ERR_HIP;
    c->abort.store(0);
    std::fill(c->calls.begin(), c->calls.end(), 0);
    for (int r = 0; r < c->n; ++r) c->posted[(size_t)r].store(0);
    return CCAL_OK;
}

// which of its two buffers the rank's NEXT collective uses (the reduce kernel in front of inproc_post writes there)
int inproc_parity(const void* user) {
    const InprocRank* h = static_cast<const InprocRank*>(user);
    return (int)(h->c->calls[(size_t)h->rank] & 1);
}
// Post `buf` (count doubles, complete once everything enqueued on `hip_stream` so far has run) as this rank's contribution to
// its next collective, make the stream wait for every peer's contribution, and hand back all ranks' buffers in rank order.
// Returns 0, or 1 after an abort / timeout / HIP error (the transport is aborted then).
int inproc_post(void* user, const double* buf, size_t count, void* hip_stream, PeerView* out) {
    InprocRank* h = static_cast<InprocRank*>(user);
    InprocComm* c = h->c;
    const int r = h->rank, n = c->n;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    if (c->abort.load(std::memory_order_acquire)) return 1;
    const int par = (int)(c->calls[(size_t)r] & 1);
    const uint64_t call = ++c->calls[(size_t)r];
    c->src[(size_t)(par * n + r)] = buf; c->cnt[(size_t)(par * n + r)] = count;
    if (hipEventRecord(c->ready[(size_t)(par * n + r)], st) != hipSuccess) { inproc_abort(c); return 1; }
    c->posted[(size_t)r].store(call, std::memory_order_seq_cst);
    if (c->sleepers.load(std::memory_order_seq_cst) > 0) { std::lock_guard<std::mutex> lk(c->m); c->cv.notify_all(); }
    out->n = n;
    for (int q = 0; q < n; ++q) {
        if (q == r) { out->src[q] = buf; continue; }
        if (!inproc_wait_posted(c, q, call)) return 1;
        // every rank posts the same buffer of the same step; the peer's record precedes this wait (acquire above)
        if (c->cnt[(size_t)(par * n + q)] != count || hipStreamWaitEvent(st, c->ready[(size_t)(par * n + q)], 0) != hipSuccess) { inproc_abort(c); return 1; }
        out->src[q] = c->src[(size_t)(par * n + q)];
    }
    for (int q = n; q < kMaxPeers; ++q) out->src[q] = nullptr;
    return 0;
}

}  // namespace ccal

// ---------------------------------------------------------------------------------------------------------------------
struct ccal_multi {
    int n = 0;
    std::vector<int> device;
    std::vector<ccal_ctx*> ctx;
    int transport = CCAL_TRANSPORT_NONE;
    std::vector<void*> comms;                  // RCCL: one ncclComm_t per device
    InprocComm* inproc = nullptr;
    bool broken = false;                       // a sharded solve failed with the RCCL communicators in an unknown state
    std::string err;
    int n_problems = 0;
    bool destroy_requested = false;
};
struct ccal_multi_problem {
    ccal_multi* m = nullptr;
    int n_slots = 0, n_obs = 0, n_cams = 0;
    int64_t n_corners = 0;
    std::vector<ccal_problem*> shard;
    std::vector<int32_t> first;                // [n + 1] slot range of every shard
    std::vector<std::vector<int32_t>> obs_of;  // shard -> its observation frames (indices into the caller's description)
    // ccal_multi_validation: where the shards' errors of one camera meet on the first shard's GPU - [values |
    // selection's work area], grown on demand and kept between calls (like ccal_problem::d_scratch for the single-GPU validation())
    char* d_gather = nullptr; size_t gather_bytes = 0;
};

static void multi_free(ccal_multi* m) {
    if (m->inproc) inproc_destroy(m->inproc);
    for (void* c : m->comms) if (c) { if (m->broken) rccl_comm_abort(c); else (void)ccal_rccl_comm_destroy(c); }
    for (ccal_ctx* c : m->ctx) if (c) ccal_ctx_destroy(c);
    delete m;
}
static int mfail(ccal_multi* m, int code, const std::string& msg) { if (m) { try { m->err = msg; } catch (...) { } } return code; }
// run fn(i) for every shard: shard 0 on the caller's thread, the others on their contexts' persistent helper threads (the ones
// ccal_multi_solve / ccal_solve_batch use: created on first use, asleep on a condition variable in between)
template <class F>
static void for_each_shard(ccal_multi* m, int n, F&& fn) {
    int started = 0;
    try { for (int i = 1; i < n; ++i) { ctx_worker_submit(m->ctx[(size_t)i], [&fn, i] { fn(i); }); started = i; } }
    catch (...) { for (int i = 1; i <= started; ++i) ctx_worker_wait(m->ctx[(size_t)i]); throw; }
    fn(0);
    for (int i = 1; i < n; ++i) ctx_worker_wait(m->ctx[(size_t)i]);
}
// why the last ccal_ctx_create / ccal_multi_create* of this thread failed (there is no handle to ask)
static thread_local std::string t_create_err;
namespace ccal { void note_create_error(const std::string& msg) noexcept { try { t_create_err = msg; } catch (...) { } } }

extern "C" {

const char* ccal_create_last_error(void) { return t_create_err.c_str(); }

// transport: -1 = automatic (RCCL when the devices are all different and RCCL can be resolved, else the in-process transport;
// the environment variable CCAL_MULTI_TRANSPORT=inproc|rccl overrides the automatic choice), CCAL_TRANSPORT_RCCL = communicators
// even for ONE device (how a 1-GPU box runs ncclCommInitAll and the library-issued collective), CCAL_TRANSPORT_INPROC = the
// in-process transport even when RCCL is there.  A transport asked for by name is not replaced by another one when it fails.
int ccal_multi_create_transport(const int* device_ids, int n_dev, int transport, ccal_multi** out) {
    if (!out) return CCAL_ERR_INVALID_ARG;
    *out = nullptr;
    note_create_error("");
    if (!device_ids || n_dev < 1 || n_dev > CCAL_MULTI_MAX_DEVICES || transport < -1 || transport > CCAL_TRANSPORT_INPROC) { note_create_error("ccal_multi_create: bad arguments"); return CCAL_ERR_INVALID_ARG; }
    CCAL_API_TRY
    std::unique_ptr<ccal_multi> m(new ccal_multi());
    m->n = n_dev;
    m->device.assign(device_ids, device_ids + n_dev);
    m->ctx.assign((size_t)n_dev, nullptr);
    auto bail = [&](int rc, const std::string& why) { note_create_error(why); ccal_multi* raw = m.release(); multi_free(raw); return rc; };
    for (int i = 0; i < n_dev; ++i) {
        const int rc = ccal_ctx_create(device_ids[i], nullptr, &m->ctx[i]);
        if (rc != CCAL_OK) return bail(rc, "ccal_multi_create: no context on device " + std::to_string(device_ids[i]) + " (" + t_create_err + ")");
    }
    if (transport < 0) {
        const char* force = std::getenv("CCAL_MULTI_TRANSPORT");
        if (force && force[0] == 'r') transport = CCAL_TRANSPORT_RCCL;
        else if (force && force[0] == 'i') transport = CCAL_TRANSPORT_INPROC;
    }
    if (n_dev == 1 && transport == CCAL_TRANSPORT_INPROC)      // asked for by name: never replaced silently by "none"
        return bail(CCAL_ERR_UNSUPPORTED, "ccal_multi_create: the in-process transport needs at least two shards (list the device twice)");
    if (n_dev > 1 || transport == CCAL_TRANSPORT_RCCL) {
        bool distinct = true;
        for (int i = 0; i < n_dev; ++i) for (int k = 0; k < i; ++k) distinct = distinct && device_ids[i] != device_ids[k];
        std::string err_rccl, err_inproc;
        const bool want_rccl = transport == CCAL_TRANSPORT_RCCL || (transport < 0 && distinct && ccal_rccl_available());
        if (want_rccl) {
            if (!distinct) err_rccl = "RCCL needs every device listed once";
            else {
                m->comms.assign((size_t)n_dev, nullptr);
                const int rc = rccl_comm_init_all(device_ids, n_dev, m->comms.data(), &err_rccl);
                if (rc == CCAL_OK) m->transport = CCAL_TRANSPORT_RCCL;
                else {                                       // whatever communicators the failed call did make must not leak
                    for (void*& c : m->comms) if (c) { rccl_comm_abort(c); c = nullptr; }
                    m->comms.clear();
                }
            }
            if (m->transport != CCAL_TRANSPORT_RCCL && transport == CCAL_TRANSPORT_RCCL) return bail(CCAL_ERR_UNSUPPORTED, "ccal_multi_create: RCCL transport: " + err_rccl);
        }
        if (m->transport == CCAL_TRANSPORT_NONE && n_dev > 1) {      // (automatic: peer access may still do where RCCL did not)
            m->inproc = inproc_create(n_dev, device_ids, &err_inproc);
            if (!m->inproc) return bail(CCAL_ERR_UNSUPPORTED, "ccal_multi_create: " + err_inproc + (err_rccl.empty() ? std::string("; RCCL not tried") : "; RCCL: " + err_rccl));
            m->transport = CCAL_TRANSPORT_INPROC;
            if (!err_rccl.empty()) m->err = "RCCL transport not used: " + err_rccl;       // readable through ccal_multi_last_error
        }
    }
    *out = m.release();
    return CCAL_OK;
    CCAL_API_CATCH((ccal_ctx*)nullptr)
}
int ccal_multi_create(const int* device_ids, int n_dev, ccal_multi** out) { return ccal_multi_create_transport(device_ids, n_dev, -1, out); }
// ranks of the device set's communicators as RCCL counts them (ncclCommCount of the first one); 0: the transport is not RCCL
int ccal_multi_rccl_ranks(const ccal_multi* m) {
    if (!m) return -1;
    if (m->transport != CCAL_TRANSPORT_RCCL || m->comms.empty() || !m->comms[0]) return 0;
    return ccal_rccl_comm_count(m->comms[0]);
}
void ccal_multi_destroy(ccal_multi* m) {
    if (!m) return;
    if (m->n_problems > 0) { m->destroy_requested = true; return; }       // freed by its last problem
    multi_free(m);
}
int ccal_multi_num_devices(const ccal_multi* m) { return m ? m->n : -1; }
int ccal_multi_transport(const ccal_multi* m) { return m ? m->transport : -1; }
ccal_ctx* ccal_multi_ctx(ccal_multi* m, int i) { return (m && i >= 0 && i < m->n) ? m->ctx[i] : nullptr; }
const char* ccal_multi_last_error(const ccal_multi* m) { return m ? m->err.c_str() : "null device set"; }
int ccal_multi_set_model_conventions(ccal_multi* m, const ccal_model_conventions* in) {
    if (!m) return CCAL_ERR_INVALID_ARG;
    for (ccal_ctx* c : m->ctx) { const int rc = ccal_set_model_conventions(c, in); if (rc != CCAL_OK) return rc; }
    return CCAL_OK;
}
int ccal_multi_sync(ccal_multi* m) {
    if (!m) return CCAL_ERR_INVALID_ARG;
    for (ccal_ctx* c : m->ctx) { const int rc = ccal_sync(c); if (rc != CCAL_OK) return mfail(m, rc, ccal_last_error(c)); }
    return CCAL_OK;
}

void ccal_multi_problem_destroy(ccal_multi_problem* mp) {
    if (!mp) return;
    for (ccal_problem* p : mp->shard) if (p) {
        // the transport outlives its users: early-exit groups still queued hold its events / communicator
        (void)drain_pending_groups(p);
        ccal_problem_destroy(p);
    }
    ccal_multi* m = mp->m;
    if (mp->d_gather && m) { (void)hipSetDevice(m->ctx[0]->device); (void)hipFree(mp->d_gather); }
    delete mp;
    if (m && --m->n_problems == 0 && m->destroy_requested) multi_free(m);
}

// Where a description is cut: contiguous slot ranges balanced by corner count - boundary r = the first slot at which the corners
// of the slots before it reach r / n of all corners ("balanced by corner count (CSR offsets)"); all cameras'
// observations of a slot stay on one shard.  Pure host code (ccal_partition_slots: the same cut for one process per GPU).
static int partition_slots(const ccal_problem_desc* d, int n, std::vector<int32_t>& first, int64_t* total_out, const char** why) {
    if (d->n_slots < 0 || d->n_obs < 0 || (d->n_obs > 0 && (!d->obs_slot || !d->obs_offsets))) { *why = "bad problem description"; return CCAL_ERR_INVALID_ARG; }
    std::vector<int64_t> before((size_t)d->n_slots + 1, 0);
    for (int o = 0; o < d->n_obs; ++o) {
        const int s = d->obs_slot[o];
        const int64_t c = d->obs_offsets[o + 1] - d->obs_offsets[o];
        if (s < 0 || s >= d->n_slots || c < 0) { *why = "bad observation frame table"; return CCAL_ERR_INVALID_ARG; }
        before[(size_t)s + 1] += c;
    }
    for (int s = 0; s < d->n_slots; ++s) before[(size_t)s + 1] += before[(size_t)s];
    const int64_t total = before[(size_t)d->n_slots];
    first.assign((size_t)n + 1, 0);
    first[(size_t)n] = d->n_slots;
    for (int r = 1; r < n; ++r) {
        int s;
        if (total > 0) {
            const int64_t target = (int64_t)(((__int128)total * r) / n);
            s = (int)(std::lower_bound(before.begin(), before.end(), target) - before.begin());
        } else s = (int)((int64_t)d->n_slots * r / n);
        first[(size_t)r] = std::min(std::max(s, first[(size_t)r - 1]), d->n_slots);
    }
    if (total_out) *total_out = total;
    return CCAL_OK;
}
int ccal_partition_slots(const ccal_problem_desc* d, int n_shards, int32_t* first_out) {
    if (!d || !first_out || n_shards < 1) return CCAL_ERR_INVALID_ARG;
    CCAL_API_TRY
    std::vector<int32_t> first;
    const char* why = "";
    const int rc = partition_slots(d, n_shards, first, nullptr, &why);
    if (rc != CCAL_OK) return rc;
    std::memcpy(first_out, first.data(), ((size_t)n_shards + 1) * sizeof(int32_t));
    return CCAL_OK;
    CCAL_API_CATCH((ccal_ctx*)nullptr)
}

int ccal_multi_problem_create(ccal_multi* m, const ccal_problem_desc* d, ccal_multi_problem** out) {
    if (!m || !d || !out) return CCAL_ERR_INVALID_ARG;
    *out = nullptr;
    CCAL_API_TRY
    if (d->n_cams < 1 || d->n_cams > CCAL_MAX_CAMS || d->n_slots < 0 || d->n_obs < 0 || !d->model ||
        (d->n_obs > 0 && (!d->obs_cam || !d->obs_slot || !d->obs_offsets)))
        return mfail(m, CCAL_ERR_INVALID_ARG, "bad problem description");
    const int n = m->n;
    std::unique_ptr<ccal_multi_problem> mp(new ccal_multi_problem());
    int64_t total = 0;
    { const char* why = ""; const int rc = partition_slots(d, n, mp->first, &total, &why); if (rc != CCAL_OK) return mfail(m, rc, why); }
    mp->m = m; mp->n_slots = d->n_slots; mp->n_obs = d->n_obs; mp->n_cams = d->n_cams; mp->n_corners = total;
    mp->obs_of.assign((size_t)n, {});
    std::vector<int> shard_of_slot((size_t)std::max(d->n_slots, 1), 0);
    for (int r = 0; r < n; ++r) for (int s = mp->first[(size_t)r]; s < mp->first[(size_t)r + 1]; ++s) shard_of_slot[(size_t)s] = r;
    for (int o = 0; o < d->n_obs; ++o) mp->obs_of[(size_t)shard_of_slot[(size_t)d->obs_slot[o]]].push_back(o);
    mp->shard.assign((size_t)n, nullptr);
    m->n_problems += 1;                                       // from here on ccal_multi_problem_destroy undoes everything
    auto bail = [&](int rc, const std::string& msg) { ccal_multi_problem* raw = mp.release(); m->err = msg; ccal_multi_problem_destroy(raw); return rc; };
    for (int r = 0; r < n; ++r) {
        const std::vector<int32_t>& obs = mp->obs_of[(size_t)r];
        std::vector<int32_t> cam(obs.size()), slot(obs.size());
        std::vector<int64_t> off(obs.size() + 1, 0);
        for (size_t i = 0; i < obs.size(); ++i) {
            const int o = obs[i];
            cam[i] = d->obs_cam[o]; slot[i] = d->obs_slot[o] - mp->first[(size_t)r];
            off[i + 1] = off[i] + (d->obs_offsets[o + 1] - d->obs_offsets[o]);
        }
        const size_t nc = (size_t)off.back();
        std::vector<float> x(nc), y(nc), z(nc), u(nc), v(nc);
        if (nc && (!d->p3d_x || !d->p3d_y || !d->p3d_z || !d->p2d_u || !d->p2d_v)) return bail(CCAL_ERR_INVALID_ARG, "null corner arrays");
        for (size_t i = 0; i < obs.size(); ++i) {
            const int64_t s0 = d->obs_offsets[obs[i]], cnt = off[i + 1] - off[i];
            if (!cnt) continue;
            std::memcpy(&x[(size_t)off[i]], d->p3d_x + s0, (size_t)cnt * sizeof(float)); std::memcpy(&y[(size_t)off[i]], d->p3d_y + s0, (size_t)cnt * sizeof(float));
            std::memcpy(&z[(size_t)off[i]], d->p3d_z + s0, (size_t)cnt * sizeof(float)); std::memcpy(&u[(size_t)off[i]], d->p2d_u + s0, (size_t)cnt * sizeof(float));
            std::memcpy(&v[(size_t)off[i]], d->p2d_v + s0, (size_t)cnt * sizeof(float));
        }
        ccal_problem_desc sd = *d;
        sd.n_slots = mp->first[(size_t)r + 1] - mp->first[(size_t)r];
        sd.n_obs = (int32_t)obs.size();
        sd.obs_cam = cam.data(); sd.obs_slot = slot.data(); sd.obs_offsets = off.data();
        sd.p3d_x = x.data(); sd.p3d_y = y.data(); sd.p3d_z = z.data(); sd.p2d_u = u.data(); sd.p2d_v = v.data();
        const int rc = ccal_problem_create(m->ctx[(size_t)r], &sd, &mp->shard[(size_t)r]);
        if (rc != CCAL_OK) return bail(rc, std::string("shard ") + std::to_string(r) + ": " + ccal_last_error(m->ctx[(size_t)r]));
        ccal_problem* p = mp->shard[(size_t)r];
        if (m->transport == CCAL_TRANSPORT_RCCL) p->rccl_comm = m->comms[(size_t)r];
        else if (m->transport == CCAL_TRANSPORT_INPROC) p->peer = inproc_rank_handle(m->inproc, r);
    }
    *out = mp.release();
    return CCAL_OK;
    CCAL_API_CATCH((ccal_ctx*)nullptr)
}

int ccal_multi_problem_num_shards(const ccal_multi_problem* mp) { return mp ? (int)mp->shard.size() : -1; }
ccal_problem* ccal_multi_problem_shard(ccal_multi_problem* mp, int i) { return (mp && i >= 0 && i < (int)mp->shard.size()) ? mp->shard[(size_t)i] : nullptr; }
int ccal_multi_problem_slot_range(const ccal_multi_problem* mp, int i, int32_t* first_slot, int32_t* n_slots) {
    if (!mp || i < 0 || i >= (int)mp->shard.size()) return CCAL_ERR_INVALID_ARG;
    if (first_slot) *first_slot = mp->first[(size_t)i];
    if (n_slots) *n_slots = mp->first[(size_t)i + 1] - mp->first[(size_t)i];
    return CCAL_OK;
}

// constraints: the same on every shard (ccal_solve_sharded checks it)
#define CCAL_MULTI_FORALL(call)                                                                   \
    if (!mp) return CCAL_ERR_INVALID_ARG;                                                          \
    for (ccal_problem* p : mp->shard) { const int rc = (call); if (rc != CCAL_OK) return rc; }     \
    return CCAL_OK;
int ccal_multi_set_bounds(ccal_multi_problem* mp, int cam, int eff_idx, double lo, double hi) { CCAL_MULTI_FORALL(ccal_set_bounds(p, cam, eff_idx, lo, hi)) }
int ccal_multi_clear_bounds(ccal_multi_problem* mp, int cam, int eff_idx) { CCAL_MULTI_FORALL(ccal_clear_bounds(p, cam, eff_idx)) }
int ccal_multi_fix_param(ccal_multi_problem* mp, int cam, int eff_idx) { CCAL_MULTI_FORALL(ccal_fix_param(p, cam, eff_idx)) }
int ccal_multi_unfix_param(ccal_multi_problem* mp, int cam, int eff_idx) { CCAL_MULTI_FORALL(ccal_unfix_param(p, cam, eff_idx)) }
int ccal_multi_apply_reference_bounds(ccal_multi_problem* mp) { CCAL_MULTI_FORALL(ccal_apply_reference_bounds(p)) }
int ccal_multi_disable_distortions(ccal_multi_problem* mp, int n_disabled, double* intr_io) { CCAL_MULTI_FORALL(ccal_disable_distortions(p, n_disabled, intr_io)) }
#undef CCAL_MULTI_FORALL

int ccal_multi_upload_params(ccal_multi_problem* mp, const double* intr, const double* poses, const double* extr) {
    if (!mp) return CCAL_ERR_INVALID_ARG;
    for (size_t r = 0; r < mp->shard.size(); ++r) {
        const int rc = ccal_upload_params(mp->shard[r], intr, poses ? poses + 6 * (size_t)mp->first[r] : nullptr, extr);
        if (rc != CCAL_OK) return mfail(mp->m, rc, ccal_last_error(mp->m->ctx[r]));
    }
    return CCAL_OK;
}
// Mode E needs no collective: every shard's launches are enqueued on its context's stream and run side by side.
// r_dev[i] / J_dev[i]: device buffers on shard i's GPU, sized for ITS corners (ccal_num_corners / ccal_jacobian_len of the shard).
int ccal_multi_eval_dev(ccal_multi_problem* mp, int apply_loss, double* const* r_dev, double* const* J_dev) {
    if (!mp || !r_dev || !J_dev) return CCAL_ERR_INVALID_ARG;
    for (size_t r = 0; r < mp->shard.size(); ++r) {
        const int rc = ccal_eval_dev(mp->shard[r], apply_loss, r_dev[r], J_dev[r]);
        if (rc != CCAL_OK) return mfail(mp->m, rc, ccal_last_error(mp->m->ctx[r]));
    }
    return CCAL_OK;
}

int ccal_multi_init_poses(ccal_multi_problem* mp, const double* intr, int min_points, double* poses_obs, int32_t* n_used) {
    if (!mp || !intr || !poses_obs || !n_used) return CCAL_ERR_INVALID_ARG;
    CCAL_API_TRY
    const int n = (int)mp->shard.size();
    std::vector<int> rc((size_t)n, CCAL_OK);
    std::vector<std::vector<double>> po((size_t)n);
    std::vector<std::vector<int32_t>> nu((size_t)n);
    for (int r = 0; r < n; ++r) { po[(size_t)r].assign(std::max<size_t>(mp->obs_of[(size_t)r].size(), 1) * 6, 0.0); nu[(size_t)r].assign(std::max<size_t>(mp->obs_of[(size_t)r].size(), 1), 0); }
    for_each_shard(mp->m, n, [&](int r) { rc[(size_t)r] = ccal_init_poses(mp->shard[(size_t)r], intr, min_points, po[(size_t)r].data(), nu[(size_t)r].data()); });
    for (int r = 0; r < n; ++r) {
        if (rc[(size_t)r] != CCAL_OK) return mfail(mp->m, rc[(size_t)r], ccal_last_error(mp->m->ctx[(size_t)r]));
        const std::vector<int32_t>& obs = mp->obs_of[(size_t)r];
        for (size_t i = 0; i < obs.size(); ++i) {
            std::memcpy(poses_obs + 6 * (size_t)obs[i], &po[(size_t)r][6 * i], 6 * sizeof(double));
            n_used[obs[i]] = nu[(size_t)r][i];
        }
    }
    return CCAL_OK;
    CCAL_API_CATCH((ccal_ctx*)nullptr)
}

// validation() over the shards (src/util.rs:721-795): every shard evaluates its corners' errors on its GPU; the camera's errors of
// all shards meet on the first shard's GPU, where the single-GPU statistics code sorts and sums them - the same values, the same
// sorted order, the same reduction: the same bits as ccal_validation on one GPU.
int ccal_multi_validation(ccal_multi_problem* mp, int cam, const double* intr, const double* poses, const double* extr,
                          double* avg_99_percent, double* median) {
    if (!mp || cam < 0 || cam >= mp->n_cams || !intr || (!poses && mp->n_slots) || (!extr && mp->n_cams > 1) || !avg_99_percent || !median) return CCAL_ERR_INVALID_ARG;
    ccal_multi* m = mp->m;
    CCAL_API_TRY
    const int n = (int)mp->shard.size();
    std::vector<double*> d_part((size_t)n, nullptr);       // slices of the shards' own scratch blocks (nothing to free)
    std::vector<int64_t> cnt((size_t)n, 0);
    std::vector<int> rc((size_t)n, CCAL_OK);
    for_each_shard(m, n, [&](int r) {
        ccal_problem* p = mp->shard[(size_t)r];
        rc[(size_t)r] = reprojection_errors_dev(p, intr, poses ? poses + 6 * (size_t)mp->first[(size_t)r] : nullptr, extr);
        if (rc[(size_t)r] == CCAL_OK && camera_errors_device(p, cam, p->d_err, &d_part[(size_t)r], &cnt[(size_t)r], p->ctx->stream) != hipSuccess) {
            rc[(size_t)r] = CCAL_ERR_HIP; note_error(p->ctx, "ccal_multi_validation: gathering the camera's errors failed");
        }
    });
    int64_t total = 0;
    for (int r = 0; r < n; ++r) {
        if (rc[(size_t)r] != CCAL_OK) return mfail(m, rc[(size_t)r], std::string("shard ") + std::to_string(r) + ": " + ccal_last_error(m->ctx[(size_t)r]));
        total += cnt[(size_t)r];
    }
    if (total <= 0) return mfail(m, CCAL_ERR_INVALID_ARG, "camera has no observations");
    ccal_ctx* c0 = m->ctx[0];
    if (hipSetDevice(c0->device) != hipSuccess) return mfail(m, CCAL_ERR_HIP, "ccal_multi_validation: hipSetDevice failed");
    const size_t need = order_stats_block_bytes(total, c0->stream);
    if (need == 0) return mfail(m, CCAL_ERR_HIP, "ccal_multi_validation: sizing the selection's work area failed");
    if (mp->gather_bytes < need) {
        if (mp->d_gather) { (void)hipFree(mp->d_gather); mp->d_gather = nullptr; mp->gather_bytes = 0; }
        const size_t want = std::max(need, order_stats_block_bytes(std::max<int64_t>(mp->n_corners, 1), c0->stream));     // every camera of the problem fits
        if (hipMalloc((void**)&mp->d_gather, want) != hipSuccess) { (void)hipGetLastError(); return mfail(m, CCAL_ERR_NO_MEMORY, "ccal_multi_validation: out of device memory"); }
        mp->gather_bytes = want;
    }
    double* d_all = reinterpret_cast<double*>(mp->d_gather);
    int64_t at = 0;
    for (int r = 0; r < n; ++r) {                            // shard order = slot order: the single-GPU gather's order
        if (!cnt[(size_t)r]) continue;                       // (the shards' gathers have completed: camera_errors_device waits for its stream)
        if (hipMemcpyAsync(d_all + at, d_part[(size_t)r], (size_t)cnt[(size_t)r] * sizeof(double), hipMemcpyDeviceToDevice, c0->stream) != hipSuccess)
            return mfail(m, CCAL_ERR_HIP, "ccal_multi_validation: gathering the shards' errors failed");
        at += cnt[(size_t)r];
    }
    if (order_stats_block(mp->d_gather, mp->gather_bytes, total, avg_99_percent, median, c0->stream) != hipSuccess)
        return mfail(m, CCAL_ERR_HIP, "ccal_multi_validation: the statistics kernels failed");
    return CCAL_OK;
    CCAL_API_CATCH((ccal_ctx*)nullptr)
}

// Per-corner reprojection errors in the CALLER's corner order (the order of the description handed to ccal_multi_problem_create).
int ccal_multi_reprojection_errors(ccal_multi_problem* mp, const double* intr, const double* poses, const double* extr, double* err_out,
                                   const int64_t* obs_offsets /* the description's [n_obs + 1] */) {
    if (!mp || !intr || (!poses && mp->n_slots) || (!extr && mp->n_cams > 1) || !err_out || !obs_offsets) return CCAL_ERR_INVALID_ARG;
    CCAL_API_TRY
    const int n = (int)mp->shard.size();
    std::vector<int> rc((size_t)n, CCAL_OK);
    std::vector<std::vector<double>> part((size_t)n);
    for (int r = 0; r < n; ++r) part[(size_t)r].assign((size_t)std::max<int64_t>(ccal_num_corners(mp->shard[(size_t)r]), 1), 0.0);
    for_each_shard(mp->m, n, [&](int r) {
        rc[(size_t)r] = ccal_reprojection_errors(mp->shard[(size_t)r], intr, poses ? poses + 6 * (size_t)mp->first[(size_t)r] : nullptr, extr, part[(size_t)r].data());
    });
    for (int r = 0; r < n; ++r) {
        if (rc[(size_t)r] != CCAL_OK) return mfail(mp->m, rc[(size_t)r], ccal_last_error(mp->m->ctx[(size_t)r]));
        int64_t at = 0;
        for (int32_t o : mp->obs_of[(size_t)r]) {
            const int64_t c = obs_offsets[o + 1] - obs_offsets[o];
            std::memcpy(err_out + obs_offsets[o], &part[(size_t)r][(size_t)at], (size_t)c * sizeof(double));
            at += c;
        }
    }
    return CCAL_OK;
    CCAL_API_CATCH((ccal_ctx*)nullptr)
}

int ccal_multi_solve(ccal_multi_problem* mp, const ccal_solver_opts* o, double* intr_io, double* poses_io, double* extr_io, ccal_report* rep) {
    if (!mp || !o || !intr_io || (!poses_io && mp->n_slots) || (!extr_io && mp->n_cams > 1)) return CCAL_ERR_INVALID_ARG;
    ccal_multi* m = mp->m;
    CCAL_API_TRY
    const int n = (int)mp->shard.size();
    if (n == 1) {
        const int rc = ccal_solve(mp->shard[0], o, intr_io, poses_io, extr_io, rep);
        if (rc != CCAL_OK) m->err = ccal_last_error(m->ctx[0]);
        return rc;
    }
    if (m->broken) return mfail(m, CCAL_ERR_HIP, "an earlier sharded solve failed inside a collective: the RCCL communicators of this ccal_multi were aborted - destroy it and create a new one");
    std::vector<double*> pp((size_t)n);
    for (int r = 0; r < n; ++r) pp[(size_t)r] = poses_io ? poses_io + 6 * (size_t)mp->first[(size_t)r] : nullptr;
    if (m->inproc) inproc_set_timeout(m->inproc, o->timeout_s > 0 ? (double)o->timeout_s : 600.0);
    const int rc = solve_sharded_run(mp->shard.data(), n, o, intr_io, pp.data(), extr_io, rep, m->inproc);
    const bool verdict = rc == CCAL_OK || rc == CCAL_ERR_NONFINITE || rc == CCAL_ERR_NOT_PD || rc == CCAL_ERR_NO_CONVERGENCE;
    if (rc != CCAL_OK) m->err = ccal_last_error(m->ctx[0]);
    if (!verdict) {
        // a rank left the shared sequence of collectives.  In-process transport: drain the devices, start over.  RCCL: peers
        // may sit inside ncclAllReduce with no partner - abort the communicators, this device set is finished
        if (m->inproc) {
            if (inproc_recover(m->inproc) != CCAL_OK) m->broken = true;
            else for (ccal_problem* p : mp->shard) if (p->nws) { p->nws->tail_pending = false; if (p->nws->fws) p->nws->fws->tail_pending = false; }
        }
        else if (m->transport == CCAL_TRANSPORT_RCCL) {
            for (size_t r = 0; r < m->comms.size(); ++r) { rccl_comm_abort(m->comms[r]); m->comms[r] = nullptr; mp->shard[r]->rccl_comm = nullptr; }
            m->broken = true;
        }
    }
    return rc;
    CCAL_API_CATCH((ccal_ctx*)nullptr)
}

}  // extern "C"
