// The decision / camera-solve step of the single-camera loop as device functions: k_head runs them as a kernel of its own
// (sharded solves, large problems), k_gram1v<.., ITER> in front of its evaluation (single-launch groups, session sizes).
#pragma once
#include "ccal_gram_common.hpp"

namespace ccal {

struct HeadShared {               // LDS of the decision / solve step
    DevState S0;
    double red[2 * 100 + 2];
    double S[10 * 11];
    double x[10];                 // the camera step dc (zero when the camera system was not positive definite)
    double cur_intr[CCAL_PMAX];   // current intrinsics (full layout)
    double cand[CCAL_PMAX];       // candidate intrinsics (full layout)
    int fx[10];                   // fixed flags of the camera columns
    int bad, solve;
    int early_pub;                // the group's status word went out before the camera solve (a group that does not finish)
    int entry_done;               // the solve had finished before this group
};

struct HeadIO {
    const DevState* st_in; DevState* st_out; HostStatus* hs;
    const double* red_g;          // the (all-)reduced sums in global memory; NULL: the caller has put them into HeadShared::red
    const ColInfo* cols;
    double* intr[2]; double* dc;
    int32_t K, seq, publish_all;
    double min_diag, max_diag;
};

// S (LDS, row stride 11) x = rhs (LDS) by Cholesky, entirely in registers of every lane; lane 0 writes x back.
template <int K>
__device__ __forceinline__ bool chol_solve_reg(const double* S, double* x) {
    double M[K * K], v[K];
#pragma unroll
    for (int i = 0; i < K; ++i) {
        v[i] = x[i];
#pragma unroll
        for (int j = 0; j <= i; ++j) M[i * K + j] = S[i * 11 + j];
    }
    bool ok = true;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        double d = M[j * K + j];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= M[j * K + k] * M[j * K + k];
        ok = ok && (d > 0.0) && (d < 1.7e308);
        double sq, rs;
        fast_sqrt_rsqrt(ok ? d : 1.0, sq, rs);
        M[j * K + j] = rs;                               // inverted diagonal
#pragma unroll
        for (int i = j + 1; i < K; ++i) {
            double t = M[i * K + j];
#pragma unroll
            for (int k = 0; k < j; ++k) t -= M[i * K + k] * M[j * K + k];
            M[i * K + j] = t * rs;
        }
    }
#pragma unroll
    for (int i = 0; i < K; ++i) {
        double t = v[i];
#pragma unroll
        for (int k = 0; k < i; ++k) t -= M[i * K + k] * v[k];
        v[i] = t * M[i * K + i];
    }
#pragma unroll
    for (int i = K - 1; i >= 0; --i) {
        double t = v[i];
#pragma unroll
        for (int k = i + 1; k < K; ++k) t -= M[k * K + i] * v[k];
        v[i] = t * M[i * K + i];
    }
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < K; ++i) x[i] = v[i];
    }
    return ok;
}


// What head_wave reads from global memory, one element per lane - requested ahead of whatever the caller does first (the
// single-launch form: the sum over the previous launch's rows), so that the decision does not start with a memory round trip.
struct HeadPre { ColInfo ci; double intr_a, intr_b, stw; double redv[4]; };      // redv: this lane's entries of the reduced sums (k_head: they sit in global memory)
__device__ __forceinline__ HeadPre head_prefetch(const HeadIO& a, const int lane) {
    HeadPre p;
    p.ci = ColInfo{}; p.intr_a = 0.0; p.intr_b = 0.0; p.stw = 0.0;
    if (lane < a.K) p.ci = a.cols[lane];
    if (lane < CCAL_PMAX) { p.intr_a = a.intr[0][lane]; p.intr_b = a.intr[1][lane]; }
    static_assert(sizeof(DevState) / sizeof(double) <= 64, "one element of the state per lane");
    if (lane < (int)(sizeof(DevState) / sizeof(double))) p.stw = reinterpret_cast<const double*>(a.st_in)[lane];
    // the (all-)reduced sums travel with the same round trip (they were requested after the state had arrived: a second one)
    static_assert(fused_red_size(kFusedMaxK) <= 4 * 64, "four entries of the sums per lane");
#pragma unroll
    for (int q = 0; q < 4; ++q) p.redv[q] = (a.red_g && lane + 64 * q < fused_red_size(a.K)) ? a.red_g[lane + 64 * q] : 0.0;
    return p;
}

// ONE wavefront.  Decision (optimizer_decide, ccal_fused.hpp) on the all-reduced sums, then - when the sums at hand are the
// system to solve - the K x K camera solve and the candidate intrinsics.
// red = [A_dir (K1*K1) | Y^T Y (K1*K1) | mc_pose | failed pose blocks]
// The state is staged in LDS (hs.S0) and stays there for the caller: decided state, camera step hs.x, candidate hs.cand.
// `writer`: this wavefront also writes them to global memory and tells the host (k_head: always; single-launch groups:
// workgroup 0 - every workgroup computes the same values from the same sums).
__device__ __forceinline__ void head_wave(const HeadIO& a, HeadShared& hs, const int lane, const bool writer, const HeadPre& pre) {
    DevState& S0 = hs.S0;
    double* red = hs.red; double* S = hs.S; double* x = hs.x; int& bad = hs.bad;
    const int K = a.K, K1 = K + 1;
    // everything the solve needs from global memory was requested up front (head_prefetch), next to the state: one memory
    // latency instead of a chain of three (state -> column info -> intrinsics)
    const ColInfo ci = pre.ci;
    const double intr_a = pre.intr_a, intr_b = pre.intr_b;
    {   // stage state + reduced sums
        double* dst = reinterpret_cast<double*>(&S0);
        if (lane < (int)(sizeof(DevState) / sizeof(double))) dst[lane] = pre.stw;
        if (a.red_g) {
#pragma unroll
            for (int q = 0; q < 4; ++q) if (lane + 64 * q < fused_red_size(K)) red[lane + 64 * q] = pre.redv[q];
        }
        if (lane < K) hs.fx[lane] = ci.fixed;
        if (lane == 0) { hs.early_pub = 0; hs.entry_done = 0; }
    }
    wsync();
    DevState* st = &S0;
    if (st->done) {               // the host still waits for this group's number (head_finish publishes it)
        if (lane == 0) hs.entry_done = 1;
        if (writer && a.st_out != a.st_in) {
            const double* src = reinterpret_cast<const double*>(&S0);
            double* dst = reinterpret_cast<double*>(a.st_out);
            for (int e = lane; e < (int)(sizeof(DevState) / sizeof(double)); e += 64) dst[e] = src[e];
        }
        wsync();
        return;
    }
    const double* Ad = red;
    const double* Yt = red + K1 * K1;
    const bool lm = st->method == CCAL_METHOD_LM;
    {   // the decision on a register copy of the state (every lane the same wave-uniform work; through the LDS copy each of the
        // rule's ~40 dependent field accesses was an LDS round trip), written back by lane 0
        DevState loc = S0;
        const bool sv = optimizer_decide(&loc, Ad[K * K1 + K], red[2 * K1 * K1], red[2 * K1 * K1 + 1] > 0.0, a.seq);
        wsync();
        if (lane == 0) { S0 = loc; hs.solve = sv ? 1 : 0; }
    }
    wsync();
    // A group that does NOT finish the solve tells the host so as soon as that is certain - after the decision for a group that
    // does not solve (a re-elimination follows), after the camera factorisation otherwise (Gauss-Newton ends the solve there when
    // the system is not positive definite: what the host reads for a group must not depend on WHEN it looks, or sharded ranks
    // would enqueue different numbers of collectives).  The word only says "group seq has decided, go on": the host answers by
    // enqueueing a later group behind the ones already in the stream, nothing it does depends on what this kernel still writes,
    // and the store's trip across the bus overlaps the rest of the kernel instead of sitting in front of its end.  A finishing
    // group publishes last (its report and result must be complete first); so does a verbose solve's full report.
    if (!st->done && !a.publish_all && !hs.solve) { if (lane == 0) { hs.early_pub = 1; if (writer) a.hs->word = status_word(a.seq, 0, 0); } }
    if (hs.solve) {
        const double lambda = st->lambda;
        const int cur = st->cur;
        if (lane == 0) bad = 0;
        for (int e = lane; e < K * K; e += 64) {
            const int i = e / K, j = e - i * K;
            double v = Ad[i * K1 + j] - Yt[i * K1 + j];
            const bool fi = hs.fx[i] != 0, fj = hs.fx[j] != 0;
            if (fi || fj) v = (i == j) ? 1.0 : 0.0;
            else if (i == j && lambda > 0.0) v += lambda * clampd1(Ad[i * K1 + i], a.min_diag, a.max_diag);
            S[i * 11 + j] = v;
        }
        if (lane < K) x[lane] = ci.fixed ? 0.0 : -(Ad[lane * K1 + K] - Yt[lane * K1 + K]);
        wsync();
        // K <= 9: Cholesky + both triangular solves in registers (every lane the same wave-uniform work, no
        // LDS round trips or barriers inside the factorisation)
        {
            bool okc = true;
            switch (K) {
                case 4: okc = chol_solve_reg<4>(S, x); break;
                case 5: okc = chol_solve_reg<5>(S, x); break;
                case 6: okc = chol_solve_reg<6>(S, x); break;
                case 7: okc = chol_solve_reg<7>(S, x); break;
                case 8: okc = chol_solve_reg<8>(S, x); break;
                default: okc = chol_solve_reg<9>(S, x); break;
            }
            if (lane == 0 && !okc) bad = 1;
        }
        wsync();
        if (bad) {
            if (lane == 0) {
                if (!lm) { st->done = CCAL_ERR_NOT_PD + 1; if (!st->done_seq) st->done_seq = a.seq; }
                else st->cam_failed = 1;             // LM: the next decision rejects and shrinks the radius
                st->mc_cam = 0.0; st->lambda_solve = lambda;
            }
            if (lane < K) x[lane] = 0.0;
        }
        wsync();
        if (!st->done && !a.publish_all) { if (lane == 0) { hs.early_pub = 1; if (writer) a.hs->word = status_word(a.seq, 0, 0); } }      // (st->done: set just above, GN only)
        if (!bad || lm) {
            // candidate intrinsics = clamp(x + dc) into the other set; model decrease of the camera block
            const double keep = cur ? intr_b : intr_a;          // current intrinsics, full layout, element `lane`
            if (lane < CCAL_PMAX) { hs.cand[lane] = keep; hs.cur_intr[lane] = keep; }
            wsync();
            double mc = 0.0;
            if (lane < K && !bad) {
                const double d = x[lane];
                const double Dii = lambda > 0.0 ? lambda * clampd1(Ad[lane * K1 + lane], a.min_diag, a.max_diag) : 0.0;
                if (!ci.fixed) {
                    mc = d * (Dii * d - Ad[lane * K1 + K]);
                    double v = hs.cur_intr[ci.dst] + d;
                    if (ci.has_bound) v = fmin(fmax(v, ci.lo), ci.hi);
                    hs.cand[ci.dst] = v;
                    if (ci.dst2 >= 0) hs.cand[ci.dst2] = v;
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) mc += __shfl_down(mc, off, 64);
            if (lane == 0 && !bad) { st->mc_cam = mc; st->lambda_solve = lambda; }
            wsync();
            if (writer && lane < CCAL_PMAX) (cur ? a.intr[0] : a.intr[1])[lane] = hs.cand[lane];      // (no run-time index into the argument block: it would live in scratch)
        }
        if (writer && lane < K) a.dc[lane] = x[lane];
    }
    wsync();
    if (writer) {   // the decided state
        const double* src = reinterpret_cast<const double*>(&S0);
        double* dst = reinterpret_cast<double*>(a.st_out);
        for (int e = lane; e < (int)(sizeof(DevState) / sizeof(double)); e += 64) dst[e] = src[e];
    }
}

// PUBLISH-LAST INVARIANT (FusedJob::end relies on it): the host copies the result out of pinned memory as soon as it reads a
// status word that says `done`, without synchronising the stream.  That is correct only because (1) every store of the result below
// precedes the system-scope fence, (2) the fence and a workgroup barrier precede the store of the status word, and (3) NOTHING
// is written to result_host - or to anything else the host reads on `done` - after the word.  A future store behind the publish
// breaks the host's result silently; tests/test_gpu_iter.py::test_zero_copy_result_is_the_device_state holds the host's copy
// against the device's after a synchronise.
// The whole workgroup of the writer, behind head_wave and a workgroup barrier: a group that finishes the solve copies the
// accepted point straight to the caller's side of the bus (session-sized solves with host pointers, result_host != NULL) -
// every wavefront of the workgroup, 16-byte stores (tools/ubench/host_publish.hip: 30 KB in 2.9 us with four wavefronts,
// 8.2 us with one and 8-byte stores) - and then, last, the status word.  The accepted poses were written by an earlier launch;
// the accepted intrinsics: no group writes set `cur` (candidates go to cur ^ 1), global memory holds them.
// SPREAD (single-launch groups of problems beyond the session sizes, spread_poses != NULL): every workgroup of the launch has taken the
// same decision on the same sums, so every workgroup copies ITS slice of the accepted poses to the caller's side of the bus (up to 256 KB:
// one workgroup would need ~12 us, a DMA behind the launch costs the host a stream round trip; beyond that the DMA's larger packets win) and counts itself in
// (*done_cnt, device scope, behind its own system-scope fence); the workgroup that counts last resets the counter and publishes the word:
// (1) - (3) hold per workgroup, and the word is behind every workgroup's fence.
__device__ __forceinline__ void head_finish(const HeadIO& a, HeadShared& hs, double* result_host, const double* poses0, const double* poses1,
                                            const int64_t np6, const bool writer, double* spread_poses = nullptr, int32_t* done_cnt = nullptr) {
    const DevState* st = &hs.S0;
    const bool fin = st->done && result_host && !hs.entry_done;
    const bool spread = fin && spread_poses != nullptr;
    if (!writer && !spread) return;
    if (fin) {
        typedef double dv2 __attribute__((ext_vector_type(2)));
        const int cur = st->cur;
        const dv2* pi = reinterpret_cast<const dv2*>(cur ? a.intr[1] : a.intr[0]);
        const dv2* pp = reinterpret_cast<const dv2*>(cur ? poses1 : poses0);
        dv2* out = reinterpret_cast<dv2*>(result_host);
        static_assert(CCAL_PMAX % 2 == 0, "16-byte stores");
        if (writer) for (int e = threadIdx.x; e < CCAL_PMAX / 2; e += blockDim.x) out[e] = pi[e];
        if (spread) {
            dv2* outp = reinterpret_cast<dv2*>(spread_poses);
            const int64_t n2 = np6 / 2, per = (n2 + gridDim.x - 1) / gridDim.x;          // (np6 = 6 x slots: even)
            const int64_t e1 = per * (blockIdx.x + 1) < n2 ? per * (blockIdx.x + 1) : n2;
            for (int64_t e = per * blockIdx.x + threadIdx.x; e < e1; e += blockDim.x) outp[e] = pp[e];
        } else {
            for (int64_t e = threadIdx.x; e < np6 / 2; e += blockDim.x) out[CCAL_PMAX / 2 + e] = pp[e];
        }
        __threadfence_system();
    }
    __syncthreads();
    if (spread) {
        if (threadIdx.x == 0) {
            const int old = __hip_atomic_fetch_add(done_cnt, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (old == (int)gridDim.x - 1) {
                __hip_atomic_store(done_cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __threadfence_system();
                if (!hs.early_pub) publish_host_status(a.hs, st, a.seq, a.publish_all != 0);
            }
        }
        return;
    }
    if (threadIdx.x == 0 && !hs.early_pub) publish_host_status(a.hs, st, a.seq, a.publish_all != 0);
}

}  // namespace ccal
