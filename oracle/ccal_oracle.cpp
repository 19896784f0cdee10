// ORACLE -- TEST INFRASTRUCTURE ONLY (see ccal_oracle.hpp for provenance and pinning).
// extern "C" entry points used by tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg through ctypes.  Nothing in the product links this file.
#include "ccal_oracle.hpp"
#include "../include/ccal.h"   // POD structs only (problem description, solver options, report)

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <thread>

#include <sched.h>

namespace oracle {

void factor_jacobian_dyn(int D, int model, bool xy, bool other, int p_eff,
                         const double* params, const double* pose0, const double* pose1,
                         const float p3d[3], const float p2d[2], double r[2], double* J) {
    switch (D) {
#define ORACLE_CASE(n) case n: factor_jacobian<n>(model, xy, other, p_eff, params, pose0, pose1, p3d, p2d, r, J); break;
        ORACLE_CASE(10) ORACLE_CASE(11) ORACLE_CASE(12) ORACLE_CASE(13) ORACLE_CASE(14) ORACLE_CASE(15)
        ORACLE_CASE(16) ORACLE_CASE(17) ORACLE_CASE(18) ORACLE_CASE(19) ORACLE_CASE(20) ORACLE_CASE(21)
#undef ORACLE_CASE
        default: r[0] = r[1] = NAN; break;
    }
}

// ---- problem bookkeeping ------------------------------------------------------------------
struct Layout {
    int n_cams = 0; bool xy = false;
    int model[CCAL_MAX_CAMS]; int P[CCAL_MAX_CAMS]; int Peff[CCAL_MAX_CAMS];
    int D[CCAL_MAX_CAMS];              // block Jacobian width
    int col_theta[CCAL_MAX_CAMS];      // column of theta_c in the reduced system
    int col_extr[CCAL_MAX_CAMS];       // column of rvec_c_0 (c>0)
    int K = 0;
};
static bool make_layout(const ccal_problem_desc* d, Layout& L) {
    if (!d || d->n_cams < 1 || d->n_cams > CCAL_MAX_CAMS) return false;
    L.n_cams = d->n_cams; L.xy = d->xy_same_focal != 0; L.K = 0;
    for (int c = 0; c < d->n_cams; ++c) {
        L.model[c] = d->model[c]; L.P[c] = model_nparams(d->model[c]);
        if (L.P[c] < 0) return false;
        L.Peff[c] = L.P[c] - (L.xy ? 1 : 0);
        L.D[c] = L.Peff[c] + (c == 0 ? 6 : 12);
        L.col_theta[c] = L.K; L.K += L.Peff[c];
        L.col_extr[c] = -1;
        if (c > 0) { L.col_extr[c] = L.K; L.K += 6; }
    }
    return L.K <= CCAL_KMAX;
}
static void full_to_eff(const Layout& L, int c, const double* full, double* eff) {
    if (L.xy) { eff[0] = full[0]; for (int i = 2; i < L.P[c]; ++i) eff[i - 1] = full[i]; }
    else for (int i = 0; i < L.P[c]; ++i) eff[i] = full[i];
}
static void eff_to_full(const Layout& L, int c, const double* eff, double* full) {
    if (L.xy) { full[0] = eff[0]; full[1] = eff[0]; for (int i = 2; i < L.P[c]; ++i) full[i] = eff[i - 1]; }
    else for (int i = 0; i < L.P[c]; ++i) full[i] = eff[i];
}

// One block (corner k of observation frame o): raw r, J (2 x D, row-major).
static inline void eval_block(const ccal_problem_desc* d, const Layout& L, int cam, int64_t k,
                              const double* eff, const double* pose0, const double* pose1,
                              double r[2], double* J, bool heap_duals = false) {
    const float p3[3] = { d->p3d_x[k], d->p3d_y[k], d->p3d_z[k] };
    const float p2[2] = { d->p2d_u[k], d->p2d_v[k] };
    if (heap_duals) factor_jacobian_heap(L.D[cam], L.model[cam], L.xy, cam > 0, L.Peff[cam], eff, pose0, pose1, p3, p2, r, J);
    else factor_jacobian_dyn(L.D[cam], L.model[cam], L.xy, cam > 0, L.Peff[cam], eff, pose0, pose1, p3, p2, r, J);
}

// Dense symmetric Cholesky (lower), in place; returns false if not PD.
static bool cholesky(double* A, int n) {
    for (int j = 0; j < n; ++j) {
        double s = A[j * n + j];
        for (int k = 0; k < j; ++k) s -= A[j * n + k] * A[j * n + k];
        if (!(s > 0.0) || !std::isfinite(s)) return false;
        const double l = std::sqrt(s);
        A[j * n + j] = l;
        for (int i = j + 1; i < n; ++i) {
            double t = A[i * n + j];
            for (int k = 0; k < j; ++k) t -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = t / l;
        }
    }
    return true;
}
static void chol_solve(const double* Lm, int n, double* x) {   // x <- (L L^T)^-1 x
    for (int i = 0; i < n; ++i) { double t = x[i]; for (int k = 0; k < i; ++k) t -= Lm[i * n + k] * x[k]; x[i] = t / Lm[i * n + i]; }
    for (int i = n - 1; i >= 0; --i) { double t = x[i]; for (int k = i + 1; k < n; ++k) t -= Lm[k * n + i] * x[k]; x[i] = t / Lm[i * n + i]; }
}

struct SlotBlocks {             // per frame slot, kept for back-substitution
    double C[36];               // H_pp (undamped)
    std::vector<double> B;      // 6 x K   H_pc
    double g[6];                // g_p
};

struct Normal {
    int K = 0;
    std::vector<double> Hcc;    // K x K (undamped, before Schur)
    std::vector<double> gc;     // K
    std::vector<double> S;      // K x K reduced (damped with lambda)
    std::vector<double> b;      // K reduced
    double cost = 0.0;
    std::vector<SlotBlocks> slots;
};

static inline double clampd(double v, double lo, double hi) { return std::min(std::max(v, lo), hi); }

// Threads of the solver's two passes over the blocks (normal equations, cost): 1 = the serial restatement every parity test
// uses; bench.py's all-cores Gauss-Newton baseline sets more (oracle_set_solve_threads).  A thread takes a contiguous range
// of frame SLOTS (all cameras' observations of a slot: its pose block is complete), accumulates into its own camera block,
// the blocks are added in thread order.
static int g_solve_threads = 1;
template <class F>
static void for_slot_ranges(int n_slots, F&& fn) {
    const int T = std::max(1, std::min(g_solve_threads, std::max(n_slots, 1)));
    if (T == 1) { fn(0, 0, n_slots); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back([&fn, t, T, n_slots] { fn(t, (int)((int64_t)n_slots * t / T), (int)((int64_t)n_slots * (t + 1) / T)); });
    for (auto& x : th) x.join();
}

// Accumulate the full arrow-shaped normal equations from dual-number Jacobians and reduce.
static void build_normal(const ccal_problem_desc* d, const Layout& L, const double* intr, const double* poses,
                         const double* extr, double lambda, double min_diag, double max_diag, Normal& N) {
    const int K = L.K;
    N.K = K; N.Hcc.assign((size_t)K * K, 0.0); N.gc.assign(K, 0.0); N.cost = 0.0;
    N.slots.resize(d->n_slots);
    for (auto& s : N.slots) { std::memset(s.C, 0, sizeof s.C); s.B.assign((size_t)6 * K, 0.0); std::memset(s.g, 0, sizeof s.g); }
    double eff[CCAL_MAX_CAMS][9];
    for (int c = 0; c < L.n_cams; ++c) full_to_eff(L, c, intr + (size_t)c * CCAL_PMAX, eff[c]);
    const int T = std::max(1, std::min(g_solve_threads, std::max(d->n_slots, 1)));
    std::vector<std::vector<double>> tH(T > 1 ? T : 0), tg(T > 1 ? T : 0);
    std::vector<double> tcost(T, 0.0);
    for_slot_ranges(d->n_slots, [&](int t, int s_lo, int s_hi) {
    std::vector<double>& Hcc = T > 1 ? tH[t] : N.Hcc;
    std::vector<double>& gcv = T > 1 ? tg[t] : N.gc;
    if (T > 1) { Hcc.assign((size_t)K * K, 0.0); gcv.assign(K, 0.0); }
    double cost_local = 0.0;
    double& cost = T > 1 ? cost_local : N.cost;       // (a thread's running sum stays off the shared cache line)
    double J[2 * 21], r[2];
    int colmap[21];
    for (int o = 0; o < d->n_obs; ++o) {
        const int cam = d->obs_cam[o], slot = d->obs_slot[o];
        if (slot < s_lo || slot >= s_hi) continue;
        const int D = L.D[cam], Pe = L.Peff[cam];
        for (int i = 0; i < Pe; ++i) colmap[i] = L.col_theta[cam] + i;
        for (int i = 0; i < 6; ++i) colmap[Pe + i] = K + i;                       // pose columns (local)
        if (cam > 0) for (int i = 0; i < 6; ++i) colmap[Pe + 6 + i] = L.col_extr[cam] + i;
        SlotBlocks& sb = N.slots[slot];
        for (int64_t k = d->obs_offsets[o]; k < d->obs_offsets[o + 1]; ++k) {
            eval_block(d, L, cam, k, eff[cam], poses + (size_t)slot * 6, extr + (size_t)cam * 6, r, J);
            const double s = r[0] * r[0] + r[1] * r[1];
            const double w = d->huber_delta > 0.0 ? huber_weight(s, d->huber_delta) : 1.0;
            cost += w * s;
            for (int row = 0; row < 2; ++row) {
                const double* Jr = J + row * D;
                const double wr = w * r[row];
                for (int a = 0; a < D; ++a) {
                    const int ca = colmap[a];
                    const double wja = w * Jr[a];
                    if (ca < K) gcv[ca] += Jr[a] * wr; else sb.g[ca - K] += Jr[a] * wr;
                    for (int bq = 0; bq < D; ++bq) {
                        const int cb = colmap[bq];
                        const double v = wja * Jr[bq];
                        if (ca < K && cb < K) Hcc[(size_t)ca * K + cb] += v;
                        else if (ca >= K && cb >= K) sb.C[(ca - K) * 6 + (cb - K)] += v;
                        else if (ca >= K && cb < K) sb.B[(size_t)(ca - K) * K + cb] += v;
                    }
                }
            }
        }
    }
    if (T > 1) tcost[t] = cost_local;
    });
    if (T > 1) for (int t = 0; t < T; ++t) {
        for (size_t i = 0; i < N.Hcc.size(); ++i) N.Hcc[i] += tH[t][i];
        for (int i = 0; i < K; ++i) N.gc[i] += tg[t][i];
        N.cost += tcost[t];
    }
    // Schur complement of every pose block (Marquardt damping lambda * clamp(diag)).
    // (the camera block's own damping lambda * clamp(diag Hcc) is added at solve time, after any all-reduce)
    N.S = N.Hcc; N.b = N.gc;
    std::vector<std::vector<double>> tS(T > 1 ? T : 0), tb(T > 1 ? T : 0);
    std::vector<char> tbad(T, 0);
    for_slot_ranges(d->n_slots, [&](int t, int s_lo, int s_hi) {
    std::vector<double>& Sv = T > 1 ? tS[t] : N.S;
    std::vector<double>& bv = T > 1 ? tb[t] : N.b;
    if (T > 1) { Sv.assign((size_t)K * K, 0.0); bv.assign(K, 0.0); }
    std::vector<double> Y((size_t)6 * (K + 1));
    for (int si = s_lo; si < s_hi; ++si) {
        SlotBlocks& sb = N.slots[si];
        double Cl[36]; std::memcpy(Cl, sb.C, sizeof Cl);
        bool any = false; for (int i = 0; i < 6; ++i) any |= Cl[i * 6 + i] != 0.0;
        if (!any) continue;                                        // slot without observations
        if (lambda > 0.0) for (int i = 0; i < 6; ++i) Cl[i * 6 + i] += lambda * clampd(sb.C[i * 6 + i], min_diag, max_diag);
        if (!cholesky(Cl, 6)) { tbad[t] = 1; return; }
        // Y = L^-1 [B | g]
        for (int j = 0; j <= K; ++j) {
            for (int i = 0; i < 6; ++i) {
                double t = (j < K) ? sb.B[(size_t)i * K + j] : sb.g[i];
                for (int k = 0; k < i; ++k) t -= Cl[i * 6 + k] * Y[(size_t)k * (K + 1) + j];
                Y[(size_t)i * (K + 1) + j] = t / Cl[i * 6 + i];
            }
        }
        for (int a = 0; a < K; ++a) {
            for (int bq = 0; bq < K; ++bq) {
                double t = 0.0; for (int k = 0; k < 6; ++k) t += Y[(size_t)k * (K + 1) + a] * Y[(size_t)k * (K + 1) + bq];
                Sv[(size_t)a * K + bq] -= t;
            }
            double t = 0.0; for (int k = 0; k < 6; ++k) t += Y[(size_t)k * (K + 1) + a] * Y[(size_t)k * (K + 1) + K];
            bv[a] -= t;
        }
    }
    });
    for (int t = 0; t < T; ++t) if (tbad[t]) { N.cost = NAN; return; }
    if (T > 1) for (int t = 0; t < T; ++t) {
        for (size_t i = 0; i < N.S.size(); ++i) N.S[i] += tS[t][i];
        for (int i = 0; i < K; ++i) N.b[i] += tb[t][i];
    }
}

static double total_cost(const ccal_problem_desc* d, const Layout& L, const double* intr, const double* poses, const double* extr) {
    using T = double;
    double eff[CCAL_MAX_CAMS][9];
    for (int c = 0; c < L.n_cams; ++c) full_to_eff(L, c, intr + (size_t)c * CCAL_PMAX, eff[c]);
    const int NT = std::max(1, std::min(g_solve_threads, std::max(d->n_slots, 1)));
    std::vector<double> tcost(NT, 0.0);
    for_slot_ranges(d->n_slots, [&](int t, int s_lo, int s_hi) {
    double cost = 0.0;
    for (int o = 0; o < d->n_obs; ++o) {
        const int cam = d->obs_cam[o], slot = d->obs_slot[o];
        if (slot < s_lo || slot >= s_hi) continue;
        const double* ps = poses + (size_t)slot * 6; const double* ex = extr + (size_t)cam * 6;
        for (int64_t k = d->obs_offsets[o]; k < d->obs_offsets[o + 1]; ++k) {
            const float p3[3] = { d->p3d_x[k], d->p3d_y[k], d->p3d_z[k] };
            const float p2[2] = { d->p2d_u[k], d->p2d_v[k] };
            T r[2];
            if (cam > 0) other_cam_reprojection_factor<T>(L.model[cam], L.xy, eff[cam], ps, ps + 3, ex, ex + 3, p3, p2, r);
            else reprojection_factor<T>(L.model[cam], L.xy, eff[cam], ps, ps + 3, p3, p2, r);
            const double s = r[0] * r[0] + r[1] * r[1];
            cost += (d->huber_delta > 0.0 ? huber_weight(s, d->huber_delta) : 1.0) * s;
        }
    }
    tcost[t] = cost;
    });
    double cost = 0.0;
    for (int t = 0; t < NT; ++t) cost += tcost[t];
    return cost;
}

struct Constraints { const double* lo; const double* hi; const uint8_t* has_bound; const uint8_t* fixed; };

// One linear solve + update.  Returns CCAL_OK or an error; fills xc_* with the candidate and
// model_change with dx^T (lambda D dx - g) (the predicted decrease of sum w s).
static int step(const ccal_problem_desc* d, const Layout& L, const Constraints& cs, const Normal& N, double lambda,
                double min_diag, double max_diag, const double* intr, const double* poses, const double* extr,
                double* intr_c, double* poses_c, double* extr_c, double* mc_cam, double* mc_pose) {
    const int K = N.K;
    std::vector<double> S = N.S, dc(K);
    if (lambda > 0.0) for (int i = 0; i < K; ++i) S[(size_t)i * K + i] += lambda * clampd(N.Hcc[(size_t)i * K + i], min_diag, max_diag);
    std::vector<uint8_t> fx(K, 0);
    for (int c = 0; c < L.n_cams; ++c) for (int i = 0; i < L.Peff[c]; ++i)
        if (cs.fixed && cs.fixed[c * CCAL_PMAX + i]) fx[L.col_theta[c] + i] = 1;
    for (int i = 0; i < K; ++i) dc[i] = fx[i] ? 0.0 : -N.b[i];
    for (int i = 0; i < K; ++i) if (fx[i]) { for (int j = 0; j < K; ++j) { S[(size_t)i * K + j] = 0.0; S[(size_t)j * K + i] = 0.0; } S[(size_t)i * K + i] = 1.0; }
    if (!cholesky(S.data(), K)) return CCAL_ERR_NOT_PD;
    chol_solve(S.data(), K, dc.data());
    double mc = 0.0;
    for (int i = 0; i < K; ++i) {
        const double Dii = lambda > 0.0 ? lambda * clampd(N.Hcc[(size_t)i * K + i], min_diag, max_diag) : 0.0;
        if (!fx[i]) mc += dc[i] * (Dii * dc[i] - N.gc[i]);
    }
    *mc_cam = mc; mc = 0.0;
    // shared block update, clamp to bounds, keep fixed
    for (int c = 0; c < L.n_cams; ++c) {
        double eff[9]; full_to_eff(L, c, intr + (size_t)c * CCAL_PMAX, eff);
        for (int i = 0; i < L.Peff[c]; ++i) {
            if (fx[L.col_theta[c] + i]) continue;
            double v = eff[i] + dc[L.col_theta[c] + i];
            if (cs.has_bound && cs.has_bound[c * CCAL_PMAX + i]) v = std::min(std::max(v, cs.lo[c * CCAL_PMAX + i]), cs.hi[c * CCAL_PMAX + i]);
            eff[i] = v;
        }
        for (int i = 0; i < CCAL_PMAX; ++i) intr_c[(size_t)c * CCAL_PMAX + i] = intr[(size_t)c * CCAL_PMAX + i];
        eff_to_full(L, c, eff, intr_c + (size_t)c * CCAL_PMAX);
        for (int i = 0; i < 6; ++i) extr_c[c * 6 + i] = extr[c * 6 + i] + (c > 0 ? dc[L.col_extr[c] + i] : 0.0);
    }
    // back-substitution of every pose block
    for (int s = 0; s < d->n_slots; ++s) {
        const SlotBlocks& sb = N.slots[s];
        double Cl[36]; std::memcpy(Cl, sb.C, sizeof Cl);
        bool any = false; for (int i = 0; i < 6; ++i) any |= Cl[i * 6 + i] != 0.0;
        if (!any) { for (int i = 0; i < 6; ++i) poses_c[(size_t)s * 6 + i] = poses[(size_t)s * 6 + i]; continue; }
        double Dp[6];
        for (int i = 0; i < 6; ++i) { Dp[i] = lambda > 0.0 ? lambda * clampd(sb.C[i * 6 + i], min_diag, max_diag) : 0.0; Cl[i * 6 + i] += Dp[i]; }
        if (!cholesky(Cl, 6)) return CCAL_ERR_NOT_PD;
        double rhs[6];
        for (int i = 0; i < 6; ++i) { double t = sb.g[i]; for (int j = 0; j < K; ++j) t += sb.B[(size_t)i * K + j] * dc[j]; rhs[i] = -t; }
        chol_solve(Cl, 6, rhs);
        for (int i = 0; i < 6; ++i) { poses_c[(size_t)s * 6 + i] = poses[(size_t)s * 6 + i] + rhs[i]; mc += rhs[i] * (Dp[i] * rhs[i] - sb.g[i]); }
    }
    *mc_pose = mc;
    return CCAL_OK;
}

// Frame-sharded solves: the same hook signature as ccal_allreduce_fn, on a HOST buffer.
typedef int (*allreduce_fn)(void* user, double* buf, size_t count, void* stream);
static int allreduce_normal(Normal& N, allreduce_fn fn, void* user) {
    if (!fn) return 0;
    const int K = N.K;
    std::vector<double> buf((size_t)K * K + 3 * K + 1);
    double* q = buf.data();
    std::memcpy(q, N.S.data(), sizeof(double) * K * K); q += (size_t)K * K;
    std::memcpy(q, N.b.data(), sizeof(double) * K); q += K;
    for (int i = 0; i < K; ++i) q[i] = N.Hcc[(size_t)i * K + i];
    q += K;
    std::memcpy(q, N.gc.data(), sizeof(double) * K); q += K;
    *q = N.cost;
    if (fn(user, buf.data(), buf.size(), nullptr) != 0) return 1;
    q = buf.data();
    std::memcpy(N.S.data(), q, sizeof(double) * K * K); q += (size_t)K * K;
    std::memcpy(N.b.data(), q, sizeof(double) * K); q += K;
    for (int i = 0; i < K; ++i) N.Hcc[(size_t)i * K + i] = q[i];
    q += K;
    std::memcpy(N.gc.data(), q, sizeof(double) * K); q += K;
    N.cost = *q;
    return 0;
}
static int allreduce_scalars(double* v, int n, allreduce_fn fn, void* user) { return fn ? fn(user, v, (size_t)n, nullptr) : 0; }

}  // namespace oracle

using namespace oracle;

// ModelConvertFactor evaluation (see oracle_convert_model below)
namespace {
template <int P>
void convert_eval(int src_model, const double* src, int tgt_model, const double* th, const std::vector<double>& rays,
                  double* H, double* g, double* s_out) {
    using DT = Dual<P>;
    DT tv[P]; for (int i = 0; i < P; ++i) tv[i] = DT::seed(th[i], i);
    for (int i = 0; i < P * P; ++i) H[i] = 0.0;
    for (int i = 0; i < P; ++i) g[i] = 0.0;
    double s = 0.0;
    const size_t M = rays.size() / 3;
    for (size_t k = 0; k < M; ++k) {
        const double* q = &rays[3 * k];
        const bool ok = project_valid(src_model, src, q[0], q[1], q[2]) && project_valid(tgt_model, th, q[0], q[1], q[2]);
        if (!ok) { s += 2.0 * 10000.0 * 10000.0; continue; }              // factors.rs:71: constant, no Jacobian
        double u0, v0; project_one<double>(src_model, src, { q[0], q[1], q[2] }, u0, v0);
        DT u1, v1; project_one<DT>(tgt_model, tv, { DT(q[0]), DT(q[1]), DT(q[2]) }, u1, v1);
        const DT r[2] = { DT(u0) - u1, DT(v0) - v1 };
        for (int a = 0; a < 2; ++a) {
            s += r[a].re * r[a].re;
            for (int i = 0; i < P; ++i) {
                g[i] += r[a].eps[i] * r[a].re;
                for (int j = 0; j < P; ++j) H[i * P + j] += r[a].eps[i] * r[a].eps[j];
            }
        }
    }
    *s_out = s;
}
typedef void (*convert_eval_fn)(int, const double*, int, const double*, const std::vector<double>&, double*, double*, double*);
}  // namespace

extern "C" {

int oracle_model_num_params(int model) { return model_nparams(model); }

int oracle_project(int model, const double* params, int n, const double* xyz, double* uv) {
    if (model_nparams(model) < 0) return CCAL_ERR_INVALID_ARG;
    for (int i = 0; i < n; ++i) {
        V3<double> p = { xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2] };
        project_one<double>(model, params, p, uv[2 * i], uv[2 * i + 1]);
    }
    return CCAL_OK;
}

// ReprojectionFactor / OtherCamReprojectionFactor residual_func with T = f64 (J == NULL) or dual.
int oracle_factor(int model, int xy_same_focal, int other_cam, const double* params_eff,
                  const double* pose0, const double* pose1, const float* p3d, const float* p2d,
                  double* r, double* J) {
    const int P = model_nparams(model); if (P < 0) return CCAL_ERR_INVALID_ARG;
    const int pe = P - (xy_same_focal ? 1 : 0);
    if (!J) {
        if (other_cam) other_cam_reprojection_factor<double>(model, xy_same_focal, params_eff, pose0, pose0 + 3, pose1, pose1 + 3, p3d, p2d, r);
        else reprojection_factor<double>(model, xy_same_focal, params_eff, pose0, pose0 + 3, p3d, p2d, r);
        return CCAL_OK;
    }
    factor_jacobian_dyn(pe + (other_cam ? 12 : 6), model, xy_same_focal, other_cam, pe, params_eff, pose0, pose1, p3d, p2d, r, J);
    return CCAL_OK;
}

// RvecTvec -> Isometry3 -> RvecTvec (tests/types_test.rs:5-20).
int oracle_rvec_tvec_roundtrip(const double* pose_in, double* pose_out) {
    Iso3<double> m = iso_new<double>({ pose_in[3], pose_in[4], pose_in[5] }, { pose_in[0], pose_in[1], pose_in[2] });
    quat_scaled_axis(m.q, pose_out);
    pose_out[3] = m.t.x; pose_out[4] = m.t.y; pose_out[5] = m.t.z;
    return CCAL_OK;
}
// (T_a * T_b).to_rvec_tvec()  -- saved per-camera poses are T_i_0 * T_0_b (src/bin/camera_calibration.rs:279-287)
int oracle_pose_compose(const double* a, const double* b, double* out) {
    Iso3<double> ma = iso_new<double>({ a[3], a[4], a[5] }, { a[0], a[1], a[2] });
    Iso3<double> mb = iso_new<double>({ b[3], b[4], b[5] }, { b[0], b[1], b[2] });
    Iso3<double> m = iso_mul(ma, mb);
    quat_scaled_axis(m.q, out); out[3] = m.t.x; out[4] = m.t.y; out[5] = m.t.z;
    return CCAL_OK;
}
int oracle_pose_inverse(const double* a, double* out) {
    Iso3<double> m = iso_inverse(iso_new<double>({ a[3], a[4], a[5] }, { a[0], a[1], a[2] }));
    quat_scaled_axis(m.q, out); out[3] = m.t.x; out[4] = m.t.y; out[5] = m.t.z;
    return CCAL_OK;
}
int oracle_pose_apply(const double* a, const double* p, double* out) {
    V3<double> q = iso_apply(iso_new<double>({ a[3], a[4], a[5] }, { a[0], a[1], a[2] }), V3<double>{ p[0], p[1], p[2] });
    out[0] = q.x; out[1] = q.y; out[2] = q.z;
    return CCAL_OK;
}

// Mode E over a whole problem, the reference way: one dual-number factor evaluation per corner,
// model and exp-map rebuilt per corner (src/optimization/factors.rs:152-173).  `threads` static
// partition over observation frames (tiny-solver evaluates blocks from a rayon pool).
static int oracle_eval_impl(const ccal_problem_desc* d, const double* intr, const double* poses, const double* extr,
                            int apply_loss, int threads, int reps, double* r_out, double* J_out, bool heap_duals = false) {
    Layout L; if (!make_layout(d, L)) return CCAL_ERR_INVALID_ARG;
    double eff[CCAL_MAX_CAMS][9];
    for (int c = 0; c < L.n_cams; ++c) full_to_eff(L, c, intr + (size_t)c * CCAL_PMAX, eff[c]);
    std::vector<int64_t> joff(d->n_obs + 1, 0);
    for (int o = 0; o < d->n_obs; ++o) joff[o + 1] = joff[o] + (d->obs_offsets[o + 1] - d->obs_offsets[o]) * 2 * L.D[d->obs_cam[o]];
    auto work = [&](int o0, int o1) {
        for (int rep = 0; rep < reps; ++rep)
        for (int o = o0; o < o1; ++o) {
            const int cam = d->obs_cam[o], slot = d->obs_slot[o], D = L.D[cam];
            for (int64_t k = d->obs_offsets[o]; k < d->obs_offsets[o + 1]; ++k) {
                double* r = r_out + 2 * k;
                double* J = J_out + joff[o] + (k - d->obs_offsets[o]) * 2 * D;
                eval_block(d, L, cam, k, eff[cam], poses + (size_t)slot * 6, extr ? extr + (size_t)cam * 6 : nullptr, r, J, heap_duals);
                if (apply_loss && d->huber_delta > 0.0) {
                    const double sw = std::sqrt(huber_weight(r[0] * r[0] + r[1] * r[1], d->huber_delta));
                    r[0] *= sw; r[1] *= sw; for (int i = 0; i < 2 * D; ++i) J[i] *= sw;
                }
            }
        }
    };
    if (threads <= 1) { work(0, d->n_obs); return CCAL_OK; }
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t) {
        const int o0 = (int)((int64_t)d->n_obs * t / threads), o1 = (int)((int64_t)d->n_obs * (t + 1) / threads);
        pool.emplace_back(work, o0, o1);
    }
    for (auto& th : pool) th.join();
    return CCAL_OK;
}

int oracle_eval(const ccal_problem_desc* d, const double* intr, const double* poses, const double* extr,
                int apply_loss, int threads, double* r_out, double* J_out) {
    return oracle_eval_impl(d, intr, poses, extr, apply_loss, threads, 1, r_out, J_out);
}
// CPU-baseline timing helper: every thread repeats its static share `reps` times (amortises thread
// start-up the way a long-lived rayon pool would); returns wall seconds of the whole call.
double oracle_eval_timed(const ccal_problem_desc* d, const double* intr, const double* poses, const double* extr,
                         int threads, int reps, double* r_out, double* J_out) {
    const auto t0 = std::chrono::steady_clock::now();
    if (oracle_eval_impl(d, intr, poses, extr, 0, threads, reps, r_out, J_out) != CCAL_OK) return -1.0;
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// the same with heap-backed duals (DualH, the container tiny-solver instantiates the factors with)
double oracle_eval_timed_heap(const ccal_problem_desc* d, const double* intr, const double* poses, const double* extr,
                              int threads, int reps, double* r_out, double* J_out) {
    const auto t0 = std::chrono::steady_clock::now();
    if (oracle_eval_impl(d, intr, poses, extr, 0, threads, reps, r_out, J_out, true) != CCAL_OK) return -1.0;
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
int oracle_eval_heap(const ccal_problem_desc* d, const double* intr, const double* poses, const double* extr,
                     int apply_loss, int threads, double* r_out, double* J_out) {
    return oracle_eval_impl(d, intr, poses, extr, apply_loss, threads, 1, r_out, J_out, true);
}

int oracle_reduced_dim(const ccal_problem_desc* d) { Layout L; return make_layout(d, L) ? L.K : -1; }

int oracle_build_normal(const ccal_problem_desc* d, const double* intr, const double* poses, const double* extr,
                        double lambda, double min_diag, double max_diag,
                        double* S, double* b, double* cost, double* Hcc_diag, double* gc) {
    Layout L; if (!make_layout(d, L)) return CCAL_ERR_INVALID_ARG;
    Normal N; build_normal(d, L, intr, poses, extr, lambda, min_diag, max_diag, N);
    if (S) {
        std::memcpy(S, N.S.data(), sizeof(double) * N.S.size());
        if (lambda > 0.0) for (int i = 0; i < L.K; ++i) S[(size_t)i * L.K + i] += lambda * clampd(N.Hcc[(size_t)i * L.K + i], min_diag, max_diag);
    }
    if (b) std::memcpy(b, N.b.data(), sizeof(double) * N.b.size());
    if (cost) *cost = N.cost;
    if (Hcc_diag) for (int i = 0; i < L.K; ++i) Hcc_diag[i] = N.Hcc[(size_t)i * L.K + i];
    if (gc) std::memcpy(gc, N.gc.data(), sizeof(double) * L.K);
    return std::isfinite(N.cost) ? CCAL_OK : CCAL_ERR_NOT_PD;
}

double oracle_cost(const ccal_problem_desc* d, const double* intr, const double* poses, const double* extr) {
    Layout L; if (!make_layout(d, L)) return NAN;
    return total_cost(d, L, intr, poses, extr);
}

static inline double stop_error(double cost, int metric) { return metric ? std::sqrt(std::max(cost, 0.0)) : cost; }
// GaussNewtonOptimizer::optimize restated (defaults in ccal_set_defaults) + LM mode.
// lo/hi/has_bound/fixed: [n_cams][CCAL_PMAX] in eff index space (may be NULL).
int oracle_solve(const ccal_problem_desc* d, const double* lo, const double* hi, const uint8_t* has_bound,
                 const uint8_t* fixed, const ccal_solver_opts* o,
                 double* intr, double* poses, double* extr, ccal_report* rep,
                 oracle::allreduce_fn ar, void* ar_user) {
    Layout L; if (!make_layout(d, L) || !o) return CCAL_ERR_INVALID_ARG;
    Constraints cs = { lo, hi, has_bound, fixed };
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<double> ic((size_t)L.n_cams * CCAL_PMAX), pc((size_t)d->n_slots * 6), ec((size_t)L.n_cams * 6);
    std::vector<double> ex0((size_t)L.n_cams * 6, 0.0);
    if (!extr) extr = ex0.data();
    Normal N;
    ccal_report R = {}; R.status = CCAL_OK;
    double cur = total_cost(d, L, intr, poses, extr);
    if (allreduce_scalars(&cur, 1, ar, ar_user)) return CCAL_ERR_HIP;
    R.initial_cost = cur;
    int status = CCAL_OK;
    if (o->method == CCAL_METHOD_GN) {
        for (int it = 0; it < o->max_iterations; ++it) {
            const double last = cur;
            build_normal(d, L, intr, poses, extr, 0.0, 0.0, 0.0, N);
            if (allreduce_normal(N, ar, ar_user)) return CCAL_ERR_HIP;
            if (!std::isfinite(N.cost)) { status = CCAL_ERR_NOT_PD; break; }
            double mcc, mcp;
            status = step(d, L, cs, N, 0.0, 0.0, 0.0, intr, poses, extr, ic.data(), pc.data(), ec.data(), &mcc, &mcp);
            if (status != CCAL_OK) break;
            std::memcpy(intr, ic.data(), sizeof(double) * ic.size());
            std::memcpy(poses, pc.data(), sizeof(double) * pc.size());
            std::memcpy(extr == ex0.data() ? ex0.data() : extr, ec.data(), sizeof(double) * ec.size());
            cur = total_cost(d, L, intr, poses, extr);
            if (allreduce_scalars(&cur, 1, ar, ar_user)) return CCAL_ERR_HIP;
            R.iterations++;
            if (o->verbose) std::printf("[oracle GN] iter %d cost %.12g\n", it, cur);
            // what tiny-solver calls "error": squared norm (metric 0) or norm (metric 1) of the loss-corrected residuals
            const double le = stop_error(last, o->error_metric), ce = stop_error(cur, o->error_metric);
            if (ce < o->min_error) break;
            if (std::isnan(cur)) { status = CCAL_ERR_NONFINITE; break; }
            if (std::fabs(le - ce) < o->min_abs_error_decrease) break;
            if (std::fabs(le - ce) / le < o->min_rel_error_decrease) break;
            if (it == o->max_iterations - 1) status = CCAL_ERR_NO_CONVERGENCE;
        }
    } else {
        double radius = o->lm_initial_radius, dec = 2.0;
        bool have = false;
        for (int it = 0; it < o->max_iterations; ++it) {
            const double lambda = 1.0 / radius;
            // Gram at x is lambda-independent, but this oracle simply rebuilds.
            build_normal(d, L, intr, poses, extr, lambda, o->lm_min_diagonal, o->lm_max_diagonal, N);
            (void)have;
            if (allreduce_normal(N, ar, ar_user)) return CCAL_ERR_HIP;
            if (!std::isfinite(N.cost)) { status = CCAL_ERR_NOT_PD; break; }
            double mcc = 0.0, mcp = 0.0;
            int st = step(d, L, cs, N, lambda, o->lm_min_diagonal, o->lm_max_diagonal, intr, poses, extr, ic.data(), pc.data(), ec.data(), &mcc, &mcp);
            R.iterations++;
            double cand = NAN, rho = -1.0, mc = 0.0;
            double pair[2] = { st == CCAL_OK ? total_cost(d, L, ic.data(), pc.data(), ec.data()) : NAN, mcp };
            if (allreduce_scalars(pair, 2, ar, ar_user)) return CCAL_ERR_HIP;
            if (st == CCAL_OK) { cand = pair[0]; mc = mcc + pair[1]; rho = (cur - cand) / mc; }
            const double mce = o->error_metric ? std::sqrt(std::max(cur, 0.0)) - std::sqrt(std::max(cur - mc, 0.0)) : mc;
            if (st == CCAL_OK && std::isfinite(cand) && mc >= 0.0 &&
                (mce < o->min_abs_error_decrease || mce < o->min_rel_error_decrease * stop_error(cur, o->error_metric))) {
                // predicted decrease below the thresholds: converged
                if (cand < cur) {
                    std::memcpy(intr, ic.data(), sizeof(double) * ic.size());
                    std::memcpy(poses, pc.data(), sizeof(double) * pc.size());
                    std::memcpy(extr == ex0.data() ? ex0.data() : extr, ec.data(), sizeof(double) * ec.size());
                    cur = cand; R.lm_accepted++;
                }
                break;
            }
            if (st == CCAL_OK && std::isfinite(cand) && mc > 0.0 && rho > 0.0) {
                const double last = cur;
                std::memcpy(intr, ic.data(), sizeof(double) * ic.size());
                std::memcpy(poses, pc.data(), sizeof(double) * pc.size());
                std::memcpy(extr == ex0.data() ? ex0.data() : extr, ec.data(), sizeof(double) * ec.size());
                cur = cand; R.lm_accepted++;
                const double t = 2.0 * rho - 1.0;
                radius = std::min(1e16, radius / std::max(1.0 / 3.0, 1.0 - t * t * t));
                dec = 2.0;
                if (o->verbose) std::printf("[oracle LM] iter %d accept cost %.12g rho %.3g radius %.3g\n", it, cur, rho, radius);
                const double le = stop_error(last, o->error_metric), ce = stop_error(cur, o->error_metric);
                if (ce < o->min_error) break;
                if (std::fabs(le - ce) < o->min_abs_error_decrease) break;
                if (std::fabs(le - ce) / le < o->min_rel_error_decrease) break;
            } else {
                R.lm_rejected++;
                radius /= dec; dec *= 2.0;
                if (o->verbose) std::printf("[oracle LM] iter %d reject (cand %.12g) radius %.3g\n", it, cand, radius);
                if (radius < 1e-32) { status = CCAL_ERR_NO_CONVERGENCE; break; }
            }
            if (it == o->max_iterations - 1) status = CCAL_ERR_NO_CONVERGENCE;
        }
    }
    R.final_cost = cur; R.status = status;
    R.solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (rep) *rep = R;
    return status;
}

// The reference's literal formulation for small problems: the FULL normal equations over
// [theta | extr | every pose] solved by one dense Cholesky per iteration (what tiny-solver does with a
// sparse Cholesky).  Used by tests to show that the Schur-reduced solve above is the same step.
int oracle_gn_step_dense(const ccal_problem_desc* d, const uint8_t* fixed,
                         const double* intr, const double* poses, const double* extr, double* dx_out /* K + 6 n_slots */) {
    Layout L; if (!make_layout(d, L)) return CCAL_ERR_INVALID_ARG;
    const int K = L.K, n = K + 6 * d->n_slots;
    std::vector<double> H((size_t)n * n, 0.0), g(n, 0.0);
    double eff[CCAL_MAX_CAMS][9];
    for (int c = 0; c < L.n_cams; ++c) full_to_eff(L, c, intr + (size_t)c * CCAL_PMAX, eff[c]);
    double J[2 * 21], r[2]; int colmap[21];
    std::vector<double> ex0((size_t)L.n_cams * 6, 0.0); if (!extr) extr = ex0.data();
    for (int o = 0; o < d->n_obs; ++o) {
        const int cam = d->obs_cam[o], slot = d->obs_slot[o], D = L.D[cam], Pe = L.Peff[cam];
        for (int i = 0; i < Pe; ++i) colmap[i] = L.col_theta[cam] + i;
        for (int i = 0; i < 6; ++i) colmap[Pe + i] = K + 6 * slot + i;
        if (cam > 0) for (int i = 0; i < 6; ++i) colmap[Pe + 6 + i] = L.col_extr[cam] + i;
        for (int64_t k = d->obs_offsets[o]; k < d->obs_offsets[o + 1]; ++k) {
            eval_block(d, L, cam, k, eff[cam], poses + (size_t)slot * 6, extr + (size_t)cam * 6, r, J);
            const double s = r[0] * r[0] + r[1] * r[1];
            const double w = d->huber_delta > 0.0 ? huber_weight(s, d->huber_delta) : 1.0;
            for (int row = 0; row < 2; ++row) for (int a = 0; a < D; ++a) {
                g[colmap[a]] += w * J[row * D + a] * r[row];
                for (int bq = 0; bq < D; ++bq) H[(size_t)colmap[a] * n + colmap[bq]] += w * J[row * D + a] * J[row * D + bq];
            }
        }
    }
    for (int c = 0; c < L.n_cams; ++c) for (int i = 0; i < L.Peff[c]; ++i) if (fixed && fixed[c * CCAL_PMAX + i]) {
        const int q = L.col_theta[c] + i;
        for (int j = 0; j < n; ++j) { H[(size_t)q * n + j] = 0.0; H[(size_t)j * n + q] = 0.0; } H[(size_t)q * n + q] = 1.0; g[q] = 0.0;
    }
    if (!cholesky(H.data(), n)) return CCAL_ERR_NOT_PD;
    for (int i = 0; i < n; ++i) dx_out[i] = -g[i];
    chol_solve(H.data(), n, dx_out);
    return CCAL_OK;
}

// validation(): per-corner Euclidean reprojection error (src/util.rs:733-745).
int oracle_reprojection_errors(const ccal_problem_desc* d, const double* intr, const double* poses, const double* extr, double* err) {
    Layout L; if (!make_layout(d, L)) return CCAL_ERR_INVALID_ARG;
    std::vector<double> ex0((size_t)L.n_cams * 6, 0.0); if (!extr) extr = ex0.data();
    double eff[CCAL_MAX_CAMS][9];
    for (int c = 0; c < L.n_cams; ++c) full_to_eff(L, c, intr + (size_t)c * CCAL_PMAX, eff[c]);
    for (int o = 0; o < d->n_obs; ++o) {
        const int cam = d->obs_cam[o], slot = d->obs_slot[o];
        const double* ps = poses + (size_t)slot * 6; const double* ex = extr + (size_t)cam * 6;
        for (int64_t k = d->obs_offsets[o]; k < d->obs_offsets[o + 1]; ++k) {
            const float p3[3] = { d->p3d_x[k], d->p3d_y[k], d->p3d_z[k] };
            const float p2[2] = { d->p2d_u[k], d->p2d_v[k] };
            double r[2];
            if (cam > 0) other_cam_reprojection_factor<double>(L.model[cam], L.xy, eff[cam], ps, ps + 3, ex, ex + 3, p3, p2, r);
            else reprojection_factor<double>(L.model[cam], L.xy, eff[cam], ps, ps + 3, p3, p2, r);
            err[k] = std::sqrt(r[0] * r[0] + r[1] * r[1]);
        }
    }
    return CCAL_OK;
}
// median = e[len/2]; avg_99 = sum_{i < len*99/100} e_i / (len*99/100)   (src/util.rs:782-794)
int oracle_validation_stats(const double* err, int64_t n, double* avg_99, double* median) {
    if (n <= 0) return CCAL_ERR_INVALID_ARG;
    std::vector<double> e(err, err + n);
    std::sort(e.begin(), e.end());
    *median = e[n / 2];
    const int64_t n99 = n * 99 / 100;
    double s = 0.0; for (int64_t i = 0; i < n99; ++i) s += e[i] / (double)n99;
    *avg_99 = s;
    return CCAL_OK;
}

// One SE3Factor block with forward-mode duals: r[6] and J[6][6] (row-major, columns rvec | tvec)
int oracle_se3_factor(const double* pose_0_b, const double* pose_i_b, const double* x, double* r, double* J) {
    using DT = Dual<6>;
    DT xv[6]; for (int i = 0; i < 6; ++i) xv[i] = DT::seed(x[i], i);
    DT rr[6]; se3_factor<DT>(pose_0_b, pose_i_b, xv, xv + 3, rr);
    for (int i = 0; i < 6; ++i) { r[i] = rr[i].re; for (int c = 0; c < 6; ++c) J[i * 6 + c] = rr[i].eps[c]; }
    return CCAL_OK;
}

// init_camera_extrinsic (src/util.rs:511-561): SE3Factor per common frame, HuberLoss(0.5), Gauss-Newton
// with dual-number Jacobians and tiny-solver's default thresholds, from t_i_b[0] * t_0_b[0]^-1.
int oracle_init_camera_extrinsic(const double* poses0, const double* posesi, int n, double* out6, int* iters) {
    if (n < 1) return CCAL_ERR_INVALID_ARG;
    using DT = Dual<6>;
    const double delta = 0.5;
    Iso3<double> a = iso_new<double>({ poses0[3], poses0[4], poses0[5] }, { poses0[0], poses0[1], poses0[2] });
    Iso3<double> b = iso_new<double>({ posesi[3], posesi[4], posesi[5] }, { posesi[0], posesi[1], posesi[2] });
    Iso3<double> init = iso_mul(b, iso_inverse(a));
    double x[6]; quat_scaled_axis(init.q, x); x[3] = init.t.x; x[4] = init.t.y; x[5] = init.t.z;
    auto total = [&](const double* xx) {
        double c = 0;
        for (int k = 0; k < n; ++k) {
            double r[6]; se3_factor<double>(poses0 + 6 * k, posesi + 6 * k, xx, xx + 3, r);
            double s = 0; for (int i = 0; i < 6; ++i) s += r[i] * r[i];
            c += huber_weight(s, delta) * s;
        }
        return c;
    };
    double cur = total(x);
    int it_done = 0, status = CCAL_OK;
    for (int it = 0; it < 100; ++it) {
        const double last = cur;
        double H[36] = { 0 }, g[6] = { 0 };
        DT xv[6]; for (int i = 0; i < 6; ++i) xv[i] = DT::seed(x[i], i);
        for (int k = 0; k < n; ++k) {
            DT r[6]; se3_factor<DT>(poses0 + 6 * k, posesi + 6 * k, xv, xv + 3, r);
            double s = 0; for (int i = 0; i < 6; ++i) s += r[i].re * r[i].re;
            const double w = huber_weight(s, delta);
            for (int i = 0; i < 6; ++i) for (int c = 0; c < 6; ++c) {
                g[c] += w * r[i].eps[c] * r[i].re;
                for (int e = 0; e < 6; ++e) H[c * 6 + e] += w * r[i].eps[c] * r[i].eps[e];
            }
        }
        if (!cholesky(H, 6)) { status = CCAL_ERR_NOT_PD; break; }
        double dx[6]; for (int i = 0; i < 6; ++i) dx[i] = -g[i];
        chol_solve(H, 6, dx);
        for (int i = 0; i < 6; ++i) x[i] += dx[i];
        cur = total(x); ++it_done;
        if (cur < 1e-10) break;
        if (std::isnan(cur)) { status = CCAL_ERR_NONFINITE; break; }
        if (std::fabs(last - cur) < 1e-5) break;
        if (std::fabs(last - cur) / last < 1e-5) break;
    }
    if (iters) *iters = it_done;
    std::memcpy(out6, x, sizeof x);
    return status;
}

// util::convert_model (src/util.rs:224-282) with ModelConvertFactor (src/optimization/factors.rs:10-76).
// ONE residual block of dimension 2 M over the M grid points the source model can unproject, HuberLoss(1.0) on
// the whole block, variable "params" = all target intrinsics, Gauss-Newton with the reference's bounds
// (lo/hi/has_bound, [CCAL_PMAX]) and the last `disabled` distortion parameters fixed at 0.
// n_points_out: M.  Returns the solver status.

int oracle_convert_model(int src_model, const double* src, int tgt_model, double* tgt_io, double width, double height,
                         int disabled, const double* lo, const double* hi, const uint8_t* has_bound,
                         int* n_points_out, ccal_report* rep) {
    const int P = model_nparams(tgt_model);
    if (P < 0 || model_nparams(src_model) < 0 || disabled < 0 || disabled > P - 4) return CCAL_ERR_INVALID_ARG;
    ccal_report R = {};
    if (src_model == UCM && tgt_model == EUCM) {                              // src/util.rs:229-235
        for (int i = 0; i < 5; ++i) tgt_io[i] = src[i];
        tgt_io[5] = 1.0;
        if (n_points_out) *n_points_out = 0;
        if (rep) *rep = R;
        return CCAL_OK;
    }
    const double big = std::max(width, height);
    const uint32_t edge = (uint32_t)big / 100u;                               // src/util.rs:245
    const size_t steps = (size_t)(big / 30.0);                                // :246
    if (steps == 0 || (uint32_t)height <= edge || (uint32_t)width <= edge) return CCAL_ERR_INVALID_ARG;
    std::vector<double> rays;
    for (uint32_t r = edge; r < (uint32_t)height - edge; r += (uint32_t)steps)          // factors.rs:35-39
        for (uint32_t c = edge; c < (uint32_t)width - edge; c += (uint32_t)steps) {
            double q[3];
            if (unproject_one(src_model, src, (double)c, (double)r, q)) { rays.push_back(q[0]); rays.push_back(q[1]); rays.push_back(q[2]); }
        }
    if (n_points_out) *n_points_out = (int)(rays.size() / 3);
    double th[9];
    for (int i = 0; i < P; ++i) th[i] = tgt_io[i];
    for (int i = 0; i < 4; ++i) th[i] = src[i];                               // src/util.rs:256-258
    std::vector<uint8_t> fx(P, 0);
    for (int i = 0; i < disabled; ++i) { fx[P - 1 - i] = 1; th[P - 1 - i] = 0.0; }   // src/util.rs:58-70
    const convert_eval_fn ev = P == 5 ? convert_eval<5> : P == 6 ? convert_eval<6> : P == 8 ? convert_eval<8> : convert_eval<9>;
    double H[81], g[9], s;
    ev(src_model, src, tgt_model, th, rays, H, g, &s);
    double cur = huber_weight(s, 1.0) * s;
    R.initial_cost = cur;
    int status = CCAL_OK;
    for (int it = 0; it < 100; ++it) {
        const double last = cur;
        // the corrector multiplies J and r by sqrt(rho'): one block => one common factor on H and g
        const double w = huber_weight(s, 1.0);
        double S[81], dx[9];
        for (int i = 0; i < P * P; ++i) S[i] = w * H[i];
        for (int i = 0; i < P; ++i) dx[i] = fx[i] ? 0.0 : -w * g[i];
        for (int i = 0; i < P; ++i) if (fx[i]) { for (int j = 0; j < P; ++j) { S[i * P + j] = 0.0; S[j * P + i] = 0.0; } S[i * P + i] = 1.0; }
        if (!cholesky(S, P)) { status = CCAL_ERR_NOT_PD; break; }
        chol_solve(S, P, dx);
        for (int i = 0; i < P; ++i) {
            if (fx[i]) continue;
            double v = th[i] + dx[i];
            if (has_bound && has_bound[i]) v = std::min(std::max(v, lo[i]), hi[i]);
            th[i] = v;
        }
        ev(src_model, src, tgt_model, th, rays, H, g, &s);
        cur = huber_weight(s, 1.0) * s;
        R.iterations++;
        if (cur < 1e-10) break;
        if (std::isnan(cur)) { status = CCAL_ERR_NONFINITE; break; }
        if (std::fabs(last - cur) < 1e-5) break;
        if (std::fabs(last - cur) / last < 1e-5) break;
        if (it == 99) status = CCAL_ERR_NO_CONVERGENCE;
    }
    R.final_cost = cur; R.status = status;
    if (rep) *rep = R;
    for (int i = 0; i < P; ++i) tgt_io[i] = th[i];
    return status;
}

// threads of oracle_solve's passes over the blocks (bench.py's all-cores CPU baseline of the optimizer); 1 = serial
int oracle_set_solve_threads(int n) { const int old = g_solve_threads; g_solve_threads = n > 0 ? n : 1; return old; }
int oracle_hardware_threads(void) { return (int)std::thread::hardware_concurrency(); }
// CPUs this process may actually run on: the scheduler affinity mask (a container / cgroup cpuset shrinks it)
int oracle_usable_cpus(void) {
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) != 0) return (int)std::thread::hardware_concurrency();
    return CPU_COUNT(&set);
}

}  // extern "C"
