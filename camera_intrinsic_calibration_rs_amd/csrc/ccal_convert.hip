// util::convert_model (src/util.rs:224-282) with ModelConvertFactor (src/optimization/factors.rs:10-76):
// fit the target model's intrinsics so that it reproduces the source model over a pixel grid.
//   grid      rows/cols from edge = max(w,h)/100 in steps of max(w,h)/30            (util.rs:245-246, factors.rs:35-39)
//   rays      source.unproject(grid), the points it can unproject                   (factors.rs:40-44)
//   residual  ONE block of dimension 2 M: source.project(ray) - target.project(ray); 10000 where either
//             projection is undefined (factors.rs:58-73); HuberLoss(1.0) on the whole block (util.rs:252)
//   solve     Gauss-Newton on "params" = all target intrinsics, first four initialised from the source
//             (util.rs:256-258), the reference's bounds, last k distortion parameters fixed at 0 (util.rs:265-274)
// About 900 points and at most 9 unknowns: two small kernels (rays once; Gram per iteration on one workgroup,
// lane-private triangle + shuffle/LDS reduction, fixed order) and the 9 x 9 solve on the host.
#include <algorithm>
#include <cmath>
#include <cstring>

#include <cstdio>
#include <string>

#include "ccal_device.hpp"
#include "ccal_internal.hpp"

#define HIP_TRY(ctx, expr)                                                                         \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                        \
            return CCAL_ERR_HIP;                                                                   \
        }                                                                                          \
    } while (0)
#define HIP_TRYN(ctx, expr) HIP_TRY(ctx, expr)      /* inside lambdas returning int */

namespace ccal {

// Model inverse as published (UCM/EUCM: Usenko et al. 2018; KB4: Newton on theta; OPENCV5: fixed-point
// undistortion); rays may point backwards (z <= 0) for fisheye models.  false = `unproject` gives None.
template <int MODEL>
__device__ bool unproject_ray(const double* th, double small_radius, double u, double v, double& x, double& y, double& z) {
    const double mx = (u - th[2]) / th[0], my = (v - th[3]) / th[1];
    const double r2 = mx * mx + my * my;
    if constexpr (MODEL == kUCM || MODEL == kEUCM) {
        const double alpha = th[4], beta = (MODEL == kEUCM) ? th[5] : 1.0;
        if (alpha > 0.5 && r2 > 1.0 / (beta * (2.0 * alpha - 1.0))) return false;
        const double t1 = 1.0 - (2.0 * alpha - 1.0) * beta * r2;
        if (t1 < 0.0) return false;
        const double mz = (1.0 - beta * alpha * alpha * r2) / (alpha * sqrt(t1) + (1.0 - alpha));
        const double n = sqrt(r2 + mz * mz);
        x = mx / n; y = my / n; z = mz / n;
        return true;
    } else if constexpr (MODEL == kKB4) {
        const double r = sqrt(r2);
        if (r < small_radius) { x = mx; y = my; z = 1.0; return true; }
        double t = r;
        for (int it = 0; it < 20; ++it) {
            const double t2 = t * t;
            const double f = t * (1.0 + t2 * (th[4] + t2 * (th[5] + t2 * (th[6] + t2 * th[7])))) - r;
            const double fp = 1.0 + t2 * (3.0 * th[4] + t2 * (5.0 * th[5] + t2 * (7.0 * th[6] + t2 * 9.0 * th[7])));
            const double dt = f / fp;
            t -= dt;
            if (fabs(dt) < 1e-14) break;
        }
        if (!(t > 0.0) || !(t < 3.141592653589793)) return false;
        const double s = sin(t) / r;
        x = mx * s; y = my * s; z = cos(t);
        return true;
    } else {
        const double k1 = th[OCV5_K1], k2 = th[OCV5_K2], p1 = th[OCV5_P1], p2 = th[OCV5_P2], k3 = th[OCV5_K3];
        double xx = mx, yy = my;
        for (int it = 0; it < 50; ++it) {
            const double q = xx * xx + yy * yy;
            const double rad = 1.0 + q * (k1 + q * (k2 + q * k3));
            const double dx = 2.0 * p1 * xx * yy + p2 * (q + 2.0 * xx * xx);
            const double dy = p1 * (q + 2.0 * yy * yy) + 2.0 * p2 * xx * yy;
            xx = (mx - dx) / rad; yy = (my - dy) / rad;
        }
        const double q = xx * xx + yy * yy, rad = 1.0 + q * (k1 + q * (k2 + q * k3));
        const double ex = xx * rad + 2.0 * p1 * xx * yy + p2 * (q + 2.0 * xx * xx) - mx;
        const double ey = yy * rad + p1 * (q + 2.0 * yy * yy) + 2.0 * p2 * xx * yy - my;
        if (!(fabs(ex) + fabs(ey) < 1e-9)) return false;
        const double n = sqrt(q + 1.0);
        x = xx / n; y = yy / n; z = 1.0 / n;
        return true;
    }
}

// Is `project` defined for this camera-frame point (the Option of GenericModel::project)?
template <int MODEL>
__device__ bool project_valid(const double* th, double x, double y, double z) {
    if constexpr (MODEL == kUCM || MODEL == kEUCM) {
        const double alpha = th[4], beta = (MODEL == kEUCM) ? th[5] : 1.0;
        const double d = sqrt(beta * (x * x + y * y) + z * z);
        const double w = alpha <= 0.5 ? alpha / (1.0 - alpha) : (1.0 - alpha) / alpha;
        return z > -w * d;
    } else if constexpr (MODEL == kKB4) {
        return x * x + y * y + z * z > 0.0;
    } else {
        return z > 1e-9;
    }
}

constexpr int CONV_REC = 6;          // x, y, z, u0, v0, state (0 dropped, 1 both projections defined so far, 2 source undefined)

struct ConvArgs {
    const double* src; const double* tgt;       // device copies of the parameter vectors
    double* rays;                               // [n_grid][CONV_REC]
    int32_t n_rows, n_cols, edge, steps;
    double* out;                                // [P (P+1)/2 | P | s | n_points]
    ModelRt rt;                                 // the context's conventions; the OPENCV5 order is the canonical one here: the
                                                // parameter vectors are permuted at the API boundary (ccal_convert_model)
};

template <int SRC>
__global__ __launch_bounds__(256) void k_convert_rays(const ConvArgs a) {
    const int n = a.n_rows * a.n_cols;
    double th[th_len<SRC>()];
    load_theta<SRC, false>(a.src, a.rt, th);
    for (int k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256) {
        const int r = a.edge + (k / a.n_cols) * a.steps, c = a.edge + (k % a.n_cols) * a.steps;
        double x = 0.0, y = 0.0, z = 0.0, u0 = 0.0, v0 = 0.0, state = 0.0;
        if (unproject_ray<SRC>(th, a.rt.unproject_eps, (double)c, (double)r, x, y, z)) {
            if (project_valid<SRC>(th, x, y, z)) { project_uv<SRC>(th, x, y, z, u0, v0); state = 1.0; }
            else state = 2.0;
        }
        double* o = a.rays + (int64_t)k * CONV_REC;
        o[0] = x; o[1] = y; o[2] = z; o[3] = u0; o[4] = v0; o[5] = state;
    }
}

template <int TGT>
__global__ __launch_bounds__(256) void k_convert_gram(const ConvArgs a) {
    constexpr int P = model_np(TGT), ND = P - 4, NT = P * (P + 1) / 2, NA = NT + P + 2;
    __shared__ double part[4][NA];
    double th[th_len<TGT>()];
    load_theta<TGT, false>(a.tgt, a.rt, th);
    double acc[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) acc[i] = 0.0;
    const int n = a.n_rows * a.n_cols;
    for (int k = threadIdx.x; k < n; k += 256) {
        const double* q = a.rays + (int64_t)k * CONV_REC;
        const double state = q[5];
        if (state == 0.0) continue;
        acc[NT + P + 1] += 1.0;
        const double x = q[0], y = q[1], z = q[2];
        if (state == 2.0 || !project_valid<TGT>(th, x, y, z)) { acc[NT + P] += 2.0 * 10000.0 * 10000.0; continue; }
        double mx, my, dmx[3], dmy[3], ddx[ND], ddy[ND];
        project_partials<TGT>(th, x, y, z, mx, my, dmx, dmy, ddx, ddy);
        // r = source - target, so dr/dtheta = -d(target)/dtheta
        double Ju[P], Jv[P];
        Ju[0] = -mx; Ju[1] = 0.0; Ju[2] = -1.0; Ju[3] = 0.0;
        Jv[0] = 0.0; Jv[1] = -my; Jv[2] = 0.0; Jv[3] = -1.0;
#pragma unroll
        for (int i = 0; i < ND; ++i) { Ju[4 + i] = -th[0] * ddx[i]; Jv[4 + i] = -th[1] * ddy[i]; }
        const double ru = q[3] - (th[0] * mx + th[2]), rv = q[4] - (th[1] * my + th[3]);
        int e = 0;
#pragma unroll
        for (int i = 0; i < P; ++i) {
#pragma unroll
            for (int j = 0; j <= i; ++j) { acc[e] += Ju[i] * Ju[j] + Jv[i] * Jv[j]; ++e; }
        }
#pragma unroll
        for (int i = 0; i < P; ++i) acc[NT + i] += Ju[i] * ru + Jv[i] * rv;
        acc[NT + P] += ru * ru + rv * rv;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        double v = acc[i];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) part[wave][i] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NA; i += 256) a.out[i] = (part[0][i] + part[1][i]) + (part[2][i] + part[3][i]);
}

namespace {
bool chol_host(double* A, int n) {
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
void chol_solve_host(const double* L, int n, double* x) {
    for (int i = 0; i < n; ++i) { double t = x[i]; for (int k = 0; k < i; ++k) t -= L[i * n + k] * x[k]; x[i] = t / L[i * n + i]; }
    for (int i = n - 1; i >= 0; --i) { double t = x[i]; for (int k = i + 1; k < n; ++k) t -= L[k * n + i] * x[k]; x[i] = t / L[i * n + i]; }
}
// bounds of set_problem_parameter_bound (src/util.rs:29-48); distortion part from the context's conventions table
void reference_bounds(const ccal_model_conventions& cv, int model, double w, double h, double* lo, double* hi) {
    lo[0] = 0.0; hi[0] = 10000.0; lo[1] = 0.0; hi[1] = 10000.0; lo[2] = 0.0; hi[2] = w; lo[3] = 0.0; hi[3] = h;
    for (int i = 4; i < model_np(model); ++i) { lo[i] = cv.dist_lo[model][i - 4]; hi[i] = cv.dist_hi[model][i - 4]; }
}
inline double huber_w(double s, double delta) { return s <= delta * delta ? 1.0 : delta / std::sqrt(s); }
}  // namespace

}  // namespace ccal

using namespace ccal;

extern "C" int ccal_convert_model(ccal_ctx* ctx, int src_model, const double* src_params, int tgt_model,
                                  double* tgt_params_io, double width, double height, int disabled_distortions,
                                  const ccal_solver_opts* opts, ccal_report* rep) {
    if (!ctx) return CCAL_ERR_INVALID_ARG;
    CCAL_API_TRY
    const int PS = ccal_model_num_params(src_model), P = ccal_model_num_params(tgt_model);
    if (PS < 0 || P < 0 || !src_params || !tgt_params_io || disabled_distortions < 0 || disabled_distortions > P - 4 ||
        !(width >= 1.0) || !(height >= 1.0)) { ctx->err = "ccal_convert_model: invalid argument"; return CCAL_ERR_INVALID_ARG; }
    ccal_report R = {};
    if (src_model == kUCM && tgt_model == kEUCM) {                  // src/util.rs:229-235: closed form
        for (int i = 0; i < 5; ++i) tgt_params_io[i] = src_params[i];
        tgt_params_io[5] = 1.0;
        if (rep) *rep = R;
        return CCAL_OK;
    }
    if (src_model == kUCM && tgt_model == CCAL_MODEL_EUCMT) {      // src/util.rs:236-243: beta = 1, both tangential terms 0
        for (int i = 0; i < 5; ++i) tgt_params_io[i] = src_params[i];
        tgt_params_io[5] = 1.0; tgt_params_io[6] = 0.0; tgt_params_io[7] = 0.0;
        if (rep) *rep = R;
        return CCAL_OK;
    }
    if (src_model >= kNumModels || tgt_model >= kNumModels) {
        // EUCMT is a parameter container here: its projection lives only in the absent camera-intrinsic-model crate
        ctx->err = "ccal_convert_model: EUCMT can only be the target of the closed-form UCM conversion";
        return CCAL_ERR_UNSUPPORTED;
    }
    ccal_solver_opts o;
    if (opts) o = *opts; else ccal_set_defaults(&o);
    const double big = std::max(width, height);
    const uint32_t edge = (uint32_t)big / 100u;
    const int steps = (int)(big / 30.0);
    if (steps < 1 || (uint32_t)height <= 2 * edge || (uint32_t)width <= 2 * edge) { ctx->err = "ccal_convert_model: image too small for the grid"; return CCAL_ERR_INVALID_ARG; }
    const int n_rows = (int)(((uint32_t)height - 2 * edge + steps - 1) / steps);
    const int n_cols = (int)(((uint32_t)width - 2 * edge + steps - 1) / steps);
    const int n_grid = n_rows * n_cols;
    const int NT = P * (P + 1) / 2, NA = NT + P + 2;

    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    double* d_buf = nullptr;                    // [src 10 | tgt 10 | out 64 | rays]
    HIP_TRY(ctx, hipMalloc((void**)&d_buf, sizeof(double) * (size_t)(20 + 64 + (size_t)n_grid * CONV_REC)));
    struct Free { double* p; ~Free() { (void)hipFree(p); } } guard{ d_buf };
    // The fit runs in the kernels' canonical OPENCV5 order (k1, k2, p1, p2, k3): both parameter vectors are permuted HERE, at the
    // boundary (ccal_model_conventions.ocv5_order), the kernels get the identity
    ModelRt rt = model_rt(ctx);
    rt.ocv5_perm = kOcv5IdentityPerm;
    const int32_t* ord = ctx->conv.ocv5_order;
    auto canon = [&](int model, int i) { return (model == kOCV5 && i >= 4) ? 4 + ord[i - 4] : i; };     // canonical index -> caller's index
    ConvArgs a{ d_buf, d_buf + 10, d_buf + 84, n_rows, n_cols, (int32_t)edge, steps, d_buf + 20, rt };

    double th[CCAL_PMAX] = { 0 }, lo[CCAL_PMAX], hi[CCAL_PMAX], src_c[CCAL_PMAX] = { 0 };
    for (int i = 0; i < PS; ++i) src_c[i] = src_params[canon(src_model, i)];
    for (int i = 0; i < P; ++i) th[i] = tgt_params_io[canon(tgt_model, i)];
    for (int i = 0; i < 4; ++i) th[i] = src_params[i];                      // util.rs:256-258
    bool fx[CCAL_PMAX] = { false };
    for (int i = 0; i < disabled_distortions; ++i) {                        // the LAST k of the caller's vector (src/util.rs:58-70)
        for (int c = 0; c < P; ++c) if (canon(tgt_model, c) == P - 1 - i) { fx[c] = true; th[c] = 0.0; }
    }
    reference_bounds(ctx->conv, tgt_model, width, height, lo, hi);

    HIP_TRY(ctx, hipMemcpyAsync(d_buf, src_c, sizeof(double) * PS, hipMemcpyHostToDevice, st));
    switch (src_model) {
        case kUCM: hipLaunchKernelGGL(k_convert_rays<kUCM>, dim3((n_grid + 255) / 256), dim3(256), 0, st, a); break;
        case kEUCM: hipLaunchKernelGGL(k_convert_rays<kEUCM>, dim3((n_grid + 255) / 256), dim3(256), 0, st, a); break;
        case kKB4: hipLaunchKernelGGL(k_convert_rays<kKB4>, dim3((n_grid + 255) / 256), dim3(256), 0, st, a); break;
        default: hipLaunchKernelGGL(k_convert_rays<kOCV5>, dim3((n_grid + 255) / 256), dim3(256), 0, st, a); break;
    }
    HIP_TRY(ctx, hipGetLastError());

    double out[64];
    auto eval = [&](const double* t) -> int {
        HIP_TRYN(ctx, hipMemcpyAsync(d_buf + 10, t, sizeof(double) * P, hipMemcpyHostToDevice, st));
        switch (tgt_model) {
            case kUCM: hipLaunchKernelGGL(k_convert_gram<kUCM>, dim3(1), dim3(256), 0, st, a); break;
            case kEUCM: hipLaunchKernelGGL(k_convert_gram<kEUCM>, dim3(1), dim3(256), 0, st, a); break;
            case kKB4: hipLaunchKernelGGL(k_convert_gram<kKB4>, dim3(1), dim3(256), 0, st, a); break;
            default: hipLaunchKernelGGL(k_convert_gram<kOCV5>, dim3(1), dim3(256), 0, st, a); break;
        }
        HIP_TRYN(ctx, hipGetLastError());
        HIP_TRYN(ctx, hipMemcpyAsync(out, d_buf + 20, sizeof(double) * NA, hipMemcpyDeviceToHost, st));
        HIP_TRYN(ctx, hipStreamSynchronize(st));
        return 0;
    };
    if (eval(th) != 0) return CCAL_ERR_HIP;
    if (out[NT + P + 1] < 1.0) { ctx->err = "ccal_convert_model: the source model cannot unproject any grid point"; return CCAL_ERR_INVALID_ARG; }
    double s = out[NT + P];
    double cur = huber_w(s, 1.0) * s;
    R.initial_cost = cur;
    int status = CCAL_OK;
    if (!std::isfinite(cur)) status = CCAL_ERR_NONFINITE;
    for (int it = 0; status == CCAL_OK && it < o.max_iterations; ++it) {
        const double last = cur;
        const double w = huber_w(s, 1.0);                 // one block: the corrector is a common factor of H and g
        double S[81], dx[9];
        int e = 0;
        for (int i = 0; i < P; ++i) for (int j = 0; j <= i; ++j) { S[i * P + j] = S[j * P + i] = w * out[e]; ++e; }
        for (int i = 0; i < P; ++i) dx[i] = fx[i] ? 0.0 : -w * out[NT + i];
        for (int i = 0; i < P; ++i) if (fx[i]) { for (int j = 0; j < P; ++j) { S[i * P + j] = 0.0; S[j * P + i] = 0.0; } S[i * P + i] = 1.0; }
        if (!chol_host(S, P)) { status = CCAL_ERR_NOT_PD; break; }
        chol_solve_host(S, P, dx);
        for (int i = 0; i < P; ++i) if (!fx[i]) th[i] = std::min(std::max(th[i] + dx[i], lo[i]), hi[i]);
        if (eval(th) != 0) return CCAL_ERR_HIP;
        s = out[NT + P];
        cur = huber_w(s, 1.0) * s;
        R.iterations++;
        if (o.verbose) std::printf("[ccal convert_model] iter %d cost %.12g\n", it, cur);
        const double le = o.error_metric ? std::sqrt(std::max(last, 0.0)) : last, ce = o.error_metric ? std::sqrt(std::max(cur, 0.0)) : cur;
        if (ce < o.min_error) break;
        if (std::isnan(cur)) { status = CCAL_ERR_NONFINITE; break; }
        if (std::fabs(le - ce) < o.min_abs_error_decrease) break;
        if (std::fabs(le - ce) / le < o.min_rel_error_decrease) break;
        if (it == o.max_iterations - 1) status = CCAL_ERR_NO_CONVERGENCE;
    }
    R.final_cost = cur; R.status = status;
    if (rep) *rep = R;
    if (status == CCAL_OK || status == CCAL_ERR_NO_CONVERGENCE) for (int i = 0; i < P; ++i) tgt_params_io[canon(tgt_model, i)] = th[i];
    if (status != CCAL_OK && ctx->err.empty()) ctx->err = "ccal_convert_model: solve failed";
    return status;
    CCAL_API_CATCH(ctx)
}
