// Mode N kernels: weighted Gram of [J | r] per observation frame on the f64 matrix cores, per-slot
// Schur complement, deterministic reductions, the small camera-system solve and the pose
// back-substitution.  Replaces tiny-solver's sparse J^T J assembly + sparse Cholesky (call sites
// src/util.rs:455, 670) by the exact arrow-structure elimination described in SURVEY 8(e).
#include <algorithm>

#include "ccal_device.hpp"
#include "ccal_fused.hpp"

namespace ccal {

typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct GramArgs {
    KArgs k;
    const int64_t* goff;
    double* G;
    double* cost_o;
    // device-resident loop: st != NULL evaluates the set the state prescribes (eval_set: the starting point in the first
    // group, the candidate afterwards; index 0 = this set, 1 = the second one); finished solves and re-elimination
    // groups make the launch empty
    const DevState* st;
    const double* intr2; const double* poses2; const double* extr2; double* G2; double* cost_o2;
};

#ifdef CCAL_LEGACY_KERNELS      // matrix-core Gram of the general loop (19-column blocks): libccal_hip_legacy.so only
// ---------------------------------------------------------------------------------------------
// k_gram: one wavefront per observation frame.  Every lane evaluates one corner's two weighted
// rows sqrt(w) [J | r] (NC = D + 1 columns), the wave stages them in LDS as a [rows][16 T] image and
// v_mfma_f64_16x16x4_f64 accumulates G += rows^T rows, 4 rows (2 corners) per instruction: the
// lane that feeds A[i][k] also feeds B[k][j] (same register), so a Gram tile costs one ds_read_b64
// and one MFMA.  T = 1 (NC <= 16: every single-camera model) or 2 (other-camera blocks, NC <= 22).
// ---------------------------------------------------------------------------------------------
template <int MODEL, bool OF, bool OTHER>
__global__ __launch_bounds__(256) void k_gram(const GramArgs ga) {
    const KArgs& a = ga.k;
    if (ga.st && (ga.st->done || ga.st->redo)) return;
    const bool second = ga.st && eval_set(ga.st) == 1;
    const double* p_intr = second ? ga.intr2 : a.intr;
    const double* p_poses = second ? ga.poses2 : a.poses;
    const double* p_extr = second ? ga.extr2 : a.extr;
    double* p_G = second ? ga.G2 : ga.G;
    double* p_cost = second ? ga.cost_o2 : ga.cost_o;
    constexpr int D = block_dim(MODEL, OF, OTHER);
    // Other-camera blocks: d r / d tvec_0_b = (d r / d tvec_c_0) R_c0, so those three columns are linear combinations
    // of three others with per-camera constant coefficients.  When the remaining D - 3 + 1 columns fit ONE 16 x 16
    // matrix-core tile (UCM / EUCM rigs) the Gram is built without them (one MFMA per corner pair instead of
    // three) and k_schur expands it with R_c0, which is stored behind the tile.
    constexpr bool CMP = gram_compact(MODEL, OF, OTHER);
    constexpr int PE = D - (OTHER ? 12 : 6);
    constexpr int NC = CMP ? D - 2 : D + 1;   // staged columns (the last one is the residual)
    constexpr int RC = NC - 1;                // index of the residual column
    constexpr int T = NC <= 16 ? 1 : 2;
    constexpr int RS = T == 1 ? 16 : 24;      // doubles per staged row
    constexpr int CS = 2 * RS + 2;            // doubles per corner (two rows + pad)
    constexpr int NCP = 16 * T;
    constexpr int FCN = OTHER ? FC_SIZE : FC_N0P;
    constexpr int WS = FCN + GRAM_TILE_CORNERS * CS;   // rows are staged 32 corners at a time (LDS -> occupancy)
    extern __shared__ double smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int widx = blockIdx.x * WAVES_PER_BLOCK + wave;
    if (widx >= a.n_list) return;
    double* fc = smem + wave * WS;
    double* tile = fc + FCN;

    const int o = __builtin_amdgcn_readfirstlane(a.list[widx]);
    const int slot = __builtin_amdgcn_readfirstlane(a.obs_slot[o]);
    const int64_t start = a.obs_off[o];
    const int n = (int)(a.obs_off[o + 1] - start);
    const double* th_g = p_intr + a.cam * CCAL_PMAX;
    double th[th_len<MODEL>()];
    load_theta<MODEL, OF>(th_g, a.rt, th);
    {
        double pose[6], ex[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) pose[i] = p_poses[(int64_t)slot * 6 + i];
        if constexpr (OTHER) {
#pragma unroll
            for (int i = 0; i < 6; ++i) ex[i] = p_extr[a.cam * 6 + i];
        }
        double fcr[OTHER ? FC_SIZE : FC_N0];
        frame_setup<OTHER>(pose, ex, fcr);
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < (OTHER ? FC_SIZE : FC_N0); ++i) fc[i] = fcr[i];
        }
    }
    wave_sync_lds();

    d4 acc00a = { 0, 0, 0, 0 }, acc00b = { 0, 0, 0, 0 }, acc01 = { 0, 0, 0, 0 }, acc11 = { 0, 0, 0, 0 };
    // read position of this lane inside a corner pair: corner (lane >> 5), row (lane >> 4) & 1, column lane & 15
    const int rd_off = (lane >> 5) * CS + ((lane >> 4) & 1) * RS + (lane & 15);
    const bool hi_valid = (lane & 15) < (RS - 16);          // T == 2: columns 16..RS-1 exist, the rest are zero

    for (int base = 0; base < n; base += 64) {
        const int c = base + lane;
        const bool valid = c < n;
        const int64_t g = start + (valid ? c : 0);
        const double X = a.x[g], Y = a.y[g], Z = a.z[g], uo = a.u[g], vo = a.v[g];
        double ru, rv, J[2 * D];
        corner_block<MODEL, OF, OTHER>(th, fc, X, Y, Z, uo, vo, ru, rv, J, J + D);
        // Huber corrector: both rows scaled by sqrt(rho'); invalid lanes contribute zero rows
        const double sw = valid ? huber_sqrt_weight(ru * ru + rv * rv, a.huber_delta) : 0.0;
        const int nv = min(64, n - base);
#pragma unroll
        for (int half = 0; half < 64 / GRAM_TILE_CORNERS; ++half) {
            if (half * GRAM_TILE_CORNERS >= nv) break;                    // wave-uniform
            if ((lane / GRAM_TILE_CORNERS) == half) {
                double* row = tile + (lane % GRAM_TILE_CORNERS) * CS;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int i = 0; i < RS; i += 2) {
                        // staged column -> Jacobian column (compact form skips tvec_0_b: PE+3 .. PE+5)
                        const int s0 = (CMP && i >= PE + 3) ? i + 3 : i, s1 = (CMP && i + 1 >= PE + 3) ? i + 4 : i + 1;
                        const double v0 = i < RC ? sw * J[h * D + (i < RC ? s0 : 0)] : (i == RC ? sw * (h ? rv : ru) : 0.0);
                        const double v1 = (i + 1) < RC ? sw * J[h * D + ((i + 1) < RC ? s1 : 0)] : ((i + 1) == RC ? sw * (h ? rv : ru) : 0.0);
                        *reinterpret_cast<double2*>(row + h * RS + i) = make_double2(v0, v1);
                    }
                }
            }
            wave_sync_lds();
            const int npairs = (min(GRAM_TILE_CORNERS, nv - half * GRAM_TILE_CORNERS) + 1) >> 1;
            const double* rd = tile + rd_off;
            if constexpr (T == 1) {
                int m = 0;
                for (; m + 1 < npairs; m += 2) {
                    const double p = rd[(2 * m) * CS];
                    const double q = rd[(2 * m + 2) * CS];
                    acc00a = __builtin_amdgcn_mfma_f64_16x16x4f64(p, p, acc00a, 0, 0, 0);
                    acc00b = __builtin_amdgcn_mfma_f64_16x16x4f64(q, q, acc00b, 0, 0, 0);
                }
                if (m < npairs) {
                    const double p = rd[(2 * m) * CS];
                    acc00a = __builtin_amdgcn_mfma_f64_16x16x4f64(p, p, acc00a, 0, 0, 0);
                }
            } else {
                for (int m = 0; m < npairs; ++m) {
                    const double p = rd[(2 * m) * CS];
                    const double q = hi_valid ? rd[(2 * m) * CS + 16] : 0.0;
                    acc00a = __builtin_amdgcn_mfma_f64_16x16x4f64(p, p, acc00a, 0, 0, 0);
                    acc01 = __builtin_amdgcn_mfma_f64_16x16x4f64(p, q, acc01, 0, 0, 0);
                    acc11 = __builtin_amdgcn_mfma_f64_16x16x4f64(q, q, acc11, 0, 0, 0);
                }
            }
            wave_sync_lds();
        }
    }

    // C/D layout of v_mfma_f64_16x16x4_f64: lane l, register v holds D[(l >> 4) + 4 v][l & 15]
    double* Go = p_G + ga.goff[o];
    const int gi = lane >> 4, gj = lane & 15;
    if constexpr (CMP) {
        // compact tile -> the block-upper-triangular 32-stride layout k_schur reads (rows 0..15 x all columns, rows 16.. x
        // columns 16..), with the tvec_0_b rows / columns restored: column a = sum_m R_c0[m][a] tvec_c_0[m].
        // Three branch-free passes: the tile's own entries straight from the accumulator registers, the 3 x 16 restored
        // rows (+ their mirror), the 3 x 3 restored block.
        const d4 acc = acc00a + acc00b;
        constexpr int TC = PE + 6;                          // first tvec_c_0 column of the compact tile
        wave_sync_lds();
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int ci = gi + 4 * v, cj = gj;
            tile[ci * 17 + cj] = acc[v];
            const int i = ci < PE + 3 ? ci : ci + 3, j = cj < PE + 3 ? cj : cj + 3;     // compact -> virtual column
            if (!(i >= 16 && j < 16)) Go[i * 32 + j] = acc[v];                          // lower-left tile is never read
            if (ci == RC && cj == RC) p_cost[o] = acc[v];
        }
        wave_sync_lds();
        const double* R1 = fc + FC_R1;
        if (lane < 48) {
            const int aa = lane >> 4, cj = lane & 15;
            double val = 0.0;
#pragma unroll
            for (int m = 0; m < 3; ++m) val += R1[m * 3 + aa] * tile[(TC + m) * 17 + cj];
            const int i = PE + 3 + aa, j = cj < PE + 3 ? cj : cj + 3;
            Go[i * 32 + j] = val;
            if (j < 16) Go[j * 32 + i] = val;
        } else if (lane < 57) {
            const int aa = (lane - 48) / 3, ab = (lane - 48) % 3;
            double val = 0.0;
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int n2 = 0; n2 < 3; ++n2) val += R1[m * 3 + aa] * R1[n2 * 3 + ab] * tile[(TC + m) * 17 + TC + n2];
            Go[(PE + 3 + aa) * 32 + PE + 3 + ab] = val;
        }
    } else if constexpr (T == 1) {
        const d4 acc = acc00a + acc00b;
#pragma unroll
        for (int v = 0; v < 4; ++v) Go[(gi + 4 * v) * NCP + gj] = acc[v];
        if (gi + 4 * (RC / 4) == RC && gj == RC) p_cost[o] = acc[RC / 4];
    } else {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            Go[(gi + 4 * v) * NCP + gj] = acc00a[v];
            Go[(gi + 4 * v) * NCP + 16 + gj] = acc01[v];
            Go[(16 + gi + 4 * v) * NCP + 16 + gj] = acc11[v];
        }
        constexpr int dl = RC - 16;  // residual column lives in tile (1,1)
        if (gi + 4 * (dl / 4) == dl && gj == dl) p_cost[o] = acc11[dl / 4];
    }
}

template <int MODEL, bool OF, bool OTHER>
static hipError_t launch_gram_t(const GramArgs& ga, hipStream_t s) {
    constexpr int D = block_dim(MODEL, OF, OTHER);
    constexpr int T = (gram_compact(MODEL, OF, OTHER) ? D - 2 : D + 1) <= 16 ? 1 : 2;
    constexpr int RS = T == 1 ? 16 : 24;
    constexpr int WS = (OTHER ? FC_SIZE : FC_N0P) + GRAM_TILE_CORNERS * (2 * RS + 2);
    const size_t lds = sizeof(double) * WS * WAVES_PER_BLOCK;
    const int blocks = (ga.k.n_list + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    if (blocks == 0) return hipSuccess;
    static DynLdsGuard lds_guard;
    if (hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(&k_gram<MODEL, OF, OTHER>), lds, lds_guard); e != hipSuccess) return e;
    hipLaunchKernelGGL((k_gram<MODEL, OF, OTHER>), dim3(blocks), dim3(256), lds, s, ga);
    return hipGetLastError();
}

#endif  // CCAL_LEGACY_KERNELS

#define CCAL_DISPATCH(FN, model, of, other, ...)                                                     \
    do {                                                                                             \
        const int key_ = (model) * 4 + ((of) ? 2 : 0) + ((other) ? 1 : 0);                           \
        switch (key_) {                                                                              \
            case 0: return FN<kUCM, false, false>(__VA_ARGS__);   case 1: return FN<kUCM, false, true>(__VA_ARGS__);   \
            case 2: return FN<kUCM, true, false>(__VA_ARGS__);    case 3: return FN<kUCM, true, true>(__VA_ARGS__);    \
            case 4: return FN<kEUCM, false, false>(__VA_ARGS__);  case 5: return FN<kEUCM, false, true>(__VA_ARGS__);  \
            case 6: return FN<kEUCM, true, false>(__VA_ARGS__);   case 7: return FN<kEUCM, true, true>(__VA_ARGS__);   \
            case 8: return FN<kKB4, false, false>(__VA_ARGS__);   case 9: return FN<kKB4, false, true>(__VA_ARGS__);   \
            case 10: return FN<kKB4, true, false>(__VA_ARGS__);   case 11: return FN<kKB4, true, true>(__VA_ARGS__);   \
            case 12: return FN<kOCV5, false, false>(__VA_ARGS__); case 13: return FN<kOCV5, false, true>(__VA_ARGS__); \
            case 14: return FN<kOCV5, true, false>(__VA_ARGS__);  case 15: return FN<kOCV5, true, true>(__VA_ARGS__);  \
            default: return hipErrorInvalidValue;                                                    \
        }                                                                                            \
    } while (0)

hipError_t launch_gram(const ccal_problem* p, int cam, bool cand, int gbuf, hipStream_t s) {
#ifndef CCAL_LEGACY_KERNELS
    (void)p; (void)cam; (void)cand; (void)gbuf; (void)s;
    return hipErrorNotSupported;          // the matrix-core pair is not in the product build (normal_ws_ensure never selects it)
#else
    const NormalWs* w = p->nws;
    GramArgs ga = {};
    KArgs& a = ga.k;
    a.x = p->d_x; a.y = p->d_y; a.z = p->d_z; a.u = p->d_u; a.v = p->d_v;
    a.obs_off = p->d_obs_off; a.obs_slot = p->d_obs_slot; a.joff = p->d_joff;
    a.list = p->cams[cam].d_obs; a.n_list = (int32_t)p->cams[cam].obs.size(); a.cam = cam;
    a.intr = cand ? p->d_intr_c : p->d_intr; a.poses = cand ? p->d_poses_c : p->d_poses; a.extr = cand ? p->d_extr_c : p->d_extr;
    a.huber_delta = p->huber_delta; a.rt = model_rt(p->ctx);
    ga.goff = w->d_goff; ga.G = w->G[gbuf]; ga.cost_o = w->cost_o[gbuf];
    CCAL_DISPATCH(launch_gram_t, p->cams[cam].model, p->one_focal, cam > 0, ga, s);
#endif
}
// device-resident loop: set 0 = (p->d_*, G[w->cur]), set 1 = (p->d_*_c, G[w->cur ^ 1])
hipError_t launch_gram_dev_all(const ccal_problem* p, const DevState* st, hipStream_t s) {
    const NormalWs* w = p->nws;
    if (w->merged_gram) return launch_gram_dev(p, -1, st, s);
    for (int c = 0; c < p->n_cams; ++c)
        if (hipError_t e = launch_gram_dev(p, c, st, s); e != hipSuccess) return e;
    return hipSuccess;
}

// ragged frames: the launch's list sorted by corner count and its bins, when normal_ws_ensure_general planned them (ccal_solver.hip)
static void set_gen_bins(FusedArgs& fa, const NormalWs* w, int slot) {
    const GramBins& gb = w->gen_bins[slot];
    if (gb.n_bins <= 0 || !w->d_gen_sorted[slot]) return;
    fa.list = w->d_gen_sorted[slot];
    fa.n_bins = gb.n_bins;
    for (int b = 0; b < kGramMaxBins; ++b) { fa.bin_lpf[b] = gb.lpf[b]; fa.bin_first[b] = gb.first[b]; fa.bin_count[b] = gb.count[b]; fa.bin_wg0[b] = gb.wg0[b]; }
    fa.bin_wg0[kGramMaxBins] = gb.wg0[kGramMaxBins];
}
// cam < 0: the merged launch (w->merged_gram)
hipError_t launch_gram_dev(const ccal_problem* p, int cam, const DevState* st, hipStream_t s) {
    const NormalWs* w = p->nws;
    if (w->register_gram && cam < 0) {
        // Two launches of 2 000 wavefronts each leave the FP64 pipes half empty twice (the second wavefront of every SIMD
        // runs alone for its last ~9 us); one launch refills the slot of a finished wavefront at once
        FusedArgs fa = {};
        fa.x = p->d_x; fa.y = p->d_y; fa.z = p->d_z; fa.u = p->d_u; fa.v = p->d_v;
        fa.obs_off = p->d_obs_off; fa.obs_slot = p->d_obs_slot;
        fa.list = w->d_all_obs; fa.n_obs = p->n_obs; fa.rec_off = w->d_goff; fa.obs_cam = w->d_obs_cam;
        fa.K = p->cams[0].Peff; fa.huber_delta = p->huber_delta; fa.rt = model_rt(p->ctx);
        fa.intr[0] = p->d_intr; fa.intr[1] = p->d_intr_c;
        fa.poses[0] = p->d_poses; fa.poses[1] = p->d_poses_c;
        fa.extr[0] = p->d_extr; fa.extr[1] = p->d_extr_c; fa.cam = 0;
        fa.praw[0] = w->G[w->cur]; fa.praw[1] = w->G[w->cur ^ 1];
        fa.st = st;
        if (st && w->gen_backsub) {          // device-resident loop: the candidate poses are formed in the kernel's prologue
            fa.gen_backsub = 1; fa.g_K = w->K; fa.g_PF = w->PF; fa.g_pf = w->pf; fa.g_dc = w->dc; fa.g_mc_slot = w->mc_slot;
            fa.g_owner = w->d_obs_owner; fa.min_diag = w->lm_min_diag; fa.max_diag = w->lm_max_diag;
        }
        fa.avg_corners = (int32_t)(p->n_corners / std::max(p->n_obs, 1));
        set_gen_bins(fa, w, 0);
        return launch_gram1v_general(p->cams[0].model, p->one_focal, fa, s);
    }
    if (w->register_gram) {
        // every camera's blocks (6 + P_eff + 1 columns at the composed pose: a triangle of <= 136 entries) through the
        // register Gram kernels of the single-camera loop, record format; k_schur expands the records (caminfo NCP = 0).
        // Two EUCM cameras x 10 000 frames: 32 + 60 us (matrix-core kernel for camera 1, 19 columns) -> 2 x ~32 us
        FusedArgs fa = {};
        fa.x = p->d_x; fa.y = p->d_y; fa.z = p->d_z; fa.u = p->d_u; fa.v = p->d_v;
        fa.obs_off = p->d_obs_off; fa.obs_slot = p->d_obs_slot;
        fa.list = p->cams[cam].d_obs; fa.n_obs = (int32_t)p->cams[cam].obs.size(); fa.rec_off = w->d_goff;
        fa.K = p->cams[cam].Peff; fa.huber_delta = p->huber_delta; fa.rt = model_rt(p->ctx);
        fa.intr[0] = p->d_intr + cam * CCAL_PMAX; fa.intr[1] = p->d_intr_c + cam * CCAL_PMAX;
        fa.poses[0] = p->d_poses; fa.poses[1] = p->d_poses_c;
        fa.extr[0] = p->d_extr; fa.extr[1] = p->d_extr_c; fa.cam = cam;
        fa.praw[0] = w->G[w->cur]; fa.praw[1] = w->G[w->cur ^ 1];
        fa.st = st;
        if (st && w->gen_backsub) {          // device-resident loop: the candidate poses are formed in the kernel's prologue
            fa.gen_backsub = 1; fa.g_K = w->K; fa.g_PF = w->PF; fa.g_pf = w->pf; fa.g_dc = w->dc; fa.g_mc_slot = w->mc_slot;
            fa.g_owner = w->d_obs_owner; fa.min_diag = w->lm_min_diag; fa.max_diag = w->lm_max_diag;
        }
        fa.avg_corners = (int32_t)(p->n_corners / std::max(p->n_obs, 1));
        set_gen_bins(fa, w, 1 + cam);
        return launch_gram1v_general(p->cams[cam].model, p->one_focal, fa, s);
    }
#ifndef CCAL_LEGACY_KERNELS
    return hipErrorNotSupported;
#else
    GramArgs ga = {};
    KArgs& a = ga.k;
    a.x = p->d_x; a.y = p->d_y; a.z = p->d_z; a.u = p->d_u; a.v = p->d_v;
    a.obs_off = p->d_obs_off; a.obs_slot = p->d_obs_slot; a.joff = p->d_joff;
    a.list = p->cams[cam].d_obs; a.n_list = (int32_t)p->cams[cam].obs.size(); a.cam = cam;
    a.intr = p->d_intr; a.poses = p->d_poses; a.extr = p->d_extr;
    ga.intr2 = p->d_intr_c; ga.poses2 = p->d_poses_c; ga.extr2 = p->d_extr_c;
    a.huber_delta = p->huber_delta; a.rt = model_rt(p->ctx);
    ga.goff = w->d_goff; ga.G = w->G[w->cur]; ga.cost_o = w->cost_o[w->cur]; ga.G2 = w->G[w->cur ^ 1]; ga.cost_o2 = w->cost_o[w->cur ^ 1];
    ga.st = st;
    CCAL_DISPATCH(launch_gram_t, p->cams[cam].model, p->one_focal, cam > 0, ga, s);
#endif
}

// ---------------------------------------------------------------------------------------------
// k_schur: persistent wavefronts, each walks frame slots with stride (number of waves) and keeps the
// reduced system A[(K+1)^2] + extras in its own LDS accumulators (flushed once at the end).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double clampd(double v, double lo, double hi) { return fmin(fmax(v, lo), hi); }

#ifndef CCAL_SCHUR_MINW
#define CCAL_SCHUR_MINW 4
#endif
// WPB wavefronts per workgroup: 4 (four wavefronts per SIMD: <= 128 VGPRs), or 1 for reduced systems of 64 .. 127 columns,
// whose (K + 1)^2 accumulators (up to 131 KB) leave room for one wavefront per CU only
template <bool REC, int WPB>
__global__ __launch_bounds__(64 * WPB, WPB == 4 ? CCAL_SCHUR_MINW : 1) void k_schur(const SchurArgs a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int K = a.K, K1 = a.K + 1, RB = a.RB;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;     // wave-uniform values in SGPRs
    const int gw = blockIdx.x * WPB + wave;
    // per-wave LDS: acc[RB] | Baug[6][K1] | C[36] | Y[6][K1] | record staging [STG]
    const int WS = ((RB + 12 * K1 + 36 + 1) & ~1) + a.STG;
    double* acc = smem + wave * WS;
    double* Baug = acc + RB;
    double* Cm = Baug + 6 * K1;
    double* Ym = Cm + 36;
    double* stg = acc + ((RB + 12 * K1 + 36 + 1) & ~1);        // 16-byte aligned rows
    // Scatter table, built once per workgroup: for every entry e of a camera's (D+1) x (D+1) Gram block where it is
    // read from (k_gram leaves the whole 16 x 16 tile or the block-upper three tiles of a 32-stride block) and which
    // LDS accumulator(s) receive it - packed src | (dst + 1) << 16 | (extra + 1) << 32.  The per-slot loop is then
    // load, load, ds_add: the index arithmetic and its divergent branches cost as many scalar as vector
    // instructions before (SQ_INSTS_SALU = SQ_INSTS_VALU = 5 k per wavefront).
    __shared__ int32_t cbase[CCAL_MAX_CAMS + 1], cnc2[CCAL_MAX_CAMS], cinfo[CCAL_MAX_CAMS][4];
    if (threadIdx.x < a.n_cams * 4) cinfo[threadIdx.x >> 2][threadIdx.x & 3] = a.caminfo[threadIdx.x];
    int64_t* tab = reinterpret_cast<int64_t*>(smem + WPB * WS);
    constexpr bool TRI = REC && WPB == 4;              // K + 1 <= 64: (i, j) of a lower-triangle entry from a table
    __shared__ uint16_t tri[TRI ? 65 * 66 / 2 : 1];     // lower-triangle entry -> i | j << 8
    if constexpr (TRI) {
        for (int e = threadIdx.x; e < K1 * K1; e += 64 * WPB) {
            const int i = e / K1, j = e - i * K1;
            if (j <= i) tri[i * (i + 1) / 2 + j] = (uint16_t)(i | (j << 8));
        }
    }
    if constexpr (!REC) {
        int base = 0;
        for (int c = 0; c < a.n_cams; ++c) {
            const int Pe = a.caminfo[c * 4 + 0], ct = a.caminfo[c * 4 + 1], ce = a.caminfo[c * 4 + 2], NCP = a.caminfo[c * 4 + 3];
            const int D = Pe + (c > 0 ? 12 : 6), NC = D + 1;
            if (NCP == 0) continue;                 // record format: expanded in the slot loop, no table
            if (threadIdx.x == 0) { cbase[c] = base; cnc2[c] = NC * NC; }
            for (int e = threadIdx.x; e < NC * NC; e += 64 * WPB) {
                const int i = e / NC, j = e - i * NC;
                // local column -> (kind, index): kind 0 = camera-system column (r maps to K), 1 = pose
                int ki, ii, kj, jj;
                if (i < Pe) { ki = 0; ii = ct + i; } else if (i < Pe + 6) { ki = 1; ii = i - Pe; } else if (i < D) { ki = 0; ii = ce + (i - Pe - 6); } else { ki = 0; ii = K; }
                if (j < Pe) { kj = 0; jj = ct + j; } else if (j < Pe + 6) { kj = 1; jj = j - Pe; } else if (j < D) { kj = 0; jj = ce + (j - Pe - 6); } else { kj = 0; jj = K; }
                int off = -1, xoff = -1;
                if (ki == 0) {
                    if (kj == 0) {
                        off = ii * K1 + jj;
                        if (i == j && ii < K) xoff = K1 * K1 + ii;                    // hdiag
                        else if (jj == K && ii < K) xoff = K1 * K1 + K + ii;           // g_c
                        else if (ii == K && jj == K) xoff = K1 * K1 + 2 * K;           // cost = sum rho' s
                    }                                                                  // camera row x pose column: its transpose is taken
                } else if (kj == 0) {
                    off = RB + ii * K1 + jj;                                           // Baug
                } else {
                    off = RB + 6 * K1 + ii * 6 + jj;                                   // Cm
                }
                int src = (i >= 16 && j < 16) ? j * NCP + i : i * NCP + j;
                tab[base + e] = (int64_t)src | ((int64_t)(off + 1) << 16) | ((int64_t)(xoff + 1) << 32);
            }
            base += NC * NC;
        }
    }
    for (int e = lane; e < RB; e += 64) acc[e] = 0.0;
    // record format: the dense 12 x 6 image of E^T sits behind the largest record; its zeros and the three ones of the
    // tvec_c_0 rows are the same for every record - set once; a record's 36 stored values go to their places (et_pos)
    const int EOD = a.STG - 144;
    const int et_pos = gen_et_pos(lane < GEN_EC ? lane : 0);
    if constexpr (REC) {
        for (int e = lane; e < 72; e += 64) { const int row = e / 6, k = e - 6 * row; stg[EOD + e] = (row >= 9 && k - 3 == row - 9) ? 1.0 : 0.0; }
    }
    __syncthreads();
    if (gw >= a.n_pw || (a.st && a.st->done)) return;
    const int g_set = a.st ? schur_set(a.st) : 0;
    const double* p_G = a.Gs[g_set];
    const double lambda = a.st ? schur_lambda(a.st) : a.lambda;

    int o0n = gw < a.n_slots ? a.slot_off[gw] : 0, o1n = gw < a.n_slots ? a.slot_off[gw + 1] : 0;
    for (int s = gw; s < a.n_slots; s += a.n_pw) {
        const int o0 = o0n, o1 = o1n;
        if (s + a.n_pw < a.n_slots) { o0n = a.slot_off[s + a.n_pw]; o1n = a.slot_off[s + a.n_pw + 1]; }   // next slot's range: off the critical path
        double* pf = a.pf + (int64_t)s * a.PF;
        if (lane == 0) acc[RB - 2] += a.mc_slot[s];       // model decrease of this slot's pose block for the step under decision
        if (o0 == o1) {                                   // slot without observations
            for (int e = lane; e < a.PF; e += 64) pf[e] = 0.0;
            continue;
        }
        for (int e = lane; e < 6 * K1 + 36; e += 64) Baug[e] = 0.0;
        wave_sync_lds();
        for (int oi = o0; oi < o1; ++oi) {
            const int64_t desc = a.slot_desc[oi];
            const int cam = (int)(desc & 7);
            const double* Go = p_G + (desc >> 3);
            if constexpr (REC) {
                // Record of the register Gram kernels (GEN, layout in ccal_fused.hpp): the 6 + Pe + 1 column Gram at the
                // COMPOSED pose - C (pose x pose, 6 x 6) | [B|g]^T (K1c rows of 6, r last) | A (K1c x K1c) - and E^T (12 x 6),
                // the matrix that turns the composed pose's (phi, delta) into the block's reference columns
                // rvec_0_b, tvec_0_b | rvec_c_0, tvec_c_0 (frame_setup_composed):
                //   pose x pose, pose x extrinsics, extrinsics x extrinsics = E^T C E;   (camera | r) x those = [B|g]^T E.
                // Every row that meets E is six contiguous doubles (three ds_read_b128), lanes = (row or column group, b)
                // without a division; each accumulator receives one addend per observation frame (ordered sums).
                const int Pe = cinfo[cam][0], ct = cinfo[cam][1], ce = cinfo[cam][2];
                const int K1c = Pe + 1, EO = gen_e_off(Pe), NEc = cam > 0 ? 12 : 6;
                const double* ept = stg + EOD;            // E^T, 12 x 6 (dense image)
                double* wvt = stg + EOD + 72;             // (C E)^T, 12 x 6
                {   // all loads first (one memory latency per record), then the LDS image: C | [B|g]^T | A (packed lower triangle)
                    // as stored, E^T expanded from its 36 stored non-zeros to the dense 12 x 6 the products read
                    constexpr int RV = (gen_e_off(9) + 63) / 64;           // P_eff <= 9 (OPENCV5)
                    double rv[RV], ecv = 0.0;
#pragma unroll
                    for (int t = 0; t < RV; ++t) rv[t] = (lane + 64 * t) < EO ? Go[lane + 64 * t] : 0.0;
                    if (lane < GEN_EC) ecv = Go[EO + lane];
#pragma unroll
                    for (int t = 0; t < RV; ++t) if ((lane + 64 * t) < EO) stg[lane + 64 * t] = rv[t];
                    // the dense image's zeros and the three ones of the tvec_c_0 rows are the same for every record (set once, below)
                    if (lane < GEN_EC) stg[EOD + et_pos] = ecv;
                }
                wave_sync_lds();
                // (row group, column b) of the two products: 4 x 16 lanes for twelve columns, 8 x 8 for camera 0's six
                const int sh = cam > 0 ? 4 : 3, b = lane & ((1 << sh) - 1), rg = lane >> sh, rstep = 64 >> sh;
                const bool b_ok = b < NEc;
                double2 eb0 = { 0, 0 }, eb1 = { 0, 0 }, eb2 = { 0, 0 };
                if (b_ok) { const double2* er = reinterpret_cast<const double2*>(ept + 6 * b); eb0 = er[0]; eb1 = er[1]; eb2 = er[2]; }
                // Only the LOWER triangle of the (K + 1) x (K + 1) system is kept (r = row K: b^T, cost): k_solve and
                // ccal_build_normal read nothing else.  The camera | r block needs no product: straight in (+ hdiag, g_c, cost)
                {
                    const int j = lane & 15;
                    for (int i = lane >> 4; i < K1c; i += 4) {
                        if (j > i) continue;
                        const int ii = i < Pe ? ct + i : K, jj = j < Pe ? ct + j : K;
                        const double v = stg[gen_a_off(Pe) + i * (i + 1) / 2 + j];
                        __hip_atomic_fetch_add(acc + ii * K1 + jj, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        const int x = i == j ? (i < Pe ? K1 * K1 + ii : K1 * K1 + 2 * K) : ((i == Pe) ? K1 * K1 + K + jj : -1);
                        if (x >= 0) __hip_atomic_fetch_add(acc + x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
                // rows 0..5: C E (kept, transposed, for the second product); rows 6..: [B|g]^T E (final: scattered at once)
                for (int row = rg; row < 6 + K1c; row += rstep) {
                    if (!b_ok) continue;
                    const double2* rr = reinterpret_cast<const double2*>(stg + 6 * row);
                    const double2 r0 = rr[0], r1 = rr[1], r2 = rr[2];
                    const double t = ((r0.x * eb0.x + r0.y * eb0.y) + (r1.x * eb1.x + r1.y * eb1.y)) + (r2.x * eb2.x + r2.y * eb2.y);
                    if (row < 6) { wvt[6 * b + row] = t; continue; }
                    const int i = row - 6, ci = i < Pe ? ct + i : K;
                    if (b < 6) {
                        __hip_atomic_fetch_add(Baug + b * K1 + ci, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    } else {
                        const int cj = ce + (b - 6), hi = ci > cj ? ci : cj, lo = ci > cj ? cj : ci;
                        __hip_atomic_fetch_add(acc + hi * K1 + lo, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (i == Pe) __hip_atomic_fetch_add(acc + K1 * K1 + K + cj, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
                wave_sync_lds();
                double2 w0 = { 0, 0 }, w1 = { 0, 0 }, w2 = { 0, 0 };
                if (b_ok) { const double2* wr = reinterpret_cast<const double2*>(wvt + 6 * b); w0 = wr[0]; w1 = wr[1]; w2 = wr[2]; }
                for (int aa = rg; aa < NEc; aa += rstep) {
                    // pose x pose and extrinsics x extrinsics: lower triangle; pose x extrinsics: all of it
                    if (!b_ok || ((aa < 6) == (b < 6) ? b > aa : aa >= 6)) continue;
                    const double2* ar = reinterpret_cast<const double2*>(ept + 6 * aa);
                    const double2 a0 = ar[0], a1 = ar[1], a2 = ar[2];
                    const double t = ((a0.x * w0.x + a0.y * w0.y) + (a1.x * w1.x + a1.y * w1.y)) + (a2.x * w2.x + a2.y * w2.y);
                    if (aa < 6) {
                        double* dst = b < 6 ? Cm + aa * 6 + b : Baug + aa * K1 + ce + (b - 6);
                        __hip_atomic_fetch_add(dst, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    } else {
                        __hip_atomic_fetch_add(acc + (ce + aa - 6) * K1 + ce + (b - 6), t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (aa == b) __hip_atomic_fetch_add(acc + K1 * K1 + ce + aa - 6, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
                wave_sync_lds();
            } else {
            const int64_t* tb = tab + cbase[cam];
            const int nc2 = cnc2[cam];
            // table entries, then all of this lane's Gram entries (NC <= 22: at most 8 per lane; one memory latency per
            // frame), then the adds: LDS atomics (ds_add_f64, nothing waits on them); within one observation frame
            // every accumulator receives exactly one addend, so the sums stay ordered
            constexpr int GV = (22 * 22 + 63) / 64;
            int64_t ent[GV];
            double gv[GV];
#pragma unroll
            for (int t = 0; t < GV; ++t) { const int e = lane + 64 * t; ent[t] = e < nc2 ? tb[e] : 0; }
#pragma unroll
            for (int t = 0; t < GV; ++t) gv[t] = (lane + 64 * t) < nc2 ? Go[ent[t] & 0xffff] : 0.0;
#pragma unroll
            for (int t = 0; t < GV; ++t) {
                const int off = (int)((ent[t] >> 16) & 0xffff) - 1, xoff = (int)((ent[t] >> 32) & 0xffff) - 1;
                if (off >= 0) __hip_atomic_fetch_add(acc + off, gv[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (xoff >= 0) __hip_atomic_fetch_add(acc + xoff, gv[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            wave_sync_lds();
            }
        }
        // Cholesky of C + lambda clamp(diag C): every lane runs the same 6x6 factorisation
        double L[21], dC[6];
        bool ok = true;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            dC[i] = Cm[i * 6 + i];
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                double t = Cm[i * 6 + j];
                if (i == j && lambda > 0.0) t += lambda * clampd(dC[i], a.min_diag, a.max_diag);
#pragma unroll
                for (int k = 0; k < j; ++k) t -= L[i * (i + 1) / 2 + k] * L[j * (j + 1) / 2 + k];
                if (i == j) {
                    ok = ok && (t > 0.0) && (t < 1.7e308);
                    double sq, rsq;
                    fast_sqrt_rsqrt(ok ? t : 1.0, sq, rsq);
                    L[i * (i + 1) / 2 + i] = ok ? rsq : 0.0;                  // diagonal stored inverted
                } else {
                    L[i * (i + 1) / 2 + j] = t * L[j * (j + 1) / 2 + j];
                }
            }
        }
        if (!ok) {
            if (lane == 0) acc[RB - 1] += 1.0;            // failed pose blocks (all-reduced with the sums: every rank sees them)
            for (int e = lane; e < a.PF; e += 64) pf[e] = 0.0;
            wave_sync_lds();
            continue;
        }
        // Y = L^-1 [B | g_p], one column per lane
        for (int j = lane; j < K1; j += 64) {
            double y[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                double t = Baug[i * K1 + j];
#pragma unroll
                for (int k = 0; k < i; ++k) t -= L[i * (i + 1) / 2 + k] * y[k];
                y[i] = t * L[i * (i + 1) / 2 + i];
                Ym[i * K1 + j] = y[i];
                pf[21 + i * K1 + j] = y[i];
            }
        }
        // L and diag C to the slot's record: every lane holds them; lane 0 parks them in the (now dead) C block, 21 + 6
        // lanes store (two ds_write / ds_read pairs instead of 27 three-instruction select chains)
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 21; ++i) Cm[i] = L[i];
#pragma unroll
            for (int i = 0; i < 6; ++i) Cm[21 + i] = dC[i];
        }
        wave_sync_lds();
        if (lane < 21) pf[lane] = Cm[lane];
        if (lane < 6) { pf[21 + 6 * K1 + lane] = Baug[lane * K1 + K]; pf[21 + 6 * K1 + 6 + lane] = Cm[21 + lane]; }
        // A -= Y^T Y  (record mode: lower triangle only, (i, j) from the workgroup's table)
        if constexpr (REC) {
            const int NT = K1 * (K1 + 1) / 2;
            for (int e = lane; e < NT; e += 64) {
                int i, j;
                if constexpr (TRI) { const int ij = tri[e]; i = ij & 255; j = ij >> 8; }
                else {           // large systems: decoded (no room for the table beside the accumulators)
                    i = (int)((__builtin_sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
                    if ((i + 1) * (i + 2) / 2 <= e) ++i;
                    if (i * (i + 1) / 2 > e) --i;
                    j = e - i * (i + 1) / 2;
                }
                double t = 0.0;
#pragma unroll
                for (int k = 0; k < 6; ++k) t += Ym[k * K1 + i] * Ym[k * K1 + j];
                __hip_atomic_fetch_add(acc + i * K1 + j, -t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        } else {
            const float rk1 = 1.0f / (float)K1;
            for (int e = lane; e < K1 * K1; e += 64) {
                const int i = (int)(((float)e + 0.5f) * rk1), j = e - i * K1;
                double t = 0.0;
#pragma unroll
                for (int k = 0; k < 6; ++k) t += Ym[k * K1 + i] * Ym[k * K1 + j];
                __hip_atomic_fetch_add(acc + e, -t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        wave_sync_lds();
    }
    // the four wavefronts of the workgroup combine in LDS (fixed order): a quarter of the partial sums to write and for
    // k_reduce to read (4 096 wavefronts x 367 doubles were 12 MB per group for two EUCM cameras)
    __syncthreads();
    const int nblk = a.n_pw / WPB;
    for (int e = threadIdx.x; e < RB; e += 64 * WPB) {
        if constexpr (WPB == 4) a.partial[(int64_t)e * nblk + blockIdx.x] = (smem[e] + smem[WS + e]) + (smem[2 * WS + e] + smem[3 * WS + e]);
        else a.partial[(int64_t)e * nblk + blockIdx.x] = smem[e];
    }
}

hipError_t launch_schur(const ccal_problem* p, int gbuf, double lambda, double min_diag, double max_diag, hipStream_t s,
                        const DevState* st) {
    const NormalWs* w = p->nws;
    SchurArgs a = {};
    a.st = st; a.Gs[1] = w->G[gbuf ^ 1];
    a.Gs[0] = w->G[gbuf]; a.slot_desc = w->schurq ? w->d_slot_rec : w->d_slot_desc; a.slot_off = w->d_slot_off;
    a.caminfo = w->d_caminfo; a.n_cams = p->n_cams;
    a.n_slots = p->n_slots; a.K = w->K; a.RB = w->RB; a.PF = w->PF; a.n_pw = w->n_pw;
    a.lambda = lambda; a.min_diag = min_diag; a.max_diag = max_diag;
    a.partial = w->partial; a.pf = w->pf; a.mc_slot = w->mc_slot;
    const int K1 = w->K + 1;
    // record-format cameras: staging for the largest record (with E^T) + (C E)^T (72)
    int stg = 0;
    if (w->register_gram)
        for (int c = 0; c < p->n_cams; ++c) stg = std::max(stg, gen_e_off(p->cams[c].Peff) + 144);       // record + dense E^T + (C E)^T
    a.STG = stg;
    const int WS = ((w->RB + 12 * K1 + 36 + 1) & ~1) + stg;
    int tab_entries = 0;
    if (!w->register_gram) for (int c = 0; c < p->n_cams; ++c) tab_entries += (p->cams[c].D + 1) * (p->cams[c].D + 1);
    const size_t lds = sizeof(double) * ((size_t)WS * WAVES_PER_BLOCK + tab_entries);
    if (w->schurq) return launch_schurq(a, p->cams[0].Peff, w->n_rows, w->schurq_slots, s);
    if (w->schur_wpb == 1) {           // 64 .. 127 columns: one wavefront per workgroup (record format only: normal_ws_ensure)
        const size_t lds1 = sizeof(double) * (size_t)WS;
        static DynLdsGuard lds_guard_big;
        if (hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(&k_schur<true, 1>), lds1, lds_guard_big); e != hipSuccess) return e;
        hipLaunchKernelGGL((k_schur<true, 1>), dim3(w->n_pw), dim3(64), lds1, s, a);
        return hipGetLastError();
    }
    static DynLdsGuard lds_guard;
    const int blocks = (w->n_pw + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    if (w->register_gram) {
        static DynLdsGuard lds_guard_rec;
        if (hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(&k_schur<true, 4>), lds, lds_guard_rec); e != hipSuccess) return e;
        hipLaunchKernelGGL((k_schur<true, 4>), dim3(blocks), dim3(256), lds, s, a);
        return hipGetLastError();
    }
#ifdef CCAL_LEGACY_KERNELS          // the table-driven elimination of the matrix-core Gram's 16 x 16 / 32-stride tiles
    if (hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(&k_schur<false, 4>), lds, lds_guard); e != hipSuccess) return e;
    hipLaunchKernelGGL((k_schur<false, 4>), dim3(blocks), dim3(256), lds, s, a);
    return hipGetLastError();
#else
    (void)lds_guard;
    return hipErrorNotSupported;
#endif
}

// ---------------------------------------------------------------------------------------------
// k_reduce: red[e] = sum_w partial[e][w], one workgroup per element, fixed summation order.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double block_sum_256(double v, double* sh) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void k_reduce(const double* partial, int n_pw, double* red, const DevState* st) {
    __shared__ double sh[4];
    if (st && st->done) return;
    const double* src = partial + (int64_t)blockIdx.x * n_pw;
    double v = 0.0;
    for (int i = threadIdx.x; i < n_pw; i += 256) v += src[i];
    const double t = block_sum_256(v, sh);
    if (threadIdx.x == 0) red[blockIdx.x] = t;
}
hipError_t launch_reduce(const ccal_problem* p, hipStream_t s, const DevState* st) {
    const NormalWs* w = p->nws;
    hipLaunchKernelGGL(k_reduce, dim3(w->RB), dim3(256), 0, s, w->partial, w->n_rows, w->red_out ? w->red_out : w->red, st);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// k_solve: one wavefront.  Device-resident loop (st != NULL): first the optimizer's decision on the all-reduced sums
// (optimizer_decide, ccal_fused.hpp - the same function as the single-camera loop), then, when the sums at hand are
// the system to solve:  S = A[0:K,0:K] + lambda clamp(hdiag), rhs = -b, fixed columns -> identity, in-LDS Cholesky,
// dc, candidate intrinsics/extrinsics = clamp(x + dc), model decrease of the camera block; status word for the host.
// ---------------------------------------------------------------------------------------------
struct SolveArgs {
    const double* red; const ColInfo* cols; int32_t K, RB;
    double lambda, min_diag, max_diag;
    const double* intr; const double* extr; double* intr_c; double* extr_c; int32_t n_intr, n_extr;
    double* dc; double* scal; int32_t* flags;
    DevState* st;                  // device-resident loop: current set = st->cur (0: intr/extr, 1: intr_c/extr_c), lambda from the state
    HostStatus* hs; int32_t seq, publish_all;
    PeerView peers;                // in-process transport: every rank's sums of this step (rank order), added here; n == 0: `red`
};
// BIG: reduced systems of 64 .. 127 columns - two wavefronts (thread i still owns row i), the matrix in dynamic LDS with a
// run-time row stride, pivots and finished components travel through LDS instead of wavefront shuffles
// the value of lane j (wave-uniform j) in every lane: two v_readlane_b32 through scalar registers - a ds_bpermute (what __shfl
// compiles to) is an LDS-pipe round trip, and the factorisation waits for one per column
__device__ __forceinline__ double bcast_lane(double v, int j) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), j), hi = __builtin_amdgcn_readlane(__double2hiint(v), j);
    return __hiloint2double(hi, lo);
}
// The factorisation of k_solve for K <= KT with the rows in REGISTERS: lane i holds row i of the lower triangle (lane K: the
// right-hand side riding along), column j is t = S[i][j] - sum_{k<j} L[i][k] L[j][k] with L[j][k] read from lane j's registers
// (v_readlane: j and k are compile-time), the same operations in the same order as the LDS loop below - the same bits - without
// its LDS round trips and its barrier per column (K = 18: ~6 us -> ~1.5).  L goes back to LDS for the back substitution.
// Returns false when a pivot is not positive (the factor is then incomplete, as in the LDS loop).
template <int KT>
__device__ __forceinline__ bool chol_rows_reg(double* S, const int LD, const int K, const int lane) {
    double r[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) r[k] = (k < K && lane <= K) ? S[lane * LD + k] : 0.0;
    bool ok = true;
#pragma unroll
    for (int j = 0; j < KT; ++j) {
        if (j < K && ok) {                         // wave-uniform
            // lane j's row first, every entry into scalar registers of its own (one pair reused for all of them makes each FMA wait
            // for "its" v_readlane: ~65 cycles per entry measured), then the dependent chain
            double lj[KT];
#pragma unroll
            for (int k = 0; k < j; ++k) lj[k] = bcast_lane(r[k], j);
            double t = r[j];
#pragma unroll
            for (int k = 0; k < j; ++k) t -= r[k] * lj[k];
            const double piv = bcast_lane(t, j);
            if (!(piv > 0.0) || !(piv < 1.7e308)) ok = false;
            else {
                double sq, inv;
                fast_sqrt_rsqrt(piv, sq, inv);
                r[j] = lane == j ? inv : t * inv;  // (lanes above the diagonal: never written back)
            }
        }
    }
#pragma unroll
    for (int k = 0; k < KT; ++k) if (k < K && lane <= K && (k <= lane)) S[lane * LD + k] = r[k];
    return ok;
}
template <bool BIG>
__global__ __launch_bounds__(BIG ? 128 : 64) void k_solve(const SolveArgs a0) {
    constexpr int NTH = BIG ? 128 : 64;
    SolveArgs a = a0;
    __shared__ int go;
    __shared__ DevState S0;            // the optimizer state is staged in LDS: the decision touches ~20 fields one after the
                                       // other, each a global round trip for a single lane otherwise (k_solve 19 -> ~9 us)
    const int lane = threadIdx.x;
    DevState* const gst = a0.st;
    // Everything the kernel reads from global memory is requested HERE, in one go - the state, the three sums of the decision, the
    // column table, this lane's entries of the system (up to PRE_S of them: K <= 24 with one wavefront), right-hand side, diagonal
    // and gradient, and the parameter value of this lane's column in BOTH sets (the decision picks the current one): the kernel
    // was a chain of three dependent round trips - state and sums, then columns and vectors, then the matrix (session-sized rigs:
    // k_solve 15.6 us of a 50-us group)
    constexpr int PRE_S = 9;
    const int Kp = a.K, K1p = Kp + 1;
    // entry idx of the (all-)reduced sums: the all-reduce buffer, or - in-process transport - the ranks' buffers added in rank order
    auto R = [&a0](const int idx) -> double { return a0.peers.n > 0 ? peer_sum(a0.peers, idx) : a0.red[idx]; };
    const bool pre = !BIG && Kp * Kp <= PRE_S * NTH;
    ColInfo ci = {};
    if (lane < Kp) ci = a.cols[lane];
    double sv[PRE_S];
    if (pre) {
        const float rk = 1.0f / (float)Kp;
#pragma unroll
        for (int q = 0; q < PRE_S; ++q) {
            const int e = lane + NTH * q;
            const int ec = e < Kp * Kp ? e : 0;
            const int i = (int)(((float)ec + 0.5f) * rk), j = ec - i * Kp;
            sv[q] = R(i >= j ? i * K1p + j : j * K1p + i);
        }
    }
    const double rhs = lane < Kp ? R(Kp * K1p + lane) : 0.0;          // the system is kept as its lower triangle (row K = b^T)
    const double hd = lane < Kp ? R(K1p * K1p + lane) : 0.0, gcl = lane < Kp ? R(K1p * K1p + Kp + lane) : 0.0;
    const double xs0 = lane < Kp ? (ci.is_extr ? a0.extr : a0.intr)[ci.dst] : 0.0;
    const double xs1 = (lane < Kp && a0.st) ? (ci.is_extr ? a0.extr_c : a0.intr_c)[ci.dst] : 0.0;
    // the candidate set starts as a copy of the current one: both sets' values of this lane's element, requested now as well
    const bool pre_c = a.n_intr <= NTH && a.n_extr <= NTH;
    double vi0 = 0.0, vi1 = 0.0, ve0 = 0.0, ve1 = 0.0;
    if (pre_c) {
        if (lane < a.n_intr) { vi0 = a0.intr[lane]; if (a0.st) vi1 = a0.intr_c[lane]; }
        if (lane < a.n_extr) { ve0 = a0.extr[lane]; if (a0.st) ve1 = a0.extr_c[lane]; }
    }
    if (a.st) {
        // the three sums the decision needs are requested together with the state (one memory latency, not two)
        const double d_cost = R(a.RB - 3), d_mc = R(a.RB - 2), d_fail = R(a.RB - 1);
        {
            const double* src = reinterpret_cast<const double*>(gst);
            double* dst = reinterpret_cast<double*>(&S0);
            for (int e = lane; e < (int)(sizeof(DevState) / sizeof(double)); e += NTH) dst[e] = src[e];
        }
        __syncthreads();
        a.st = &S0;
        if (S0.done) { if (lane == 0) publish_host_status(a.hs, &S0, a.seq, a.publish_all != 0); return; }
        {   // the decision on a register copy of the state (every lane the same work; through LDS each of its ~40 dependent field
            // accesses was an LDS round trip), written back by lane 0
            DevState loc = S0;
            const bool sv_go = optimizer_decide(&loc, d_cost, d_mc, d_fail > 0.0, a.seq);
            __syncthreads();
            if (lane == 0) { S0 = loc; go = sv_go ? 1 : 0; }
        }
        __syncthreads();
        if (!go) {
            const double* src = reinterpret_cast<const double*>(&S0);
            double* dst = reinterpret_cast<double*>(gst);
            for (int e = lane; e < (int)(sizeof(DevState) / sizeof(double)); e += NTH) dst[e] = src[e];
            __syncthreads();
            if (lane == 0) publish_host_status(a.hs, &S0, a.seq, a.publish_all != 0);
            return;
        }
        a.lambda = S0.lambda;
        if (S0.cur) { a.intr = a0.intr_c; a.extr = a0.extr_c; a.intr_c = const_cast<double*>(a0.intr); a.extr_c = const_cast<double*>(a0.extr); }
    }
    constexpr int KS = 64;                                          // the one-wavefront form: K + 1 <= 64 rows
    __shared__ double S_small[BIG ? 1 : (KS + 1) * (KS + 1)];       // rows 0..K-1: the system, row K: the right-hand side
    extern __shared__ __attribute__((aligned(16))) double S_big[];
    __shared__ double x[CCAL_KMAX];
    __shared__ double pivs;
    __shared__ double mcs[2];
    __shared__ int bad;
    const int K = a.K, K1 = K + 1;
    double* const S = BIG ? S_big : S_small;
    const int LD = BIG ? (K1 | 1) : KS + 1;                          // odd row stride: a column walk touches every bank
    __shared__ int fxs[CCAL_KMAX];
    if (lane < K) fxs[lane] = ci.fixed;
    if (lane == 0) bad = 0;
    if (pre_c) {
        const bool c1 = a0.st && S0.cur;
        if (lane < a.n_intr) a.intr_c[lane] = c1 ? vi1 : vi0;
        if (lane < a.n_extr) a.extr_c[lane] = c1 ? ve1 : ve0;
    } else {
        for (int e = lane; e < a.n_intr; e += NTH) a.intr_c[e] = a.intr[e];
        for (int e = lane; e < a.n_extr; e += NTH) a.extr_c[e] = a.extr[e];
    }
    const double xsrc = (a0.st && S0.cur) ? xs1 : xs0;                  // (a.intr / a.extr are the current set: swapped above when cur == 1)
    __syncthreads();
    {
        const float rk = 1.0f / (float)K;
        if (pre) {
#pragma unroll
            for (int q = 0; q < PRE_S; ++q) {
                const int e = lane + NTH * q;
                if (e < K * K) {
                    const int i = (int)(((float)e + 0.5f) * rk), j = e - i * K;
                    double v = sv[q];
                    if (fxs[i] || fxs[j]) v = (i == j) ? 1.0 : 0.0;
                    S[i * LD + j] = v;
                }
            }
        } else {
#pragma unroll 4
            for (int e = lane; e < K * K; e += NTH) {
                const int i = (int)(((float)e + 0.5f) * rk), j = e - i * K;
                double v = R(i >= j ? i * K1 + j : j * K1 + i);
                if (fxs[i] || fxs[j]) v = (i == j) ? 1.0 : 0.0;
                S[i * LD + j] = v;
            }
        }
    }
    __syncthreads();
    if (lane < K && !ci.fixed && a.lambda > 0.0) S[lane * LD + lane] += a.lambda * clampd(hd, a.min_diag, a.max_diag);
    // the right-hand side rides along as row K of the matrix: the factorisation's own recurrence turns it into L^-1 rhs
    // (the forward substitution costs nothing extra: lane K is one more row)
    if (lane < K) S[K * LD + lane] = ci.fixed ? 0.0 : -rhs;
    __syncthreads();
    // left-looking Cholesky, lane i owns row i: t = S[i][j] - sum_{k<j} L[i][k] L[j][k] (no stores inside the sum,
    // so the LDS reads pipeline), the pivot travels by shuffle; one barrier per column.  Same operation order as
    // the right-looking form.
    constexpr int KT_REG = 32;
    bool in_regs = false;
    if constexpr (!BIG) {
        if (K <= KT_REG) {
            in_regs = true;
            if (!chol_rows_reg<KT_REG>(S, LD, K, lane) && lane == 0) bad = 1;
        }
    }
    for (int j = 0; j < (in_regs ? 0 : K); ++j) {
        double t = 0.0;
        if (lane >= j && lane <= K) {
            t = S[lane * LD + j];
#pragma unroll 4
            for (int k = 0; k < j; ++k) t -= S[lane * LD + k] * S[j * LD + k];
        }
        double piv;
        if constexpr (BIG) { if (lane == j) pivs = t; __syncthreads(); piv = pivs; }
        else piv = bcast_lane(t, j);
        if (!(piv > 0.0) || !(piv < 1.7e308)) { if (lane == 0) bad = 1; break; }      // uniform
        double sq, inv;
        fast_sqrt_rsqrt(piv, sq, inv);                    // hardware seed + Newton steps (<= 2 ulp), not the IEEE sqrt + division expansions
        if (lane == j) S[j * LD + j] = inv;               // the diagonal is kept inverted: the solves below multiply
        else if (lane > j && lane <= K) S[lane * LD + j] = t * inv;
        __syncthreads();
    }
    __syncthreads();
    if (bad) {
        if (lane == 0) {
            a.flags[1] = 1; a.scal[2] = 0.0;
            if (a.st) {
                // Gauss-Newton has no step (tiny-solver: None); LM rejects the (unchanged) candidate at the next decision
                if (a.st->method != CCAL_METHOD_LM) { a.st->done = CCAL_ERR_NOT_PD + 1; if (!a.st->done_seq) a.st->done_seq = a.seq; }
                else a.st->cam_failed = 1;
                a.st->mc_cam = 0.0; a.st->lambda_solve = a.lambda;
            }
        }
        if (lane < K) a.dc[lane] = 0.0;
        __syncthreads();
        if (a.st) {
            const double* src = reinterpret_cast<const double*>(&S0);
            double* dst = reinterpret_cast<double*>(gst);
            for (int e = lane; e < (int)(sizeof(DevState) / sizeof(double)); e += NTH) dst[e] = src[e];
            __syncthreads();
            if (lane == 0) publish_host_status(a.hs, &S0, a.seq, a.publish_all != 0);
        }
        return;
    }
    {   // back substitution on y = L^-1 rhs (row K): lane i owns x_i, the finished component travels by shuffle
        double xi = lane < K ? S[K * LD + lane] : 0.0;
        for (int j = K - 1; j >= 0; --j) {
            if (lane == j) xi = xi * S[j * LD + j];
            double xj;
            if constexpr (BIG) { if (lane == j) pivs = xi; __syncthreads(); xj = pivs; __syncthreads(); }
            else xj = bcast_lane(xi, j);
            if (lane < j) xi -= S[j * LD + lane] * xj;
        }
        if (lane < K) x[lane] = xi;
    }
    __syncthreads();
    double mc = 0.0;
    if (lane < K) {
        const double d = x[lane];
        a.dc[lane] = d;
        const double Dii = a.lambda > 0.0 ? a.lambda * clampd(hd, a.min_diag, a.max_diag) : 0.0;
        if (!ci.fixed) {
            mc = d * (Dii * d - gcl);
            double* dst = ci.is_extr ? a.extr_c : a.intr_c;
            double v = xsrc + d;
            if (ci.has_bound) v = fmin(fmax(v, ci.lo), ci.hi);     // tiny-solver: max(lo).min(hi)
            dst[ci.dst] = v;
            if (ci.dst2 >= 0) dst[ci.dst2] = v;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mc += __shfl_down(mc, off, 64);
    if constexpr (BIG) {
        if ((lane & 63) == 0) mcs[lane >> 6] = mc;
        __syncthreads();
        mc = mcs[0] + mcs[1];
    }
    if (lane == 0) {
        a.scal[2] = mc;
        if (a.st) { a.st->mc_cam = mc; a.st->lambda_solve = a.lambda; }
    }
    __syncthreads();
    if (a.st) {
        const double* src = reinterpret_cast<const double*>(&S0);
        double* dst = reinterpret_cast<double*>(gst);
        for (int e = lane; e < (int)(sizeof(DevState) / sizeof(double)); e += NTH) dst[e] = src[e];
        __syncthreads();
        if (lane == 0) publish_host_status(a.hs, &S0, a.seq, a.publish_all != 0);
    }
}
hipError_t launch_solve(const ccal_problem* p, double lambda, double min_diag, double max_diag, hipStream_t s, DevState* st,
                        HostStatus* hs, int seq, bool publish_all) {
    const NormalWs* w = p->nws;
    SolveArgs a = {};
    a.st = st; a.hs = hs; a.seq = seq; a.publish_all = publish_all ? 1 : 0;
    a.red = w->red; a.cols = w->cols; a.K = w->K; a.RB = w->RB; a.lambda = lambda; a.min_diag = min_diag; a.max_diag = max_diag;
    a.peers = w->peers;
    a.intr = p->d_intr; a.extr = p->d_extr; a.intr_c = p->d_intr_c; a.extr_c = p->d_extr_c;
    a.n_intr = p->n_cams * CCAL_PMAX; a.n_extr = p->n_cams * 6;
    a.dc = w->dc; a.scal = w->scal; a.flags = w->flags;
    if (w->K >= 64) {
        const int K1 = w->K + 1;
        const size_t lds = sizeof(double) * (size_t)(K1 | 1) * K1;
        static DynLdsGuard guard;
        if (hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(&k_solve<true>), lds, guard); e != hipSuccess) return e;
        hipLaunchKernelGGL(k_solve<true>, dim3(1), dim3(128), lds, s, a);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_solve<false>, dim3(1), dim3(64), 0, s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// k_backsub: 16 lanes per frame slot (16 slots per workgroup).  dp = -L^-T (y_r + Y dc); poses_c = poses + dp;
// mc_slot = dp^T (lambda D_p dp - g_p).  The slot's record is staged in LDS with coalesced loads (one thread per slot
// walking its 148-double record alone cost 13 us at 10 000 slots), six lanes form the rows of y_r + Y dc, one lane
// back-substitutes - the same operation order as before.
// ---------------------------------------------------------------------------------------------
struct BacksubArgs {
    const double* pf; const double* dc; double* mc_slot;
    double* poses_s[2];            // parameter set 0 / 1; read [cur], write [cur ^ 1] - indexed, never selected (see SchurArgs::Gs)
    int32_t n_slots, K, PF; double lambda, min_diag, max_diag;
    const DevState* st;
};
__global__ __launch_bounds__(256) void k_backsub(const BacksubArgs a0) {
    // every fused multiply-add spelled out and the step kept out of the pose update's addition by an optimisation barrier, here and
    // in gen_backsub_pose (ccal_gram_common.hpp: the same work in the GEN Gram kernels' prologue for session-sized rigs): the two
    // forms give the same bits
    __shared__ double dcs[CCAL_KMAX];
    extern __shared__ double smem[];                       // [16][PF + 6]
    BacksubArgs a = a0;
    if (a.st) {
        if (a.st->done || a.st->redo) return;      // finished, or this group's decision asked for a re-elimination (nothing was solved)
        a.lambda = a.st->lambda;
    }
    const int cur_set = a.st ? a.st->cur : 0;
    const double* const poses = a0.poses_s[cur_set];          // the kernel argument itself: one scalar load at a computed offset
    double* const poses_c = a0.poses_s[cur_set ^ 1];
    if (threadIdx.x < a.K) dcs[threadIdx.x] = a.dc[threadIdx.x];
    const int g = threadIdx.x >> 4, gl = threadIdx.x & 15;
    const int s = blockIdx.x * 16 + g;
    const bool active = s < a.n_slots;
    const int K1 = a.K + 1, RS = a.PF + 6;
    double* R = smem + g * RS;
    if (active) {
        const double* pf = a.pf + (int64_t)s * a.PF;
        for (int e = gl; e < a.PF; e += 16) R[e] = pf[e];
    }
    __syncthreads();
    if (active && gl < 6) {
        const double* yr = R + 21 + gl * K1;
        double t = yr[a.K];
        for (int j = 0; j < a.K; ++j) t = __builtin_fma(yr[j], dcs[j], t);
        R[a.PF + gl] = -t;
    }
    __syncthreads();
    if (!active || gl != 0) return;
    if (R[0] == 0.0) {          // no observations / failed factorisation: pose unchanged
#pragma unroll
        for (int i = 0; i < 6; ++i) poses_c[(int64_t)s * 6 + i] = poses[(int64_t)s * 6 + i];
        a.mc_slot[s] = 0.0;
        return;
    }
    double L[21], dp[6];
#pragma unroll
    for (int i = 0; i < 21; ++i) L[i] = R[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) dp[i] = R[a.PF + i];
#pragma unroll
    for (int i = 5; i >= 0; --i) {     // L^T x = rhs, diagonal stored inverted
        double t = dp[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k) t = __builtin_fma(-L[k * (k + 1) / 2 + i], dp[k], t);
        dp[i] = t * L[i * (i + 1) / 2 + i];
        asm volatile("" : "+v"(dp[i]));            // the product is a value of its own
    }
    double mc = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const double gp = R[21 + 6 * K1 + i], dC = R[21 + 6 * K1 + 6 + i];
        const double Dii = a.lambda > 0.0 ? a.lambda * clampd(dC, a.min_diag, a.max_diag) : 0.0;
        mc = __builtin_fma(dp[i], __builtin_fma(Dii, dp[i], -gp), mc);
        poses_c[(int64_t)s * 6 + i] = poses[(int64_t)s * 6 + i] + dp[i];
    }
    a.mc_slot[s] = mc;
}
hipError_t launch_backsub(const ccal_problem* p, double lambda, double min_diag, double max_diag, hipStream_t s, const DevState* st) {
    const NormalWs* w = p->nws;
    if (p->n_slots == 0) return hipSuccess;
    BacksubArgs a = {};
    a.st = st;
    a.pf = w->pf; a.dc = w->dc; a.poses_s[0] = p->d_poses; a.poses_s[1] = p->d_poses_c; a.mc_slot = w->mc_slot;
    a.n_slots = p->n_slots; a.K = w->K; a.PF = w->PF; a.lambda = lambda;
    a.min_diag = min_diag; a.max_diag = max_diag;
    const size_t lds = sizeof(double) * 16 * (size_t)(w->PF + 6);
    static DynLdsGuard lds_guard;
    if (hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(&k_backsub), lds, lds_guard); e != hipSuccess) return e;
    hipLaunchKernelGGL(k_backsub, dim3((p->n_slots + 15) / 16), dim3(256), lds, s, a);
    return hipGetLastError();
}

}  // namespace ccal
