// Single-camera fast path of the optimizer: the whole Gauss-Newton / LM iteration is device-resident,
// the host only enqueues kernels and watches a status word in pinned memory.
//
//   k_unpack1       the starting point of a solve (state, column table, intrinsics in the argument block; poses from pinned host memory)
//   k_gram1v        LPF lanes per frame, the whole upper triangle of [J | r]^T W [J | r] in registers, the frame's pose block
//                   eliminated in the kernel's tail (gram_fused_tail): one row of partial sums per wavefront.  UCM / EUCM / KB4
//                   below 2 000 frames (k_gram2, ccal_kernels_gram2.hip, from there on and for OPENCV5)
//   k_gram1v<ITER>  the same with the PREVIOUS launch's rows summed, the decision taken and the camera system solved in front of
//                   the evaluation: a whole optimizer step in one launch (session sizes)
//   k_gram1v_batch  that launch for a whole batch of problems (ccal_solve_batch: blockIdx.y = problem, lockstep)
//   k_reduce1       fixed-order sum over the rows -> red (the all-reduce buffer of sharded solves)
//   k_head          one wavefront: accept / reject / convergence tests (tiny-solver's rules or the Ceres-style trust region),
//                   K x K solve, candidate intrinsics, status to pinned host memory; in-process transport: adds the ranks' sums
//   second library only (-DCCAL_LEGACY_KERNELS / -DCCAL_DEV_SWITCHES): k_gram1 (matrix core), k_gram1w (LDS accumulators),
//                   k_schur1m (the separate elimination launch), the OPENCV5 instantiations of k_gram1v
// Same arithmetic and decision sequence as the general loop in ccal_solver.hip (multi-camera problems);
// parity tests cover both.
#include <algorithm>
#include <cstdlib>

#include "ccal_device.hpp"
#include "ccal_fused.hpp"
#include "ccal_gram_common.hpp"
#include "ccal_head.hpp"
#include "ccal_devopt.hpp"

namespace ccal {

// ---------------------------------------------------------------------------------------------
// k_unpack1: the starting point of a solve in ONE launch.  Optimizer state, column table and intrinsics ride in the kernel's
// argument block (round 4; they were a staging copy the kernel then waited for), the poses are read where the caller left them:
// pinned host memory for session-sized problems (zero-copy: no H2D operation at all), a staged device copy for large ones.
// Both parameter sets start from the caller's values; slots without observations never change.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_unpack1(const UnpackArgs a) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x, nt = (int64_t)gridDim.x * 256;
    // ccal_solve_dev: the starting point already sits in set 0 on the device - only mirror it into set 1
    const double* poses = a.poses_on_device ? a.poses0 : a.poses_src;
    for (int64_t e = t; e < a.np6; e += nt) { const double v = poses[e]; if (!a.poses_on_device) a.poses0[e] = v; a.poses1[e] = v; }
    if (blockIdx.x == 0) {
        constexpr int NS = (int)(sizeof(DevState) / sizeof(double)), NC1 = (int)(sizeof(ColInfo) / sizeof(double));
        for (int e = threadIdx.x; e < CCAL_PMAX; e += 256) {
            const double v = a.poses_on_device ? a.intr0[e] : a.intr_h[e];
            if (!a.poses_on_device) a.intr0[e] = v;
            a.intr1[e] = v;
        }
        for (int e = threadIdx.x; e < NS; e += 256) reinterpret_cast<double*>(a.st)[e] = reinterpret_cast<const double*>(&a.st0)[e];
        for (int e = threadIdx.x; e < a.n_cols * NC1; e += 256) reinterpret_cast<double*>(a.cols)[e] = reinterpret_cast<const double*>(a.col0)[e];
    }
}
hipError_t launch_unpack1(const UnpackArgs& a, hipStream_t s) {
    const int blocks = (int)std::min<int64_t>(std::max<int64_t>((a.np6 + 255) / 256, 1), 1024);
    hipLaunchKernelGGL(k_unpack1, dim3(blocks), dim3(256), 0, s, a);
    return hipGetLastError();
}

#ifdef CCAL_LEGACY_KERNELS      // the matrix-core Gram kernel of round 1: superseded by the register kernels for every model; kept in
                                // libccal_hip_legacy.so as the independent second implementation the parity tests hold the product kernels against
// ---------------------------------------------------------------------------------------------
// k_gram1
// ---------------------------------------------------------------------------------------------
template <int MODEL, bool OF>
__global__ __launch_bounds__(256, CCAL_GRAM_MINW) void k_gram1(const FusedArgs a) {
    constexpr int D = block_dim(MODEL, OF, false);
    constexpr int K = D - 6, K1 = K + 1;
    constexpr int RS = 16, CS = 2 * RS + 2;
    constexpr int WS = FC_N0P + GRAM_TILE_CORNERS * CS;
    extern __shared__ double smem[];
    const DevState* st = a.st;
    if (st->done || st->redo) return;            // finished, or a re-elimination group (no evaluation)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int f = blockIdx.x * WAVES_PER_BLOCK + wave;
    if (f >= a.n_obs) return;
    double* fc = smem + wave * WS;
    double* tile = fc + FC_N0P;
    const int cur = st->cur, first = st->first;
    const int es = first ? cur : (cur ^ 1);
    const double* th_g = a.intr[es];
    double th[th_len<MODEL>()];
    load_theta<MODEL, OF>(th_g, a.rt, th);
    const int64_t start = a.obs_off[f];
    const int n = (int)(a.obs_off[f + 1] - start);
    // software prefetch: the first pass's corner rows are requested before the (long, latency-bound)
    // back-substitution + exp-map below; every later pass is requested one pass ahead
    float pX, pY, pZ, pU, pV;
    {
        const int64_t g0 = start + (lane < n ? lane : 0);
        pX = a.x[g0]; pY = a.y[g0]; pZ = a.z[g0]; pU = a.u[g0]; pV = a.v[g0];
    }
    {
        // candidate pose = accepted pose + back-substitution of the previous camera solve
        // (dp = -L^-T (y_r + Y dc)), then the frame constants; wave-uniform, scalar loads
        const int slot = __builtin_amdgcn_readfirstlane(a.obs_slot[f]);
        double pose[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) pose[i] = a.poses[cur][(int64_t)slot * 6 + i];
        double mc = 0.0;
        if (!first) {
            const double* pf = a.pf[cur] + (int64_t)slot * a.PF;
            if (pf[0] != 0.0) {
                double dp[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const double* yr = pf + 21 + i * K1;
                    double t = yr[K];
#pragma unroll
                    for (int j = 0; j < K; ++j) t += yr[j] * a.dc[j];
                    dp[i] = -t;
                }
#pragma unroll
                for (int i = 5; i >= 0; --i) {
                    double t = dp[i];
#pragma unroll
                    for (int k = i + 1; k < 6; ++k) t -= pf[k * (k + 1) / 2 + i] * dp[k];
                    dp[i] = t * pf[i * (i + 1) / 2 + i];
                }
                const double lam = st->lambda_solve;
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const double gp = pf[21 + 6 * K1 + i], dCi = pf[21 + 6 * K1 + 6 + i];
                    const double Dii = lam > 0.0 ? lam * clampd1(dCi, a.min_diag, a.max_diag) : 0.0;
                    mc += dp[i] * (Dii * dp[i] - gp);
                    pose[i] += dp[i];
                }
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) if (lane == i) a.poses[es][(int64_t)slot * 6 + i] = pose[i];
        }
        if (lane == 0) a.mc_f[f] = mc;
        double fcr[FC_N0];
        frame_setup<false>(pose, nullptr, fcr);
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < FC_N0; ++i) fc[i] = fcr[i];
        }
    }
    wsync();

    d4 acc0 = { 0, 0, 0, 0 }, acc1 = { 0, 0, 0, 0 };
    const int rd_off = (lane >> 5) * CS + ((lane >> 4) & 1) * RS + (lane & 15);
    for (int base = 0; base < n; base += 64) {
        const int c = base + lane;
        const bool valid = c < n;
        const double X = pX, Y = pY, Z = pZ, uo = pU, vo = pV;
        if (base + 64 < n) {                                              // wave-uniform: next pass in flight
            const int cn = base + 64 + lane;
            const int64_t gn = start + (cn < n ? cn : 0);
            pX = a.x[gn]; pY = a.y[gn]; pZ = a.z[gn]; pU = a.u[gn]; pV = a.v[gn];
        }
        double ru, rv, J[2 * D];
        corner_block<MODEL, OF, false, true>(th, fc, X, Y, Z, uo, vo, ru, rv, J, J + D);      // rotation columns in the phi basis
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
                        const double v0 = i < D ? sw * J[h * D + (i < D ? i : 0)] : (i == D ? sw * (h ? rv : ru) : 0.0);
                        const double v1 = (i + 1) < D ? sw * J[h * D + ((i + 1) < D ? (i + 1) : 0)] : ((i + 1) == D ? sw * (h ? rv : ru) : 0.0);
                        *reinterpret_cast<double2*>(row + h * RS + i) = make_double2(v0, v1);
                    }
                }
            }
            wsync();
            const int npairs = (min(GRAM_TILE_CORNERS, nv - half * GRAM_TILE_CORNERS) + 1) >> 1;
            const double* rd = tile + rd_off;
            int m = 0;
            for (; m + 1 < npairs; m += 2) {
                const double p = rd[(2 * m) * CS];
                const double q = rd[(2 * m + 2) * CS];
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(p, p, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(q, q, acc1, 0, 0, 0);
            }
            if (m < npairs) {
                const double p = rd[(2 * m) * CS];
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(p, p, acc0, 0, 0, 0);
            }
            wsync();
        }
    }
    // Gram -> LDS (aliases the tile), then the compact record.  Columns: camera 0..K-1, pose K..K+5, r = D.
    double* G = tile;
    {
        const d4 acc = acc0 + acc1;
        const int gi = lane >> 4, gj = lane & 15;
#pragma unroll
        for (int v = 0; v < 4; ++v) G[(gi + 4 * v) * 16 + gj] = acc[v];
    }
    wsync();
    double* rec = a.praw[es] + (int64_t)f * a.PRAW;
    if (lane < 21) {
        int i = 0, r = lane;
        while (r > i) { r -= i + 1; ++i; }          // lane -> (i, r) of the packed lower triangle
        rec[lane] = G[(K + i) * 16 + (K + r)];
    }
    for (int e = lane; e < 6 * K1; e += 64) {
        const int i = e / K1, j = e - i * K1;
        rec[21 + e] = G[(K + i) * 16 + (j < K ? j : D)];
    }
    for (int e = lane; e < K1 * K1; e += 64) {
        const int i = e / K1, j = e - i * K1;
        rec[21 + 6 * K1 + e] = G[(i < K ? i : D) * 16 + (j < K ? j : D)];
    }
    if (lane == 0) a.cost_f[f] = G[D * 16 + D];
    if (lane < 9) rec[praw_jl_off(K) + lane] = fc[FC_A + lane];      // the frame's left Jacobian: the elimination maps phi -> rvec with it
}

#endif  // CCAL_LEGACY_KERNELS

// ---------------------------------------------------------------------------------------------
// k_gram1v: the same per-frame record as k_gram1, without the LDS transposition the matrix cores need.
// Measured (profiles/r01, ablations in DESIGN.md): staging sqrt(w)[J|r] rows through LDS in MFMA operand
// order costs ~45 us of k_gram1's 69 us at 10 000 frames, and f64 MFMA shares the FP64 datapath with the
// VALU on gfx950 (no overlap).  Here 16 lanes own one frame (4 frames per wavefront; 144 corners = 9 x 16,
// no idle lanes), every lane keeps the upper triangle of [J|r]^T W [J|r] in registers (structural zeros of
// the intrinsic columns skipped at compile time) and the 16 partial Grams of a frame meet once in LDS.
// ---------------------------------------------------------------------------------------------
// Where each accumulated triangle entry goes inside the compact per-frame record
// C (21) | [B|g] (6 x K1) | A (K1 x K1, both halves), worked out at compile time: dst | mirror << 16 (0xffff = none).
// W = number of leading camera columns whose products with the pose columns are left out (k_gram1w keeps those in LDS
// and writes them separately).
template <int K, int W, bool GEN = false>
struct RecMap {
    static constexpr int D = K + 6, NC = D + 1, K1 = K + 1;
    static constexpr int N = NC * (NC + 1) / 2 - 6 * W;
    uint32_t d[N];
    constexpr RecMap() : d{} {
        int r = 0;
        for (int i = 0; i < NC; ++i)
            for (int j = i; j < NC; ++j) {
                const bool ip = i >= K && i < D, jp = j >= K && j < D;      // pose columns
                if (!ip && jp && i < W) continue;
                const int ci = i < K ? i : K, cj = j < K ? j : K;           // camera-block index (r -> K)
                uint32_t a = 0, b = 0xffff;
                if (!GEN) {
                    if (ip && jp) a = (j - K) * (j - K + 1) / 2 + (i - K);
                    else if (!ip && jp) a = 21 + (j - K) * K1 + ci;             // i camera, j pose
                    else if (ip && !jp) a = 21 + (i - K) * K1 + K;              // i pose, j = r
                    else { a = 21 + 6 * K1 + ci * K1 + cj; b = 21 + 6 * K1 + cj * K1 + ci; }
                } else {                                                        // GEN record (ccal_fused.hpp)
                    if (ip && jp) { a = (i - K) * 6 + (j - K); if (i != j) b = (j - K) * 6 + (i - K); }
                    else if (!ip && jp) a = 36 + ci * 6 + (j - K);
                    else if (ip && !jp) a = 36 + K * 6 + (i - K);
                    else a = gen_a_off(K) + cj * (cj + 1) / 2 + ci;            // packed lower triangle (ci <= cj)
                }
                d[r++] = a | (b << 16);
            }
    }
};
template <int K, int W, bool GEN = false> __device__ const RecMap<K, W, GEN> g_recmap = RecMap<K, W, GEN>();

#ifndef CCAL_GRAMV_WPB
#define CCAL_GRAMV_WPB 2          // wavefronts per workgroup (4 frames each)
#endif
// structural zeros of the block Jacobian rows (u row: no fy, cy; v row: no fx, cx); f feeds both rows
template <bool OF> __device__ constexpr bool nz_u(int i) { return OF ? (i != 2) : (i != 1 && i != 3); }
template <bool OF> __device__ constexpr bool nz_v(int i) { return OF ? (i != 1) : (i != 0 && i != 2); }

// LPF = lanes per frame (16, 32 or 64): small problems spread a frame over more lanes so that the chip still fills.
// GEN: one camera's blocks of the general (multi-camera) loop - the frames are those of a.list, the slot pose is read as it
// stands (k_backsub has formed it) and composed with the camera's extrinsics (frame_setup_composed): the six pose columns
// are those of the COMPOSED pose, in the phi basis, for every camera; the record goes to the observation frame's place in
// the Gram buffer (a.rec_off) together with the blocks of the matrix that expands it to the block's reference columns.
// ITER (single-camera loop, session sizes; IterArgs in ccal_fused.hpp): a group in ONE launch - four wavefronts per workgroup;
// in front of the evaluation every workgroup sums the previous launch's rows (reduce_partial_rows: k_head's order), its first
// wavefront decides and solves the camera system (head_wave), and the evaluation takes state, camera step and candidate
// intrinsics from LDS; behind it the four wavefronts' rows of partial sums are added in LDS: one row per workgroup.
// What changes from one single-launch group of a solve to the next - in the argument block (IterArgs) when a launch serves ONE
// problem, derived from the launch number when it serves a whole batch (k_gram1v_batch: the table of argument blocks is written once).
struct IterDyn { int32_t seq, skip_head, fold; const DevState* st_in; DevState* st_out; const double* partial_in; double* partial_out; };
template <int MODEL, bool OF, int LPF, bool GEN, bool ITER>
__device__ __forceinline__ void gram1v_body(const FusedArgs& a, const IterDyn& dy) {
    static_assert(!(ITER && GEN), "single-launch groups: single-camera loop only");
    constexpr int WPB = ITER ? 4 : CCAL_GRAMV_WPB;
    constexpr int G = 64 / LPF;                     // frames per wavefront
    constexpr int D = block_dim(MODEL, OF, false);
    constexpr int K = D - 6, K1 = K + 1;
    constexpr int NC = D + 1;                       // columns of [J | r]
    constexpr int NE = NC * (NC + 1) / 2;           // upper triangle
    constexpr int HALF = (NE + 1) / 2;              // entries reduced per LDS round
    constexpr int LS = HALF | 1;                    // odd row stride (doubles): conflict-free column sums
    constexpr int WSL = G * FC_N0P + 64 * LS;           // per wave: G frames' constants | reduction buffer
    constexpr int NQ = (G * HALF + 63) / 64;        // (frame, entry) sums per lane and round
    extern __shared__ double smem[];
#ifdef CCAL_STAMPS          // diagnostic build (tools/stamps_g1v.py): the phases of every wavefront, 8 stamps per wavefront in the per-frame scratch
    long long ts[12] = {};
    ts[0] = wall_clock64();
#define G1V_STAMP(i) ts[i] = wall_clock64()
#else
#define G1V_STAMP(i)
#endif
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // LPF need not divide 64 (12 lanes x 5 frames, 6 x 10): the lanes beyond G * LPF idle along with group G - 1
    const bool lane_ok = lane < G * LPF;
    const int grp = lane_ok ? lane / LPF : G - 1, gl = lane % LPF;
    const int f = (blockIdx.x * WPB + wave) * G + grp;
    const bool active = lane_ok && f < a.n_obs;
    const int fa_ = GEN ? a.list[active ? f : 0] : (active ? f : 0);      // observation frame (GEN: the camera's list)
    // ITER: everything the prologue below reads from global memory is requested in FRONT of the decision - the frame's offsets
    // and slot before the sum over the previous launch's rows, pose / elimination record / first corner rows (both parameter
    // sets: the decision picks one) behind that sum's loads and before its workgroup barriers: the prologue's two dependent round
    // trips run beside the decision instead of behind it (stamps, 625 frames: prologue 3.2 -> 2.1 us)
    constexpr int PFLEN = 33 + 6 * (D - 6 + 1);                // the slot's elimination record: L (21) | Y (6 x K1) | g_p (6) | diag C (6)
    constexpr int NPFQ = ITER ? (PFLEN + LPF - 1) / LPF : 1;
    int64_t start_p = 0; int n_p = 0, slot_p = 0;
    float pX = 0.f, pY = 0.f, pZ = 0.f, pU = 0.f, pV = 0.f;
    double pose_p[2][ITER ? 6 : 1], pf_p[2][NPFQ];
    if constexpr (ITER) {
        start_p = a.obs_off[fa_];
        n_p = active ? (int)(a.obs_off[fa_ + 1] - start_p) : 0;
        slot_p = a.obs_slot[fa_];
    }
    bool fold_first = false;                        // ITER: this launch is also the solve's k_unpack1 (IterArgs::fold)
    if constexpr (ITER) fold_first = dy.skip_head != 0 && dy.fold != 0;
    auto iter_prefetch2 = [&]() {
      if constexpr (ITER) {
        const int64_t g0 = start_p + (gl < n_p ? gl : 0);
        pX = a.x[g0]; pY = a.y[g0]; pZ = a.z[g0]; pU = a.u[g0]; pV = a.v[g0];
        // (fold: the starting pose from where the caller left it - pinned host memory, or set 0 for device-resident solves)
        const double* ps0 = fold_first ? (a.it.poses_on_device ? a.poses[0] : a.it.poses_src) : a.poses[0];
        const double* ps1 = fold_first ? ps0 : a.poses[1];
#pragma unroll
        for (int i = 0; i < 6; ++i) { pose_p[0][i] = ps0[(int64_t)slot_p * 6 + i]; pose_p[1][i] = ps1[(int64_t)slot_p * 6 + i]; }
#pragma unroll
        for (int q = 0; q < NPFQ; ++q) {
            const int e = min(gl + LPF * q, PFLEN - 1);
            pf_p[0][q] = a.pf[0][(int64_t)slot_p * a.PF + e]; pf_p[1][q] = a.pf[1][(int64_t)slot_p * a.PF + e];
        }
      }
    };
    // what the evaluation needs of the optimizer state, in registers (the single-launch form decides in this kernel: LDS; else global)
    struct { int done, redo, cur, first, method; double lambda_solve, lam_schur; } g;
    constexpr int PROW = ITER ? iter_row_len(D - 6) : 1;       // ITER: a workgroup's row of partial sums, symmetric blocks packed
    __shared__ double it_rows[ITER ? 4 : 1][ITER ? fused_red_size(D - 6) : 1];
    __shared__ typename std::conditional<ITER, HeadShared, int>::type hsh;
    bool from_lds = false;                          // ITER, not the first launch: state, camera step and candidate are in hsh
    if constexpr (ITER) {
        const IterArgs& it = a.it;
        if (!dy.skip_head) {
            // (the rows are summed also for a solve that has finished: asking first would put a memory round trip in front of every group's loads)
            HeadIO io;
            io.st_in = dy.st_in; io.st_out = dy.st_out; io.hs = it.hs; io.red_g = nullptr; io.cols = it.cols;
            io.intr[0] = a.intr[0]; io.intr[1] = a.intr[1]; io.dc = it.dc_out; io.K = D - 6; io.seq = dy.seq;
            io.min_diag = a.min_diag; io.max_diag = a.max_diag; io.publish_all = it.publish_all;
            HeadPre hpre = {};
            if (threadIdx.x < 64) hpre = head_prefetch(io, (int)threadIdx.x);
            __shared__ double shr[4][(PROW + 63) / 64][64];
            double vsum[(PROW + 63) / 64][4];
            iter_reduce_load<D - 6>(dy.partial_in, it.n_part_in, vsum);
            iter_prefetch2();
            iter_reduce_combine<D - 6>(vsum, hsh.red, shr);
            G1V_STAMP(7);                                   // the previous launch's rows are summed
            const bool writer = blockIdx.x == 0;
            if (threadIdx.x < 64) head_wave(io, hsh, (int)threadIdx.x, writer, hpre);
            __syncthreads();
            G1V_STAMP(8);                                   // decided, camera system solved
            head_finish(io, hsh, it.result_host, a.poses[0], a.poses[1], it.np6, writer, it.result_poses, it.done_cnt);
            G1V_STAMP(9);
            from_lds = true;
            const DevState& S = hsh.S0;
            g.done = S.done; g.redo = S.redo; g.cur = S.cur; g.first = S.first; g.method = S.method;
            g.lambda_solve = S.lambda_solve; g.lam_schur = schur_lambda(&S);
        } else if (fold_first) {
            iter_prefetch2();
            // the solve's first launch AND its k_unpack1: state, columns and intrinsics from the argument block
            constexpr int NS = (int)(sizeof(DevState) / sizeof(double)), NC1 = (int)(sizeof(ColInfo) / sizeof(double));
            if (threadIdx.x < CCAL_PMAX) hsh.cand[threadIdx.x] = it.poses_on_device ? a.intr[0][threadIdx.x] : reinterpret_cast<const double*>(it.intr_h)[threadIdx.x];
            if (blockIdx.x == 0 && threadIdx.x < 64) {
                for (int e = threadIdx.x; e < NS; e += 64) reinterpret_cast<double*>(dy.st_out)[e] = reinterpret_cast<const double*>(&it.st0)[e];
                for (int e = threadIdx.x; e < it.n_cols * NC1; e += 64) reinterpret_cast<double*>(it.cols_out)[e] = reinterpret_cast<const double*>(it.col0)[e];
                if (threadIdx.x < CCAL_PMAX) {
                    const double v = it.poses_on_device ? a.intr[0][threadIdx.x] : reinterpret_cast<const double*>(it.intr_h)[threadIdx.x];
                    if (!it.poses_on_device) a.intr[0][threadIdx.x] = v;
                    a.intr[1][threadIdx.x] = v;
                }
                if (threadIdx.x == 0) it.hs->word = status_word(dy.seq, 0, 0);
            }
            __syncthreads();
            from_lds = true;                       // (the intrinsics: hsh.cand; the camera step is not read in a first evaluation)
            const DevState& S = it.st0;
            g.done = S.done; g.redo = S.redo; g.cur = S.cur; g.first = S.first; g.method = S.method;
            g.lambda_solve = S.lambda_solve; g.lam_schur = schur_lambda(&S);
        } else {
            iter_prefetch2();
            // the solve's first launch: the starting state passes through to the buffer the next launch reads
            if (blockIdx.x == 0 && threadIdx.x < 64) {
                const double* src = reinterpret_cast<const double*>(dy.st_in);
                double* dst = reinterpret_cast<double*>(dy.st_out);
                for (int e = threadIdx.x; e < (int)(sizeof(DevState) / sizeof(double)); e += 64) dst[e] = src[e];
                if (threadIdx.x == 0) it.hs->word = status_word(dy.seq, 0, 0);
            }
            const DevState* st = dy.st_in;
            g.done = st->done; g.redo = st->redo; g.cur = st->cur; g.first = st->first; g.method = st->method;
            g.lambda_solve = st->lambda_solve; g.lam_schur = schur_lambda(st);
        }
    } else {
        const DevState* st = a.st;
        g.done = st->done; g.redo = st->redo; g.cur = st->cur; g.first = st->first; g.method = st->method;
        g.lambda_solve = st->lambda_solve; g.lam_schur = schur_lambda(st);
    }
    // the four wavefronts' rows -> the workgroup's row (fixed order), the two symmetric blocks as their upper triangles
    auto iter_row_out = [&]() {
        if constexpr (ITER) {
            __syncthreads();
            if (threadIdx.x < 64) for (int pe = threadIdx.x; pe < PROW; pe += 64) {
                const int e = iter_row_src(D - 6, pe);
                dy.partial_out[(int64_t)blockIdx.x * PROW + pe] = ((it_rows[0][e] + it_rows[1][e]) + it_rows[2][e]) + it_rows[3][e];
            }
        }
    };
    // fused elimination (launch_gram1v_t decides): no separate elimination launch; a re-elimination group then runs here too
    const bool fuse = !GEN && a.fuse_elim != 0;
    // the records in HBM are what a re-elimination group reads: Gauss-Newton never has one, so a fused GN group skips the stores
    const bool keep_rec = GEN || !fuse || g.method == CCAL_METHOD_LM;
    if (g.done || (g.redo && !fuse)) return;            // finished, or a re-elimination group without fusion (no evaluation)
    G1V_STAMP(1);                                           // the state has arrived
    const int camf = (GEN && a.obs_cam) ? a.obs_cam[fa_] : a.cam;          // GEN, merged launch: the frame's camera
    double* fcw = smem + wave * WSL;
    double* fc = fcw + grp * FC_N0P;
    double* red = fcw + G * FC_N0P;
    const int cur = g.cur, first = g.first;
    const int es = first ? cur : (cur ^ 1);
    if constexpr (!GEN) {
        constexpr int REC_ = praw_jl_off(K) + 9, GS_ = (REC_ + 6 * K1 + 1) & ~1;
        static_assert(G * GS_ <= 64 * LS, "the frames' records fit the reduction buffer");
        if (g.redo) {
            // re-elimination group (LM: rejected step or missed speculation): the accepted set's stored records, new damping
            double* R = red + grp * GS_;
            double mcv = 0.0;
            int slot_r = 0;
            if (active) {
                const double* rec = a.praw[cur] + (int64_t)f * a.PRAW;
                slot_r = a.obs_slot[f];
                for (int e = gl; e < REC_; e += LPF) R[e] = rec[e];
                if (gl == 0) mcv = a.mc_f[f];
            }
            wsync();
            gram_fused_tail<K, LPF>(a, g.lam_schur, red, ITER ? it_rows[wave] : a.partial + (int64_t)(blockIdx.x * WPB + wave) * fused_red_size(K), grp, gl, lane_ok, active, slot_r, cur, mcv);
            iter_row_out();
            return;
        }
    }
    double th[th_len<MODEL>()];
    if constexpr (ITER) {
        if (from_lds) load_theta<MODEL, OF>(hsh.cand, a.rt, th);           // the candidate this launch has just formed
        else load_theta<MODEL, OF>(a.intr[es], a.rt, th);
    } else {
        load_theta<MODEL, OF>(a.intr[es] + ((GEN && a.obs_cam) ? camf * CCAL_PMAX : 0), a.rt, th);
    }
    const int64_t start = ITER ? start_p : a.obs_off[fa_];
    const int n = ITER ? n_p : (active ? (int)(a.obs_off[fa_ + 1] - start) : 0);
    if constexpr (!ITER) {
        const int64_t g0 = start + (gl < n ? gl : 0);
        pX = a.x[g0]; pY = a.y[g0]; pZ = a.z[g0]; pU = a.u[g0]; pV = a.v[g0];
    }
    {
        // candidate pose of this group's frame (back-substitution of the previous camera solve) + constants;
        // the 16 lanes of a group compute the same values, the 4 groups work on 4 frames at once
        int slot;                           // (identity table: no round trip in front of the pose and the elimination record)
        if constexpr (ITER) slot = slot_p; else { if (!GEN && a.slot_ident) slot = fa_; else slot = a.obs_slot[fa_]; }
        double pose[6];
        if constexpr (ITER) {
#pragma unroll
            for (int i = 0; i < 6; ++i) pose[i] = cur ? pose_p[1][i] : pose_p[0][i];
            // the accepted set's elimination record of the frame's slot: from the lanes' registers into LDS (the reduction
            // buffer is free until the corner loop has ended)
            constexpr int PFS = (PFLEN + 1) & ~1;
            static_assert(G * PFS <= 64 * LS, "the frames' elimination records fit the reduction buffer");
            if (lane_ok) {
#pragma unroll
                for (int q = 0; q < NPFQ; ++q) if (gl + LPF * q < PFLEN) red[grp * PFS + gl + LPF * q] = cur ? pf_p[1][q] : pf_p[0][q];
            }
            wsync();
        } else {
            // GEN: k_backsub has formed the candidate - or, FusedArgs::gen_backsub, it is formed here from the accepted pose
            const bool gbs = GEN && a.gen_backsub != 0 && !first;
#pragma unroll
            for (int i = 0; i < 6; ++i) pose[i] = a.poses[(GEN && !gbs) ? es : cur][(int64_t)slot * 6 + i];
            if constexpr (GEN) {
                if (gbs) {
                    const double mcg = gen_backsub_pose<LPF, G, 64 * LS>(a, slot, g.lambda_solve, pose, red, grp, gl, lane_ok);
                    if (active && a.g_owner[fa_]) {
#pragma unroll
                        for (int i = 0; i < 6; ++i) if (gl == i) a.poses[es][(int64_t)slot * 6 + i] = pose[i];
                        if (gl == 0) a.g_mc_slot[slot] = mcg;
                    }
                }
            }
        }
        double mc = 0.0;
        if (!GEN && !first) {
            const double* pf;
            if constexpr (ITER) pf = red + grp * ((PFLEN + 1) & ~1); else pf = a.pf[cur] + (int64_t)slot * a.PF;
            double dcr[K];                                  // the camera step
            if constexpr (ITER) {
#pragma unroll
                for (int j = 0; j < K; ++j) dcr[j] = hsh.x[j];    // (not the first launch: this launch's own solve)
            } else {
#pragma unroll
                for (int j = 0; j < K; ++j) dcr[j] = a.dc[j];
            }
            if (pf[0] != 0.0) {
                double dp[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const double* yr = pf + 21 + i * K1;
                    double t = yr[K];
#pragma unroll
                    for (int j = 0; j < K; ++j) t += yr[j] * dcr[j];
                    dp[i] = -t;
                }
#pragma unroll
                for (int i = 5; i >= 0; --i) {
                    double t = dp[i];
#pragma unroll
                    for (int k = i + 1; k < 6; ++k) t -= pf[k * (k + 1) / 2 + i] * dp[k];
                    dp[i] = t * pf[i * (i + 1) / 2 + i];
                }
                const double lam = g.lambda_solve;
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const double gp = pf[21 + 6 * K1 + i], dCi = pf[21 + 6 * K1 + 6 + i];
                    const double Dii = lam > 0.0 ? lam * clampd1(dCi, a.min_diag, a.max_diag) : 0.0;
                    mc += dp[i] * (Dii * dp[i] - gp);
                    pose[i] += dp[i];
                }
            }
            if (active) {
#pragma unroll
                for (int i = 0; i < 6; ++i) if (gl == i) a.poses[es][(int64_t)slot * 6 + i] = pose[i];
            }
        }
        if constexpr (ITER) {
            if (fold_first && active) {            // k_unpack1's part of this frame: the starting pose into the parameter sets
#pragma unroll
                for (int i = 0; i < 6; ++i) if (gl == i) { if (!a.it.poses_on_device) a.poses[0][(int64_t)slot * 6 + i] = pose[i]; a.poses[1][(int64_t)slot * 6 + i] = pose[i]; }
            }
        }
        if (!GEN && active && gl == 0) a.mc_f[f] = mc;
        if constexpr (GEN) {
            // composed pose T_c0 o T_0b and the 6 x 12 expansion matrix E (frame_setup_composed); E^T goes straight into
            // the frame's record, where k_schur finds it
            const bool other = camf > 0;
            double ex[6], fcr[12], ept[GEN_EPT];
#pragma unroll
            for (int i = 0; i < 6; ++i) ex[i] = other ? a.extr[es][camf * 6 + i] : 0.0;
            frame_setup_composed(pose, ex, fcr, ept);
            if (gl == 0 && lane_ok) {
#pragma unroll
                for (int i = 0; i < 12; ++i) fc[i] = fcr[i];
            }
            if (gl == 0 && active) {
                double ec[GEN_EC];
                gen_et_compact(ept, ec);
                double2* rec = reinterpret_cast<double2*>(a.praw[es] + a.rec_off[fa_] + gen_e_off(K));
#pragma unroll
                for (int i = 0; i < GEN_EC / 2; ++i) rec[i] = make_double2(ec[2 * i], ec[2 * i + 1]);
            }
        } else {
            double fcr[FC_N0];
            frame_setup<false>(pose, nullptr, fcr);
            if (gl == 0 && lane_ok) {
#pragma unroll
                for (int i = 0; i < FC_N0; ++i) fc[i] = fcr[i];
            }
        }
    }
    wsync();
    G1V_STAMP(2);                                           // prologue done: frame constants in LDS

    double acc[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) acc[e] = 0.0;
    // same trip count for the whole wave: the largest frame of the four
    int nmax = n;
#pragma unroll
    for (int off = ((LPF & (LPF - 1)) == 0 ? LPF : 1); off < 64; off <<= 1) nmax = max(nmax, __shfl_xor(nmax, off, 64));
    for (int base = 0; base < nmax; base += LPF) {
        const int c = base + gl;
        const bool valid = c < n;
        const double X = pX, Y = pY, Z = pZ, uo = pU, vo = pV;
        if (base + LPF < nmax) {
            const int cn = base + LPF + gl;
            const int64_t gn = start + (cn < n ? cn : 0);
            pX = a.x[gn]; pY = a.y[gn]; pZ = a.z[gn]; pU = a.u[gn]; pV = a.v[gn];
        }
        double ru, rv, J[2 * D];
        corner_block<MODEL, OF, false, true>(th, fc, X, Y, Z, uo, vo, ru, rv, J, J + D);       // rotation columns in the phi basis
        const double sw = valid ? huber_sqrt_weight(ru * ru + rv * rv, a.huber_delta) : 0.0;
        // sqrt(w)-scaled rows (only the structurally non-zero entries are ever touched)
        double su[NC], sv[NC];
#pragma unroll
        for (int i = 0; i < D; ++i) { su[i] = sw * J[i]; sv[i] = sw * J[D + i]; }
        su[D] = sw * ru; sv[D] = sw * rv;
        int e = 0;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
#pragma unroll
            for (int j = i; j < NC; ++j) {
                const bool bu = (i >= K || nz_u<OF>(i)) && (j >= K || nz_u<OF>(j));
                const bool bv = (i >= K || nz_v<OF>(i)) && (j >= K || nz_v<OF>(j));
                if (bu) acc[e] = __builtin_fma(su[i], su[j], acc[e]);
                if (bv) acc[e] = __builtin_fma(sv[i], sv[j], acc[e]);
                ++e;
            }
        }
    }

    G1V_STAMP(3);                                           // corner loop done
    // 16 partial Grams per frame -> one, through LDS, in two halves of the triangle
    double res[2][NQ];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        wsync();
#pragma unroll
        for (int t = 0; t < HALF; ++t) { const int e = h * HALF + t; if (e < NE) red[lane * LS + t] = acc[e < NE ? e : 0]; }
        wsync();
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int idx = lane + 64 * q;                      // (group, entry) pairs
            double sum = 0.0;
            if (idx < G * HALF) {
                const int g = idx / HALF, t = idx - g * HALF;
                const double* src = red + (g * LPF) * LS + t;
#pragma unroll
                for (int l = 0; l < LPF; ++l) sum += src[l * LS];
            }
            res[h][q] = sum;
        }
    }
    G1V_STAMP(4);                                           // lane sums reduced
    // scatter the upper triangle into the compact record  C (21) | [B|g] (6 x K1) | A (K1 x K1)
    const int fbase = (blockIdx.x * WPB + wave) * G;
    // fused elimination: what its tail needs from memory is requested now, behind the reductions
    int slot_t = 0;
    double mc_t = 0.0;
    if constexpr (!GEN) {
        if (fuse && active) { slot_t = a.obs_slot[f]; if (gl == 0) mc_t = a.mc_f[f]; }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int idx = lane + 64 * q;
            if (idx >= G * HALF) continue;
            const int g = idx / HALF, t = idx - g * HALF, e = h * HALF + t;
            const int ff = fbase + g;
            if (e >= NE || ff >= a.n_obs) continue;
            const uint32_t m = g_recmap<K, 0, GEN>.d[e];
            const double v = res[h][q];
            double* rec = a.praw[es] + (GEN ? a.rec_off[a.list[ff]] : (int64_t)ff * a.PRAW);
            if (keep_rec) {
                rec[m & 0xffff] = v;
                if ((m >> 16) != 0xffff) rec[m >> 16] = v;
                if (!GEN && e == NE - 1) a.cost_f[ff] = v;               // the last entry is r x r
            }
            if constexpr (!GEN) {
                if (fuse) {                                          // fused elimination: the same record in LDS (red is free: every sum is in registers)
                    constexpr int GS_ = (praw_jl_off(K) + 9 + 6 * K1 + 1) & ~1;
                    red[g * GS_ + (m & 0xffff)] = v;
                    if ((m >> 16) != 0xffff) red[g * GS_ + (m >> 16)] = v;
                }
            }
        }
    }
    // the frame's left Jacobian (phi -> rvec map of the elimination)
    if (!GEN && active && keep_rec) for (int e = gl; e < 9; e += LPF) a.praw[es][(int64_t)f * a.PRAW + praw_jl_off(K) + e] = fc[FC_A + e];
    if constexpr (!GEN) {
        if (fuse) {
            constexpr int GS_ = (praw_jl_off(K) + 9 + 6 * K1 + 1) & ~1;
            if (lane_ok) for (int e = gl; e < 9; e += LPF) red[grp * GS_ + praw_jl_off(K) + e] = fc[FC_A + e];
            wsync();
            G1V_STAMP(5);                                   // record assembled (LDS and / or HBM)
            gram_fused_tail<K, LPF>(a, g.lam_schur, red, ITER ? it_rows[wave] : a.partial + (int64_t)(blockIdx.x * WPB + wave) * fused_red_size(K), grp, gl, lane_ok, active, slot_t, es, mc_t);
            iter_row_out();
        }
    }
#ifdef CCAL_STAMPS
    if (!GEN && lane == 0) { ts[6] = wall_clock64(); const int wg = blockIdx.x * WPB + wave; for (int i = 0; i < 10; ++i) a.fcbuf[16 * wg + i] = (double)ts[i]; }
#endif
#undef G1V_STAMP
}
template <int MODEL, bool OF, int LPF, bool GEN, bool ITER = false>
__global__ __launch_bounds__(64 * (ITER ? 4 : CCAL_GRAMV_WPB), 1) void k_gram1v(const FusedArgs) {
    // The argument block is read where it lies - the kernarg segment, offset 0 - through a pointer: handing the by-value parameter
    // to the (inlined) body by reference made the compiler copy all 1 088 bytes of it into scratch, because the body indexes its
    // pointer arrays (a.intr[set], a.poses[set], ...) at run time; a pointer into constant memory keeps those scalar loads at a
    // computed offset, which is also exactly what the batched kernel below does with its table.
    const FusedArgs& a = *(const FusedArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    IterDyn dy = {};
    if constexpr (ITER) {
        dy.seq = a.it.seq; dy.skip_head = a.it.skip_head; dy.fold = a.it.fold;
        dy.st_in = a.it.st_in; dy.st_out = a.it.st_out; dy.partial_in = a.it.partial_in; dy.partial_out = a.partial;
    }
    gram1v_body<MODEL, OF, LPF, GEN, ITER>(a, dy);
}
// ccal_solve_batch: ONE launch per optimizer step for a whole batch of session-sized problems (same model, focal mode and lane
// mapping): blockIdx.y = the problem, `tab` = the problems' argument blocks in device memory, written once per batch; what a
// single-problem launch finds in IterArgs per step follows from the launch number s here - the two state buffers and the two
// halves of the partial-sum buffer alternate by s (tab[.].it.st_in / tab[.].partial: their bases), the first launch has nothing to
// decide and (it.fold) unpacks the starting point.  n problems x 4 launches per solve from n host threads were serialised by the
// runtime's launch path (eight 625-frame sessions: 0.48 ms per batch for ~0.1 ms of device work); this is 4-5 launches per BATCH.
template <int MODEL, bool OF, int LPF>
__global__ __launch_bounds__(256, 1) void k_gram1v_batch(const FusedArgs* __restrict__ tab, const int s) {
    const FusedArgs& a = tab[blockIdx.y];
    if ((int)blockIdx.x >= a.n_part) return;                   // (problems of a batch differ in size: the grid is the largest one's)
    constexpr int RB1 = fused_red_size(block_dim(MODEL, OF, false) - 6);
    IterDyn dy;
    dy.seq = s; dy.skip_head = s == 1 ? 1 : 0; dy.fold = (s == 1 && a.it.fold != 0) ? 1 : 0;
    DevState* const sb = const_cast<DevState*>(a.it.st_in);
    dy.st_in = sb + (s & 1); dy.st_out = sb + ((s + 1) & 1);
    const size_t half = (size_t)a.n_part * RB1;
    dy.partial_in = a.partial + (size_t)((s - 1) & 1) * half; dy.partial_out = a.partial + (size_t)(s & 1) * half;
    gram1v_body<MODEL, OF, LPF, false, true>(a, dy);
}

// k_gram1w: k_gram1v with two wavefronts per SIMD.  k_gram1v needs 256 VGPRs + 66 AGPRs (91 accumulators and the
// hoisted frame constants), i.e. one wavefront per SIMD and nothing to hide the f64 dependency stalls behind.
// Here the 6 K camera x pose accumulators of every lane live in LDS ([entry][lane], stride 65: conflict-free both
// for the lane-private ds_add_f64 of the corner loop - fire and forget, nothing waits on it - and for the column
// sums afterwards), and the frame constants are re-read from LDS every corner instead of being hoisted.
template <int MODEL, bool OF, int LPF, bool GEN>
__global__ __launch_bounds__(64 * CCAL_GRAMV_WPB, 2) void k_gram1w(const FusedArgs a) {
    constexpr int G = 64 / LPF;                     // frames per wavefront
    constexpr int D = block_dim(MODEL, OF, false);
    constexpr int K = D - 6, K1 = K + 1;
    constexpr int NC = D + 1;                       // columns of [J | r]
    constexpr int NE = NC * (NC + 1) / 2;           // upper triangle
#ifndef CCAL_GRAMW_NLC
#define CCAL_GRAMW_NLC 3          // measured at 10 000 frames (EUCM): 6 -> 53.2, 4 -> 51.8, 3 -> 51.0, 2 -> 56.1 us per build
#endif
    // camera columns whose products with the pose columns are LDS accumulators: as few as keeps two wavefronts per SIMD
    constexpr int NLC = CCAL_GRAMW_NLC < K ? CCAL_GRAMW_NLC : K;
    constexpr int NL = 6 * NLC;
    constexpr int NR = NE - NL;                     // the rest: registers
    constexpr int LSA = 65;                         // lane stride of an LDS accumulator row
    constexpr int HALF = (NR + 1) / 2;              // register entries reduced per LDS round
    constexpr int LS = HALF | 1;                    // odd row stride (doubles): conflict-free column sums
    constexpr int RED = NL * LSA > 64 * LS ? NL * LSA : 64 * LS;
    constexpr int WSL = G * FC_N0P + RED;               // per wave: G frames' constants | accumulators / reduction buffer
    constexpr int NQ = (G * HALF + 63) / 64;        // (frame, entry) sums per lane and round
    constexpr int NQA = (G * NL + 63) / 64;
    extern __shared__ double smem[];
    const DevState* st = a.st;
    // fused elimination (launch_gram1v_t decides): no separate elimination launch; a re-elimination group then runs here too
    const bool fuse = !GEN && a.fuse_elim != 0;
    // the records in HBM are what a re-elimination group reads: Gauss-Newton never has one, so a fused GN group skips the stores
    const bool keep_rec = GEN || !fuse || st->method == CCAL_METHOD_LM;
    if (st->done || (st->redo && !fuse)) return;            // finished, or a re-elimination group without fusion (no evaluation)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#ifdef CCAL_STAMPS          // diagnostic build (tools/stamps_gram.py): start / end time of every wavefront into the per-frame scratch
    const long long t_start = wall_clock64();
#endif
    // LPF need not divide 64 (12 lanes x 5 frames, 6 x 10): the lanes beyond G * LPF idle along with group G - 1
    const bool lane_ok = lane < G * LPF;
    const int grp = lane_ok ? lane / LPF : G - 1, gl = lane % LPF;
    const int f = (blockIdx.x * CCAL_GRAMV_WPB + wave) * G + grp;
    const bool active = lane_ok && f < a.n_obs;
    const int fa_ = GEN ? a.list[active ? f : 0] : (active ? f : 0);      // observation frame (GEN: the camera's list)
    const int camf = (GEN && a.obs_cam) ? a.obs_cam[fa_] : a.cam;          // GEN, merged launch: the frame's camera
    double* fcw = smem + wave * WSL;
    double* fc = fcw + grp * FC_N0P;
    double* red = fcw + G * FC_N0P;
    const int cur = st->cur, first = st->first;
    const int es = first ? cur : (cur ^ 1);
    if constexpr (!GEN) {
        constexpr int REC_ = praw_jl_off(K) + 9, GS_ = (REC_ + 6 * K1 + 1) & ~1;
        static_assert(G * GS_ <= RED, "the frames' records fit the reduction buffer");
        if (st->redo) {
            // re-elimination group (LM: rejected step or missed speculation): the accepted set's stored records, new damping
            double* R = red + grp * GS_;
            double mcv = 0.0;
            int slot_r = 0;
            if (active) {
                const double* rec = a.praw[cur] + (int64_t)f * a.PRAW;
                slot_r = a.obs_slot[f];
                for (int e = gl; e < REC_; e += LPF) R[e] = rec[e];
                if (gl == 0) mcv = a.mc_f[f];
            }
            wsync();
            gram_fused_tail<K, LPF>(a, schur_lambda(st), red, a.partial + (int64_t)(blockIdx.x * CCAL_GRAMV_WPB + wave) * fused_red_size(K), grp, gl, lane_ok, active, slot_r, cur, mcv);
            return;
        }
    }
    const double* th_g = a.intr[es] + ((GEN && a.obs_cam) ? camf * CCAL_PMAX : 0);
    double th[th_len<MODEL>()];
    load_theta<MODEL, OF>(th_g, a.rt, th);
    const int64_t start = a.obs_off[fa_];
    const int n = active ? (int)(a.obs_off[fa_ + 1] - start) : 0;
    float pX, pY, pZ, pU, pV;
    {
        const int64_t g0 = start + (gl < n ? gl : 0);
        pX = a.x[g0]; pY = a.y[g0]; pZ = a.z[g0]; pU = a.u[g0]; pV = a.v[g0];
    }
    {
        // candidate pose of this group's frame (back-substitution of the previous camera solve) + constants;
        // the 16 lanes of a group compute the same values, the 4 groups work on 4 frames at once
        const int slot = a.obs_slot[fa_];
        double pose[6];
        // GEN: k_backsub has formed the candidate - or, FusedArgs::gen_backsub, it is formed here from the accepted pose
        const bool gbs = GEN && a.gen_backsub != 0 && !first;
#pragma unroll
        for (int i = 0; i < 6; ++i) pose[i] = a.poses[(GEN && !gbs) ? es : cur][(int64_t)slot * 6 + i];
        if constexpr (GEN) {
            if (gbs) {
                const double mcg = gen_backsub_pose<LPF, G, 0>(a, slot, st->lambda_solve, pose, red, grp, gl, lane_ok);      // (this kernel's buffer holds its LDS accumulators: the record is walked in global memory)
                if (active && a.g_owner[fa_]) {
#pragma unroll
                    for (int i = 0; i < 6; ++i) if (gl == i) a.poses[es][(int64_t)slot * 6 + i] = pose[i];
                    if (gl == 0) a.g_mc_slot[slot] = mcg;
                }
            }
        }
        double mc = 0.0;
        if (!GEN && !first) {
            const double* pf = a.pf[cur] + (int64_t)slot * a.PF;
            if (pf[0] != 0.0) {
                double dp[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const double* yr = pf + 21 + i * K1;
                    double t = yr[K];
#pragma unroll
                    for (int j = 0; j < K; ++j) t += yr[j] * a.dc[j];
                    dp[i] = -t;
                }
#pragma unroll
                for (int i = 5; i >= 0; --i) {
                    double t = dp[i];
#pragma unroll
                    for (int k = i + 1; k < 6; ++k) t -= pf[k * (k + 1) / 2 + i] * dp[k];
                    dp[i] = t * pf[i * (i + 1) / 2 + i];
                }
                const double lam = st->lambda_solve;
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const double gp = pf[21 + 6 * K1 + i], dCi = pf[21 + 6 * K1 + 6 + i];
                    const double Dii = lam > 0.0 ? lam * clampd1(dCi, a.min_diag, a.max_diag) : 0.0;
                    mc += dp[i] * (Dii * dp[i] - gp);
                    pose[i] += dp[i];
                }
            }
            if (active) {
#pragma unroll
                for (int i = 0; i < 6; ++i) if (gl == i) a.poses[es][(int64_t)slot * 6 + i] = pose[i];
            }
        }
        if (!GEN && active && gl == 0) a.mc_f[f] = mc;
        if constexpr (GEN) {
            // composed pose T_c0 o T_0b and the 6 x 12 expansion matrix E (frame_setup_composed); E^T goes straight into
            // the frame's record, where k_schur finds it
            const bool other = camf > 0;
            double ex[6], fcr[12], ept[GEN_EPT];
#pragma unroll
            for (int i = 0; i < 6; ++i) ex[i] = other ? a.extr[es][camf * 6 + i] : 0.0;
            frame_setup_composed(pose, ex, fcr, ept);
            if (gl == 0 && lane_ok) {
#pragma unroll
                for (int i = 0; i < 12; ++i) fc[i] = fcr[i];
            }
            if (gl == 0 && active) {
                double ec[GEN_EC];
                gen_et_compact(ept, ec);
                double2* rec = reinterpret_cast<double2*>(a.praw[es] + a.rec_off[fa_] + gen_e_off(K));
#pragma unroll
                for (int i = 0; i < GEN_EC / 2; ++i) rec[i] = make_double2(ec[2 * i], ec[2 * i + 1]);
            }
        } else {
            double fcr[FC_N0];
            frame_setup<false>(pose, nullptr, fcr);
            if (gl == 0 && lane_ok) {
#pragma unroll
                for (int i = 0; i < FC_N0; ++i) fc[i] = fcr[i];
            }
        }
    }
    wsync();

    double acc[NR];
#pragma unroll
    for (int e = 0; e < NR; ++e) acc[e] = 0.0;
#pragma unroll
    for (int t = 0; t < NL; ++t) red[t * LSA + lane] = 0.0;
    // same trip count for the whole wave: the largest frame of the four
    int nmax = n;
#pragma unroll
    for (int off = ((LPF & (LPF - 1)) == 0 ? LPF : 1); off < 64; off <<= 1) nmax = max(nmax, __shfl_xor(nmax, off, 64));
#ifndef CCAL_GRAMW_HOIST
#define CCAL_GRAMW_HOIST 1      // measured: 10 000 frames 42.5 -> 41.7 us per build, 20 000: 72.7 -> 70.6 (244 VGPRs, still two wavefronts per SIMD)
#endif
#if CCAL_GRAMW_HOIST
    // phi basis: a corner needs R and t only (12 doubles): in registers, no LDS read (and no wait for one) per corner
    double fcl[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) fcl[i] = fc[i];
    const double* fcp = fcl;
#else
    const double* fcp = fc;
#endif
#ifdef CCAL_STAMPS
    const long long t_loop = wall_clock64();
#endif
    for (int base = 0; base < nmax; base += LPF) {
#if !CCAL_GRAMW_HOIST
        asm volatile("" ::: "memory");      // frame constants stay in LDS: no 78 registers of hoisted copies
#endif
        const int c = base + gl;
        const bool valid = c < n;
        const double X = pX, Y = pY, Z = pZ, uo = pU, vo = pV;
        if (base + LPF < nmax) {
            const int cn = base + LPF + gl;
            const int64_t gn = start + (cn < n ? cn : 0);
            pX = a.x[gn]; pY = a.y[gn]; pZ = a.z[gn]; pU = a.u[gn]; pV = a.v[gn];
        }
        double ru, rv, J[2 * D];
        corner_block<MODEL, OF, false, true>(th, fcp, X, Y, Z, uo, vo, ru, rv, J, J + D);      // rotation columns in the phi basis
        const double sw = valid ? huber_sqrt_weight(ru * ru + rv * rv, a.huber_delta) : 0.0;
        // sqrt(w)-scaled rows (only the structurally non-zero entries are ever touched)
        double su[NC], sv[NC];
#pragma unroll
        for (int i = 0; i < D; ++i) { su[i] = sw * J[i]; sv[i] = sw * J[D + i]; }
        su[D] = sw * ru; sv[D] = sw * rv;
        int e = 0, el = 0;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
#pragma unroll
            for (int j = i; j < NC; ++j) {
                const bool bu = (i >= K || nz_u<OF>(i)) && (j >= K || nz_u<OF>(j));
                const bool bv = (i >= K || nz_v<OF>(i)) && (j >= K || nz_v<OF>(j));
                if (i < NLC && j >= K && j < D) {        // camera x pose: LDS accumulator of this lane
                    double t = bu ? su[i] * su[j] : 0.0;
                    if (bv) t = __builtin_fma(sv[i], sv[j], t);
                    __hip_atomic_fetch_add(red + el * LSA + lane, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    ++el;
                } else {
                    if (bu) acc[e] = __builtin_fma(su[i], su[j], acc[e]);
                    if (bv) acc[e] = __builtin_fma(sv[i], sv[j], acc[e]);
                    ++e;
                }
            }
        }
    }
#ifdef CCAL_STAMPS
    const long long t_after = wall_clock64();
#endif
    const int fbase = (blockIdx.x * CCAL_GRAMV_WPB + wave) * G;
    // fused elimination: what its tail needs from memory is requested now, behind the reductions
    int slot_t = 0;
    double mc_t = 0.0;
    if constexpr (!GEN) {
        if (fuse && active) { slot_t = a.obs_slot[f]; if (gl == 0) mc_t = a.mc_f[f]; }
    }
    // camera x pose: LPF lane-private LDS sums per frame -> record  [B|g][pose j][camera i]
    wsync();
    double resa[NQA];
    {
#pragma unroll
        for (int q = 0; q < NQA; ++q) {
            const int idx = lane + 64 * q;
            double sum = 0.0;
            if (idx < G * NL) {
                const int g = idx / NL, t = idx - g * NL;
                const double* src = red + t * LSA + g * LPF;
#pragma unroll
                for (int l = 0; l < LPF; ++l) sum += src[l];
            }
            resa[q] = sum;
        }
#pragma unroll
        for (int q = 0; q < NQA; ++q) {
            const int idx = lane + 64 * q;
            if (idx >= G * NL) continue;
            const int g = idx / NL, t = idx - g * NL;
            const int ff = fbase + g;
            if (ff >= a.n_obs) continue;
            const int i = t / 6, jp = t - 6 * i;
            if (keep_rec) a.praw[es][GEN ? a.rec_off[a.list[ff]] + 36 + i * 6 + jp : (int64_t)ff * a.PRAW + 21 + jp * K1 + i] = resa[q];
        }
    }

    // 16 partial Grams per frame -> one, through LDS, in two halves of the triangle
    double res[2][NQ];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        wsync();
#pragma unroll
        for (int t = 0; t < HALF; ++t) { const int e = h * HALF + t; if (e < NR) red[lane * LS + t] = acc[e < NR ? e : 0]; }
        wsync();
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int idx = lane + 64 * q;                      // (group, entry) pairs
            double sum = 0.0;
            if (idx < G * HALF) {
                const int g = idx / HALF, t = idx - g * HALF;
                const double* src = red + (g * LPF) * LS + t;
#pragma unroll
                for (int l = 0; l < LPF; ++l) sum += src[l * LS];
            }
            res[h][q] = sum;
        }
    }
    // scatter the upper triangle into the compact record  C (21) | [B|g] (6 x K1) | A (K1 x K1)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int idx = lane + 64 * q;
            if (idx >= G * HALF) continue;
            const int g = idx / HALF, t = idx - g * HALF, e = h * HALF + t;
            const int ff = fbase + g;
            if (e >= NR || ff >= a.n_obs) continue;
            const uint32_t m = g_recmap<K, NLC, GEN>.d[e];
            const double v = res[h][q];
            double* rec = a.praw[es] + (GEN ? a.rec_off[a.list[ff]] : (int64_t)ff * a.PRAW);
            if (keep_rec) {
                rec[m & 0xffff] = v;
                if ((m >> 16) != 0xffff) rec[m >> 16] = v;
                if (!GEN && e == NR - 1) a.cost_f[ff] = v;               // the last entry is r x r
            }
            if constexpr (!GEN) {
                if (fuse) {                                          // fused elimination: the same record in LDS (red is free: every sum is in registers)
                    constexpr int GS_ = (praw_jl_off(K) + 9 + 6 * K1 + 1) & ~1;
                    red[g * GS_ + (m & 0xffff)] = v;
                    if ((m >> 16) != 0xffff) red[g * GS_ + (m >> 16)] = v;
                }
            }
        }
    }
    // the frame's left Jacobian (phi -> rvec map of the elimination)
    if (!GEN && active && keep_rec) for (int e = gl; e < 9; e += LPF) a.praw[es][(int64_t)f * a.PRAW + praw_jl_off(K) + e] = fc[FC_A + e];
    if constexpr (!GEN) {
        if (fuse) {
            constexpr int GS_ = (praw_jl_off(K) + 9 + 6 * K1 + 1) & ~1;
#pragma unroll
            for (int q = 0; q < NQA; ++q) {
                const int idx = lane + 64 * q;
                if (idx >= G * NL) continue;
                const int g = idx / NL, t = idx - g * NL, ci = t / 6, jp = t - 6 * ci;
                red[g * GS_ + 21 + jp * K1 + ci] = resa[q];
            }
            if (lane_ok) for (int e = gl; e < 9; e += LPF) red[grp * GS_ + praw_jl_off(K) + e] = fc[FC_A + e];
        }
    }
    if constexpr (!GEN) {
        if (fuse) {
            // the records are in LDS as well (written along with the global stores above): the elimination right here
            wsync();
            gram_fused_tail<K, LPF>(a, schur_lambda(st), red, a.partial + (int64_t)(blockIdx.x * CCAL_GRAMV_WPB + wave) * fused_red_size(K), grp, gl, lane_ok, active, slot_t, es, mc_t);
        }
    }
#ifdef CCAL_STAMPS
    if (!GEN && lane == 0) { const int wg = blockIdx.x * CCAL_GRAMV_WPB + wave; a.fcbuf[2 * wg] = (double)t_start; a.fcbuf[2 * wg + 1] = (double)wall_clock64(); a.fcbuf[8192 + 2 * wg] = (double)t_loop; a.fcbuf[8192 + 2 * wg + 1] = (double)t_after; }
#endif
}

template <int MODEL, bool OF, int LPF, bool W, bool GEN>
static hipError_t launch_gram1v_l(const FusedArgs& a, hipStream_t s) {
    constexpr int G = 64 / LPF;
    constexpr int NC = block_dim(MODEL, OF, false) + 1;
    constexpr int NE = NC * (NC + 1) / 2, NL = 6 * (CCAL_GRAMW_NLC < NC - 7 ? CCAL_GRAMW_NLC : NC - 7);
    constexpr int HALF = ((W ? NE - NL : NE) + 1) / 2;
    constexpr int RED = (W && NL * 65 > 64 * (HALF | 1)) ? NL * 65 : 64 * (HALF | 1);
    constexpr int WSL = G * FC_N0P + RED;
    const size_t lds = sizeof(double) * WSL * CCAL_GRAMV_WPB;
    void (*kern)(const FusedArgs);
    if constexpr (W) kern = k_gram1w<MODEL, OF, LPF, GEN>; else kern = k_gram1v<MODEL, OF, LPF, GEN>;
    static DynLdsGuard lds_guard;
    if (hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds, lds_guard); e != hipSuccess) return e;
    const int fpb = G * CCAL_GRAMV_WPB;
    if (a.n_obs <= 0) return hipSuccess;
    hipLaunchKernelGGL(kern, dim3((a.n_obs + fpb - 1) / fpb), dim3(64 * CCAL_GRAMV_WPB), lds, s, a);
    return hipGetLastError();
}
// Lanes per frame: few lanes = many corner passes per lane but fewer wavefronts, and a wavefront's prologue, reductions and
// (fused) elimination amortised over more corners.  Cost of a launch in units of one corner pass (~1.75 us):
//   k_gram1v (one wavefront per SIMD):   rounds x (c0 + passes),   rounds = ceil(wavefronts / 1 024)
//   k_gram1w (two wavefronts per SIMD):  g(n) x (c0 + passes),     n = wavefronts / 1 024 SIMDs,
//     g = 1 (n <= 1: every wavefront alone on its SIMD), 1 + 0.3 (n - 1) up to n = 2 (the younger partner runs at ~0.57 of
//     the solo speed until it is alone), 0.65 + 0.43 n beyond (wavefronts past 2 048 wait for a slot: a step at n = 2, then
//     pipelined) - fitted to tools/sweep_lpf.py with the fused elimination (6 / 8 / 12 / 16 / 32 / 64 lanes, 1 500-50 000 frames;
//     EUCM us per build, best in brackets: 5 000 frames 37.4 40.2 33.7 [31.2] 37.2 50.7; 10 000: 52.9 45.8 [40.8] 45.7 58.3 82.4;
//     20 000: 85.2 [67.6] 72.1 74.6 98.0 117; 50 000: 159 [138] 151 154 191 267;  KB4 (k_gram1v) 10 000: [54.6] 78.1 63.4 74.3
//     88.4 138; 20 000: [99.4] 115 116 120 168 214).  c0 = 6 passes' worth of prologue + epilogue (6 lanes: 8).
// The second library's CCAL_GRAMV_LPF overrides.  Mappings whose wavefronts would not fit the rows of the partial-sum buffer
// (`max_waves`: the single-camera loop's fused elimination writes one row per wavefront) are left out; six lanes per frame
// (the fewest wavefronts) always fit (fused_ws_ensure sizes the buffer for them).
static int gram_lanes_per_frame(int n_obs, int avg_corners, int slots, int64_t max_waves = (int64_t)1 << 40, int share = 1) {
    static const int cand[6] = { 64, 32, 16, 12, 8, 6 };
    // developer override: only the instantiated mappings (anything else would make the launcher's wavefront count and the
    // kernel it falls back to disagree)
    static const int lpf_env = [] {
        const int v = dev_env_int("CCAL_GRAMV_LPF", 0);
        for (int c : cand) if (v == c) return v;
        return 0;
    }();
    if (lpf_env) return lpf_env;
    int best = 6;
    double best_cost = 1e300;
    for (int i = 0; i < 6; ++i) {
        const int lpf = cand[i], g = 64 / lpf;
        const int64_t waves = ((int64_t)n_obs + g - 1) / g;
        if (((waves + CCAL_GRAMV_WPB - 1) / CCAL_GRAMV_WPB) * CCAL_GRAMV_WPB > max_waves && lpf != 6) continue;
        const int passes = (std::max(avg_corners, 1) + lpf - 1) / lpf;
        const double c0 = lpf == 6 ? 8.0 : 6.0;
        double occ;
        const int simds = std::max(1024 / std::max(share, 1), 64);      // side-by-side sessions (ccal_solve_batch) share the chip
        if (slots > 1024) {
            const double n = (double)waves / (double)simds;
            occ = n <= 1.0 ? 1.0 : (n <= 2.0 ? 1.0 + 0.3 * (n - 1.0) : 0.65 + 0.43 * n);
        } else {
            occ = (double)((waves + simds - 1) / simds);
        }
        const double cost = occ * (c0 + passes);
        if (cost < best_cost) { best_cost = cost; best = lpf; }        // ties: the wider mapping (listed first)
    }
    return best;
}
template <int MODEL, bool OF, bool GEN>
static hipError_t launch_gram1v_t(FusedArgs& a, hipStream_t s) {
    // More wavefronts than SIMDs (>= 2000 frames): k_gram1w, two wavefronts per SIMD (10 000 frames: 40 vs 53 us).
    // Below that every wavefront has a SIMD to itself and k_gram1v's all-register accumulators are a little faster.
    // CCAL_GRAMV_LDSACC=0|1 forces one or the other.
    static const int force = [] { const char* e = dev_env("CCAL_GRAMV_LDSACC"); return e ? (e[0] == '1' ? 1 : 0) : -1; }();
    constexpr int NCt = block_dim(MODEL, OF, false) + 1;
    // larger triangles (KB4: 105 / 120 entries, OPENCV5: 120 / 136) do not fit two wavefronts per SIMD without scratch
    // (one-focal KB4 through k_gram1w: 124 us instead of 64 at 10 000 frames): k_gram1v there
    // (k_gram1w is not even instantiated for them: CCAL_GRAMV_LDSACC=1 has no effect there)
    // The PRODUCT launchers never reach k_gram1w (UCM / EUCM from 2 000 frames go to k_gram2, use_gram2) nor any OPENCV5
    // instantiation of k_gram1v (k_gram2 for every size): they are compiled into the second library only (CCAL_GRAM2=0 there)
#ifdef CCAL_DEV_SWITCHES
    constexpr bool W_OK = NCt * (NCt + 1) / 2 <= 91;
#else
    constexpr bool W_OK = false; (void)NCt;
#endif
    const bool w = W_OK && (force >= 0 ? force == 1 : a.n_obs >= 2000);
    const int lpf = gram_lanes_per_frame(a.n_obs, a.avg_corners, w ? 2048 : 1024, (GEN || !a.fuse_elim) ? (int64_t)1 << 40 : a.part_cap, a.share);
    // fused elimination (single-camera loop): one row of partial sums per wavefront
    const int waves = ((a.n_obs + 64 / lpf - 1) / (64 / lpf) + CCAL_GRAMV_WPB - 1) / CCAL_GRAMV_WPB * CCAL_GRAMV_WPB;
    // (every size: 300 / 625 / 1 000 / 1 280 frames GN 0.135-0.155 ms fused against 0.145-0.172 with a separate elimination launch and
    // the head's own reduction of its <= 40 rows; the second library's CCAL_FUSE_ELIM=0 still takes the separate launch)
    const bool fuse = !GEN && a.fuse_elim != 0 && waves <= a.part_cap;
    a.fuse_elim = fuse ? 1 : 0;
    a.elim_fused = fuse ? 1 : 0;
    if (fuse) a.n_part = waves;
#define CCAL_LPF_CASE(L) case L: if constexpr (W_OK) { if (w) return launch_gram1v_l<MODEL, OF, L, true, GEN>(a, s); } return launch_gram1v_l<MODEL, OF, L, false, GEN>(a, s);
    switch (lpf) {
        CCAL_LPF_CASE(6) CCAL_LPF_CASE(8) CCAL_LPF_CASE(12) CCAL_LPF_CASE(16) CCAL_LPF_CASE(32)
        default: if constexpr (W_OK) { if (w) return launch_gram1v_l<MODEL, OF, 64, true, GEN>(a, s); } return launch_gram1v_l<MODEL, OF, 64, false, GEN>(a, s);
    }
#undef CCAL_LPF_CASE
}
template <bool GEN>
static hipError_t launch_gram1v_m(int model, bool one_focal, FusedArgs& a, hipStream_t s) {
    switch (model * 2 + (one_focal ? 1 : 0)) {
        case 0: return launch_gram1v_t<kUCM, false, GEN>(a, s);
        case 1: return launch_gram1v_t<kUCM, true, GEN>(a, s);
        case 2: return launch_gram1v_t<kEUCM, false, GEN>(a, s);
        case 3: return launch_gram1v_t<kEUCM, true, GEN>(a, s);
        case 4: return launch_gram1v_t<kKB4, false, GEN>(a, s);
        case 5: return launch_gram1v_t<kKB4, true, GEN>(a, s);
#ifdef CCAL_DEV_SWITCHES
        case 6: return launch_gram1v_t<kOCV5, false, GEN>(a, s);
        case 7: return launch_gram1v_t<kOCV5, true, GEN>(a, s);
#else
        case 6: case 7: return hipErrorNotSupported;          // (OPENCV5: k_gram2 for every size)
#endif
        default: return hipErrorInvalidValue;
    }
}
// ---- single-launch groups (k_gram1v<.., ITER>) ----
constexpr int kIterWpb = 4;
constexpr size_t kLdsPerCu = 160 * 1024;
template <int MODEL, bool OF, int LPF>
static size_t iter_lds_bytes() {
    constexpr int G = 64 / LPF, NC = block_dim(MODEL, OF, false) + 1, NE = NC * (NC + 1) / 2, HALF = (NE + 1) / 2;
    constexpr int WSL = G * FC_N0P + 64 * (HALF | 1);
    return sizeof(double) * WSL * kIterWpb;
}
template <int MODEL, bool OF, int LPF>
static hipError_t launch_gram_iter_l(FusedArgs& a, hipStream_t s) {
    constexpr int G = 64 / LPF;
    const size_t lds = iter_lds_bytes<MODEL, OF, LPF>();
    void (*kern)(const FusedArgs) = k_gram1v<MODEL, OF, LPF, false, true>;
    static DynLdsGuard lds_guard;
    if (hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds, lds_guard); e != hipSuccess) return e;
    const int fpb = G * kIterWpb;
    a.fuse_elim = 1; a.elim_fused = 1;
    a.n_part = (a.n_obs + fpb - 1) / fpb;
    hipLaunchKernelGGL(kern, dim3(a.n_part), dim3(64 * kIterWpb), lds, s, a);
    return hipGetLastError();
}
// static LDS of the ITER form beside the dynamic part: HeadShared + the reduction's 4 x 64 + the four rows
template <int MODEL, bool OF>
static constexpr size_t iter_static_lds() { return sizeof(HeadShared) + 4 * 2 * 64 * 8 + 4 * 8 * (size_t)fused_red_size(block_dim(MODEL, OF, false) - 6) + 256; }
template <int MODEL, bool OF>
static int iter_rows_t(int n_obs, int avg_corners, int share, bool launch, FusedArgs* a, hipStream_t s, hipError_t* err) {
    // CCAL_ITER_ROWS: most rows (= workgroups, each of which reads every row of the launch before) for which a group is one
    // launch; 0 = never.  One wavefront per SIMD, every workgroup resident at once: <= 256 workgroups.
    static const int max_rows = std::min(dev_env_int("CCAL_ITER_ROWS", 256), 256);
    if (n_obs <= 0 || max_rows <= 0) return 0;
    const int lpf = gram_lanes_per_frame(n_obs, avg_corners, 1024, (int64_t)1 << 40, share);
    const int g = 64 / lpf, rows = (n_obs + g * kIterWpb - 1) / (g * kIterWpb);
    if (rows > max_rows) return 0;
    size_t lds = 0;
    switch (lpf) {
#define CCAL_IT_CASE(L) case L: lds = iter_lds_bytes<MODEL, OF, L>(); if (launch && lds + iter_static_lds<MODEL, OF>() <= kLdsPerCu) *err = launch_gram_iter_l<MODEL, OF, L>(*a, s); break;
        CCAL_IT_CASE(6) CCAL_IT_CASE(8) CCAL_IT_CASE(12) CCAL_IT_CASE(16) CCAL_IT_CASE(32) CCAL_IT_CASE(64)
#undef CCAL_IT_CASE
        default: return 0;
    }
    if (lds + iter_static_lds<MODEL, OF>() > kLdsPerCu) return 0;
    return rows;
}
// rows (= workgroups) of a single-launch group without reference to a launcher: 0 if the form does not apply
template <int MODEL, bool OF>
static int iter_rows_only(int n_obs, int avg_corners, int share) {
    if (n_obs <= 0) return 0;
    const int lpf = gram_lanes_per_frame(n_obs, avg_corners, 1024, (int64_t)1 << 40, share);
    const int g = 64 / lpf, rows = (n_obs + g * kIterWpb - 1) / (g * kIterWpb);
    if (rows > 256) return 0;
    size_t lds = 0;
    switch (lpf) {
        case 6: lds = iter_lds_bytes<MODEL, OF, 6>(); break;   case 8: lds = iter_lds_bytes<MODEL, OF, 8>(); break;
        case 12: lds = iter_lds_bytes<MODEL, OF, 12>(); break; case 16: lds = iter_lds_bytes<MODEL, OF, 16>(); break;
        case 32: lds = iter_lds_bytes<MODEL, OF, 32>(); break; case 64: lds = iter_lds_bytes<MODEL, OF, 64>(); break;
        default: return 0;
    }
    return lds + iter_static_lds<MODEL, OF>() <= kLdsPerCu ? rows : 0;
}
static int iter_rows_m(int model, bool one_focal, int n_obs, int avg_corners, int share, bool launch, FusedArgs* a, hipStream_t s, hipError_t* err) {
    switch (model * 2 + (one_focal ? 1 : 0)) {
        case 0: return iter_rows_t<kUCM, false>(n_obs, avg_corners, share, launch, a, s, err);
        case 1: return iter_rows_t<kUCM, true>(n_obs, avg_corners, share, launch, a, s, err);
        case 2: return iter_rows_t<kEUCM, false>(n_obs, avg_corners, share, launch, a, s, err);
        case 3: return iter_rows_t<kEUCM, true>(n_obs, avg_corners, share, launch, a, s, err);
        case 4: return iter_rows_t<kKB4, false>(n_obs, avg_corners, share, launch, a, s, err);
        case 5: return iter_rows_t<kKB4, true>(n_obs, avg_corners, share, launch, a, s, err);
#ifdef CCAL_DEV_SWITCHES
        case 6: return iter_rows_t<kOCV5, false>(n_obs, avg_corners, share, launch, a, s, err);
        case 7: return iter_rows_t<kOCV5, true>(n_obs, avg_corners, share, launch, a, s, err);
#else
        // OPENCV5 in the product: only the batched form exists (k_gram1v_batch) - the row count without the single-problem launcher
        case 6: return launch ? 0 : iter_rows_only<kOCV5, false>(n_obs, avg_corners, share);
        case 7: return launch ? 0 : iter_rows_only<kOCV5, true>(n_obs, avg_corners, share);
#endif
        default: return 0;
    }
}
// ---- the batched form (k_gram1v_batch) ----
template <int MODEL, bool OF, int LPF>
static hipError_t launch_iter_batch_l(const FusedArgs* tab, int n, int max_rows, int s_no, hipStream_t s) {
    const size_t lds = iter_lds_bytes<MODEL, OF, LPF>();
    if (lds + iter_static_lds<MODEL, OF>() > kLdsPerCu) return hipErrorInvalidValue;
    void (*kern)(const FusedArgs*, int) = k_gram1v_batch<MODEL, OF, LPF>;
    static DynLdsGuard lds_guard;
    if (hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds, lds_guard); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(max_rows, n), dim3(64 * kIterWpb), lds, s, tab, s_no);
    return hipGetLastError();
}
template <int MODEL, bool OF>
static hipError_t launch_iter_batch_t(int lpf, const FusedArgs* tab, int n, int max_rows, int s_no, hipStream_t s) {
    switch (lpf) {
        case 6: return launch_iter_batch_l<MODEL, OF, 6>(tab, n, max_rows, s_no, s);
        case 8: return launch_iter_batch_l<MODEL, OF, 8>(tab, n, max_rows, s_no, s);
        case 12: return launch_iter_batch_l<MODEL, OF, 12>(tab, n, max_rows, s_no, s);
        case 16: return launch_iter_batch_l<MODEL, OF, 16>(tab, n, max_rows, s_no, s);
        case 32: return launch_iter_batch_l<MODEL, OF, 32>(tab, n, max_rows, s_no, s);
        case 64: return launch_iter_batch_l<MODEL, OF, 64>(tab, n, max_rows, s_no, s);
        default: return hipErrorInvalidValue;
    }
}
// one launch for n problems of one model / focal mode / lane mapping; tab: device memory, n FusedArgs (bases, see the kernel)
hipError_t launch_gram_iter_batch(int model, bool one_focal, int lpf, const FusedArgs* tab, int n, int max_rows, int s_no, hipStream_t s) {
    switch (model * 2 + (one_focal ? 1 : 0)) {
        case 0: return launch_iter_batch_t<kUCM, false>(lpf, tab, n, max_rows, s_no, s);
        case 1: return launch_iter_batch_t<kUCM, true>(lpf, tab, n, max_rows, s_no, s);
        case 2: return launch_iter_batch_t<kEUCM, false>(lpf, tab, n, max_rows, s_no, s);
        case 3: return launch_iter_batch_t<kEUCM, true>(lpf, tab, n, max_rows, s_no, s);
        case 4: return launch_iter_batch_t<kKB4, false>(lpf, tab, n, max_rows, s_no, s);
        case 5: return launch_iter_batch_t<kKB4, true>(lpf, tab, n, max_rows, s_no, s);
        case 6: return launch_iter_batch_t<kOCV5, false>(lpf, tab, n, max_rows, s_no, s);
        case 7: return launch_iter_batch_t<kOCV5, true>(lpf, tab, n, max_rows, s_no, s);
        default: return hipErrorNotSupported;
    }
}
// the lane mapping a single-launch group of this problem takes (what launch_gram_iter dispatches on)
int fused_iter_lpf(int n_obs, int avg_corners, int share) { return gram_lanes_per_frame(n_obs, avg_corners, 1024, (int64_t)1 << 40, share); }

// CCAL_GRAM2_ITER=0 (second library) / -DCCAL_NO_GRAM2_ITER (A/B builds): k_gram1v's single-launch form at every size
static bool use_gram2_iter(bool batch, int share) {
#ifdef CCAL_NO_GRAM2_ITER
    return false;
#else
    static const bool on = dev_env_int("CCAL_GRAM2_ITER", 1) != 0 && dev_env_int("CCAL_ITER_ROWS", 256) > 0;      // (CCAL_ITER_ROWS=0: no single-launch groups at all)
    return on && !batch && share <= 1;
#endif
}
int fused_iter_rows(int model, bool one_focal, int n_obs, int avg_corners, int K, int share, bool batch) {
    if (K != block_dim(model, one_focal, false) - 6) return 0;      // (the kernel's compile-time column count is the problem's)
    // OPENCV5 alone: k_gram2 (fewer AGPR copies) + reduce + head stays ahead - 625 frames GN 0.132 ms against 0.139 in the single-launch
    // form (112-double rows: two chunks per lane to sum, 4.9 us) - so only members of a lockstep batch take it (one launch per step
    // for the whole batch beats n x three).  The second library's CCAL_ITER_OCV5=1 forces the single-launch form.
    static const bool ocv5 = dev_env_int("CCAL_ITER_OCV5", 0) == 1;
    if (model == kOCV5 && !ocv5 && !batch) return 0;
    // UCM / EUCM from 2 000 frames, a problem that has the GPU to itself: the single-launch form of the two-wavefronts-per-SIMD kernel
    // (k_gram2i, ccal_kernels_gram2.hip) - 10 000 frames: ~30 us per group against 36 with k_gram1v's 1 000 one-per-SIMD wavefronts
    if (use_gram2_iter(batch, share)) { const int r = gram2_iter_rows(model, one_focal, n_obs, avg_corners, K); if (r > 0) return r; }
    return iter_rows_m(model, one_focal, n_obs, avg_corners, share, false, nullptr, nullptr, nullptr);
}
hipError_t launch_gram_iter(int model, bool one_focal, FusedArgs& a, hipStream_t s) {
    if (use_gram2_iter(false, a.share) && gram2_iter_rows(model, one_focal, a.n_obs, a.avg_corners, a.K) > 0) return launch_gram2_iter(model, one_focal, a, s);
    hipError_t err = hipErrorInvalidValue;
    const int rows = iter_rows_m(model, one_focal, a.n_obs, a.avg_corners, a.share, true, &a, s, &err);
    return rows > 0 ? err : hipErrorInvalidValue;
}

// single-camera loop; a.fuse_elim in: fusion allowed, out: fusion done (then a.n_part = rows of partial sums, a.elim_fused = 1)
// Which register Gram kernel.  k_gram2 (the block's rows on neighbouring lanes, ccal_kernels_gram2.hip) where it measured faster
// (10 000 / 2 500 frames, us per build, k_gram2 vs k_gram1v|w): OPENCV5 47.1 / 25.6 vs 51.2 / 27.3, EUCM 35.8 / 21.1 vs 36.9 / 22.2
// (from 2 000 frames: two wavefronts per SIMD, no LDS accumulators; 625 frames 15.9 vs 14.2 with k_gram1v); KB4 since round 5
// (neighbouring-lane form): 46.7 / 24.4 vs 49.7 / 25.8, one focal 45.9 vs 47.2.  CCAL_GRAM2=0|1 forces (second library).
static bool use_gram2(int model, int n_obs) {
    static const int force = [] { const char* e = dev_env("CCAL_GRAM2"); return e ? (e[0] == '1' ? 1 : 0) : -1; }();
    if (force >= 0) return force == 1;
    if (model == kOCV5) return true;
    return n_obs >= 2000;
}
static int gram2_lpf_force() { static const int v = dev_env_int("CCAL_GRAM2_LPF", 0); return v; }
hipError_t launch_gram1v(int model, bool one_focal, FusedArgs& a, hipStream_t s) {
    if (use_gram2(model, a.n_obs)) { a.lpf_force = gram2_lpf_force(); return launch_gram2(model, one_focal, a, s); }
    return launch_gram1v_m<false>(model, one_focal, a, s);
}
// one camera's blocks of a multi-camera problem: a.list / a.rec_off / a.n_obs = that camera's observation frames
hipError_t launch_gram1v_general(int model, bool one_focal, const FusedArgs& a0, hipStream_t s) {
    FusedArgs a = a0; a.fuse_elim = 0;
    if (use_gram2(model, a0.n_obs)) { a.lpf_force = gram2_lpf_force(); return launch_gram2_general(model, one_focal, a, s); }
    return launch_gram1v_m<true>(model, one_focal, a, s);
}
#ifdef CCAL_LEGACY_KERNELS
template <int MODEL, bool OF>
static hipError_t launch_gram1_t(const FusedArgs& a, hipStream_t s) {
    constexpr int WS = FC_N0P + GRAM_TILE_CORNERS * 34;
    const size_t lds = sizeof(double) * WS * WAVES_PER_BLOCK;
    static DynLdsGuard lds_guard;
    if (hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(&k_gram1<MODEL, OF>), lds, lds_guard); e != hipSuccess) return e;
    hipLaunchKernelGGL((k_gram1<MODEL, OF>), dim3((a.n_obs + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK), dim3(256), lds, s, a);
    return hipGetLastError();
}
hipError_t launch_gram1(int model, bool one_focal, const FusedArgs& a, hipStream_t s) {
    switch (model * 2 + (one_focal ? 1 : 0)) {
        case 0: return launch_gram1_t<kUCM, false>(a, s);
        case 1: return launch_gram1_t<kUCM, true>(a, s);
        case 2: return launch_gram1_t<kEUCM, false>(a, s);
        case 3: return launch_gram1_t<kEUCM, true>(a, s);
        case 4: return launch_gram1_t<kKB4, false>(a, s);
        case 5: return launch_gram1_t<kKB4, true>(a, s);
        case 6: return launch_gram1_t<kOCV5, false>(a, s);
        case 7: return launch_gram1_t<kOCV5, true>(a, s);
        default: return hipErrorInvalidValue;
    }
}

#else
hipError_t launch_gram1(int, bool, const FusedArgs&, hipStream_t) { return hipErrorNotSupported; }      // not in the product build
#endif

// ---------------------------------------------------------------------------------------------
// The separate elimination launch - superseded by the Gram kernels' fused tail (gram_fused_tail) for every model and size; kept
// in the SECOND library only (-DCCAL_LEGACY_KERNELS), where the matrix-core Gram k_gram1 and CCAL_FUSE_ELIM=0 still use it and
// tools/count_flops.py reads the elimination's operation count off its ISA.
// ---------------------------------------------------------------------------------------------
#ifdef CCAL_LEGACY_KERNELS
// k_schur1m: the elimination with FOUR frames per wavefront (16 lanes each), one pass per wavefront: the grid covers all frames
// (n_pw = 4 ceil(n_obs / 16)).
constexpr int SCHUR1M_WAVES = 8;          // 32 frames per workgroup: a quarter of the partial sums k_head / k_reduce1 have to add up
template <int K>
__global__ __launch_bounds__(64 * SCHUR1M_WAVES) void k_schur1m(const FusedArgs a) {
    constexpr int K1 = K + 1, NA = K1 * K1;
    constexpr int NQ = (NA + 15) / 16;                    // A / Y^T Y entries per lane
    constexpr int REC = 21 + 6 * K1 + NA + 9;             // C (21) | [B|g] (6 x K1) | A (K1 x K1) | J_l (9)
    constexpr int GS = (REC + 6 * K1 + 1) & ~1;           // per frame in LDS: record | Y (6 x K1); later the frame's sums
    constexpr int NF = 4 * SCHUR1M_WAVES;                 // frames per workgroup
    static_assert(2 * NA + 2 <= GS, "a frame's sums reuse its record row");
    __shared__ double smem[NF * GS];
    const DevState* st = a.st;
    if (st->done) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int grp = lane >> 4, gl = lane & 15;
    const int f = (blockIdx.x * SCHUR1M_WAVES + wave) * 4 + grp;
    const bool active = f < a.n_obs;
    double* R = smem + (wave * 4 + grp) * GS;
    double* Ym = R + REC;
    const int set = schur_set(st);
    const double lambda = schur_lambda(st);
    double accA[NQ], accY[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) { accA[q] = 0.0; accY[q] = 0.0; }
    double mcv = 0.0;
    if (active) {
        const double* rec = a.praw[set] + (int64_t)f * a.PRAW;
        for (int e = gl; e < REC; e += 16) R[e] = rec[e];
        if (gl == 0) mcv = a.mc_f[f];
    }
    const int slot = active ? a.obs_slot[f] : 0;
    wsync();
    const bool ok = eliminate_frame<K, 16>(R, Ym, gl, active, lambda, a.min_diag, a.max_diag,
                                          a.pf[set] + (int64_t)slot * a.PF, a.PF, accA, accY);
    // the frames of the workgroup combine in LDS (fixed order), one flush per workgroup; a frame's sums reuse its row
    wsync();
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int e = gl + 16 * q;
        if (e < NA) { R[e] = accA[q]; R[NA + e] = accY[q]; }
    }
    if (gl == 0) { R[2 * NA] = mcv; R[2 * NA + 1] = (active && !ok) ? 1.0 : 0.0; }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * NA + 2; e += 64 * SCHUR1M_WAVES) {
        double t = 0.0;
#pragma unroll
        for (int g = 0; g < NF; ++g) t += smem[g * GS + e];
        a.partial[(int64_t)blockIdx.x * (2 * NA + 2) + e] = t;
    }
}
hipError_t launch_schur1m(FusedArgs& a, hipStream_t s) {
    const dim3 grid((a.n_obs + 4 * SCHUR1M_WAVES - 1) / (4 * SCHUR1M_WAVES)), blk(64 * SCHUR1M_WAVES);
    a.n_part = (int32_t)grid.x;
    if (grid.x == 0) return hipSuccess;
    switch (a.K) {
        case 4: hipLaunchKernelGGL(k_schur1m<4>, grid, blk, 0, s, a); break;
        case 5: hipLaunchKernelGGL(k_schur1m<5>, grid, blk, 0, s, a); break;
        case 6: hipLaunchKernelGGL(k_schur1m<6>, grid, blk, 0, s, a); break;
        case 7: hipLaunchKernelGGL(k_schur1m<7>, grid, blk, 0, s, a); break;
        case 8: hipLaunchKernelGGL(k_schur1m<8>, grid, blk, 0, s, a); break;
        case 9: hipLaunchKernelGGL(k_schur1m<9>, grid, blk, 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

#else
hipError_t launch_schur1m(FusedArgs&, hipStream_t) { return hipErrorNotSupported; }       // not in the product build
#endif

// ---------------------------------------------------------------------------------------------
// k_reduce1: red[e] = sum over workgroups of partial[w][e], fixed order (no atomics: bitwise reproducible)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_reduce1(const double* partial, int n_part, int rb, double* red, const DevState* st) {
    if (st->done) return;
    __shared__ double sh[4][64];
    const int e = blockIdx.x * 64 + (threadIdx.x & 63);
    const double t = reduce_partial_rows(partial, n_part, rb, e, sh);
    if (threadIdx.x < 64 && e < rb) red[e] = t;
}
// Many rows (> kHeadReduceRows, where k_head never reduces itself): one workgroup per entry, 256 threads over the rows,
// shuffle tree + four wavefront sums - the parallel form (10 000 frames = 313 rows: 4.4 us instead of 7.8)
__global__ __launch_bounds__(256) void k_reduce1_wide(const double* partial, int n_part, int rb, double* red, const DevState* st) {
    if (st->done) return;
    __shared__ double sh[4];
    const double* src = partial + blockIdx.x;
    double v0 = 0.0, v1 = 0.0;
    int r = threadIdx.x;
    for (; r + 256 < n_part; r += 512) { v0 += src[(int64_t)r * rb]; v1 += src[(int64_t)(r + 256) * rb]; }
    if (r < n_part) v0 += src[(int64_t)r * rb];
    double v = v0 + v1;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) red[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
hipError_t launch_reduce1(const FusedArgs& a, hipStream_t s) {
    const int rb = fused_red_size(a.K);
    // up to kHeadReduceRows rows the order must be k_head's own (reduce_partial_rows): sharded and single-GPU solves of the
    // same frames then agree bit for bit; beyond, both go through the wide form
    if (a.n_part <= kHeadReduceRows) hipLaunchKernelGGL(k_reduce1, dim3((rb + 63) / 64), dim3(256), 0, s, a.partial, a.n_part, rb, a.red, a.st);
    else hipLaunchKernelGGL(k_reduce1_wide, dim3(rb), dim3(256), 0, s, a.partial, a.n_part, rb, a.red, a.st);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// k_head: one wavefront.  Decision (optimizer_decide, ccal_fused.hpp) on the all-reduced sums, then - when the sums at
// hand are the system to solve - the K x K camera solve and the candidate intrinsics.
// red = [A_dir (K1*K1) | Y^T Y (K1*K1) | mc_pose | failed pose blocks]
// The optimizer state is staged in LDS once (the global copy is touched twice per launch).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_head(const HeadArgs a) {
    __shared__ HeadShared hs;
    if (a.partial) {
        // single-GPU loop: the partial sums of the elimination kernel are added up here in k_reduce1's order (a launch and a
        // kernel boundary less per group); then wavefront 0 decides alone
        if (a.st->done) { if (threadIdx.x == 0) publish_host_status(a.hs, a.st, a.seq, a.publish_all != 0); return; }
        __shared__ double shr[4][64];
        const int rb = fused_red_size(a.K);
        for (int e0 = 0; e0 < rb; e0 += 64) {
            const int e = e0 + (threadIdx.x & 63);
            const double t = reduce_partial_rows(a.partial, a.n_part, rb, e, shr);
            if (threadIdx.x < 64 && e < rb) hs.red[e] = t;
        }
        __syncthreads();
    } else if (a.peers.n > 0) {
        // in-process transport: this step's sums of every rank, added here in rank order (no launch, no buffer in between)
        if (a.st->done) { if (threadIdx.x == 0) publish_host_status(a.hs, a.st, a.seq, a.publish_all != 0); return; }
        const int rb = fused_red_size(a.K);
        for (int e = threadIdx.x; e < rb; e += 256) hs.red[e] = peer_sum(a.peers, e);
        __syncthreads();
    }
    HeadIO io;
    io.st_in = a.st; io.st_out = a.st; io.hs = a.hs; io.red_g = (a.partial || a.peers.n > 0) ? nullptr : a.red; io.cols = a.cols;
    io.intr[0] = a.intr[0]; io.intr[1] = a.intr[1]; io.dc = a.dc; io.K = a.K; io.seq = a.seq;
    io.min_diag = a.min_diag; io.max_diag = a.max_diag; io.publish_all = a.publish_all;
    if (threadIdx.x < 64) head_wave(io, hs, (int)threadIdx.x, true, head_prefetch(io, (int)threadIdx.x));
    __syncthreads();
    head_finish(io, hs, a.result_host, a.poses[0], a.poses[1], a.np6, true);
}
// ccal_build_normal_dev on a single camera: evaluate set 0 as a first evaluation (no pose update) with this damping
__global__ void k_state_eval(DevState* st, double lambda) {
    st->done = 0; st->cur = 0; st->first = 1; st->redo = 0; st->iter = 0;
    st->lambda = lambda; st->lambda_spec = lambda; st->lambda_solve = 0.0;
    st->method = lambda > 0.0 ? CCAL_METHOD_LM : CCAL_METHOD_GN;
}
hipError_t launch_state_eval(DevState* st, double lambda, hipStream_t s) {
    hipLaunchKernelGGL(k_state_eval, dim3(1), dim3(1), 0, s, st, lambda);
    return hipGetLastError();
}
hipError_t launch_head(const HeadArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_head, dim3(1), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace ccal
