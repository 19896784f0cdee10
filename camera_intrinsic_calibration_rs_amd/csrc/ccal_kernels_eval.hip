// Mode E: residual + Jacobian of every block, materialised in HBM (the reference's
// Factor::residual_func evaluated with dual numbers per block, src/optimization/factors.rs:152-173,
// 204-228), plus the residual-only reprojection-error kernel behind validation() (src/util.rs:733-745).
//
// Mapping: one wavefront per observation frame, CCAL_EVAL_WPB (2) frames per 128-thread workgroup.  The frame's
// pose-dependent constants are computed once per wave and staged in LDS; corner rows are read as
// coalesced f32 SoA streams; every lane produces one block (r[2], J[2][D]).  The block Jacobians of
// 64 consecutive corners form one contiguous 64*2*D*8-byte tile of J_out, so they are transposed
// through LDS and written with full 16-B-per-lane coalesced stores.  HBM-write-bound by design.
#include "ccal_device.hpp"
#include "ccal_internal.hpp"

#include <algorithm>
#include <type_traits>

namespace ccal {

__device__ __forceinline__ void wave_lds_sync() {
    // producer and consumer are lanes of the same wavefront: LDS ops of one wave execute in order,
    // this only has to stop the compiler from moving accesses across the hand-off.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#ifndef CCAL_EVAL_WPB
#define CCAL_EVAL_WPB 2          // wavefronts (= observation frames in flight) per workgroup of k_eval
#endif
#ifndef CCAL_EVAL_NT
#define CCAL_EVAL_NT 1           // non-temporal stores for the streamed J tile
#endif
#ifndef CCAL_EVAL_NTR
#define CCAL_EVAL_NTR 0
#endif
#ifndef CCAL_EVAL_PREFETCH
#define CCAL_EVAL_PREFETCH -1    // -1: chosen per launch (see launch_eval_t); 0 / 1 force one form
#endif
#ifndef CCAL_EVAL_PERSIST
#define CCAL_EVAL_PERSIST 0      // >0: that many workgroups per CU, each wave strides over frames
#endif

constexpr int eval_tile_stride(int D) { return 2 * D + ((D & 1) ? 4 : 2); }
template <bool OTHER> constexpr int fc_doubles() { return OTHER ? FC_SIZE : FC_N0P; }   // padded to keep 16-B alignment

// PF: request a pass's corner rows one pass ahead (first pass: before the exponential map).  While the inputs of a problem
// stay in the 256 MiB Infinity Cache from one launch to the next (up to ~46 000 frames x 144 corners), HBM sees a pure
// write stream and the plain form is 1-2 % faster; beyond that the 8 % of reads queue behind a saturated write stream
// and their latency, not bandwidth, sets the pace (tools/eval_cliff.py: 40 000 frames run at 6.4 TB/s with warm inputs
// and at 5.0 TB/s when the inputs are evicted between launches) - there PF hides one memory latency per pass:
// 50 000 frames 5.4 -> 6.0 TB/s.
// AL: the Huber corrector applied to r and J (a compile-time fact: as a run-time flag the compiler multiplied every entry by a
// weight of 1.0 when it was off - 2 D + 2 FP64 multiplies per corner on the benchmark's path).
// The J tile of a FULL pass (64 corners) leaves with a compile-time number of stores (TW / 2 per lane, unrolled): with a run-time
// trip count the compiler cannot count the stores in flight and drains them all (s_waitcnt vmcnt(0)) before the next pass may
// use its - long since arrived - corner rows; counted, the next pass computes under the stores of this one (what holds the wide
// blocks back is wavefront-level concurrency).
// (A launch's time is T(frames) = 5.3 us + 4.7 us per 1 000 frames - 6.96 TB/s asymptotically, 6.3 at 10 000 frames.  One wavefront
//  per PASS instead of per frame, to shorten the tail, was built and measured in round 5: it loses 6-22 % - every pass then walks the
//  list -> slot -> pose -> exponential-map chain.  Persistent wavefronts striding over frames: -4 %.  Two / four frames per wavefront
//  as one corner stream (lane utilisation 75 % -> 90 / 100 %): -6 / -11 %.  Many short wavefronts are what this chip wants; EXPERIMENTS.md.)
template <int MODEL, bool OF, bool OTHER, bool PF, bool AL>
__global__ __launch_bounds__(64 * CCAL_EVAL_WPB) void k_eval(const KArgs a) {
    constexpr int D = block_dim(MODEL, OF, OTHER);
    constexpr int TW = 2 * D;            // doubles per block Jacobian
    // padded LDS row stride: even (16-B aligned rows) with TS / 2 odd, so that the ds_write_b128 of 16 consecutive
    // lanes land in 16 different bank quads.  TW + 2 does that only for even D: with D = 15 (OPENCV5) the stride was
    // 32 doubles = every lane on the same banks (79 us instead of 64 for 10 000 frames)
    constexpr int TS = eval_tile_stride(D);
    constexpr int FCN = fc_doubles<OTHER>();
    constexpr int WS = FCN + 64 * TS;    // doubles of LDS per wave
    extern __shared__ double smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double* fc = smem + wave * WS;
    double* tile = fc + FCN;
    const int wstride = CCAL_EVAL_PERSIST > 0 ? (int)gridDim.x * CCAL_EVAL_WPB : a.n_list;
    for (int widx = blockIdx.x * CCAL_EVAL_WPB + wave; widx < a.n_list; widx += wstride) {   // no workgroup barrier inside
    const int o = __builtin_amdgcn_readfirstlane(a.list[widx]);
    const int slot = __builtin_amdgcn_readfirstlane(a.obs_slot[o]);
    const int64_t start = a.obs_off[o];
    const int n = (int)(a.obs_off[o + 1] - start);
    const double* th_g = a.intr + a.cam * CCAL_PMAX;
    double th[th_len<MODEL>()];
    load_theta<MODEL, OF>(th_g, a.rt, th);

    // software prefetch: the first pass's corner rows are requested before the (latency-bound) exponential map below,
    // every later pass one pass ahead - one exposed memory latency per frame instead of one per pass (it is the read
    // latency under a saturated write stream that limits this kernel once the inputs no longer sit in the Infinity Cache)
    float pX = 0.f, pY = 0.f, pZ = 0.f, pU = 0.f, pV = 0.f;
    if constexpr (PF) {
        if (n > 0) {                      // an empty last frame has start == n_corners: nothing to read there
            const int64_t g0 = start + (lane < n ? lane : 0);
            pX = a.x[g0]; pY = a.y[g0]; pZ = a.z[g0]; pU = a.u[g0]; pV = a.v[g0];
        }
    }
    {   // frame constants -> LDS (every lane computes, lane 0 stores)
        double pose[6], ex[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) pose[i] = a.poses[(int64_t)slot * 6 + i];
        if constexpr (OTHER) {
#pragma unroll
            for (int i = 0; i < 6; ++i) ex[i] = a.extr[a.cam * 6 + i];
        }
        double fcr[OTHER ? FC_SIZE : FC_N0];
        frame_setup<OTHER>(pose, ex, fcr);
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < (OTHER ? FC_SIZE : FC_N0); ++i) fc[i] = fcr[i];
        }
    }
    wave_lds_sync();

    const int64_t jbase = a.joff[o];
    // One pass = 64 corners.  FULL passes (all 64 lanes hold a corner) run in a loop of their own whose every memory operation is
    // unconditional and counted at compile time - r as one 16-byte store, the J tile as TW / 2 stores per lane - so that the
    // compiler's wait for the NEXT pass's corner rows (PF: requested before this pass's stores) is s_waitcnt vmcnt(<stores since>)
    // and not vmcnt(0): the next pass computes under this pass's stores.  The frame's last, partial pass follows the loop.
    auto pass = [&](auto full_tag, const int base) {
        constexpr bool FULL = decltype(full_tag)::value;
        const int c = base + lane;
        const bool valid = FULL || c < n;
        const int64_t g = start + (valid ? c : 0);
        double X, Y, Z, uo, vo;
        if constexpr (PF) {
            X = pX; Y = pY; Z = pZ; uo = pU; vo = pV;
            if (base + 64 < n) {          // wave-uniform: the next pass's rows are in flight while this one computes and stores
                const int cn = base + 64 + lane;
                const int64_t gn = start + (cn < n ? cn : 0);
                pX = a.x[gn]; pY = a.y[gn]; pZ = a.z[gn]; pU = a.u[gn]; pV = a.v[gn];
            }
        } else {
            X = a.x[g]; Y = a.y[g]; Z = a.z[g]; uo = a.u[g]; vo = a.v[g];
        }
        double ru, rv, J[TW];
        double* Ju = J;
        double* Jv = J + D;
#if defined(CCAL_EVAL_DIAG) && CCAL_EVAL_DIAG == 2       // diagnostic build: no projection / chain rule - the store path alone
        ru = X + uo; rv = Y + vo;
#pragma unroll
        for (int i = 0; i < D; ++i) { Ju[i] = X * (double)(i + 1) + fc[0]; Jv[i] = Z * (double)(i + 1) + vo; }
#else
        corner_block<MODEL, OF, OTHER>(th, fc, X, Y, Z, uo, vo, ru, rv, Ju, Jv);
#endif
        if constexpr (AL) {
            const double sw = huber_sqrt_weight(ru * ru + rv * rv, a.huber_delta);
            ru *= sw; rv *= sw;
#pragma unroll
            for (int i = 0; i < D; ++i) { Ju[i] *= sw; Jv[i] *= sw; }
        }
#if CCAL_EVAL_NTR
        if (valid) { __builtin_nontemporal_store(ru, a.r_out + 2 * g); __builtin_nontemporal_store(rv, a.r_out + 2 * g + 1); }
#else
        if (valid) *reinterpret_cast<double2*>(a.r_out + 2 * g) = make_double2(ru, rv);
#endif
        // block Jacobian [Ju | Jv] -> LDS row of this lane
        double* row = tile + lane * TS;
#pragma unroll
        for (int i = 0; i < D; ++i) *reinterpret_cast<double2*>(row + 2 * i) = make_double2(J[2 * i], J[2 * i + 1]);
        if constexpr (MODEL == kOCV5) {
            // the block's columns follow the CALLER's parameter order: under a non-default ccal_model_conventions.ocv5_order
            // the five distortion columns of both rows move (wave-uniform branch, never taken with OpenCV's own order)
            if (a.rt.ocv5_perm != kOcv5IdentityPerm) {
                constexpr int D0 = OF ? 3 : 4;
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    const int at = D0 + ocv5_pos(a.rt.ocv5_perm, i);
                    row[at] = J[D0 + i]; row[D + at] = J[D + D0 + i];
                }
            }
        }
        wave_lds_sync();
        // contiguous tile of J_out: 16 B per lane, 1 KiB per wave-instruction
        double* dst = a.J_out + jbase + (int64_t)base * TW;
        auto put = [&](const int e) {
            const int cr = e / TW;
            const int k = e - cr * TW;
            const double2 val = *reinterpret_cast<const double2*>(tile + cr * TS + k);
#if defined(CCAL_EVAL_DIAG) && CCAL_EVAL_DIAG == 1       // diagnostic build: the computation and the LDS round trip alone (one store per pass keeps them alive)
            if (e != lane * 2 || val.x != 12345.678) return;
#endif
#if CCAL_EVAL_NT
            __builtin_nontemporal_store(val.x, dst + e);
            __builtin_nontemporal_store(val.y, dst + e + 1);
#else
            *reinterpret_cast<double2*>(dst + e) = val;
#endif
        };
        if constexpr (FULL) {
#pragma unroll
            for (int it = 0; it < TW / 2; ++it) put(lane * 2 + 128 * it);
        } else {
            const int tot = (n - base) * TW;
            for (int e = lane * 2; e < tot; e += 128) put(e);
        }
        wave_lds_sync();
    };
    const int n_full = n & ~63;
    for (int base = 0; base < n_full; base += 64) pass(std::true_type{}, base);
    if (n_full < n) pass(std::false_type{}, n_full);
    }
}

// Euclidean reprojection error per corner with the current parameters (no Jacobian).
template <int MODEL, bool OF, bool OTHER>
__global__ __launch_bounds__(256) void k_reproj_err(const KArgs a) {
    constexpr int FCN = fc_doubles<OTHER>();
    __shared__ double smem[WAVES_PER_BLOCK * FCN];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int widx = blockIdx.x * WAVES_PER_BLOCK + wave;
    if (widx >= a.n_list) return;
    double* fc = smem + wave * FCN;
    const int o = __builtin_amdgcn_readfirstlane(a.list[widx]);
    const int slot = __builtin_amdgcn_readfirstlane(a.obs_slot[o]);
    const int64_t start = a.obs_off[o];
    const int n = (int)(a.obs_off[o + 1] - start);
    const double* th_g = a.intr + a.cam * CCAL_PMAX;
    double th[th_len<MODEL>()];
    load_theta<MODEL, OF>(th_g, a.rt, th);
    {
        double pose[6], ex[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) pose[i] = a.poses[(int64_t)slot * 6 + i];
        if constexpr (OTHER) {
#pragma unroll
            for (int i = 0; i < 6; ++i) ex[i] = a.extr[a.cam * 6 + i];
        }
        double fcr[OTHER ? FC_SIZE : FC_N0];
        frame_setup<OTHER>(pose, ex, fcr);
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 12; ++i) fc[i] = fcr[i];
        }
    }
    wave_lds_sync();
    for (int c = lane; c < n; c += 64) {
        const int64_t g = start + c;
        double px, py, pz, u, v;
        transform_point(fc, a.x[g], a.y[g], a.z[g], px, py, pz);
        project_uv<MODEL>(th, px, py, pz, u, v);
        const double du = u - (double)a.u[g], dv = v - (double)a.v[g];
        a.err_out[g] = sqrt(du * du + dv * dv);
    }
}

template <int MODEL, bool OF, bool OTHER, bool PF, bool AL>
static hipError_t launch_eval_pf(const KArgs& a, hipStream_t s) {
    constexpr int D = block_dim(MODEL, OF, OTHER);
    constexpr int WS = fc_doubles<OTHER>() + 64 * eval_tile_stride(D);
    const size_t lds = sizeof(double) * WS * CCAL_EVAL_WPB;
    int blocks = (a.n_list + CCAL_EVAL_WPB - 1) / CCAL_EVAL_WPB;
    if (blocks == 0) return hipSuccess;
    if (CCAL_EVAL_PERSIST > 0) blocks = std::min(blocks, 256 * CCAL_EVAL_PERSIST);
    static DynLdsGuard lds_guard;
    if (hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(&k_eval<MODEL, OF, OTHER, PF, AL>), lds, lds_guard); e != hipSuccess) return e;
    hipLaunchKernelGGL((k_eval<MODEL, OF, OTHER, PF, AL>), dim3(blocks), dim3(64 * CCAL_EVAL_WPB), lds, s, a);
    return hipGetLastError();
}
// blocks for which the prefetching form wins whatever the input size (measured; 0 = none yet: CCAL_EVAL_PF_MIN_D)
#ifndef CCAL_EVAL_PF_MIN_D
#define CCAL_EVAL_PF_MIN_D 99
#endif
template <int MODEL, bool OF, bool OTHER> constexpr bool eval_prefetch_always() { return block_dim(MODEL, OF, OTHER) >= CCAL_EVAL_PF_MIN_D; }
template <int MODEL, bool OF, bool OTHER>
static hipError_t launch_eval_t(const KArgs& a, bool big_inputs, hipStream_t s) {
    const bool pf = CCAL_EVAL_PREFETCH < 0 ? (big_inputs || eval_prefetch_always<MODEL, OF, OTHER>()) : CCAL_EVAL_PREFETCH != 0;
    if (a.apply_loss) return pf ? launch_eval_pf<MODEL, OF, OTHER, true, true>(a, s) : launch_eval_pf<MODEL, OF, OTHER, false, true>(a, s);
    return pf ? launch_eval_pf<MODEL, OF, OTHER, true, false>(a, s) : launch_eval_pf<MODEL, OF, OTHER, false, false>(a, s);
}
template <int MODEL, bool OF, bool OTHER>
static hipError_t launch_err_t(const KArgs& a, hipStream_t s) {
    const int blocks = (a.n_list + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    if (blocks == 0) return hipSuccess;
    hipLaunchKernelGGL((k_reproj_err<MODEL, OF, OTHER>), dim3(blocks), dim3(256), 0, s, a);
    return hipGetLastError();
}

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

hipError_t launch_eval(const ccal_problem* p, int cam, const KArgs& a, hipStream_t s) {
    // corner rows (20 B each) of the whole problem beyond ~half the Infinity Cache: they will not survive the output
    // stream from one launch to the next (measured cliff: 46 000 -> 50 000 frames x 144 corners = 132 -> 144 MB)
    const bool big_inputs = p->n_corners * 20 > (int64_t)120 << 20;
    CCAL_DISPATCH(launch_eval_t, p->cams[cam].model, p->one_focal, cam > 0, a, big_inputs, s);
}
hipError_t launch_reproj_err(const ccal_problem* p, int cam, const KArgs& a, hipStream_t s) {
    CCAL_DISPATCH(launch_err_t, p->cams[cam].model, p->one_focal, cam > 0, a, s);
}

}  // namespace ccal
