// Pieces shared by the register Gram kernels (ccal_kernels_fused.hip: k_gram1v / k_gram1w; ccal_kernels_gram2.hip: k_gram2):
// the wave-level LDS hand-off, the exact elimination of one frame's pose block, and the fused tail that runs it on the
// records a Gram kernel has just left in LDS.
#pragma once
#include "ccal_device.hpp"
#include "ccal_fused.hpp"

namespace ccal {

typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void wsync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ double clampd1(double v, double lo, double hi) { return fmin(fmax(v, lo), hi); }

// Exact elimination of ONE frame's pose block, LPE lanes per frame (all wavefront lanes call it; `active` marks the lanes
// of frames that exist).  R = the frame's record in LDS: C (21, phi basis) | [B|g] (6 x K1) | A (K1 x K1) | J_l (9);
// Ym = 6 x K1 doubles of LDS behind it.  phi -> rvec map of C and [B|g], C + lambda clamp(diag C) = L L^T (every lane of the
// frame runs the same 6 x 6 factorisation), Y = L^-1 [B|g]; the slot's record pf = L (inverted diagonal) | Y | g_p | diag C.
// Returns this lane's entries e = gl + LPE q of A (accA) and of Y^T Y (accY), and whether the block was positive definite.
#ifdef CCAL_STAMPS      // diagnostic builds: the 100 MHz clock at the stations of the fused tail (tools/stamps_g2.py)
#define CCAL_TAIL_STAMP(i) do { if (dbg) dbg[i] = wall_clock64(); } while (0)
#define CCAL_TAIL_DBG_PARAM , long long* dbg = nullptr
#define CCAL_TAIL_DBG_ARG , dbg
#else
#define CCAL_TAIL_STAMP(i) do { } while (0)
#define CCAL_TAIL_DBG_PARAM
#define CCAL_TAIL_DBG_ARG
#endif
// The e-th entry i <= j of a K1 x K1 triangle, row by row: i | j << 4 | (i K1 + j) << 8 | (j K1 + i) << 16.  Y^T Y is symmetric - a
// frame's lanes compute the K1 (K1 + 1) / 2 entries of the triangle and park each on both sides (SYM; the sums are the same bits:
// fma(a, b, c) = fma(b, a, c))
template <int K1> struct TriTable {
    uint32_t ij[K1 * (K1 + 1) / 2];
    constexpr TriTable() : ij{} {
        int n = 0;
        for (int i = 0; i < K1; ++i) for (int j = i; j < K1; ++j) ij[n++] = (uint32_t)(i | j << 4 | (i * K1 + j) << 8 | (j * K1 + i) << 16);
    }
};
template <int K1> __device__ const TriTable<K1> g_tri_table = TriTable<K1>();
template <int K, int LPE, bool SYM = false>
__device__ __forceinline__ bool eliminate_frame(double* R, double* Ym, const int gl, const bool active, const double lambda,
                                                const double min_diag, const double max_diag, double* pf, const int PF,
                                                double* accA, double* accY, const uint32_t* tri_ij = nullptr CCAL_TAIL_DBG_PARAM) {
    constexpr int K1 = K + 1, NA = K1 * K1;
    constexpr int NQ = (NA + LPE - 1) / LPE;
    static_assert(K1 <= 15, "TriTable packs i, j in four bits and the positions in eight");
    bool ok = true;
    if (active) {
        double Cr[21], jl[9];
#pragma unroll
        for (int i = 0; i < 21; ++i) Cr[i] = R[i];
#pragma unroll
        for (int i = 0; i < 9; ++i) jl[i] = R[praw_jl_off(K) + i];
        phi_to_rvec_C(Cr, jl);
        CCAL_TAIL_STAMP(0);
        double L[21], dC[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            dC[i] = Cr[i * (i + 1) / 2 + i];
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                double t = Cr[i * (i + 1) / 2 + j];
                if (i == j && lambda > 0.0) t += lambda * clampd1(dC[i], min_diag, max_diag);
#pragma unroll
                for (int k = 0; k < j; ++k) t -= L[i * (i + 1) / 2 + k] * L[j * (j + 1) / 2 + k];
                if (i == j) {
                    ok = ok && (t > 0.0) && (t < 1.7e308);
                    double sq, rsq;
                    fast_sqrt_rsqrt(ok ? t : 1.0, sq, rsq);
                    L[i * (i + 1) / 2 + i] = ok ? rsq : 0.0;
                } else {
                    L[i * (i + 1) / 2 + j] = t * L[j * (j + 1) / 2 + j];
                }
            }
        }
        CCAL_TAIL_STAMP(1);
        const double* Bm = R + 21;
        if (!ok) {
            for (int e = gl; e < PF; e += LPE) pf[e] = 0.0;
            for (int e = gl; e < 6 * K1; e += LPE) Ym[e] = 0.0;
        } else {
            // a lane's columns c = gl, gl + LPE, ... side by side: the arithmetic runs unconditionally on a clamped column (two
            // forward substitutions interleave where K1 > LPE: KB4 / OPENCV5 with six lanes per frame), only the stores are guarded
            constexpr int NCQ = (K1 + LPE - 1) / LPE;
            double bcq[NCQ][6], yq[NCQ][6];
#pragma unroll
            for (int q = 0; q < NCQ; ++q) {
                const int c = gl + LPE * q, cc = c < K1 ? c : K1 - 1;
                double* bc = bcq[q];
                double* y = yq[q];
#pragma unroll
                for (int i = 0; i < 6; ++i) bc[i] = Bm[i * K1 + cc];
                phi_to_rvec_col(bc, jl);
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    double t = bc[i];
#pragma unroll
                    for (int k = 0; k < i; ++k) t -= L[i * (i + 1) / 2 + k] * y[k];
                    y[i] = t * L[i * (i + 1) / 2 + i];
                }
            }
#pragma unroll
            for (int q = 0; q < NCQ; ++q) {
                const int c = gl + LPE * q;
                if (c < K1) {
#pragma unroll
                    for (int i = 0; i < 6; ++i) { Ym[i * K1 + c] = yq[q][i]; pf[21 + i * K1 + c] = yq[q][i]; }
                    if (c == K) {                  // g_p in the rvec basis
#pragma unroll
                        for (int i = 0; i < 6; ++i) pf[21 + 6 * K1 + i] = bcq[q][i];
                    }
                }
            }
            // L and diag C: every lane holds them; lane 0 of the frame parks them in the record's C | B area (read above,
            // dead now), the frame's lanes store them coalesced after the fence below
            if (gl == 0) {
#pragma unroll
                for (int i = 0; i < 21; ++i) R[i] = L[i];
#pragma unroll
                for (int i = 0; i < 6; ++i) R[21 + i] = dC[i];
            }
        }
    }
    CCAL_TAIL_STAMP(2);
    wsync();
    if (active && ok) {
        for (int e = gl; e < 21; e += LPE) pf[e] = R[e];
        for (int e = gl; e < 6; e += LPE) pf[21 + 6 * K1 + 6 + e] = R[21 + e];
        if constexpr (SYM) {
            constexpr int NT = K1 * (K1 + 1) / 2, NQT = (NT + LPE - 1) / LPE;
#pragma unroll
            for (int q = 0; q < NQT; ++q) {
                if (gl + LPE * q < NT) {
                    const int i = tri_ij[q] & 15, j = (tri_ij[q] >> 4) & 15;
                    double t = 0.0;
#pragma unroll
                    for (int k = 0; k < 6; ++k) t += Ym[k * K1 + i] * Ym[k * K1 + j];
                    accY[q] = t;
                }
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) { const int e = gl + LPE * q; if (e < NA) accA[q] = R[21 + 6 * K1 + e]; }
        } else {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = gl + LPE * q;
            if (e < NA) {
                const int i = e / K1, j = e - i * K1;
                double t = 0.0;
#pragma unroll
                for (int k = 0; k < 6; ++k) t += Ym[k * K1 + i] * Ym[k * K1 + j];
                accY[q] = t;
                accA[q] = R[21 + 6 * K1 + e];
            }
        }
        }
    } else if (active) {
        if constexpr (SYM) {
            constexpr int NQT = (K1 * (K1 + 1) / 2 + LPE - 1) / LPE;
#pragma unroll
            for (int q = 0; q < NQT; ++q) accY[q] = 0.0;
#pragma unroll
            for (int q = 0; q < NQ; ++q) { const int e = gl + LPE * q; if (e < NA) accA[q] = R[21 + 6 * K1 + e]; }
        } else {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int e = gl + LPE * q;
                if (e < NA) { accY[q] = 0.0; accA[q] = R[21 + 6 * K1 + e]; }
            }
        }
    }
    CCAL_TAIL_STAMP(3);
    return ok;
}

// Fused elimination (single-camera loop, k_gram1w): the wavefront that built G frames' Grams eliminates their pose blocks
// itself - records in LDS at red + g GS (what k_schur1m loads from HBM), LPF lanes per frame - and writes ONE row of
// partial sums per wavefront, [A_dir | Y^T Y | model decrease | failed blocks], frames added in a fixed order, to `row`
// (the wavefront's row of FusedArgs::partial; single-launch groups: a row in LDS that the workgroup adds up).
template <int K, int LPF>
__device__ __forceinline__ void gram_fused_tail(const FusedArgs& a, const double lambda, double* red, double* row,
                                                const int grp, const int gl, const bool lane_ok, const bool active, const int slot,
                                                const int set, const double mcv CCAL_TAIL_DBG_PARAM) {
    constexpr int G = 64 / LPF, K1 = K + 1, NA = K1 * K1;
    constexpr int REC = praw_jl_off(K) + 9, GS = (REC + 6 * K1 + 1) & ~1, NQ = (NA + LPF - 1) / LPF;
    static_assert(2 * NA + 2 <= GS, "a frame's sums reuse its record row");
    double* R = red + grp * GS;
    double* Ym = R + REC;
    // the triangle of Y^T Y where that takes fewer rounds of the frame's lanes than the square (many lanes per frame: one round either way)
    constexpr int NT = K1 * (K1 + 1) / 2, NQT = (NT + LPF - 1) / LPF;
    constexpr bool SYM = NQT < NQ;
    constexpr int NQY = SYM ? NQT : NQ;
    double accA[NQ], accY[NQY];
    uint32_t tri_ij[NQY];                  // SYM: which entries of the triangle this lane takes: requested now, needed after the factorisation
#pragma unroll
    for (int q = 0; q < NQY; ++q) tri_ij[q] = SYM ? g_tri_table<K1>.ij[gl + LPF * q < NT ? gl + LPF * q : 0] : 0u;
#pragma unroll
    for (int q = 0; q < NQ; ++q) accA[q] = 0.0;
#pragma unroll
    for (int q = 0; q < NQY; ++q) accY[q] = 0.0;
    const bool ok = eliminate_frame<K, LPF, SYM>(R, Ym, gl, active, lambda, a.min_diag, a.max_diag,
                                                a.pf[set] + (int64_t)slot * a.PF, a.PF, accA, accY, tri_ij CCAL_TAIL_DBG_ARG);
    wsync();
    if (lane_ok) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) { const int e = gl + LPF * q; if (e < NA) { R[e] = accA[q]; if constexpr (!SYM) R[NA + e] = accY[q]; } }
        if constexpr (SYM) {
#pragma unroll
            for (int q = 0; q < NQY; ++q) {
                if (gl + LPF * q < NT) {
                    R[NA + ((tri_ij[q] >> 8) & 255)] = accY[q];
                    R[NA + (tri_ij[q] >> 16)] = accY[q];
                }
            }
        }
        if (gl == 0) { R[2 * NA] = active ? mcv : 0.0; R[2 * NA + 1] = (active && !ok) ? 1.0 : 0.0; }
    }
    wsync();
    CCAL_TAIL_STAMP(4);
    const int lane = threadIdx.x & 63;
    for (int e = lane; e < 2 * NA + 2; e += 64) {
        double t = 0.0;
#pragma unroll
        for (int g = 0; g < G; ++g) t += red[g * GS + e];
        row[e] = t;
    }
}


// General loop, GEN kernels with FusedArgs::gen_backsub: the candidate pose of slot `slot` from the accepted pose, the slot's
// elimination record pf = L (21, inverted diagonal) | Y (6 x K1) | g_p (6) | diag C (6) and the camera step dc (K = columns of the
// reduced system, run time) - what k_backsub did per slot in a launch of its own (ccal_kernels_normal.hip).  The frame's LPF lanes
// stage the record and the camera step in LDS together (`buf`: the wavefront's reduction buffer, REGION doubles, free until the
// corner loop has ended; G frames side by side), six of them take one row of Y dc each, then every lane finishes the 6 x 6 solve
// for itself: walking the record from global memory lane by lane cost the prologue 5.4 us (K = 18: 126 dependent-latency loads).
// Records too long for the buffer (many cameras) take that slow way.  Returns the pose block's model decrease; a slot whose
// record is empty (failed factorisation) keeps its pose.  Called by all lanes of the wavefront (wave-level hand-offs inside).
template <int LPF, int G, int REGION>
__device__ __forceinline__ double gen_backsub_pose(const FusedArgs& a, const int slot, const double lambda, double* pose, double* buf,
                                                   const int grp, const int gl, const bool lane_ok) {
    // every fused multiply-add is spelled out, here and in k_backsub, and the products that must NOT be fused into the pose update
    // pass an optimisation barrier: the two forms of the general loop give the same bits - the back end (-ffp-contract=fast) had
    // turned pose + t * L_ii into one FMA here and not there (1 ulp in a few poses per step, enough to move a verdict on a system
    // that is singular to rounding)
    const int K = a.g_K, K1 = K + 1, PF = a.g_PF;
    const int RS = (PF + K + 6 + 1) & ~1;
    const double* pfg = a.g_pf + (int64_t)slot * PF;
    double dp[6];
    const double* pf;
    if (G * RS <= REGION) {                    // wave-uniform
        double* R = buf + grp * RS;
        if (lane_ok) {
            for (int e = gl; e < PF; e += LPF) R[e] = pfg[e];
            for (int e = gl; e < K; e += LPF) R[PF + e] = a.g_dc[e];
        }
        wsync();
        if (lane_ok && gl < 6) {
            const double* yr = R + 21 + gl * K1;
            const double* dc = R + PF;
            double t = yr[K];
#pragma unroll 6
            for (int j = 0; j < K; ++j) t = __builtin_fma(yr[j], dc[j], t);       // (k_backsub's sum; unrolled: the LDS reads of six steps travel together)
            R[PF + K + gl] = -t;
        }
        wsync();
#pragma unroll
        for (int i = 0; i < 6; ++i) dp[i] = R[PF + K + i];
        pf = R;
    } else {
        const double* y0 = pfg + 21;
        double t0 = y0[K], t1 = y0[K1 + K], t2 = y0[2 * K1 + K], t3 = y0[3 * K1 + K], t4 = y0[4 * K1 + K], t5 = y0[5 * K1 + K];
        for (int j = 0; j < K; ++j) {
            const double d = a.g_dc[j];
            t0 = __builtin_fma(y0[j], d, t0); t1 = __builtin_fma(y0[K1 + j], d, t1); t2 = __builtin_fma(y0[2 * K1 + j], d, t2);
            t3 = __builtin_fma(y0[3 * K1 + j], d, t3); t4 = __builtin_fma(y0[4 * K1 + j], d, t4); t5 = __builtin_fma(y0[5 * K1 + j], d, t5);
        }
        dp[0] = -t0; dp[1] = -t1; dp[2] = -t2; dp[3] = -t3; dp[4] = -t4; dp[5] = -t5;
        pf = pfg;
    }
    double mc = 0.0;
    if (pf[0] != 0.0) {
#pragma unroll
        for (int i = 5; i >= 0; --i) {     // L^T x = rhs, diagonal stored inverted
            double t = dp[i];
#pragma unroll
            for (int k = i + 1; k < 6; ++k) t = __builtin_fma(-pf[k * (k + 1) / 2 + i], dp[k], t);
            dp[i] = t * pf[i * (i + 1) / 2 + i];
            asm volatile("" : "+v"(dp[i]));        // the product is a value of its own
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const double gp = pf[21 + 6 * K1 + i], dC = pf[21 + 6 * K1 + 6 + i];
            const double Dii = lambda > 0.0 ? lambda * clampd1(dC, a.min_diag, a.max_diag) : 0.0;
            mc = __builtin_fma(dp[i], __builtin_fma(Dii, dp[i], -gp), mc);
            pose[i] = pose[i] + dp[i];
        }
    }
    wsync();                                   // (the buffer is the caller's again)
    return mc;
}

#ifndef CCAL_GRAMV_WPB
#define CCAL_GRAMV_WPB 2          // wavefronts per workgroup of the register Gram kernels
#endif

}  // namespace ccal
