// k_schurq: the per-slot elimination of a TWO-camera rig whose cameras have the same number of free intrinsics (same model,
// same focal mode - the reference's stereo case, e.g. TUM-VI cam0 + cam1), with FOUR LANES per frame slot and everything
// about the layout known at compile time.
//
// k_schur (ccal_kernels_normal.hip) is generic in the number of cameras and their block sizes: every phase of a slot carries
// run-time index arithmetic, table look-ups, exec-mask regions and a wavefront hand-off, ~750 - 1 250 instructions per slot
// however the lanes are arranged (a 16-lanes-per-slot variant of it measured 38 us against 29.6 us: the time goes with the
// instruction count, not with the idle lanes).  Here the system is [theta_0 (PE) | theta_1 (PE) | extrinsics (6) | r], all
// loops are unrolled, every LDS address is a per-lane base plus a constant, a wavefront works on 16 slots in straight-line
// code and the four lanes of a slot share the work by "index = q (mod 4)".  ~3 500 instructions per wavefront = ~220 per slot.
//
// Sums are reproducible: a slot's contributions go to the slot's own image of the reduced system in LDS in program order,
// the 16 images of a wavefront are added in a fixed order, one row of partial sums per wavefront goes to k_reduce.
// Inputs / outputs as k_schur<true>: GEN records of the register Gram kernels (ccal_fused.hpp), slot records pf for
// k_backsub.  Reference contract: the linear solve of one tiny-solver step (src/util.rs:670).
#include <algorithm>
#include <cstdlib>

#include "ccal_gram_common.hpp"
#include "ccal_normal.hpp"

namespace ccal {

// Slots per wavefront: 16 with FOUR lanes each or 8 with EIGHT lanes each (LQ = 64 / SLOTS lanes share a slot's work by
// "index = q (mod LQ)").  The kernel is a latency chain - one wavefront per SIMD (340 registers), ~3 300 straight-line
// instructions at ~11 cycles each, measured: halving the LDS bank conflicts (43 % -> 11 % of the LDS cycles) did not move its
// 20 us, and half-empty wavefronts of 8 four-lane slots take the same 17 us each (and two rounds: 31 us).  Eight lanes per
// slot shorten the chain itself: every lane carries half the rows / columns.  CCAL_SCHURQ_SLOTS=8|16 picks the form.

// entry e of a row-major lower triangle -> (i, j)
__device__ __forceinline__ void sq_tri_decode(int e, int& i, int& j) {
    i = (int)((__builtin_sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
    if ((i + 1) * (i + 2) / 2 <= e) ++i;
    if (i * (i + 1) / 2 > e) --i;
    j = e - i * (i + 1) / 2;
}
__host__ __device__ constexpr int sq_tri(int i, int j) { return i * (i + 1) / 2 + j; }

// per-slot LDS area (doubles)
template <int PE, int LQ> struct SqLayout {
    static constexpr int K1c = PE + 1, K = 2 * PE + 6, K1 = K + 1, NT = K1 * (K1 + 1) / 2;
    static constexpr int XH = NT, XG = NT + K, XC = NT + 2 * K, XM = XC + 1, XF = XC + 2, ACCN = XC + 3;
    static constexpr int HS = 36 + 6 * K1c;                              // record head: C | [B|g]^T
    static constexpr int H0 = 0, E0 = HS, H1 = HS + 36, E1 = 2 * HS + 36;   // staging of the two records (E as stored: [M | N | J1 | SJ], ccal_fused.hpp)
    static constexpr int STG = 2 * HS + 72;
    static constexpr int IMG = 0;                                        // the slot's image of the reduced system: over the (dead) staging
    static constexpr int ISH = 4;                                        // a slot's image starts 0 or ISH doubles into its area (sq_img_shift)
    static constexpr int YL = ((STG > ACCN + ISH ? STG : ACCN + ISH) + 1) & ~1;       // [B|g] of the slot, then Y: K1 columns of 6
    static constexpr int CX = YL + 6 * K1;                               // C of the slot (6 x 6)
    static constexpr int DUM = (CX + 36 + 15) & ~15;                     // where stores nobody wants go: see dump() in the kernel
    static constexpr int SS0 = DUM + 16 + LQ + ISH;
    // Slot stride = 8 (mod 16) doubles.  What decides: the 16-byte stores and loads whose four lanes of a slot touch consecutive
    // (staging, zeroing) or 48-byte-strided (rows of [B|g]^T, columns of Y) pieces - ds_write_b128 works in groups of 8 lanes = 2
    // slots over 32 banks, ds_read_b128 in groups of 4 slots over 64: with the stride = 2 (mod 4) of round 3 the second slot of
    // a group sat 16 bytes beside the first and three of its four pieces collided (model of every access of the kernel,
    // tools/lds_bank_model.py: 46 % of the LDS cycles were conflicts, measured 42 %); at 64 bytes (mod 128) the pieces of a
    // group tile the banks.  Broadcast reads (one address per slot) stay conflict-free.  CCAL_SQ_SS_OLD: the old stride (A/B)
#ifdef CCAL_SQ_SS_OLD
    static constexpr int SS = SS0 + ((2 - SS0 % 4) + 4) % 4;
#else
    static constexpr int SS = SS0 + ((8 - SS0 % 16) + 16) % 16;
#endif
};

// Where a slot's image of the reduced system starts inside its area.  Every access of the image by the four lanes of a slot
// is ROW-FIXED, COLUMN-CONSECUTIVE (the lanes own columns j = q + 4 t): four consecutive doubles per slot.  With the slot
// stride = 8 (mod 16) the slots of an 8-byte store group (4 slots, 32 banks) sit at 0, 8, 0, 8 and those of a load group (8
// slots, 64 banks) at 0, 8, 16, 24, 0, 8, 16, 24 doubles: this shift makes them 0, 8, 4, 12 and 0, 8, 20, 28, 4, 12, 16, 24 -
// the groups tile the banks whatever the (common) offset inside the image.
__host__ __device__ constexpr int sq_img_shift(int sl) { return 4 * (((sl >> 1) ^ (sl >> 2)) & 1); }
#ifdef CCAL_STAMPS      // diagnostic build: 100 MHz clock at the phase boundaries, parked behind the partial sums (tools/stamps_sq.py)
#define SQ_STAMP(i) do { sq_stamps[i] = wall_clock64(); } while (0)
#else
#define SQ_STAMP(i) do { } while (0)
#endif

// (eight lanes per slot: 1 250 wavefronts at 10 000 slots - two per SIMD, i.e. at most 256 registers, or half of them wait for a
// second round)
template <int PE, int SLOTS>
__global__ __launch_bounds__(64, SLOTS == 8 ? 2 : 1) void k_schurq(const SchurArgs a) {
    static_assert(SLOTS == 8 || SLOTS == 16, "8 or 16 slots per wavefront");
    constexpr int LQ = 64 / SLOTS;                          // lanes per slot
#ifdef CCAL_STAMPS
    long long sq_stamps[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
#endif
    SQ_STAMP(0);
    using Lt = SqLayout<PE, LQ>;
    constexpr int K1c = Lt::K1c, K = Lt::K, K1 = Lt::K1, NT = Lt::NT, HS = Lt::HS, ACCN = Lt::ACCN;
    constexpr int CT0 = 0, CT1 = PE, CE = 2 * PE;
    constexpr int TR = (K1c + LQ - 1) / LQ;                       // rows of a record's [B|g]^T per lane
    constexpr int TY = (K1 + LQ - 1) / LQ;                        // columns / rows of the slot's Y per lane
    extern __shared__ __attribute__((aligned(16))) double smem[];
    if (a.st && a.st->done) return;
    const int g_set = a.st ? schur_set(a.st) : 0;
    // (an explicit global-address-space pointer: picked out of the argument block's array the compiler cannot prove where it
    // points and emits flat_load, which also counts against the LDS counter)
    typedef const __attribute__((address_space(1))) double* gptr_t;
    const gptr_t p_G = (gptr_t)a.Gs[g_set];
    const double lambda = a.st ? schur_lambda(a.st) : a.lambda;
    const int lane = threadIdx.x, sl = lane / LQ, q = lane % LQ;
    const int s = blockIdx.x * SLOTS + sl;
    const bool has = s < a.n_slots;
    double* sb = smem + sl * Lt::SS;
    // Stores of the lanes that have no entry in the row at hand (right of the diagonal) go to a dump - branch-free - and the
    // dump address continues the valid lanes' pattern: "address = X + q (mod 16 doubles)" for all four lanes of the slot, so the
    // store group tiles the banks whether a lane's value is wanted or not (a fixed dump address per lane collided with the valid
    // lanes of other slots: the rest of the bank conflicts in the model, tools/lds_bank_model.py)
    double* dq = sb + Lt::DUM + sq_img_shift(sl) + q;
    auto dump = [&](const int X) { return dq + (X & 15); };

    // ---- the two records of the slot -> LDS (16 bytes per lane and load).  They sit side by side at (2 s + camera) x record size
    // (normal_ws_ensure): requested at once, no look-up first; a missing record is a hole of zeros.  The table only says
    // whether the slot has an observation at all (wanted much later)
    const int64_t d0 = has ? 2 * (int64_t)s * gen_rec_size(PE) : 0, d1 = d0 + (has ? gen_rec_size(PE) : 0);
    const int64_t p0 = has ? a.slot_desc[2 * (int64_t)s] : -1, p1 = has ? a.slot_desc[2 * (int64_t)s + 1] : -1;
    const bool live = p0 >= 0 || p1 >= 0;
    const double mc_s = (has && q == 0) ? a.mc_slot[s] : 0.0;          // model decrease of this slot's pose block for the step under decision
    // the camera | r blocks of the records (direct terms of the reduced system, wanted only once the products are done): this
    // lane's COLUMNS j = q + 4 t of the lower triangles (rows i >= j), held in registers from the staging on
    double av[2][TR][K1c];
    const gptr_t r0 = p_G + d0;
    const gptr_t r1 = p_G + d1;
    // Four lanes per slot: requested with the staging, as 8-byte loads straight into registers they arrive while the products
    // run (through 16-byte pieces into LDS they sat on the staging's critical path: 11.4 us against 8.4).  Eight lanes per slot
    // (two wavefronts per SIMD: 256 registers): requested AFTER the products, they arrive behind the factorisation - held across
    // the products they cost 130 bytes of scratch
    constexpr bool AV_LATE = LQ == 8;
    auto load_av = [&]() {
#pragma unroll
        for (int t = 0; t < TR; ++t) {
            const int j = q + LQ * t;
#pragma unroll
            for (int i = 0; i < K1c; ++i) {
                if (i < LQ * t) { av[0][t][i] = 0.0; av[1][t][i] = 0.0; continue; }          // compile time: j <= i impossible
                const bool in = j < K1c && j <= i;
                av[0][t][i] = (in && has) ? r0[HS + i * (i + 1) / 2 + j] : 0.0;      // packed lower triangle, entry (i, j)
                av[1][t][i] = (in && has) ? r1[HS + i * (i + 1) / 2 + j] : 0.0;
            }
        }
    };
    {
        constexpr int NH = HS / 2;                                      // 16-byte pieces of a head
        constexpr int TH = (NH + LQ - 1) / LQ, TE = (18 + LQ - 1) / LQ;
        typedef double dv2 __attribute__((ext_vector_type(2)));                       // (a plain vector: HIP's double2 class has no copy from address space 1)
        typedef const __attribute__((address_space(1))) dv2* gptr2_t;
        const gptr2_t g0 = (gptr2_t)r0;
        const gptr2_t g1 = (gptr2_t)r1;
        double2* w = reinterpret_cast<double2*>(sb);
        {
            double2 h0[TH], e0[TE];
#pragma unroll
            // (always a load from a valid address, then a select of the VALUE, component by component: `cond ? g0[c] : z2` is
            // compiled as a select of POINTERS into two address spaces - flat_load plus a scratch slot for the zero)
            for (int t = 0; t < TH; ++t) { const int c = q + LQ * t; const dv2 v = g0[c < NH ? c : NH - 1]; const bool in = has && c < NH; h0[t] = make_double2(in ? v.x : 0.0, in ? v.y : 0.0); }
#pragma unroll
            for (int t = 0; t < TE; ++t) { const int c = q + LQ * t; const dv2 v = g0[gen_e_off(PE) / 2 + (c < 18 ? c : 17)]; const bool in = has && c < 18; e0[t] = make_double2(in ? v.x : 0.0, in ? v.y : 0.0); }
#pragma unroll
            for (int t = 0; t < TH; ++t) { const int c = q + LQ * t; if (c < NH) w[Lt::H0 / 2 + c] = h0[t]; }
#pragma unroll
            for (int t = 0; t < TE; ++t) { const int c = q + LQ * t; if (c < 18) w[Lt::E0 / 2 + c] = e0[t]; }
        }
        {
            double2 h1[TH], e1[TE];
#pragma unroll
            for (int t = 0; t < TH; ++t) { const int c = q + LQ * t; const dv2 v = g1[c < NH ? c : NH - 1]; const bool in = has && c < NH; h1[t] = make_double2(in ? v.x : 0.0, in ? v.y : 0.0); }
#pragma unroll
            for (int t = 0; t < TE; ++t) { const int c = q + LQ * t; const dv2 v = g1[gen_e_off(PE) / 2 + (c < 18 ? c : 17)]; const bool in = has && c < 18; e1[t] = make_double2(in ? v.x : 0.0, in ? v.y : 0.0); }
#pragma unroll
            for (int t = 0; t < TH; ++t) { const int c = q + LQ * t; if (c < NH) w[Lt::H1 / 2 + c] = h1[t]; }
#pragma unroll
            for (int t = 0; t < TE; ++t) { const int c = q + LQ * t; if (c < 18) w[Lt::E1 / 2 + c] = e1[t]; }
        }
        if constexpr (!AV_LATE) load_av();
    }
    wsync();
    SQ_STAMP(1);

    // ---- products with E (frame_setup_composed): E_p = diag(M, N) maps the composed pose's (phi, delta) to (rvec_0_b, tvec_0_b),
    // E_x (camera 1) to (rvec_1_0, tvec_1_0).  Stored as [M | N | J1 | SJ], X(m, b) at 3 b + m (ccal_fused.hpp)
    double racc[6] = { 0, 0, 0, 0, 0, 0 };                  // [B|g] column r: both cameras' row r meet in the same lane
    double cmb[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };          // this lane's 3 x 3 block (rh, ch) of the slot's C
    double cross[TR][6];                                    // camera 1: rows of [B|g]^T times E_x -> theta_1 (| r) x extrinsics
    constexpr int TX = (6 + LQ - 1) / LQ;                   // extrinsic columns per lane
    double xx[TX][6];                                        // E_x^T (C E_x) columns j = q, q + 4
    const int rh = (q & 3) >> 1, ch = q & 1;          // (eight lanes per slot: lanes q and q + 4 do the same block)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const double* H = sb + (c == 0 ? Lt::H0 : Lt::H1);
        const double* Ec = sb + (c == 0 ? Lt::E0 : Lt::E1);        // [M | N | J1 | SJ], X(m, b) at 3 b + m
        // E_p^T rows 0..5 in registers: row b = [M(., b) | 0], row 3 + b = [0 | N(., b)] - only the non-zero halves are used
        double ep[6][6];
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                ep[b][m] = Ec[3 * b + m]; ep[b][3 + m] = 0.0;
                ep[3 + b][m] = 0.0; ep[3 + b][3 + m] = Ec[9 + 3 * b + m];
            }
        {   // block (rh, ch) of E_p^T C E_p = P_rh^T C_(rh,ch) P_ch,  P_0 = M, P_1 = N
            double cb[3][3], pr[3][3], pc[3][3], tm[3][3];
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int n = 0; n < 3; ++n) {
                    cb[m][n] = H[(3 * rh + m) * 6 + 3 * ch + n];
                    pr[m][n] = Ec[9 * rh + 3 * n + m];        // P_rh(m, n)
                    pc[m][n] = Ec[9 * ch + 3 * n + m];
                }
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int n = 0; n < 3; ++n) tm[m][n] = cb[m][0] * pc[0][n] + cb[m][1] * pc[1][n] + cb[m][2] * pc[2][n];
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int n = 0; n < 3; ++n) cmb[3 * m + n] += pr[0][m] * tm[0][n] + pr[1][m] * tm[1][n] + pr[2][m] * tm[2][n];
        }
        double ex[6][6];                                    // camera 1: E_x^T rows j (E^T rows 6 + j)
        if (c == 1) {
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    ex[j][k] = Ec[18 + 3 * j + k]; ex[j][3 + k] = Ec[27 + 3 * j + k];         // rvec_1_0: [J1(., j) | SJ(., j)]
                    ex[3 + j][k] = 0.0; ex[3 + j][3 + k] = k == j ? 1.0 : 0.0;                // tvec_1_0: unit vectors (folded by the compiler)
                }
        }
        // rows i = q + 4 t of [B|g]^T: E_p^T b_i -> column theta_c,i of the slot's [B|g] (i = PE: the r column)
#pragma unroll
        for (int t = 0; t < TR; ++t) {
            const int i = q + LQ * t, ic = i < K1c ? i : 0;
            const double2* r = reinterpret_cast<const double2*>(H + 36 + 6 * ic);
            const double2 r0 = r[0], r1 = r[1], r2 = r[2];
            const double bi[6] = { r0.x, r0.y, r1.x, r1.y, r2.x, r2.y };
            double y[6];
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                y[b] = ep[b][0] * bi[0] + ep[b][1] * bi[1] + ep[b][2] * bi[2];
                y[3 + b] = ep[3 + b][3] * bi[3] + ep[3 + b][4] * bi[4] + ep[3 + b][5] * bi[5];
            }
            const bool is_r = i == PE;
#pragma unroll
            for (int k = 0; k < 6; ++k) racc[k] += is_r ? y[k] : 0.0;
            if (i < PE) {
                double2* yo = reinterpret_cast<double2*>(sb + Lt::YL + 6 * ((c == 0 ? CT0 : CT1) + i));
                yo[0] = double2{ y[0], y[1] }; yo[1] = double2{ y[2], y[3] }; yo[2] = double2{ y[4], y[5] };
            }
            if (c == 1) {
#pragma unroll
                for (int j = 0; j < 6; ++j)
                    cross[t][j] = ((ex[j][0] * bi[0] + ex[j][1] * bi[1]) + (ex[j][2] * bi[2] + ex[j][3] * bi[3])) + (ex[j][4] * bi[4] + ex[j][5] * bi[5]);
            }
        }
        if (c == 1) {
            // columns j = q, q + 4 of E_x: z = C E_x[:, j];  E_p^T z -> extrinsics column of [B|g];  E_x^T z -> extrinsics block
            double z[TX][6];
            double exj[TX][6];
#pragma unroll
            for (int t = 0; t < TX; ++t) {
                const int j = q + LQ * t, jr = j < 3 ? j : 0;             // column j of E_x: rotation columns stored, translation columns unit
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    exj[t][k] = j < 3 ? Ec[18 + 3 * jr + k] : 0.0;
                    exj[t][3 + k] = j < 3 ? Ec[27 + 3 * jr + k] : (j - 3 == k ? 1.0 : 0.0);
                }
            }
#pragma unroll
            for (int m = 0; m < 6; ++m) {
                const double2* r = reinterpret_cast<const double2*>(H + 6 * m);
                const double2 r0 = r[0], r1 = r[1], r2 = r[2];
#pragma unroll
                for (int t = 0; t < TX; ++t)
                    z[t][m] = ((r0.x * exj[t][0] + r0.y * exj[t][1]) + (r1.x * exj[t][2] + r1.y * exj[t][3])) + (r2.x * exj[t][4] + r2.y * exj[t][5]);
            }
#pragma unroll
            for (int t = 0; t < TX; ++t) {
                const int j = q + LQ * t;
                double y[6];
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    y[b] = ep[b][0] * z[t][0] + ep[b][1] * z[t][1] + ep[b][2] * z[t][2];
                    y[3 + b] = ep[3 + b][3] * z[t][3] + ep[3 + b][4] * z[t][4] + ep[3 + b][5] * z[t][5];
                }
                if (j < 6) {
                    double2* yo = reinterpret_cast<double2*>(sb + Lt::YL + 6 * (CE + j));
                    yo[0] = double2{ y[0], y[1] }; yo[1] = double2{ y[2], y[3] }; yo[2] = double2{ y[4], y[5] };
                }
#pragma unroll
                for (int jp = 0; jp < 6; ++jp)
                    xx[t][jp] = ((ex[jp][0] * z[t][0] + ex[jp][1] * z[t][1]) + (ex[jp][2] * z[t][2] + ex[jp][3] * z[t][3])) + (ex[jp][4] * z[t][4] + ex[jp][5] * z[t][5]);
            }
        }
    }
    if (q == (PE % LQ)) {                                    // the lane that had row r of both records
        double2* yo = reinterpret_cast<double2*>(sb + Lt::YL + 6 * K);
        yo[0] = double2{ racc[0], racc[1] }; yo[1] = double2{ racc[2], racc[3] }; yo[2] = double2{ racc[4], racc[5] };
    }
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) sb[Lt::CX + (3 * rh + m) * 6 + 3 * ch + n] = cmb[3 * m + n];
    if constexpr (AV_LATE) load_av();
    wsync();
    SQ_STAMP(2);

    // ---- C + lambda clamp(diag C) = L L^T: the four lanes of a slot run the same 6 x 6 factorisation
    double L[21], dC[6];
    bool ok = true;
    {
        double cm[21];
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) cm[i * (i + 1) / 2 + j] = sb[Lt::CX + i * 6 + j];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            dC[i] = cm[i * (i + 1) / 2 + i];
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                double t = cm[i * (i + 1) / 2 + j];
                if (i == j && lambda > 0.0) t += lambda * clampd1(dC[i], a.min_diag, a.max_diag);
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
    }
    const bool go = live && ok;
    SQ_STAMP(3);
    double* pf = a.pf + (int64_t)(has ? s : 0) * a.PF;
    double* img = sb + Lt::IMG + sq_img_shift(sl);
    // ---- the slot's image of the reduced system (over the staging, dead now): the direct terms.  The image is NOT cleared
    // first (round 4): every entry of the triangle is written by the Y^T Y pass below, which knows - at compile time per row and
    // column block - where no direct term was stored (theta_1 x theta_0, extrinsics x theta_0: camera 0 has no share in camera 1's
    // blocks) and starts from 0 there instead of reading; every extra (hdiag, g_c, cost, model decrease, failed) has its writer
    if (q == 0) {
        img[Lt::XM] = has ? mc_s : 0.0;
        img[Lt::XF] = (live && !ok) ? 1.0 : 0.0;             // failed pose block (all-reduced with the sums: every rank sees it)
    }
    // Direct terms: every entry has ONE writer (plain stores over the zeros), except (r, r) = cost, where the two records'
    // entries meet in the same lane.  camera | r blocks of the records, row by row (the row is compile time, the lanes hold its
    // columns j = q + 4 t: consecutive addresses): + hdiag, g_c, cost
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int ct = c == 0 ? CT0 : CT1;
#pragma unroll
        for (int i = 0; i < K1c; ++i) {
            const int ii = i < PE ? ct + i : K;              // the image's row (compile time)
#pragma unroll
            for (int t = 0; t < TR; ++t) {
                if (LQ * t > i) continue;                    // compile time: j <= i impossible
                const int j = q + LQ * t;
                const double v = av[c][t][i];
                const bool in = j <= i && j < PE;            // (j == PE: the residual column - only (r, r), below)
                *(in ? img + sq_tri(ii, 0) + ct + j : dump(sq_tri(ii, 0) + ct + LQ * t)) = v;
                if (i < PE) { if (i / LQ == t) *((in && j == i) ? img + Lt::XH + ii : dump(Lt::XH + ii - (i % LQ))) = v; }      // hdiag
                else *(in ? img + Lt::XG + ct + j : dump(Lt::XG + ct + LQ * t)) = v;                                          // g_c
            }
        }
    }
    {
        constexpr int t = PE / LQ;                            // the lane whose column is the residual's: j == PE
        const int j = q + LQ * t;
        const double v = av[0][t][PE] + av[1][t][PE];
        *(j == PE ? img + sq_tri(K, K) : dump(sq_tri(K, K) - (PE % LQ))) = v;
        *(j == PE ? img + Lt::XC : dump(Lt::XC - (PE % LQ))) = v;
    }
    // camera 1: theta_1 (| r) x extrinsics, extrinsics x extrinsics
#pragma unroll
    for (int t = 0; t < TR; ++t) {
        const int i = q + LQ * t;
        const int col = i < PE ? CT1 + i : K;                // i == PE: row r of the system (g_c of the extrinsic columns)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int hi = i < PE ? CE + j : K, lo = i < PE ? col : CE + j;
            *(i < K1c ? img + hi * (hi + 1) / 2 + lo : dump(sq_tri(CE + j, 0) + CT1 + LQ * t)) = cross[t][j];
            if (PE / LQ == t) *(i == PE ? img + Lt::XG + CE + j : dump(Lt::XG + CE + j - (PE % LQ))) = cross[t][j];
        }
    }
#pragma unroll
    for (int t = 0; t < TX; ++t) {
        const int j = q + LQ * t;
#pragma unroll
        for (int jp = 0; jp < 6; ++jp) {
            if (jp < LQ * t) continue;                        // compile time: jp >= j impossible
            const bool in = j < 6 && jp >= j;
            *(in ? img + sq_tri(CE + jp, 0) + CE + j : dump(sq_tri(CE + jp, 0) + CE + LQ * t)) = xx[t][jp];
            if (jp <= LQ * t + LQ - 1) *((in && jp == j) ? img + Lt::XH + CE + j : dump(Lt::XH + CE + jp - (jp % LQ))) = xx[t][jp];
        }
    }
    SQ_STAMP(4);
    // ---- Y = L^-1 [B | g_p], columns c = q + 4 t, in place; the slot's record for k_backsub
    if (has && !go) for (int e = q; e < a.PF; e += LQ) pf[e] = 0.0;          // no observations or a failed block
#pragma unroll
    for (int t = 0; t < TY; ++t) {
        const int c = q + LQ * t, cc = c < K1 ? c : 0;
        double2* yp = reinterpret_cast<double2*>(sb + Lt::YL + 6 * cc);
        const double2 b0 = yp[0], b1 = yp[1], b2 = yp[2];
        const double bc[6] = { b0.x, b0.y, b1.x, b1.y, b2.x, b2.y };
        double y[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            double v = bc[i];
#pragma unroll
            for (int k = 0; k < i; ++k) v -= L[i * (i + 1) / 2 + k] * y[k];
            y[i] = go ? v * L[i * (i + 1) / 2 + i] : 0.0;
        }
        if (c < K1) {
            yp[0] = double2{ y[0], y[1] }; yp[1] = double2{ y[2], y[3] }; yp[2] = double2{ y[4], y[5] };
            if (go) {
#pragma unroll
                for (int i = 0; i < 6; ++i) pf[21 + i * K1 + c] = y[i];
                if (c == K) {
#pragma unroll
                    for (int i = 0; i < 6; ++i) pf[21 + 6 * K1 + i] = bc[i];
                }
            }
        }
    }
    if (go) {
        // L (21) and diag C (6): every lane holds them; lane q stores entries q, q + 4, ...
#pragma unroll
        for (int e = 0; e < 21; ++e) if ((e % LQ) == q) pf[e] = L[e];
#pragma unroll
        for (int e = 0; e < 6; ++e) if ((e % LQ) == q) pf[21 + 6 * K1 + 6 + e] = dC[e];
    }
    wsync();
    SQ_STAMP(5);
    // ---- image -= Y^T Y: the lane owns COLUMNS j = q + 4 t (Y_j in registers), the rows i stream through (one broadcast read of Y_i
    // serves the four lanes; the entries (i, j) they update are consecutive doubles of row i)
    {
        double yj[TY][6];
#pragma unroll
        for (int t = 0; t < TY; ++t) {
            const int j = q + LQ * t, jc = j < K1 ? j : 0;
            const double2* r = reinterpret_cast<const double2*>(sb + Lt::YL + 6 * jc);
            const double2 r0 = r[0], r1 = r[1], r2 = r[2];
            yj[t][0] = r0.x; yj[t][1] = r0.y; yj[t][2] = r1.x; yj[t][3] = r1.y; yj[t][4] = r2.x; yj[t][5] = r2.y;
        }
        // software pipeline by hand: row i + 1 (Y_i and the direct terms in it) is read before row i is worked on - one writer
        // per entry, plain read - modify - write (a ds_add_f64 per entry cost 32 cycles of the LDS pipe each)
        double2 c0[2], c1[2], c2[2];
        double dv[2][TY];
        auto fetch = [&](const int i, const int b) {
            const double2* r = reinterpret_cast<const double2*>(sb + Lt::YL + 6 * i);
            c0[b] = r[0]; c1[b] = r[1]; c2[b] = r[2];
#pragma unroll
            for (int t = 0; t < TY; ++t) {
                if (LQ * t > i) continue;                    // compile time: every column of this t lies right of the diagonal
                const int j = q + LQ * t;
                // rows of theta_1 and of the extrinsics have no direct term in the theta_0 columns (j < PE): nothing was stored there
                const bool row_all = i < PE || i == K;       // compile time
                if (!row_all && LQ * t + LQ - 1 < PE) { dv[b][t] = 0.0; continue; }
                const double rd = img[sq_tri(i, 0) + (j <= i ? j : i)];       // (right of the diagonal: the diagonal entry again - a broadcast, value unused)
                dv[b][t] = (row_all || LQ * t >= PE || j >= PE) ? rd : 0.0;
            }
        };
        fetch(0, 0);
#pragma unroll
        for (int i = 0; i < K1; ++i) {
            const int b = i & 1;
            if (i + 1 < K1) fetch(i + 1, b ^ 1);
#pragma unroll
            for (int t = 0; t < TY; ++t) {
                if (LQ * t > i) continue;
                const int j = q + LQ * t;
                const double v = ((yj[t][0] * c0[b].x + yj[t][1] * c0[b].y) + (yj[t][2] * c1[b].x + yj[t][3] * c1[b].y)) + (yj[t][4] * c2[b].x + yj[t][5] * c2[b].y);
                *(j <= i ? img + sq_tri(i, 0) + j : dump(sq_tri(i, 0) + LQ * t)) = dv[b][t] - v;
            }
        }
    }
    wsync();
    SQ_STAMP(6);
    // ---- the wavefront's 16 images in a fixed order -> one row of partial sums (k_reduce's layout: the full (K+1)^2 image,
    // lower triangle filled, then the extras)
    // (rows of the upper triangle are never written: the buffer is cleared once when the workspace is made)
    for (int e = lane; e < ACCN; e += 64) {
        int dst;
        if (e < NT) { int i, j; sq_tri_decode(e, i, j); dst = i * K1 + j; }
        else dst = K1 * K1 + (e - NT);
        double t = 0.0;
#pragma unroll
        for (int g = 0; g < SLOTS; g += 4)
            t += (smem[g * Lt::SS + Lt::IMG + sq_img_shift(g) + e] + smem[(g + 1) * Lt::SS + Lt::IMG + sq_img_shift(g + 1) + e]) +
                 (smem[(g + 2) * Lt::SS + Lt::IMG + sq_img_shift(g + 2) + e] + smem[(g + 3) * Lt::SS + Lt::IMG + sq_img_shift(g + 3) + e]);
        a.partial[(int64_t)dst * gridDim.x + blockIdx.x] = t;
    }
#ifdef CCAL_STAMPS
    sq_stamps[7] = wall_clock64();
    if (lane == 0) for (int i = 0; i < 8; ++i) a.partial[(int64_t)a.RB * gridDim.x + 8 * blockIdx.x + i] = (double)sq_stamps[i];
#endif
}

// two cameras, the same number of free intrinsics - at most 6 (UCM, EUCM) -, the standard column layout [theta_0 | theta_1 |
// extrinsics_1].  KB4 / OPENCV5 blocks (7 .. 9): a slot's area grows to 3.5 - 4.7 KB, two wavefronts of 16 slots per CU instead
// of three, i.e. two rounds of wavefronts - two KB4 cameras x 10 000 frames: 147.7 us per build against 137.5 with k_schur
// (136.9 / 117.6 one-focal), also with half-full wavefronts of 8 slots that are all resident; they stay with k_schur
bool schurq_fits(int n_cams, const int* peff, const int* col_theta, const int* col_extr) {
    if (n_cams != 2 || peff[0] != peff[1] || peff[0] < 4 || peff[0] > 6) return false;
    return col_theta[0] == 0 && col_theta[1] == peff[0] && col_extr[1] == 2 * peff[0];
}
// slots per wavefront (16: four lanes per slot, 8: eight lanes per slot), chosen when a problem's workspace is created.
// Measured (two EUCM cameras, whole build, us; eight | four lanes | generic k_schur): 1 000 slots 27.5 | 30.6 | 28.2; 3 000: 37.6 |
// 40.0 | 42.6; 6 000: 56.1 | 57.3 | 63.8; 10 000: 71.6 | 71.8; 20 000: 134.5 | 135.9; two one-focal UCM cameras x 10 000: 64.4 | 62.9 -
// the eight-lane form's shorter chain wins while its wavefronts (twice as many) still get a SIMD to themselves.
// (the second library's CCAL_SCHURQ_SLOTS=8|16 overrides it where the workspace is made, ccal_solver.hip)
int schurq_slots_per_wave(int n_slots) { return n_slots <= 8192 ? 8 : 16; }
int schurq_rows(int n_slots, int slots_per_wave) { return (std::max(n_slots, 1) + slots_per_wave - 1) / slots_per_wave; }

template <int PE, int SLOTS>
static hipError_t launch_schurq_s(const SchurArgs& a, int rows, hipStream_t s) {
    const size_t lds = sizeof(double) * (size_t)SLOTS * SqLayout<PE, 64 / SLOTS>::SS;
    static DynLdsGuard guard;
    if (hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(&k_schurq<PE, SLOTS>), lds, guard); e != hipSuccess) return e;
    hipLaunchKernelGGL((k_schurq<PE, SLOTS>), dim3(rows), dim3(64), lds, s, a);
    return hipGetLastError();
}
template <int PE>
static hipError_t launch_schurq_t(const SchurArgs& a, int rows, int slots_per_wave, hipStream_t s) {
    return slots_per_wave == 8 ? launch_schurq_s<PE, 8>(a, rows, s) : launch_schurq_s<PE, 16>(a, rows, s);
}

hipError_t launch_schurq(const SchurArgs& a, int peff, int rows, int slots_per_wave, hipStream_t s) {
    switch (peff) {
        case 4: return launch_schurq_t<4>(a, rows, slots_per_wave, s);
        case 5: return launch_schurq_t<5>(a, rows, slots_per_wave, s);
        case 6: return launch_schurq_t<6>(a, rows, slots_per_wave, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace ccal
