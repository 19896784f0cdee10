// k_gram2: the register Gram kernel with the block's two rows accumulated on two NEIGHBOURING lanes (even lanes: u rows, odd lanes:
// v rows), rows traded with v_mov_b32_dpp quad_perm:[1,0,3,2].
//
// What k_gram1v (ccal_kernels_fused.hip) keeps per lane is the upper triangle of [J | r]^T W [J | r] of a whole 2-row block: 91 (EUCM)
// ... 136 (OPENCV5) f64 accumulators next to two scaled rows of 13 ... 16 doubles - more than the 256 registers the VALU can address,
// hence 105-136 accumulators behind v_accvgpr copies for KB4 / OPENCV5 (176-224 of the 600-625 instructions of a corner pass).
//
// The block's u row does not touch fy, cy and its v row does not touch fx, cx (one focal: f is shared).  In ROW-LOCAL column order
//       [ f_row | c_row | distortion (ND) | phi (3) | t (3) | r_row ]                    NCR = ND + 9 columns
// both rows have the SAME dense structure.  Every lane evaluates ONE corner completely (transform, projection, both Jacobian rows:
// nothing is computed twice), then lanes 2 k and 2 k + 1 - two corners of the same frame - trade the row the other one accumulates.
// The odd lanes work in MIRRORED coordinates (x and y of the camera-frame point, fx / fy, cx / cy, observed u / v, OPENCV5's p1 / p2
// exchanged): every model is symmetric under that reflection, so one instruction stream forms "the row of my kind" and "the row for my
// neighbour" on every lane, and the trade is one DPP move per register half (round 4 traded between lanes 32 apart with
// v_permlane32_swap: twice the cost).  A lane adds the two rows of its kind into NCR (NCR + 1) / 2 accumulators (UCM 55, EUCM 66, KB4
// 91 - 3 for its Hankel block, OPENCV5 105) - the same number of FMAs per corner as the whole triangle, half the accumulators, no
// structural zeros left to skip.  The per-frame reduction un-mirrors the odd lanes' sums (RowMap), adds the two rows' Grams where they
// meet (distortion, pose and residual columns; f with one focal) and lands in the same per-frame record as k_gram1v's, so the pose
// update in front of the loop and the fused elimination behind it (ccal_gram_common.hpp) are shared.
//   UCM / EUCM: two wavefronts per SIMD (226 registers); KB4 / OPENCV5: one (296 / 322 registers, 30 / 65 AGPR copies per pass).
#include <algorithm>
#include <cstdio>
#include <cstring>

#include "ccal_head.hpp"

namespace ccal {

// ---- compile-time maps -------------------------------------------------------------------------------------------
// The v lanes work in MIRRORED coordinates (x and y of the camera-frame point exchanged, see the kernel): their row-local
// columns are the u lanes' with  phi_0 <-> phi_1,  t_0 <-> t_1  (OPENCV5: p1 <-> p2)  and the three phi columns negated.
template <int MODEL>
__host__ __device__ constexpr int g2_mirror(int a) {
    constexpr int ND = model_np(MODEL) - 4, PH = 2 + ND, T = PH + 3;
    if (a == PH || a == PH + 1) return PH + 1 - (a - PH);
    if (a == T || a == T + 1) return T + 1 - (a - T);
    if (MODEL == kOCV5 && (a == 2 + OCV5_P1 - 4 || a == 2 + OCV5_P2 - 4)) return a == 2 + OCV5_P1 - 4 ? 2 + OCV5_P2 - 4 : 2 + OCV5_P1 - 4;
    return a;
}
// KB4: the distortion columns of a row are  D t^3, D t^5, D t^7, D t^9  (D = sw f x / r): the product of columns i and j depends on i + j
// only - a Hankel block, 7 different sums instead of the 10 of its triangle.  Pairs (p <= q) of one anti-diagonal share ONE accumulator;
// the pair with the smallest p is the one the corner loop adds.
template <int MODEL> __host__ __device__ constexpr bool g2_hankel(int p, int q) { return MODEL == kKB4 && p >= 2 && q < 6; }
template <int MODEL> __host__ __device__ constexpr int g2_hankel_rep_p(int p, int q) { const int s_ = p + q - 4; return (s_ > 3 ? s_ - 3 : 0) + 2; }
template <int MODEL> __host__ __device__ constexpr bool g2_is_rep(int p, int q) { return !g2_hankel<MODEL>(p, q) || p == g2_hankel_rep_p<MODEL>(p, q); }
template <int MODEL> __host__ __device__ constexpr bool g2_is_phi(int a) { return a >= model_np(MODEL) - 2 && a < model_np(MODEL) + 1; }
// full column (order of the block Jacobian: camera P_eff | pose 6 | r) of row-local column a of row `row` (0 = u, 1 = v)
template <int MODEL, bool OF>
__host__ __device__ constexpr int g2_fullcol(int row, int a) {
    if (row) a = g2_mirror<MODEL>(a);
    if (OF) return a == 0 ? 0 : (a == 1 ? (row ? 2 : 1) : a + 1);
    return a == 0 ? (row ? 1 : 0) : (a == 1 ? (row ? 3 : 2) : a + 2);
}
// where entry (i <= j) of the full triangle goes in the per-frame record: dst | mirror << 16 (0xffff = none); the maps of
// RecMap (ccal_kernels_fused.hip) with W = 0
template <int K, bool GEN>
__host__ __device__ constexpr uint32_t g2_rec_dst(int i, int j) {
    constexpr int D = K + 6, K1 = K + 1;
    const bool ip = i >= K && i < D, jp = j >= K && j < D;
    const int ci = i < K ? i : K, cj = j < K ? j : K;
    uint32_t a = 0, b = 0xffff;
    if (!GEN) {
        if (ip && jp) a = (j - K) * (j - K + 1) / 2 + (i - K);
        else if (!ip && jp) a = 21 + (j - K) * K1 + ci;
        else if (ip && !jp) a = 21 + (i - K) * K1 + K;
        else { a = 21 + 6 * K1 + ci * K1 + cj; b = 21 + 6 * K1 + cj * K1 + ci; }
    } else {
        if (ip && jp) { a = (i - K) * 6 + (j - K); if (i != j) b = (j - K) * 6 + (i - K); }
        else if (!ip && jp) a = 36 + ci * 6 + (j - K);
        else if (ip && !jp) a = 36 + K * 6 + (i - K);
        else a = gen_a_off(K) + cj * (cj + 1) / 2 + ci;                    // packed lower triangle (ci <= cj)
    }
    return a | (b << 16);
}
// One ITEM per entry (i <= j) of the full triangle:
//   src = t_u | t_v << 8 | mask << 16 | neg << 18    the accumulator the u lanes hold it in, the one the v lanes hold it in (the
//         mirrored pair), mask bit 0 / 1: the u / v lanes contribute (0: a structural zero of the block), neg: the v lanes' sum
//         enters negated (exactly one of its two columns is a phi column)
//   rec = where it goes in the record
// The accumulators are reduced in NS slices (LDS per wavefront = 64 x slice); they are NUMBERED so that an entry and its mirror
// image sit in the same slice (`where`): an item finds both its sources in one pass.  first[s] = first item of slice s.
// (The four-role form of round 3 - the row-local triangle split between two lane roles, rows staged in LDS - was measured slower
//  for every model and is gone: EXPERIMENTS.md.)
template <int MODEL, bool OF, bool GEN, int NS>
struct RowMap {
    static constexpr int P = model_np(MODEL), ND = P - 4, K = P - (OF ? 1 : 0), D = K + 6, NCF = D + 1, NEF = NCF * (NCF + 1) / 2;
    static constexpr int NCR = ND + 9;
    static constexpr int NER = NCR * (NCR + 1) / 2 - (MODEL == kKB4 ? 3 : 0);      // accumulators per lane (KB4: the Hankel block's 7 for 10)
    static constexpr int CH = (NER + NS - 1) / NS;
    uint32_t rec[NEF];
    uint32_t src[NEF];
    int first[NS + 1];
    uint8_t num[NCR][NCR];                               // accumulator of the row-local pair (p <= q)
    bool ok;                                             // every slice filled without splitting a mirrored pair
    constexpr int where(int p, int q) const { return p <= q ? num[p][q] : num[q][p]; }
    constexpr RowMap() : rec{}, src{}, first{}, num{}, ok(true) {
        // number the accumulators: slice by slice, mirrored pairs first (never split), entries that are their own image after
        bool placed[NCR][NCR] = {};
        int t = 0;
        for (int s = 0; s < NS; ++s) {
            const int end = (s + 1) * CH < NER ? (s + 1) * CH : NER;
            for (int pass = 0; pass < 2; ++pass)
                for (int p = 0; p < NCR; ++p)
                    for (int q = p; q < NCR; ++q) {
                        int mp = g2_mirror<MODEL>(p), mq = g2_mirror<MODEL>(q);
                        if (mp > mq) { const int x = mp; mp = mq; mq = x; }
                        const bool self = mp == p && mq == q;
                        if (!g2_is_rep<MODEL>(p, q)) continue;                    // (numbered below, like its anti-diagonal's first pair)
                        if (placed[p][q] || self != (pass == 1) || t + (self ? 1 : 2) > end) continue;
                        num[p][q] = (uint8_t)t++; placed[p][q] = true;
                        if (!self) { num[mp][mq] = (uint8_t)t++; placed[mp][mq] = true; }
                    }
            if (t != end) ok = false;
        }
        for (int p = 0; p < NCR; ++p)
            for (int q = p; q < NCR; ++q)
                if (!g2_is_rep<MODEL>(p, q)) { const int rp = g2_hankel_rep_p<MODEL>(p, q); num[p][q] = num[rp][p + q - rp]; }
        // what the u lanes / the v lanes hold of every entry of the full triangle
        int tu_of[NCF][NCF] = {}, tv_of[NCF][NCF] = {};
        bool neg_of[NCF][NCF] = {};
        for (int i = 0; i < NCF; ++i) for (int j = 0; j < NCF; ++j) { tu_of[i][j] = -1; tv_of[i][j] = -1; }
        for (int a = 0; a < NCR; ++a)
            for (int b = a; b < NCR; ++b) {
                const int iu = g2_fullcol<MODEL, OF>(0, a), ju = g2_fullcol<MODEL, OF>(0, b);
                int iv = g2_fullcol<MODEL, OF>(1, a), jv = g2_fullcol<MODEL, OF>(1, b);
                if (iv > jv) { const int x = iv; iv = jv; jv = x; }
                tu_of[iu][ju] = where(a, b);
                tv_of[iv][jv] = where(a, b);
                neg_of[iv][jv] = g2_is_phi<MODEL>(a) != g2_is_phi<MODEL>(b);
            }
        // items, slice by slice
        int n = 0;
        for (int s = 0; s < NS; ++s) {
            first[s] = n;
            for (int i = 0; i < NCF; ++i)
                for (int j = i; j < NCF; ++j) {
                    int tu = tu_of[i][j], tv = tv_of[i][j];
                    const int sl = tu >= 0 ? tu / CH : (tv >= 0 ? tv / CH : NS - 1);        // structural zeros of the block: last slice
                    if (sl != s) continue;
                    if (tu >= 0 && tv >= 0 && tv / CH != s) ok = false;
                    const int mask = (tu >= 0 ? 1 : 0) | (tv >= 0 ? 2 : 0);
                    if (tu < 0) tu = tv >= 0 ? tv : s * CH;
                    if (tv < 0) tv = tu;
                    rec[n] = g2_rec_dst<K, GEN>(i, j);
                    src[n] = (uint32_t)tu | ((uint32_t)tv << 8) | ((uint32_t)mask << 16) | ((uint32_t)(neg_of[i][j] ? 1 : 0) << 18);
                    ++n;
                }
        }
        first[NS] = n;
        if (n != NEF) ok = false;
    }
    static constexpr int max_items() {              // most items any slice holds
        RowMap m;
        int mx = 0;
        for (int s = 0; s < NS; ++s) mx = m.first[s + 1] - m.first[s] > mx ? m.first[s + 1] - m.first[s] : mx;
        return mx;
    }
};
template <int MODEL, bool OF, bool GEN, int NS> __device__ const RowMap<MODEL, OF, GEN, NS> g_rowmap = RowMap<MODEL, OF, GEN, NS>();

// slices of the reduction: the LDS a wavefront needs is 64 lanes x (slice | 1) doubles; two wavefronts per SIMD = 8 per CU
// have to share 160 KB with the frames' constants
template <int MODEL> __host__ __device__ constexpr int g2_slices() { return (MODEL == kKB4 || MODEL == kOCV5) ? 3 : 2; }

// diagnostic build (-DCCAL_STAMPS, tools/stamps_g2.py): the 100 MHz clock at the phase boundaries of every wavefront, parked
// in the per-frame scratch (8 stamps per wavefront)
#ifdef CCAL_STAMPS       // kept in (scalar) registers, stored once at the end: a store per stamp would sit in front of the next fence
#define G2_STAMP(i) do { g2_stamps[i] = wall_clock64(); } while (0)
#define G2_STAMPS_DECL long long g2_stamps[6] = { 0, 0, 0, 0, 0, 0 }; long long g2_cyc[5] = { 0, 0, 0, 0, 0 }; long long g2_c0 = 0; long long g2_ep[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }; long long g2_pro[4] = { 0, 0, 0, 0 }
#define G2_EP(i) do { g2_ep[i] = wall_clock64(); } while (0)
#define G2_PRO(i, dep) do { asm volatile("" :: "v"(dep)); g2_pro[i] = wall_clock64(); } while (0)     /* a station of the prologue: once `dep` has arrived */
#define G2_EP_PTR , g2_ep + 3
#define G2_STAMPS_FLUSH do { if (!GEN && lane == 0) { for (int i_ = 0; i_ < 6; ++i_) a.fcbuf[32 * wrow + i_] = (double)g2_stamps[i_]; \
                                                       for (int i_ = 0; i_ < 5; ++i_) a.fcbuf[32 * wrow + 8 + i_] = (double)g2_cyc[i_]; \
                                                       for (int i_ = 0; i_ < 8; ++i_) a.fcbuf[32 * wrow + 16 + i_] = (double)g2_ep[i_]; \
                                                       for (int i_ = 0; i_ < 4; ++i_) a.fcbuf[32 * wrow + 24 + i_] = (double)g2_pro[i_]; } } while (0)
// -DCCAL_STAMPS=2: shader cycles (s_memtime) spent in the sections of the corner loop, summed over the passes:
// 2 projection + rows + DPP, 3 Gram products, 4 passes
#if CCAL_STAMPS >= 2
#define G2_CYC_BEGIN() do { __builtin_amdgcn_sched_barrier(0); g2_c0 = clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define G2_CYC(i) do { __builtin_amdgcn_sched_barrier(0); const long long c_ = clock64(); g2_cyc[i] += c_ - g2_c0; g2_c0 = c_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define G2_CYC_BEGIN() do { } while (0)
#define G2_CYC(i) do { } while (0)
#endif
#else
#define G2_STAMP(i) do { } while (0)
#define G2_STAMPS_DECL do { } while (0)
#define G2_STAMPS_FLUSH do { } while (0)
#define G2_EP(i) do { } while (0)
#define G2_PRO(i, dep) do { } while (0)
#define G2_EP_PTR
#define G2_CYC_BEGIN() do { } while (0)
#define G2_CYC(i) do { } while (0)
#endif
// wavefronts per SIMD the kernel is compiled for: two where 2 NER accumulators + two rows fit 256 registers (UCM, EUCM);
// KB4 / OPENCV5 (182 / 210 + 52 / 56) do not without scratch or LDS accumulators, both of which cost more than they buy
// (measured, 10 000 frames, build: scratch 99 / 98 us, 16 / 32 LDS accumulators 68 / 60 us, one wavefront per SIMD 52 / 50 us)
#ifndef CCAL_G2_MINW
#define CCAL_G2_MINW(MODEL) (((MODEL) == kKB4 || (MODEL) == kOCV5) ? 1 : 2)
#endif
// R and t of the lane's frame in registers (24 of them) instead of twelve LDS reads per corner: only where they fit
#ifndef CCAL_G2_HOIST
#define CCAL_G2_HOIST(MODEL) ((MODEL) == kUCM)
#endif

// R | t of a frame with the x and y rows exchanged: what the v lanes transform their corners with
__device__ __forceinline__ void g2_store_mirrored(const double* fcr, double* dst) {
#pragma unroll
    for (int i = 0; i < 3; ++i) { dst[i] = fcr[3 + i]; dst[3 + i] = fcr[i]; dst[6 + i] = fcr[6 + i]; }
    dst[9] = fcr[10]; dst[10] = fcr[9]; dst[11] = fcr[11];
}
// the value of the neighbouring lane (lane ^ 1): two v_mov_b32 with a quad_perm [1,0,3,2] DPP control
__device__ __forceinline__ double g2_from_partner(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0xB1, 0xF, 0xF, true);      // (every lane has a source: nothing is kept of the old value)
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0xB1, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

// LPF = lanes per frame, EVEN: LPF corners of a frame per pass, every lane evaluates one.  The frame's lanes are contiguous (grp = lane / LPF), so
// the prologue and the fused tail are those of k_gram1w.
// The body of a workgroup.  BIN = false: workgroup `wg` of a launch in which every frame has LPF lanes; the frames are a.n_obs
// observation frames in their own order (GEN: a.list).  BIN = true (single camera, ragged frames): the launch is cut into BINS of
// frames sorted by corner count, each with the lanes per frame ITS frames need (k_gram2b below) - this workgroup is number `wg` of a bin
// whose frames are positions frame0 .. frame0 + nfr - 1 of the sorted table a.bin_tab (frame | first corner | corners | slot:
// ONE 16-byte load instead of the offsets' and the slot table's).  The per-frame buffers of the loop (records, model decrease, cost)
// are then indexed by POSITION in that table - they never leave the single-camera kernels -, poses and elimination records by slot
// as ever; a wavefront's row of partial sums is its number in the launch, so the rows are added in the same order every time.
// ITER (k_gram2i; single-camera loop, 2 000 .. ~10 000 frames of UCM / EUCM): a whole group in ONE launch, as k_gram1v<.., ITER> does for
// session sizes (ccal_kernels_fused.hip) - workgroups of EIGHT wavefronts (two per SIMD: one workgroup per compute unit, <= 256 of them);
// in front of the evaluation every workgroup sums the previous launch's rows, its first wavefront decides and solves the camera
// system (head_wave: the same arithmetic on the same sums in every workgroup, workgroup 0 writes), the evaluation takes state, camera
// step and candidate intrinsics from LDS, and behind it the eight wavefronts' rows are added in LDS and leave as ONE packed row.
template <int MODEL, bool OF, int LPF, bool GEN, bool BIN, bool ITER = false>
__device__ __forceinline__ void gram2_body(const FusedArgs& a, double* smem, const int wg, const int frame0, const int nfr) {
    static_assert(LPF % 2 == 0, "a frame's lanes: LPF / 2 pairs of a u lane and a v lane");
    static_assert(!(GEN && BIN), "bins: the single-camera loop");
    static_assert(!(ITER && GEN), "single-launch groups: the single-camera loop");      // (ITER && BIN: ONE bin - the frames in sorted order, see k_gram2i)
    constexpr int WPB = ITER ? 8 : CCAL_GRAMV_WPB;
    constexpr int NS = g2_slices<MODEL>();
    using Map = RowMap<MODEL, OF, GEN, NS>;
    static_assert(Map().ok, "accumulator numbering: a slice splits a mirrored pair");
    constexpr int L2 = LPF / 2;                     // pairs of lanes per frame
    constexpr int G = 32 / L2;                      // frames per wavefront
    constexpr int FCS = FC_N0P + 12;                // a frame's constants + R, t once more with the x and y rows exchanged (v lanes)
    constexpr int P = model_np(MODEL), ND = P - 4;
    constexpr int D = block_dim(MODEL, OF, false);
    constexpr int K = D - 6, K1 = K + 1;
    constexpr int NCR = Map::NCR, NER = Map::NER, CH = Map::CH, NEF = Map::NEF;
    constexpr int LS = CH | 1;                      // odd row stride (doubles): conflict-free column sums
    constexpr int REC_ = praw_jl_off(K) + 9, GS_ = (REC_ + 6 * K1 + 1) & ~1;
    constexpr int RED = (!GEN && G * GS_ > 64 * LS) ? G * GS_ : 64 * LS;
    constexpr int WSL = (G * FCS + RED + NEF + 1) & ~1;        // per wave: G frames' constants | reduction buffer / records | item table
    static_assert(!ITER || fused_red_size(K) <= G * FCS, "a wavefront's row of partial sums is parked where its frames' constants were");
    const bool fuse = !GEN && a.fuse_elim != 0;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // frame g: lanes [g LPF, (g + 1) LPF); even lanes take the u rows, odd lanes the v rows; gl = the lane's index within its frame
    const int role = lane & 1;
    const bool lane_ok = lane < G * LPF;
    const int grp = lane_ok ? lane / LPF : G - 1;
    const int gl = lane_ok ? lane - grp * LPF : (lane & 1);
    G2_STAMPS_DECL;
    G2_STAMP(0);
    const int wrow = blockIdx.x * WPB + wave;                     // the wavefront's number in the launch: its row of partial sums
    // which group of G frames this wavefront takes.  ITER over the sorted table: a workgroup's wavefronts w and w + 4 share a SIMD, so the
    // first four take groups from the front of the table (large frames), the last four from behind the first 4 x gridDim groups (small
    // ones): every SIMD pairs a long wavefront with a short one
    const int mwave = (ITER && BIN) ? (wave < 4 ? wg * 4 + wave : 4 * (int)gridDim.x + wg * 4 + (wave - 4)) : wg * WPB + wave;
    const int fpos = mwave * G + grp;                              // the frame's position among the nfr frames this launch / bin works on
    const bool active = lane_ok && fpos < nfr;
    const int f = frame0 + (active ? fpos : 0);                    // index of the loop's per-frame buffers (BIN: the position in the sorted table)
    int4 btab = make_int4(0, 0, 0, 0);
    if constexpr (BIN) btab = reinterpret_cast<const int4*>(a.bin_tab)[f];
    const int fa_ = BIN ? btab.x : (GEN ? a.list[f] : (active ? fpos : 0));
    const int camf = (GEN && a.obs_cam) ? a.obs_cam[fa_] : a.cam;          // GEN, merged launch: the frame's camera
    double* fcw = smem + wave * WSL;
    double* fc = fcw + grp * FCS;
    double* red = fcw + G * FCS;
    // the per-item tables of the reduction (src | rec << 32), requested NOW and parked in LDS before the corner loop: the
    // reduction and the scatter then look them up at LDS latency instead of waiting for global memory twice per slice
    unsigned long long* tab = reinterpret_cast<unsigned long long*>(red + RED);
    constexpr int NTQ = (NEF + 63) / 64;
    unsigned long long tabv[NTQ];
    {
        const Map& gmap = g_rowmap<MODEL, OF, GEN, NS>;
#pragma unroll
        for (int q = 0; q < NTQ; ++q) {
            const int it = lane + 64 * q;
            tabv[q] = it < NEF ? ((unsigned long long)gmap.src[it] | ((unsigned long long)gmap.rec[it] << 32)) : 0ull;
        }
    }
    // Single-camera loop: everything the prologue needs from memory that does NOT depend on the optimizer state is requested before
    // the state is looked at - the frame's offsets and slot, its pose and the intrinsics in BOTH parameter sets (the state picks one),
    // the first corner rows: one memory round trip in front of the exponential map instead of three (stations of the -DCCAL_STAMPS
    // build, 10 000 frames: state + intrinsics 0.8 us, then pose 0.9, then exponential map 0.3, first rows 0.6)
    // (rigs: the camera's extrinsics of both sets too; a set that another wavefront is writing - the candidate poses of
    // FusedArgs::gen_backsub - is the one the state does NOT pick)
    double th_b[2][th_len<MODEL>()], pose_b[2][6], ex_b[2][GEN ? 6 : 1];
    int slot_e;
    int64_t start_e, end_e;
    if constexpr (BIN) { slot_e = btab.w; start_e = btab.y; end_e = (int64_t)btab.y + btab.z; }
    else {
        if (!GEN && a.slot_ident) slot_e = fa_; else slot_e = a.obs_slot[fa_];
        start_e = a.obs_off[fa_]; end_e = a.obs_off[fa_ + 1];
    }
    {
        const int th_off = (GEN && a.obs_cam) ? camf * CCAL_PMAX : 0;
        load_theta<MODEL, OF>(a.intr[0] + th_off, a.rt, th_b[0]);
        load_theta<MODEL, OF>(a.intr[1] + th_off, a.rt, th_b[1]);
        // (ITER, the solve's first launch as its own k_unpack1: the starting pose from where the caller left it - pinned host memory, or set 0)
        const bool fold_src = ITER && a.it.skip_head != 0 && a.it.fold != 0;
        const double* ps0 = fold_src ? (a.it.poses_on_device ? a.poses[0] : a.it.poses_src) : a.poses[0];
        const double* ps1 = fold_src ? ps0 : a.poses[1];
#pragma unroll
        for (int i = 0; i < 6; ++i) { pose_b[0][i] = ps0[(int64_t)slot_e * 6 + i]; pose_b[1][i] = ps1[(int64_t)slot_e * 6 + i]; }
        if constexpr (GEN) {
#pragma unroll
            for (int i = 0; i < 6; ++i) { ex_b[0][i] = camf > 0 ? a.extr[0][camf * 6 + i] : 0.0; ex_b[1][i] = camf > 0 ? a.extr[1][camf * 6 + i] : 0.0; }
        }
    }
    // what the evaluation needs of the optimizer state, in registers (ITER: this launch decides itself - LDS; else global memory)
    struct { int done, redo, cur, first, method; double lambda_solve, lam_schur; } g;
    constexpr int PROW = ITER ? iter_row_len(K) : 1;
    __shared__ typename std::conditional<ITER, HeadShared, int>::type hsh;
    bool from_lds = false, fold_first = false;
    if constexpr (ITER) {
        const IterArgs& it = a.it;
        fold_first = it.skip_head != 0 && it.fold != 0;
        if (!it.skip_head) {
            HeadIO io;
            io.st_in = it.st_in; io.st_out = it.st_out; io.hs = it.hs; io.red_g = nullptr; io.cols = it.cols;
            io.intr[0] = a.intr[0]; io.intr[1] = a.intr[1]; io.dc = it.dc_out; io.K = K; io.seq = it.seq;
            io.min_diag = a.min_diag; io.max_diag = a.max_diag; io.publish_all = it.publish_all;
            HeadPre hpre = {};
            if (threadIdx.x < 64) hpre = head_prefetch(io, (int)threadIdx.x);
            __shared__ double shr[4][(PROW + 63) / 64][64];
            double vsum[(PROW + 63) / 64][4];
            iter_reduce_load<K>(it.partial_in, it.n_part_in, vsum);           // (wavefronts 4 .. 7 find no rows of theirs: zeros)
            iter_reduce_combine<K>(vsum, hsh.red, shr);
            const bool writer = blockIdx.x == 0;
            if (threadIdx.x < 64) head_wave(io, hsh, (int)threadIdx.x, writer, hpre);
            __syncthreads();
            head_finish(io, hsh, it.result_host, a.poses[0], a.poses[1], it.np6, writer, it.result_poses, it.done_cnt);
            from_lds = true;
            const DevState& S = hsh.S0;
            g.done = S.done; g.redo = S.redo; g.cur = S.cur; g.first = S.first; g.method = S.method;
            g.lambda_solve = S.lambda_solve; g.lam_schur = schur_lambda(&S);
        } else if (fold_first) {
            // the solve's first launch AND its k_unpack1: state, columns and intrinsics from the argument block
            constexpr int NSW = (int)(sizeof(DevState) / sizeof(double)), NC1 = (int)(sizeof(ColInfo) / sizeof(double));
            if (threadIdx.x < CCAL_PMAX) hsh.cand[threadIdx.x] = it.poses_on_device ? a.intr[0][threadIdx.x] : reinterpret_cast<const double*>(it.intr_h)[threadIdx.x];
            if (blockIdx.x == 0 && threadIdx.x < 64) {
                for (int e = threadIdx.x; e < NSW; e += 64) reinterpret_cast<double*>(it.st_out)[e] = reinterpret_cast<const double*>(&it.st0)[e];
                for (int e = threadIdx.x; e < it.n_cols * NC1; e += 64) reinterpret_cast<double*>(it.cols_out)[e] = reinterpret_cast<const double*>(it.col0)[e];
                if (threadIdx.x < CCAL_PMAX) {
                    const double v = it.poses_on_device ? a.intr[0][threadIdx.x] : reinterpret_cast<const double*>(it.intr_h)[threadIdx.x];
                    if (!it.poses_on_device) a.intr[0][threadIdx.x] = v;
                    a.intr[1][threadIdx.x] = v;
                }
                if (threadIdx.x == 0) it.hs->word = status_word(it.seq, 0, 0);
            }
            __syncthreads();
            from_lds = true;                       // (the intrinsics: hsh.cand; the camera step is not read in a first evaluation)
            const DevState& S = it.st0;
            g.done = S.done; g.redo = S.redo; g.cur = S.cur; g.first = S.first; g.method = S.method;
            g.lambda_solve = S.lambda_solve; g.lam_schur = schur_lambda(&S);
        } else {
            // the solve's first launch: the starting state passes through to the buffer the next launch reads
            if (blockIdx.x == 0 && threadIdx.x < 64) {
                const double* src = reinterpret_cast<const double*>(it.st_in);
                double* dst = reinterpret_cast<double*>(it.st_out);
                for (int e = threadIdx.x; e < (int)(sizeof(DevState) / sizeof(double)); e += 64) dst[e] = src[e];
                if (threadIdx.x == 0) it.hs->word = status_word(it.seq, 0, 0);
            }
            const DevState* st = it.st_in;
            g.done = st->done; g.redo = st->redo; g.cur = st->cur; g.first = st->first; g.method = st->method;
            g.lambda_solve = st->lambda_solve; g.lam_schur = schur_lambda(st);
        }
    } else {
        g.done = g.redo = g.cur = g.first = g.method = 0; g.lambda_solve = 0.0; g.lam_schur = 0.0;      // (unused: see below)
    }
    // The fields of the state as the evaluation reads them: ITER from this launch's decision (copied out of LDS above); else straight from
    // the state in global memory WHERE THEY ARE NEEDED, exactly as before the single-launch form existed - read up front into a struct they
    // cost KB4 / OPENCV5, which have no register to spare, 1 us per build (profiles/r06/ab_g2_iter_body.txt)
    auto st_done = [&]() -> int { if constexpr (ITER) return g.done; else return a.st->done; };
    auto st_redo = [&]() -> int { if constexpr (ITER) return g.redo; else return a.st->redo; };
    auto st_cur = [&]() -> int { if constexpr (ITER) return g.cur; else return a.st->cur; };
    auto st_first = [&]() -> int { if constexpr (ITER) return g.first; else return a.st->first; };
    auto st_method = [&]() -> int { if constexpr (ITER) return g.method; else return a.st->method; };
    auto st_lambda_solve = [&]() -> double { if constexpr (ITER) return g.lambda_solve; else return a.st->lambda_solve; };
    auto st_lam_schur = [&]() -> double { if constexpr (ITER) return g.lam_schur; else return schur_lambda(a.st); };
    // ITER: the eight wavefronts' rows (parked where their frames' constants were) -> the workgroup's row, fixed order, the two symmetric
    // blocks as their upper triangles
    auto iter_row_out = [&]() {
        if constexpr (ITER) {
            __syncthreads();
            if (threadIdx.x < 64) for (int pe = threadIdx.x; pe < PROW; pe += 64) {
                const int e = iter_row_src(K, pe);
                double t = smem[e];
#pragma unroll
                for (int w = 1; w < WPB; ++w) t += smem[w * WSL + e];
                a.partial[(int64_t)blockIdx.x * PROW + pe] = t;
            }
        }
    };
    const bool keep_rec = GEN || !fuse || st_method() == CCAL_METHOD_LM;
    if (st_done() || (st_redo() && !fuse)) return;
    const int cur = st_cur(), first = st_first();
    const int es = first ? cur : (cur ^ 1);
    if constexpr (!GEN) {
        if (st_redo()) {
            // re-elimination group (LM: rejected step or missed speculation): the accepted set's stored records, new damping
            double* R = red + grp * GS_;
            double mcv = 0.0;
            int slot_r = 0;
            if (active) {
                const double* rec = a.praw[cur] + (int64_t)f * a.PRAW;
                slot_r = BIN ? slot_e : a.obs_slot[f];
                for (int e = gl; e < REC_; e += LPF) R[e] = rec[e];
                if (gl == 0) mcv = a.mc_f[f];
            }
            wsync();
            gram_fused_tail<K, LPF>(a, st_lam_schur(), red, ITER ? fcw : a.partial + (int64_t)wrow * fused_red_size(K), grp, gl, lane_ok, active, slot_r, cur, mcv);
            iter_row_out();
            return;
        }
    }
    const int cl = gl;                              // every lane has its own corner: LPF corners of a frame per pass
    double th[th_len<MODEL>()];
#pragma unroll
    for (int i = 0; i < th_len<MODEL>(); ++i) th[i] = es ? th_b[1][i] : th_b[0][i];
    if constexpr (ITER) { if (from_lds) load_theta<MODEL, OF>(hsh.cand, a.rt, th); }      // the candidate this launch has just formed (or, fold, the start)
    G2_PRO(0, th[0] + (double)first);
    // the lane's view of the intrinsics: MINE = the row it accumulates (u lanes: fx, cx; v lanes: fy, cy), OTHER = the row it
    // forms for its partner; the v lanes see the distortion in mirrored coordinates (OPENCV5: p1 and p2 exchanged)
    if (role) {
        if constexpr (!OF) { const double x = th[0]; th[0] = th[1]; th[1] = x; }
        { const double x = th[2]; th[2] = th[3]; th[3] = x; }
        if constexpr (MODEL == kOCV5) { const double x = th[OCV5_P1]; th[OCV5_P1] = th[OCV5_P2]; th[OCV5_P2] = x; }
    }
    const int64_t start = start_e;
    const int n = active ? (int)(end_e - start) : 0;
    // corner rows by 32-bit byte offsets from the (wave-uniform) stream pointers: no 64-bit address arithmetic per load
    // (ccal_problem_create refuses more than 2^30 - 1 corners per problem)
    auto ldf = [](const float* base, uint32_t byte_off) { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off); };
    const uint32_t ob0 = (uint32_t)start * 4u;
    float pX, pY, pZ, pU, pV;
    {
        const uint32_t g0 = ob0 + 4u * (uint32_t)(cl < n ? cl : 0);
        pX = ldf(a.x, g0); pY = ldf(a.y, g0); pZ = ldf(a.z, g0); pU = ldf(a.u, g0); pV = ldf(a.v, g0);
    }
    {
        // candidate pose of this group's frame (back-substitution of the previous camera solve) + constants; the lanes of
        // a group compute the same values, the G groups work on G frames at once
        // the frame's slot: the table - or, when the table is the identity, the frame itself: pose and elimination record are
        // requested with the frame's offsets instead of a memory round trip after them
        const int slot = slot_e;
        double pose[6];
        // GEN: k_backsub has formed the candidate - or, FusedArgs::gen_backsub, it is formed here from the accepted pose
        const bool gbs = GEN && a.gen_backsub != 0 && !first;
        const int pset = (GEN && !gbs) ? es : cur;
#pragma unroll
        for (int i = 0; i < 6; ++i) pose[i] = pset ? pose_b[1][i] : pose_b[0][i];
        if constexpr (GEN) {
            if (gbs) {
                const double mcg = gen_backsub_pose<LPF, G, RED>(a, slot, st_lambda_solve(), pose, red, grp, gl, lane_ok);
                if (active && a.g_owner[fa_]) {
#pragma unroll
                    for (int i = 0; i < 6; ++i) if (gl == i) a.poses[es][(int64_t)slot * 6 + i] = pose[i];
                    if (gl == 0) a.g_mc_slot[slot] = mcg;
                }
            }
        }
        double mc = 0.0;
        const double* dcv = a.dc;                    // the camera step: ITER - this launch's own solve, in LDS
        if constexpr (ITER) { if (from_lds) dcv = hsh.x; }
        if (!GEN && !first) {
            const double* pf = a.pf[cur] + (int64_t)slot * a.PF;
            if (pf[0] != 0.0) {
                double dp[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const double* yr = pf + 21 + i * K1;
                    double t = yr[K];
#pragma unroll
                    for (int j = 0; j < K; ++j) t += yr[j] * dcv[j];
                    dp[i] = -t;
                }
#pragma unroll
                for (int i = 5; i >= 0; --i) {
                    double t = dp[i];
#pragma unroll
                    for (int k = i + 1; k < 6; ++k) t -= pf[k * (k + 1) / 2 + i] * dp[k];
                    dp[i] = t * pf[i * (i + 1) / 2 + i];
                }
                const double lam = st_lambda_solve();
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
            double ex[6], fcr[12], ept[GEN_EPT];
#pragma unroll
            for (int i = 0; i < 6; ++i) ex[i] = es ? ex_b[1][i] : ex_b[0][i];
            frame_setup_composed(pose, ex, fcr, ept);
            if (gl == 0 && lane_ok) {
#pragma unroll
                for (int i = 0; i < 12; ++i) fc[i] = fcr[i];
            }
            if (gl == 1 && lane_ok) g2_store_mirrored(fcr, fc + FC_N0P);
            if (gl == 0 && active) {
                double ec[GEN_EC];
                gen_et_compact(ept, ec);
                double2* rec = reinterpret_cast<double2*>(a.praw[es] + a.rec_off[fa_] + gen_e_off(K));
#pragma unroll
                for (int i = 0; i < GEN_EC / 2; ++i) rec[i] = make_double2(ec[2 * i], ec[2 * i + 1]);
            }
        } else {
            G2_PRO(1, pose[0] + pose[5]);
            double fcr[FC_N0];
            frame_setup<false>(pose, nullptr, fcr);
            G2_PRO(2, fcr[0] + fcr[FC_N0 - 1]);
            if (gl == 0 && lane_ok) {
#pragma unroll
                for (int i = 0; i < FC_N0; ++i) fc[i] = fcr[i];
            }
            if (gl == 1 && lane_ok) g2_store_mirrored(fcr, fc + FC_N0P);
        }
    }
    G2_PRO(3, pX + pV);                             // (the first pass's corner rows)
    wsync();
    G2_STAMP(1);

    double acc[NER];
#pragma unroll
    for (int e = 0; e < NER; ++e) acc[e] = 0.0;
    // same trip count for the whole wave: the largest frame of the group
    int nmax = n;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) nmax = max(nmax, __shfl_xor(nmax, off, 64));
#pragma unroll
    for (int q = 0; q < NTQ; ++q) if (lane + 64 * q < NEF) tab[lane + 64 * q] = tabv[q];
    constexpr bool HOIST = CCAL_G2_HOIST(MODEL);
    // a corner needs R and t only (phi basis): in registers where they fit, else re-read from LDS per corner
    const double* fcm = role ? fc + FC_N0P : fc;           // v lanes: x and y rows of R, t exchanged
    double fcl[12];
    if constexpr (HOIST) {
#pragma unroll
        for (int i = 0; i < 12; ++i) fcl[i] = fcm[i];
    }
    const double* fcp = HOIST ? fcl : fcm;
    // One corner: transform, projection and its partials, residual, weight, and the two scaled rows in row-local column order
    // [f | c | distortion | phi | t | r] (the weight rides on the focal length, folded into the factors a row's partials share):
    // MINE, the row of the lane's own kind, and OTHER, the row its neighbour accumulates, which it gets from there by DPP.
    // The v lanes do all this in MIRRORED coordinates - x and y of the camera-frame point exchanged (their R, t come from the
    // exchanged copy in LDS), fx / fy, cx / cy, the observed u / v (and OPENCV5's p1 / p2) exchanged: every model here is
    // symmetric under that reflection, so the SAME instructions give a v lane its v row as "the first row" and the u row as
    // "the second" - no per-lane selects.  In mirrored coordinates the pose columns come out as t' = (t_y, t_x, t_z),
    // phi' = -(phi_y, phi_x, phi_z); OTHER is written in the PARTNER's convention (components exchanged, cross product
    // reversed: free), and RowMap un-mirrors the v lanes' sums where the frame's lanes meet.
    // (project_partials of ccal_device.hpp, which mode E and k_gram1v use, is the same arithmetic before the folding.)
    // the pose columns and the residual of both rows from the scaled translation partials (u: MINE, v: OTHER, in the partner's
    // convention), then the trade
    auto rows_finish = [&](double rx, double ry, double rz, double u0, double u1, double u2, double v0, double v1, double v2,
                           double sw, double ru, double rv, double* su, double* so, double* sv) {
        su[2 + ND + 0] = ry * u2 - rz * u1; su[2 + ND + 1] = rz * u0 - rx * u2; su[2 + ND + 2] = rx * u1 - ry * u0;   // (R X) x j
        so[2 + ND + 0] = rx * v2 - rz * v0; so[2 + ND + 1] = rz * v1 - ry * v2; so[2 + ND + 2] = ry * v0 - rx * v1;
        su[2 + ND + 3] = u0; su[2 + ND + 4] = u1; su[2 + ND + 5] = u2;
        so[2 + ND + 3] = v1; so[2 + ND + 4] = v0; so[2 + ND + 5] = v2;
        su[NCR - 1] = sw * ru;          so[NCR - 1] = sw * rv;
#pragma unroll
        for (int i = 0; i < NCR; ++i) sv[i] = g2_from_partner(so[i]);
    };
    // UCM / EUCM: steps A and B in one, with the weight and the focal length folded into the two factors every partial of a row
    // shares (e = sw f / n, a = e m): 9 FP64 instructions fewer per corner than scaling project_partials' outputs
    auto rows_ucm = [&](double X, double Y, double Z, double uo, double vo, bool valid, double* su, double* sv) {
      if constexpr (MODEL == kUCM || MODEL == kEUCM) {
            if constexpr (!HOIST) asm volatile("" ::: "memory");
            const double rx = fcp[0] * X + fcp[1] * Y + fcp[2] * Z, ry = fcp[3] * X + fcp[4] * Y + fcp[5] * Z, rz = fcp[6] * X + fcp[7] * Y + fcp[8] * Z;
            const double px = rx + fcp[9], py = ry + fcp[10], pz = rz + fcp[11];
            const double alpha = th[4], beta = (MODEL == kEUCM) ? th[5] : 1.0;
            const double r2 = px * px + py * py;
            double rho, irho;
            fast_sqrt_rsqrt(beta * r2 + pz * pz, rho, irho);
            const double inv = fast_rcp(alpha * rho + (1.0 - alpha) * pz);
            const double mx = px * inv, my = py * inv;
            const double ru = th[0] * mx + th[2] - uo, rv = th[1] * my + th[3] - vo;
            const double sw = valid ? huber_sqrt_weight(ru * ru + rv * rv, a.huber_delta) : 0.0;
            const double ab = (alpha * beta) * irho;
            const double nx = ab * px, ny = ab * py, nz = (alpha * pz) * irho + (1.0 - alpha);
            const double eu = (sw * th[0]) * inv, ev = (sw * th[1]) * inv;
            const double au = eu * mx, av = ev * my;
            double so[NCR];
            su[0] = sw * mx;                so[0] = sw * my;
            su[1] = sw;                     so[1] = sw;
            const double na = rho - pz;
            su[2] = -au * na;               so[2] = -av * na;
            if constexpr (MODEL == kEUCM) { const double nb = (0.5 * alpha * r2) * irho; su[3] = -au * nb; so[3] = -av * nb; }
            const double u0 = __builtin_fma(-au, nx, eu), u1 = -au * ny, u2 = -au * nz;
            const double v0 = -av * nx, v1 = __builtin_fma(-av, ny, ev), v2 = -av * nz;
            rows_finish(rx, ry, rz, u0, u1, u2, v0, v1, v2, sw, ru, rv, su, so, sv);
      }
    };
    // KB4 / OPENCV5: the same folding (project_partials' formulas, ccal_device.hpp, with sw f carried into the common factors of a
    // row's partials): 10 FP64 instructions fewer per corner each
    auto rows_kb4 = [&](double X, double Y, double Z, double uo, double vo, bool valid, double* su, double* sv) {
      if constexpr (MODEL == kKB4) {
            if constexpr (!HOIST) asm volatile("" ::: "memory");
            const double rx = fcp[0] * X + fcp[1] * Y + fcp[2] * Z, ry = fcp[3] * X + fcp[4] * Y + fcp[5] * Z, rz = fcp[6] * X + fcp[7] * Y + fcp[8] * Z;
            const double px = rx + fcp[9], py = ry + fcp[10], pz = rz + fcp[11];
            const double r2 = px * px + py * py;
            double r, ir;
            fast_sqrt_rsqrt(r2, r, ir);          // r2 == 0 gives NaN here and falls into the pinhole branch below
            double s_, g, sz, t;                 // m = s (x, y);  d m / d (x, y, z) from s, g, sz;  theta
            if (r > th[model_np(kKB4)]) {          // run-time convention slot (load_theta): ccal_model_conventions.kb4_small_radius
                t = fast_atan2_pos(r, pz);
                const double t2 = t * t;
                const double k1 = th[4], k2 = th[5], k3 = th[6], k4 = th[7];
                const double td = t * (1.0 + t2 * (k1 + t2 * (k2 + t2 * (k3 + t2 * k4))));
                const double tdp = 1.0 + t2 * (3.0 * k1 + t2 * (5.0 * k2 + t2 * (7.0 * k3 + t2 * 9.0 * k4)));
                s_ = td * ir;
                const double id2 = fast_rcp(r2 + pz * pz);
                const double tq = pz * ir * id2, tz = -r * id2;
                g = ir * (tdp * tq - s_ * ir);
                sz = ir * tdp * tz;
            } else {                             // pinhole limit: m = (x, y) / z, no distortion partials
                const double iz = fast_rcp(pz);
                s_ = iz; g = 0.0; sz = -iz * iz; t = 0.0; ir = 0.0;
            }
            const double mx = px * s_, my = py * s_;
            const double ru = th[0] * mx + th[2] - uo, rv = th[1] * my + th[3] - vo;
            const double sw = valid ? huber_sqrt_weight(ru * ru + rv * rv, a.huber_delta) : 0.0;
            const double fxs = sw * th[0], fys = sw * th[1];
            const double A = fxs * px, B = fys * py, Ag = A * g, Bg = B * g;
            const double u0 = __builtin_fma(Ag, px, fxs * s_), u1 = Ag * py, u2 = A * sz;
            const double v0 = Bg * px, v1 = __builtin_fma(Bg, py, fys * s_), v2 = B * sz;
            const double t2 = t * t, t3 = t2 * t, t5 = t3 * t2, t7 = t5 * t2, t9 = t7 * t2;
            const double Du = A * ir, Dv = B * ir;
            double so[NCR];
            su[0] = sw * mx;                so[0] = sw * my;
            su[1] = sw;                     so[1] = sw;
            su[2] = Du * t3; su[3] = Du * t5; su[4] = Du * t7; su[5] = Du * t9;
            so[2] = Dv * t3; so[3] = Dv * t5; so[4] = Dv * t7; so[5] = Dv * t9;
            rows_finish(rx, ry, rz, u0, u1, u2, v0, v1, v2, sw, ru, rv, su, so, sv);
      }
    };
    auto rows_ocv5 = [&](double X, double Y, double Z, double uo, double vo, bool valid, double* su, double* sv) {
      if constexpr (MODEL == kOCV5) {
            if constexpr (!HOIST) asm volatile("" ::: "memory");
            const double rx = fcp[0] * X + fcp[1] * Y + fcp[2] * Z, ry = fcp[3] * X + fcp[4] * Y + fcp[5] * Z, rz = fcp[6] * X + fcp[7] * Y + fcp[8] * Z;
            const double px = rx + fcp[9], py = ry + fcp[10], pz = rz + fcp[11];
            const double k1 = th[OCV5_K1], k2 = th[OCV5_K2], p1 = th[OCV5_P1], p2 = th[OCV5_P2], k3 = th[OCV5_K3];      // (v lanes: p1, p2 exchanged)
            const double iz = fast_rcp(pz);
            const double xn = px * iz, yn = py * iz;
            const double xx = xn * xn, yy = yn * yn, xy = xn * yn;
            const double r2 = xx + yy;
            const double rad = 1.0 + r2 * (k1 + r2 * (k2 + r2 * k3));
            const double drad = k1 + r2 * (2.0 * k2 + r2 * 3.0 * k3);
            const double txy = 2.0 * xy, qx = r2 + 2.0 * xx, qy = r2 + 2.0 * yy;
            const double mx = xn * rad + p1 * txy + p2 * qx;
            const double my = yn * rad + p1 * qy + p2 * txy;
            const double xd_x = rad + 2.0 * xx * drad + 2.0 * p1 * yn + 6.0 * p2 * xn;
            const double xd_y = txy * drad + 2.0 * p1 * xn + 2.0 * p2 * yn;
            const double yd_y = rad + 2.0 * yy * drad + 6.0 * p1 * yn + 2.0 * p2 * xn;
            const double ru = th[0] * mx + th[2] - uo, rv = th[1] * my + th[3] - vo;
            const double sw = valid ? huber_sqrt_weight(ru * ru + rv * rv, a.huber_delta) : 0.0;
            const double fxs = sw * th[0], fys = sw * th[1];
            const double eu = fxs * iz, ev = fys * iz;
            const double u0 = eu * xd_x, u1 = eu * xd_y, u2 = -(u0 * xn + u1 * yn);
            const double v0 = ev * xd_y, v1 = ev * yd_y, v2 = -(v0 * xn + v1 * yn);
            const double Xu = fxs * xn, Yv = fys * yn;
            const double r4 = r2 * r2, r6 = r4 * r2;
            double so[NCR];
            su[0] = sw * mx;                so[0] = sw * my;
            su[1] = sw;                     so[1] = sw;
            su[2 + OCV5_K1 - 4] = Xu * r2; su[2 + OCV5_K2 - 4] = Xu * r4; su[2 + OCV5_K3 - 4] = Xu * r6;
            su[2 + OCV5_P1 - 4] = fxs * txy; su[2 + OCV5_P2 - 4] = fxs * qx;
            so[2 + OCV5_K1 - 4] = Yv * r2; so[2 + OCV5_K2 - 4] = Yv * r4; so[2 + OCV5_K3 - 4] = Yv * r6;
            so[2 + OCV5_P2 - 4] = fys * qy; so[2 + OCV5_P1 - 4] = fys * txy;          // d v / d p1, d v / d p2 in the partner's (mirrored) columns
            rows_finish(rx, ry, rz, u0, u1, u2, v0, v1, v2, sw, ru, rv, su, so, sv);
      }
    };
    auto gram = [&](const double* su, const double* sv) {       // su: the lane's own corner, sv: its neighbour's - both rows of the lane's kind
        constexpr Map nm = Map();
#pragma unroll
        for (int i = 0; i < NCR; ++i) {
#pragma unroll
            for (int j = i; j < NCR; ++j) {
                if (!g2_is_rep<MODEL>(i, j)) continue;                  // (KB4's Hankel block: one pair per anti-diagonal)
                const int e = nm.num[i][j];
                acc[e] = __builtin_fma(su[i], su[j], acc[e]);
                acc[e] = __builtin_fma(sv[i], sv[j], acc[e]);
            }
        }
    };
    for (int base = 0; base < nmax; base += LPF) {
        const bool valid = base + cl < n;
        double X = pX, Y = pY, Z = pZ, uo = role ? pV : pU, vo = role ? pU : pV;
        asm volatile("" : "+v"(X), "+v"(Y), "+v"(Z), "+v"(uo), "+v"(vo));      // converted BEFORE the next rows are requested: they land in the same registers
        if (base + LPF < nmax) {
            const int cn = base + LPF + cl;
            const uint32_t gn = ob0 + 4u * (uint32_t)(cn < n ? cn : 0);
            pX = ldf(a.x, gn); pY = ldf(a.y, gn); pZ = ldf(a.z, gn); pU = ldf(a.u, gn); pV = ldf(a.v, gn);
        }
        double su[NCR], sv[NCR];
        G2_CYC_BEGIN();
        if constexpr (MODEL == kUCM || MODEL == kEUCM) rows_ucm(X, Y, Z, uo, vo, valid, su, sv);
        else if constexpr (MODEL == kKB4) rows_kb4(X, Y, Z, uo, vo, valid, su, sv);
        else rows_ocv5(X, Y, Z, uo, vo, valid, su, sv);
        G2_CYC(2);
        gram(su, sv);
        G2_CYC(3);
#if defined(CCAL_STAMPS) && CCAL_STAMPS >= 2
        g2_cyc[4] += 1;
#endif
    }

    G2_STAMP(2);
    const int pbase = mwave * G;              // first of the wavefront's frames among the launch's / the bin's
    // fused elimination: what its tail needs from memory is requested now, behind the reductions
    int slot_t = 0;
    double mc_t = 0.0;
    if constexpr (!GEN) {
        if (fuse && active) { slot_t = BIN ? slot_e : (a.slot_ident ? f : a.obs_slot[f]); if (gl == 0) mc_t = a.mc_f[f]; }
    }
    // the frame's LPF partial row Grams -> one Gram of the block, through LDS, slice by slice: an item = one entry of the
    // full triangle = the sum over the frame's u lanes and / or v lanes of one row-local entry
    constexpr Map cm = Map();                                  // slice boundaries: compile-time
    constexpr int NQM = (G * Map::max_items() + 63) / 64;      // items per lane and slice
    constexpr int RLEN = GEN ? gen_e_off(K) : REC_;              // doubles of a record that this kernel forms (GEN: E^T went out in the prologue)
    constexpr int RSTR = GEN ? ((RLEN + 1) & ~1) : GS_;           // stride of a frame's record in LDS
    static_assert(G * RSTR <= RED, "the frames' records fit the reduction buffer");
    double res[NS][NQM];
    int dst_a[NS][NQM], dst_b[NS][NQM];       // where the item goes in the LDS records (second place: the mirror entry of A, or -1): worked out
                                              // here, in the shadow of the LDS reads, so that the assembly pass is one or two writes per item
    static_assert(cm.first[NS] == NEF, "one item per entry of the full triangle");
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        wsync();
#pragma unroll
        for (int t = 0; t < CH; ++t) { const int e = s * CH + t; if (e < NER) red[lane * LS + t] = acc[e < NER ? e : 0]; }
        wsync();
        if (s == 0) G2_EP(0);
        const int i0 = cm.first[s], ni = cm.first[s + 1] - i0;
#pragma unroll
        for (int q = 0; q < NQM; ++q) {
            const int idx = lane + 64 * q;
            double sum = 0.0;
            if (idx < G * ni) {
                const int g = idx / ni, it = i0 + (idx - g * ni);
                const unsigned long long tv = tab[it];
                const uint32_t sc = (uint32_t)tv;
                const uint32_t m = (uint32_t)(tv >> 32);
                dst_a[s][q] = g * RSTR + (int)(m & 0xffff);
                dst_b[s][q] = (m >> 16) != 0xffff ? g * RSTR + (int)(m >> 16) : -1;
                double s_u = 0.0, s_v = 0.0;
                const double* src_u = red + (g * LPF) * LS + ((int)(sc & 0xff) - s * CH);              // the frame's u lanes: the even ones
                const double* src_v = red + (g * LPF + 1) * LS + ((int)((sc >> 8) & 0xff) - s * CH);   // its v lanes hold the entry under its mirrored number
#pragma unroll
                for (int l = 0; l < L2; ++l) { s_u += src_u[2 * l * LS]; s_v += src_v[2 * l * LS]; }
                sum = ((sc >> 16) & 1 ? s_u : 0.0) + ((sc >> 17) & 1 ? ((sc >> 18) & 1 ? -s_v : s_v) : 0.0);
            }
            res[s][q] = sum;
        }
        if (s == 0) G2_EP(1);
        if (s == NS - 1) G2_EP(2);
    }
    wsync();
    G2_STAMP(3);
    // The records are ASSEMBLED IN LDS - C (21) | [B|g] (6 x K1) | A (K1 x K1) | J_l (9), or the general loop's record
    // C (36) | [B|g]^T | A - with one or two ds_write per item (every sum is in registers: the reduction buffer is free), and
    // go out to HBM as whole records with coalesced stores (they were 8-byte stores scattered by a table: ~180 per lane)
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int ni = cm.first[s + 1] - cm.first[s];
#pragma unroll
        for (int q = 0; q < NQM; ++q) {
            if (lane + 64 * q < G * ni) {
                const double v = res[s][q];
                red[dst_a[s][q]] = v;
                if (dst_b[s][q] >= 0) red[dst_b[s][q]] = v;
            }
        }
    }
    // the frame's left Jacobian (phi -> rvec map of the elimination)
    if constexpr (!GEN) { if (lane_ok) for (int e = gl; e < 9; e += LPF) red[grp * GS_ + praw_jl_off(K) + e] = fc[FC_A + e]; }
    wsync();
    if (keep_rec) {
        double* const praw_es = a.praw[es];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (pbase + g >= nfr) break;                               // wave-uniform
            const int ff = frame0 + pbase + g;
            double* rec = praw_es + (GEN ? a.rec_off[a.list[ff]] : (int64_t)ff * a.PRAW);
            for (int e = lane; e < RLEN; e += 64) rec[e] = red[g * RSTR + e];
            if (!GEN && lane == 0) a.cost_f[ff] = red[g * RSTR + 21 + 6 * K1 + K * K1 + K];      // r x r
        }
    }
    if constexpr (!GEN) {
        if (fuse) {
            if (keep_rec) wsync();                                     // the tail reuses the records' rows
            G2_STAMP(4);
            gram_fused_tail<K, LPF>(a, st_lam_schur(), red, ITER ? fcw : a.partial + (int64_t)wrow * fused_red_size(K), grp, gl, lane_ok, active, slot_t, es, mc_t G2_EP_PTR);
            iter_row_out();
        }
    }
    G2_STAMP(5);
    G2_STAMPS_FLUSH;
}

template <int MODEL, bool OF, int LPF, bool GEN>
__global__ __launch_bounds__(64 * CCAL_GRAMV_WPB, CCAL_G2_MINW(MODEL)) void k_gram2(const FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    gram2_body<MODEL, OF, LPF, GEN, false>(a, smem, blockIdx.x, 0, a.n_obs);
}
// Single-launch groups (ITER, see gram2_body): eight wavefronts per workgroup, one workgroup per compute unit.  SORTED: the problem has
// a table of its frames sorted by corner count (ragged frames, gram2_bin_plan) - the launch walks it as ONE bin of LPF lanes per frame:
// a wavefront's five frames are alike, so the first 1 024 wavefronts (the large frames) share their SIMDs with the short ones of the
// second half instead of every wavefront running to the trip count of the largest of five frames picked in table order.
template <int MODEL, bool OF, int LPF, bool SORTED>
__global__ __launch_bounds__(64 * 8, CCAL_G2_MINW(MODEL)) void k_gram2i(const FusedArgs) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    // (the argument block where it lies - the kernarg segment -, not the by-value parameter: the unpack branch indexes it at run time,
    //  which would make the compiler copy all of it into scratch; k_gram1v, ccal_kernels_fused.hip)
    const FusedArgs& a = *(const FusedArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    gram2_body<MODEL, OF, LPF, false, SORTED, true>(a, smem, blockIdx.x, 0, a.n_obs);
}
// Ragged frames (real sessions hold 24 .. 144 corners per frame, /root/reference/src/data_loader.rs:15): ONE launch whose workgroups
// belong to bins of frames with different lanes per frame (gram2_bin_plan), so that a wavefront's trip count is what ITS frames need
// and not the largest frame's of a group picked in table order.  Bin b = workgroups bin_wg0[b] .. bin_wg0[b + 1] - 1.  (A FOLDED plan is
// one bin over a table whose second half runs smallest first: nothing for the kernel to know - a wavefront's trip count is the largest
// of ITS frames wherever they sit.)
template <int MODEL, bool OF>
__global__ __launch_bounds__(64 * CCAL_GRAMV_WPB, CCAL_G2_MINW(MODEL)) void k_gram2b(const FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    // (constant indices only: a dynamically indexed kernel argument would be copied to scratch)
    int wg0 = 0, frame0 = a.bin_first[0], nfr = a.bin_count[0], lpf = a.bin_lpf[0];
#pragma unroll
    for (int i = 1; i < kGramMaxBins; ++i)
        if (i < a.n_bins && (int)blockIdx.x >= a.bin_wg0[i]) { wg0 = a.bin_wg0[i]; frame0 = a.bin_first[i]; nfr = a.bin_count[i]; lpf = a.bin_lpf[i]; }
    const int wg = blockIdx.x - wg0;
    switch (lpf) {
        case 6: gram2_body<MODEL, OF, 6, false, true>(a, smem, wg, frame0, nfr); break;
        case 8: gram2_body<MODEL, OF, 8, false, true>(a, smem, wg, frame0, nfr); break;
        case 12: gram2_body<MODEL, OF, 12, false, true>(a, smem, wg, frame0, nfr); break;
        case 16: gram2_body<MODEL, OF, 16, false, true>(a, smem, wg, frame0, nfr); break;
#ifdef CCAL_G2_MORE_LPF      // experiment builds: lane mappings that leave lanes idle (10: six frames on 60 lanes, 20: three)
        case 10: gram2_body<MODEL, OF, 10, false, true>(a, smem, wg, frame0, nfr); break;
        case 20: gram2_body<MODEL, OF, 20, false, true>(a, smem, wg, frame0, nfr); break;
#endif
        default: gram2_body<MODEL, OF, 32, false, true>(a, smem, wg, frame0, nfr); break;
    }
}

// The same for one camera's (or, merged, every camera's) blocks of a rig: the launch's LIST of observation frames comes sorted by corner
// count (normal_ws_ensure_general), the bins are ranges of it; records, slots and cameras are looked up by observation frame as ever.
template <int MODEL, bool OF>
__global__ __launch_bounds__(64 * CCAL_GRAMV_WPB, CCAL_G2_MINW(MODEL)) void k_gram2g(const FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    int wg0 = 0, frame0 = a.bin_first[0], nfr = a.bin_count[0], lpf = a.bin_lpf[0];
#pragma unroll
    for (int i = 1; i < kGramMaxBins; ++i)
        if (i < a.n_bins && (int)blockIdx.x >= a.bin_wg0[i]) { wg0 = a.bin_wg0[i]; frame0 = a.bin_first[i]; nfr = a.bin_count[i]; lpf = a.bin_lpf[i]; }
    const int wg = blockIdx.x - wg0;
    switch (lpf) {
        case 6: gram2_body<MODEL, OF, 6, true, false>(a, smem, wg, frame0, nfr); break;
        case 8: gram2_body<MODEL, OF, 8, true, false>(a, smem, wg, frame0, nfr); break;
        case 12: gram2_body<MODEL, OF, 12, true, false>(a, smem, wg, frame0, nfr); break;
        case 16: gram2_body<MODEL, OF, 16, true, false>(a, smem, wg, frame0, nfr); break;
        default: gram2_body<MODEL, OF, 32, true, false>(a, smem, wg, frame0, nfr); break;
    }
}

#ifdef CCAL_G2_PROBE      // register-allocation probes (developer): a few instantiations, no launchers
template __global__ void k_gram2<kEUCM, false, 12, false>(const FusedArgs);
template __global__ void k_gram2<kKB4, false, 12, false>(const FusedArgs);
template __global__ void k_gram2<kOCV5, false, 12, false>(const FusedArgs);
template __global__ void k_gram2<kOCV5, true, 12, true>(const FusedArgs);
template __global__ void k_gram2b<kEUCM, false>(const FusedArgs);
template __global__ void k_gram2b<kKB4, false>(const FusedArgs);
template __global__ void k_gram2i<kEUCM, false, 12, false>(const FusedArgs);
template __global__ void k_gram2i<kEUCM, false, 12, true>(const FusedArgs);
}  // namespace ccal
#else
template <int MODEL, bool OF, int LPF, bool GEN>
static hipError_t launch_gram2_l(const FusedArgs& a, hipStream_t s) {
    constexpr int NS = g2_slices<MODEL>();
    using Map = RowMap<MODEL, OF, GEN, NS>;
    constexpr int G = 64 / LPF, K = block_dim(MODEL, OF, false) - 6, K1 = K + 1;
    constexpr int LS = Map::CH | 1;
    constexpr int GS_ = (praw_jl_off(K) + 9 + 6 * K1 + 1) & ~1;
    constexpr int RED = (!GEN && G * GS_ > 64 * LS) ? G * GS_ : 64 * LS;
    constexpr int WSL = (G * (FC_N0P + 12) + RED + Map::NEF + 1) & ~1;
    const size_t lds = sizeof(double) * WSL * CCAL_GRAMV_WPB;
    void (*kern)(const FusedArgs) = k_gram2<MODEL, OF, LPF, GEN>;
    static DynLdsGuard lds_guard;
    if (hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds, lds_guard); e != hipSuccess) return e;
    const int fpb = G * CCAL_GRAMV_WPB;
    if (a.n_obs <= 0) return hipSuccess;
    hipLaunchKernelGGL(kern, dim3((a.n_obs + fpb - 1) / fpb), dim3(64 * CCAL_GRAMV_WPB), lds, s, a);
    return hipGetLastError();
}

// Lanes per frame (even: 64, 32, 16, 12, 8, 6): the cost model of gram_lanes_per_frame (ccal_kernels_fused.hip).
// `force` (FusedArgs::lpf_force: a developer switch of the second library) overrides; mappings whose wavefronts would not fit
// the rows of the partial-sum buffer (`max_waves`, single-camera loop: one row per wavefront) are left out.
static int gram2_lanes_per_frame(int n_obs, int avg_corners, bool two_per_simd, bool gen, int force, int64_t max_waves, int share) {
    static const int cand[6] = { 64, 32, 16, 12, 8, 6 };
    for (int c : cand) if (force == c) return c;
    int best = 6;
    double best_cost = 1e300;
    for (int i = 0; i < 6; ++i) {
        const int lpf = cand[i], g = 64 / lpf;
        const int64_t waves = ((int64_t)n_obs + g - 1) / g;
        if (((waves + CCAL_GRAMV_WPB - 1) / CCAL_GRAMV_WPB) * CCAL_GRAMV_WPB > max_waves && lpf != 6) continue;
        const int passes = (std::max(avg_corners, 1) + lpf - 1) / lpf;
        // prologue + reductions + elimination, in passes; the general loop's launches (GEN) have no elimination in their tail
        // (two EUCM cameras x 10 000 frames in one launch, whole build: 6 lanes 77.0 us, 8: 81.5, 12: 83.1, 16: 85.0)
        // - but what they measure at 20 000 frames is a larger fixed cost per wavefront (the record goes to HBM, the
        // occupancy term below is optimistic beyond four wavefronts per SIMD)
        // (single camera, 20 000 frames, whole build: 6 lanes 55.1 us, 8: 59.1, 12: 63.8; 50 000 frames: 133.8, 123.5, 136.8)
        const double c0 = gen ? (lpf == 6 ? 8.0 : 7.0) : (lpf == 6 ? 7.0 : 6.0);
        double occ;
        const int simds = std::max(1024 / std::max(share, 1), 64);      // side-by-side sessions (ccal_solve_batch) share the chip
        const double nw = (double)waves / (double)simds;
        if (two_per_simd) occ = nw <= 1.0 ? 1.0 : (nw <= 2.0 ? 1.0 + 0.3 * (nw - 1.0) : 0.65 + 0.43 * nw);
        else occ = (double)((waves + simds - 1) / simds);
        const double cost = occ * (c0 + passes);
        if (cost < best_cost) { best_cost = cost; best = lpf; }
    }
    return best;
}

// ---- ragged frames: the bins of one launch ------------------------------------------------------------------------------------
// What a wavefront costs the SIMD it runs on, in corner passes: its trip count + what its prologue, reductions and fused elimination are
// worth - which grow with the frames it holds (4 + frames per wavefront).  Fitted, with the pair rule below, to the PICKS among 198 hand-made
// plans (equalised T = 3 .. 20, folds of 8 .. 32 lanes) over sixteen problem sizes from 2 000 to 30 000 ragged frames
// (profiles/r06/ab_g2_plans_folded.txt, ab_g2_plans_between.txt, ab_g2_plans_small.txt; a constant 6 ranked equalised plans of many narrow
// bins in front of the 16-lane fold that wins from 4 000 to 8 000 frames by 5-9 %).
static inline double g2_wave_cost(int lpf, int passes) { return 4.0 + (double)(64 / lpf) + (double)passes; }
// The time of a launch whose wavefronts cost c[0 .. W) in launch order on `simds` SIMDs: wavefront i, i + simds, i + 2 simds ... share
// a SIMD (the dispatcher fills the chip in launch order).  Two resident wavefronts each advance at 1 / 1.3 of the rate of one alone
// (gram2_lanes_per_frame's occupancy fit: a pair of equals takes 0.65 of the summed cost), a wavefront whose partner has finished runs
// at full rate: a pair costs max + 0.3 min - EQUAL partners are the cheapest way to get through a given sum (measured, folded against
// equalised plans: profiles/r06/ab_g2_fold.txt).  One per SIMD (KB4 / OPENCV5: 296+ registers): one after the other.
static double g2_launch_cost(const std::vector<double>& c, int simds, bool two_per_simd) {
    double worst = 0.0;
    const int W = (int)c.size();
    for (int i = 0; i < simds && i < W; ++i) {
        double t = 0.0;
        if (!two_per_simd) { for (int k = i; k < W; k += simds) t += c[(size_t)k]; }
        else {
            double r0 = c[(size_t)i], r1 = 0.0;
            int k = i + simds;
            if (k < W) { r1 = c[(size_t)k]; k += simds; }
            while (r0 > 0.0 && r1 > 0.0) {
                const double m = std::min(r0, r1);
                t += 1.3 * m; r0 -= m; r1 -= m;
                if (r0 <= 0.0 && k < W) { r0 = c[(size_t)k]; k += simds; }
                if (r1 <= 0.0 && k < W) { r1 = c[(size_t)k]; k += simds; }
            }
            t += std::max(r0, r1);
        }
        worst = std::max(worst, t);
    }
    return worst;
}
static int g2_iter_lpf(int n_obs, int avg_corners);
GramBins gram2_bin_plan(const int64_t* off, int n_obs, bool two_per_simd, std::vector<int32_t>* order, bool rig_list) {
    GramBins none;
#ifdef CCAL_G2_NO_BINS        // A/B builds (tools/build_tu_variants.sh ccal_kernels_gram2 "nobins:-DCCAL_G2_NO_BINS"): the launch without bins
    return none;
#endif
#ifdef CCAL_G2_PLAN_ENV       // experiment builds: the plan from the environment, "lpf:frames+lpf:frames+..." in launch order over the frames sorted by size
    int only_T = 0, only_fold = 0;                // "T:9": the equalised plan of that trip-count limit; "fold:16": the folded plan of that mapping
    if (const char* e = std::getenv("CCAL_G2_PLAN"); e && !std::strncmp(e, "T:", 2)) only_T = std::atoi(e + 2);
    else if (e && !std::strncmp(e, "fold:", 5)) only_fold = std::atoi(e + 5);
    else if (e && off && order && n_obs > 0) {
        std::vector<int32_t> ord((size_t)n_obs);
        for (int o = 0; o < n_obs; ++o) ord[(size_t)o] = o;
        std::stable_sort(ord.begin(), ord.end(), [&](int32_t x, int32_t y) { return off[x + 1] - off[x] > off[y + 1] - off[y]; });
        GramBins gb;
        int pos = 0, wgs = 0;
        while (*e && gb.n_bins < kGramMaxBins && pos < n_obs) {
            const int lpf = std::atoi(e); while (*e && *e != ':') ++e; if (*e) ++e;
            int cnt = std::atoi(e); while (*e && *e != ',' && *e != '+') ++e; if (*e) ++e;
            cnt = std::min(cnt > 0 ? cnt : n_obs, n_obs - pos);
            const int k = gb.n_bins++, g = 64 / lpf;
            gb.lpf[k] = lpf; gb.first[k] = pos; gb.count[k] = cnt; gb.wg0[k] = wgs;
            wgs += ((cnt + g - 1) / g + CCAL_GRAMV_WPB - 1) / CCAL_GRAMV_WPB;
            pos += cnt;
        }
        gb.wg0[gb.n_bins] = wgs;
        if (pos == n_obs) { *order = std::move(ord); return gb; }
    }
#endif
    if (!off || n_obs < 2000 || !order) return none;                 // (below: k_gram1v's single-launch groups, one wavefront per SIMD)
    static const int lpfs[kGramMaxBins] = { 32, 16, 12, 8, 6 };       // launch order: the bins of the large frames first
    const int simds = 1024;
    int64_t nmin = INT64_MAX, nmax = 0, total = 0;
    for (int o = 0; o < n_obs; ++o) { const int64_t n = off[o + 1] - off[o]; nmin = std::min(nmin, n); nmax = std::max(nmax, n); total += n; }
    if (nmin == nmax || nmax > 32 * 64) return none;                  // uniform frames: nothing to sort
    // frames by corner count, largest first (stable: ties in table order) - for every trip-count limit T the bins are contiguous ranges
    std::vector<int32_t> ord((size_t)n_obs);
    for (int o = 0; o < n_obs; ++o) ord[(size_t)o] = o;
    std::stable_sort(ord.begin(), ord.end(), [&](int32_t x, int32_t y) { return off[x + 1] - off[x] > off[y + 1] - off[y]; });
    auto cnt = [&](int pos) { return (int)(off[ord[(size_t)pos] + 1] - off[ord[(size_t)pos]]); };
    // the launch as it is without bins: one mapping for all frames (gram2_lanes_per_frame), frames in table order
    double cost_plain;
    {
        const int avg = (int)(total / std::max(n_obs, 1));
        const int lpf = gram2_lanes_per_frame(n_obs, avg, two_per_simd, false, 0, (int64_t)1 << 40, 1), g = 64 / lpf;
        std::vector<double> c;
        for (int o = 0; o < n_obs; o += g) {
            int64_t mx = 0;
            for (int q = o; q < std::min(n_obs, o + g); ++q) mx = std::max(mx, off[q + 1] - off[q]);
            c.push_back(g2_wave_cost(lpf, (int)((mx + lpf - 1) / lpf)));
        }
        cost_plain = g2_launch_cost(c, simds, two_per_simd);
    }
    GramBins best;
    double best_cost = 1e300;
    std::vector<double> c;
    for (int T = 2; T <= 64; ++T) {
        if ((nmax + 31) / 32 > T) continue;                          // the widest mapping cannot cover the largest frame in T passes
#ifdef CCAL_G2_PLAN_ENV
        if ((only_T && T != only_T) || only_fold) continue;
#endif
        GramBins gb;
        c.clear();
        int pos = 0, wgs = 0;
        for (int b = 0; b < kGramMaxBins && pos < n_obs; ++b) {
            const int lpf = lpfs[b], g = 64 / lpf;
            // this bin: the frames that need more than T passes with the next narrower mapping (the narrowest takes the rest)
            int end = pos;
            if (b == kGramMaxBins - 1) end = n_obs;
            else while (end < n_obs && cnt(end) > T * lpfs[b + 1]) ++end;
            if (end == pos) continue;
            const int k = gb.n_bins++;
            gb.lpf[k] = lpf; gb.first[k] = pos; gb.count[k] = end - pos; gb.wg0[k] = wgs;
            const int waves = (end - pos + g - 1) / g;
            for (int w = 0; w < waves; ++w) c.push_back(g2_wave_cost(lpf, (cnt(pos + w * g) + lpf - 1) / lpf));
            if (waves % CCAL_GRAMV_WPB) c.push_back(0.0);            // (a workgroup belongs to one bin: its idle second wavefront)
            wgs += (waves + CCAL_GRAMV_WPB - 1) / CCAL_GRAMV_WPB;
            pos = end;
        }
        gb.wg0[gb.n_bins] = wgs;
        const double cost = g2_launch_cost(c, simds, two_per_simd);
        if (cost < best_cost) { best_cost = cost; best = gb; }
    }
#ifndef CCAL_G2_NO_FOLD
    // Two wavefronts per SIMD: ONE bin whose table is FOLDED - the larger half of the frames largest first, then the smaller half
    // smallest first - so that the wavefronts that share a SIMD (i and i + simds of the launch; k_gram2i: w and w + 4 of a workgroup)
    // are a long one and a short one and every SIMD has the same work, whatever the distribution of the frame sizes.  Where the
    // mapping makes ~2 048 wavefronts of the frames this is the plan (8 000 / 16 000 / 20 000 frames U{24..144}: 26.7 / 37.6 / 45.5 us
    // against 28.6 / 41.4 / 48.4 for the best equalised plan).  Where the solve runs single-launch groups of k_gram2i (12 lanes) the
    // table is folded for THAT launch whatever the model says of the build (10 000 frames: build 30.5 against 30.3 us, GN 0.213
    // against 0.223 ms; profiles/r06/ab_g2_plans_folded.txt).
    bool forced = false;
    if (two_per_simd) {
        // two sizes ranges where the table's shape is not the model's to choose: the single-launch groups' 12 lanes, and 3 500 .. 8 192 frames
        // of clearly ragged sets, where the 16-lane fold was the best of every plan tried at 4 000, 5 000, 6 000, 7 000 and 8 000 frames
        // (20.8 / 23.0 / 24.5 / 25.8 / 26.7 us; the plain launch at 4 000 / 5 000: 22.5 / 25.3) and the model sees it only from 5 000
        const bool iter12 = !rig_list && g2_iter_lpf(n_obs, (int)(total / std::max(n_obs, 1))) == 12;       // (a rig's Gram lists: the model's choice only)
        const bool mid16 = !rig_list && !iter12 && n_obs > 3500 && n_obs <= 8192 && nmax <= 16 * 64 && (double)total <= 0.8 * (double)nmax * (double)n_obs;
        const int force_lpf = iter12 ? 12 : (mid16 ? 16 : 0);
        for (int b = 0; b < kGramMaxBins; ++b) {
            const int lpf = lpfs[b], g = 64 / lpf;
            const int waves = (n_obs + g - 1) / g;
            if (waves > 2 * simds || (nmax + lpf - 1) / lpf > 64) continue;
#ifndef CCAL_G2_PLAN_ENV
            if (lpf == 6) continue;                   // (20 000 frames: 45.8 us folded against 44.8 for the equalised plan the model ranks behind it)
            if (lpf == 32 && n_obs > 3500) continue;  // (4 000 frames: 22.9 us against 20.8 with 16 lanes - pairs of 32-lane wavefronts slow each other more than the model's 1.3)
#else
            if ((only_fold && lpf != only_fold) || only_T) continue;
#endif
            const int fold = std::min(n_obs, 4 * ((waves + 7) / 8) * g);           // (k_gram2i's split: the first four wavefronts of its workgroups)
            auto at = [&](int pos) { return pos < fold ? cnt(pos) : cnt(n_obs - 1 - (pos - fold)); };
            c.clear();
            for (int w = 0; w < waves; ++w) {
                int mx = 0;
                for (int q = w * g; q < std::min(n_obs, (w + 1) * g); ++q) mx = std::max(mx, at(q));
                c.push_back(g2_wave_cost(lpf, (mx + lpf - 1) / lpf));
            }
            double cost = g2_launch_cost(c, simds, true);
#ifndef CCAL_G2_PLAN_ENV
            if (force_lpf) { if (lpf != force_lpf) continue; cost = std::min(cost, best_cost); forced = mid16; }
#endif
            if (cost <= best_cost) {
                best_cost = cost;
                best = GramBins();
                best.n_bins = 1; best.lpf[0] = lpf; best.first[0] = 0; best.count[0] = n_obs; best.wg0[0] = 0;
                best.wg0[1] = (waves + CCAL_GRAMV_WPB - 1) / CCAL_GRAMV_WPB;
                best.fold = fold;
            }
        }
        (void)force_lpf;
    }
#endif
#ifdef CCAL_G2_PLAN_ENV
    if (std::getenv("CCAL_G2_PLAN_PRINT")) std::fprintf(stderr, "gram2_bin_plan %d frames: model cost %.2f (no bins: %.2f), %d bins, fold %d\n", n_obs, best_cost, cost_plain, best.n_bins, best.fold);
    if ((only_T || only_fold) && best.n_bins > 0) cost_plain = 1e300;
#endif
    if (best.n_bins == 0 || (!forced && best_cost > 0.97 * cost_plain)) return none;       // (the sorted table costs the prologue a dependent load: it has to pay)
    if (best.fold > 0) std::reverse(ord.begin() + best.fold, ord.end());
    *order = std::move(ord);
    return best;
}

// the binned launch: a.n_bins / a.bin_* filled in by the caller (single camera: make_fused_args, + a.bin_tab; rigs: launch_gram_dev, a.list sorted)
template <int MODEL, bool OF, bool GEN>
static hipError_t launch_gram2_binned(FusedArgs& a, hipStream_t s) {
    constexpr int NS = g2_slices<MODEL>();
    using Map = RowMap<MODEL, OF, GEN, NS>;
    constexpr int K = block_dim(MODEL, OF, false) - 6, K1 = K + 1;
    constexpr int LS = Map::CH | 1;
    constexpr int GS_ = (praw_jl_off(K) + 9 + 6 * K1 + 1) & ~1;
    size_t lds = 0;
    for (int b = 0; b < a.n_bins; ++b) {
        const int G = 64 / a.bin_lpf[b];
        const int RED = (!GEN && G * GS_ > 64 * LS) ? G * GS_ : 64 * LS;
        const int WSL = (G * (FC_N0P + 12) + RED + Map::NEF + 1) & ~1;
        lds = std::max(lds, sizeof(double) * WSL * CCAL_GRAMV_WPB);
    }
    void (*kern)(const FusedArgs) = GEN ? k_gram2g<MODEL, OF> : k_gram2b<MODEL, OF>;
    static DynLdsGuard lds_guard;
    if (hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds, lds_guard); e != hipSuccess) return e;
    const int wgs = a.bin_wg0[a.n_bins];
    if constexpr (!GEN) { a.n_part = wgs * CCAL_GRAMV_WPB; a.fuse_elim = 1; a.elim_fused = 1; }
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(64 * CCAL_GRAMV_WPB), lds, s, a);
    return hipGetLastError();
}

template <int MODEL, bool OF, bool GEN>
static hipError_t launch_gram2_t(FusedArgs& a, hipStream_t s) {
    if constexpr (!GEN) {
        // ragged frames: the problem came with a plan (ccal_problem_create); the fused elimination's rows must fit, and a forced
        // mapping (developer switch of the second library) means the plain launch
        if (a.n_bins > 0 && a.fuse_elim && !a.lpf_force && a.bin_wg0[a.n_bins] * CCAL_GRAMV_WPB <= a.part_cap) return launch_gram2_binned<MODEL, OF, false>(a, s);
    } else {
        if (a.n_bins > 0 && !a.lpf_force) return launch_gram2_binned<MODEL, OF, true>(a, s);      // rigs: the list came sorted, with its bins
    }
    const int lpf = gram2_lanes_per_frame(a.n_obs, a.avg_corners, CCAL_G2_MINW(MODEL) >= 2, GEN, a.lpf_force, (GEN || !a.fuse_elim) ? (int64_t)1 << 40 : a.part_cap, a.share);
    const int waves = ((a.n_obs + 64 / lpf - 1) / (64 / lpf) + CCAL_GRAMV_WPB - 1) / CCAL_GRAMV_WPB * CCAL_GRAMV_WPB;
    const bool fuse = !GEN && a.fuse_elim != 0 && waves <= a.part_cap;
    a.fuse_elim = fuse ? 1 : 0;
    a.elim_fused = fuse ? 1 : 0;
    if (fuse) a.n_part = waves;
    switch (lpf) {
        case 6: return launch_gram2_l<MODEL, OF, 6, GEN>(a, s);
        case 8: return launch_gram2_l<MODEL, OF, 8, GEN>(a, s);
        case 12: return launch_gram2_l<MODEL, OF, 12, GEN>(a, s);
        case 16: return launch_gram2_l<MODEL, OF, 16, GEN>(a, s);
        case 32: return launch_gram2_l<MODEL, OF, 32, GEN>(a, s);
        default: return launch_gram2_l<MODEL, OF, 64, GEN>(a, s);
    }
}
// ---- single-launch groups on k_gram2i ------------------------------------------------------------------------------------------
// Where it applies: UCM / EUCM (two wavefronts of 256 registers per SIMD), all wavefronts resident at once (<= 2 048) in <= 256
// workgroups of eight 12-lane wavefronts - and at least 224 of them: a workgroup of eight wavefronts takes a whole compute unit, so
// the form only pays where the launch fills the chip (profiles/r06/ab_g2_single_launch_groups.txt, GN with host pointers against
// k_gram1v's single-launch form: 10 000 frames 0.237 against 0.253 ms, 8 000 - 16 lanes - 0.212 = 0.211, 5 000: 0.188 against 0.180,
// 2 500: 0.168 against 0.147).  10 000 frames: 250 workgroups; the window is 8 960 .. 10 240 frames.
constexpr int kG2IterWpb = 8;
template <int MODEL, bool OF, int LPF>
static constexpr size_t g2_iter_lds() {
    constexpr int NS = g2_slices<MODEL>();
    using Map = RowMap<MODEL, OF, false, NS>;
    constexpr int G = 64 / LPF, K = block_dim(MODEL, OF, false) - 6, K1 = K + 1;
    constexpr int LS = Map::CH | 1, GS_ = (praw_jl_off(K) + 9 + 6 * K1 + 1) & ~1;
    constexpr int RED = (G * GS_ > 64 * LS) ? G * GS_ : 64 * LS;
    constexpr int WSL = (G * (FC_N0P + 12) + RED + Map::NEF + 1) & ~1;
    return sizeof(double) * WSL * kG2IterWpb;
}
template <int MODEL, bool OF>
static constexpr size_t g2_iter_static_lds() { return sizeof(HeadShared) + 4 * 2 * 64 * 8 + 512; }      // decision + the row sum's 4 x 2 x 64 + alignment
static int g2_iter_lpf(int n_obs, int avg_corners) {
    int best = 0;
    double best_cost = 1e300;
    for (int lpf : { 12 }) {
        const int g = 64 / lpf;
        const int64_t waves = ((int64_t)n_obs + g - 1) / g, wgs = (waves + kG2IterWpb - 1) / kG2IterWpb;
        if (waves > 2048 || wgs > 256 || wgs < 224) continue;
        const double nw = (double)waves / 1024.0;
        const double occ = nw <= 1.0 ? 1.0 : 1.0 + 0.3 * (nw - 1.0);
        const double cost = occ * (6.0 + (std::max(avg_corners, 1) + lpf - 1) / lpf);
        if (cost < best_cost) { best_cost = cost; best = lpf; }
    }
    return best;
}
template <int MODEL, bool OF>
static int g2_iter_rows_t(int n_obs, int avg_corners, FusedArgs* a, hipStream_t s, hipError_t* err) {
    const int lpf = g2_iter_lpf(n_obs, avg_corners);
    if (!lpf) return 0;
    const int g = 64 / lpf, rows = (int)((((int64_t)n_obs + g - 1) / g + kG2IterWpb - 1) / kG2IterWpb);
    const size_t lds = g2_iter_lds<MODEL, OF, 12>();
    if (lpf != 12 || lds + g2_iter_static_lds<MODEL, OF>() > 160 * 1024) return 0;
    if (a) {
        const bool sorted = a->n_bins > 0 && a->bin_tab != nullptr;      // ragged frames: the sorted table of ccal_problem_create
        void (*kern)(const FusedArgs) = sorted ? k_gram2i<MODEL, OF, 12, true> : k_gram2i<MODEL, OF, 12, false>;
        static DynLdsGuard guard_plain, guard_sorted;
        if (hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds, sorted ? guard_sorted : guard_plain); e != hipSuccess) { *err = e; return rows; }
        a->fuse_elim = 1; a->elim_fused = 1; a->n_part = rows;
        hipLaunchKernelGGL(kern, dim3(rows), dim3(64 * kG2IterWpb), lds, s, *a);
        *err = hipGetLastError();
    }
    return rows;
}
static int g2_iter_rows_m(int model, bool one_focal, int n_obs, int avg_corners, int K, FusedArgs* a, hipStream_t s, hipError_t* err) {
    if (n_obs < 2000 || K != block_dim(model, one_focal, false) - 6) return 0;
    switch (model * 2 + (one_focal ? 1 : 0)) {
        case 0: return g2_iter_rows_t<kUCM, false>(n_obs, avg_corners, a, s, err);
        case 1: return g2_iter_rows_t<kUCM, true>(n_obs, avg_corners, a, s, err);
        case 2: return g2_iter_rows_t<kEUCM, false>(n_obs, avg_corners, a, s, err);
        case 3: return g2_iter_rows_t<kEUCM, true>(n_obs, avg_corners, a, s, err);
        default: return 0;                                 // KB4 / OPENCV5: one wavefront per SIMD - k_gram1v's single-launch form
    }
}
int gram2_iter_rows(int model, bool one_focal, int n_obs, int avg_corners, int K) { return g2_iter_rows_m(model, one_focal, n_obs, avg_corners, K, nullptr, nullptr, nullptr); }
hipError_t launch_gram2_iter(int model, bool one_focal, FusedArgs& a, hipStream_t s) {
    hipError_t err = hipErrorInvalidValue;
    const int rows = g2_iter_rows_m(model, one_focal, a.n_obs, a.avg_corners, a.K, &a, s, &err);
    return rows > 0 ? err : hipErrorInvalidValue;
}

template <bool GEN>
static hipError_t launch_gram2_m(int model, bool one_focal, FusedArgs& a, hipStream_t s) {
    switch (model * 2 + (one_focal ? 1 : 0)) {
        case 0: return launch_gram2_t<kUCM, false, GEN>(a, s);
        case 1: return launch_gram2_t<kUCM, true, GEN>(a, s);
        case 2: return launch_gram2_t<kEUCM, false, GEN>(a, s);
        case 3: return launch_gram2_t<kEUCM, true, GEN>(a, s);
        case 4: return launch_gram2_t<kKB4, false, GEN>(a, s);
        case 5: return launch_gram2_t<kKB4, true, GEN>(a, s);
        case 6: return launch_gram2_t<kOCV5, false, GEN>(a, s);
        case 7: return launch_gram2_t<kOCV5, true, GEN>(a, s);
        default: return hipErrorInvalidValue;
    }
}
// single-camera loop; a.fuse_elim in: fusion allowed, out: fusion done (then a.n_part = rows of partial sums, a.elim_fused = 1)
hipError_t launch_gram2(int model, bool one_focal, FusedArgs& a, hipStream_t s) { return launch_gram2_m<false>(model, one_focal, a, s); }
// one camera's blocks of a multi-camera problem
hipError_t launch_gram2_general(int model, bool one_focal, const FusedArgs& a0, hipStream_t s) { FusedArgs a = a0; a.fuse_elim = 0; return launch_gram2_m<true>(model, one_focal, a, s); }

}  // namespace ccal
#endif
