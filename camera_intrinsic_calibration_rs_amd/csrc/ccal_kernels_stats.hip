// validation() statistics on the device (src/util.rs:778-795): gather one camera's reprojection errors, then
//     median = e_sorted[len / 2],   avg_99 = sum_{i < len * 99 / 100} e_sorted[i] / (len * 99 / 100).
// Neither needs the sorted array: both are ORDER STATISTICS.  An MSB-first radix select over the values' bit patterns (errors are
// >= 0: bit order == value order; six digits of 11 bits) finds, for both ranks at once, the exact value K at the rank and how many
// copies of it lie at or below the rank; the 99 % mean is then  sum_{e < K} e / len_99  +  copies x K / len_99,  accumulated in
// 128-bit FIXED POINT (2^-80 px): integer addition is associative, so the result does not depend on the order of the values at all
// (what the sort used to guarantee) - the same bits on one GPU and over the shards of several, whatever the frame order.  Hand-written: the
// header-only library sort this replaced (hipCUB radix sort) was 5 MB of the 13 MB library for this one small step.
#include "ccal_internal.hpp"

namespace ccal {

__global__ __launch_bounds__(256) void k_gather_err(const double* err, const int64_t* obs_off, const int32_t* list, int n_list,
                                                    const int64_t* dst_off, double* out) {
    // one workgroup per observation frame of the camera: contiguous copy
    const int o = list[blockIdx.x];
    const int64_t s = obs_off[o], n = obs_off[o + 1] - s, d = dst_off[blockIdx.x];
    for (int64_t i = threadIdx.x; i < n; i += 256) out[d + i] = err[s + i];
}

// ---- radix select ------------------------------------------------------------------------------------------------------
constexpr int kSelBits = 11, kSelBins = 1 << kSelBits, kSelPasses = 6;        // 5 x 11 + 9 bits
constexpr int kSelBlocks = 256;                                              // workgroups of the histogram / sum kernels
__host__ __device__ constexpr int sel_shift(int pass) { return pass < 5 ? 64 - kSelBits * (pass + 1) : 0; }
__host__ __device__ constexpr int sel_bins(int pass) { return pass < 5 ? kSelBins : 1 << 9; }
struct SelState { unsigned long long prefix[2]; long long rank[2]; };        // per target: the digits found so far (right-aligned), rank among the keys that share them
typedef unsigned __int128 u128;
struct SelWork {                                                              // the work area (device): zeroed in front of every use
    uint32_t hist[kSelPasses][2][kSelBins];
    SelState state[kSelPasses + 1];
    unsigned long long part_lo[kSelBlocks], part_hi[kSelBlocks];              // per-workgroup sums, fixed point
    uint32_t part_bad[kSelBlocks];                                            // a value that is not finite (or beyond 2^40 px)
    double out[2];
};
constexpr int kFixFrac = 80;                                                  // fractional bits of the fixed-point sums
// v >= 0 as an integer multiple of 2^-80 (bits below are dropped: < 2^-80 px per term); *bad: NaN, infinity or >= 2^40
__device__ __forceinline__ u128 to_fixed(const double v, bool* bad) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const int ex = (int)((b >> 52) & 0x7ff);
    const unsigned long long man = (b & 0xfffffffffffffull) | (ex ? (1ull << 52) : 0ull);
    const int sh = (ex ? ex : 1) - 1075 + kFixFrac;                            // value = man x 2^(ex - 1075)
    if ((b >> 63) || ex >= 1023 + 40) { *bad = true; return 0; }
    if (sh <= -64) return 0;
    return sh >= 0 ? ((u128)man << sh) : (u128)(man >> (-sh));
}
__device__ __forceinline__ double from_fixed(const u128 f) {
    const double hi = (double)(unsigned long long)(f >> 64), lo = (double)(unsigned long long)f;
    return (hi * 18446744073709551616.0 + lo) * 8.271806125530277e-25;          // x 2^-80
}
size_t order_stats_work_bytes() { return (sizeof(SelWork) + 255) & ~(size_t)255; }

// What the histogram of pass p - 1 says about both targets: every workgroup derives it for itself (the same counts: the same
// answer), workgroup 0 leaves it for the next launch.  state[p] = state at the ENTRY of pass p.  Wavefront t of the workgroup takes
// target t: lane l owns nb / 64 consecutive bins (requested together), a shuffle scan over the lanes' sums finds the lane whose
// range holds the rank, that lane walks its own bins in registers.
__device__ __forceinline__ SelState sel_advance(SelWork* w, const int pass, SelState* sh_state) {
    const int pp = pass - 1, nb = sel_bins(pp), per = nb / 64;              // 32 or 8 bins per lane
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < 2) {
        const int t = wave;
        const SelState prev = w->state[pp];
        const uint32_t* h = w->hist[pp][t] + lane * per;
        uint32_t c[32];
        uint32_t loc = 0;
#pragma unroll
        for (int i = 0; i < 32; ++i) { c[i] = i < per ? h[i] : 0u; loc += c[i]; }
        long long incl = loc;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const long long o = __shfl_up(incl, off, 64); if (lane >= off) incl += o; }
        const long long excl = incl - loc, r = prev.rank[t];
        // the lane whose range holds rank r (the last lane takes whatever lies beyond: cannot happen for a rank < count)
        const bool mine = (excl <= r && r < incl) || (lane == 63 && r >= incl);
        if (mine) {
            long long cum = excl;
            int b = 0;
#pragma unroll
            for (int i = 0; i < 31; ++i) if (b == i && i < per - 1 && cum + (long long)c[i] <= r) { cum += c[i]; ++b; }
            sh_state->prefix[t] = (prev.prefix[t] << (pp < 5 ? kSelBits : 9)) | (unsigned long long)(lane * per + b);
            sh_state->rank[t] = r - cum;
        }
    }
    __syncthreads();
    const SelState st = *sh_state;
    if (blockIdx.x == 0 && threadIdx.x == 0) w->state[pass] = st;
    return st;
}
__global__ __launch_bounds__(256) void k_sel_hist(const double* __restrict__ vals, const long long n, SelWork* w, const int pass,
                                                  const long long rank0, const long long rank1) {
    __shared__ uint32_t hist[2][kSelBins];
    __shared__ SelState shs;
    SelState st;
    if (pass == 0) {                        // the two ranks arrive in the argument block
        st.prefix[0] = 0; st.prefix[1] = 0; st.rank[0] = rank0; st.rank[1] = rank1;
        if (blockIdx.x == 0 && threadIdx.x == 0) w->state[0] = st;
    } else st = sel_advance(w, pass, &shs);
    const int nb = sel_bins(pass), shift = sel_shift(pass);
    const int hi_shift = pass == 0 ? 0 : (pass < 5 ? 64 - kSelBits * pass : 9);      // bits above this pass's digit: key >> hi_shift == prefix
    for (int b = threadIdx.x; b < 2 * kSelBins; b += 256) (&hist[0][0])[b] = 0;
    __syncthreads();
    const unsigned long long* keys = reinterpret_cast<const unsigned long long*>(vals);
    const long long stride = (long long)gridDim.x * 256;
    for (long long i0 = (long long)blockIdx.x * 256 + threadIdx.x; i0 < n; i0 += 4 * stride) {
        unsigned long long k[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) k[u] = i0 + u * stride < n ? keys[i0 + u * stride] : 0ull;       // four loads in flight
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i0 + u * stride >= n) continue;
            const uint32_t d = (uint32_t)(k[u] >> shift) & (uint32_t)(nb - 1);
            const unsigned long long hi = pass == 0 ? 0ull : (k[u] >> hi_shift);
            if (pass == 0 || hi == st.prefix[0]) atomicAdd(&hist[0][d], 1u);
            if (pass == 0 || hi == st.prefix[1]) atomicAdd(&hist[1][d], 1u);
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < 2 * kSelBins; b += 256) {
        const uint32_t c = (&hist[0][0])[b];
        if (c) atomicAdd(&w->hist[pass][0][0] + b, c);         // integer counts: the order of the additions does not matter
    }
}
// sum over the values below the 99 % key, each scaled (e x 1 / len_99, one rounding per value) and added as a 128-bit integer:
// per thread, then per workgroup (thread 0 adds its 256 threads' sums), one partial per workgroup; k_sel_finish adds the partials
__global__ __launch_bounds__(256) void k_sel_sum(const double* __restrict__ vals, const long long n, SelWork* w, const double inv) {
    __shared__ SelState shs;
    __shared__ unsigned long long s_lo[256], s_hi[256];
    __shared__ uint32_t s_bad[256];
    const SelState st = sel_advance(w, kSelPasses, &shs);
    const unsigned long long K99 = st.prefix[1];
    const unsigned long long* keys = reinterpret_cast<const unsigned long long*>(vals);
    u128 acc = 0;
    bool bad = false;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        if (keys[i] < K99) acc += to_fixed(vals[i] * inv, &bad);
    s_lo[threadIdx.x] = (unsigned long long)acc; s_hi[threadIdx.x] = (unsigned long long)(acc >> 64); s_bad[threadIdx.x] = bad ? 1u : 0u;
    __syncthreads();
    if (threadIdx.x == 0) {
        u128 t = 0;
        uint32_t anyb = 0;
        for (int i = 0; i < 256; ++i) { t += ((u128)s_hi[i] << 64) | (u128)s_lo[i]; anyb |= s_bad[i]; }
        w->part_lo[blockIdx.x] = (unsigned long long)t; w->part_hi[blockIdx.x] = (unsigned long long)(t >> 64); w->part_bad[blockIdx.x] = anyb;
    }
}
__global__ __launch_bounds__(64) void k_sel_finish(SelWork* w, const int n_part, const double inv, const int have99) {
    if (threadIdx.x != 0) return;
    const SelState st = w->state[kSelPasses];
    const double k99 = __longlong_as_double((long long)st.prefix[1]);
    w->out[0] = __longlong_as_double((long long)st.prefix[0]);                        // median = e_sorted[len / 2]
    double avg = 0.0;
    if (have99) {
        u128 t = 0;
        bool bad = false;
        for (int i = 0; i < n_part; ++i) { t += ((u128)w->part_hi[i] << 64) | (u128)w->part_lo[i]; bad = bad || w->part_bad[i] != 0; }
        t += to_fixed(k99 * inv, &bad) * (u128)(unsigned long long)(st.rank[1] + 1);   // + the copies of the key at or below the rank
        avg = bad ? (k99 != k99 ? k99 : __longlong_as_double(0x7ff0000000000000ll)) : from_fixed(t);
    }
    w->out[1] = avg;
}

// The whole selection in ONE launch of one workgroup for small arrays (a few thousand errors: nine launches of the general form are
// ~45 us of launch latency for no work; beyond ~8 000 values one workgroup's loads are the slower way - measured: 51 000 values 0.1 ms).
// Same digits, same keys, the same fixed-point sum: the same bits as the general form.
constexpr int kSelOneMax = 1 << 13;
__global__ __launch_bounds__(1024) void k_sel_one(const double* __restrict__ vals, const int n, SelWork* w, const double inv,
                                                  const long long rank0, const long long rank1, const int have99) {
    __shared__ uint32_t hist[2][kSelBins];
    __shared__ SelState shs;
    __shared__ unsigned long long s_lo[16], s_hi[16];
    __shared__ uint32_t s_bad[16];
    const unsigned long long* keys = reinterpret_cast<const unsigned long long*>(vals);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    SelState st;
    st.prefix[0] = 0; st.prefix[1] = 0; st.rank[0] = rank0; st.rank[1] = rank1;
    for (int pass = 0; pass < kSelPasses; ++pass) {
        const int nb = sel_bins(pass), shift = sel_shift(pass), per = nb / 64;
        const int hi_shift = pass == 0 ? 0 : (pass < 5 ? 64 - kSelBits * pass : 9);
        for (int b = threadIdx.x; b < 2 * kSelBins; b += 1024) (&hist[0][0])[b] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += 1024) {
            const unsigned long long k = keys[i];
            const uint32_t d = (uint32_t)(k >> shift) & (uint32_t)(nb - 1);
            const unsigned long long hi = pass == 0 ? 0ull : (k >> hi_shift);
            if (pass == 0 || hi == st.prefix[0]) atomicAdd(&hist[0][d], 1u);
            if (pass == 0 || hi == st.prefix[1]) atomicAdd(&hist[1][d], 1u);
        }
        __syncthreads();
        if (wave < 2) {                                     // sel_advance's search on the workgroup's own histogram
            const int t = wave;
            const uint32_t* h = hist[t] + lane * per;
            uint32_t c[32];
            uint32_t loc = 0;
#pragma unroll
            for (int i = 0; i < 32; ++i) { c[i] = i < per ? h[i] : 0u; loc += c[i]; }
            long long incl = loc;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const long long o = __shfl_up(incl, off, 64); if (lane >= off) incl += o; }
            const long long excl = incl - loc, r = st.rank[t];
            if ((excl <= r && r < incl) || (lane == 63 && r >= incl)) {
                long long cum = excl;
                int b = 0;
#pragma unroll
                for (int i = 0; i < 31; ++i) if (b == i && i < per - 1 && cum + (long long)c[i] <= r) { cum += c[i]; ++b; }
                shs.prefix[t] = (st.prefix[t] << (pass < 5 ? kSelBits : 9)) | (unsigned long long)(lane * per + b);
                shs.rank[t] = r - cum;
            }
        }
        __syncthreads();
        st = shs;
        __syncthreads();
    }
    const unsigned long long K99 = st.prefix[1];
    u128 acc = 0;
    bool bad = false;
    if (have99) for (int i = threadIdx.x; i < n; i += 1024) if (keys[i] < K99) acc += to_fixed(vals[i] * inv, &bad);
    // 128-bit integer sums: any order gives the same bits - lanes by shuffle, wavefronts through LDS
    unsigned long long lo = (unsigned long long)acc, hi = (unsigned long long)(acc >> 64);
    uint32_t bd = bad ? 1u : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long olo = __shfl_down(lo, off, 64), ohi = __shfl_down(hi, off, 64);
        const u128 t = (((u128)hi << 64) | lo) + (((u128)ohi << 64) | olo);
        lo = (unsigned long long)t; hi = (unsigned long long)(t >> 64);
        bd |= __shfl_down(bd, off, 64);
    }
    if (lane == 0) { s_lo[wave] = lo; s_hi[wave] = hi; s_bad[wave] = bd; }
    __syncthreads();
    if (threadIdx.x == 0) {
        u128 t = 0;
        bool anyb = false;
        for (int i = 0; i < 16; ++i) { t += ((u128)s_hi[i] << 64) | (u128)s_lo[i]; anyb = anyb || s_bad[i] != 0; }
        const double k99 = __longlong_as_double((long long)K99);
        double avg = 0.0;
        if (have99) {
            t += to_fixed(k99 * inv, &anyb) * (u128)(unsigned long long)(st.rank[1] + 1);
            avg = anyb ? (k99 != k99 ? k99 : __longlong_as_double(0x7ff0000000000000ll)) : from_fixed(t);
        }
        w->out[0] = __longlong_as_double((long long)st.prefix[0]);
        w->out[1] = avg;
    }
}

// median and 99 % mean of n non-negative values on the device; `work`: order_stats_work_bytes() of device memory
hipError_t order_stats_device(const double* d_vals, int64_t n, char* work, double* avg_99, double* median, hipStream_t s) {
    if (n <= 0 || !work) return hipErrorInvalidValue;
    SelWork* w = reinterpret_cast<SelWork*>(work);
    const int64_t n99 = n * 99 / 100;
    hipError_t e;
#define TRY(x) do { e = (x); if (e != hipSuccess) return e; } while (0)
    const double inv = n99 > 0 ? 1.0 / (double)n99 : 0.0;
    double h[2] = { 0.0, 0.0 };
    if (n <= kSelOneMax) {
        hipLaunchKernelGGL(k_sel_one, dim3(1), dim3(1024), 0, s, d_vals, (int)n, w, inv, (long long)(n / 2), (long long)(n99 > 0 ? n99 - 1 : 0), n99 > 0 ? 1 : 0);
        TRY(hipGetLastError());
        TRY(hipMemcpyAsync(h, w->out, sizeof h, hipMemcpyDeviceToHost, s));
        TRY(hipStreamSynchronize(s));
        *median = h[0]; *avg_99 = h[1];
        return hipSuccess;
    }
    TRY(hipMemsetAsync(w, 0, sizeof(SelWork), s));
    const int blocks = (int)std::min<int64_t>(kSelBlocks, (n + 256 * 8 - 1) / (256 * 8));
    for (int pass = 0; pass < kSelPasses; ++pass) {
        hipLaunchKernelGGL(k_sel_hist, dim3(blocks), dim3(256), 0, s, d_vals, (long long)n, w, pass, (long long)(n / 2), (long long)(n99 > 0 ? n99 - 1 : 0));
        TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(k_sel_sum, dim3(blocks), dim3(256), 0, s, d_vals, (long long)n, w, inv);
    TRY(hipGetLastError());
    hipLaunchKernelGGL(k_sel_finish, dim3(1), dim3(64), 0, s, w, blocks, inv, n99 > 0 ? 1 : 0);
    TRY(hipGetLastError());
    TRY(hipMemcpyAsync(h, w->out, sizeof h, hipMemcpyDeviceToHost, s));
    TRY(hipStreamSynchronize(s));
#undef TRY
    *median = h[0]; *avg_99 = h[1];
    return hipSuccess;
}

static hipError_t ensure_scratch(ccal_problem* p, size_t total) {
    if (p->scratch_bytes >= total) return hipSuccess;
    if (p->d_scratch) { (void)hipStreamSynchronize(p->ctx->stream); ctx_release(p->ctx, p->d_scratch, false); p->d_scratch = nullptr; p->scratch_bytes = 0; }
    const size_t want = std::max(total, problem_scratch_hint(p));
    const hipError_t e = ctx_dev_alloc(p->ctx, (void**)&p->d_scratch, want);
    if (e == hipSuccess) p->scratch_bytes = want;
    return e;
}
static inline size_t up256(size_t b) { return (b + 255) & ~(size_t)255; }

// One camera's errors out of the per-corner errors of the whole problem (device), packed in observation-frame order into the
// problem's scratch block [offsets | values | room for the statistics' work area] (*d_out: the values, valid until the next call
// that uses the scratch; *n_out doubles; complete when the function returns).  A camera without corners: *d_out = NULL, *n_out = 0.
static hipError_t gather_camera_errors(ccal_problem* p, int cam, const double* d_err, double** d_out, int64_t* n_out, hipStream_t s,
                                       std::vector<int64_t>& dst, bool sync) {
    *d_out = nullptr; *n_out = 0;
    const CamLayout& cl = p->cams[cam];
    const int n_list = (int)cl.obs.size();
    dst.assign((size_t)n_list + 1, 0);
    for (int i = 0; i < n_list; ++i) dst[i + 1] = dst[i] + (p->h_obs_off[cl.obs[i] + 1] - p->h_obs_off[cl.obs[i]]);
    const int64_t n = dst[n_list];
    if (n <= 0) return hipSuccess;
    const size_t b_off = up256((size_t)(n_list + 1) * sizeof(int64_t)), b_val = up256((size_t)n * sizeof(double));
    hipError_t e = ensure_scratch(p, b_off + b_val + order_stats_work_bytes());
    if (e != hipSuccess) return e;
    int64_t* d_dst = reinterpret_cast<int64_t*>(p->d_scratch);
    double* d_a = reinterpret_cast<double*>(p->d_scratch + b_off);
    e = hipMemcpyAsync(d_dst, dst.data(), (size_t)(n_list + 1) * sizeof(int64_t), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) { hipLaunchKernelGGL(k_gather_err, dim3(n_list), dim3(256), 0, s, d_err, p->d_obs_off, cl.d_obs, n_list, d_dst, d_a); e = hipGetLastError(); }
    if (e == hipSuccess && sync) e = hipStreamSynchronize(s);       // (the multi-GPU form copies d_a across devices next; else the caller's own
                                                                    // synchronise covers `dst`, which it keeps until then)
    if (e != hipSuccess) return e;
    *d_out = d_a; *n_out = n;
    return hipSuccess;
}
hipError_t camera_errors_device(ccal_problem* p, int cam, const double* d_err, double** d_out, int64_t* n_out, hipStream_t s) {
    std::vector<int64_t> dst;
    return gather_camera_errors(p, cam, d_err, d_out, n_out, s, dst, true);
}

// d_err: per-corner errors of the whole problem (device).  Returns the two statistics of camera `cam`: the gather above, then the
// selection on the gathered values (the multi-GPU form runs the same selection on the shards' values side by side: the same values
// in the same order, the same kernels - the same bits).  Every temporary is a slice of the problem's scratch block.
hipError_t validation_stats_device(ccal_problem* p, int cam, const double* d_err, double* avg_99, double* median, hipStream_t s) {
    double* d_a = nullptr;
    int64_t n = 0;
    std::vector<int64_t> dst;                               // (read by an asynchronous copy: alive until order_stats_device has synchronised)
    hipError_t e = gather_camera_errors(p, cam, d_err, &d_a, &n, s, dst, false);
    if (e != hipSuccess) return e;
    if (n <= 0) return hipErrorInvalidValue;
    char* work = reinterpret_cast<char*>(d_a) + up256((size_t)n * sizeof(double));
    return order_stats_device(d_a, n, work, avg_99, median, s);
}

// the multi-GPU form: block = [values (n) | work area], sized by order_stats_block_bytes and kept by the caller between calls
size_t order_stats_block_bytes(int64_t n, hipStream_t) { return n <= 0 ? 0 : up256((size_t)n * sizeof(double)) + order_stats_work_bytes(); }
hipError_t order_stats_block(char* block, size_t block_bytes, int64_t n, double* avg_99, double* median, hipStream_t s) {
    if (n <= 0 || !block || order_stats_block_bytes(n, s) > block_bytes) return hipErrorInvalidValue;
    return order_stats_device(reinterpret_cast<const double*>(block), n, block + up256((size_t)n * sizeof(double)), avg_99, median, s);
}

}  // namespace ccal
