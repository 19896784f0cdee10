// validation() statistics on the device (src/util.rs:778-795): gather one camera's reprojection errors,
// radix-sort them (hipCUB/rocPRIM header-only device primitives), median = e[len/2],
// avg_99 = sum_{i < len*99/100} e_i / (len*99/100).
#include <hipcub/hipcub.hpp>

#include "ccal_internal.hpp"

namespace ccal {

__global__ __launch_bounds__(256) void k_gather_err(const double* err, const int64_t* obs_off, const int32_t* list, int n_list,
                                                    const int64_t* dst_off, double* out) {
    // one workgroup per observation frame of the camera: contiguous copy
    const int o = list[blockIdx.x];
    const int64_t s = obs_off[o], n = obs_off[o + 1] - s, d = dst_off[blockIdx.x];
    for (int64_t i = threadIdx.x; i < n; i += 256) out[d + i] = err[s + i];
}
__global__ __launch_bounds__(256) void k_scale(double* v, int64_t n, double inv) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) v[i] *= inv;
}

static hipError_t ensure_scratch(ccal_problem* p, size_t total) {
    if (p->scratch_bytes >= total) return hipSuccess;
    if (p->d_scratch) { (void)hipFree(p->d_scratch); p->d_scratch = nullptr; p->scratch_bytes = 0; }
    const size_t want = std::max(total, problem_scratch_hint(p));
    const hipError_t e = hipMalloc((void**)&p->d_scratch, want);
    if (e == hipSuccess) p->scratch_bytes = want;
    return e;
}
static inline size_t up256(size_t b) { return (b + 255) & ~(size_t)255; }

// One camera's errors out of the per-corner errors of the whole problem (device), packed in observation-frame order into the
// problem's scratch block (*d_out: a slice of it, valid until the next call that uses the scratch; *n_out doubles; complete when
// the function returns).  A camera without corners: *d_out = NULL, *n_out = 0.
hipError_t camera_errors_device(ccal_problem* p, int cam, const double* d_err, double** d_out, int64_t* n_out, hipStream_t s) {
    *d_out = nullptr; *n_out = 0;
    const CamLayout& cl = p->cams[cam];
    const int n_list = (int)cl.obs.size();
    std::vector<int64_t> dst(n_list + 1, 0);
    for (int i = 0; i < n_list; ++i) dst[i + 1] = dst[i] + (p->h_obs_off[cl.obs[i] + 1] - p->h_obs_off[cl.obs[i]]);
    const int64_t n = dst[n_list];
    if (n <= 0) return hipSuccess;
    const size_t b_off = up256((size_t)(n_list + 1) * sizeof(int64_t));
    hipError_t e = ensure_scratch(p, b_off + up256((size_t)n * sizeof(double)));
    if (e != hipSuccess) return e;
    int64_t* d_dst = reinterpret_cast<int64_t*>(p->d_scratch);
    double* d_a = reinterpret_cast<double*>(p->d_scratch + b_off);
    e = hipMemcpyAsync(d_dst, dst.data(), (size_t)(n_list + 1) * sizeof(int64_t), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) { hipLaunchKernelGGL(k_gather_err, dim3(n_list), dim3(256), 0, s, d_err, p->d_obs_off, cl.d_obs, n_list, d_dst, d_a); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipStreamSynchronize(s);       // (dst is a host vector of this frame; the caller copies d_a across devices next)
    if (e != hipSuccess) return e;
    *d_out = d_a; *n_out = n;
    return hipSuccess;
}

// d_err: per-corner errors of the whole problem (device).  Returns the two statistics of camera `cam`.  The same kernels in the
// same order as camera_errors_device + sorted_stats_device (the multi-GPU form's pieces: the same bits), with every temporary a
// slice of the problem's scratch block (kept for the next call): five hipMalloc / hipFree pairs - a hipFree waits for the device -
// were half of validation()'s 0.18 ms at 600 frames.
hipError_t validation_stats_device(ccal_problem* p, int cam, const double* d_err, double* avg_99, double* median, hipStream_t s) {
    const CamLayout& cl = p->cams[cam];
    const int n_list = (int)cl.obs.size();
    std::vector<int64_t> dst(n_list + 1, 0);
    for (int i = 0; i < n_list; ++i) dst[i + 1] = dst[i] + (p->h_obs_off[cl.obs[i] + 1] - p->h_obs_off[cl.obs[i]]);
    const int64_t n = dst[n_list];
    if (n <= 0) return hipErrorInvalidValue;
    const int64_t n99 = n * 99 / 100;
    size_t tmp_sort = 0, tmp_red = 0;
    hipError_t e;
#define TRY(x) do { e = (x); if (e != hipSuccess) return e; } while (0)
    TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_sort, (const double*)nullptr, (double*)nullptr, (int)n, 0, 64, s));
    TRY(hipcub::DeviceReduce::Sum(nullptr, tmp_red, (const double*)nullptr, (double*)nullptr, (int)std::max<int64_t>(n99, 1), s));
    const size_t tmp_bytes = std::max<size_t>(std::max(tmp_sort, tmp_red), 16);
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t b_off = up((size_t)(n_list + 1) * sizeof(int64_t)), b_val = up((size_t)n * sizeof(double)), b_sum = 256, b_tmp = up(tmp_bytes);
    const size_t total = b_off + 2 * b_val + b_sum + b_tmp;
    TRY(ensure_scratch(p, total));
    char* q = p->d_scratch;
    int64_t* d_dst = reinterpret_cast<int64_t*>(q); q += b_off;
    double* d_a = reinterpret_cast<double*>(q); q += b_val;
    double* d_b = reinterpret_cast<double*>(q); q += b_val;
    double* d_sum = reinterpret_cast<double*>(q); q += b_sum;
    void* d_tmp = q;
    double h[2] = { 0.0, 0.0 };
    size_t tb = tmp_bytes;
    TRY(hipMemcpyAsync(d_dst, dst.data(), (size_t)(n_list + 1) * sizeof(int64_t), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_gather_err, dim3(n_list), dim3(256), 0, s, d_err, p->d_obs_off, cl.d_obs, n_list, d_dst, d_a);
    TRY(hipGetLastError());
    TRY(hipcub::DeviceRadixSort::SortKeys(d_tmp, tb, d_a, d_b, (int)n, 0, 64, s));       // errors are >= 0: bit order == value order
    TRY(hipMemcpyAsync(&h[0], d_b + n / 2, sizeof(double), hipMemcpyDeviceToHost, s));          // median = e[len / 2]
    if (n99 > 0) {
        hipLaunchKernelGGL(k_scale, dim3((unsigned)((n99 + 255) / 256)), dim3(256), 0, s, d_b, n99, 1.0 / (double)n99);   // e_i / len_99, then sum
        TRY(hipGetLastError());
        tb = tmp_bytes;
        TRY(hipcub::DeviceReduce::Sum(d_tmp, tb, d_b, d_sum, (int)n99, s));
        TRY(hipMemcpyAsync(&h[1], d_sum, sizeof(double), hipMemcpyDeviceToHost, s));
    }
    TRY(hipStreamSynchronize(s));          // (dst and h are host objects of this frame)
#undef TRY
    *median = h[0]; *avg_99 = h[1];
    return hipSuccess;
}

// median = e[len / 2] and avg_99 = sum_{i < len * 99 / 100} e_i / (len * 99 / 100) of n non-negative values on the device.  The
// multi-GPU form gathers the shards' values into the front of one block and calls this: same values, same sorted order, same
// reduction - the same bits as on one GPU.  block = [values (n) | sort buffer (n) | sum | hipCUB's temporary], sized by
// sorted_stats_scratch_bytes and kept by the caller between calls (no allocation here).
static hipError_t sorted_stats_sizes(int64_t n, hipStream_t s, size_t* b_val, size_t* tmp_bytes) {
    size_t tmp_sort = 0, tmp_red = 0;
    const int64_t n99 = n * 99 / 100;
    hipError_t e = hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_sort, (const double*)nullptr, (double*)nullptr, (int)n, 0, 64, s);
    if (e == hipSuccess) e = hipcub::DeviceReduce::Sum(nullptr, tmp_red, (const double*)nullptr, (double*)nullptr, (int)std::max<int64_t>(n99, 1), s);
    *b_val = up256((size_t)n * sizeof(double)); *tmp_bytes = std::max<size_t>(std::max(tmp_sort, tmp_red), 16);
    return e;
}
size_t sorted_stats_scratch_bytes(int64_t n, hipStream_t s) {
    size_t b_val = 0, tmp = 0;
    if (n <= 0 || sorted_stats_sizes(n, s, &b_val, &tmp) != hipSuccess) return 0;
    return 2 * b_val + 256 + up256(tmp);
}
hipError_t sorted_stats_device(char* block, size_t block_bytes, int64_t n, double* avg_99, double* median, hipStream_t s) {
    if (n <= 0 || !block) return hipErrorInvalidValue;
    size_t b_val = 0, tmp_bytes = 0;
    hipError_t e;
#define TRY(x) do { e = (x); if (e != hipSuccess) return e; } while (0)
    TRY(sorted_stats_sizes(n, s, &b_val, &tmp_bytes));
    if (2 * b_val + 256 + up256(tmp_bytes) > block_bytes) return hipErrorInvalidValue;
    double* d_a = reinterpret_cast<double*>(block);
    double* d_b = reinterpret_cast<double*>(block + b_val);
    double* d_sum = reinterpret_cast<double*>(block + 2 * b_val);
    void* d_tmp = block + 2 * b_val + 256;
    const int64_t n99 = n * 99 / 100;
    double h[2] = { 0.0, 0.0 };
    size_t tb = tmp_bytes;
    TRY(hipcub::DeviceRadixSort::SortKeys(d_tmp, tb, d_a, d_b, (int)n, 0, 64, s));       // errors are >= 0: bit order == value order
    TRY(hipMemcpyAsync(&h[0], d_b + n / 2, sizeof(double), hipMemcpyDeviceToHost, s));          // median = e[len / 2]
    if (n99 > 0) {
        hipLaunchKernelGGL(k_scale, dim3((unsigned)((n99 + 255) / 256)), dim3(256), 0, s, d_b, n99, 1.0 / (double)n99);   // e_i / len_99, then sum
        TRY(hipGetLastError());
        tb = tmp_bytes;
        TRY(hipcub::DeviceReduce::Sum(d_tmp, tb, d_b, d_sum, (int)n99, s));
        TRY(hipMemcpyAsync(&h[1], d_sum, sizeof(double), hipMemcpyDeviceToHost, s));
    }
    TRY(hipStreamSynchronize(s));
#undef TRY
    *median = h[0]; *avg_99 = h[1];
    return hipSuccess;
}

}  // namespace ccal
