// Developer check: does a kernel argument struct of 136 bytes arrive intact (member at offset 128), also through one
// s_load_dwordx16 at offset 0x48?  It does - this ruled the argument block out when the elimination kernels misbehaved with a
// 136-byte SchurArgs; the cause was a miscompiled select of two kernel-argument pointers.
//   hipcc --offload-arch=gfx950 -O3 -o kernarg136.bin kernarg136.hip && ./kernarg136.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
struct St;
struct Args {
    const double* G; const int64_t* slot_desc; const int32_t* slot_off; const int32_t* caminfo; int32_t n_cams;
    int32_t n_slots, K, RB, PF, n_pw; int32_t STG;
    double lambda, min_diag, max_diag;
    double* partial; double* pf; const double* mc_slot;
    const St* st; const double* G2;
    const int64_t* extra;          // offset 128
};
__global__ __launch_bounds__(256, 4) void k(const Args a, unsigned long long* out) {
    // every member is used, as in the kernel that misbehaved: the compiler then fetches the tail of the block with one
    // s_load_dwordx16 at offset 0x48 (72 .. 135), not 64-byte aligned
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        out[0] = (unsigned long long)a.extra; out[1] = (unsigned long long)a.G2; out[2] = (unsigned long long)a.st; out[3] = (unsigned long long)a.n_pw;
        out[4] = (unsigned long long)a.partial; out[5] = (unsigned long long)a.pf; out[6] = (unsigned long long)a.mc_slot;
        out[7] = (unsigned long long)__double_as_longlong(a.lambda); out[8] = (unsigned long long)__double_as_longlong(a.min_diag); out[9] = (unsigned long long)__double_as_longlong(a.max_diag);
        out[10] = (unsigned long long)a.G; out[11] = (unsigned long long)a.slot_desc; out[12] = (unsigned long long)a.slot_off; out[13] = (unsigned long long)a.caminfo;
        out[14] = (unsigned long long)(a.n_cams + a.n_slots + a.K + a.RB + a.PF + a.STG);
        // the load the compiler chose for the kernel that misbehaved: 16 dwords at offset 0x48 (bytes 72 .. 135)
        typedef unsigned int u16v __attribute__((ext_vector_type(16)));
        auto kp = __builtin_amdgcn_kernarg_segment_ptr();
        u16v v;
        asm volatile("s_load_dwordx16 %0, %1, 0x48\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(kp));
        out[15] = (unsigned long long)v[14] | ((unsigned long long)v[15] << 32);      // bytes 128 .. 135 = extra
        out[16] = (unsigned long long)v[12] | ((unsigned long long)v[13] << 32);      // bytes 120 .. 127 = G2
        out[17] = (unsigned long long)v[0] | ((unsigned long long)v[1] << 32);        // bytes 72 .. 79 = min_diag
        out[18] = (unsigned long long)(size_t)kp;
    }
}
int main() {
    printf("sizeof(Args) = %zu, offsetof(extra) = %zu\n", sizeof(Args), offsetof(Args, extra));
    unsigned long long* d; hipMalloc(&d, 256);
    Args a = {};
    a.partial = (double*)0x0101010101010101ull; a.pf = (double*)0x0202020202020202ull; a.mc_slot = (const double*)0x0303030303030303ull;
    a.lambda = 1.5; a.min_diag = 2.5; a.max_diag = 3.5; a.G = (const double*)0x0404040404040404ull;
    a.extra = (const int64_t*)0x1111222233334444ull; a.G2 = (const double*)0x5555666677778888ull; a.st = (const St*)0x9999aaaabbbbccccull; a.n_pw = 4096;
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipLaunchKernelGGL(k, dim3(4), dim3(256), 1024, s, a, d);
    hipStreamSynchronize(s);
    unsigned long long h[20]; hipMemcpy(h, d, 160, hipMemcpyDeviceToHost);
    printf("x16 @0x48: extra %llx  G2 %llx  min_diag %llx   kernarg segment at %llx (mod 64 = %llu)\n", h[15], h[16], h[17], h[18], h[18] % 64);
    printf("partial %llx pf %llx mc %llx lambda %llx min %llx max %llx G %llx\n", h[4], h[5], h[6], h[7], h[8], h[9], h[10]);
    printf("extra %llx  G2 %llx  st %llx  n_pw %llu  -> %s\n", h[0], h[1], h[2], h[3], (h[0] == 0x1111222233334444ull && h[1] == 0x5555666677778888ull) ? "intact" : "CORRUPTED");
    return 0;
}
