// What kind of neighbour slows a latency-bound wave of dependent VALU work (the shape of a serial-chain wave)?  W one-wave
// workgroups run ~340 dependent v_mad per block and touch NO memory; beside them, on a second stream, a neighbour kernel that fills
// every wave slot (8192 workgroups x 256 threads, grid-stride) and does one of:
//   copy     16-byte loads and stores, saturating HBM          sleep    nothing but s_sleep
//   load     loads only (sum kept in a register)                valu     independent v_mad, no memory
//   store    stores only                                        lds      ds_read / ds_write on 8 KB per workgroup
// Each neighbour also runs with 80 KB of (unused) dynamic LDS per workgroup, i.e. 2 of its waves per SIMD instead of 8.
// build: hipcc -O3 --offload-arch=gfx950 neighbour.hip -o neighbour
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))
__global__ __launch_bounds__(64) void chain(int iters, uint32_t *out) {
    uint32_t x = threadIdx.x, y = 0x9E3779B1u;
    for (int it = 0; it < iters; it++) {
        R64(asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x) : "v"(y));)
        R64(asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x) : "v"(y));)
        R64(asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x) : "v"(y));)
        R64(asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x) : "v"(y));)
        R64(asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x) : "v"(y));)
    }
    if (x == 0x12345) out[0] = x;
}
template <int KIND>
__global__ __launch_bounds__(256) void neighbour(const uint4 *__restrict__ a, uint4 *__restrict__ b, size_t n, int reps, uint32_t *out) {
    extern __shared__ uint32_t lds[];
    uint32_t acc = threadIdx.x;
    for (int r = 0; r < reps; r++) {
        if (KIND == 3) { for (int k = 0; k < 2000; k++) __builtin_amdgcn_s_sleep(100); continue; }
        if (KIND == 4) { uint32_t y = acc | 1; for (int k = 0; k < 4000; k++) { R16(asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(acc) : "v"(y));) } continue; }
        if (KIND == 5) { for (int k = 0; k < 20000; k++) { lds[(threadIdx.x + k) & 2047] = acc; acc += lds[(threadIdx.x * 7 + k) & 2047]; } continue; }
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
            if (KIND == 0) b[i] = a[i];
            if (KIND == 1) { const uint4 v = a[i]; acc += v.x ^ v.w; }
            if (KIND == 2) b[i] = make_uint4(acc, 1, 2, 3);
        }
    }
    if (acc == 0x12345) out[1] = acc;
}
int main() {
    const size_t cbytes = 2ull << 30;
    uint8_t *ca, *cb; uint32_t *out;
    CHK(hipMalloc((void **)&ca, cbytes)); CHK(hipMalloc((void **)&cb, cbytes)); CHK(hipMalloc((void **)&out, 64));
    CHK(hipMemset(ca, 2, cbytes));
    hipStream_t s1, s2; CHK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CHK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t a0, a1; CHK(hipEventCreate(&a0)); CHK(hipEventCreate(&a1));
    const int iters = 16000;
    const char *names[6] = {"copy", "load", "store", "sleep", "valu", "lds"};
    for (int W : {256, 1024}) {
        chain<<<W, 64, 0, s1>>>(100, out); CHK(hipStreamSynchronize(s1));
        CHK(hipEventRecord(a0, s1)); chain<<<W, 64, 0, s1>>>(iters, out); CHK(hipEventRecord(a1, s1)); CHK(hipEventSynchronize(a1));
        float alone; CHK(hipEventElapsedTime(&alone, a0, a1));
        printf("W=%4d chain waves alone: %.2f ms = %.0f ns per block\n", W, alone, alone * 1e6 / iters);
        for (int kind = 0; kind < 6; kind++) {
            for (size_t pad : {(size_t)8192, (size_t)80 * 1024}) {
                auto nb = [&](int reps) {
                    switch (kind) {
                    case 0: neighbour<0><<<8192, 256, pad, s2>>>((const uint4 *)ca, (uint4 *)cb, cbytes / 16, reps, out); break;
                    case 1: neighbour<1><<<8192, 256, pad, s2>>>((const uint4 *)ca, (uint4 *)cb, cbytes / 16, reps, out); break;
                    case 2: neighbour<2><<<8192, 256, pad, s2>>>((const uint4 *)ca, (uint4 *)cb, cbytes / 16, reps, out); break;
                    case 3: neighbour<3><<<8192, 256, pad, s2>>>((const uint4 *)ca, (uint4 *)cb, cbytes / 16, reps, out); break;
                    case 4: neighbour<4><<<8192, 256, pad, s2>>>((const uint4 *)ca, (uint4 *)cb, cbytes / 16, reps, out); break;
                    default: neighbour<5><<<8192, 256, pad, s2>>>((const uint4 *)ca, (uint4 *)cb, cbytes / 16, reps, out); break;
                    }
                };
                // how long is one rep of this neighbour alone?
                nb(1); CHK(hipStreamSynchronize(s2));
                hipEvent_t b0, b1; CHK(hipEventCreate(&b0)); CHK(hipEventCreate(&b1));
                CHK(hipEventRecord(b0, s2)); nb(1); CHK(hipEventRecord(b1, s2)); CHK(hipEventSynchronize(b1));
                float one; CHK(hipEventElapsedTime(&one, b0, b1));
                const int reps = (int)(3.0f * alone / one) + 2;  // outlast the chain kernel even if it runs 3x slower
                CHK(hipEventRecord(a0, s1)); chain<<<W, 64, 0, s1>>>(iters, out); CHK(hipEventRecord(a1, s1));
                nb(reps);
                CHK(hipEventSynchronize(a1));
                float beside; CHK(hipEventElapsedTime(&beside, a0, a1));
                const bool still = hipStreamQuery(s2) == hipErrorNotReady;
                CHK(hipStreamSynchronize(s2));
                printf("   beside %-5s (%2zu KB LDS per workgroup, one pass alone %.2f ms): %7.2f ms = %5.0f ns per block (%.2fx)%s\n", names[kind], pad / 1024, one, beside,
                       beside * 1e6 / iters, beside / alone, still ? "" : "  [neighbour ended first]");
            }
        }
    }
    return 0;
}
