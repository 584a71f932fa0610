// How far ahead must a serial-chain wave request its per-block 16 bytes so that a kernel saturating HBM beside it does not
// slow it down?  W one-wave workgroups; every lane walks its own stream (spread layout, as the chain kernels do), ~3000 cycles
// of dependent VALU per block, one 16-byte load requested D blocks ahead (D register sets, the loop unrolled D times so that
// no register copy - i.e. no early wait - is needed) and one 16-byte store per block.  Timed alone and beside a device copy
// that runs for the kernel's whole life.
// build: hipcc -O3 --offload-arch=gfx950 depth.hip -o depth
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))
template <int D, bool STORE>
__global__ __launch_bounds__(64) void slow(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, uint64_t stride, int iters, uint32_t *out, int prio) {
    if (prio) __builtin_amdgcn_s_setprio(3);
    const uint64_t s = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    const uint8_t *p = src + s * stride;
    uint8_t *q = dst + s * stride;
    uint4 f[D];
#pragma unroll
    for (int d = 0; d < D; d++) f[d] = *reinterpret_cast<const uint4 *>(p + 16 * d);
    uint32_t x = threadIdx.x, y = 0x9E3779B1u;
    const uint64_t c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();  // shader clock / constant 100 MHz clock
    for (int it = 0; it < iters; it += D) {  // iters is a multiple of 8
#pragma unroll
        for (int d = 0; d < D; d++) {
            const uint4 cur = f[d];
            const int nb = it + d + D < iters ? it + d + D : it + d;
            f[d] = *reinterpret_cast<const uint4 *>(p + 16ull * nb);  // D blocks ahead
            x ^= cur.x;
            R64(asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x) : "v"(y));)
            R64(asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x) : "v"(y));)
            R64(asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x) : "v"(y));)
            R64(asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x) : "v"(y));)
            R64(asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x) : "v"(y));)
            if (STORE) *reinterpret_cast<uint4 *>(q + 16ull * (it + d)) = make_uint4(x, cur.y, cur.z, cur.w);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const uint64_t c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        reinterpret_cast<uint64_t *>(out)[1] = c1 - c0; reinterpret_cast<uint64_t *>(out)[2] = r1 - r0;
    }
    if (x == 0x12345) out[0] = x;
}
__global__ __launch_bounds__(256) void copyk(const uint4 *__restrict__ a, uint4 *__restrict__ b, size_t n) {
    extern __shared__ uint32_t pad_lds[];  // (dynamic LDS only limits how many copy waves a CU holds)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
int main(int argc, char **argv) {
    const int prio = argc > 1 ? atoi(argv[1]) : 0;
    const size_t copy_lds = argc > 2 ? (size_t)atoi(argv[2]) * 1024 : 0;
    printf("slow kernel s_setprio %d, copy kernel with %zu KB of LDS per 256-thread workgroup\n", prio ? 3 : 0, copy_lds / 1024);
    const uint64_t GB = 1ull << 30, work = 16 * GB;
    uint8_t *sa, *sb, *ca, *cb; uint32_t *out;
    CHK(hipMalloc((void **)&sa, work)); CHK(hipMalloc((void **)&sb, work));
    const size_t cbytes = 2 * GB;
    CHK(hipMalloc((void **)&ca, cbytes)); CHK(hipMalloc((void **)&cb, cbytes)); CHK(hipMalloc((void **)&out, 64));
    CHK(hipMemset(sa, 1, work)); CHK(hipMemset(ca, 2, cbytes));
    hipStream_t s1, s2; CHK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CHK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t a0, a1; CHK(hipEventCreate(&a0)); CHK(hipEventCreate(&a1));
    const int iters = 16000;  // 250 KB per stream
    for (int W : {256, 1024}) {
        const uint64_t ns = (uint64_t)W * 64;
        const uint64_t stride = ((work / ns - 1024) & ~511ull) | 256;
        if (stride < (uint64_t)iters * 16 + 256 || ns * stride > work) { printf("W=%d does not fit\n", W); continue; }
        for (int cfg : {0, 2, 5}) {
            auto launch = [&](int it) {
                switch (cfg) {
                case 0: slow<1, true><<<W, 64, 0, s1>>>(sa, sb, stride, it, out, prio); break;
                case 1: slow<2, true><<<W, 64, 0, s1>>>(sa, sb, stride, it, out, prio); break;
                case 2: slow<4, true><<<W, 64, 0, s1>>>(sa, sb, stride, it, out, prio); break;
                case 3: slow<1, false><<<W, 64, 0, s1>>>(sa, sb, stride, it, out, prio); break;
                case 4: slow<2, false><<<W, 64, 0, s1>>>(sa, sb, stride, it, out, prio); break;
                default: slow<4, false><<<W, 64, 0, s1>>>(sa, sb, stride, it, out, prio); break;
                }
            };
            launch(64); CHK(hipStreamSynchronize(s1));
            CHK(hipEventRecord(a0, s1)); launch(iters); CHK(hipEventRecord(a1, s1)); CHK(hipEventSynchronize(a1));
            float alone; CHK(hipEventElapsedTime(&alone, a0, a1));
            uint64_t ck[3]; CHK(hipMemcpy(ck, out, 24, hipMemcpyDeviceToHost));
            const double mhz_alone = ck[2] ? 100.0 * (double)ck[1] / (double)ck[2] : 0;
            // beside a copy that keeps running: enqueue enough copies to outlast the kernel three times over, then drain
            CHK(hipEventRecord(a0, s1)); launch(iters); CHK(hipEventRecord(a1, s1));
            const int reps = (int)(3.0f * alone / 0.9f) + 4;
            for (int r = 0; r < reps; r++) copyk<<<8192, 256, copy_lds, s2>>>((const uint4 *)ca, (uint4 *)cb, cbytes / 16);
            CHK(hipEventSynchronize(a1));
            float beside; CHK(hipEventElapsedTime(&beside, a0, a1));
            CHK(hipStreamSynchronize(s2));
            CHK(hipMemcpy(ck, out, 24, hipMemcpyDeviceToHost));
            const double mhz_beside = ck[2] ? 100.0 * (double)ck[1] / (double)ck[2] : 0;
            printf("W=%5d depth %d %-9s: alone %7.2f ms = %5.0f ns/block, shader clock %4.0f MHz | beside a saturating copy %7.2f ms = %5.0f ns/block (%.2fx), shader clock %4.0f MHz\n", W,
                   cfg % 3 == 0 ? 1 : cfg % 3 == 1 ? 2 : 4, cfg < 3 ? "ld+st" : "ld only", alone, alone * 1e6 / iters, mhz_alone, beside, beside * 1e6 / iters, beside / alone, mhz_beside);
        }
    }
    return 0;
}
