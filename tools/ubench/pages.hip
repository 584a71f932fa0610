// Do many SLOW per-lane streams - the memory shape of the serial-chain kernels: every lane of a wave walks a stream of its own,
// 16 bytes every ~1.3 us, and the 64 streams of a wave lie in 64 different regions of a >100 GB workspace - throttle a streaming
// kernel beside them (and themselves) through address translation / DRAM page locality rather than through bandwidth?
//   slow<0> "spread":       stream s lives at base + s * stride        (stride ~ 0.4 .. 3 MB: every lane of a wave in its own 2 MB page)
//   slow<1> "interleaved":  block b of stream s at base + (b * nstreams + s) * 16  (a wave's 64 loads = 1 KB contiguous)
//   slow<2>, slow<3> "spread, 64 / 128-byte units": the layout of slow<0>, but a lane moves 4 (8) consecutive 16-byte pieces
//                           back to back every 4th (8th) iteration instead of one piece per iteration
// Both move the same bytes (one 16-byte load one block ahead + one 16-byte store per lane and iteration, ~SPIN dependent VALU
// instructions in between).  Measured: time per iteration of the slow kernel alone, a 2 GB device copy alone, and the same copy
// started while the slow kernel is running.
// build: hipcc -O3 --offload-arch=gfx950 pages.hip -o pages
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))
template <int MODE>
__global__ __launch_bounds__(64) void slow(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, uint64_t nstreams, uint64_t stride, int iters, uint32_t *out, uint32_t wrap) {
    const uint64_t s = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    // wrap (spread modes): a stream re-walks its first `wrap` units of 128 bytes: the same access pattern inside a smaller footprint
    auto addr = [&](int it) -> uint64_t { return MODE != 1 ? s * stride + (uint64_t)((uint32_t)it % wrap) * 16 : ((uint64_t)it * nstreams + s) * 16; };
    constexpr int U = MODE == 2 ? 4 : MODE == 3 ? 8 : 1;  // pieces per unit
    uint4 nx[U], cu[U], ob[U];
#pragma unroll
    for (int u = 0; u < U; u++) { nx[u] = *reinterpret_cast<const uint4 *>(src + addr(u)); cu[u] = nx[u]; ob[u] = nx[u]; }
    uint32_t x = threadIdx.x, y = 0x9E3779B1u;
    for (int it = 0; it < iters; it++) {
        uint4 cur;
        if (U == 1) {
            cur = nx[0];
            nx[0] = *reinterpret_cast<const uint4 *>(src + addr(it + 1 < iters ? it + 1 : it));  // one block ahead
        } else {
            if (it % U == 0) {  // (iters is a multiple of 8) the unit requested a unit ago lands, the next one is requested
#pragma unroll
                for (int u = 0; u < U; u++) cu[u] = nx[u];
                const int nb = it + U < iters ? it + U : it;
#pragma unroll
                for (int u = 0; u < U; u++) nx[u] = *reinterpret_cast<const uint4 *>(src + addr(nb + u));
            }
            cur = cu[0];
#pragma unroll
            for (int u = 0; u + 1 < U; u++) cu[u] = cu[u + 1];  // (rotate: piece it % U)
        }
        x ^= cur.x;
        R64(asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x) : "v"(y));)
        R64(asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x) : "v"(y));)
        R64(asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x) : "v"(y));)
        R64(asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x) : "v"(y));)
        R64(asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x) : "v"(y));)
        R16(asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x) : "v"(y));)  // ~340 dependent instructions ~ 3000 cycles: one 8-step block of a chain
        if (U == 1) *reinterpret_cast<uint4 *>(dst + addr(it)) = make_uint4(x, cur.y, cur.z, cur.w);
        else {
#pragma unroll
            for (int u = 0; u + 1 < U; u++) ob[u] = ob[u + 1];
            ob[U - 1] = make_uint4(x, cur.y, cur.z, cur.w);
            if (it % U == U - 1) {
#pragma unroll
                for (int u = 0; u < U; u++) *reinterpret_cast<uint4 *>(dst + addr(it - (U - 1) + u)) = ob[u];
            }
        }
    }
    if (x == 0x12345) out[0] = x;
}
__global__ __launch_bounds__(256) void copyk(const uint4 *__restrict__ a, uint4 *__restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
int main(int argc, char **argv) {
    const uint64_t GB = 1ull << 30;
    const uint64_t work = (argc > 1 ? (uint64_t)atoi(argv[1]) : 48) * GB;  // bytes of each of the two stream areas
    const uint64_t foot = argc > 2 ? (uint64_t)atoi(argv[2]) * GB : work;  // the spread streams are confined to the first `foot` bytes (they wrap inside their stride)
    uint8_t *sa, *sb, *ca, *cb; uint32_t *out;
    CHK(hipMalloc((void **)&sa, work + (1 << 20))); CHK(hipMalloc((void **)&sb, work + (1 << 20)));
    const size_t cbytes = 2 * GB;
    CHK(hipMalloc((void **)&ca, cbytes)); CHK(hipMalloc((void **)&cb, cbytes)); CHK(hipMalloc((void **)&out, 4));
    CHK(hipMemset(sa, 1, work)); CHK(hipMemset(ca, 2, cbytes));
    hipStream_t s1, s2; CHK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CHK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    auto copy_ms = [&](int reps) {
        CHK(hipEventRecord(e0, s2));
        for (int r = 0; r < reps; r++) copyk<<<8192, 256, 0, s2>>>((const uint4 *)ca, (uint4 *)cb, cbytes / 16);
        CHK(hipEventRecord(e1, s2)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps;
    };
    copy_ms(2);
    const float alone = copy_ms(8);
    printf("stream areas 2 x %llu GB, spread footprint %llu GB\n", (unsigned long long)(work / GB), (unsigned long long)(foot / GB)); printf("copy alone: %.3f ms per 2 GB = %.0f GB/s (read + write)\n", alone, 2.0 * cbytes / alone / 1e6);
    for (int W : {512, 1024, 1536, 2048}) {
        const uint64_t ns = (uint64_t)W * 64;
        int iters = 20000;
        if ((uint64_t)iters * ns * 16 > work) iters = (int)(work / (ns * 16));
        uint64_t stride = ((foot / ns - 1024) & ~511ull) | 256;
        uint32_t wrap = (uint32_t)((stride - 256) / 16) & ~7u;  // pieces of a stream before it wraps (a multiple of 8)
        if (wrap >= (uint32_t)iters) wrap = 1u << 30;  // odd multiple of 256 bytes (no channel aliasing between the streams), ns * stride < work
        if (foot / ns < 2048 || wrap < 8 || ns * stride > work || (uint64_t)iters * ns * 16 > work) { printf("W=%d: streams do not fit\n", W); continue; }
        iters &= ~7;
        for (int mode = 0; mode < 4; mode++) {
            auto launch = [&](int it) {
                if (mode == 0) slow<0><<<W, 64, 0, s1>>>(sa, sb, ns, stride, it, out, wrap);
                else if (mode == 1) slow<1><<<W, 64, 0, s1>>>(sa, sb, ns, stride, it, out, wrap);
                else if (mode == 2) slow<2><<<W, 64, 0, s1>>>(sa, sb, ns, stride, it, out, wrap);
                else slow<3><<<W, 64, 0, s1>>>(sa, sb, ns, stride, it, out, wrap);
            };
            launch(200); CHK(hipStreamSynchronize(s1));
            hipEvent_t a0, a1; CHK(hipEventCreate(&a0)); CHK(hipEventCreate(&a1));
            CHK(hipEventRecord(a0, s1)); launch(iters); CHK(hipEventRecord(a1, s1)); CHK(hipEventSynchronize(a1));
            float ms; CHK(hipEventElapsedTime(&ms, a0, a1));
            // the copy beside it: the slow kernel runs `ms`; start it, wait a little, time copies inside its lifetime
            CHK(hipEventRecord(a0, s1)); launch(iters); CHK(hipEventRecord(a1, s1));
            copy_ms(1);
            const int reps = (int)(0.6f * ms / alone / 3) > 1 ? (int)(0.6f * ms / alone / 3) : 1;
            const float beside = copy_ms(reps);
            const bool still = hipEventQuery(a1) == hipErrorNotReady;
            CHK(hipEventSynchronize(a1));
            float ms2; CHK(hipEventElapsedTime(&ms2, a0, a1));
            printf("W=%5d %-11s stride %7.0f KB: slow alone %7.2f ms = %6.0f ns/iter | copy beside %.3f ms = %5.0f GB/s (%.2fx alone)%s | slow beside %7.2f ms\n", W,
                   mode == 0 ? "spread" : mode == 1 ? "interleaved" : mode == 2 ? "spread 64B" : "spread 128B", mode != 1 ? stride / 1024.0 : 0.0, ms, ms * 1e6 / iters, beside, 2.0 * cbytes / beside / 1e6, beside / alone,
                   still ? "" : " (slow kernel had ended)", ms2);
        }
    }
    return 0;
}
