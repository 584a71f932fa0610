// (1) Semantics of the gfx950 LDS-DMA load used as a register-free prefetch: global_load_lds_dwordx4 vaddr, off with M0 = LDS byte
//     address -> lane l's 16 bytes land at LDS[M0 + 16 l] (checked against a plain copy).
// (2) The depth experiment of depth.hip done right: W one-wave workgroups of dependent VALU work (~1.1 us per block), every lane
//     reads 16 bytes of its own stream per block THROUGH an LDS ring filled by LDS-DMA loads issued D blocks ahead (no registers
//     held, no register copies, the wait is a manual s_waitcnt vmcnt(D - 1): loads return in order and the kernel issues no
//     other vector-memory instruction).  Timed alone and beside a device copy that saturates HBM for the kernel's whole life.
// build: hipcc -O3 --offload-arch=gfx950 ldsdma.hip -o ldsdma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))
typedef __attribute__((address_space(3))) uint8_t lds8;
__device__ __forceinline__ void dma16(uint32_t lds_byte_addr, const void *gaddr) {  // every lane: 16 bytes from its own address -> LDS[addr + 16 lane]
    asm volatile("s_mov_b32 m0, %0\n s_nop 0\n global_load_lds_dwordx4 %1, off" :: "s"(lds_byte_addr), "v"(gaddr) : "memory");
}
__global__ __launch_bounds__(64) void sem(const uint4 *src, uint4 *dst) {
    extern __shared__ __align__(16) uint8_t lds[];
    const uint32_t base = (uint32_t)(uintptr_t)(lds8 *)lds;
    // lane l reads element (63 - l) * 3 of its block's 256-element source: a scattered pattern
    dma16(base + 1024, src + blockIdx.x * 256 + (63 - threadIdx.x) * 3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    dst[blockIdx.x * 64 + threadIdx.x] = reinterpret_cast<uint4 *>(lds + 1024)[threadIdx.x];
}
template <int D>
__global__ __launch_bounds__(64) void slow(const uint8_t *__restrict__ src, uint64_t stride, int iters, uint32_t *out) {
    extern __shared__ __align__(16) uint8_t lds[];  // ring of 8 slots x 1 KB
    const uint32_t base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds8 *)lds);
    const uint8_t *p = src + ((uint64_t)blockIdx.x * 64 + threadIdx.x) * stride;
    for (int d = 0; d < D; d++) dma16(base + 1024u * (uint32_t)d, p + 16ull * d);
    uint32_t x = threadIdx.x, y = 0x9E3779B1u;
    for (int it = 0; it < iters; it++) {
        // block `it`: its bytes were requested D blocks ago; D - 1 younger requests may still be in flight
        if (D == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (D == 2) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        if (D == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        if (D == 6) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        const uint4 cur = *reinterpret_cast<const uint4 *>(lds + 1024u * ((uint32_t)it & 7u) + 16u * threadIdx.x);
        x ^= cur.x + cur.w;
        const int nb = it + D < iters ? it + D : iters - 1;
        dma16(base + 1024u * ((uint32_t)(it + D) & 7u), p + 16ull * nb);  // (slot of block it + D: read last D - 8 ... blocks ago, i.e. free)
        R64(asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x) : "v"(y));)
        R64(asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x) : "v"(y));)
        R64(asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x) : "v"(y));)
        R64(asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x) : "v"(y));)
        R64(asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x) : "v"(y));)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (x == 0x12345) out[0] = x;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[1] = x;
}
__global__ __launch_bounds__(256) void copyk(const uint4 *__restrict__ a, uint4 *__restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
int main() {
    {   // ---- (1)
        const int NB = 64;
        std::vector<uint32_t> h(NB * 256 * 4), r(NB * 64 * 4);
        for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u);
        uint4 *ds, *dd;
        CHK(hipMalloc((void **)&ds, h.size() * 4)); CHK(hipMalloc((void **)&dd, r.size() * 4));
        CHK(hipMemcpy(ds, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        sem<<<NB, 64, 4096>>>(ds, dd);
        CHK(hipDeviceSynchronize());
        CHK(hipMemcpy(r.data(), dd, r.size() * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (int b = 0; b < NB; b++) for (int l = 0; l < 64; l++) for (int q = 0; q < 4; q++)
            if (r[((size_t)b * 64 + l) * 4 + q] != h[((size_t)b * 256 + (63 - l) * 3) * 4 + q]) bad++;
        printf("LDS-DMA semantics (lane l's 16 bytes at M0 + 16 l): %s (%zu mismatches)\n", bad ? "DIFFERENT" : "confirmed", bad);
        if (bad) return 1;
    }
    const uint64_t GB = 1ull << 30, work = 16 * GB;
    uint8_t *sa, *ca, *cb; uint32_t *out;
    CHK(hipMalloc((void **)&sa, work)); CHK(hipMemset(sa, 1, work));
    const size_t cbytes = 2 * GB;
    CHK(hipMalloc((void **)&ca, cbytes)); CHK(hipMalloc((void **)&cb, cbytes)); CHK(hipMalloc((void **)&out, 64));
    CHK(hipMemset(ca, 2, cbytes));
    hipStream_t s1, s2; CHK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CHK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t a0, a1; CHK(hipEventCreate(&a0)); CHK(hipEventCreate(&a1));
    const int iters = 16000;
    for (int W : {256, 1024, 2048}) {
        const uint64_t ns = (uint64_t)W * 64;
        const uint64_t stride = ((work / ns - 1024) & ~511ull) | 256;
        if (stride < (uint64_t)iters * 16 + 256 || ns * stride > work) { printf("W=%d does not fit\n", W); continue; }
        for (int D : {1, 2, 4, 6}) {
            auto launch = [&](int it) {
                if (D == 1) slow<1><<<W, 64, 8192, s1>>>(sa, stride, it, out);
                else if (D == 2) slow<2><<<W, 64, 8192, s1>>>(sa, stride, it, out);
                else if (D == 4) slow<4><<<W, 64, 8192, s1>>>(sa, stride, it, out);
                else slow<6><<<W, 64, 8192, s1>>>(sa, stride, it, out);
            };
            launch(64); CHK(hipStreamSynchronize(s1));
            CHK(hipEventRecord(a0, s1)); launch(iters); CHK(hipEventRecord(a1, s1)); CHK(hipEventSynchronize(a1));
            float alone; CHK(hipEventElapsedTime(&alone, a0, a1));
            CHK(hipEventRecord(a0, s1)); launch(iters); CHK(hipEventRecord(a1, s1));
            const int reps = (int)(3.0f * alone / 0.9f) + 4;
            for (int r = 0; r < reps; r++) copyk<<<8192, 256, 0, s2>>>((const uint4 *)ca, (uint4 *)cb, cbytes / 16);
            CHK(hipEventSynchronize(a1));
            float beside; CHK(hipEventElapsedTime(&beside, a0, a1));
            const bool still = hipStreamQuery(s2) == hipErrorNotReady;
            CHK(hipStreamSynchronize(s2));
            printf("W=%5d requests %d block%s ahead: alone %6.2f ms = %5.0f ns/block | beside a saturating copy %6.2f ms = %5.0f ns/block (%.2fx)%s\n", W, D, D > 1 ? "s" : " ",
                   alone, alone * 1e6 / iters, beside, beside * 1e6 / iters, beside / alone, still ? "" : " [copy ended first]");
        }
    }
    return 0;
}
