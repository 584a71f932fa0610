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
// (3) the same with one 16-byte STORE per lane and block, as the encode chains have: [wait] [read block] [store] [request block + 3].
//     MODE 0 waits with s_waitcnt vmcnt(1): loads and stores retire in order on gfx9, so the store of the previous boundary (one block
//     old) has to be acknowledged too.  MODE 1 never waits on vmcnt inside the loop: the ring slot is filled with a sentinel before
//     its request is issued and the reader polls it (the source bytes never equal the sentinel).
template <int MODE>
__global__ __launch_bounds__(64) void slow2(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, uint64_t stride, int iters, uint32_t *out) {
    extern __shared__ __align__(16) uint8_t lds[];
    constexpr int D = 3;
    const uint32_t base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds8 *)lds);
    const uint8_t *p = src + ((uint64_t)blockIdx.x * 64 + threadIdx.x) * stride;
    uint8_t *q = dst + ((uint64_t)blockIdx.x * 64 + threadIdx.x) * stride;
    uint4 *ring = reinterpret_cast<uint4 *>(lds);
    const uint4 sent = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    for (int d = 0; d < 8; d++) ring[64 * d + threadIdx.x] = sent;
    for (int d = 0; d < D; d++) dma16(base + 1024u * (uint32_t)d, p + 16ull * d);
    uint32_t x = threadIdx.x, y = 0x9E3779B1u;
    for (int it = 0; it < iters; it++) {
        uint4 cur;
        // vector-memory instructions in program order: ... st(it-2) DMA(it+1) st(it-1) DMA(it+2) | this iteration: st(it) DMA(it+3)
        if (MODE == 0) {
            asm volatile("s_waitcnt vmcnt(1)" ::: "memory");  // as a kernel whose store count per block is unknown must wait: everything but the newest request, i.e. st(it-1) too
            cur = ring[64 * (it & 7) + threadIdx.x];
        } else if (MODE == 2) {
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // exact: DMA(it) is the fifth youngest
            cur = ring[64 * (it & 7) + threadIdx.x];
        } else {
            int spins = 0;
            do {
                asm volatile("" ::: "memory");  // (re-read the slot every round)
                cur = ring[64 * (it & 7) + threadIdx.x];
            } while (__ballot(cur.w == 0xFFFFFFFFu) != 0 && ++spins < (1 << 20));
            if (spins >= (1 << 20)) { out[2] = 1; break; }  // (never: the request was issued three blocks ago)
            ring[64 * (it & 7) + threadIdx.x] = sent;  // (re-armed for the request that will reuse this slot, 5 blocks from now)
        }
        x ^= cur.x + cur.w;
        *reinterpret_cast<uint4 *>(q + 16ull * it) = make_uint4(x, cur.y, cur.z, cur.w);
        const int nb = it + D < iters ? it + D : iters - 1;
        dma16(base + 1024u * ((uint32_t)(it + D) & 7u), p + 16ull * nb);
        R64(asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x) : "v"(y));)
        R64(asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x) : "v"(y));)
        R64(asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x) : "v"(y));)
        R64(asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x) : "v"(y));)
        R64(asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x) : "v"(y));)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (x == 0x12345) out[0] = x;
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
    uint8_t *sa, *sb2, *ca, *cb; uint32_t *out;
    CHK(hipMalloc((void **)&sa, work)); CHK(hipMemset(sa, 1, work));
    CHK(hipMalloc((void **)&sb2, work));
    const size_t cbytes = 2 * GB;
    CHK(hipMalloc((void **)&ca, cbytes)); CHK(hipMalloc((void **)&cb, cbytes)); CHK(hipMalloc((void **)&out, 64)); CHK(hipMemset(out, 0, 64));
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
    for (int W : {256, 1024}) {
        const uint64_t ns = (uint64_t)W * 64;
        const uint64_t stride = ((work / ns - 1024) & ~511ull) | 256;
        if (stride < (uint64_t)iters * 16 + 256 || ns * stride > work) continue;
        for (int mode = 0; mode < 3; mode++) {
            auto launch = [&](int it) {
                if (mode == 0) slow2<0><<<W, 64, 8192, s1>>>(sa, sb2, stride, it, out);
                else if (mode == 1) slow2<1><<<W, 64, 8192, s1>>>(sa, sb2, stride, it, out);
                else slow2<2><<<W, 64, 8192, s1>>>(sa, sb2, stride, it, out);
            };
            launch(64); CHK(hipStreamSynchronize(s1));
            CHK(hipEventRecord(a0, s1)); launch(iters); CHK(hipEventRecord(a1, s1)); CHK(hipEventSynchronize(a1));
            float alone; CHK(hipEventElapsedTime(&alone, a0, a1));
            CHK(hipEventRecord(a0, s1)); launch(iters); CHK(hipEventRecord(a1, s1));
            const int reps = (int)(3.0f * alone / 0.9f) + 4;
            for (int r = 0; r < reps; r++) copyk<<<8192, 256, 0, s2>>>((const uint4 *)ca, (uint4 *)cb, cbytes / 16);
            CHK(hipEventSynchronize(a1));
            float beside; CHK(hipEventElapsedTime(&beside, a0, a1));
            CHK(hipStreamSynchronize(s2));
            printf("W=%5d 3 blocks ahead + a 16-byte store per block, %s: alone %6.2f ms = %5.0f ns/block | beside a saturating copy %6.2f ms = %5.0f ns/block (%.2fx)\n", W,
                   mode == 0 ? "s_waitcnt vmcnt(1): the previous block's store included" : mode == 1 ? "polling the ring slot, no vmcnt                      " : "s_waitcnt vmcnt(4): exact, stores stay in flight     ", alone, alone * 1e6 / iters, beside, beside * 1e6 / iters, beside / alone);
        }
    }
    return 0;
}
