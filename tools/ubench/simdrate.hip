// What does one SIMD of gfx950 retire per cycle when N wavefronts share it, each running a DEPENDENT chain of one instruction kind
// (the shape of the serial-chain kernels' streams)?  256 x N workgroups of 256 threads (one wavefront per SIMD each; dynamic LDS sized
// so that exactly N workgroups fit a compute unit), every wavefront issues ITERS x 64 instructions of the kind; printed: cycles per
// instruction as ONE wavefront sees them, and instructions per cycle per SIMD.  Cycles from s_memtime deltas inside the kernel.
// build: hipcc -O3 --offload-arch=gfx950 simdrate.hip -o simdrate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))
constexpr int ITERS = 2000;
template <int MODE>
__global__ __launch_bounds__(256) void k(uint64_t *cyc, uint32_t *out, uint32_t a, uint32_t b) {
    extern __shared__ uint32_t lds[];
    uint32_t x = a + threadIdx.x, y = b | 1, z = a * 7 + threadIdx.x;
    uint64_t q = ((uint64_t)a << 32) | (b + threadIdx.x);
    for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = (i * 2654435761u) >> 21;  // 11-bit values: a random walk inside the 8 KB
    __syncthreads();
    const uint64_t t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < ITERS; i++) {
        if (MODE == 0) { R64(asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(y));) }
        if (MODE == 1) { R64(asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(x) : "v"(y));) }
        if (MODE == 2) { R64(asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));) }
        if (MODE == 3) { R64(asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));) }
        if (MODE == 4) { R64(asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x) : "v"(y));) }
        if (MODE == 5) { R64(asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(y));) }
        if (MODE == 6) { R64(asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(y));) }
        if (MODE == 7) { R64(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q) : "v"(z), "v"(y) : "vcc");) }
        if (MODE == 8) { R64(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(y) : );) }
        if (MODE == 9) { R64(asm volatile("v_bfe_u32 %0, %0, 1, 31" : "+v"(x));) }
        if (MODE == 10) { R64(asm volatile("v_lshlrev_b32 %0, 2, %0\n ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)" : "+v"(x) : : "memory");) }  // dependent random LDS gather (2 instructions)
        if (MODE == 11) { R64(asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(y));) }
        if (MODE == 12) { R64(asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));) }
        if (MODE == 13) { R64(asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(y));) }
        if (MODE == 14) { R64(asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x));) }
        if (MODE == 15) { R64(asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(x) : "v"(y) : "vcc");) }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    if (x == 0x12345 && (uint32_t)q == 77) out[0] = x;
}
static const char *NAMES[16] = {"v_add_u32", "v_lshl_add_u32", "v_and_or_b32", "v_alignbit_b32", "v_mad_u32_u24", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32",
                                "v_cndmask_b32", "v_bfe_u32", "lshl+ds_read_b32+wait", "v_add_f32", "v_fma_f32", "v_xor_b32", "v_mov_dpp row_shr", "v_add_co_u32"};
template <int MODE>
static void run(uint64_t *d_cyc, uint32_t *d_out, int cus) {
    printf("%-24s", NAMES[MODE]);
    for (int N : {1, 2, 3, 4, 6, 8}) {
        const size_t lds = std::max<size_t>(8192, (160 * 1024) / N - 1024);
        CHK(hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const int grid = cus * N;
        k<MODE><<<grid, 256, lds>>>(d_cyc, d_out, 3, 5);
        CHK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        CHK(hipEventRecord(e0)); k<MODE><<<grid, 256, lds>>>(d_cyc, d_out, 3, 5); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<uint64_t> h(grid * 4);
        CHK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        const double med = (double)h[h.size() / 2], n_inst = (double)ITERS * 64 * (MODE == 10 ? 1 : 1);
        // counter ticks per instruction for the median wavefront; ticks -> the wall clock: ticks of the kernel's longest wavefront ~ ms
        const double ticks_per_ms = (double)h.back() / ms;
        printf(" | N=%d %5.2f ns/inst %5.2f inst/ns/SIMD (%4.2f t/inst, %.0f t/us)", N, ms * 1e6 / n_inst, N * n_inst / (ms * 1e6), med / n_inst, ticks_per_ms / 1e3);
        CHK(hipEventDestroy(e0)); CHK(hipEventDestroy(e1));
    }
    printf("\n");
}
int main() {
    uint64_t *d_cyc; uint32_t *d_out;
    CHK(hipMalloc((void **)&d_cyc, 8 * 4 * 256 * 8 + 64)); CHK(hipMalloc((void **)&d_out, 64));
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("%d compute units; ticks = s_memtime / __builtin_readcyclecounter units (MHz = ticks per wall microsecond)\n", cus);
    run<0>(d_cyc, d_out, cus); run<1>(d_cyc, d_out, cus); run<2>(d_cyc, d_out, cus); run<3>(d_cyc, d_out, cus); run<4>(d_cyc, d_out, cus);
    run<5>(d_cyc, d_out, cus); run<6>(d_cyc, d_out, cus); run<7>(d_cyc, d_out, cus); run<8>(d_cyc, d_out, cus); run<9>(d_cyc, d_out, cus);
    run<10>(d_cyc, d_out, cus); run<11>(d_cyc, d_out, cus); run<12>(d_cyc, d_out, cus); run<13>(d_cyc, d_out, cus); run<14>(d_cyc, d_out, cus); run<15>(d_cyc, d_out, cus);
    return 0;
}
