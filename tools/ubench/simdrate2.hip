// Issue rates of single instruction kinds on one SIMD of gfx950 with N wavefronts sharing it (see simdrate.hip; this is the long table).
// Per kind and N: ns per instruction GROUP as one wavefront sees it (a group = the 1 or 2 instructions of the kind) and groups per ns per
// SIMD.  build: hipcc -O3 --offload-arch=gfx950 simdrate2.hip -o simdrate2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <algorithm>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))
constexpr int ITERS = 1000;
template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t a, uint32_t b) {
    extern __shared__ uint32_t lds[];
    uint32_t x = a + threadIdx.x, y = b | 1, z = a * 7 + threadIdx.x, w = b * 3 + threadIdx.x;
    uint64_t q = ((uint64_t)a << 32) | (b + threadIdx.x), m = 0x5555555555555555ull;
    lds[threadIdx.x] = x;
    __syncthreads();
#pragma unroll 1
    for (int i = 0; i < ITERS; i++) {
        if (MODE == 0) { R64(asm volatile("v_add_u32 %0, %0, %1" : "+v"(x), "+v"(y), "+v"(z), "+s"(m), "+v"(w), "+v"(q) : : "vcc", "scc", "s20", "s21", "memory");) }
        if (MODE == 1) { R64(asm volatile("v_add_u32_e64 %0, %0, %1" : "+v"(x), "+v"(y), "+v"(z), "+s"(m), "+v"(w), "+v"(q) : : "vcc", "scc", "s20", "s21", "memory");) }
        if (MODE == 2) { R64(asm volatile("v_add_u32 %0, 0x12345, %0" : "+v"(x), "+v"(y), "+v"(z), "+s"(m), "+v"(w), "+v"(q) : : "vcc", "scc", "s20", "s21", "memory");) }
        if (MODE == 3) { R64(asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x), "+v"(y), "+v"(z), "+s"(m), "+v"(w), "+v"(q) : : "vcc", "scc", "s20", "s21", "memory");) }
        if (MODE == 4) { R64(asm volatile("v_and_b32 %0, %0, %1" : "+v"(x), "+v"(y), "+v"(z), "+s"(m), "+v"(w), "+v"(q) : : "vcc", "scc", "s20", "s21", "memory");) }
        if (MODE == 5) { R64(asm volatile("v_or_b32 %0, %0, %1" : "+v"(x), "+v"(y), "+v"(z), "+s"(m), "+v"(w), "+v"(q) : : "vcc", "scc", "s20", "s21", "memory");) }
        if (MODE == 6) { R64(asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(x), "+v"(y), "+v"(z), "+s"(m), "+v"(w), "+v"(q) : : "vcc", "scc", "s20", "s21", "memory");) }
        if (MODE == 7) { R64(asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(x), "+v"(y), "+v"(z), "+s"(m), "+v"(w), "+v"(q) : : "vcc", "scc", "s20", "s21", "memory");) }
        if (MODE == 8) { R64(asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(x), "+v"(y), "+v"(z), "+s"(m), "+v"(w), "+v"(q) : : "vcc", "scc", "s20", "s21", "memory");) }
        if (MODE == 9) { R64(asm volatile("v_min_u32 %0, %0, %1" : "+v"(x), "+v"(y), "+v"(z), "+s"(m), "+v"(w), "+v"(q) : : "vcc", "scc", "s20", "s21", "memory");) }
        if (MODE == 10) { R64(asm volatile("v_max_u32 %0, %0, %1" : "+v"(x), "+v"(y), "+v"(z), "+s"(m), "+v"(w), "+v"(q) : : "vcc", "scc", "s20", "s21", "memory");) }
        if (MODE == 11) { R64(asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x), "+v"(y), "+v"(z), "+s"(m), "+v"(w), "+v"(q) : : "vcc", "scc", "s20", "s21", "memory");) }
        if (MODE == 12) { R64(asm volatile("v_mov_b32 %0, %0" : "+v"(x), "+v"(y), "+v"(z), "+s"(m), "+v"(w), "+v"(q) : : "vcc", "scc", "s20", "s21", "memory");) }
        if (MODE == 13) { R64(asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(x), "+v"(y), "+v"(z), "+s"(m), "+v"(w), "+v"(q) : : "vcc", "scc", "s20", "s21", "memory");) }
        if (MODE == 14) { R64(asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x), "+v"(y), "+v"(z), "+s"(m), "+v"(w), "+v"(q) : : "vcc", "scc", "s20", "s21", "memory");) }
        if (MODE == 15) { R64(asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x), "+v"(y), "+v"(z), "+s"(m), "+v"(w), "+v"(q) : : "vcc", "scc", "s20", "s21", "memory");) }
        if (MODE == 16) { R64(asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(x), "+v"(y), "+v"(z), "+s"(m), "+v"(w), "+v"(q) : : "vcc", "scc", "s20", "s21", "memory");) }
        if (MODE == 17) { R64(asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(x), "+v"(y), "+v"(z), "+s"(m), "+v"(w), "+v"(q) : : "vcc", "scc", "s20", "s21", "memory");) }
        if (MODE == 18) { R64(asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(x), "+v"(y), "+v"(z), "+s"(m), "+v"(w), "+v"(q) : : "vcc", "scc", "s20", "s21", "memory");) }
        if (MODE == 19) { R64(asm volatile("v_and_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "+v"(x), "+v"(y), "+v"(z), "+s"(m), "+v"(w), "+v"(q) : : "vcc", "scc", "s20", "s21", "memory");) }
        if (MODE == 20) { R64(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x), "+v"(y), "+v"(z), "+s"(m), "+v"(w), "+v"(q) : : "vcc", "scc", "s20", "s21", "memory");) }
        if (MODE == 21) { R64(asm volatile("v_cndmask_b32_e64 %0, %0, %1, %3" : "+v"(x), "+v"(y), "+v"(z), "+s"(m), "+v"(w), "+v"(q) : : "vcc", "scc", "s20", "s21", "memory");) }
        if (MODE == 22) { R64(asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(x), "+v"(y), "+v"(z), "+s"(m), "+v"(w), "+v"(q) : : "vcc", "scc", "s20", "s21", "memory");) }
        if (MODE == 23) { R64(asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cmp_lt_u32 vcc, %0, %2" : "+v"(x), "+v"(y), "+v"(z), "+s"(m), "+v"(w), "+v"(q) : : "vcc", "scc", "s20", "s21", "memory");) }
        if (MODE == 24) { R64(asm volatile("v_cmp_lt_u32_e64 %3, %0, %1" : "+v"(x), "+v"(y), "+v"(z), "+s"(m), "+v"(w), "+v"(q) : : "vcc", "scc", "s20", "s21", "memory");) }
        if (MODE == 25) { R64(asm volatile("v_readfirstlane_b32 s20, %0" : "+v"(x), "+v"(y), "+v"(z), "+s"(m), "+v"(w), "+v"(q) : : "vcc", "scc", "s20", "s21", "memory");) }
        if (MODE == 26) { R64(asm volatile("v_readlane_b32 s20, %0, 5\n v_add_u32 %0, s20, %0" : "+v"(x), "+v"(y), "+v"(z), "+s"(m), "+v"(w), "+v"(q) : : "vcc", "scc", "s20", "s21", "memory");) }
        if (MODE == 27) { R64(asm volatile("v_mbcnt_lo_u32_b32 %0, -1, %0" : "+v"(x), "+v"(y), "+v"(z), "+s"(m), "+v"(w), "+v"(q) : : "vcc", "scc", "s20", "s21", "memory");) }
        if (MODE == 28) { R64(asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(x), "+v"(y), "+v"(z), "+s"(m), "+v"(w), "+v"(q) : : "vcc", "scc", "s20", "s21", "memory");) }
        if (MODE == 29) { R64(asm volatile("s_add_u32 s20, s20, s21" : "+v"(x), "+v"(y), "+v"(z), "+s"(m), "+v"(w), "+v"(q) : : "vcc", "scc", "s20", "s21", "memory");) }
        if (MODE == 30) { R64(asm volatile("s_lshl_b32 s20, s20, 1\n s_and_b32 s20, s20, s21" : "+v"(x), "+v"(y), "+v"(z), "+s"(m), "+v"(w), "+v"(q) : : "vcc", "scc", "s20", "s21", "memory");) }
        if (MODE == 31) { R64(asm volatile("v_add_u32 %0, %0, %1\n s_add_u32 s20, s20, s21" : "+v"(x), "+v"(y), "+v"(z), "+s"(m), "+v"(w), "+v"(q) : : "vcc", "scc", "s20", "s21", "memory");) }
        if (MODE == 32) { R64(asm volatile("v_lshl_add_u32 %0, %0, 3, %1\n s_add_u32 s20, s20, s21" : "+v"(x), "+v"(y), "+v"(z), "+s"(m), "+v"(w), "+v"(q) : : "vcc", "scc", "s20", "s21", "memory");) }
        if (MODE == 33) { R64(asm volatile("v_lshl_add_u32 %0, %0, 3, %1\n v_lshl_add_u32 %4, %4, 3, %1" : "+v"(x), "+v"(y), "+v"(z), "+s"(m), "+v"(w), "+v"(q) : : "vcc", "scc", "s20", "s21", "memory");) }
        if (MODE == 34) { R64(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %4, %4, %1" : "+v"(x), "+v"(y), "+v"(z), "+s"(m), "+v"(w), "+v"(q) : : "vcc", "scc", "s20", "s21", "memory");) }
        if (MODE == 35) { R64(asm volatile("v_add_u32 %0, %0, %1\n v_lshl_add_u32 %4, %4, 3, %1" : "+v"(x), "+v"(y), "+v"(z), "+s"(m), "+v"(w), "+v"(q) : : "vcc", "scc", "s20", "s21", "memory");) }
        if (MODE == 36) { R64(asm volatile("v_mul_lo_u32 %0, %0, %1\n v_add_u32 %4, %4, %1" : "+v"(x), "+v"(y), "+v"(z), "+s"(m), "+v"(w), "+v"(q) : : "vcc", "scc", "s20", "s21", "memory");) }
        if (MODE == 37) { R64(asm volatile("v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %4, vcc, %4, %2, vcc" : "+v"(x), "+v"(y), "+v"(z), "+s"(m), "+v"(w), "+v"(q) : : "vcc", "scc", "s20", "s21", "memory");) }
        if (MODE == 38) { R64(asm volatile("v_lshlrev_b64 %5, 1, %5" : "+v"(x), "+v"(y), "+v"(z), "+s"(m), "+v"(w), "+v"(q) : : "vcc", "scc", "s20", "s21", "memory");) }
        if (MODE == 39) { R64(asm volatile("v_lshrrev_b64 %5, 1, %5" : "+v"(x), "+v"(y), "+v"(z), "+s"(m), "+v"(w), "+v"(q) : : "vcc", "scc", "s20", "s21", "memory");) }
        if (MODE == 40) { R64(asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(x), "+v"(y), "+v"(z), "+s"(m), "+v"(w), "+v"(q) : : "vcc", "scc", "s20", "s21", "memory");) }
        if (MODE == 41) { R64(asm volatile("v_bfe_i32 %0, %0, 1, 31" : "+v"(x), "+v"(y), "+v"(z), "+s"(m), "+v"(w), "+v"(q) : : "vcc", "scc", "s20", "s21", "memory");) }
        if (MODE == 42) { R64(asm volatile("v_ffbh_u32 %0, %0" : "+v"(x), "+v"(y), "+v"(z), "+s"(m), "+v"(w), "+v"(q) : : "vcc", "scc", "s20", "s21", "memory");) }
        if (MODE == 43) { R64(asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(x), "+v"(y), "+v"(z), "+s"(m), "+v"(w), "+v"(q) : : "vcc", "scc", "s20", "s21", "memory");) }

    }
    if (x == 0x12345 && (uint32_t)q == 77 && w == 5) out[0] = x;
}
template <int MODE>
static void run(const char *name, uint32_t *d_out, int cus) {
    printf("%-26s", name);
    for (int N : {1, 2, 4, 8}) {
        const size_t lds = std::max<size_t>(8192, (160 * 1024) / N - 1024);
        CHK(hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const int grid = cus * N;
        k<MODE><<<grid, 256, lds>>>(d_out, 3, 5);
        CHK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        CHK(hipEventRecord(e0)); k<MODE><<<grid, 256, lds>>>(d_out, 3, 5); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        const double n = (double)ITERS * 64;
        printf(" | N=%d %6.2f ns %5.2f /ns/SIMD", N, ms * 1e6 / n, N * n / (ms * 1e6));
        CHK(hipEventDestroy(e0)); CHK(hipEventDestroy(e1));
    }
    printf("\n");
}
int main() {
    uint32_t *d_out; CHK(hipMalloc((void **)&d_out, 64));
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    run<0>("v_add_u32", d_out, cus);
    run<1>("v_add_u32_e64", d_out, cus);
    run<2>("v_add_u32 lit", d_out, cus);
    run<3>("v_sub_u32", d_out, cus);
    run<4>("v_and_b32", d_out, cus);
    run<5>("v_or_b32", d_out, cus);
    run<6>("v_lshlrev_b32", d_out, cus);
    run<7>("v_lshrrev_b32", d_out, cus);
    run<8>("v_lshrrev_b32 v", d_out, cus);
    run<9>("v_min_u32", d_out, cus);
    run<10>("v_max_u32", d_out, cus);
    run<11>("v_mul_u32_u24", d_out, cus);
    run<12>("v_mov_b32", d_out, cus);
    run<13>("v_bfi_b32", d_out, cus);
    run<14>("v_perm_b32", d_out, cus);
    run<15>("v_add3_u32", d_out, cus);
    run<16>("v_xad_u32", d_out, cus);
    run<17>("v_lshl_or_b32", d_out, cus);
    run<18>("v_or3_b32", d_out, cus);
    run<19>("v_and_b32 sdwa", d_out, cus);
    run<20>("v_cndmask vcc", d_out, cus);
    run<21>("v_cndmask sgpr", d_out, cus);
    run<22>("v_cmp+v_cndmask", d_out, cus);
    run<23>("v_cmp_lt_u32 (x2 indep)", d_out, cus);
    run<24>("v_cmp_e64 sgpr", d_out, cus);
    run<25>("v_readfirstlane", d_out, cus);
    run<26>("v_readlane+v_add", d_out, cus);
    run<27>("v_mbcnt_lo", d_out, cus);
    run<28>("ds_bpermute+wait", d_out, cus);
    run<29>("s_add_u32", d_out, cus);
    run<30>("s_lshl_b32+s_and", d_out, cus);
    run<31>("v_add + s_add (mixed)", d_out, cus);
    run<32>("v_lshl_add + s_add", d_out, cus);
    run<33>("2 indep v_lshl_add", d_out, cus);
    run<34>("2 indep v_add", d_out, cus);
    run<35>("v_add + v_lshl_add indep", d_out, cus);
    run<36>("v_mul_lo + v_add indep", d_out, cus);
    run<37>("v_add_co+v_addc", d_out, cus);
    run<38>("v_lshlrev_b64", d_out, cus);
    run<39>("v_lshrrev_b64", d_out, cus);
    run<40>("v_pk_add_u16", d_out, cus);
    run<41>("v_bfe_i32", d_out, cus);
    run<42>("v_ffbh_u32", d_out, cus);
    run<43>("v_cvt_f32_u32", d_out, cus);

    return 0;
}
