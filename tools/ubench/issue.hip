// Issue / dependent-latency costs of the instructions the serial-chain kernels are made of, for a LONE wavefront on gfx950.
// Each line: ns and cycles (at the measured wall rate) per instruction.  build: hipcc -O3 --offload-arch=gfx950 issue.hip -o issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))
constexpr int ITERS = 20000;
template <int MODE>
__global__ __launch_bounds__(64) void k(uint64_t *out, uint32_t a, uint32_t b) {
    uint32_t x0 = a + threadIdx.x, x1 = a * 3, x2 = a * 5, x3 = a * 7, y = b | 1;
    uint64_t q0 = ((uint64_t)a << 32) | b, q1 = q0 * 3, q2 = q0 * 5, q3 = q0 * 7;
    double d0 = a, d1 = b, d2 = 1.000001;
    uint32_t s0 = a, s1 = b;
    __shared__ uint32_t lds[1024];
    lds[threadIdx.x] = threadIdx.x & 7; lds[threadIdx.x + 64] = 1;
    __syncthreads();
#pragma unroll 1
    for (int i = 0; i < ITERS; i++) {
        if (MODE == 0) { R16(asm volatile("v_add_u32 %0, %0, %1" : "+v"(x0) : "v"(y));) }                      // dependent add
        if (MODE == 1) { R4(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(y));) }  // independent adds
        if (MODE == 2) { R16(asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x0) : "v"(y));) }
        if (MODE == 3) { R4(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(y));) }
        if (MODE == 4) { R16(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q0) : "v"(x1), "v"(y) : "vcc");) }   // dependent through the 64-bit addend
        if (MODE == 5) { R4(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(x1), "v"(y) : "vcc");) }
        if (MODE == 6) { R16(q0 = __umul64hi(q0, q1 | (1ull << 63)) + q2; asm volatile("" : "+v"(q0));) }       // the encoder's 64x64 high multiply
        if (MODE == 7) { R16(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d0) : "v"(d2), "v"(d1));) }
        if (MODE == 8) { R16(asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc");) }
        if (MODE == 9) { R16(asm volatile("v_readlane_b32 %0, %1, %0\n s_and_b32 %0, %0, 7" : "+s"(s0) : "v"(x0) : "scc");) }   // readlane with the lane select from the previous readlane
        if (MODE == 10) { R16(asm volatile("v_readfirstlane_b32 %1, %0\n v_add_u32 %0, %1, %2" : "+v"(x0), "+s"(s0) : "v"(y));) }  // VALU -> SALU -> VALU round trip
        if (MODE == 11) { R16(asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %2, %3, vcc" : "+v"(x0) : "v"(y), "v"(x1), "v"(x2) : "vcc");) }
        if (MODE == 12) { R16(asm volatile("v_lshrrev_b64 %0, 3, %0\n v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q0) : "v"(q1));) }
        if (MODE == 13) { R16(asm volatile("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n v_lshlrev_b32 %0, 2, %0" : "+v"(x0));) }  // dependent LDS read
        if (MODE == 14) { R16(asm volatile("s_lshl_b32 m0, %0, 1\n s_nop 0\n s_movrels_b32 %0, s40\n s_and_b32 %0, %0, 3" : "+s"(s0) : : "m0", "scc");) }  // SGPR-relative read chain
        if (MODE == 15) { R16(asm volatile("v_cmp_eq_u32 vcc, %0, %1\n s_and_b64 %2, vcc, exec\n s_cselect_b32 %3, 1, 2\n v_add_u32 %0, %0, %3" : "+v"(x0), "+v"(y), "=s"(q1), "+s"(s0) : : "vcc", "scc");) }
    }
    if (threadIdx.x == 0) out[0] = x0 + x1 + x2 + x3 + q0 + q1 + q2 + q3 + (uint64_t)d0 + s0;
}
template <int M> static void run(uint64_t *d, const char *name, int per_iter) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<M><<<1, 64>>>(d, 3u, 5u); hipDeviceSynchronize();
    hipEventRecord(e0); k<M><<<1, 64>>>(d, 3u, 5u); hipEventRecord(e1); hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / ((double)ITERS * per_iter);
    printf("%-64s %7.2f ns  %6.1f cycles @2.4GHz (per instruction / unit)\n", name, ns, ns * 2.4);
}
int main() {
    uint64_t *d; hipMalloc(&d, 16);
    run<0>(d, "v_add_u32 dependent", 16);
    run<1>(d, "v_add_u32 4 independent chains", 16);
    run<2>(d, "v_mul_lo_u32 dependent", 16);
    run<3>(d, "v_mul_lo_u32 4 independent chains", 16);
    run<4>(d, "v_mad_u64_u32 dependent (addend)", 16);
    run<5>(d, "v_mad_u64_u32 4 independent chains", 16);
    run<6>(d, "__umul64hi + add, dependent (unit = one mulhi)", 16);
    run<7>(d, "v_fma_f64 dependent", 16);
    run<8>(d, "s_add_u32 dependent", 16);
    run<9>(d, "v_readlane(sel = prev) + s_and chain (unit = pair)", 16);
    run<10>(d, "v_readfirstlane -> v_add round trip (unit = pair)", 16);
    run<11>(d, "v_cmp -> v_cndmask dependent (unit = pair)", 16);
    run<12>(d, "v_lshrrev_b64 + v_add dependent (unit = pair)", 16);
    run<13>(d, "ds_read_b32 -> wait -> shl dependent (unit = triple)", 16);
    run<14>(d, "s_movrels chain: m0 <- s, s_movrels, s_and (unit = 4 instr)", 16);
    run<15>(d, "v_cmp -> s_and -> s_cselect -> v_add (unit = 4 instr)", 16);
    return 0;
}
