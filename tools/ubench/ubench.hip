// Dependent-chain latency microbenchmarks for the serial-chain kernels (one wave, one workgroup).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N 4096
#define BENCH(name, body)                                                         \
    __global__ void name(uint64_t *out, uint32_t a, uint32_t b) {                 \
        __shared__ uint32_t lds[1024];                                            \
        lds[threadIdx.x] = threadIdx.x * 4; __syncthreads();                      \
        uint32_t x = a + threadIdx.x, y = b, z = a ^ b; uint64_t q = ((uint64_t)a << 32) | b;  \
        uint64_t t0 = __builtin_readcyclecounter();                               \
        _Pragma("unroll 1") for (int i = 0; i < N / 16; i++) {                    \
            _Pragma("unroll") for (int j = 0; j < 16; j++) { body; }              \
        }                                                                         \
        uint64_t t1 = __builtin_readcyclecounter();                               \
        if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = x + y + z + (uint32_t)q + (uint32_t)(q >> 32) + lds[5]; }   \
    }
BENCH(k_add, x = x + y)
BENCH(k_xor_add, x = (x ^ y) + z)
BENCH(k_mul_lo, x = x * y + 1)
BENCH(k_mul_hi, x = __umulhi(x, y) + z)
BENCH(k_mad64, q = (uint64_t)(uint32_t)q * y + q)
BENCH(k_shr64, q = (q >> (y & 31)) + 0x100000000ull)
BENCH(k_alignbit, x = __builtin_amdgcn_alignbit(x, z, y & 31) + 1)
BENCH(k_umul64hi, q = __umul64hi(q, 0x8000000000000123ull) + q)
BENCH(k_cndmask, x = (x & 1) ? y : (x + z))
BENCH(k_cmp64, q = (q < 0x80000000ull) ? (q << 32 | y) : q - 7)
BENCH(k_lds_read, x = lds[(x & 1023)])
BENCH(k_lds_write_read, lds[threadIdx.x] = x; x = lds[threadIdx.x] + 1)
BENCH(k_lds_write_wait, lds[threadIdx.x] = x; __builtin_amdgcn_s_waitcnt(0xC07F); x = x + y)
BENCH(k_readlane, x = (uint32_t)__builtin_amdgcn_readlane((int)x, (int)(y & 63)) + 1; y = __builtin_amdgcn_readfirstlane((int)x))
BENCH(k_ballot, x = x + (uint32_t)__builtin_popcountll(__ballot(x & 1)))
BENCH(k_dpp, x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, false) + y)
BENCH(k_bfe, x = ((x >> 8) & 0xFF) + z)
BENCH(k_mad24, x = __umul24(x, y) + z)
#define RUN(name, threads) do { hipMemset(d, 0, 16); name<<<1, threads>>>(d, 12345u, 678u); hipDeviceSynchronize(); uint64_t h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost); printf("%-20s threads=%3d  %6.1f cycles/iter\n", #name, threads, (double)h[0] / N); } while (0)
int main() {
    uint64_t *d; hipMalloc(&d, 16);
    for (int th : {64}) {
        RUN(k_add, th); RUN(k_xor_add, th); RUN(k_mul_lo, th); RUN(k_mul_hi, th); RUN(k_mad64, th); RUN(k_shr64, th); RUN(k_alignbit, th);
        RUN(k_umul64hi, th); RUN(k_cndmask, th); RUN(k_cmp64, th); RUN(k_lds_read, th); RUN(k_lds_write_read, th); RUN(k_lds_write_wait, th);
        RUN(k_readlane, th); RUN(k_ballot, th); RUN(k_dpp, th); RUN(k_bfe, th); RUN(k_mad24, th);
    }
    return 0;
}
