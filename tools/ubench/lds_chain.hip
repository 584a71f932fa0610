// Dependent-chain costs of LDS accesses for a lone wavefront (the shape of the serial-chain kernels' steps).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N (4096 * 64)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(64) void k(uint64_t *out, uint32_t a) {
    __shared__ u32x4 h[16 * 64];
    __shared__ uint32_t r[16 * 64];
    const uint32_t lane = threadIdx.x;
    for (int i = lane; i < 16 * 64; i += 64) { u32x4 v = {(uint32_t)(i * 7 + a) & 15u, 0u, 1u, 0u}; h[i] = v; r[i] = (i * 5 + a) & 15u; }
    __syncthreads();
    uint32_t cur = lane & 15, acc = 0;
    uint64_t t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < N / 8; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (MODE == 0) { cur = r[cur * 64 + lane]; }                                   // b32 read chain
            if (MODE == 1) { const u32x4 v = h[cur * 64 + lane]; cur = v.x & 15u; acc += v.z; }   // b128 read chain
            if (MODE == 2) { const u32x4 v = h[cur * 64 + lane]; u32x4 w = v; w.y = v.y + 1; h[cur * 64 + lane] = w; cur = (v.x + (w.y & 0)) & 15u; }  // read, write back, chain
            if (MODE == 3) { const uint32_t x = r[cur * 64 + lane]; const uint32_t y = r[((x + 1) & 15u) * 64 + lane]; cur = y; }  // two dependent b32 reads
            if (MODE == 4) { cur = (cur * 5 + 1) & 15u; cur = (cur ^ (cur >> 1)) & 15u; cur = (cur + acc) & 15u; acc += cur; }  // 8 dependent VALU
            if (MODE == 5) { const bool c = cur > 7; const uint32_t x = c ? acc : cur + 3; cur = x & 15u; acc += 1; }  // cmp -> vcc -> cndmask chain
        }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    if (lane == 0) { out[0] = t1 - t0; out[1] = cur + acc; }
}
#define RUN(M, name) do { hipMemset(d, 0, 16); hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); k<M><<<1, 64>>>(d, 3u); hipDeviceSynchronize(); hipEventRecord(e0); k<M><<<1, 64>>>(d, 3u); hipEventRecord(e1); hipDeviceSynchronize(); float ms = 0; hipEventElapsedTime(&ms, e0, e1); uint64_t hh[2]; hipMemcpy(hh, d, 16, hipMemcpyDeviceToHost); printf("%-44s %7.1f ticks/iter  %7.1f ns/iter\n", name, (double)hh[0] / N, ms * 1e6 / N); } while (0)
int main() {
    uint64_t *d; hipMalloc(&d, 16);
    RUN(0, "ds_read_b32 dependent chain");
    RUN(1, "ds_read_b128 dependent chain");
    RUN(2, "ds_read_b128 + ds_write_b128 + chain");
    RUN(3, "two dependent ds_read_b32");
    RUN(4, "8 dependent VALU (mul/xor/add/and)");
    RUN(5, "v_cmp -> v_cndmask -> and chain");
    return 0;
}
