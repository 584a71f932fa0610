// Background load generators for tools/interference.py: each fills the GPU with ONE kind of pressure (VALU issue, LDS,
// HBM streaming, scattered 16-byte HBM accesses, instruction-cache footprint) on the stream it is given, so that the slowdown
// of the serial-chain kernels beside the bandwidth kernels of other pipeline slots can be attributed to one mechanism.
// build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC hammer.hip -o hammer.so
#include <hip/hip_runtime.h>
#include <cstdint>
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))
#define R256(x) R4(R64(x))

__global__ __launch_bounds__(256) void k_valu(uint32_t *out, uint32_t a, int iters) {
    uint32_t x0 = a + threadIdx.x, x1 = a * 3, x2 = a * 5, x3 = a * 7;
    for (int i = 0; i < iters; i++) {
        R16(asm volatile("v_mad_u32_u24 %0, %0, %4, %1\n v_mad_u32_u24 %1, %1, %4, %2\n v_mad_u32_u24 %2, %2, %4, %3\n v_mad_u32_u24 %3, %3, %4, %0"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a));)
    }
    if (x0 + x1 + x2 + x3 == 0x12345) out[0] = 1;
}

__global__ __launch_bounds__(256) void k_lds(uint32_t *out, uint32_t a, int iters) {
    __shared__ uint32_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = (i * 7 + a) & 4095;
    __syncthreads();
    uint32_t x = threadIdx.x;
    for (int i = 0; i < iters; i++) {
        R16(x = lds[x]; lds[(x + 64) & 4095] = x;)
    }
    if (x == 0x12345) out[0] = x;
}

__global__ __launch_bounds__(256) void k_stream(const uint4 *__restrict__ src, uint4 *__restrict__ dst, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) dst[i] = src[i];
}

// 16-byte loads and stores with a 64-row stride pattern like the band reconstruction's (each lane its own 4 KB-apart line)
__global__ __launch_bounds__(64) void k_scatter(const uint4 *__restrict__ src, uint4 *__restrict__ dst, uint64_t n_rows, uint32_t row_u4, int iters) {
    const uint64_t row = ((uint64_t)blockIdx.x * 64 + threadIdx.x) % n_rows;
    const uint4 *s = src + row * row_u4;
    uint4 *d = dst + row * row_u4;
    for (int i = 0; i < iters; i++)
        for (uint32_t c = 0; c < row_u4; c++) d[c] = s[c];
}

// a big straight-line body: ~256 KB of code walked in a loop by every wave, to evict other kernels' lines from the
// instruction cache (64 KB shared by two CUs)
__global__ __launch_bounds__(64) void k_icache(uint32_t *out, uint32_t a, int iters) {
    uint32_t x = a + threadIdx.x;
    for (int i = 0; i < iters; i++) {
        R4(R256(R64(asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(a));)))
    }
    if (x == 0x12345) out[0] = x;
}

extern "C" {
int hammer_valu(void *stream, uint32_t *scratch, int blocks, int iters) {
    k_valu<<<blocks, 256, 0, (hipStream_t)stream>>>(scratch, 3, iters);
    return (int)hipGetLastError();
}
int hammer_lds(void *stream, uint32_t *scratch, int blocks, int iters) {
    k_lds<<<blocks, 256, 0, (hipStream_t)stream>>>(scratch, 3, iters);
    return (int)hipGetLastError();
}
int hammer_stream(void *stream, const void *src, void *dst, uint64_t bytes, int blocks) {
    k_stream<<<blocks, 256, 0, (hipStream_t)stream>>>((const uint4 *)src, (uint4 *)dst, bytes / 16);
    return (int)hipGetLastError();
}
int hammer_scatter(void *stream, const void *src, void *dst, uint64_t bytes, int blocks, int iters) {
    const uint32_t row_u4 = 111;  // a 444-pixel tile row
    k_scatter<<<blocks, 64, 0, (hipStream_t)stream>>>((const uint4 *)src, (uint4 *)dst, bytes / 16 / row_u4, row_u4, iters);
    return (int)hipGetLastError();
}
int hammer_icache(void *stream, uint32_t *scratch, int blocks, int iters) {
    k_icache<<<blocks, 64, 0, (hipStream_t)stream>>>(scratch, 3, iters);
    return (int)hipGetLastError();
}
}
