// Does the LDS return the right bytes for ds_read_b128 / b64 / b32 at every byte alignment on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4_a1 __attribute__((ext_vector_type(4), aligned(1)));
typedef uint32_t u32x2_a1 __attribute__((ext_vector_type(2), aligned(1)));
typedef uint32_t u32_a1 __attribute__((aligned(1)));
__global__ void k(uint32_t *out) {
    __shared__ __align__(16) uint8_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint8_t)(i * 7 + 3);
    __syncthreads();
    const uint32_t off = threadIdx.x;  // byte offsets 0..63
    const u32x4_a1 a = *reinterpret_cast<const u32x4_a1 *>(lds + 100 + off);
    const u32x2_a1 b = *reinterpret_cast<const u32x2_a1 *>(lds + 1000 + off);
    const u32_a1 c = *reinterpret_cast<const u32_a1 *>(lds + 2000 + off);
    uint32_t *o = out + threadIdx.x * 8;
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = c;
}
int main() {
    uint32_t *d; hipMalloc(&d, 64 * 32); k<<<1, 64>>>(d); uint32_t h[64 * 8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad128 = 0, bad64 = 0, bad32 = 0;
    auto by = [](int i) { return (uint32_t)(uint8_t)(i * 7 + 3); };
    auto w = [&](int i) { return by(i) | by(i + 1) << 8 | by(i + 2) << 16 | by(i + 3) << 24; };
    for (int t = 0; t < 64; t++) {
        for (int q = 0; q < 4; q++) if (h[t * 8 + q] != w(100 + t + 4 * q)) { bad128++; printf("b128 off %d dword %d: got %08x want %08x\n", t, q, h[t * 8 + q], w(100 + t + 4 * q)); break; }
        for (int q = 0; q < 2; q++) if (h[t * 8 + 4 + q] != w(1000 + t + 4 * q)) { bad64++; break; }
        if (h[t * 8 + 6] != w(2000 + t)) bad32++;
    }
    printf("unaligned LDS reads wrong at: b128 %d/64 offsets, b64 %d/64, b32 %d/64\n", bad128, bad64, bad32);
    return 0;
}
