// Can a wave read its own outstanding vector-memory count?  s_getreg_b32 hwreg(HW_REG_IB_STS): vm_cnt = bits [3:0] | bits [23:22] << 4
// (gfx9 layout).  The kernel issues N independent 16-byte loads from far-apart addresses, reads the register right behind them
// (expected: close to N, the loads take ~1 us) and again after s_waitcnt vmcnt(0) (expected: 0).
// build: hipcc -O3 --offload-arch=gfx950 ibsts.hip -o ibsts
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e_)); return 1; } } while (0)
__device__ __forceinline__ uint32_t vmcnt_now() {
    const uint32_t v = __builtin_amdgcn_s_getreg((31 << 11) | 7);  // HW_REG_IB_STS, all 32 bits
    return (v & 15u) | (((v >> 22) & 3u) << 4);
}
template <int N>
__global__ void k(const uint4 *src, uint32_t *out, uint64_t stride) {
    uint4 v[N];
    const uint4 *p = src + threadIdx.x;
#pragma unroll
    for (int i = 0; i < N; i++) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[i]) : "v"(p + (uint64_t)i * stride) : "memory");
    const uint32_t raw_a = __builtin_amdgcn_s_getreg((31 << 11) | 7);
    const uint32_t a = (raw_a & 15u) | (((raw_a >> 22) & 3u) << 4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint32_t b = vmcnt_now();
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < N; i++) acc += v[i].x;
    if (threadIdx.x == 0) { out[0] = a; out[1] = b; out[2] = raw_a; }
    if (acc == 0x12345) out[3] = acc;
}
int main() {
    uint4 *src; uint32_t *out, h[4];
    const uint64_t stride = 1 << 16;  // uint4 units: 1 MB apart
    CHK(hipMalloc((void **)&src, 64ull << 20)); CHK(hipMemset(src, 1, 64ull << 20)); CHK(hipMalloc((void **)&out, 64));
#define RUN(N) k<N><<<1, 64>>>(src, out, stride); CHK(hipDeviceSynchronize()); CHK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost)); printf("N=%2d: vm_cnt behind the loads %2u, after s_waitcnt vmcnt(0) %u, raw IB_STS 0x%08x\n", N, h[0], h[1], h[2]);
    RUN(1) RUN(1) RUN(5) RUN(10) RUN(15) RUN(16) RUN(17) RUN(18) RUN(24) RUN(32) RUN(40) RUN(18) RUN(18)
    return 0;
}
