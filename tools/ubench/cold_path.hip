// device check of the packed 8-way cumulative-count compare used by k_rans2_dec_chain<BIG>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_a2 __attribute__((ext_vector_type(4), aligned(2)));
typedef __attribute__((address_space(3))) u32x4_a2 lds128u;
typedef uint32_t u32_a2 __attribute__((aligned(2)));
typedef __attribute__((address_space(3))) u32_a2 lds32u;
typedef __attribute__((address_space(3))) uint8_t lds8;
__global__ void k(const uint16_t *cum, const uint8_t *coarse, const uint32_t *slots, uint32_t *out, int nslots) {
    __shared__ __align__(16) uint8_t ltab[4 + 516 + 512 + 64];
    for (int i = threadIdx.x; i < 258; i += 64) ((uint16_t *)(ltab + 4))[i] = cum[i];
    for (int i = threadIdx.x; i < 512; i += 64) ltab[4 + 516 + i] = coarse[i];
    __syncthreads();
    const uint32_t a_fc = (uint32_t)(uintptr_t)(lds8 *)ltab + 4, a_co = a_fc + 516;
    for (int i = threadIdx.x; i < nslots; i += 64) {
        const uint32_t slot = slots[i];
        uint32_t sym = *(const lds8 *)(uintptr_t)(a_co + (slot >> 6));
        const uint32_t slot2 = slot | (slot << 16);
        uint32_t rounds = 0;
        for (int guard = 0; guard < 40; guard++) {
            const u32x4_a2 v = *(const lds128u *)(uintptr_t)(a_fc + 2 * sym + 2);
            uint32_t m[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t vq = v[q];
                const u16x2 d = __builtin_bit_cast(u16x2, slot2) - __builtin_bit_cast(u16x2, vq);
                m[q] = __builtin_bit_cast(uint32_t, d) & 0x80008000u;
            }
            uint32_t all = (m[0] >> 15) | (m[1] >> 13) | (m[2] >> 11) | (m[3] >> 9);
            all = (all | (all >> 15)) & 0xFFu;
            const uint32_t t = (uint32_t)__builtin_ctz(all | 0x100u);
            sym += t; rounds++;
            if (__ballot(t == 8 && sym < 256u) == 0) break;
        }
        sym = sym < 255u ? sym : 255u;
        const uint32_t cc = *(const lds32u *)(uintptr_t)(a_fc + 2 * sym);
        out[2 * i] = sym | (rounds << 16); out[2 * i + 1] = cc;
    }
}
int main() {
    const int pb = 15, N = 256, NS = 64 * 64;
    std::vector<uint32_t> F(N, 1); F[0] = 27000; F[4] = 32768 - 27000 - 254;
    std::vector<uint16_t> cum(258); uint32_t c = 0; for (int i = 0; i < N; i++) { cum[i] = c; c += F[i]; } cum[256] = 1u << pb; cum[257] = 0x8000;
    std::vector<uint8_t> co(512);
    for (int g = 0; g < 512; g++) { uint32_t s = g << 6; int lo = 0; for (int i = 0; i < N; i++) if (cum[i] <= s) lo = i; co[g] = lo; }
    std::vector<uint32_t> slots(NS); srand(3); for (auto &s : slots) s = (rand() % 4 == 0) ? 27000 + rand() % 5768 : rand() % 32768;
    uint16_t *dc; uint8_t *dco; uint32_t *ds, *dout;
    hipMalloc(&dc, 516); hipMalloc(&dco, 512); hipMalloc(&ds, NS * 4); hipMalloc(&dout, NS * 8);
    hipMemcpy(dc, cum.data(), 516, hipMemcpyHostToDevice); hipMemcpy(dco, co.data(), 512, hipMemcpyHostToDevice); hipMemcpy(ds, slots.data(), NS * 4, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dc, dco, ds, dout, NS); std::vector<uint32_t> out(NS * 2); hipMemcpy(out.data(), dout, NS * 8, hipMemcpyDeviceToHost);
    int bad = 0, maxr = 0;
    for (int i = 0; i < NS; i++) {
        int want = 0; for (int j = 0; j < N; j++) if (cum[j] <= slots[i]) want = j;
        const int got = out[2 * i] & 0xFFFF, r = out[2 * i] >> 16; if (r > maxr) maxr = r;
        if (got != want || out[2 * i + 1] != (cum[want] | (uint32_t)cum[want + 1] << 16)) { if (bad < 8) printf("slot %u: got %d want %d (cc %08x)\n", slots[i], got, want, out[2 * i + 1]); bad++; }
    }
    printf("bad %d of %d, max rounds %d\n", bad, NS, maxr);
}
