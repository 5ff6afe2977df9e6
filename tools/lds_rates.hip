// lds_rates.hip -- what do the LDS access shapes of the encode stage and of the second-level probe cost on gfx950?
// One workgroup per CU, W waves; every wave issues REPS x 16 LDS instructions of one shape (results xor-folded so that
// nothing is dropped); prints ns per wave-instruction per CU and the equivalent LDS clocks at 2.4 GHz.
//   0 b64 aligned, stride 8          1 b64 at byte address lane (unaligned, overlapping windows: the encode row read)
//   2 b64 aligned at lane & ~7       3 b32 at byte address lane (unaligned)       4 b32 aligned at lane & ~3
//   5 u8 at lane                     6 b128 unaligned, random offsets in 64 KiB   7 b128 aligned, random offsets
//   8 b8 writes at (q%3)*80 + q/3    9 read2_b32 + b32 aligned (12 bytes around lane) + 2 v_alignbyte
//  10 b64 unaligned, random offsets  11 b64 aligned random offsets
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 2; } } while (0)
struct B16 { uint32_t w[4]; };
template <int SHAPE>
__global__ __launch_bounds__(1024) void k(uint32_t *out, int reps, uint32_t seed)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    for (uint32_t i = threadIdx.x; i < 65536 / 4; i += blockDim.x) ((uint32_t *)lds)[i] = i * 2654435761u;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint8_t *base = lds + wave * 1024;
    uint32_t acc = 0;
    uint32_t rnd = (threadIdx.x * 2654435761u + seed) >> 7;
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint32_t o = (uint32_t)j * 8u;         // moves the window a little so that loads are not hoisted
            if (SHAPE == 0) { uint2 v; __builtin_memcpy(&v, base + ((lane * 8 + o) & 1023 & ~7u), 8); acc ^= v.x ^ v.y; }
            if (SHAPE == 1) { uint2 v; __builtin_memcpy(&v, base + lane + o, 8); acc ^= v.x ^ v.y; }
            if (SHAPE == 2) { uint2 v; __builtin_memcpy(&v, base + ((lane + o) & ~7u), 8); acc ^= v.x ^ v.y; }
            if (SHAPE == 3) { uint32_t v; __builtin_memcpy(&v, base + lane + o, 4); acc ^= v; }
            if (SHAPE == 4) { uint32_t v; __builtin_memcpy(&v, base + ((lane + o) & ~3u), 4); acc ^= v; }
            if (SHAPE == 5) { acc ^= base[lane + o]; }
            if (SHAPE == 6) { B16 v; rnd = rnd * 1664525u + 1013904223u; __builtin_memcpy(&v, lds + ((rnd >> 8) & 0xFFEFu), 16); acc ^= v.w[0] ^ v.w[1] ^ v.w[2] ^ v.w[3]; }
            if (SHAPE == 7) { B16 v; rnd = rnd * 1664525u + 1013904223u; __builtin_memcpy(&v, lds + ((rnd >> 8) & 0xFFE0u), 16); acc ^= v.w[0] ^ v.w[1] ^ v.w[2] ^ v.w[3]; }
            if (SHAPE == 8) { const uint32_t q = lane + 64u * (j & 3), t = (q * 171u) >> 9, f = q - 3u * t; base[f * 80u + t] = (uint8_t)(acc + j); }
            if (SHAPE == 9) {
                const uint32_t a = lane + o, a4 = a & ~3u, sh = a & 3u;
                const uint32_t *p = (const uint32_t *)(base + a4);
                const uint32_t w0 = p[0], w1 = p[1], w2 = p[2];
                acc ^= __builtin_amdgcn_alignbyte(w1, w0, sh) ^ __builtin_amdgcn_alignbyte(w2, w1, sh);
            }
            if (SHAPE == 10) { uint2 v; rnd = rnd * 1664525u + 1013904223u; __builtin_memcpy(&v, lds + ((rnd >> 8) & 0xFFF7u), 8); acc ^= v.x ^ v.y; }
            if (SHAPE == 11) { uint2 v; rnd = rnd * 1664525u + 1013904223u; __builtin_memcpy(&v, lds + ((rnd >> 8) & 0xFFF0u), 8); acc ^= v.x ^ v.y; }
        }
        if (SHAPE == 8) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    }
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}
template <int SHAPE> int run(int waves, int reps, uint32_t *d)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipFuncSetAttribute((const void *)k<SHAPE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 1024));
    hipLaunchKernelGGL(k<SHAPE>, dim3(256), dim3(64 * waves), 65536 + 1024, 0, d, 10, 1u);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k<SHAPE>, dim3(256), dim3(64 * waves), 65536 + 1024, 0, d, reps, 7u);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    const double n = (double)reps * 16 * waves;           // wave-instructions per CU
    printf("{\"shape\": %d, \"waves\": %d, \"ms\": %.3f, \"ns_per_wave_instr_per_cu\": %.3f, \"clk_at_2.4GHz\": %.2f}\n", SHAPE, waves, ms,
           ms * 1e6 / n, ms * 1e6 / n * 2.4);
    return 0;
}
int main(int argc, char **argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 4000;
    uint32_t *d; CK(hipMalloc((void **)&d, 4096 * 4));
    for (int waves : {4, 16}) {
        run<0>(waves, reps, d); run<1>(waves, reps, d); run<2>(waves, reps, d); run<3>(waves, reps, d); run<4>(waves, reps, d);
        run<5>(waves, reps, d); run<6>(waves, reps, d); run<7>(waves, reps, d); run<8>(waves, reps, d); run<9>(waves, reps, d);
        run<10>(waves, reps, d); run<11>(waves, reps, d);
    }
    return 0;
}
