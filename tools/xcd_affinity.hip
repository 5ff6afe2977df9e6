// xcd_affinity.hip -- does "blockIdx % 8" really select an XCD (and so an L2)?  Eight regions of <region_MiB>;
// every workgroup reads random 16-byte words from ONE region chosen by
//   mode 0: blockIdx % 8        mode 1: HW_REG_XCC_ID        mode 2: (blockIdx / 8) % 8  (anti-affine)
// If the choice matches the XCD, each L2 holds one region and the rate is the L2-resident rate.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}
__global__ __launch_bounds__(256) void gather(const uint8_t *__restrict__ tab, uint64_t region_bytes, int mode, int rounds, uint32_t *out, uint32_t *xcc_hist)
{
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xF;
    if (threadIdx.x == 0) atomicAdd(&xcc_hist[(blockIdx.x & 7) * 16 + xcc], 1u);
    uint32_t region = mode == 0 ? (blockIdx.x & 7) : mode == 1 ? (xcc & 7) : ((blockIdx.x >> 3) & 7);
    const uint8_t *base = tab + (uint64_t)region * region_bytes;
    uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (int r = 0; r < rounds; r++) {
        uint4 v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint64_t h = mix(id * 1315423911ull + (uint64_t)(r * 4 + k));
            uint64_t off = __umul64hi(h, region_bytes - 64) & ~15ull;
            __builtin_memcpy(&v[k], base + off, 16);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main(int argc, char **argv)
{
    uint64_t region = (uint64_t)(argc > 1 ? atoll(argv[1]) : 2) << 20;
    int wgs = argc > 2 ? atoi(argv[2]) : 1024, rounds = argc > 3 ? atoi(argv[3]) : 256;
    uint8_t *d; uint32_t *d_out, *d_hist;
    CK(hipMalloc((void **)&d, region * 8)); CK(hipMalloc((void **)&d_out, 64)); CK(hipMalloc((void **)&d_hist, 128 * 4));
    CK(hipMemset(d, 0x5A, region * 8));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int mode = 0; mode < 3; mode++) {
        float ms = 0;
        CK(hipMemset(d_hist, 0, 128 * 4));
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(gather, dim3(wgs), dim3(256), 0, 0, d, region, mode, rounds, d_out, d_hist);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
        }
        double loads = (double)wgs * 256 * rounds * 4;
        printf("{\"region_MiB\": %llu, \"wgs\": %d, \"mode\": %d, \"ms\": %.3f, \"Gloads_per_s\": %.1f}\n",
               (unsigned long long)(region >> 20), wgs, mode, ms, loads / ms / 1e6);
    }
    uint32_t h[128]; CK(hipMemcpy(h, d_hist, sizeof h, hipMemcpyDeviceToHost));
    printf("blockIdx%%8 -> XCC_ID histogram (last launch, 3 reps):\n");
    for (int i = 0; i < 8; i++) { printf("  %d:", i); for (int x = 0; x < 16; x++) if (h[i * 16 + x]) printf(" xcc%d=%u", x, h[i * 16 + x]); printf("\n"); }
    return 0;
}
