// gather_flavors.hip -- does a cache-policy modifier change what a random 16-byte read costs the memory side?
// Same loop as gather_ceiling, the load issued as inline asm with different modifiers.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}
// Four loads and their wait in ONE asm statement with early-clobber outputs: the compiler must not reuse a
// destination register (e.g. for the next address) while a load is still in flight.
#define LD4(MOD)                                                                                              \
    asm volatile("global_load_dwordx4 %0, %4, off " MOD "\n\tglobal_load_dwordx4 %1, %5, off " MOD "\n\t"       \
                 "global_load_dwordx4 %2, %6, off " MOD "\n\tglobal_load_dwordx4 %3, %7, off " MOD "\n\t"       \
                 "s_waitcnt vmcnt(0)"                                                                          \
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory")
template <int FLAVOR>
__global__ __launch_bounds__(256) void gather(const uint8_t *__restrict__ tab, uint64_t nbytes, int rounds, uint32_t *out)
{
    uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (int r = 0; r < rounds; r++) {
        const uint8_t *p0 = tab + (__umul64hi(mix(id * 1315423911ull + (uint64_t)(r * 4 + 0)), nbytes - 64) & ~15ull);
        const uint8_t *p1 = tab + (__umul64hi(mix(id * 1315423911ull + (uint64_t)(r * 4 + 1)), nbytes - 64) & ~15ull);
        const uint8_t *p2 = tab + (__umul64hi(mix(id * 1315423911ull + (uint64_t)(r * 4 + 2)), nbytes - 64) & ~15ull);
        const uint8_t *p3 = tab + (__umul64hi(mix(id * 1315423911ull + (uint64_t)(r * 4 + 3)), nbytes - 64) & ~15ull);
        uint4 v0, v1, v2, v3;
        if (FLAVOR == 0) LD4("");
        if (FLAVOR == 1) LD4("nt");
        if (FLAVOR == 2) LD4("sc1");
        if (FLAVOR == 3) LD4("sc0 sc1");
        if (FLAVOR == 4) LD4("sc0 sc1 nt");
        if (FLAVOR == 5) LD4("sc0");
        acc ^= v0.x ^ v0.y ^ v0.z ^ v0.w ^ v1.x ^ v1.y ^ v1.z ^ v1.w ^ v2.x ^ v2.y ^ v2.z ^ v2.w ^ v3.x ^ v3.y ^ v3.z ^ v3.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
template <int F> int run(const uint8_t *d, uint64_t nbytes, int rounds, int wgs, uint32_t *d_out)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(gather<F>, dim3(wgs), dim3(256), 0, 0, d, nbytes, rounds, d_out);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
    }
    double loads = (double)wgs * 256 * rounds * 4;
    printf("{\"table_MiB\": %llu, \"flavor\": %d, \"loads\": %.0f, \"ms\": %.3f, \"Gloads_per_s\": %.2f}\n",
           (unsigned long long)(nbytes >> 20), F, loads, ms, loads / ms / 1e6);
    return 0;
}
int main(int argc, char **argv)
{
    uint64_t nbytes = (uint64_t)(argc > 1 ? atoll(argv[1]) : 1400) << 20;
    int only = argc > 2 ? atoi(argv[2]) : -1;
    uint8_t *d; uint32_t *d_out;
    CK(hipMalloc((void **)&d, nbytes)); CK(hipMalloc((void **)&d_out, 64)); CK(hipMemset(d, 0x5A, nbytes));
    int rounds = 16, wgs = 8192;
    if (only < 0 || only == 0) run<0>(d, nbytes, rounds, wgs, d_out);
    if (only < 0 || only == 1) run<1>(d, nbytes, rounds, wgs, d_out);
    if (only < 0 || only == 2) run<2>(d, nbytes, rounds, wgs, d_out);
    if (only < 0 || only == 3) run<3>(d, nbytes, rounds, wgs, d_out);
    if (only < 0 || only == 4) run<4>(d, nbytes, rounds, wgs, d_out);
    if (only < 0 || only == 5) run<5>(d, nbytes, rounds, wgs, d_out);
    return 0;
}
