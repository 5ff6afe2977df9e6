// gather_width.hip -- does the WIDTH of a random L2-resident read change its cost?  Random reads of 1 / 2 / 4 / 8 / 16 bytes
// from a table of argv[1] MiB (default 2: resident in every XCD's L2, the tag pass's situation), four independent loads per
// lane and round, same loop as gather_flavors.hip.  Prints loads per second per width.  (Tuning aid, round 4: would a
// one-byte exact home index be cheaper to probe than the 16-tag window?)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}
#define LD4(INS, T)                                                                                            \
    { T v0, v1, v2, v3;                                                                                        \
    asm volatile(INS " %0, %4, off\n\t" INS " %1, %5, off\n\t" INS " %2, %6, off\n\t" INS " %3, %7, off\n\t"   \
                 "s_waitcnt vmcnt(0)"                                                                          \
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory"); \
    acc ^= fold(v0) ^ fold(v1) ^ fold(v2) ^ fold(v3); }
__device__ __forceinline__ uint32_t fold(uint32_t v) { return v; }
__device__ __forceinline__ uint32_t fold(uint2 v) { return v.x ^ v.y; }
__device__ __forceinline__ uint32_t fold(uint4 v) { return v.x ^ v.y ^ v.z ^ v.w; }
template <int W>
__global__ __launch_bounds__(256) void gather(const uint8_t *__restrict__ tab, uint64_t nbytes, int rounds, uint32_t *out)
{
    uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (int r = 0; r < rounds; r++) {
        const uint8_t *p0 = tab + (__umul64hi(mix(id * 1315423911ull + (uint64_t)(r * 4 + 0)), nbytes - 64) & ~(uint64_t)(W - 1));
        const uint8_t *p1 = tab + (__umul64hi(mix(id * 1315423911ull + (uint64_t)(r * 4 + 1)), nbytes - 64) & ~(uint64_t)(W - 1));
        const uint8_t *p2 = tab + (__umul64hi(mix(id * 1315423911ull + (uint64_t)(r * 4 + 2)), nbytes - 64) & ~(uint64_t)(W - 1));
        const uint8_t *p3 = tab + (__umul64hi(mix(id * 1315423911ull + (uint64_t)(r * 4 + 3)), nbytes - 64) & ~(uint64_t)(W - 1));
        if (W == 1) LD4("global_load_ubyte", uint32_t)
        if (W == 2) LD4("global_load_ushort", uint32_t)
        if (W == 4) LD4("global_load_dword", uint32_t)
        if (W == 8) LD4("global_load_dwordx2", uint2)
        if (W == 16) LD4("global_load_dwordx4", uint4)
    }
    if (acc == 0x12345678u) out[0] = acc;
}
template <int W> int run(const uint8_t *d, uint64_t nbytes, int rounds, int wgs, uint32_t *d_out)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(gather<W>, dim3(wgs), dim3(256), 0, 0, d, nbytes, rounds, d_out);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
    }
    double loads = (double)wgs * 256 * rounds * 4;
    printf("{\"table_MiB\": %llu, \"width_bytes\": %d, \"loads\": %.0f, \"ms\": %.3f, \"Gloads_per_s\": %.2f}\n",
           (unsigned long long)(nbytes >> 20), W, loads, ms, loads / ms / 1e6);
    return 0;
}
int main(int argc, char **argv)
{
    uint64_t nbytes = (uint64_t)(argc > 1 ? atoll(argv[1]) : 2) << 20;
    uint8_t *d; uint32_t *d_out;
    CK(hipMalloc((void **)&d, nbytes)); CK(hipMalloc((void **)&d_out, 64)); CK(hipMemset(d, 0x5A, nbytes));
    int rounds = 16, wgs = 8192;
    if (run<1>(d, nbytes, rounds, wgs, d_out)) return 1;
    if (run<2>(d, nbytes, rounds, wgs, d_out)) return 1;
    if (run<4>(d, nbytes, rounds, wgs, d_out)) return 1;
    if (run<8>(d, nbytes, rounds, wgs, d_out)) return 1;
    if (run<16>(d, nbytes, rounds, wgs, d_out)) return 1;
    return 0;
}
