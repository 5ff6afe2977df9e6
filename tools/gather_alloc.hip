// gather_alloc.hip -- does the way a table is allocated change what a random small read costs the memory side?
// hipMalloc (cached, 128-byte line fills) vs hipExtMallocWithFlags fine-grained / uncached (MTYPE UC: no L2 allocation),
// load widths 4 / 8 / 16 bytes.  Same loop as gather_ceiling (4 independent loads per lane per round).
//   gather_alloc <table_MiB> [wgs=8192] [rounds=16]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}
template <int W>
__global__ __launch_bounds__(256) void gather(const uint8_t *__restrict__ tab, uint64_t nbytes, int rounds, uint32_t *out)
{
    uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (int r = 0; r < rounds; r++) {
        const uint8_t *p[4];
#pragma unroll
        for (int k = 0; k < 4; k++) p[k] = tab + (__umul64hi(mix(id * 1315423911ull + (uint64_t)(r * 4 + k)), nbytes - 64) & ~15ull);
        if (W == 16) {
            uint4 v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = *reinterpret_cast<const uint4 *>(p[k]);
#pragma unroll
            for (int k = 0; k < 4; k++) acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
        } else if (W == 8) {
            uint2 v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = *reinterpret_cast<const uint2 *>(p[k]);
#pragma unroll
            for (int k = 0; k < 4; k++) acc ^= v[k].x ^ v[k].y;
        } else {
            uint32_t v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = *reinterpret_cast<const uint32_t *>(p[k]);
#pragma unroll
            for (int k = 0; k < 4; k++) acc ^= v[k];
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}
template <int W> int run(const char *alloc, const uint8_t *d, uint64_t nbytes, int rounds, int wgs, uint32_t *d_out)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(gather<W>, dim3(wgs), dim3(256), 0, 0, d, nbytes, rounds, d_out);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
    }
    double loads = (double)wgs * 256 * rounds * 4;
    printf("{\"alloc\": \"%s\", \"table_MiB\": %llu, \"load_bytes\": %d, \"loads\": %.0f, \"ms\": %.3f, \"Gloads_per_s\": %.2f}\n",
           alloc, (unsigned long long)(nbytes >> 20), W, loads, ms, loads / ms / 1e6);
    fflush(stdout);
    return 0;
}
int main(int argc, char **argv)
{
    uint64_t nbytes = (uint64_t)(argc > 1 ? atoll(argv[1]) : 1400) << 20;
    int wgs = argc > 2 ? atoi(argv[2]) : 8192, rounds = argc > 3 ? atoi(argv[3]) : 16;
    uint32_t *d_out; CK(hipMalloc((void **)&d_out, 64));
    const char *names[3] = {"hipMalloc", "finegrained", "uncached"};
    for (int mode = 0; mode < 3; mode++) {
        uint8_t *d = nullptr;
        hipError_t e = mode == 0 ? hipMalloc((void **)&d, nbytes)
                                 : hipExtMallocWithFlags((void **)&d, nbytes, mode == 1 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached);
        if (e != hipSuccess) { printf("{\"alloc\": \"%s\", \"error\": \"%s\"}\n", names[mode], hipGetErrorString(e)); continue; }
        CK(hipMemset(d, 0x5A, nbytes)); CK(hipDeviceSynchronize());
        if (run<16>(names[mode], d, nbytes, rounds, wgs, d_out)) return 1;
        if (run<8>(names[mode], d, nbytes, rounds, wgs, d_out)) return 1;
        if (run<4>(names[mode], d, nbytes, rounds, wgs, d_out)) return 1;
        CK(hipFree(d));
    }
    return 0;
}
