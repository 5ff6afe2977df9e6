// gather_ceiling.hip -- micro-benchmark: how many independent random 16-byte reads per second does
// an MI355X sustain from a table of a given size?  This is the ceiling of the probe step of the scan
// kernel (one 16-tag load per query k-mer, SURVEY.md section 8d asks for it to be measured).
//
//   gather_ceiling <table_MiB> <ilp> <aligned 0|1> [loads_per_lane=64] [wgs=4096]
//
// Every lane issues `ilp` independent loads per round at pseudo-random byte offsets (splitmix64 of a
// counter), xors the results and writes one word at the end so nothing is optimised away.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

template <int ILP, bool ALIGNED>
__global__ __launch_bounds__(256) void gather(const uint8_t *__restrict__ tab, uint64_t nbytes, uint64_t magic, int rounds,
                                             uint32_t *out)
{
    uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (int r = 0; r < rounds; r++) {
        uint4 v[ILP];
#pragma unroll
        for (int k = 0; k < ILP; k++) {
            uint64_t h = mix(id * 1315423911ull + (uint64_t)(r * ILP + k));
            uint64_t off = __umul64hi(h, nbytes - 64);          // uniform in [0, nbytes-64)
            if (ALIGNED) off &= ~15ull;
            __builtin_memcpy(&v[k], tab + off, 16);
        }
#pragma unroll
        for (int k = 0; k < ILP; k++) acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int ILP>
int run(bool aligned, const uint8_t *d, uint64_t nbytes, int rounds, int wgs, uint32_t *d_out, float *ms)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(a));
        if (aligned) hipLaunchKernelGGL((gather<ILP, true>), dim3(wgs), dim3(256), 0, 0, d, nbytes, 0ull, rounds, d_out);
        else hipLaunchKernelGGL((gather<ILP, false>), dim3(wgs), dim3(256), 0, 0, d, nbytes, 0ull, rounds, d_out);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(ms, a, b));
    }
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: %s table_MiB ilp aligned [loads_per_lane] [wgs]\n", argv[0]); return 2; }
    uint64_t nbytes = (uint64_t)atoll(argv[1]) << 20;
    int ilp = atoi(argv[2]);
    bool aligned = atoi(argv[3]) != 0;
    int per_lane = argc > 4 ? atoi(argv[4]) : 64;
    int wgs = argc > 5 ? atoi(argv[5]) : 4096;
    uint8_t *d = nullptr; uint32_t *d_out = nullptr;
    CK(hipMalloc((void **)&d, nbytes));
    CK(hipMalloc((void **)&d_out, 64));
    CK(hipMemset(d, 0x5A, nbytes));
    int rounds = per_lane / ilp;
    float ms = 0;
    int rc = 0;
    switch (ilp) {
    case 1: rc = run<1>(aligned, d, nbytes, rounds, wgs, d_out, &ms); break;
    case 2: rc = run<2>(aligned, d, nbytes, rounds, wgs, d_out, &ms); break;
    case 3: rc = run<3>(aligned, d, nbytes, rounds, wgs, d_out, &ms); break;
    case 4: rc = run<4>(aligned, d, nbytes, rounds, wgs, d_out, &ms); break;
    case 6: rc = run<6>(aligned, d, nbytes, rounds, wgs, d_out, &ms); break;
    case 8: rc = run<8>(aligned, d, nbytes, rounds, wgs, d_out, &ms); break;
    default: fprintf(stderr, "ilp must be 1,2,3,4,6,8\n"); return 2;
    }
    if (rc) return rc;
    double loads = (double)wgs * 256.0 * rounds * ilp;
    printf("{\"table_MiB\": %llu, \"ilp\": %d, \"aligned\": %d, \"wgs\": %d, \"loads\": %.0f, \"ms\": %.3f, "
           "\"Gloads_per_s\": %.2f, \"GBps_at_64B_per_load\": %.0f}\n",
           (unsigned long long)(nbytes >> 20), ilp, (int)aligned, wgs, loads, ms, loads / ms / 1e6, loads * 64 / ms / 1e6);
    return 0;
}
