// lds_unaligned.hip -- does gfx950 serve the unaligned LDS reads hipcc emits for byte-aligned memcpy (ds_read_b32 / b64 /
// b128, and the ds_read2_b64 its load/store optimizer merges two of them into)?  The encode stage (kg_device.hpp) and
// the second-level probe (kg_partition2.hpp) rely on them.  Every lane reads at every byte offset 0..63 and compares
// with a byte-wise read.  Prints one JSON line; exit code 1 on a mismatch.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 2; } } while (0)
struct B16 { uint32_t w[4]; };
__global__ void probe(uint32_t *bad)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[2048];
    for (uint32_t i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = (uint8_t)(i * 37u + 11u);
    __syncthreads();
    uint32_t nbad = 0;
    for (uint32_t shift = 0; shift < 64; shift++) {
        const uint32_t a = threadIdx.x * 3u + shift;           // every alignment, lanes at different alignments
        uint32_t x4; uint2 x8, y8; B16 x16;
        __builtin_memcpy(&x4, lds + a, 4);
        __builtin_memcpy(&x8, lds + a, 8);
        __builtin_memcpy(&y8, lds + a + 240, 8);               // (merged with the read above into one ds_read2_b64)
        __builtin_memcpy(&x16, lds + a, 16);
        uint8_t ref[16], ref2[8];
        for (int k = 0; k < 16; k++) ref[k] = (uint8_t)((a + k) * 37u + 11u);
        for (int k = 0; k < 8; k++) ref2[k] = (uint8_t)((a + 240 + k) * 37u + 11u);
        uint32_t r4; uint2 r8, s8; B16 r16;
        __builtin_memcpy(&r4, ref, 4); __builtin_memcpy(&r8, ref, 8); __builtin_memcpy(&s8, ref2, 8); __builtin_memcpy(&r16, ref, 16);
        nbad += x4 != r4;
        nbad += x8.x != r8.x || x8.y != r8.y;
        nbad += y8.x != s8.x || y8.y != s8.y;
        nbad += x16.w[0] != r16.w[0] || x16.w[1] != r16.w[1] || x16.w[2] != r16.w[2] || x16.w[3] != r16.w[3];
    }
    if (nbad) atomicAdd(bad, nbad);
}
int main()
{
    uint32_t *d = nullptr, h = 0;
    CK(hipMalloc((void **)&d, 4));
    CK(hipMemset(d, 0, 4));
    hipLaunchKernelGGL(probe, dim3(4), dim3(256), 0, 0, d);
    CK(hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost));
    printf("{\"unaligned_lds_reads_wrong\": %u}\n", h);
    return h ? 1 : 0;
}
