// valu_rates.hip -- issue rate of the integer multiplies the encode / split arithmetic can be built from
// (gfx950).  Each kernel runs ITER x 16 independent instructions of one kind per lane (inline asm, 16 accumulators)
// on every SIMD (256 CUs x 4 SIMDs x 2 waves); prints wave-instructions per ns per SIMD-clock equivalents.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/valu_rates.bin tools/valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITER 2048

#define BODY16(ASM)                                                                                                  \
    _Pragma("unroll 1") for (int it = 0; it < ITER; it++) {                                                          \
        asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7) ASM(8) ASM(9) ASM(10) ASM(11) ASM(12)    \
                         ASM(13) ASM(14) ASM(15)                                                                     \
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), \
                       "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]),       \
                       "+v"(a[15])                                                                                   \
                     : "v"(m));                                                                                      \
    }

#define A_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %16\n"
#define A_MULHI(i) "v_mul_hi_u32 %" #i ", %" #i ", %16\n"
#define A_MUL24(i) "v_mul_u32_u24 %" #i ", %" #i ", %16\n"
#define A_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %16, %" #i "\n"
#define A_ADD(i) "v_add_u32 %" #i ", %" #i ", %16\n"
#define A_XOR(i) "v_xor_b32 %" #i ", %" #i ", %16\n"
#define A_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 3, %16\n"

#define KERNEL(NAME, ASM)                                                          \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t m)         \
    {                                                                              \
        uint32_t a[16];                                                            \
        for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 16 + i;                  \
        BODY16(ASM)                                                                \
        uint32_t s = 0;                                                            \
        for (int i = 0; i < 16; i++) s ^= a[i];                                    \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                   \
    }

KERNEL(k_mullo, A_MULLO)
KERNEL(k_mulhi, A_MULHI)
KERNEL(k_mul24, A_MUL24)
KERNEL(k_mad24, A_MAD24)
KERNEL(k_add, A_ADD)
KERNEL(k_xor, A_XOR)
KERNEL(k_lshladd, A_LSHLADD)

__global__ __launch_bounds__(256) void k_mad64(uint32_t *out, uint32_t m)
{
    uint64_t a[8];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 8 + i;
#pragma unroll 1
    for (int it = 0; it < ITER; it++) {
#define M64(i) "v_mad_u64_u32 %" #i ", vcc, %8, %8, %" #i "\n"
        asm volatile(M64(0) M64(1) M64(2) M64(3) M64(4) M64(5) M64(6) M64(7) M64(0) M64(1) M64(2) M64(3) M64(4) M64(5) M64(6) M64(7)
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(m)
                     : "vcc");
    }
    uint64_t s = 0;
    for (int i = 0; i < 8; i++) s ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}

// LDS read rates: ds_read_u8 / ds_read_b32, stride-1 addresses
__global__ __launch_bounds__(256) void k_ds_u8(uint32_t *out, uint32_t m)
{
    __shared__ uint8_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = (uint8_t)i;
    __syncthreads();
    uint32_t s = 0, at = threadIdx.x;
#pragma unroll 1
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int k = 0; k < 16; k++) s += ((volatile uint8_t *)lds)[(at + k * 3) & 4095];
        at += m;
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_ds_b32(uint32_t *out, uint32_t m)
{
    __shared__ uint32_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i;
    __syncthreads();
    uint32_t s = 0, at = threadIdx.x;
#pragma unroll 1
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int k = 0; k < 16; k++) s += ((volatile uint32_t *)lds)[(at + k * 3) & 4095];
        at += m;
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename K>
void run(const char *name, K kern, uint32_t *d_out)
{
    const int grid = 256 * 2;    // 2 workgroups of 4 waves per CU -> 2 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d_out, 3u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d_out, 3u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 2 waves x ITER x 16 wave-instructions
    double per_simd = 2.0 * ITER * 16;
    printf("{\"op\": \"%s\", \"ms\": %.4f, \"ns_per_wave_instr_per_simd\": %.3f}\n", name, ms, ms * 1e6 / per_simd);
}

int main()
{
    uint32_t *d_out;
    hipMalloc(&d_out, 256 * 2 * 256 * 4);
    run("v_add_u32", k_add, d_out);
    run("v_xor_b32", k_xor, d_out);
    run("v_lshl_add_u32", k_lshladd, d_out);
    run("v_mul_u32_u24", k_mul24, d_out);
    run("v_mad_u32_u24", k_mad24, d_out);
    run("v_mul_lo_u32", k_mullo, d_out);
    run("v_mul_hi_u32", k_mulhi, d_out);
    run("v_mad_u64_u32", k_mad64, d_out);
    run("ds_read_u8 x16", k_ds_u8, d_out);
    run("ds_read_b32 x16", k_ds_b32, d_out);
    hipFree(d_out);
    return 0;
}
