/*
 * sad_ubench.hip -- issue-rate micro-benchmark of the packed-SAD VALU
 * instructions on gfx950 ("measure first").
 * Prints giga-instructions/s and tera abs-diff/s per instruction for a grid
 * that fills every SIMD with `waves` waves.
 */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int OP>
__global__ __launch_bounds__(256) void k(unsigned *out, int iters, unsigned seed) {
    unsigned a[8], b = seed * 2654435761u + threadIdx.x, c = threadIdx.x * 0x01010101u + seed;
    unsigned long long q[4];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x + i;
#pragma unroll
    for (int i = 0; i < 4; i++) q[i] = threadIdx.x + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0) a[i] = __builtin_amdgcn_sad_u8(b + i, c, a[i]);
                if (OP == 1) a[i] = __builtin_amdgcn_msad_u8(b + i, c, a[i]);
                if (OP == 2) a[i] = __builtin_amdgcn_sad_u16(b + i, c, a[i]);
                if (OP == 3) q[i & 3] = __builtin_amdgcn_qsad_pk_u16_u8(((unsigned long long)b << 32) | (c + i), c, q[i & 3]);
                if (OP == 4) q[i & 3] = __builtin_amdgcn_mqsad_pk_u16_u8(((unsigned long long)b << 32) | (c + i), c, q[i & 3]);
                if (OP == 5) a[i] = __builtin_amdgcn_alignbyte(a[i], b + i, c);
                if (OP == 6) a[i] = a[i] + (b ^ (c + i));
                if (OP == 7) a[i] = __builtin_amdgcn_sad_hi_u8(b + i, c, a[i]);
            }
        }
        b += 0x10203u;
    }
    unsigned r = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r ^= a[i];
#pragma unroll
    for (int i = 0; i < 4; i++) r ^= (unsigned)q[i] ^ (unsigned)(q[i] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP>
static void run(const char *name, double absdiff_per_lane_instr, int waves_per_simd) {
    const int blocks = 256 * waves_per_simd; /* 256-thread blocks = 4 waves = one per SIMD of a CU */
    unsigned *out;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    const int iters = 4096;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 64, 1u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 2u);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double wave_instr = (double)blocks * 4 * iters * 64;
    const double lane_instr = wave_instr * 64;
    printf("%-22s waves/SIMD=%d  %8.3f ms  %8.1f G wave-instr/s  %7.2f cyc/wave-instr/SIMD@2.4GHz  %8.1f T absdiff/s\n", name,
           waves_per_simd, ms, wave_instr / ms / 1e6, (ms * 1e-3 * 2.4e9) / (wave_instr / 1024.0),
           lane_instr * absdiff_per_lane_instr / ms / 1e9);
    CHECK(hipFree(out));
}

int main() {
    for (int w = 1; w <= 4; w *= 2) {
        run<6>("v_add+xor (ref)", 0, w);
        run<0>("v_sad_u8", 4, w);
        run<7>("v_sad_hi_u8", 4, w);
        run<1>("v_msad_u8", 4, w);
        run<2>("v_sad_u16", 2, w);
        run<3>("v_qsad_pk_u16_u8", 16, w);
        run<4>("v_mqsad_pk_u16_u8", 16, w);
        run<5>("v_alignbyte_b32", 0, w);
    }
    return 0;
}
