/*
 * ifetch_ubench.hip -- does straight-line code cost more to issue than a loop?  The search kernel (k_me_pool, venc_me3.hip) is
 * 34 KB of unrolled code that every wave walks once: ~1000 eight-byte vector instructions per wave with no reuse inside the
 * wave.  This runs the SAME number of v_sad_u16 (8 independent chains, inline assembly) per wave two ways:
 *   loop:     a body of 64 instructions (512 bytes) executed total / 64 times
 *   straight: a body of 1024 / 2048 / 4096 instructions (8 / 16 / 32 KB), executed total / body times
 * on 512-thread blocks, many blocks per CU in sequence (20 rounds), at 4 / 6 / 8 waves per SIMD.  Output: cycles per
 * wave-instruction per SIMD.
 *   hipcc --offload-arch=gfx950 -O3 -o tools/ifetch_ubench tools/ifetch_ubench.hip && tools/ifetch_ubench
 */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int BODY, int WAVES_PER_EU>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(WAVES_PER_EU, WAVES_PER_EU))) void k(unsigned *out, int reps, unsigned b, unsigned c) {
    unsigned a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x + i;
#pragma unroll 1
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int u = 0; u < BODY / 8; u++) {
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_sad_u16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
        }
    }
    unsigned x = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) x ^= a[i];
    out[(blockIdx.x * blockDim.x + threadIdx.x) & 0xffff] = x;
}

template <int BODY, int W>
static double run(unsigned *out, int blocks, int total) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<BODY, W>), dim3(blocks), dim3(512), 0, 0, out, total / BODY, 3u, 5u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<BODY, W>), dim3(blocks), dim3(512), 0, 0, out, total / BODY, 3u, 5u);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double wave_instr = (double)blocks * 8 * total;
    return (ms * 1e-3 * 2.4e9) / (wave_instr / 1024.0);
}

template <int W>
static void row(unsigned *out, int total) {
    /* blocks: 20 rounds of (W / 2) blocks of eight waves per CU */
    const int blocks = 256 * (W / 2) * 20;
    printf("%d waves/SIMD, %5d v_sad_u16 per wave:  loop of 64: %5.2f   straight 1024: %5.2f   2048: %5.2f   4096: %5.2f\n", W, total,
           run<64, W>(out, blocks, total), run<1024, W>(out, blocks, total), run<2048, W>(out, blocks, total), run<4096, W>(out, blocks, total));
}

int main() {
    unsigned *out;
    CHECK(hipMalloc(&out, 65536 * 4));
    printf("# cycles per wave-instruction per SIMD at 2.4 GHz nominal; every wave executes `total` instructions, as a loop or as straight-line code\n");
    for (int total : {4096, 8192}) {
        row<4>(out, total);
        row<6>(out, total);
        row<8>(out, total);
    }
    return 0;
}
