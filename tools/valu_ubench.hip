/*
 * valu_ubench.hip -- issue cost of the vector instructions the search and transform kernels are made of, on gfx950:
 * cycles per wave-instruction per SIMD at 1 / 2 / 4 / 6 waves per SIMD (8 independent chains per wave, inline
 * assembly so that nothing is folded).  The cost model behind DESIGN.md section 4.1's instruction budget.
 *   hipcc --offload-arch=gfx950 -O3 -o tools/valu_ubench tools/valu_ubench.hip && tools/valu_ubench
 */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

#define OPS(X) \
    X(0, "v_add_u32", "v_add_u32 %0, %1, %0") \
    X(1, "v_and_b32", "v_and_b32 %0, %1, %0") \
    X(2, "v_lshl_or_b32", "v_lshl_or_b32 %0, %0, 3, %1") \
    X(3, "v_and_or_b32", "v_and_or_b32 %0, %0, %1, %2") \
    X(4, "v_min_u32", "v_min_u32 %0, %1, %0") \
    X(5, "v_min3_u32", "v_min3_u32 %0, %0, %1, %2") \
    X(6, "v_cndmask_b32", "v_cndmask_b32 %0, %0, %1, vcc") \
    X(7, "v_perm_b32", "v_perm_b32 %0, %0, %1, %2") \
    X(8, "v_alignbyte_b32", "v_alignbyte_b32 %0, %0, %1, %2") \
    X(9, "v_sad_u8", "v_sad_u8 %0, %1, %2, %0") \
    X(10, "v_sad_u16", "v_sad_u16 %0, %1, %2, %0") \
    X(11, "v_pk_add_u16", "v_pk_add_u16 %0, %0, %1") \
    X(12, "v_pk_sub_i16", "v_pk_sub_i16 %0, %0, %1") \
    X(13, "v_pk_max_i16", "v_pk_max_i16 %0, %0, %1") \
    X(14, "v_pk_min_u16", "v_pk_min_u16 %0, %0, %1") \
    X(15, "v_pk_mad_u16", "v_pk_mad_u16 %0, %0, %1, %2") \
    X(16, "v_pk_lshrrev_b16", "v_pk_lshrrev_b16 %0, 1, %0") \
    X(17, "v_mul_u32_u24", "v_mul_u32_u24 %0, %0, %1") \
    X(18, "v_mad_u32_u24", "v_mad_u32_u24 %0, %0, %1, %2") \
    X(19, "v_mul_lo_u32", "v_mul_lo_u32 %0, %0, %1") \
    X(20, "v_mad_u64_u32", "v_mad_u64_u32 %0, vcc, %1, %2, %0") \
    X(21, "v_add_u32 dpp row_shr:1", "v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf") \
    X(22, "v_mov_b32 dpp quad_perm", "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") \
    X(23, "v_sub_u16 sdwa", "v_sub_u16_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1") \
    X(24, "v_ashrrev_i32", "v_ashrrev_i32 %0, 1, %0") \
    X(25, "v_max_i32", "v_max_i32 %0, %1, %0") \
    X(26, "v_med3_i32", "v_med3_i32 %0, %0, %1, %2") \
    X(27, "v_add3_u32", "v_add3_u32 %0, %0, %1, %2") \
    X(28, "v_lshlrev_b32", "v_lshlrev_b32 %0, 1, %0") \
    X(29, "v_sub_u32", "v_sub_u32 %0, %0, %1") \
    X(30, "v_xad_u32", "v_xad_u32 %0, %0, %1, %2") \
    X(31, "v_bfe_u32", "v_bfe_u32 %0, %0, 3, 8") \
    X(32, "v_cmp_lt_u32 (vcc)", "v_cmp_lt_u32 vcc, %0, %1") \
    X(33, "v_pk_sub_u16 clamp", "v_pk_sub_u16 %0, %0, %1 clamp") \
    X(34, "v_pk_max_u16", "v_pk_max_u16 %0, %0, %1") \
    X(35, "v_msad_u8", "v_msad_u8 %0, %1, %2, %0") \
    X(36, "v_lshrrev_b64", "v_lshrrev_b64 %0, 8, %0") \
    X(37, "v_mbcnt_lo", "v_mbcnt_lo_u32_b32 %0, %1, %0")

template <int OP>
__global__ __launch_bounds__(256) void k(unsigned *out, int iters, unsigned seed) {
    unsigned a[8];
    unsigned long long q[8];
    unsigned b = seed * 2654435761u + threadIdx.x, c = threadIdx.x * 0x01010101u + seed;
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x + i, q[i] = threadIdx.x + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
#define X(n, name, text)                                                                                          \
    if (OP == n) {                                                                                                \
        if (n == 20 || n == 36) asm volatile(text : "+v"(q[i]) : "v"(b), "v"(c));                               \
        else asm volatile(text : "+v"(a[i]) : "v"(b), "v"(c) : "vcc");                                            \
    }
                OPS(X)
#undef X
            }
        }
    }
    unsigned r = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r ^= a[i] ^ (unsigned)q[i] ^ (unsigned)(q[i] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

/* LDS instructions: cycles per wave-instruction per CU (all waves of a CU share the LDS) */
template <int OP>
__global__ __launch_bounds__(256) void kl(unsigned *out, int iters, unsigned stride) {
    __shared__ __attribute__((aligned(16))) unsigned s[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) s[i] = i;
    __syncthreads();
    unsigned acc = 0;
    const unsigned base = (threadIdx.x & 63) * stride;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const unsigned o = (base + u * 4 + (it & 1)) & 2047u;
            if (OP == 0) acc += s[o];
            if (OP == 1) { uint2 v = *(const uint2 *)&s[o & ~1u]; acc += v.x ^ v.y; }
            if (OP == 2) { uint4 v = *(const uint4 *)&s[o & ~3u]; acc += v.x ^ v.y ^ v.z ^ v.w; }
            if (OP == 3) acc += s[o] + s[o + 2048 + 64];   /* ds_read2_b32 (offsets far apart in units of dwords) */
            if (OP == 4) acc += ((const unsigned short *)s)[o];
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

static double run_time(void (*launch)(unsigned *, int, int), unsigned *out, int blocks, int iters) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    launch(out, blocks, 16);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    launch(out, blocks, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

template <int OP>
static void launch_k(unsigned *out, int blocks, int iters) { hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 2u); }
template <int OP>
static void launch_l(unsigned *out, int blocks, int iters) { hipLaunchKernelGGL(kl<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 1u); }
template <int OP>
static void launch_l33(unsigned *out, int blocks, int iters) { hipLaunchKernelGGL(kl<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 33u); }

int main() {
    unsigned *out;
    CHECK(hipMalloc(&out, (size_t)256 * 8 * 256 * 4));
    const int iters = 2048;
    printf("# cycles per wave-instruction per SIMD at 2.4 GHz nominal (the chip may clock lower under load); 256 CUs x 4 SIMDs\n");
    printf("%-28s %8s %8s %8s %8s\n", "instruction", "1 w/SIMD", "2", "4", "6");
#define X(n, name, text)                                                                        \
    {                                                                                           \
        printf("%-28s", name);                                                                  \
        for (int w : {1, 2, 4, 6}) {                                                            \
            const double ms = run_time(launch_k<n>, out, 256 * w, iters);                       \
            const double wave_instr = (double)256 * w * 4 * iters * 64;                         \
            printf(" %8.2f", (ms * 1e-3 * 2.4e9) / (wave_instr / 1024.0));                      \
        }                                                                                       \
        printf("\n");                                                                           \
    }
    OPS(X)
#undef X
    printf("# LDS: cycles per wave-instruction per CU (conflict-free lane stride 1 dword unless noted)\n");
    const char *ln[] = {"ds_read_b32", "ds_read_b64", "ds_read_b128", "ds_read2_b32", "ds_read_u16"};
    void (*lf[])(unsigned *, int, int) = {launch_l<0>, launch_l<1>, launch_l<2>, launch_l<3>, launch_l<4>};
    for (int o = 0; o < 5; o++) {
        printf("%-28s", ln[o]);
        for (int w : {1, 2, 4, 6}) {
            const double ms = run_time(lf[o], out, 256 * w, iters);
            const double wave_instr = (double)256 * w * 4 * iters * 16;
            printf(" %8.2f", (ms * 1e-3 * 2.4e9) / (wave_instr / 256.0));
        }
        printf("\n");
    }
    CHECK(hipFree(out));
    return 0;
}
