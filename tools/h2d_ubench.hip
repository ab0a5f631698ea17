/*
 * h2d_ubench.hip -- what one frame's ingest copy costs: a 1080p I420 frame (3,110,400 B) from pinned host memory to HBM,
 * one copy per "call" as x264_encoder_encode makes it, split over 1 / 2 / 4 streams, waited for at once (depth 1) or one
 * call later (depth 2).  hipcc --offload-arch=gfx950 -O3 -o tools/h2d_ubench tools/h2d_ubench.hip
 */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
static double now() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
static void spin(hipEvent_t e) { while (hipEventQuery(e) == hipErrorNotReady) {} }
int main(int argc, char **argv) {
    const size_t fb = argc > 1 ? (size_t)atol(argv[1]) : 3110400;
    const int frames = 300, reps = 3;
    unsigned char *h, *d;
    CHECK(hipHostMalloc((void **)&h, fb * frames, hipHostMallocDefault));
    for (size_t i = 0; i < fb * frames; i += 4096) h[i] = (unsigned char)i;
    CHECK(hipMalloc((void **)&d, fb * 8));
    hipStream_t st[4];
    hipEvent_t ev[8][4];
    for (int i = 0; i < 4; i++) CHECK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
    for (int i = 0; i < 8; i++) for (int j = 0; j < 4; j++) CHECK(hipEventCreateWithFlags(&ev[i][j], hipEventDisableTiming));
    for (int depth = 1; depth <= 3; depth++)
        for (int k = 1; k <= 4; k *= 2)
            for (int blocking = 0; blocking < 2; blocking++) {
                double best = 0;
                for (int r = 0; r < reps; r++) {
                    CHECK(hipDeviceSynchronize());
                    const double t0 = now();
                    for (int f = 0; f < frames; f++) {
                        const size_t chunk = ((fb / k) + 255) & ~(size_t)255;
                        for (int j = 0; j < k; j++) {
                            const size_t off = j * chunk, len = off + chunk <= fb ? chunk : fb - off;
                            CHECK(hipMemcpyAsync(d + (f & 7) * fb + off, h + f * fb + off, len, hipMemcpyHostToDevice, st[j]));
                            CHECK(hipEventRecord(ev[f & 7][j], st[j]));
                        }
                        if (f >= depth - 1)
                            for (int j = 0; j < k; j++) {
                                if (blocking) CHECK(hipEventSynchronize(ev[(f - (depth - 1)) & 7][j]));
                                else spin(ev[(f - (depth - 1)) & 7][j]);
                            }
                    }
                    CHECK(hipDeviceSynchronize());
                    const double gbs = fb * frames / (now() - t0) / 1e9;
                    if (gbs > best) best = gbs;
                }
                printf("frame %zu B  depth %d  streams %d  %-8s  %6.2f GB/s  (%.1f us per frame)\n", fb, depth, k, blocking ? "blocking" : "spin", best, fb / best / 1e3);
            }
    return 0;
}
