// How fast does the MFMA-epilogue store pattern (8 bytes per lane, 16 rows x 32 contiguous bytes per wave instruction)
// stream to HBM compared with row-contiguous 16-byte stores of the same tile?   hipcc --offload-arch=gfx950 -O3 store_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// tile = 32 rows x 64 columns of bf16 (128 B per row) per wave iteration, N columns per row in memory
template <int MODE>
__global__ __launch_bounds__(512) void k(unsigned short* out, long rows, int N, int twice) {
    const int l = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long strips = rows / 32;
    for (long s = (long)blockIdx.x * 8 + wave; s < strips; s += (long)gridDim.x * 8) {
        for (int nb = 0; nb < N / 64; ++nb) {
            for (int rep = 0; rep <= twice; ++rep) {
                unsigned short* base = out + (long)rep * rows * N;
                if (MODE == 0) {            // MFMA pattern: lane = (row l&15, quad l>>4), 8 B per lane, per (mt, i)
                    for (int mt = 0; mt < 2; ++mt)
                        for (int i = 0; i < 4; ++i) {
                            const long m = s * 32 + mt * 16 + (l & 15);
                            const int n0 = nb * 64 + i * 16 + ((l >> 4) << 2);
                            *reinterpret_cast<uint2*>(base + m * N + n0) = make_uint2(l + i, s);
                        }
                } else {                    // row-contiguous: 8 lanes x 16 B per row, 8 rows per instruction
                    for (int it = 0; it < 4; ++it) {
                        const long m = s * 32 + it * 8 + (l >> 3);
                        const int n0 = nb * 64 + (l & 7) * 8;
                        *reinterpret_cast<uint4*>(base + m * N + n0) = make_uint4(l, s, it, nb);
                    }
                }
            }
        }
    }
}

int main() {
    const long rows = 262144; const int N = 448;
    unsigned short* d; hipMalloc(&d, rows * N * 2 * 2);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int twice = 0; twice < 2; ++twice)
        for (int mode = 0; mode < 2; ++mode) {
            float best = 1e9;
            for (int it = 0; it < 5; ++it) {
                hipEventRecord(a);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(512), dim3(512), 0, 0, d, rows, N, twice);
                else hipLaunchKernelGGL(k<1>, dim3(512), dim3(512), 0, 0, d, rows, N, twice);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
            }
            const double bytes = (double)rows * N * 2 * (twice + 1);
            printf("outputs %d  %s  %.1f us  %.2f TB/s\n", twice + 1, mode == 0 ? "mfma-pattern 8B" : "row-contig 16B", best * 1e3, bytes / best / 1e9);
        }
    return 0;
}
