// Micro-benchmark (developer tool): v_mfma_f32_16x16x4_f32 rate vs waves per SIMD and independent accumulators.
// build: hipcc --offload-arch=gfx950 -O3 mfma_occupancy.hip -o mfma_occupancy ; run: ./mfma_occupancy
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// NACC independent accumulators, VALU_PER = extra VALU ops (integer max) per 4 MFMAs
template <int NACC, int VALU_PER>
__global__ void k(float* out, int steps, int lo) {
    const int lane = threadIdx.x & 63;
    f32x4 c[NACC];
    for (int n = 0; n < NACC; ++n) c[n] = (f32x4){0, 0, 0, 0};
    float a = 0.5f + lane * 1e-3f, b = 1.0f + lane * 1e-4f;
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int r = 0; r < 64 / NACC; ++r) {
#pragma unroll
            for (int n = 0; n < NACC; ++n) {
                if (VALU_PER && (n % 4) == 0) {
#pragma unroll
                    for (int v = 0; v < VALU_PER; ++v) a = __int_as_float(max(__float_as_int(a), lo + v));
                }
                c[n] = MFMA16(a, b, c[n]);
            }
        }
    }
    float t = 0;
    for (int n = 0; n < NACC; ++n) t += c[n][0] + c[n][1] + c[n][2] + c[n][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

template <int NACC, int VALU_PER>
void run(float* out, int waves_per_simd) {
    const int steps = 400, blocks = 256 * 4, threads = 64 * waves_per_simd;   // 4 blocks per CU -> one per SIMD (approx.)
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    hipLaunchKernelGGL((k<NACC, VALU_PER>), dim3(blocks), dim3(threads), 0, 0, out, 10, (int)0x80000000);
    hipDeviceSynchronize();
    hipEventRecord(s);
    hipLaunchKernelGGL((k<NACC, VALU_PER>), dim3(blocks), dim3(threads), 0, 0, out, steps, (int)0x80000000);
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    double flops = (double)blocks * waves_per_simd * steps * 64 * 2048.0;
    printf("acc %2d valu/4mfma %d waves/SIMD~%d : %8.3f ms %7.1f TFLOP/s\n", NACC, VALU_PER, waves_per_simd, ms, flops / ms / 1e9);
}

int main() {
    float* out; hipMalloc(&out, 1024 * 1024 * 4);
    for (int w : {1, 2, 4}) {
        run<2, 0>(out, w); run<4, 0>(out, w); run<16, 0>(out, w);
        run<4, 1>(out, w); run<16, 1>(out, w); run<16, 2>(out, w); run<16, 4>(out, w);
    }
    return 0;
}
