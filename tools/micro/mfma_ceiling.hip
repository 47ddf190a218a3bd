// Micro-benchmark (developer tool): issue ceiling of v_mfma_f32_16x16x4_f32 in the conv_ts loop shape.
//   variant 0: 16 MFMAs per step on 2 alternating accumulators, operands in registers
//   variant 1: + 8 ds_read2_b32-style LDS reads of the B operands per step (read -> wait -> MFMA)
//   variant 2: variant 1 with the B reads of step s+1 issued before the MFMAs of step s (double-buffered registers)
// build: hipcc --offload-arch=gfx950 -O3 mfma_ceiling.hip -o mfma_ceiling ; run: ./mfma_ceiling
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

template <int V>
__global__ __launch_bounds__(1024) void k(float* out, int steps) {
    __shared__ float W[27 * 1024];
    for (int e = threadIdx.x; e < 27 * 1024; e += 1024) W[e] = (float)(e & 7) * 0.001f;
    __syncthreads();
    const int lane = threadIdx.x & 63, i = lane & 15, kq = lane >> 4;
    const int bofs = (4 * kq) * 32 + (i ^ ((kq & 1) << 4));
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0;
    float a[8];
    for (int e = 0; e < 8; ++e) a[e] = 0.5f + e * 0.01f + lane * 1e-3f;
    float bl[8], bh[8], nl[8], nh[8];
    for (int e = 0; e < 8; ++e) { bl[e] = 1.f; bh[e] = 2.f; }
    int o = lane & 3;   // pseudo offset, wave-uniform enough for addressing purposes
    o = __builtin_amdgcn_readfirstlane(o);
    for (int s = 0; s < steps; ++s) {
        o = (o * 5 + 3) % 27;
        const float* wb = W + o * 1024 + bofs;
        const float* wc = W + o * 1024 + (bofs ^ 16);
        if (V == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { bl[e] = wb[e * 32]; bh[e] = wc[e * 32]; bl[4 + e] = wb[(16 + e) * 32]; bh[4 + e] = wc[(16 + e) * 32]; }
        }
        if (V == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { nl[e] = wb[e * 32]; nh[e] = wc[e * 32]; nl[4 + e] = wb[(16 + e) * 32]; nh[4 + e] = wc[(16 + e) * 32]; }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { c0 = MFMA16(a[e], bl[e], c0); c1 = MFMA16(a[e], bh[e], c1); }
        if (V == 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { bl[e] = nl[e]; bh[e] = nh[e]; }
        }
    }
    out[blockIdx.x * 1024 + threadIdx.x] = c0[0] + c0[1] + c0[2] + c0[3] + c1[0] + c1[1] + c1[2] + c1[3];
}

template <int V>
void run(const char* name, float* out) {
    const int steps = 2000, blocks = 256;
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    k<V><<<blocks, 1024>>>(out, 100);
    hipDeviceSynchronize();
    hipEventRecord(s);
    k<V><<<blocks, 1024>>>(out, steps);
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    double flops = (double)blocks * 16 * steps * 16 * 2048.0;
    printf("%-40s %8.3f ms  %7.1f TFLOP/s\n", name, ms, flops / ms / 1e9);
}

int main() {
    float* out; hipMalloc(&out, 256 * 1024 * 4);
    run<0>("mfma only (regs)", out);
    run<1>("+ B from LDS, read->wait->mfma", out);
    run<2>("+ B from LDS, prefetched one step", out);
    return 0;
}
