// Micro-benchmark (developer tool): what does the conv_ts step SKELETON cost?  16 waves per CU, 110 KB of weights in LDS.
//   V0: 16 MFMAs per step, B from LDS (fixed offset sequence)                         -- the 132 TF reference
//   V1: + offsets popped from a 27-bit mask with ctz (scalar), new mask every 27 steps
//   V2: V1 + the A values pass 8 v_cndmask (row-valid select) and 8 integer-max ReLU
//   V3: V2 + queue rotation of 4 (o, idx) items + flags + "bubble" and "done" branches like TS_STEP
//   V4: V3 with the loop unrolled by 4 and a break after every step (the real structure)
// build: hipcc --offload-arch=gfx950 -O3 step_skeleton.hip -o step_skeleton
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

template <int V>
__global__ __launch_bounds__(1024) void k(float* out, const unsigned* masks, int tiles, int relu_in) {
    __shared__ float W[27 * 1024];
    for (int e = threadIdx.x; e < 27 * 1024; e += 1024) W[e] = (float)(e & 7) * 0.001f;
    __syncthreads();
    const int lane = threadIdx.x & 63, i = lane & 15, kq = lane >> 4;
    const int bofs = (4 * kq) * 32 + (i ^ ((kq & 1) << 4));
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0;
    f32x4 a0 = {0.5f + lane * 1e-3f, 0.6f, 0.7f, 0.8f}, a1 = {0.9f, 1.0f, 1.1f, 1.2f};
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int t = 0;
    unsigned m = masks[(blockIdx.x * 16 + wave) % 64];
    m = __builtin_amdgcn_readfirstlane(m);
    int oq0 = -1, oq1 = -1, oq2 = -1, oq3 = -1, iq0 = lane, iq1 = lane, iq2 = lane, iq3 = lane;
    bool fq0 = true, fq1 = false, fq2 = false, fq3 = false;
    if (V >= 3) {
        if (m) { oq0 = __builtin_ctz(m); m &= m - 1; }
        if (m) { oq1 = __builtin_ctz(m); m &= m - 1; }
        if (m) { oq2 = __builtin_ctz(m); m &= m - 1; }
        if (m) { oq3 = __builtin_ctz(m); m &= m - 1; }
    }
#define STEP()                                                                                       \
    do {                                                                                             \
        int o = 0;                                                                                   \
        if (V == 0) { o = (t * 5 + 3) % 27; ++t; }                                                   \
        if (V == 1 || V == 2) {                                                                      \
            if (m == 0) { ++t; m = __builtin_amdgcn_readfirstlane(masks[(blockIdx.x * 16 + wave + t) % 64]); } \
            o = __builtin_ctz(m); m &= m - 1;                                                        \
        }                                                                                            \
        int o4 = -1; bool f4 = false;                                                                \
        if (V >= 3) {                                                                                \
            if (m == 0 && t < tiles) { ++t; m = __builtin_amdgcn_readfirstlane(masks[(blockIdx.x * 16 + wave + t) % 64]); f4 = true; } \
            if (m && t < tiles) { o4 = __builtin_ctz(m); m &= m - 1; }                               \
            if (fq0) { out[blockIdx.x * 1024 + threadIdx.x] = c0[0] + c1[0]; c0 = c1 = (f32x4){0, 0, 0, 0}; } \
            o = oq0;                                                                                 \
        }                                                                                            \
        if (V < 3 || o >= 0) {                                                                       \
            f32x4 x0 = a0, x1 = a1;                                                                  \
            if (V >= 2) {                                                                            \
                if (iq0 < 0) { x0 = (f32x4){0, 0, 0, 0}; x1 = x0; }                                  \
                if (relu_in) {                                                                       \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                  \
                        x0[e] = __int_as_float(max(__float_as_int(x0[e]), 0));                       \
                        x1[e] = __int_as_float(max(__float_as_int(x1[e]), 0));                       \
                    }                                                                                \
                }                                                                                    \
            }                                                                                        \
            const float* wb = W + o * 1024 + bofs;                                                   \
            const float* wc = W + o * 1024 + (bofs ^ 16);                                            \
            float bl[8], bh[8];                                                                      \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) { bl[e] = wb[e * 32]; bh[e] = wc[e * 32]; bl[4 + e] = wb[(16 + e) * 32]; bh[4 + e] = wc[(16 + e) * 32]; } \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) { c0 = MFMA16(x0[e], bl[e], c0); c1 = MFMA16(x0[e], bh[e], c1); } \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) { c0 = MFMA16(x1[e], bl[4 + e], c0); c1 = MFMA16(x1[e], bh[4 + e], c1); } \
        }                                                                                            \
        if (V >= 3) {                                                                                \
            oq0 = oq1; oq1 = oq2; oq2 = oq3; oq3 = o4;                                               \
            iq0 = iq1; iq1 = iq2; iq2 = iq3; iq3 = lane + o4;                                        \
            fq0 = fq1; fq1 = fq2; fq2 = fq3; fq3 = f4;                                               \
        }                                                                                            \
    } while (0)
#define DONE() (oq0 < 0 && oq1 < 0 && oq2 < 0 && oq3 < 0 && !fq0 && !fq1 && !fq2 && !fq3 && t >= tiles)
    if (V <= 2) {
        const int steps = tiles * 27;
        for (int s = 0; s < steps; ++s) STEP();
    } else if (V == 3) {
        while (true) { STEP(); if (DONE()) break; }
    } else {
        while (true) {
            STEP(); if (DONE()) break;
            STEP(); if (DONE()) break;
            STEP(); if (DONE()) break;
            STEP(); if (DONE()) break;
        }
    }
    out[blockIdx.x * 1024 + threadIdx.x] = c0[0] + c0[1] + c0[2] + c0[3] + c1[0] + c1[1] + c1[2] + c1[3];
}

// V5: the structure proposed for conv_ts v3.  Items are popped from a 64-bit mask (current tile in bits 0..26, next tile in
// bits 27..53), so the prefetch stages run into the next tile without a branch; the tile loop runs popcount(mask) compute
// steps; the 4 named A-register sets keep rotating across tiles (the tile body exists once per starting phase).
__global__ __launch_bounds__(1024) void k5(float* out, const unsigned* masks, int tiles, int relu_in) {
    __shared__ float W[27 * 1024];
    for (int e = threadIdx.x; e < 27 * 1024; e += 1024) W[e] = (float)(e & 7) * 0.001f;
    __syncthreads();
    const int lane = threadIdx.x & 63, i = lane & 15, kq = lane >> 4;
    const int bofs = (4 * kq) * 32 + (i ^ ((kq & 1) << 4));
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int t = 0;
    unsigned long long M = __builtin_amdgcn_readfirstlane(masks[(blockIdx.x * 16 + wave) % 64]);
    M |= (unsigned long long)__builtin_amdgcn_readfirstlane(masks[(blockIdx.x * 16 + wave + 1) % 64]) << 27;
    f32x4 s0[2], s1[2], s2[2], s3[2];                  // A register sets (stand-ins for the gathered rows)
    int oq0, oq1, oq2, oq3;
#define POP(O) do { if (M) { O = __builtin_ctzll(M); M &= M - 1; } else O = -1; } while (0)
#define GATHER(S, O) do { S[0] = (f32x4){0.5f + (O) * 1e-3f + lane * 1e-4f, 0.6f, 0.7f, 0.8f}; S[1] = S[0] + 1.f; } while (0)
#define STEP5(CS, GS)                                                                                \
    do {                                                                                             \
        int o4; POP(o4);                                                                             \
        GATHER(GS, oq3);                                                                             \
        f32x4 x0 = CS[0], x1 = CS[1];                                                                \
        if (relu_in) {                                                                               \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                          \
                x0[e] = __int_as_float(max(__float_as_int(x0[e]), 0));                               \
                x1[e] = __int_as_float(max(__float_as_int(x1[e]), 0));                               \
            }                                                                                        \
        }                                                                                            \
        const int o = oq0 >= 27 ? oq0 - 27 : oq0;                                                    \
        const float* wb = W + o * 1024 + bofs;                                                       \
        const float* wc = W + o * 1024 + (bofs ^ 16);                                                \
        float bl[8], bh[8];                                                                          \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) { bl[e] = wb[e * 32]; bh[e] = wc[e * 32]; bl[4 + e] = wb[(16 + e) * 32]; bh[4 + e] = wc[(16 + e) * 32]; } \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) { c0 = MFMA16(x0[e], bl[e], c0); c1 = MFMA16(x0[e], bh[e], c1); } \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) { c0 = MFMA16(x1[e], bl[4 + e], c0); c1 = MFMA16(x1[e], bh[4 + e], c1); } \
        oq0 = oq1; oq1 = oq2; oq2 = oq3; oq3 = o4;                                                   \
    } while (0)
    POP(oq0); POP(oq1); POP(oq2); POP(oq3);
    GATHER(s0, oq0); GATHER(s1, oq1); GATHER(s2, oq2);
    int phase = 0;
    // tile body starting with set A as the compute set: n steps, then rotate the names by n & 3
#define TILE_BODY(A, B, C, D)                                                                        \
    {                                                                                                \
        int n = nk;                                                                                  \
        while (n >= 4) { STEP5(A, D); STEP5(B, A); STEP5(C, B); STEP5(D, C); n -= 4; }               \
        if (n >= 1) STEP5(A, D);                                                                     \
        if (n >= 2) STEP5(B, A);                                                                     \
        if (n >= 3) STEP5(C, B);                                                                     \
    }
    for (int tile = 0; tile < tiles; ++tile) {
        const int nk = __popcll(M & 0x7FFFFFFull) + ((oq0 >= 0 && oq0 < 27) + (oq1 >= 0 && oq1 < 27) + (oq2 >= 0 && oq2 < 27) + (oq3 >= 0 && oq3 < 27));
        if (phase == 0) TILE_BODY(s0, s1, s2, s3)
        else if (phase == 1) TILE_BODY(s1, s2, s3, s0)
        else if (phase == 2) TILE_BODY(s2, s3, s0, s1)
        else TILE_BODY(s3, s0, s1, s2)
        phase = (phase + nk) & 3;
        // tile boundary: write, shift the mask window, fetch the next-next tile's mask
        out[blockIdx.x * 1024 + threadIdx.x] = c0[0] + c1[0];
        c0 = c1 = (f32x4){0, 0, 0, 0};
        M >>= 27;
        oq0 -= 27; oq1 -= 27; oq2 -= 27; oq3 -= 27;       // queued items of the (old) next tile become current (bubbles stay < 0)
        ++t;
        if (t + 1 < tiles) M |= (unsigned long long)__builtin_amdgcn_readfirstlane(masks[(blockIdx.x * 16 + wave + t + 1) % 64]) << 27;
    }
    out[blockIdx.x * 1024 + threadIdx.x] += c0[0] + c0[1] + c0[2] + c0[3] + c1[0] + c1[1] + c1[2] + c1[3];
}

// V6/V7: where do the per-step VALU ops sit?  NV non-hoistable VALU ops per A value (they depend on the popped offset);
// V6 = all of them in a bunch before the 16 MFMAs, V7 = each A value finished right before its two MFMAs.
template <int INTERLEAVE, int NV>
__global__ __launch_bounds__(1024) void k67(float* out, const unsigned* masks, int tiles, int lo) {
    __shared__ float W[27 * 1024];
    for (int e = threadIdx.x; e < 27 * 1024; e += 1024) W[e] = (float)(e & 7) * 0.001f;
    __syncthreads();
    const int lane = threadIdx.x & 63, i = lane & 15, kq = lane >> 4;
    const int bofs = (4 * kq) * 32 + (i ^ ((kq & 1) << 4));
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0;
    float a[8];
    for (int e = 0; e < 8; ++e) a[e] = 0.5f + e * 0.01f + lane * 1e-3f;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned m = __builtin_amdgcn_readfirstlane(masks[(blockIdx.x * 16 + wave) % 64]);
    int t = 0;
    const int steps = tiles * 27;
    for (int s = 0; s < steps; ++s) {
        if (m == 0) { ++t; m = __builtin_amdgcn_readfirstlane(masks[(blockIdx.x * 16 + wave + t) % 64]); }
        const int o = __builtin_ctz(m); m &= m - 1;
        const float* wb = W + o * 1024 + bofs;
        const float* wc = W + o * 1024 + (bofs ^ 16);
        float bl[8], bh[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { bl[e] = wb[e * 32]; bh[e] = wc[e * 32]; bl[4 + e] = wb[(16 + e) * 32]; bh[4 + e] = wc[(16 + e) * 32]; }
        float x[8];
        if (!INTERLEAVE) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                int v = __float_as_int(a[e]) + o;
#pragma unroll
                for (int r = 0; r < NV; ++r) v = max(v, lo + r * o);
                x[e] = __int_as_float(v);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) { c0 = MFMA16(x[e], bl[e], c0); c1 = MFMA16(x[e], bh[e], c1); }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                int v = __float_as_int(a[e]) + o;
#pragma unroll
                for (int r = 0; r < NV; ++r) v = max(v, lo + r * o);
                x[e] = __int_as_float(v);
                c0 = MFMA16(x[e], bl[e], c0); c1 = MFMA16(x[e], bh[e], c1);
            }
        }
    }
    out[blockIdx.x * 1024 + threadIdx.x] = c0[0] + c0[1] + c0[2] + c0[3] + c1[0] + c1[1] + c1[2] + c1[3];
}

template <int INTERLEAVE, int NV>
void run67(float* out, const unsigned* masks) {
    const int tiles = 80, blocks = 256;
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    hipLaunchKernelGGL((k67<INTERLEAVE, NV>), dim3(blocks), dim3(1024), 0, 0, out, masks, 4, (int)0x80000000);
    hipDeviceSynchronize();
    hipEventRecord(s);
    hipLaunchKernelGGL((k67<INTERLEAVE, NV>), dim3(blocks), dim3(1024), 0, 0, out, masks, tiles, (int)0x80000000);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    double flops = (double)blocks * 16 * tiles * 27 * 16 * 2048.0;
    printf("V%d %2d dependent VALU ops per step, %-22s %8.3f ms  %7.1f TFLOP/s\n", INTERLEAVE ? 7 : 6, 8 * (NV + 1),
           INTERLEAVE ? "interleaved with MFMA" : "bunched before MFMA", ms, flops / ms / 1e9);
    fflush(stdout);
}

template <int V>
void run(const char* name, float* out, const unsigned* masks) {
    const int tiles = 80, blocks = 256;
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    k<V><<<blocks, 1024>>>(out, masks, 4, 1);
    hipDeviceSynchronize();
    hipEventRecord(s);
    k<V><<<blocks, 1024>>>(out, masks, tiles, 1);
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    double flops = (double)blocks * 16 * tiles * 27 * 16 * 2048.0;
    printf("%-58s %8.3f ms  %7.1f TFLOP/s\n", name, ms, flops / ms / 1e9); fflush(stdout);
}

int main() {
    float* out; hipMalloc(&out, 256 * 1024 * 4);
    unsigned h[64]; for (int j = 0; j < 64; ++j) h[j] = 0x7FFFFFFu;      // every tile has all 27 offsets
    unsigned* masks; hipMalloc(&masks, sizeof(h)); hipMemcpy(masks, h, sizeof(h), hipMemcpyHostToDevice);
    run<0>("V0 mfma + LDS B", out, masks);
    run<1>("V1 + ctz offset popping", out, masks);
    run<2>("V2 + cndmask/relu on A", out, masks);
    run<3>("V3 + item queue, flags, bubble/done branches", out, masks);
    run<4>("V4 + unrolled by 4 with breaks", out, masks);
    {
        const int tiles = 80, blocks = 256;
        hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
        k5<<<blocks, 1024>>>(out, masks, 4, 1);
        hipDeviceSynchronize();
        hipEventRecord(s);
        k5<<<blocks, 1024>>>(out, masks, tiles, 1);
        hipEventRecord(e); hipEventSynchronize(e);
        float ms; hipEventElapsedTime(&ms, s, e);
        double flops = (double)blocks * 16 * tiles * 27 * 16 * 2048.0;
        printf("%-58s %8.3f ms  %7.1f TFLOP/s\n", "V5 per-tile trip counts, 64-bit mask window, 4 phases", ms, flops / ms / 1e9); fflush(stdout);
    }
    run67<0, 1>(out, masks); run67<1, 1>(out, masks);
    run67<0, 3>(out, masks); run67<1, 3>(out, masks);
    run67<0, 7>(out, masks); run67<1, 7>(out, masks);
    return 0;
}
