// conv_os: output-stationary rule-list convolution -- the hot kernel of the sparse backbone.
//
//   Y[r] = residual[r] + bias + sum_o sum_{(i,r) in R_o} in(X[i]) . W[o']        r in a block of R output rows
//
// Why this shape (measured in profiles/r1a): a [27][N] neighbour table walked 32 rows at a time
// executes 2-3x the useful MFMA work on surface-like clouds, because the rows of a tile rarely share their offsets.
// The canonical rule list (offset-major, output row ascending) makes the rules of one offset that land in a block of R
// consecutive output rows a CONTIGUOUS range [bstart[o][b], bstart[o][b+1]).  A workgroup owns such a block:
//   * it walks those ranges in dense tiles of 16 rules (v_mfma_f32_16x16x4_f32): executed/useful = 1.05-1.15,
//   * gathers the 16 input rows of a tile straight into the A fragment (one float4 per lane per 16 channels),
//   * reads the B fragment (weights) from LDS, where the [n_off][32 x 32] weight slice of the current 32-channel
//     K-chunk and 32-column output chunk is staged once per workgroup and reused by every tile,
//   * accumulates the 16 x 32 tile result into the block's [R][32] output slab in LDS with ds_add_f32
//     (an output row receives one contribution per offset, from whichever wave processed that tile),
//   * writes the slab back with coalesced 128-B row stores, fusing bias / residual (AddTable) / ReLU-backward mask.
// No global atomics; HBM traffic is the compulsory X gather + one Y write.
//
// MFMA operand maps (cdna_hip_programming.md §3): lane l, i = l & 15, kq = l >> 4.
//   A[i][k]: rule i of the tile;  B[k][j]: j = l & 15 output column;  step (half, e) contracts the channels
//   k = 16*half + 4*kq + e, kq = 0..3 -- a fixed permutation of the summation order, identical for A and B, which
//   lets a lane fetch 4 consecutive channels with ONE 16-byte access (global float4 for A, ds_read_b128 for B).
//   C/D: acc[j] = D[4*kq + j][l & 15].
#include "scn_common.h"

using scn::S;
using scn::cdiv;

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

static constexpr int OS_THREADS = 256;          // 4 waves; 2 workgroups per CU give each SIMD a partner wave
static constexpr int OS_KC = 32;                // channels per K-chunk
static constexpr int OS_CT = 32;                // output columns per workgroup
static constexpr int OS_T = 16;                 // rules per tile
static constexpr int OS_RMAX = 1024;            // max output rows per workgroup

// LDS image of the weight slice: Ws[o][n][k], k contiguous, 16-byte chunks XOR-swizzled by (n & 7) so that the
// 16 lanes of a ds_read_b128 group (16 different n, same chunk) spread over the banks.
__device__ __forceinline__ int ws_off(int o, int n, int k) {
    return (o * OS_CT + n) * OS_KC + ((((k >> 2) ^ (n & 7)) << 2) | (k & 3));
}

// ------------------------------------------------------------------------------------------------
// bstart[o][b] = first rule of offset o whose output row is >= b*R   (b = 0..nb)
// ------------------------------------------------------------------------------------------------
__global__ void k_block_starts(const int* __restrict__ out_rows, const long long* __restrict__ prefix, int n_off,
                               long long R, long long nb, int* __restrict__ bstart) {
    const long long total = (long long)n_off * (nb + 1);
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        const int o = (int)(t / (nb + 1));
        const long long b = t - (long long)o * (nb + 1);
        long long lo = prefix[o], hi = prefix[o + 1];
        const long long target = b * R;
        while (lo < hi) {
            long long mid = (lo + hi) >> 1;
            if (out_rows[mid] < target) lo = mid + 1;
            else hi = mid;
        }
        bstart[t] = (int)lo;
    }
}

extern "C" int scn_rules_block_starts(const int32_t* out_rows, const int64_t* prefix, int n_off, int64_t n_out,
                                      int block_rows, int32_t* bstart, scn_stream_t stream) {
    SCN_REQUIRE(n_off >= 1 && n_off <= 32 && n_out >= 0 && block_rows >= 16 && block_rows <= OS_RMAX && prefix && bstart);
    const int64_t nb = cdiv(n_out, block_rows);
    hipLaunchKernelGGL(k_block_starts, dim3(scn::ew_grid((int64_t)n_off * (nb + 1), 256)), dim3(256), 0, S(stream),
                       out_rows, (const long long*)prefix, n_off, (long long)block_rows, (long long)nb, bstart);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

// ------------------------------------------------------------------------------------------------
// the kernel
// ------------------------------------------------------------------------------------------------
// Workgroup = 4 waves, owns output rows [b*R, b*R+R) x columns [n0, n0+32).  LDS: Yacc [R][32] + two 4-KB weight
// buffers (2 workgroups per CU at R = 512).  The offsets are walked in LOCKSTEP (one barrier per offset): inside an
// offset every output row occurs at most once, so the tile results are added into Yacc with plain read-modify-write --
// no LDS float atomics (measured: ds_add_f32 costs ~200 cycles per wave-instruction and dominated the first version).
// The weight slice of offset o+1 is fetched into registers while offset o computes and is written to the other buffer
// before the barrier.  Across offsets the accumulation order is fixed (o ascending), so results are reproducible.
template <bool WT, bool VEC, bool VECN>
__global__ __launch_bounds__(OS_THREADS) void k_conv_os(
    const float* __restrict__ X, int cin, const int* __restrict__ in_rows, const int* __restrict__ out_rows,
    const int* __restrict__ bstart, int n_off, long long n_out, int R, const float* __restrict__ W,
    const float* __restrict__ bias, const float* __restrict__ residual, const float* __restrict__ relu_mask,
    float* __restrict__ Y, int cout, int flags) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wb = smem;                                          // [2][32 n][32 k]
    float* Yacc = smem + 2 * OS_CT * OS_KC;                    // [R][32]
    int* seg = (int*)(Yacc + R * OS_CT);                       // seg[o] = first rule, seg[32+o] = end

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const long long nb1 = (n_out + R - 1) / R + 1;
    const long long b = blockIdx.x;
    const long long r0 = b * R;
    const int rows = (int)((n_out - r0) < R ? (n_out - r0) : R);
    const int n0 = blockIdx.y * OS_CT;
    const bool relu_in = flags & SCN_F_RELU_IN;
    const bool rev = flags & SCN_F_OFF_REVERSE;

    if (tid < n_off) {
        seg[tid] = bstart[(long long)tid * nb1 + b];
        seg[32 + tid] = bstart[(long long)tid * nb1 + b + 1];
    }
    for (int e = tid; e < R * OS_CT; e += OS_THREADS) Yacc[e] = 0.f;

    // this thread's element of a weight slice: one 16-byte piece (256 threads x 4 floats = 32 x 32)
    const int wc4 = tid & 7, wm = tid >> 3;                    // WT: (n = wm, k = 4*wc4..)   !WT: (k = wm, n = 4*wc4..)
    auto load_w = [&](int o, int kc) -> float4 {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const int wo = rev ? n_off - 1 - o : o;
        if (WT) {
            const int k = kc + 4 * wc4, ng = n0 + wm;
            if (ng < cout) {
                const float* src = W + ((long long)wo * cout + ng) * cin + k;
                if (VEC && k + 3 < cin) v = *(const float4*)src;
                else {
                    if (k < cin) v.x = src[0];
                    if (k + 1 < cin) v.y = src[1];
                    if (k + 2 < cin) v.z = src[2];
                    if (k + 3 < cin) v.w = src[3];
                }
            }
        } else {
            const int kg = kc + wm, ng = n0 + 4 * wc4;
            if (kg < cin) {
                const float* src = W + ((long long)wo * cin + kg) * cout + ng;
                if (VECN && ng + 3 < cout) v = *(const float4*)src;
                else {
                    if (ng < cout) v.x = src[0];
                    if (ng + 1 < cout) v.y = src[1];
                    if (ng + 2 < cout) v.z = src[2];
                    if (ng + 3 < cout) v.w = src[3];
                }
            }
        }
        return v;
    };
    auto store_w = [&](float* buf, float4 v) {
        if (WT) {
            *(float4*)(buf + ws_off(0, wm, 4 * wc4)) = v;
        } else {                                               // transpose into [n][k]
            buf[ws_off(0, 4 * wc4 + 0, wm)] = v.x;
            buf[ws_off(0, 4 * wc4 + 1, wm)] = v.y;
            buf[ws_off(0, 4 * wc4 + 2, wm)] = v.z;
            buf[ws_off(0, 4 * wc4 + 3, wm)] = v.w;
        }
    };

    // ---- per-wave tile stream ------------------------------------------------------------------------------
    // The tiles this wave computes form a static sequence (offset ascending, then tiles wave, wave+4, ... of the
    // offset).  Three stages run ahead of each other so the gather latency hides behind MFMAs:
    //   stage I (2 tiles ahead): rule indices;  stage G (1 ahead): the 16 gathered A rows;  stage C: MFMA + accumulate
    struct Cur { int o, p; };                       // tile position: offset, first rule (+ lane's i)
    auto first_tile = [&](Cur& c) -> bool {         // wave-uniform
        c.o = 0;
        c.p = seg[0] + wave * OS_T;
        while (c.p >= seg[32 + c.o]) {
            if (++c.o >= n_off) return false;
            c.p = seg[c.o] + wave * OS_T;
        }
        return true;
    };
    auto next_tile = [&](Cur& c) -> bool {
        c.p += 4 * OS_T;
        while (c.p >= seg[32 + c.o]) {
            if (++c.o >= n_off) return false;
            c.p = seg[c.o] + wave * OS_T;
        }
        return true;
    };
    auto load_idx = [&](const Cur& c, bool live, int& inr, int& outl) {
        const int p = c.p + i;
        const bool valid = live && p < seg[32 + c.o];
        inr = valid ? in_rows[p] : -1;
        outl = valid ? (int)(out_rows[p] - r0) : -1;
    };
    auto gather = [&](int inr, int kc, float4& a0, float4& a1) {
        a0 = make_float4(0.f, 0.f, 0.f, 0.f);
        a1 = a0;
        if (inr >= 0) {
            const float* xp = X + (long long)inr * cin + kc + 4 * kq;
            if (VEC) {
                if (kc + 4 * kq + 3 < cin) a0 = *(const float4*)xp;
                if (kc + 16 + 4 * kq + 3 < cin) a1 = *(const float4*)(xp + 16);
            } else {
                const int k = kc + 4 * kq;
                if (k < cin) a0.x = xp[0];
                if (k + 1 < cin) a0.y = xp[1];
                if (k + 2 < cin) a0.z = xp[2];
                if (k + 3 < cin) a0.w = xp[3];
                if (k + 16 < cin) a1.x = xp[16];
                if (k + 17 < cin) a1.y = xp[17];
                if (k + 18 < cin) a1.z = xp[18];
                if (k + 19 < cin) a1.w = xp[19];
            }
        }
    };

    // weight slices are numbered s = (kc/32)*n_off + o; slice s+3 is requested while slice s computes (three global
    // round trips in flight: a phase is far shorter than one HBM/L2 latency), slice s+1 is written to the other LDS
    // buffer at the end of phase s.
    const int n_slices = ((cin + OS_KC - 1) / OS_KC) * n_off;
    auto load_slice = [&](int sidx) -> float4 {
        if (sidx >= n_slices) return make_float4(0.f, 0.f, 0.f, 0.f);
        return load_w(sidx % n_off, (sidx / n_off) * OS_KC);
    };
    int phase = 0;
    store_w(Wb, load_slice(0));
    float4 w1 = load_slice(1), w2 = load_slice(2);
    __syncthreads();

    for (int kc = 0; kc < cin; kc += OS_KC) {
        // prime the pipeline for this K-chunk
        Cur tc, tg, ti;                               // compute / gather / index positions
        bool live_c = first_tile(tc);
        tg = tc;
        bool live_g = live_c && next_tile(tg);
        ti = tg;
        bool live_i = live_g && next_tile(ti);
        int inr_c, outl_c, inr_g, outl_g, inr_i, outl_i;
        load_idx(tc, live_c, inr_c, outl_c);
        load_idx(tg, live_g, inr_g, outl_g);
        load_idx(ti, live_i, inr_i, outl_i);
        float4 ac0, ac1, ag0, ag1;
        gather(inr_c, kc, ac0, ac1);
        gather(inr_g, kc, ag0, ag1);

        for (int o = 0; o < n_off; ++o, ++phase) {
            const float* Ws = Wb + (phase & 1) * (OS_CT * OS_KC);
            const float4 w3 = load_slice(phase + 3);

            if (live_c && tc.o == o) {
                // B fragments of this offset: column i (+16 for the second half-tile), channels 4kq..4kq+3 (+16)
                const float4 b00 = *(const float4*)(Ws + ws_off(0, i, 4 * kq));
                const float4 b01 = *(const float4*)(Ws + ws_off(0, i, 16 + 4 * kq));
                const float4 b10 = *(const float4*)(Ws + ws_off(0, 16 + i, 4 * kq));
                const float4 b11 = *(const float4*)(Ws + ws_off(0, 16 + i, 16 + 4 * kq));
                do {
                    // stage I for the tile after next, stage G for the next tile: loads in flight during the MFMAs
                    Cur tn = ti;
                    const bool live_n = live_i && next_tile(tn);
                    int inr_n, outl_n;
                    load_idx(tn, live_n, inr_n, outl_n);
                    float4 an0, an1;
                    gather(inr_i, kc, an0, an1);

                    float4 a0 = ac0, a1 = ac1;
                    if (relu_in) {
                        a0.x = fmaxf(a0.x, 0.f); a0.y = fmaxf(a0.y, 0.f); a0.z = fmaxf(a0.z, 0.f); a0.w = fmaxf(a0.w, 0.f);
                        a1.x = fmaxf(a1.x, 0.f); a1.y = fmaxf(a1.y, 0.f); a1.z = fmaxf(a1.z, 0.f); a1.w = fmaxf(a1.w, 0.f);
                    }
                    f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
                    c0 = MFMA16(a0.x, b00.x, c0);  c1 = MFMA16(a0.x, b10.x, c1);
                    c0 = MFMA16(a0.y, b00.y, c0);  c1 = MFMA16(a0.y, b10.y, c1);
                    c0 = MFMA16(a0.z, b00.z, c0);  c1 = MFMA16(a0.z, b10.z, c1);
                    c0 = MFMA16(a0.w, b00.w, c0);  c1 = MFMA16(a0.w, b10.w, c1);
                    c0 = MFMA16(a1.x, b01.x, c0);  c1 = MFMA16(a1.x, b11.x, c1);
                    c0 = MFMA16(a1.y, b01.y, c0);  c1 = MFMA16(a1.y, b11.y, c1);
                    c0 = MFMA16(a1.z, b01.z, c0);  c1 = MFMA16(a1.z, b11.z, c1);
                    c0 = MFMA16(a1.w, b01.w, c0);  c1 = MFMA16(a1.w, b11.w, c1);
                    // scatter-accumulate: rows of one offset are distinct -> plain read-modify-write
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int r = __shfl(outl_c, 4 * kq + j);
                        if (r >= 0) {
                            float* y = Yacc + r * OS_CT + i;
                            y[0] += c0[j];
                            y[16] += c1[j];
                        }
                    }
                    // rotate: G -> C, I -> G, N -> I
                    tc = tg; live_c = live_g; outl_c = outl_g; ac0 = ag0; ac1 = ag1;
                    tg = ti; live_g = live_i; outl_g = outl_i; ag0 = an0; ag1 = an1;
                    ti = tn; live_i = live_n; inr_i = inr_n; outl_i = outl_n;
                } while (live_c && tc.o == o);
            }
            if (phase + 1 < n_slices) store_w(Wb + ((phase + 1) & 1) * (OS_CT * OS_KC), w1);
            w1 = w2;
            w2 = w3;
            __syncthreads();
        }
    }

    // ---- epilogue: coalesced row stores -----------------------------------------------------------------------
    for (int e = tid; e < rows * OS_CT; e += OS_THREADS) {
        const int r = e >> 5, n = e & 31;
        const int ng = n0 + n;
        if (ng < cout) {
            const long long off = (r0 + r) * cout + ng;
            float y = Yacc[e];
            if (bias) y += bias[ng];
            if (residual) y += residual[off];
            if (relu_mask && !(relu_mask[off] > 0.f)) y = 0.f;
            Y[off] = y;
        }
    }
}

extern "C" int64_t scn_conv_os_lds_bytes(int n_off, int block_rows) {
    (void)n_off;
    return (int64_t)sizeof(float) * (2 * OS_CT * OS_KC + (int64_t)block_rows * OS_CT) + 4 * 64;
}

extern "C" int scn_conv_rules(const float* X, int cin, const int32_t* in_rows, const int32_t* out_rows,
                              const int32_t* bstart, int n_off, int64_t n_out, int block_rows, const float* W,
                              const float* bias, const float* residual, const float* relu_mask, float* Y, int cout,
                              int flags, scn_stream_t stream) {
    SCN_REQUIRE(n_off >= 1 && n_off <= 32 && n_out >= 0 && cin >= 1 && cout >= 1);
    SCN_REQUIRE(block_rows >= 16 && block_rows <= OS_RMAX);
    if (n_out == 0) return SCN_OK;
    SCN_REQUIRE(X && in_rows && out_rows && bstart && W && Y);
    const int64_t lds = scn_conv_os_lds_bytes(n_off, block_rows);
    SCN_REQUIRE(lds <= 160 * 1024);
    const bool wt = flags & SCN_F_W_TRANSPOSED;
    const bool vec = (cin % 4 == 0) && (((uintptr_t)X & 15) == 0) && (((uintptr_t)W & 15) == 0);
    const bool vecn = (cout % 4 == 0) && (((uintptr_t)W & 15) == 0);
    dim3 grid((unsigned)cdiv(n_out, block_rows), (unsigned)cdiv(cout, OS_CT));
#define LAUNCH_OS(T, V, VN)                                                                                       \
    do {                                                                                                          \
        SCN_HIP(hipFuncSetAttribute((const void*)k_conv_os<T, V, VN>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    160 * 1024));                                                                 \
        hipLaunchKernelGGL((k_conv_os<T, V, VN>), grid, dim3(OS_THREADS), (size_t)lds, S(stream), X, cin, in_rows, \
                           out_rows, bstart, n_off, (long long)n_out, block_rows, W, bias, residual, relu_mask, Y, \
                           cout, flags);                                                                          \
    } while (0)
    if (wt && vec) LAUNCH_OS(true, true, true);
    else if (wt) LAUNCH_OS(true, false, true);
    else if (vec && vecn) LAUNCH_OS(false, true, true);
    else if (vec) LAUNCH_OS(false, true, false);
    else if (vecn) LAUNCH_OS(false, false, true);
    else LAUNCH_OS(false, false, false);
#undef LAUNCH_OS
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}
