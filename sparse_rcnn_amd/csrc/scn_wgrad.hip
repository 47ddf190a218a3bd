// wgrad: weight gradient of every conv-type layer,   dW[o] = sum_{p in R_o} in(X[in_p])^T . dY[out_p]
//
// A tall-skinny reduction: M x N = Cin x Cout is small, K = number of rules is long and GATHERED on both operands.
// At 32-64 channels it is bound by the row gathers (8-16 FLOP per gathered byte), from 128 channels on by the fp32
// matrix cores.
//
// Workgroup (4 waves) = (offset o, K-split s, CB x CB block of dW[o]).  It walks its rule range in chunks of 32 rules:
//   * the 32 X rows (CB channels) and 32 dY rows (CB channels) of the NEXT chunk are gathered with 16-byte loads into
//     registers while the current chunk is multiplied -- all loads are unconditional (out-of-range rules read row 0 and
//     are zeroed before the LDS write) so they stay in flight across the MFMAs;
//   * they are written to the other LDS buffer ([rule][channel], row stride 80 floats: the 4 rules x 16 channels of an
//     MFMA fragment hit 64 distinct banks), one barrier per chunk;
//   * MFMA v_mfma_f32_16x16x4_f32: A[i = channel of X][k = rule], B[k = rule][j = channel of dY]; a wave owns a
//     (CB/2) x (CB/2) sub-block (2 x 2 tiles at CB = 64) so each fragment read feeds two MFMAs.
// Every workgroup writes its CB x CB partial to a slab; a second kernel sums the K-splits in fixed order
// (bitwise reproducible, no float atomics).
#include "scn_common.h"

using scn::S;
using scn::cdiv;

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

static constexpr int WG_PK = 32;        // rules per chunk
static constexpr int WG_LD = 80;        // LDS row stride in floats (64 channels + 16 pad)

struct WgradPlan {
    long long rule_start[33];           // prefix of rules per offset
    int n_off, splits, cb, nbi, nbj;    // K-splits per offset, block size, blocks along Cin / Cout
};

template <int CB>
__global__ __launch_bounds__(256) void k_wgrad_lds(const float* __restrict__ X, int cin, const float* __restrict__ dY,
                                                   int cout, const int* __restrict__ in_rows,
                                                   const int* __restrict__ out_rows, WgradPlan plan,
                                                   float* __restrict__ slabs, int relu_in) {
    constexpr int TW = CB / 32;                      // tiles per wave along each block dimension (1 or 2)
    constexpr int F4_PER_ROW = CB / 4;               // 16-byte pieces per staged row
    constexpr int ROWS_PER_PASS = 256 / F4_PER_ROW;  // rows staged per pass of the 256 threads (16 or 32)
    constexpr int PASSES = WG_PK / ROWS_PER_PASS;    // 2 or 1
    __shared__ __attribute__((aligned(16))) float lds[2][2][WG_PK * WG_LD];   // [buffer][X | dY][rule][channel]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, kq = lane >> 4;
    const int o = blockIdx.y, s = blockIdx.x;
    const int bi = blockIdx.z / plan.nbj, bj = blockIdx.z % plan.nbj;
    const int ci0 = bi * CB, co0 = bj * CB;
    const int wi = wave >> 1, wj = wave & 1;         // wave's sub-block inside the CB x CB block

    const long long p_lo = plan.rule_start[o], p_hi = plan.rule_start[o + 1];
    const long long per = ((p_hi - p_lo + plan.splits - 1) / plan.splits + WG_PK - 1) / WG_PK * WG_PK;
    long long p0 = p_lo + (long long)s * per;
    long long p1 = p0 + per < p_hi ? p0 + per : p_hi;

    f32x4 acc[TW][TW];
#pragma unroll
    for (int a = 0; a < TW; ++a)
#pragma unroll
        for (int b = 0; b < TW; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // staging role of this thread: row r_st (+ ROWS_PER_PASS per pass), 16-byte piece c4
    const int c4 = tid % F4_PER_ROW, r_st = tid / F4_PER_ROW;
    const bool x_ok = ci0 + 4 * c4 + 3 < cin, y_ok = co0 + 4 * c4 + 3 < cout;     // channel counts are multiples of 4

    auto load_idx = [&](long long pc, int (&ri)[PASSES], int (&ro)[PASSES]) {
#pragma unroll
        for (int q = 0; q < PASSES; ++q) {
            const long long p = pc + r_st + q * ROWS_PER_PASS;
            const long long pl = p < p1 ? p : (p1 > p0 ? p1 - 1 : p0);          // clamp: load stays unconditional
            const int vi = in_rows ? in_rows[pl < p_hi ? pl : 0] : (int)pl;
            const int vo = out_rows ? out_rows[pl < p_hi ? pl : 0] : (int)pl;
            ri[q] = p < p1 ? vi : -1;
            ro[q] = p < p1 ? vo : -1;
        }
    };
    auto load_rows = [&](const int (&ri)[PASSES], const int (&ro)[PASSES], float4 (&vx)[PASSES], float4 (&vy)[PASSES]) {
#pragma unroll
        for (int q = 0; q < PASSES; ++q) {
            const float* xp = X + (long long)(ri[q] < 0 ? 0 : ri[q]) * cin + ci0 + 4 * c4;
            const float* yp = dY + (long long)(ro[q] < 0 ? 0 : ro[q]) * cout + co0 + 4 * c4;
            vx[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            vy[q] = vx[q];
            if (x_ok) vx[q] = *(const float4*)xp;                 // loop-invariant per thread
            if (y_ok) vy[q] = *(const float4*)yp;
        }
    };
    auto store_rows = [&](int buf, const int (&ri)[PASSES], const int (&ro)[PASSES], float4 (&vx)[PASSES],
                          float4 (&vy)[PASSES]) {
#pragma unroll
        for (int q = 0; q < PASSES; ++q) {
            float4 a = vx[q], b = vy[q];
            if (ri[q] < 0) a = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ro[q] < 0) b = make_float4(0.f, 0.f, 0.f, 0.f);
            if (relu_in) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
            const int r = r_st + q * ROWS_PER_PASS;
            *(float4*)(&lds[buf][0][r * WG_LD + 4 * c4]) = a;
            *(float4*)(&lds[buf][1][r * WG_LD + 4 * c4]) = b;
        }
    };

    if (p0 < p1) {
        int ri[PASSES], ro[PASSES], rin[PASSES], ron[PASSES];
        float4 vx[PASSES], vy[PASSES];
        load_idx(p0, ri, ro);
        load_rows(ri, ro, vx, vy);
        load_idx(p0 + WG_PK, rin, ron);
        store_rows(0, ri, ro, vx, vy);
        __syncthreads();
        int buf = 0;
        for (long long pc = p0; pc < p1; pc += WG_PK, buf ^= 1) {
            const bool more = pc + WG_PK < p1;
            // next chunk's rows -> registers (indices were fetched one chunk earlier), indices of the chunk after
            int ri2[PASSES], ro2[PASSES];
            if (more) {
                load_rows(rin, ron, vx, vy);
                load_idx(pc + 2 * WG_PK, ri2, ro2);
            }
            // multiply the current chunk: 8 steps of 4 rules
            const float* xs = &lds[buf][0][0];
            const float* ys = &lds[buf][1][0];
#pragma unroll
            for (int st = 0; st < WG_PK / 4; ++st) {
                float a[TW], b[TW];
#pragma unroll
                for (int t = 0; t < TW; ++t) {
                    a[t] = xs[(4 * st + kq) * WG_LD + (wi * TW + t) * 16 + i];
                    b[t] = ys[(4 * st + kq) * WG_LD + (wj * TW + t) * 16 + i];
                }
#pragma unroll
                for (int ta = 0; ta < TW; ++ta)
#pragma unroll
                    for (int tb = 0; tb < TW; ++tb) acc[ta][tb] = MFMA16(a[ta], b[tb], acc[ta][tb]);
            }
            if (more) {
                store_rows(buf ^ 1, rin, ron, vx, vy);
#pragma unroll
                for (int q = 0; q < PASSES; ++q) { rin[q] = ri2[q]; ron[q] = ro2[q]; }
            }
            __syncthreads();
        }
    }

    // partial block -> slab [o][s][block][CB][CB]  (row = channel of X, col = channel of dY)
    float* slab = slabs + (((long long)o * plan.splits + s) * (plan.nbi * plan.nbj) + blockIdx.z) * (CB * CB);
#pragma unroll
    for (int ta = 0; ta < TW; ++ta)
#pragma unroll
        for (int tb = 0; tb < TW; ++tb)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = (wi * TW + ta) * 16 + 4 * kq + j;
                const int c = (wj * TW + tb) * 16 + i;
                slab[r * CB + c] = acc[ta][tb][j];
            }
}

__global__ __launch_bounds__(256) void k_wgrad_sum(const float* __restrict__ slabs, WgradPlan plan, int cin, int cout,
                                                   float* __restrict__ dW) {
    const long long total = (long long)plan.n_off * cin * cout;
    const int cb = plan.cb, nblk = plan.nbi * plan.nbj;
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const int co = (int)(e % cout);
        const int ci = (int)((e / cout) % cin);
        const int o = (int)(e / ((long long)cin * cout));
        const int blk = (ci / cb) * plan.nbj + co / cb;
        const long long base = (((long long)o * plan.splits) * nblk + blk) * (cb * cb) + (ci % cb) * cb + co % cb;
        float sum = 0.f;
        for (int s = 0; s < plan.splits; ++s) sum += slabs[base + (long long)s * nblk * cb * cb];
        dW[e] = sum;
    }
}

static int make_plan(int cin, int cout, const int64_t* prefix_host, int n_off, WgradPlan& pl) {
    pl.n_off = n_off;
    pl.cb = (cin > 32 || cout > 32) ? 64 : 32;
    pl.nbi = (int)cdiv(cin, pl.cb);
    pl.nbj = (int)cdiv(cout, pl.cb);
    int64_t maxp = 0;
    pl.rule_start[0] = prefix_host[0];
    for (int o = 0; o < n_off; ++o) {
        const int64_t cnt = prefix_host[o + 1] - prefix_host[o];
        if (cnt < 0) return SCN_EINVAL;
        if (cnt > maxp) maxp = cnt;
        pl.rule_start[o + 1] = prefix_host[o + 1];
    }
    // K-splits: aim at ~1500 workgroups, at least 4 chunks of work each
    int64_t splits = 1536 / ((int64_t)n_off * pl.nbi * pl.nbj);
    const int64_t max_useful = cdiv(maxp, 4 * WG_PK);
    if (splits > max_useful) splits = max_useful;
    if (splits < 1) splits = 1;
    pl.splits = (int)splits;
    return SCN_OK;
}

extern "C" int64_t scn_wgrad_scratch_bytes(int cin, int cout, const int64_t* prefix_host, int n_off) {
    if (!prefix_host || n_off < 1 || n_off > 32 || cin < 1 || cout < 1) return -1;
    WgradPlan pl;
    if (make_plan(cin, cout, prefix_host, n_off, pl) != SCN_OK) return -1;
    const int64_t fast = (int64_t)n_off * pl.splits * pl.nbi * pl.nbj * pl.cb * pl.cb * (int64_t)sizeof(float) + 256;
    const int64_t simple = scn::wgrad_simple_scratch_bytes(cin, cout, prefix_host, n_off);
    return fast > simple ? fast : simple;
}

extern "C" int scn_wgrad_rules(const float* X, int cin, const float* dY, int cout, const int32_t* in_rows,
                               const int32_t* out_rows, const int64_t* prefix_host, int n_off, float* dW, void* scratch,
                               int flags, scn_stream_t stream) {
    SCN_REQUIRE(prefix_host && n_off >= 1 && n_off <= 32 && cin >= 1 && cout >= 1 && dW && scratch);
    SCN_REQUIRE((in_rows == nullptr) == (out_rows == nullptr));
    SCN_REQUIRE(in_rows || n_off == 1);
    WgradPlan pl;
    SCN_REQUIRE(make_plan(cin, cout, prefix_host, n_off, pl) == SCN_OK);
    const int64_t total = prefix_host[n_off] - prefix_host[0];
    SCN_REQUIRE(total == 0 || (X && dY));
    // 16-byte row pieces need 16-byte aligned rows; otherwise fall back to a channel count the loads can take
    SCN_REQUIRE((((uintptr_t)X | (uintptr_t)dY) & 3) == 0);
    const bool aligned = (cin % 4 == 0) && (cout % 4 == 0) && ((((uintptr_t)X | (uintptr_t)dY) & 15) == 0);
    if (!aligned)
        return scn::wgrad_simple(X, cin, dY, cout, in_rows, out_rows, prefix_host, n_off, dW, scratch, flags, stream);
    dim3 grid((unsigned)pl.splits, (unsigned)n_off, (unsigned)(pl.nbi * pl.nbj));
    const int relu_in = (flags & SCN_F_RELU_IN) ? 1 : 0;
    if (pl.cb == 64)
        hipLaunchKernelGGL(k_wgrad_lds<64>, grid, dim3(256), 0, S(stream), X, cin, dY, cout, in_rows, out_rows, pl,
                           (float*)scratch, relu_in);
    else
        hipLaunchKernelGGL(k_wgrad_lds<32>, grid, dim3(256), 0, S(stream), X, cin, dY, cout, in_rows, out_rows, pl,
                           (float*)scratch, relu_in);
    SCN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_wgrad_sum, dim3(scn::ew_grid((int64_t)n_off * cin * cout, 256)), dim3(256), 0, S(stream),
                       (const float*)scratch, pl, cin, cout, dW);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}
