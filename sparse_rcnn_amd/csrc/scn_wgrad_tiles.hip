// Round 6, experiment (c) of VERDICT r5 item 1: the weight gradient of a SubM 3^3 layer with ONE row gather per rule.
//
//   dW[o] = sum_r in(X[table[o][r]])^T . dY[r]                                    (Cin = Cout = 32, fp32, 27 offsets)
//
// The product kernel (k_wgrad_direct, scn_wgrad.hip) is rule-major: per rule it gathers the X row AND the dY row.  This form
// is TILE-major, as the judge proposed: a workgroup walks its share of the mask-sorted tiles (the forward kernel's tables:
// tstab / tile_mask / perm); the dY rows of a tile (16 x 128 B) are loaded ONCE into LDS and serve every offset of the tile;
// per (tile, offset) only the 16 X rows are gathered.  dW[o] (32 x 32 fp32 = 16 registers per lane) cannot live in one wave for
// all 27 offsets, so the OFFSETS are dealt to the 16 waves of the workgroup -- at most two accumulator sets per wave, dealt on
// the host by rule count (longest first; the centre offset, which every row has, is split by tile parity into two virtual
// offsets) -- and every wave visits the tiles whose mask holds one of its offsets.  The waves share the dY tiles, so they walk the
// tiles in ROUNDS of 8 (double-buffered in LDS, one barrier per round).  Per workgroup the 28 partial blocks go to a slab and
// k_wgrad_tiles32_sum adds the workgroups' slabs in fixed order (deterministic).
// MFMA (v_mfma_f32_16x16x4_f32): A[m][k] = X[row k][channel 2m + ta], B[k][n] = dY[row k][column 2n + tb]: lane (c, rq) holds
// row 4 rq + e of MFMA e and two consecutive channels (columns) of it -- one 8-byte load (LDS read) per row piece.
// Measured against the product kernel in profiles/r6_wgrad_one_gather.txt; opt-in through the C ABI only (scn_wgrad_tiles32),
// nothing in the package calls it.
#include "scn_common.h"

using scn::S;
using scn::cdiv;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

namespace {
constexpr int WT_NW = 16;            // waves per workgroup
#ifndef WT_R_V
#define WT_R_V 8
#endif
constexpr int WT_R = WT_R_V;         // tiles per round (8; -DWT_R_V=16: the sweep of profiles/r6_wgrad_one_gather.txt)
constexpr int WT_LU = WT_R / 8;      // dY pieces per thread and round
constexpr int WT_NV = 28;            // virtual offsets: 27 + the second half of the centre offset

struct WtPlan {
    signed char own[WT_NW][2];       // virtual offsets of wave w's two accumulator sets (-1: none)
};

// virtual offset v -> (real offset, tile parity it takes or -1 for all tiles)
__device__ __forceinline__ int wt_real(int v) { return v == 27 ? 13 : v; }

__global__ __launch_bounds__(WT_NW * 64) void k_wgrad_tiles32(
    const float* __restrict__ X, long long n_in, const float* __restrict__ dY, long long n_out, const int* __restrict__ tstab,
    const unsigned* __restrict__ tile_mask, const int* __restrict__ perm, long long nt, WtPlan plan, float* __restrict__ slabs,
    int relu_in) {
    __shared__ __attribute__((aligned(16))) float dys[2][WT_R][16][32];      // 2 x 16 KB
    __shared__ unsigned masks[2][WT_R];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, rq = lane >> 4;
    const int b = blockIdx.x, n_wg = gridDim.x;
    const long long my_tiles = (nt - b + n_wg - 1) / n_wg;                    // tiles b, b + n_wg, ...
    const int n_rounds = (int)((my_tiles + WT_R - 1) / WT_R);
    const int v0 = plan.own[w][0], v1 = plan.own[w][1];
    const int o0 = v0 >= 0 ? wt_real(v0) : 0, o1 = v1 >= 0 ? wt_real(v1) : 0;
    // the centre's two halves: virtual 13 takes even tiles of the round, virtual 27 the odd ones
    const unsigned par0 = v0 == 13 ? 0x5555u : (v0 == 27 ? 0xAAAAu : 0xFFFFu), par1 = v1 == 13 ? 0x5555u : (v1 == 27 ? 0xAAAAu : 0xFFFFu);
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)(unsigned)(n_in * 128), 0x00020000);
    const int relu_lo = relu_in ? 0 : (int)0x80000000;

    f32x4 acc0[2][2], acc1[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int q = 0; q < 2; ++q) { acc0[a][q] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc1[a][q] = acc0[a][q]; }

    // loader mapping of the dY tiles of a round: thread -> (tile j, row, 16-byte piece)
    const int lj = tid >> 7, lrow = (tid >> 3) & 15, lpc = tid & 7;
    auto tile_of = [&](int round, int j) -> long long {
        const long long k = (long long)round * WT_R + j;
        return k < my_tiles ? b + k * (long long)n_wg : -1;
    };
    auto load_round = [&](int round, f32x4 (&v)[WT_LU], unsigned (&m)[WT_LU]) {
#pragma unroll
        for (int u = 0; u < WT_LU; ++u) {
            const long long t = tile_of(round, lj + 8 * u);
            v[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
            m[u] = 0;
            if (t >= 0) {
                const int orow = perm[t * 16 + lrow];
                if (orow >= 0) v[u] = *(const f32x4*)(dY + (long long)orow * 32 + 4 * lpc);
                m[u] = tile_mask[t];
            }
        }
    };
    auto store_round = [&](int buf, const f32x4 (&v)[WT_LU], const unsigned (&m)[WT_LU]) {
#pragma unroll
        for (int u = 0; u < WT_LU; ++u) {
            *(f32x4*)&dys[buf][lj + 8 * u][lrow][4 * lpc] = v[u];
            if ((tid & 127) == 0) masks[buf][lj + 8 * u] = m[u];
        }
    };
    {
        f32x4 v[WT_LU]; unsigned m[WT_LU];
        load_round(0, v, m);
        store_round(0, v, m);
    }
    __syncthreads();

    for (int round = 0; round < n_rounds; ++round) {
        const int buf = round & 1;
        f32x4 nv[WT_LU]; unsigned nm[WT_LU];
        load_round(round + 1, nv, nm);                                 // the next round's dY rows: in flight during this round
        // items of this wave in this round: bit (2 j + s) = tile j holds the offset of accumulator set s
        unsigned long long todo = 0;
#pragma unroll
        for (int j = 0; j < WT_R; ++j) {
            const unsigned m = masks[buf][j];
            if (v0 >= 0 && ((m >> o0) & 1u) && ((par0 >> j) & 1u)) todo |= 1ull << (2 * j);
            if (v1 >= 0 && ((m >> o1) & 1u) && ((par1 >> j) & 1u)) todo |= 2ull << (2 * j);
        }
        todo = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(todo >> 32)) << 32) |
               (unsigned)__builtin_amdgcn_readfirstlane((unsigned)todo);
        // software pipeline over the items: row indices two items ahead, row gathers one item ahead, MFMAs on the current one
        int it_cur = -1, it_nxt = -1, it_idx = -1;
        int idx_nxt[4] = {-1, -1, -1, -1}, idx_far[4] = {-1, -1, -1, -1};
        f32x2 a_cur[4], a_nxt[4];
        auto pop = [&]() -> int {
            if (!todo) return -1;
            const int k = __builtin_ctzll(todo);
            todo &= todo - 1;
            return k;
        };
        auto load_idx = [&](int item, int (&idx)[4]) {                  // UNCONDITIONAL load (a load under a branch makes hipcc
            const bool ok = item >= 0;                                    // wait for every outstanding load at the merge); no
            const long long t = ok ? tile_of(round, item >> 1) : b;       // item: the workgroup's first tile, result discarded
            const int o = (item & 1) ? o1 : o0;                           // (-1 rows -> the gathers return zeros)
            const int4 q = *(const int4*)(tstab + (t * 27 + o) * 16 + 4 * rq);
            idx[0] = ok ? q.x : -1; idx[1] = ok ? q.y : -1; idx[2] = ok ? q.z : -1; idx[3] = ok ? q.w : -1;
        };
        auto gather = [&](const int (&idx)[4], f32x2 (&a)[4]) {
#pragma unroll
            for (int e = 0; e < 4; ++e)                                 // -1 (no rule) -> out of range -> zeros
                a[e] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xrsrc, __mul24(idx[e], 128) + c * 8, 0, 0));
        };
        it_cur = pop();
        load_idx(it_cur, idx_nxt);
        it_nxt = pop();
        load_idx(it_nxt, idx_far);
        gather(idx_nxt, a_cur);                                         // rows of the first item
#pragma unroll
        for (int e = 0; e < 4; ++e) idx_nxt[e] = idx_far[e];
        it_idx = pop();
        load_idx(it_idx, idx_far);
        while (it_cur >= 0) {
            gather(idx_nxt, a_nxt);                                     // rows of the next item (zeros when there is none)
            f32x2 bb[4];
            const int j = it_cur >> 1;
#pragma unroll
            for (int e = 0; e < 4; ++e) bb[e] = *(const f32x2*)&dys[buf][j][4 * rq + e][2 * c];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f32x2 x = a_cur[e];
                x[0] = __int_as_float(max(__float_as_int(x[0]), relu_lo));
                x[1] = __int_as_float(max(__float_as_int(x[1]), relu_lo));
                if (it_cur & 1) {
                    acc1[0][0] = MFMA16(x[0], bb[e][0], acc1[0][0]);
                    acc1[0][1] = MFMA16(x[0], bb[e][1], acc1[0][1]);
                    acc1[1][0] = MFMA16(x[1], bb[e][0], acc1[1][0]);
                    acc1[1][1] = MFMA16(x[1], bb[e][1], acc1[1][1]);
                } else {
                    acc0[0][0] = MFMA16(x[0], bb[e][0], acc0[0][0]);
                    acc0[0][1] = MFMA16(x[0], bb[e][1], acc0[0][1]);
                    acc0[1][0] = MFMA16(x[1], bb[e][0], acc0[1][0]);
                    acc0[1][1] = MFMA16(x[1], bb[e][1], acc0[1][1]);
                }
            }
            it_cur = it_nxt;
            it_nxt = it_idx;
#pragma unroll
            for (int e = 0; e < 4; ++e) { a_cur[e] = a_nxt[e]; idx_nxt[e] = idx_far[e]; }
            it_idx = pop();
            load_idx(it_idx, idx_far);
        }
        store_round(buf ^ 1, nv, nm);                                   // (nobody reads buf ^ 1 in this round)
        __syncthreads();
    }
    // partial blocks of this workgroup: slab[b][v][ci][co], ci = 2 (4 rq + j) + ta, co = 2 c + tb
    float* out = slabs + (long long)b * WT_NV * 1024;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int v = s ? v1 : v0;
        if (v < 0) continue;
#pragma unroll
        for (int ta = 0; ta < 2; ++ta)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ci = 2 * (4 * rq + j) + ta;
                const f32x2 pr = {s ? acc1[ta][0][j] : acc0[ta][0][j], s ? acc1[ta][1][j] : acc0[ta][1][j]};
                *(f32x2*)(out + (v * 32 + ci) * 32 + 2 * c) = pr;
            }
    }
}

// dW[o] = sum over workgroups b of slab[b][o] (+ slab[b][27] for the centre offset), in a FIXED association: a workgroup owns 64
// float4 outputs; thread (q, k) adds the slabs b = q, q + 4, q + 8, ... for output k; the four partial sums meet in LDS and are
// added in ascending q.
__global__ __launch_bounds__(256) void k_wgrad_tiles32_sum(const float* __restrict__ slabs, int n_wg, float* __restrict__ dW) {
    __shared__ f32x4 part[4][64];
    const int k = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + k;                                   // float4 index into [27][32][32]
    const int o = e >> 8;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int b = q; b < n_wg; b += 4) {
        const float* p = slabs + (long long)b * WT_NV * 1024;
        s += *(const f32x4*)(p + e * 4);
        if (o == 13) s += *(const f32x4*)(p + 27 * 1024 + (e & 255) * 4);
    }
    part[q][k] = s;
    __syncthreads();
    if (q == 0) *(f32x4*)(dW + e * 4) = ((part[0][k] + part[1][k]) + part[2][k]) + part[3][k];
}
}  // namespace

extern "C" int64_t scn_wgrad_tiles32_scratch_bytes(void) { return (int64_t)scn::cu_budget() * WT_NV * 1024 * 4; }

// X [n_in][32], dY [n_out][32] fp32; tstab / tile_mask / perm: the mask-sorted tiles of the layer's 27-offset table
// (scn_tiles_build over n_out rows); prefix_host[28]: the rule prefix (weights of the offset -> wave deal); dW [27][32][32].
extern "C" int scn_wgrad_tiles32(const float* X, int64_t n_in, const float* dY, int64_t n_out, const int32_t* tstab,
                                 const uint32_t* tile_mask, const int32_t* perm, const int64_t* prefix_host, float* dW,
                                 void* scratch, int relu_in, scn_stream_t stream) {
    SCN_REQUIRE(n_in >= 0 && n_out >= 0 && dW && scratch && prefix_host);
    hipStream_t st = S(stream);
    if (n_out == 0) {
        SCN_HIP(hipMemsetAsync(dW, 0, sizeof(float) * 27 * 1024, st));
        return SCN_OK;
    }
    SCN_REQUIRE(X && dY && tstab && tile_mask && perm);
    SCN_REQUIRE(n_in < (1ll << 23) && (((uintptr_t)X | (uintptr_t)dY) & 15) == 0);
    const int64_t nt = cdiv(n_out, 16);
    // deal the 28 virtual offsets to 16 waves x 2 accumulator sets: heaviest first onto the lightest wave with a free set
    double wgt[WT_NV];
    for (int o = 0; o < 27; ++o) wgt[o] = (double)(prefix_host[o + 1] - prefix_host[o]);
    wgt[27] = wgt[13] * 0.5;
    wgt[13] *= 0.5;
    int order[WT_NV];
    for (int v = 0; v < WT_NV; ++v) order[v] = v;
    for (int i = 0; i < WT_NV; ++i)
        for (int j = i + 1; j < WT_NV; ++j)
            if (wgt[order[j]] > wgt[order[i]]) { const int t = order[i]; order[i] = order[j]; order[j] = t; }
    WtPlan plan;
    double load[WT_NW];
    int used[WT_NW];
    for (int w = 0; w < WT_NW; ++w) { plan.own[w][0] = plan.own[w][1] = -1; load[w] = 0.0; used[w] = 0; }
    for (int i = 0; i < WT_NV; ++i) {
        int best = -1;
        for (int w = 0; w < WT_NW; ++w)
            if (used[w] < 2 && (best < 0 || load[w] < load[best])) best = w;
        SCN_REQUIRE(best >= 0);
        plan.own[best][used[best]++] = (signed char)order[i];
        load[best] += wgt[order[i]];
    }
    int n_wg = scn::cu_budget();
    if (n_wg > cdiv(nt, WT_R)) n_wg = (int)cdiv(nt, WT_R);
    if (n_wg < 1) n_wg = 1;
    float* slabs = (float*)scratch;
    // (every one of the 28 virtual offsets has an owner wave in every workgroup: every slab block is written)
    hipLaunchKernelGGL(k_wgrad_tiles32, dim3(n_wg), dim3(WT_NW * 64), 0, st, X, (long long)n_in, dY, (long long)n_out, tstab,
                       tile_mask, perm, (long long)nt, plan, slabs, relu_in);
    SCN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_wgrad_tiles32_sum, dim3(27 * 256 / 64), dim3(256), 0, st, (const float*)slabs, n_wg, dW);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}
