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
#include <stdlib.h>

#include "scn_common.h"

using scn::S;
using scn::cdiv;

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

static constexpr int WG_PK = 32;        // rules per chunk

struct WgradPlan {
    long long rule_start[33];           // prefix of rules per offset
    int unit_start[33];                 // prefix of work units per offset: a unit = `per` consecutive rules of ONE offset
    long long per;                      // rules per unit (multiple of 32): units are equal-sized, so offsets with many
                                        // rules (the centre offset has N, a corner a tenth of that) get many units
    int n_off, cb, nbi, nbj;            // block size, blocks along Cin / Cout
};

// TW = tiles per wave along each side of its sub-block (1, 2 or 4); block CB = 32*TW channels on both sides, 4 waves in
// a 2 x 2 arrangement.  LDS image of a chunk: [rule][i][t] with channel = 16*t + i stored at i*(CB/16) + t, so the TW
// fragment values a lane needs for its TW tiles (same i, consecutive t) are ONE ds_read of TW floats (b32 / b64 / b128):
// TW^2 MFMAs per 2 LDS reads -- at TW = 1..2 the kernel was bound by LDS fragment reads (128 B/clk/CU of ds_read_b32).
template <int TW>
__global__ __launch_bounds__(256) void k_wgrad_lds(const float* __restrict__ X, int cin, const float* __restrict__ dY,
                                                   int cout, const int* __restrict__ in_rows,
                                                   const int* __restrict__ out_rows, WgradPlan plan,
                                                   float* __restrict__ slabs, int relu_in,
                                                   float* __restrict__ db_slabs, unsigned db_mask, int cout_pad) {
    constexpr int CB = 32 * TW;
    constexpr int NT = CB / 16;                      // tiles per block side
    constexpr int LD = CB + 4;                       // LDS row stride in floats
    constexpr int F4_PER_ROW = CB / 4;               // 16-byte pieces per staged row
    constexpr int ROWS_PER_PASS = 256 / F4_PER_ROW;  // rows staged per pass of the 256 threads
    constexpr int PASSES = WG_PK / ROWS_PER_PASS;    // 1, 2 or 4
    typedef float fragT __attribute__((ext_vector_type(TW)));
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [buffer 2][X | dY][rule][LD]
    constexpr int OPND = WG_PK * LD;                 // floats per operand image

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, kq = lane >> 4;
    const int unit = blockIdx.x;
    int o = 0;
    while (unit >= plan.unit_start[o + 1]) ++o;      // <= 27 scalar compares
    const int s = unit - plan.unit_start[o];
    const int bi = blockIdx.z / plan.nbj, bj = blockIdx.z % plan.nbj;
    const int ci0 = bi * CB, co0 = bj * CB;
    const int wi = wave >> 1, wj = wave & 1;         // wave's sub-block inside the CB x CB block

    const long long p_lo = plan.rule_start[o], p_hi = plan.rule_start[o + 1];
    const long long per = plan.per;
    const long long p0 = p_lo + (long long)s * per;
    const long long p1 = p0 + per < p_hi ? p0 + per : p_hi;

    f32x4 acc[TW][TW];
#pragma unroll
    for (int a = 0; a < TW; ++a)
#pragma unroll
        for (int b = 0; b < TW; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // staging role of this thread: row r_st (+ ROWS_PER_PASS per pass), 16-byte piece c4 = channels 4*c4 .. 4*c4+3
    const int c4 = tid % F4_PER_ROW, r_st = tid / F4_PER_ROW;
    const bool x_ok = ci0 + 4 * c4 + 3 < cin, y_ok = co0 + 4 * c4 + 3 < cout;     // channel counts are multiples of 4
    // LDS position of channel c = 4*c4 + j: tile t = c / 16, i = c % 16 -> i * NT + t
    const int st_pos = ((4 * c4) & 15) * NT + (4 * c4) / 16;

    int ri[PASSES], ro[PASSES], rin[PASSES], ron[PASSES];
    f32x4 vx[PASSES], vy[PASSES];
    // Bias gradient for free: when the offsets in db_mask together name every output row exactly once (the centre
    // offset of a submanifold conv, all 8 offsets of a Deconvolution, the identity list), db = sum of the dY rows this
    // kernel stages anyway.  Workgroups of the first Cin-block add up the 16-byte pieces they write to LDS; the
    // per-unit column sums go to db_slabs[unit][cout_pad] and k_wgrad_sum adds them in unit order.
    const bool do_db = db_slabs != nullptr && ((db_mask >> o) & 1u) && bi == 0;
    f32x4 dbacc = {0.f, 0.f, 0.f, 0.f};

#define WG_LOAD_IDX(PC, RI, RO)                                                                      \
    _Pragma("unroll") for (int q = 0; q < PASSES; ++q) {                                             \
        const long long p_ = (PC) + r_st + q * ROWS_PER_PASS;                                        \
        long long pl_ = p_ < p1 ? p_ : p1 - 1;              /* clamp: the load stays unconditional */ \
        if (pl_ < p_lo) pl_ = p_lo < p_hi ? p_lo : 0;                                                \
        const int vi_ = in_rows ? in_rows[pl_] : (int)pl_;                                           \
        const int vo_ = out_rows ? out_rows[pl_] : (int)pl_;                                         \
        RI[q] = p_ < p1 ? vi_ : -1;                                                                  \
        RO[q] = p_ < p1 ? vo_ : -1;                                                                  \
    }
#define WG_LOAD_ROWS(RI, RO)                                                                         \
    _Pragma("unroll") for (int q = 0; q < PASSES; ++q) {                                             \
        const float* xp_ = X + (long long)(RI[q] < 0 ? 0 : RI[q]) * cin + ci0 + 4 * c4;              \
        const float* yp_ = dY + (long long)(RO[q] < 0 ? 0 : RO[q]) * cout + co0 + 4 * c4;            \
        vx[q] = (f32x4){0.f, 0.f, 0.f, 0.f};                                                         \
        vy[q] = vx[q];                                                                               \
        if (x_ok) vx[q] = *(const f32x4*)xp_;               /* loop-invariant per thread */          \
        if (y_ok) vy[q] = *(const f32x4*)yp_;                                                        \
    }
#define WG_STORE_ROWS(BUF, RI, RO)                                                                   \
    _Pragma("unroll") for (int q = 0; q < PASSES; ++q) {                                             \
        f32x4 a_ = vx[q], b_ = vy[q];                                                                \
        if (RI[q] < 0) a_ = (f32x4){0.f, 0.f, 0.f, 0.f};                                             \
        if (RO[q] < 0) b_ = (f32x4){0.f, 0.f, 0.f, 0.f};                                             \
        if (relu_in) { _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) a_[e_] = fmaxf(a_[e_], 0.f); } \
        if (do_db) dbacc += b_;                 /* bias gradient: column sums of dY, see below */      \
        float* xd_ = lds + (BUF) * 2 * OPND + (r_st + q * ROWS_PER_PASS) * LD + st_pos;              \
        float* yd_ = xd_ + OPND;                                                                     \
        _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) {                                           \
            xd_[e_ * NT] = a_[e_];                                                                   \
            yd_[e_ * NT] = b_[e_];                                                                   \
        }                                                                                            \
    }

    if (p0 < p1) {
        WG_LOAD_IDX(p0, ri, ro);
        WG_LOAD_ROWS(ri, ro);
        WG_LOAD_IDX(p0 + WG_PK, rin, ron);
        WG_STORE_ROWS(0, ri, ro);
        __syncthreads();
        int buf = 0;
        for (long long pc = p0; pc < p1; pc += WG_PK, buf ^= 1) {
            const bool more = pc + WG_PK < p1;
            // next chunk's rows -> registers (indices were fetched one chunk earlier), indices of the chunk after
            int ri2[PASSES], ro2[PASSES];
            if (more) {
                WG_LOAD_ROWS(rin, ron);
                WG_LOAD_IDX(pc + 2 * WG_PK, ri2, ro2);
            }
            // multiply the current chunk: 8 steps of 4 rules; fragments of the wave's TW tiles in one LDS read each
            const float* xs = lds + buf * 2 * OPND + i * NT + wi * TW;
            const float* ys = lds + buf * 2 * OPND + OPND + i * NT + wj * TW;
#pragma unroll
            for (int st = 0; st < WG_PK / 4; ++st) {
                const fragT a = *(const fragT*)(xs + (4 * st + kq) * LD);
                const fragT b = *(const fragT*)(ys + (4 * st + kq) * LD);
#pragma unroll
                for (int ta = 0; ta < TW; ++ta)
#pragma unroll
                    for (int tb = 0; tb < TW; ++tb) {
                        acc[ta][tb] = MFMA16(a[ta], b[tb], acc[ta][tb]);
                    }
            }
            if (more) {
                WG_STORE_ROWS(buf ^ 1, rin, ron);
#pragma unroll
                for (int q = 0; q < PASSES; ++q) { rin[q] = ri2[q]; ron[q] = ro2[q]; }
            }
            __syncthreads();
        }
    }
#undef WG_LOAD_IDX
#undef WG_LOAD_ROWS
#undef WG_STORE_ROWS

    if (do_db) {        // reduce the per-thread pieces over the ROWS_PER_PASS threads that share a channel group (LDS reuse)
        __syncthreads();
        float* red = lds;
        *(f32x4*)(red + r_st * CB + 4 * c4) = dbacc;
        __syncthreads();
        if (tid < CB) {
            float t = 0.f;
            for (int r = 0; r < ROWS_PER_PASS; ++r) t += red[r * CB + tid];
            db_slabs[(long long)unit * cout_pad + co0 + tid] = t;
        }
    }

    // partial block -> slab [unit][block][CB][CB]  (row = channel of X, col = channel of dY)
    float* slab = slabs + ((long long)unit * (plan.nbi * plan.nbj) + blockIdx.z) * (CB * CB);
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

// dW[o][ci][co] = sum over the K-splits, fixed association: 4 interleaved partial sums (splits s, s+4, ...) per element
// computed by 4 threads, combined through LDS in a fixed order.  64 elements x 4 split lanes per block.
__global__ __launch_bounds__(256) void k_wgrad_sum(const float* __restrict__ slabs, WgradPlan plan, int cin, int cout,
                                                   float* __restrict__ dW, const float* __restrict__ db_slabs,
                                                   unsigned db_mask, int cout_pad, float* __restrict__ db, int main_blocks) {
    if ((int)blockIdx.x >= main_blocks) {           // trailing blocks: bias gradient, one column each
        const int co = blockIdx.x - main_blocks;
        float s = 0.f;
        for (int o = 0; o < plan.n_off; ++o) {
            if (!((db_mask >> o) & 1u)) continue;
            for (int u = plan.unit_start[o] + threadIdx.x; u < plan.unit_start[o + 1]; u += 256)
                s += db_slabs[(long long)u * cout_pad + co];
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) s += __shfl_down(s, d);
        __shared__ float w[4];
        if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) db[co] = (w[0] + w[1]) + (w[2] + w[3]);
        return;
    }
    const long long total = (long long)plan.n_off * cin * cout;
    const int cb = plan.cb, nblk = plan.nbi * plan.nbj;
    const int el = threadIdx.x & 63, sl = threadIdx.x >> 6;
    __shared__ float part[4][64];
    for (long long e0 = (long long)blockIdx.x * 64; e0 < total; e0 += (long long)main_blocks * 64) {
        const long long e = e0 + el;
        float sum = 0.f;
        if (e < total) {
            const int co = (int)(e % cout);
            const int ci = (int)((e / cout) % cin);
            const int o = (int)(e / ((long long)cin * cout));
            const int blk = (ci / cb) * plan.nbj + co / cb;
            const int u0 = plan.unit_start[o], nu = plan.unit_start[o + 1] - u0;
            const long long base = ((long long)u0 * nblk + blk) * (cb * cb) + (ci % cb) * cb + co % cb;
            for (int s = sl; s < nu; s += 4) sum += slabs[base + (long long)s * nblk * cb * cb];
        }
        part[sl][el] = sum;
        __syncthreads();
        if (sl == 0 && e < total) dW[e] = (part[0][el] + part[1][el]) + (part[2][el] + part[3][el]);
        __syncthreads();
    }
}

static int make_plan(int cin, int cout, const int64_t* prefix_host, int n_off, WgradPlan& pl) {
    pl.n_off = n_off;
    pl.cb = (cin > 64 || cout > 64) ? 128 : ((cin > 32 || cout > 32) ? 64 : 32);
    if (const char* e = getenv("SCN_WGRAD_CB")) pl.cb = atoi(e);             // developer override (tools/ablate_wgrad.py)
    pl.nbi = (int)cdiv(cin, pl.cb);
    pl.nbj = (int)cdiv(cout, pl.cb);
    pl.rule_start[0] = prefix_host[0];
    for (int o = 0; o < n_off; ++o) {
        if (prefix_host[o + 1] < prefix_host[o]) return SCN_EINVAL;
        pl.rule_start[o + 1] = prefix_host[o + 1];
    }
    // work units: ~4 per CU (2 at CB = 128) over all offsets together, at least 4 chunks of rules each
    const int64_t total = prefix_host[n_off] - prefix_host[0];
    const int64_t nblk = (int64_t)pl.nbi * pl.nbj;
    int64_t target = (pl.cb == 128 && nblk == 1) ? 512 : 1024 / nblk;      // measured: tools/ablate_wgrad.py
    if (const char* e = getenv("SCN_WGRAD_SPLITS")) target = atoi(e);       // developer override
    if (target < 1) target = 1;
    int64_t per = cdiv(cdiv(total, target), WG_PK) * WG_PK;
    if (per < 4 * WG_PK) per = 4 * WG_PK;
    pl.per = per;
    pl.unit_start[0] = 0;
    for (int o = 0; o < n_off; ++o)
        pl.unit_start[o + 1] = pl.unit_start[o] + (int)cdiv(prefix_host[o + 1] - prefix_host[o], per);
    return SCN_OK;
}

extern "C" int64_t scn_wgrad_scratch_bytes(int cin, int cout, const int64_t* prefix_host, int n_off) {
    if (!prefix_host || n_off < 1 || n_off > 32 || cin < 1 || cout < 1) return -1;
    WgradPlan pl;
    if (make_plan(cin, cout, prefix_host, n_off, pl) != SCN_OK) return -1;
    const int64_t fast = (int64_t)pl.unit_start[n_off] * ((int64_t)pl.nbi * pl.nbj * pl.cb * pl.cb + (int64_t)pl.nbj * pl.cb) *
                             (int64_t)sizeof(float) + 512;
    const int64_t simple = scn::wgrad_simple_scratch_bytes(cin, cout, prefix_host, n_off);
    return fast > simple ? fast : simple;
}

static int wgrad_impl(const float* X, int cin, const float* dY, int cout, const int32_t* in_rows,
                      const int32_t* out_rows, const int64_t* prefix_host, int n_off, float* dW, float* db,
                      unsigned db_mask, void* scratch, int flags, scn_stream_t stream);

extern "C" int scn_wgrad_rules(const float* X, int cin, const float* dY, int cout, const int32_t* in_rows,
                               const int32_t* out_rows, const int64_t* prefix_host, int n_off, float* dW, void* scratch,
                               int flags, scn_stream_t stream) {
    return wgrad_impl(X, cin, dY, cout, in_rows, out_rows, prefix_host, n_off, dW, nullptr, 0u, scratch, flags, stream);
}

extern "C" int scn_wgrad_bias_rules(const float* X, int cin, const float* dY, int cout, const int32_t* in_rows,
                                    const int32_t* out_rows, const int64_t* prefix_host, int n_off, float* dW, float* db,
                                    uint32_t db_offsets, void* scratch, int flags, scn_stream_t stream) {
    SCN_REQUIRE(db && db_offsets);
    return wgrad_impl(X, cin, dY, cout, in_rows, out_rows, prefix_host, n_off, dW, db, db_offsets, scratch, flags,
                      stream);
}

static int wgrad_impl(const float* X, int cin, const float* dY, int cout, const int32_t* in_rows,
                      const int32_t* out_rows, const int64_t* prefix_host, int n_off, float* dW, float* db,
                      unsigned db_mask, void* scratch, int flags, scn_stream_t stream) {
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
    if (!aligned) {
        int rc = scn::wgrad_simple(X, cin, dY, cout, in_rows, out_rows, prefix_host, n_off, dW, scratch, flags, stream);
        if (rc == SCN_OK && db)
            return scn::fail(SCN_EINVAL, "%sscn_wgrad_bias_rules needs channel counts that are multiples of 4 "
                                         "(use scn_wgrad_rules + scn_colsum)", "");
        return rc;
    }
    if (pl.unit_start[n_off] == 0) {
        SCN_HIP(hipMemsetAsync(dW, 0, sizeof(float) * (size_t)n_off * cin * cout, S(stream)));
        if (db) SCN_HIP(hipMemsetAsync(db, 0, sizeof(float) * (size_t)cout, S(stream)));
        return SCN_OK;
    }
    const int cout_pad = pl.nbj * pl.cb;
    float* db_slabs = db ? (float*)scratch + (int64_t)pl.unit_start[n_off] * pl.nbi * pl.nbj * pl.cb * pl.cb : nullptr;
    dim3 grid((unsigned)pl.unit_start[n_off], 1, (unsigned)(pl.nbi * pl.nbj));
    const int relu_in = (flags & SCN_F_RELU_IN) ? 1 : 0;
#define LAUNCH_WG(TW_)                                                                                      \
    do {                                                                                                        \
        const size_t lds_ = (size_t)2 * 2 * WG_PK * (32 * TW_ + 4) * sizeof(float);                             \
        static bool attr_set = false;                                                                           \
        if (!attr_set) {                                                                                        \
            SCN_HIP(hipFuncSetAttribute((const void*)k_wgrad_lds<TW_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                        160 * 1024));                                                           \
            attr_set = true;                                                                                    \
        }                                                                                                       \
        hipLaunchKernelGGL(k_wgrad_lds<TW_>, grid, dim3(256), lds_, S(stream), X, cin, dY, cout, in_rows, out_rows, \
                           pl, (float*)scratch, relu_in, db_slabs, db_mask, cout_pad);                          \
    } while (0)
    if (pl.cb == 128) LAUNCH_WG(4);
    else if (pl.cb == 64) LAUNCH_WG(2);
    else LAUNCH_WG(1);
#undef LAUNCH_WG
    SCN_LAUNCH_CHECK();
    const int main_blocks = scn::ew_grid((int64_t)n_off * cin * cout, 64);
    hipLaunchKernelGGL(k_wgrad_sum, dim3(main_blocks + (db ? cout : 0)), dim3(256), 0, S(stream),
                       (const float*)scratch, pl, cin, cout, dW, (const float*)db_slabs, db_mask, cout_pad, db,
                       main_blocks);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}
