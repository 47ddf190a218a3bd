// Feature path of libscn_mi355x: gather-GEMM-scatter on the gfx950 fp32 matrix cores.
//
// Every conv-type layer of the reference's sparse backbone / mask head is one of three GEMM shapes
// (DESIGN.md §Kernels):
//   gemm_table : output-stationary.  Y[r] = sum_o X[table[o][r]] . W[o]   (SubM, Convolution, NiN and the
//                backward-data of SubM / Deconvolution / NiN).  No atomics: an output row is owned by one wave.
//   gemm_rules : rule list, every output row appears once (Deconvolution fwd, Convolution backward-data).
//   wgrad_rules: dW[o] = sum_p X[in_p]^T . dY[out_p], K = number of rules; split over rule chunks, slabs
//                reduced in fixed order (bitwise reproducible, no float atomics).
//
// Arithmetic is exact fp32 on v_mfma_f32_32x32x2_f32 (64 FLOP/clk/SIMD; MI355X_MICROARCH.md §Matrix cores).
// A wave owns a 32x32 output tile: 16 accumulator VGPRs, A/B operands are ONE f32 VGPR each.
//   lane l: m = l & 31 (tile row for A / tile column for B), h = l >> 5 (which of the 2 k of an MFMA step)
//   K is consumed in chunks of 8: lane half h holds k = 8q + 4h + e, e = 0..3 (one float4 per lane for A),
//   MFMA step e multiplies k = 8q+e (h=0 lanes) and k = 8q+4+e (h=1 lanes): a fixed permutation of the
//   summation order, identical for A and B.
//   C/D: acc[v] = D[(v&3) + 8*(v>>2) + 4*h][l & 31]   (cdna_hip_programming.md §3 fragment layout)
#include "scn_common.h"

using scn::S;
using scn::cdiv;

namespace scn {   // scn_gemm_lt.hip: the same GEMMs with the A tile staged through LDS (8-channel groups, aligned slabs)
bool gemm_lt_usable(const void* X0, const void* X1, int cx0, int cx1, const void* W);
int gemm_lt_rows(const void* X0, int cx0, const void* X1, int cx1, int64_t n, const float* W, const float* bias,
                 const void* residual, const void* relu_mask, void* Y0, int cy0, void* Y1, int cy1, int flags, bool hb,
                 scn_stream_t stream);
int gemm_lt_rules(const void* X, int cin, const int32_t* in_rows, const int32_t* out_rows, const int64_t* prefix_host,
                  int n_off, const float* W, const float* bias, const void* relu_mask, void* Y, int cout, int flags,
                  bool hb, scn_stream_t stream);
}

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ int acc_row(int v, int h) { return (v & 3) + 8 * (v >> 2) + 4 * h; }

// One offset's contribution of a 32-row x 32-col tile:  acc += in(Xrow[0..cin)) . Wo[:, n0..n0+32)
//   xrow : this lane's gathered input row (valid iff have)
//   Wo   : weight matrix of this offset; !WT: [cin][cout] row-major;  WT: [cout][cin] row-major (used transposed)
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }

// HB: the feature rows are bf16 (uint16 bit patterns; `xrow` then points at uint16 data) and are widened exactly; the
// weights stay fp32 and the arithmetic is the fp32 kernel's.
template <bool FAST, bool WT, bool HB = false>
__device__ __forceinline__ void tile_mac(f32x16& acc, const float* __restrict__ xrow, bool have, int cin,
                                         const float* __restrict__ Wo, int cout, int n, bool n_ok, int h, bool relu_in) {
    const unsigned short* xh = (const unsigned short*)xrow;
    if (FAST) {   // cin % 8 == 0, rows 16-B aligned (HB: 8-B pieces)
        for (int q = 0; q < cin; q += 8) {
            const int k0 = q + 4 * h;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            if (HB) {
                if (have) {
                    const uint2 r = *(const uint2*)(xh + k0);
                    a = make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u),
                                    __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u));
                }
            } else if (have) a = *(const float4*)(xrow + k0);
            if (relu_in) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
            float4 b;
            if (WT) {
                b = n_ok ? *(const float4*)(Wo + (long long)n * cin + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                const float* wp = Wo + (long long)k0 * cout + n;
                b.x = n_ok ? wp[0] : 0.f;
                b.y = n_ok ? wp[cout] : 0.f;
                b.z = n_ok ? wp[2 * cout] : 0.f;
                b.w = n_ok ? wp[3 * cout] : 0.f;
            }
            acc = MFMA(a.x, b.x, acc);
            acc = MFMA(a.y, b.y, acc);
            acc = MFMA(a.z, b.z, acc);
            acc = MFMA(a.w, b.w, acc);
        }
    } else {      // any cin: guarded scalar loads
        for (int q = 0; q < cin; q += 8) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = q + 4 * h + e;
                const bool k_ok = k < cin;
                float a = (have && k_ok) ? (HB ? bf16_bits_to_f32(xh[k]) : xrow[k]) : 0.f;
                if (relu_in) a = fmaxf(a, 0.f);
                float b = 0.f;
                if (k_ok && n_ok) b = WT ? Wo[(long long)n * cin + k] : Wo[(long long)k * cout + n];
                acc = MFMA(a, b, acc);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// gemm_table
// ------------------------------------------------------------------------------------------------
template <bool FAST, bool WT, bool HB = false>
__global__ __launch_bounds__(256) void k_gemm_table(const float* __restrict__ X, int cin,
                                                    const int* __restrict__ table, int n_off, long long n_out,
                                                    const float* __restrict__ W, const float* __restrict__ bias,
                                                    const float* __restrict__ residual,
                                                    const float* __restrict__ relu_mask, float* __restrict__ Y, int cout,
                                                    int flags) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 31, h = lane >> 5;
    const long long r0 = ((long long)blockIdx.x * 4 + wave) * 32;
    if (r0 >= n_out) return;                       // wave-uniform
    const long long row = r0 + m;
    const bool row_ok = row < n_out;
    const int n0 = blockIdx.y * 32;
    const int n = n0 + m;
    const bool n_ok = n < cout;
    const bool relu_in = flags & SCN_F_RELU_IN;
    const bool rev = flags & SCN_F_OFF_REVERSE;
    const long long wstride = (long long)cin * cout;

    f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;

    int idx_next = row_ok ? (table ? table[row] : (int)row) : -1;
    for (int o = 0; o < n_off; ++o) {
        const int idx = idx_next;
        if (o + 1 < n_off) idx_next = row_ok ? table[(long long)(o + 1) * n_out + row] : -1;
        if (__ballot(idx >= 0) == 0ull) continue;  // no rule of this offset touches the tile
        const float* Wo = W + (long long)(rev ? n_off - 1 - o : o) * wstride;
        const float* xrow = HB ? (const float*)((const unsigned short*)X + (long long)(idx >= 0 ? idx : 0) * cin)
                               : X + (long long)(idx >= 0 ? idx : 0) * cin;
        tile_mac<FAST, WT, HB>(acc, xrow, idx >= 0, cin, Wo, cout, n, n_ok, h, relu_in);
    }

    const float bv = (bias && n_ok) ? bias[n] : 0.f;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const long long r = r0 + acc_row(v, h);
        if (r < n_out && n_ok) {
            const long long off = r * cout + n;
            float y = acc[v] + bv;
            if (HB) {                                   // residual, mask and result are bf16 too
                if (residual) y += bf16_bits_to_f32(((const unsigned short*)residual)[off]);
                if (relu_mask && !(bf16_bits_to_f32(((const unsigned short*)relu_mask)[off]) > 0.f)) y = 0.f;
                ((unsigned short*)Y)[off] = f32_to_bf16_bits(y);
            } else {
                if (residual) y += residual[off];
                if (relu_mask && !(relu_mask[off] > 0.f)) y = 0.f;
                Y[off] = y;
            }
        }
    }
}

static int gemm_table_impl(const float* X, int64_t n_in, int cin, const int32_t* table, int n_off, int64_t n_out,
                           const float* W, const float* bias, const float* residual, const float* relu_mask,
                           float* Y, int cout, int flags, scn_stream_t stream, bool hb) {
    SCN_REQUIRE(n_in >= 0 && n_out >= 0 && cin >= 1 && cout >= 1 && n_off >= 1);
    SCN_REQUIRE(table || (n_off == 1 && n_in == n_out));
    if (n_out == 0) return SCN_OK;
    SCN_REQUIRE(X && W && Y);
    SCN_REQUIRE(n_out * (int64_t)cout < (1LL << 40) && cdiv(n_out, 128) < 2147483647LL);
    const bool fast = (cin % 8 == 0) && (((uintptr_t)X & 15) == 0) && (((uintptr_t)W & 15) == 0);
    const bool wt = flags & SCN_F_W_TRANSPOSED;
    // identity table (NetworkInNetwork, SubM 1^3, Linear): the LDS-tiled row GEMM, same bits (SCN_F_GEMM_V1: this file's)
    if (!table && fast && !(flags & SCN_F_GEMM_V1) && n_out < 2147483647LL)
        return scn::gemm_lt_rows(X, cin, nullptr, 0, n_out, W, bias, residual, relu_mask, Y, cout, nullptr, 0, flags, hb,
                                 stream);
    dim3 grid((unsigned)cdiv(n_out, 128), (unsigned)cdiv(cout, 32));
#define LAUNCH_T(F, T, H)                                                                                     \
    hipLaunchKernelGGL((k_gemm_table<F, T, H>), grid, dim3(256), 0, S(stream), X, cin, table, n_off,          \
                       (long long)n_out, W, bias, residual, relu_mask, Y, cout, flags)
#define PICK_T(H)                                                                                             \
    do {                                                                                                      \
        if (fast && wt) LAUNCH_T(true, true, H);                                                              \
        else if (fast) LAUNCH_T(true, false, H);                                                              \
        else if (wt) LAUNCH_T(false, true, H);                                                                \
        else LAUNCH_T(false, false, H);                                                                       \
    } while (0)
    if (hb) PICK_T(true); else PICK_T(false);
#undef PICK_T
#undef LAUNCH_T
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

extern "C" int scn_gemm_table(const float* X, int64_t n_in, int cin, const int32_t* table, int n_off, int64_t n_out,
                              const float* W, const float* bias, const float* residual, const float* relu_mask,
                              float* Y, int cout, int flags, scn_stream_t stream) {
    return gemm_table_impl(X, n_in, cin, table, n_off, n_out, W, bias, residual, relu_mask, Y, cout, flags, stream, false);
}

// bf16 STORAGE of the features (X, residual, relu_mask, Y: uint16 bit patterns); W, bias fp32; fp32 arithmetic.
extern "C" int scn_gemm_table_bf16(const uint16_t* X, int64_t n_in, int cin, const int32_t* table, int n_off,
                                   int64_t n_out, const float* W, const float* bias, const uint16_t* residual,
                                   const uint16_t* relu_mask, uint16_t* Y, int cout, int flags, scn_stream_t stream) {
    return gemm_table_impl((const float*)X, n_in, cin, table, n_off, n_out, W, bias, (const float*)residual,
                           (const float*)relu_mask, (float*)Y, cout, flags, stream, true);
}

// ------------------------------------------------------------------------------------------------
// gemm_rules: tiles of 32 rules inside one offset
// ------------------------------------------------------------------------------------------------
struct SegTiles {
    long long rule_start[33];   // prefix of rules per offset (n_off <= 32)
    long long tile_start[33];   // prefix of 32-rule tiles per offset
    int n_off;
};

template <bool FAST, bool WT, bool HB = false>
__global__ __launch_bounds__(256) void k_gemm_rules(const float* __restrict__ X, int cin,
                                                    const int* __restrict__ in_rows, const int* __restrict__ out_rows,
                                                    SegTiles seg, const float* __restrict__ W,
                                                    const float* __restrict__ bias, const float* __restrict__ relu_mask,
                                                    float* __restrict__ Y, int cout, int flags) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 31, h = lane >> 5;
    const long long tile = (long long)blockIdx.x * 4 + wave;
    if (tile >= seg.tile_start[seg.n_off]) return;
    int o = 0;
    while (tile >= seg.tile_start[o + 1]) ++o;
    const long long p0 = seg.rule_start[o] + (tile - seg.tile_start[o]) * 32;
    const long long p_end = seg.rule_start[o + 1];
    const long long p = p0 + m;
    const bool p_ok = p < p_end;
    const int idx = p_ok ? in_rows[p] : -1;
    const int orow = p_ok ? out_rows[p] : -1;
    const int n0 = blockIdx.y * 32;
    const int n = n0 + m;
    const bool n_ok = n < cout;

    f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    const float* Wo = W + (long long)o * cin * cout;
    const float* xrow = HB ? (const float*)((const unsigned short*)X + (long long)(idx >= 0 ? idx : 0) * cin)
                           : X + (long long)(idx >= 0 ? idx : 0) * cin;
    tile_mac<FAST, WT, HB>(acc, xrow, idx >= 0, cin, Wo, cout, n, n_ok, h, flags & SCN_F_RELU_IN);

    const float bv = (bias && n_ok) ? bias[n] : 0.f;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int r = __shfl(orow, acc_row(v, h));
        if (r >= 0 && n_ok) {
            const long long off = (long long)r * cout + n;
            float y = acc[v] + bv;
            if (HB) {
                if (relu_mask && !(bf16_bits_to_f32(((const unsigned short*)relu_mask)[off]) > 0.f)) y = 0.f;
                ((unsigned short*)Y)[off] = f32_to_bf16_bits(y);
            } else {
                if (relu_mask && !(relu_mask[off] > 0.f)) y = 0.f;
                Y[off] = y;
            }
        }
    }
}

static int make_seg_tiles(const int64_t* prefix_host, int n_off, int tile, SegTiles& seg) {
    seg.n_off = n_off;
    seg.rule_start[0] = prefix_host[0];
    seg.tile_start[0] = 0;
    for (int o = 0; o < n_off; ++o) {
        int64_t cnt = prefix_host[o + 1] - prefix_host[o];
        if (cnt < 0) return SCN_EINVAL;
        seg.rule_start[o + 1] = prefix_host[o + 1];
        seg.tile_start[o + 1] = seg.tile_start[o] + cdiv(cnt, tile);
    }
    return SCN_OK;
}

static int gemm_rules_impl(const float* X, int cin, const int32_t* in_rows, const int32_t* out_rows,
                           const int64_t* prefix_host, int n_off, const float* W, const float* bias,
                           const float* relu_mask, float* Y, int cout, int flags, scn_stream_t stream, bool hb) {
    SCN_REQUIRE(prefix_host && n_off >= 1 && n_off <= 32 && cin >= 1 && cout >= 1);
    SegTiles seg;
    SCN_REQUIRE(make_seg_tiles(prefix_host, n_off, 32, seg) == SCN_OK);
    const long long tiles = seg.tile_start[n_off];
    if (tiles == 0) return SCN_OK;
    SCN_REQUIRE(X && in_rows && out_rows && W && Y);
    const bool fast = (cin % 8 == 0) && (((uintptr_t)X & 15) == 0) && (((uintptr_t)W & 15) == 0);
    const bool wt = flags & SCN_F_W_TRANSPOSED;
    // (one 32-column chunk: the four waves of a workgroup would share nothing -- the register kernel is faster there)
    if (fast && cout > 32 && !(flags & SCN_F_GEMM_V1))
        return scn::gemm_lt_rules(X, cin, in_rows, out_rows, prefix_host, n_off, W, bias, relu_mask, Y, cout, flags, hb,
                                  stream);
    dim3 grid((unsigned)cdiv(tiles, 4), (unsigned)cdiv(cout, 32));
#define LAUNCH_R(F, T, H)                                                                                     \
    hipLaunchKernelGGL((k_gemm_rules<F, T, H>), grid, dim3(256), 0, S(stream), X, cin, in_rows, out_rows, seg, \
                       W, bias, relu_mask, Y, cout, flags)
#define PICK_R(H)                                                                                             \
    do {                                                                                                      \
        if (fast && wt) LAUNCH_R(true, true, H);                                                              \
        else if (fast) LAUNCH_R(true, false, H);                                                              \
        else if (wt) LAUNCH_R(false, true, H);                                                                \
        else LAUNCH_R(false, false, H);                                                                       \
    } while (0)
    if (hb) PICK_R(true); else PICK_R(false);
#undef PICK_R
#undef LAUNCH_R
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

extern "C" int scn_gemm_rules(const float* X, int cin, const int32_t* in_rows, const int32_t* out_rows,
                              const int64_t* prefix_host, int n_off, const float* W, const float* bias,
                              const float* relu_mask, float* Y, int cout, int flags, scn_stream_t stream) {
    return gemm_rules_impl(X, cin, in_rows, out_rows, prefix_host, n_off, W, bias, relu_mask, Y, cout, flags, stream,
                           false);
}

extern "C" int scn_gemm_rules_bf16(const uint16_t* X, int cin, const int32_t* in_rows, const int32_t* out_rows,
                                   const int64_t* prefix_host, int n_off, const float* W, const float* bias,
                                   const uint16_t* relu_mask, uint16_t* Y, int cout, int flags, scn_stream_t stream) {
    return gemm_rules_impl((const float*)X, cin, in_rows, out_rows, prefix_host, n_off, W, bias,
                           (const float*)relu_mask, (float*)Y, cout, flags, stream, true);
}

// ------------------------------------------------------------------------------------------------
// colsum (bias gradient): two-stage, fixed order
// ------------------------------------------------------------------------------------------------
template <bool HB>
__global__ __launch_bounds__(256) void k_colsum_partial(const float* __restrict__ dY, long long n, int c,
                                                        float* __restrict__ partial) {
    const unsigned short* dYh = (const unsigned short*)dY;             // HB: bf16-stored rows, widened exactly
    // thread t owns column (t % cpad) of rows (t / cpad) + k*rows_per_pass inside this block's row range
    const long long rows_per_block = (n + gridDim.x - 1) / gridDim.x;
    const long long r_lo = blockIdx.x * rows_per_block;
    long long r_hi = r_lo + rows_per_block;
    if (r_hi > n) r_hi = n;
    __shared__ float red[256];
    for (int c0 = 0; c0 < c; c0 += 256) {
        const int width = min(256, c - c0);          // columns handled this pass
        const int lanes_per_row = width;             // one thread per column
        const int rows_par = 256 / lanes_per_row > 0 ? 256 / lanes_per_row : 1;
        const int col = threadIdx.x % lanes_per_row;
        const int rsub = threadIdx.x / lanes_per_row;
        float s = 0.f;
        if (rsub < rows_par)
            for (long long r = r_lo + rsub; r < r_hi; r += rows_par)
                s += HB ? bf16_bits_to_f32(dYh[r * c + c0 + col]) : dY[r * c + c0 + col];
        red[threadIdx.x] = (rsub < rows_par) ? s : 0.f;
        __syncthreads();
        if (threadIdx.x < width) {
            float t = 0.f;
            for (int k = 0; k < rows_par; ++k) t += red[k * lanes_per_row + threadIdx.x];
            partial[(long long)blockIdx.x * c + c0 + threadIdx.x] = t;
        }
        __syncthreads();
    }
}

// one block per column: 256 threads sum the partial rows (fixed order), wave shuffle + LDS tree
__global__ __launch_bounds__(256) void k_colsum_final(const float* __restrict__ partial, int nblk, int c,
                                                      float* __restrict__ db) {
    const int col = blockIdx.x;
    float s = 0.f;
    for (int b = threadIdx.x; b < nblk; b += 256) s += partial[(long long)b * c + col];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_down(s, d);
    __shared__ float w[4];
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) db[col] = (w[0] + w[1]) + (w[2] + w[3]);
}

static int colsum_impl(const float* dY, int64_t n, int c, float* db, void* scratch, scn_stream_t stream, bool hb) {
    SCN_REQUIRE(n >= 0 && c >= 1 && db && scratch);
    int nblk = (int)(n < SCN_COLSUM_BLOCKS * 8 ? cdiv(n, 8) : SCN_COLSUM_BLOCKS);
    if (nblk < 1) nblk = 1;
    if (n > 0) SCN_REQUIRE(dY);
    if (hb)
        hipLaunchKernelGGL(k_colsum_partial<true>, dim3(nblk), dim3(256), 0, S(stream), dY, (long long)n, c, (float*)scratch);
    else
        hipLaunchKernelGGL(k_colsum_partial<false>, dim3(nblk), dim3(256), 0, S(stream), dY, (long long)n, c, (float*)scratch);
    SCN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_colsum_final, dim3(c), dim3(256), 0, S(stream), (const float*)scratch, nblk, c, db);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

extern "C" int scn_colsum(const float* dY, int64_t n, int c, float* db, void* scratch, scn_stream_t stream) {
    return colsum_impl(dY, n, c, db, scratch, stream, false);
}

extern "C" int scn_colsum_bf16(const uint16_t* dY, int64_t n, int c, float* db, void* scratch, scn_stream_t stream) {
    return colsum_impl((const float*)dY, n, c, db, scratch, stream, true);
}
