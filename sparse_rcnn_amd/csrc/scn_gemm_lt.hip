// Row GEMMs of libscn_mi355x with the A tile staged through LDS (gfx950, fp32 MFMA 32x32x2).
//
//   rows  : Y[r]            = residual[r] + bias + in([X0[r] | X1[r]]) . W        NetworkInNetwork / SubM 1^3 / Linear,
//                                                                                 their backward-data (W transposed, one
//                                                                                 input, outputs [Y0 | Y1])
//   rules : Y[out_rows[p]]  = bias + in(X[in_rows[p]]) . W[o(p)]                  Deconvolution fwd, Convolution bwd-data
//
// The two row sources / destinations are the parts of a JoinTable (module_factory.py:298-301, 365-367): the 2C -> C
// NetworkInNetwork of a decoder level reads (up, skip) from their own slabs and its backward writes the two gradients
// into their own slabs -- no concatenated slab, no second launch with the first one's output as residual.
//
// Workgroup = 4 waves on a tile of RT = 128 / NC rows x NC column chunks of 32 (NC = 1, 2, 4 by output width): wave w
// owns row sub-tile w / NC and column chunk w % NC, a 32x32 block on sixteen accumulator registers.  K runs in chunks of
// 32 channels: the RT x 32 A chunk is loaded once per workgroup with one 16-byte load per thread and piece (a row's 128
// contiguous bytes per chunk), widened / ReLU'd there, and parked in LDS (pitch 36 floats: the float4 fragment reads are
// conflict-free); the chunk after it is already in registers while the MFMAs of this one run (register-staged double
// buffer, one barrier per chunk).  B fragments come straight from the (cache-resident) weights, one chunk ahead as well.
// Measured and not kept (tools/ablate_gemm.py, twelve launches of the cfg-2 decoder: 228 us as written, 276 us for the
// register-only kernels): the whole K extent of a tile requested at once (243 us, 158-218 VGPRs), and the B fragments
// of all its chunks too (272 us, spills) -- a wave's 128 dependent 32x32x2 MFMAs (3.4 us at K = 256) and the launch ramp,
// not a chain of memory round trips, are the floor of the deep levels.
// Summation order per output element = ascending 8-channel groups, inside a group the 32x32x2 pairing (k, k+4): the same
// as scn_conv.hip's kernels, whose results these reproduce bit for bit (single source, single destination).
#include "scn_common.h"

using scn::S;
using scn::cdiv;

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

namespace {

constexpr int KC = 32;          // channels per staged chunk (four 8-channel MFMA groups)
constexpr int PITCH = KC + 4;   // floats per LDS row: the float4 fragment reads of 16 rows hit 64 distinct banks

struct LtSeg {                  // rule form: tiles of RT rules inside one offset
    long long rule_start[33];
    long long tile_start[33];
    int n_off;
};

struct LtArgs {
    const void* X0; const void* X1; int cx0, cx1;          // sources (cx1 == 0: one)
    void* Y0; void* Y1; int cy0, cy1;                      // destinations (cy1 == 0: one)
    const float* W; const float* bias; const void* residual; const void* relu_mask;
    const int* in_rows; const int* out_rows;               // rule form only
    long long n;                                           // row form: number of rows
    int cin, cout, flags;
};

__device__ __forceinline__ float bf16w(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ unsigned short bf16n(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
__device__ __forceinline__ int acc_row(int v, int h) { return (v & 3) + 8 * (v >> 2) + 4 * h; }

template <int NC, bool WT, bool HB, bool RULES>
__global__ __launch_bounds__(256) void k_gemm_lt(LtArgs a, LtSeg seg) {
    constexpr int RT = 128 / NC;            // rows per workgroup
    constexpr int NP = RT / 32;             // 16-byte pieces a thread stages per chunk
    __shared__ float As[2][RT * PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 31, h = lane >> 5;
    const int rsub = wave / NC, cchunk = wave % NC;

    long long p0, p_end;
    const float* Wo = a.W;
    if (RULES) {
        const long long tile = blockIdx.x;
        int o = 0;
        while (tile >= seg.tile_start[o + 1]) ++o;
        p0 = seg.rule_start[o] + (tile - seg.tile_start[o]) * RT;
        p_end = seg.rule_start[o + 1];
        Wo += (long long)o * a.cin * a.cout;
    } else {
        p0 = (long long)blockIdx.x * RT;
        p_end = a.n;
    }
    const int n = blockIdx.y * (NC * 32) + cchunk * 32 + m;
    const bool n_ok = n < a.cout;
    const bool relu_in = a.flags & SCN_F_RELU_IN;

    // staging roles: piece i of this thread = row (tid + 256 i) / 8 of the tile, channels 4 * (tid % 8) ..+3 of the chunk
    int srow[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const long long p = p0 + (tid + 256 * i) / 8;
        srow[i] = -1;
        if (p < p_end) srow[i] = RULES ? a.in_rows[p] : (int)p;
    }
    const int c4 = 4 * (tid & 7);

    auto load_a = [&](int kc, float4 (&st)[NP]) {
        const int k = kc + c4;
        const bool second = a.cx1 && k >= a.cx0;
        const char* base = (const char*)(second ? a.X1 : a.X0);
        const int pitch = second ? a.cx1 : a.cx0, kk = second ? k - a.cx0 : k;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (srow[i] >= 0 && k < a.cin) {
                if (HB) {
                    const uint2 r = *(const uint2*)(base + 2 * ((long long)srow[i] * pitch + kk));
                    v = make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u),
                                    __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u));
                } else {
                    v = *(const float4*)(base + 4 * ((long long)srow[i] * pitch + kk));
                }
                if (relu_in) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            }
            st[i] = v;
        }
    };
    auto park_a = [&](int buf, const float4 (&st)[NP]) {
#pragma unroll
        for (int i = 0; i < NP; ++i) *(float4*)&As[buf][((tid + 256 * i) / 8) * PITCH + c4] = st[i];
    };
    auto load_b = [&](int kc, float4 (&b)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k0 = kc + 8 * q + 4 * h;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (n_ok && k0 < a.cin) {
                if (WT) {
                    v = *(const float4*)(Wo + (long long)n * a.cin + k0);
                } else {
                    const float* wp = Wo + (long long)k0 * a.cout + n;
                    v = make_float4(wp[0], wp[a.cout], wp[2 * a.cout], wp[3 * a.cout]);
                }
            }
            b[q] = v;
        }
    };

    f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;

    float4 st[NP], bcur[4], bnext[4];
    load_a(0, st);
    load_b(0, bcur);
    park_a(0, st);
    __syncthreads();
    const int n_chunks = (a.cin + KC - 1) / KC;
    for (int c = 0; c < n_chunks; ++c) {
        const int buf = c & 1;
        const bool more = c + 1 < n_chunks;
        if (more) {                                    // next chunk: global -> registers while this chunk's MFMAs run
            load_a((c + 1) * KC, st);
            load_b((c + 1) * KC, bnext);
        }
        const float* arow = &As[buf][(rsub * 32 + m) * PITCH + 4 * h];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 av = *(const float4*)(arow + 8 * q);
            acc = MFMA32(av.x, bcur[q].x, acc);
            acc = MFMA32(av.y, bcur[q].y, acc);
            acc = MFMA32(av.z, bcur[q].z, acc);
            acc = MFMA32(av.w, bcur[q].w, acc);
        }
        if (more) {
            park_a(buf ^ 1, st);                       // the other buffer: its last readers passed the previous barrier
#pragma unroll
            for (int q = 0; q < 4; ++q) bcur[q] = bnext[q];
        }
        __syncthreads();
    }

    // epilogue: acc[v] = D[acc_row(v, h)][m] of this wave's 32 x 32 block
    const long long prow = p0 + rsub * 32 + m;                               // this lane's tile row, for the shuffle below
    long long orow_l = -1;
    if (prow < p_end) orow_l = RULES ? (long long)a.out_rows[prow] : prow;
    const int orow_i = (int)orow_l;                                          // rows < 2^31 (checked by the host side)
    const float bv = (a.bias && n_ok) ? a.bias[n] : 0.f;
    const bool second = a.cy1 && n >= a.cy0;
    char* ybase = (char*)(second ? a.Y1 : a.Y0);
    const int ypitch = second ? a.cy1 : a.cy0, yn = second ? n - a.cy0 : n;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int r = __shfl(orow_i, acc_row(v, h));
        if (r >= 0 && n_ok) {
            const long long off = (long long)r * ypitch + yn;
            float y = acc[v] + bv;
            if (HB) {
                if (a.residual) y += bf16w(((const unsigned short*)a.residual)[off]);
                if (a.relu_mask && !(bf16w(((const unsigned short*)a.relu_mask)[off]) > 0.f)) y = 0.f;
                ((unsigned short*)ybase)[off] = bf16n(y);
            } else {
                if (a.residual) y += ((const float*)a.residual)[off];
                if (a.relu_mask && !(((const float*)a.relu_mask)[off] > 0.f)) y = 0.f;
                ((float*)ybase)[off] = y;
            }
        }
    }
}

template <bool WT, bool HB, bool RULES>
int launch_nc(const LtArgs& a, const LtSeg& seg, long long tiles_of_128, scn_stream_t stream) {
    const int nc = a.cout <= 32 ? 1 : a.cout <= 64 ? 2 : 4;
    const unsigned gy = (unsigned)cdiv(a.cout, nc * 32);
    (void)tiles_of_128;
    long long gx;
    if (RULES) gx = seg.tile_start[seg.n_off];
    else gx = cdiv(a.n, 128 / nc);
    if (gx <= 0) return SCN_OK;
    SCN_REQUIRE(gx < 2147483647LL);
    dim3 grid((unsigned)gx, gy);
    if (nc == 1) hipLaunchKernelGGL((k_gemm_lt<1, WT, HB, RULES>), grid, dim3(256), 0, S(stream), a, seg);
    else if (nc == 2) hipLaunchKernelGGL((k_gemm_lt<2, WT, HB, RULES>), grid, dim3(256), 0, S(stream), a, seg);
    else hipLaunchKernelGGL((k_gemm_lt<4, WT, HB, RULES>), grid, dim3(256), 0, S(stream), a, seg);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

template <bool RULES>
int launch_lt(const LtArgs& a, const LtSeg& seg, bool hb, scn_stream_t stream) {
    const bool wt = a.flags & SCN_F_W_TRANSPOSED;
    if (hb) return wt ? launch_nc<true, true, RULES>(a, seg, 0, stream) : launch_nc<false, true, RULES>(a, seg, 0, stream);
    return wt ? launch_nc<true, false, RULES>(a, seg, 0, stream) : launch_nc<false, false, RULES>(a, seg, 0, stream);
}

}  // namespace

namespace scn {

// Shapes the LDS-tiled kernel takes: channel counts in 8-channel groups on 16-byte aligned slabs.
bool gemm_lt_usable(const void* X0, const void* X1, int cx0, int cx1, const void* W) {
    auto al = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    return cx0 % 8 == 0 && cx1 % 8 == 0 && al(X0) && al(X1) && al(W);
}

int gemm_lt_rows(const void* X0, int cx0, const void* X1, int cx1, int64_t n, const float* W, const float* bias,
                 const void* residual, const void* relu_mask, void* Y0, int cy0, void* Y1, int cy1, int flags, bool hb,
                 scn_stream_t stream) {
    LtArgs a{};
    a.X0 = X0; a.X1 = X1; a.cx0 = cx0; a.cx1 = cx1;
    a.Y0 = Y0; a.Y1 = Y1; a.cy0 = cy0; a.cy1 = cy1;
    a.W = W; a.bias = bias; a.residual = residual; a.relu_mask = relu_mask;
    a.n = n; a.cin = cx0 + cx1; a.cout = cy0 + cy1; a.flags = flags;
    LtSeg seg{};
    return launch_lt<false>(a, seg, hb, stream);
}

int gemm_lt_rules(const void* X, int cin, const int32_t* in_rows, const int32_t* out_rows, const int64_t* prefix_host,
                  int n_off, const float* W, const float* bias, const void* relu_mask, void* Y, int cout, int flags,
                  bool hb, scn_stream_t stream) {
    LtArgs a{};
    a.X0 = X; a.cx0 = cin; a.Y0 = Y; a.cy0 = cout;
    a.W = W; a.bias = bias; a.relu_mask = relu_mask; a.in_rows = in_rows; a.out_rows = out_rows;
    a.cin = cin; a.cout = cout; a.flags = flags;
    const int nc = cout <= 32 ? 1 : cout <= 64 ? 2 : 4;
    const int rt = 128 / nc;
    LtSeg seg{};
    seg.n_off = n_off;
    seg.rule_start[0] = prefix_host[0];
    seg.tile_start[0] = 0;
    for (int o = 0; o < n_off; ++o) {
        const int64_t cnt = prefix_host[o + 1] - prefix_host[o];
        SCN_REQUIRE(cnt >= 0);
        seg.rule_start[o + 1] = prefix_host[o + 1];
        seg.tile_start[o + 1] = seg.tile_start[o] + cdiv(cnt, rt);
    }
    return launch_lt<true>(a, seg, hb, stream);
}

}  // namespace scn

/* Row GEMM over the parts of a JoinTable (include/scn_mi355x.h). */
extern "C" int scn_gemm_rows2(const void* X0, int cx0, const void* X1, int cx1, int64_t n, const float* W,
                              const float* bias, const void* residual, const void* relu_mask, void* Y0, int cy0, void* Y1,
                              int cy1, int flags, int bf16_storage, scn_stream_t stream) {
    SCN_REQUIRE(n >= 0 && cx0 >= 8 && cx1 >= 0 && cy0 >= 1 && cy1 >= 0);
    SCN_REQUIRE(cx0 % 8 == 0 && cx1 % 8 == 0);
    SCN_REQUIRE((cx1 == 0) == (X1 == nullptr) && (cy1 == 0) == (Y1 == nullptr));
    SCN_REQUIRE(!(cx1 && cy1));                                  // two sources or two destinations, not both
    SCN_REQUIRE(cy1 == 0 || (!residual && !relu_mask));          // epilogue operands exist for one destination only
    SCN_REQUIRE(n < 2147483647LL);
    if (n == 0) return SCN_OK;
    SCN_REQUIRE(X0 && W && Y0);
    SCN_REQUIRE(scn::gemm_lt_usable(X0, X1, cx0, cx1, W));
    return scn::gemm_lt_rows(X0, cx0, X1, cx1, n, W, bias, residual, relu_mask, Y0, cy0, Y1, cy1, flags,
                             bf16_storage != 0, stream);
}
