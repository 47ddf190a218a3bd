// k_conv_tss (round 4): the fp32 tile convolution OFFSET-OUTER with the weights STREAMED -- k_conv_ts (scn_conv_ts.hip)
// without its K split, for the layers whose weights do not fit one workgroup's LDS (Cin = 64 / 128: levels 1-2 of the
// benchmark U-Net, 32 of the 62 tile launches of a step).
//
//   Y[r] = residual[r] + bias + sum_o in(X[table[o][r]]) . W[o']
//
// k_conv_ts keeps the 32-channel x 32-column weight slice of ALL offsets resident (110 KB) and splits K over workgroups: a
// (tile, column chunk) is computed as n_kc partial tiles that travel through memory (write-through, tickets, a combine by
// the last arriver: ~6 % of a launch), behind a 110 KB staging front (~5 %), with ~15 vector instructions of bookkeeping per
// 16 MFMAs.  Here a workgroup owns NW tiles (one per wave) x 64 output columns with the FULL K:
//   * the offsets run in the outer loop; the 64-column slice of W[o] (Cin x 64 fp32: 16 / 32 KB) streams through a
//     double-buffered LDS stage, requested two steps ahead into registers (in the order of the row gathers: loads return
//     in order) and written when the previous step's MFMAs have been issued; one workgroup barrier per step;
//   * a wave's 16 x 64 accumulators (4 x f32x4) stay in registers across all offsets; a step is Cin/32 x 32 MFMAs
//     (64 / 128) per wave, so the per-step bookkeeping (offset pop, index read, gather addresses) is paid once per 64-128
//     MFMAs instead of once per 16;
//   * a batch takes NW consecutive tile IDS (tiles are cut from rows sorted by offset mask: nearly the same offsets) and
//     walks only the union of their masks; a wave whose tile lacks the step's offset skips its MFMAs;
//   * a row is gathered as Cin/32 x 2 16-byte pieces per lane, two steps ahead -- a step is 2-4 k cycles of MFMAs per
//     wave, four waves per SIMD: two steps cover the gather latency several times over.
// No K split, no partial tiles, no tickets, no 110 KB front, one write of Y.  LDS image of a slice: [k][64 n] with the
// columns XOR-ed with 16 on every other group of 4 channels (k_conv_ts's conflict-free B reads); a transposed layer weight
// (backward-data) is transposed by the staging write.
// Summation order per output element: offsets ascending, inside an offset the channels in k_conv_ts's step order, K-chunk
// by K-chunk -- k_conv_ts adds the offsets inside a K-chunk and then the K-chunks; the two agree to fp32 rounding
// (tests/test_gpu_exec.py), both are deterministic and independent of placement.
// RESULT: slower than k_conv_ts on the benchmark scene (see conv_tiles_stream below) -- kept as a tested, opt-in path
// (SCN_TS_STREAM=1) and as the record of the experiment VERDICT r3 / DESIGN.md section 10 asked to bound.
#include <stdlib.h>

#include "scn_common.h"

using scn::S;
using scn::cdiv;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

namespace {
constexpr int SS_T = 16;              // rows per tile
constexpr int SS_CT = 64;             // output columns per workgroup

template <int KS, int N_OFF, int NW, bool WT>
__global__ __launch_bounds__(NW * 64) void k_conv_tss(
    const float* __restrict__ X, long long n_in, int cin, const int* __restrict__ tstab,
    const unsigned* __restrict__ tile_mask, const int* __restrict__ perm, long long nt, const float* __restrict__ W,
    const float* __restrict__ bias, const float* __restrict__ residual, const float* __restrict__ relu_mask,
    float* __restrict__ Y, long long n_out, int cout, int flags, int n_chunks, int n_wgb) {
    constexpr int D = 2, THREADS = NW * 64, KT = KS * 32;
    constexpr int SLICE = KT * SS_CT;                               // floats of one offset's slice
    constexpr int PIECES = SLICE / 4;                               // 16-byte pieces
    constexpr int PW = (PIECES + THREADS - 1) / THREADS;
    constexpr int IDXW = (((N_OFF + 1) * SS_T + 63) / 64) * 64;     // row indices of a wave's tile, all offsets + a row of -1
    extern __shared__ __attribute__((aligned(16))) float Ws[];      // [2][KT][64] swizzled, int idx[NW][IDXW], union mask
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int i = lane & 15, kq = lane >> 4;
    int* idx_w = (int*)(Ws + 2 * SLICE) + w * IDXW;
    unsigned* um_s = (unsigned*)((int*)(Ws + 2 * SLICE) + NW * IDXW);
    // the column chunks of a tile batch gather the same rows: block indices 8 apart, i.e. one XCD and one L2
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int chunk = jb % n_chunks, wgb = (jb / n_chunks) * 8 + xcd;
    if (wgb >= n_wgb) return;
    const int n0 = chunk * SS_CT;
    const bool relu_in = flags & SCN_F_RELU_IN;
    const bool rev = flags & SCN_F_OFF_REVERSE;
    const bool res_last = flags & SCN_F_RESIDUAL_LAST;
    float bcol[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) bcol[nb] = bias ? bias[n0 + 16 * nb + i] : 0.f;
    const __amdgpu_buffer_rsrc_t xrsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)(unsigned)(n_in * cin * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t trsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)tstab, 0, (int)(unsigned)(nt * N_OFF * (SS_T * 4)), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, (int)(unsigned)((long long)N_OFF * cin * cout * 4), 0x00020000);
    const bool use_res = residual != nullptr, use_mask = relu_mask != nullptr;
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)residual, 0, use_res ? (int)(unsigned)(n_out * cout * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t mrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)relu_mask, 0, use_mask ? (int)(unsigned)(n_out * cout * 4) : 0, 0x00020000);
    const int row_bytes = cin * 4, lane_boff = 16 * kq, out_row_bytes = cout * 4;
    const int rf_ = relu_in ? 0 : (int)0x80000000;                  // ReLU as one integer max per value (identity: INT_MIN)
    const i32x4 relu_floor4 = {rf_, rf_, rf_, rf_};
    const int w_off_bytes = cin * cout * 4;                         // one offset of the layer's weight
    // a thread's pieces of a slice: byte offset inside W[o'] and destination inside the LDS image
    int wsrc[PW], wdst[PW];
#pragma unroll
    for (int u = 0; u < PW; ++u) {
        const int p = u * THREADS + tid;
        if (p >= PIECES) { wsrc[u] = (int)0x7FFFFFF0; wdst[u] = 0; continue; }
        if (WT) {                     // layer weight [o][n][k], k contiguous: 4 channels of column m
            const int m = p % SS_CT, c4 = p / SS_CT;
            wsrc[u] = ((n0 + m) * cin + 4 * c4) * 4;
            wdst[u] = (4 * c4) * SS_CT + (m ^ ((c4 & 1) << 4));    // + j * 64 for channel 4 c4 + j
        } else {                      // [o][k][n], n contiguous: 4 columns of channel k
            const int k = p / (SS_CT / 4), n4 = p % (SS_CT / 4);
            wsrc[u] = (k * cout + n0 + 4 * n4) * 4;
            wdst[u] = k * SS_CT + ((4 * n4) ^ (((k >> 2) & 1) << 4));
        }
    }
    // B fragment of MFMA step (ks, half, e), column block nb: channel 32 ks + 16 half + 4 kq + e, column 16 nb + i -- in the
    // swizzled image that column sits in physical block nb ^ (kq & 1)
    const float* bA = Ws + (4 * kq) * SS_CT + ((kq & 1) << 4) + i;  // blocks 0 (+32: 2)
    const float* bB = Ws + (4 * kq) * SS_CT + (((kq & 1) ^ 1) << 4) + i;   // blocks 1 (+32: 3)

#define TSS_WLOAD(O, V)                                                                                        \
    do {                                                                                                       \
        const int so_ = (rev ? N_OFF - 1 - (O) : (O)) * w_off_bytes;                                           \
        _Pragma("unroll") for (int u_ = 0; u_ < PW; ++u_)                                                      \
            V[u_] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wsrc[u_], so_, 0);                            \
    } while (0)
#define TSS_WSTORE(BUF, V)                                                                                     \
    _Pragma("unroll") for (int u_ = 0; u_ < PW; ++u_) {                                                        \
        if (u_ * THREADS + tid < PIECES) {                                                                     \
            float* d_ = Ws + (BUF) * SLICE + wdst[u_];                                                         \
            if (WT) {                                                                                          \
                const f32x4 vf_ = __builtin_bit_cast(f32x4, V[u_]);                                            \
                d_[0] = vf_[0]; d_[SS_CT] = vf_[1]; d_[2 * SS_CT] = vf_[2]; d_[3 * SS_CT] = vf_[3];            \
            } else {                                                                                           \
                *(i32x4*)d_ = V[u_];                                                                           \
            }                                                                                                  \
        }                                                                                                      \
    }
#define TSS_GATHER(O, SLOT)                                                                                    \
    do {                                                                                                       \
        const int off_ = __mul24(idx_w[(O) * SS_T + i], row_bytes) + lane_boff;     /* -1 -> out of range -> 0 */ \
        _Pragma("unroll") for (int ks_ = 0; ks_ < KS; ++ks_) {                                                 \
            A[SLOT][ks_][0] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, off_ + 128 * ks_, 0, 0);            \
            A[SLOT][ks_][1] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, off_ + 128 * ks_ + 64, 0, 0);       \
        }                                                                                                      \
    } while (0)
    // pop the lowest offset of the (workgroup-uniform) mask U: its number, or N_OFF (the all-"-1" row) when none is left
#define TSS_POP(U, O)                                                                                          \
    do {                                                                                                       \
        int f_;                                                                                                \
        asm volatile("s_ff1_i32_b32 %0, %1" : "=s"(f_) : "s"(U));                                              \
        (O) = f_ < 0 ? N_OFF : f_;                                                                             \
        (U) &= (U) - 1u;                                                                                       \
    } while (0)
    // one step: barrier; the MFMAs of this step's offset if the wave's tile has it; the next step's slice -> LDS; the register
    // sets just freed take slice and rows of the step two ahead.  SLOT = step & 1 = LDS buffer.
#define TSS_STEP(SLOT)                                                                                         \
    do {                                                                                                       \
        __syncthreads();                                                                                       \
        const int oc_ = oq[SLOT];                                                                              \
        if ((m >> oc_) & 1u) {                                                                                 \
            const float* wa_ = bA + (SLOT) * SLICE;                                                            \
            const float* wb_ = bB + (SLOT) * SLICE;                                                            \
            _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                                                \
                _Pragma("unroll") for (int hf = 0; hf < 2; ++hf) {                                             \
                    /* (whole-vector operations: element-indexed access to a bit-cast ext_vector was compiled to       \
                       element 0 in all four positions -- DESIGN.md section 8) */                                      \
                    const f32x4 af_ = __builtin_bit_cast(f32x4, __builtin_elementwise_max(A[SLOT][ks][hf], relu_floor4)); \
                    float b0_[4], b1_[4], b2_[4], b3_[4];                                                      \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                            \
                        const int ko_ = (32 * ks + 16 * hf + e) * SS_CT;                                       \
                        b0_[e] = wa_[ko_]; b1_[e] = wb_[ko_]; b2_[e] = wa_[ko_ + 32]; b3_[e] = wb_[ko_ + 32];  \
                    }                                                                                          \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                            \
                        const float a_ = af_[e];                                                               \
                        acc[0] = MFMA16(a_, b0_[e], acc[0]);                                                   \
                        acc[1] = MFMA16(a_, b1_[e], acc[1]);                                                   \
                        acc[2] = MFMA16(a_, b2_[e], acc[2]);                                                   \
                        acc[3] = MFMA16(a_, b3_[e], acc[3]);                                                   \
                    }                                                                                          \
                }                                                                                              \
            }                                                                                                  \
        }                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        TSS_WSTORE(1 - (SLOT), wv[1 - (SLOT)]);                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        TSS_POP(ua, oq[SLOT]);                                                                                 \
        TSS_WLOAD(oq[SLOT], wv[SLOT]);                                                                         \
        TSS_GATHER(oq[SLOT], SLOT);                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
    } while (0)

    for (long long bt = wgb; bt * NW < nt; bt += n_wgb) {
        const long long tl = bt * NW + w;
        const long long tile = tl < nt ? tl : -1;                   // (wave-uniform) consecutive tile ids: similar masks
        unsigned m = 0;
        int orow[4] = {-1, -1, -1, -1};
        if (tile >= 0) {
            m = __builtin_amdgcn_readfirstlane(tile_mask[tile]);
#pragma unroll
            for (int j = 0; j < 4; ++j) orow[j] = perm[tile * SS_T + 4 * kq + j];
        }
        if (tid == 0) *um_s = 0u;
        // row indices of all offsets -> this wave's LDS strip (a tile's block of the table is N_OFF x 64 contiguous bytes);
        // row N_OFF of the strip is -1: the "no offset left" slot.  A wave without a tile gets -1 everywhere.
        {
            const int tile_boff = tile >= 0 ? (int)tile * N_OFF * (SS_T * 4) : 0;
            int v[IDXW / 64];
#pragma unroll
            for (int t = 0; t < IDXW / 64; ++t)
                v[t] = (tile >= 0 && t * 64 + lane < N_OFF * SS_T)
                           ? __builtin_amdgcn_raw_buffer_load_b32(trsrc, (t * 64 + lane) * 4, tile_boff, 0) : -1;
#pragma unroll
            for (int t = 0; t < IDXW / 64; ++t) idx_w[t * 64 + lane] = v[t];
        }
        __syncthreads();
        if (lane == 0 && m) atomicOr(um_s, m);
        __syncthreads();
        unsigned ua = __builtin_amdgcn_readfirstlane(*um_s);        // offsets left to request
        const int n_steps = __popc(ua);
        i32x4 A[D][KS][2], wv[D][PW];
        int oq[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            TSS_POP(ua, oq[d]);
            TSS_WLOAD(oq[d], wv[d]);
            TSS_GATHER(oq[d], d);
        }
        TSS_WSTORE(0, wv[0]);                 // (buffer 0: the barrier above freed it)
        f32x4 acc[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) acc[nb] = (f32x4){bcol[nb], bcol[nb], bcol[nb], bcol[nb]};
        int n_left = n_steps;
        for (; n_left >= 2; n_left -= 2) {
            TSS_STEP(0);
            TSS_STEP(1);
        }
        if (n_left >= 1) TSS_STEP(0);
        // ---- epilogue: residual / ReLU-backward mask (a missing operand is a zero-record descriptor), one write
        float rs[4][4], mk[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ro = __mul24(orow[j], out_row_bytes) + (n0 + i) * 4;        // row -1 -> out of range
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                rs[j][nb] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrsrc, ro + 64 * nb, 0, 0));
                mk[j][nb] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(mrsrc, ro + 64 * nb, 0, 0));
            }
        }
        if (tile >= 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (orow[j] < 0) continue;
                float* yp = Y + (long long)orow[j] * cout + n0 + i;
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    float y = acc[nb][j] + (res_last ? 0.f : rs[j][nb]);
                    if (use_mask && !(mk[j][nb] > 0.f)) y = 0.f;
                    if (res_last) y += rs[j][nb];
                    yp[16 * nb] = y;
                }
            }
        }
        __syncthreads();                     // the next batch restages buffer 0 and the strips
    }
#undef TSS_STEP
#undef TSS_POP
#undef TSS_GATHER
#undef TSS_WSTORE
#undef TSS_WLOAD
}
}  // namespace

namespace scn {
// *launched = true: the layer ran here; false: not eligible (the caller runs k_conv_ts).  -> status
int conv_tiles_stream(const float* X, int64_t n_in, int cin, const int32_t* tstab, const uint32_t* tile_mask,
                      const int32_t* perm, int n_off, int64_t n_out, const float* W, const float* bias, const float* residual,
                      const float* relu_mask, float* Y, int cout, int flags, hipStream_t st, bool* launched) {
    *launched = false;
    // OFF by default: measured SLOWER than k_conv_ts on the cfg-2 scene (profiles/r4_tss_fp32_streaming_experiment.txt:
    // 86.5 / 114.4 us per launch at levels 1 / 2 with 4-wave workgroups, 119 / 144 with 16 / 8, against 69.4 / 65.9) -- the
    // lockstep over the union of a batch's offset masks idles the waves whose tile lacks an offset, and the mask-sorted
    // tiles of a surface scene differ too much for any batch size to hide that.  SCN_TS_STREAM=1 runs it (tests, A/B).
    if (scn::sw(scn::SW_TS_STREAM).i != 1) return SCN_OK;           // (scn_debug_set: the tests switch it inside one process)
    const int ks = cin / 32;
    const bool ok = cin % 32 == 0 && (ks == 2 || ks == 4) && cout % SS_CT == 0 && (n_off == 27 || n_off == 8) &&
                    !(flags & SCN_F_SPLIT_SUM) && (((uintptr_t)X | (uintptr_t)W) & 15) == 0 &&
                    n_in < (1ll << 23) && n_in * cin * 4 < (1ll << 32) - (1ll << 24) && n_out < (1ll << 23) &&
                    n_out * cout * 4 < (1ll << 32) - (1ll << 24) && (int64_t)n_off * cin * cout * 4 < (1ll << 31);
    if (!ok) return SCN_OK;
    const int64_t nt = cdiv(n_out, SS_T);
    const int n_chunks = cout / SS_CT;
    const bool wt = flags & SCN_F_W_TRANSPOSED;
    // waves per workgroup (= tiles per batch).  SMALL batches, many resident workgroups: a batch walks the union of its tiles'
    // offset masks in lockstep (one barrier per offset), so a wave idles through the offsets its own tile lacks -- with 16
    // consecutive tiles the union is ~25 of 27 offsets for tiles that have ~14 each, and the SIMD a stalled wave sits on has
    // nobody else to run (measured: 119 us per launch at level 1 against k_conv_ts's 69).  With 4 waves per workgroup (one per
    // SIMD) the union stays near a tile's own mask and the other resident workgroups of the CU fill the matrix pipe.
    const int nw_env = (int)scn::sw(scn::SW_TSS_NW).i;
    const int nw = (nw_env == 4 || nw_env == 8 || nw_env == 16) ? (ks == 4 && nw_env == 16 ? 8 : nw_env) : 4;
    const size_t lds0 = (size_t)2 * ks * 32 * SS_CT * sizeof(float) + (size_t)nw * (((n_off + 1) * 16 + 63) / 64 * 64) * 4 + 16;
    int64_t per_cu = (int64_t)((160 * 1024) / lds0);
    const int64_t wave_cap = (ks == 2 ? 16 : 12) / nw;              // waves per CU the register budget allows
    if (per_cu > wave_cap) per_cu = wave_cap;
    if (per_cu < 1) per_cu = 1;
    int64_t n_wgb = cdiv(nt, nw);
    const int64_t cap = (int64_t)256 * per_cu / n_chunks;           // resident workgroups
    if (n_wgb > cap) n_wgb = cap < 8 ? 8 : cap;
    const int64_t n_wgb8 = cdiv(n_wgb, 8) * 8;
    dim3 grid((unsigned)(n_wgb8 * n_chunks));
    const size_t lds = (size_t)2 * ks * 32 * SS_CT * sizeof(float) + (size_t)nw * (((n_off + 1) * 16 + 63) / 64 * 64) * 4 + 16;
#define LAUNCH_TSS(KS_, NO_, NW_, WT_)                                                                              \
    do {                                                                                                            \
        static scn::DeviceOnce attr_set;                                                                               \
        if (attr_set.needed()) {                                                                                            \
            SCN_HIP(hipFuncSetAttribute((const void*)k_conv_tss<KS_, NO_, NW_, WT_>,                                \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                   \
            attr_set.done();                                                                                        \
        }                                                                                                           \
        hipLaunchKernelGGL((k_conv_tss<KS_, NO_, NW_, WT_>), grid, dim3(NW_ * 64), lds, st, X, (long long)n_in, cin, tstab, \
                           tile_mask, perm, (long long)nt, W, bias, residual, relu_mask, Y, (long long)n_out, cout, flags, \
                           n_chunks, (int)n_wgb);                                                                   \
    } while (0)
#define PICK_WT(KS_, NO_, NW_)                                                                                      \
    do { if (wt) LAUNCH_TSS(KS_, NO_, NW_, true); else LAUNCH_TSS(KS_, NO_, NW_, false); } while (0)
#define PICK_NO(KS_, NW_)                                                                                           \
    do { if (n_off == 27) PICK_WT(KS_, 27, NW_); else PICK_WT(KS_, 8, NW_); } while (0)
    if (ks == 2) { if (nw == 16) PICK_NO(2, 16); else if (nw == 8) PICK_NO(2, 8); else PICK_NO(2, 4); }
    else { if (nw == 8) PICK_NO(4, 8); else PICK_NO(4, 4); }
#undef PICK_NO
#undef PICK_WT
#undef LAUNCH_TSS
    SCN_LAUNCH_CHECK();
    *launched = true;
    return SCN_OK;
}
}  // namespace scn
