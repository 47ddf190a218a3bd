// conv_tb: the tile convolution of scn_conv_ts.hip for bf16 STORAGE (BASELINE configs 3-5): features and
// the LDS weight image are bf16, accumulation is fp32 on v_mfma_f32_16x16x32_bf16, outputs are rounded to bf16 once
// (round-to-nearest-even).  The layer's master weights stay fp32; scn_conv_tiles_bf16_pack rounds them ONCE per use into
// a bf16 image laid out exactly as the workgroups' LDS images (slice by slice), so staging is a straight 16-byte copy of
// half the bytes (round 1 transposed fp32 weights into LDS with 2-byte stores 64 bytes apart: 8-16-way bank conflicts,
// 5-12 us of a 28-42 us launch).
//
//   Y[r] = bf16( residual[r] + bias + sum_o in(X[table[o][r]]) . bf16(W[o']) )
//
// Same work decomposition as k_conv_ts (mask-sorted 16-row tiles, LPT tile queue per workgroup, weights of all offsets
// in LDS, K split over workgroups in 32-channel chunks with fp32 slabs), different arithmetic shape:
//   * one MFMA contracts 32 channels for a 16 x 16 block, so a (tile, offset) step is NB MFMAs of 16 cycles for 16 NB
//     output columns (fp32: 16 MFMAs of 32 cycles for 32 columns) -- the matrix pipe is no longer the bound, the gather
//     and issue rate are; the workgroup therefore covers 64 columns when the layer has them (NB = 4);
//   * lane (i, kq) gathers ONE 16-byte piece per step: row tstab[t][o][i], channels kc + 8 kq .. + 7 (A fragment layout
//     of the instruction: A[row l & 15][k = 8 (l >> 4) + j]);
//   * the LDS image is [o][n][k], k contiguous: the B fragment of lane (i, kq) for column block nb is the 16 bytes at
//     ((o CT + 16 nb + i) 32 + 8 kq) -- a wave reads 1 KB contiguous with ds_read_b128.  (Lanes of a 16-lane group are
//     64 bytes apart, so SQ_LDS_BANK_CONFLICT is 0.42-0.46 of the LDS cycles; the piece order 16 kq + i, where 16 lanes
//     read 256 consecutive bytes, was built and measured in round 2c: 23.7 / 29.9 / 28.1 / 29.3 us per launch on the
//     four levels against 23.9 / 29.2 / 27.2 / 29.2 -- the kernel waits for its row gathers, not for LDS.)
//   * input ReLU is one v_pk_max_i16 per register (a negative bf16 is a negative int16).
// C/D: acc[nb][j] = D[row 4 kq + j][MFMA column i of block nb].  The image assigns ACTUAL column n0 + NB i + nb to MFMA
// column i of block nb, so a lane's NB accumulators of a row are NB CONSECUTIVE output columns: one 2 NB-byte store (and
// residual / mask load) per row instead of NB two-byte ones.
// K split over workgroups (Cin > 32 or 64): the partial tiles are added inside the launch by the wave that arrives last
// at a (tile, column chunk) -- the pipelined hand-off of k_conv_ts (scn_conv_ts.hip), fp32 partials, fixed K-chunk order.
#include <stdlib.h>
#include <type_traits>

#include "scn_common.h"

using scn::S;
using scn::cdiv;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define MFMAB(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

#ifndef TB_EXP
#define TB_EXP 0            // developer timing experiments (results invalid): 1 no K-split hand-off at all, 2 publish only,
#endif                      // 3 publish + ticket without the combine
static constexpr int TB_KC = 32;        // channels per K-chunk
static constexpr int TB_T = 16;         // rows per tile
static constexpr int TB_NW = 16;        // waves per workgroup

__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }

template <int NB, int KH, bool FUSED, bool PART>
__global__ __launch_bounds__(TB_NW * 64) __attribute__((amdgpu_waves_per_eu(NB == 2 && KH == 1 && !FUSED ? 8 : 4))) void k_conv_tb(
    const unsigned short* __restrict__ X, long long n_in, int cin, const int* __restrict__ tstab,
    const unsigned* __restrict__ tile_mask, const int* __restrict__ perm, const int* __restrict__ tile_order, int n_off,
    long long nt, const unsigned short* __restrict__ image, const float* __restrict__ bias,
    const unsigned short* __restrict__ residual, const unsigned short* __restrict__ relu_mask,
    unsigned short* __restrict__ Y, float* __restrict__ slabs, long long n_out, int cout, int flags, int n_chunks,
    int n_kc, int* __restrict__ counters, int xbins) {
    constexpr int CT = 16 * NB;
    constexpr int KC = TB_KC * KH;              // channels per K-chunk: 32, or 64 as two 32-channel planes of the image
    constexpr int THREADS = TB_NW * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned short Wb[];     // [n_off][CT][32] bf16, then the counter
    const int tid = threadIdx.x, lane = tid & 63;
    const int chunk = blockIdx.x % n_chunks;
    const int kci = (blockIdx.x / n_chunks) % n_kc;
    const int n0 = chunk * CT, kc = kci * KC;
    const bool relu_in = flags & SCN_F_RELU_IN;
    const bool res_last = flags & SCN_F_RESIDUAL_LAST;

    // ---- tile queue (as in k_conv_ts): the workgroup owns every n_tg-th entry of the LPT order --------------------
    // xbins > 1 (SCN_F_TILE_ORDER_X, round 3d): XCD-local hand-out.  Behind the LPT order sits a second list of the tiles
    // by (spatial bin of their first row, cost descending) with its nine bin starts (scn_tiles_build_x).  Workgroups are
    // dealt to XCDs round-robin by block index, so the slices of tile group tg run on the XCDs (tg n_slices + s) % 8: with
    // xbins = 8 / n_slices groups of XCDs, group tg % xbins takes the tiles of ITS run of bins -- the rows its waves gather
    // (neighbours of the tile's rows) then meet in that group's L2s instead of anywhere on the chip.  Placement only
    // decides the speed; any hand-out gives the same results.
    const int tg = blockIdx.x / (n_chunks * n_kc), n_tg = gridDim.x / (n_chunks * n_kc);
    const int* list = tile_order + tg;
    int lstride = n_tg;
    int n_tiles = (int)((nt - tg + n_tg - 1) / n_tg);
    if (xbins > 1) {
        const int bin = tg % xbins, lw = tg / xbins, nw = (n_tg - bin + xbins - 1) / xbins;
        const int* order_x = tile_order + nt;
        const int* bs = order_x + nt;
        const int s0 = bs[bin * (8 / xbins)], s1 = bs[(bin + 1) * (8 / xbins)];
        list = order_x + s0 + lw;
        lstride = nw;
        n_tiles = s1 - s0 > lw ? (s1 - s0 - lw + nw - 1) / nw : 0;
    }
    int* counter = (int*)(Wb + (size_t)n_off * CT * KC);
    if (tid == 0) *counter = 0;
    __syncthreads();
    auto grab = [&]() -> long long {
        int tl = 0;
        if (lane == 0) tl = atomicAdd(counter, 1);
        tl = __builtin_amdgcn_readfirstlane(tl);
        return tl < n_tiles ? list[(long long)tl * lstride] : -1;
    };
    const int i = lane & 15, kq = lane >> 4;
    long long tile_next = grab();
    unsigned m_next = 0;
    int orow_next[4] = {-1, -1, -1, -1};
    if (tile_next >= 0) {
        m_next = __builtin_amdgcn_readfirstlane(tile_mask[tile_next]);
#pragma unroll
        for (int j = 0; j < 4; ++j) orow_next[j] = perm[tile_next * TB_T + 4 * kq + j];
    }

    // ---- stage the weight slice: the packed image holds this workgroup's LDS image verbatim -------------------------
    {
        // image: [o][plane h = k / 32][n][k % 32]: the B fragment reads of a wave for one (column block, plane) are 1 KB
        // contiguous (conflict-free ds_read_b128) for either K-chunk size
        const int total16 = n_off * KC * CT / 8;                          // 16-byte pieces
        const uint4* src = (const uint4*)image + (size_t)(chunk * n_kc + kci) * total16;
        constexpr int SB = 4;                                             // pieces in flight per thread
        for (int base = 0; base < total16; base += THREADS * SB) {
            uint4 v[SB];
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int e = base + u * THREADS + tid;
                v[u] = make_uint4(0u, 0u, 0u, 0u);
                if (e < total16) v[u] = src[e];
            }
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int e = base + u * THREADS + tid;
                if (e < total16) ((uint4*)Wb)[e] = v[u];
            }
        }
    }
    __syncthreads();

    bool k_ok[KH];                                                  // cin % 8 == 0: a lane's 8 channels are all in or out
#pragma unroll
    for (int hh = 0; hh < KH; ++hh) k_ok[hh] = kc + 32 * hh + 8 * kq + 7 < cin;
    const bool single = n_kc == 1;
    const bool direct = single || FUSED;                            // this launch writes Y itself
    const int ncol = n0 + NB * i;                                   // first of this lane's NB consecutive output columns
    const bool n_ok = ncol < cout;                                  // cout % NB == 0: the group is in or out as a whole
    float bcol[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) bcol[nb] = (direct && bias && ncol + nb < cout) ? bias[ncol + nb] : 0.f;
    float* out_slab = slabs + (long long)kci * n_out * cout;
    const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)slabs, 0, FUSED && !single ? (int)(unsigned)(nt * n_chunks * n_kc * (TB_T * CT * 4)) : 0, 0x00020000);
    // A rows through a raw buffer descriptor that covers X exactly: a row index of -1 (no rule) or a channel group past
    // Cin becomes an out-of-range byte offset and the hardware returns zeros
    const __amdgpu_buffer_rsrc_t xrsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)(unsigned)(n_in * cin * 2), 0x00020000);
    const int row_bytes = cin * 2, lane_boff = (kc + 8 * kq) * 2;
    const unsigned short* wlane = Wb + (size_t)i * TB_KC + 8 * kq;   // + ((o KH + plane) CT + 16 nb) 32
    // Round 3d: the step below is ISSUE-bound (four 16-cycle MFMAs against ~16 vector and ~12 scalar instructions of
    // bookkeeping in the first version; SQ counters in profiles/r3_tb_where_the_time_goes.txt), so every piece of bookkeeping is
    // written for its instruction count:
    //   * row-index loads through a raw buffer descriptor of the tile table with a SCALAR offset (tile base + 64 offset):
    //     no vector address arithmetic (was a 64-bit shift-add pair per load);
    //   * the offset list is popped on the scalar unit (s_ff1_i32_b32 returns -1 for an empty mask, which is the queue's
    //     "none" marker; a finished list re-reads its last offset: olast = max(olast, popped));
    //   * the input ReLU is ONE v_pk_max_i16 per register against a scalar floor (0, or -32768 = identity), not a max
    //     plus a select on the run-time flag;
    //   * the channel-group guard of the gather address exists only in the PART instantiation (Cin % 32 != 0).
    const __amdgpu_buffer_rsrc_t trsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)tstab, 0, (int)(unsigned)(nt * n_off * (TB_T * 4)), 0x00020000);
    const int lane_i4 = i * 4;
    const short relu_floor = relu_in ? (short)0 : (short)-32768;
    const s16x8 floor8 = {relu_floor, relu_floor, relu_floor, relu_floor, relu_floor, relu_floor, relu_floor, relu_floor};

#define TB_GATHER(IDX, A)                                                                             \
    do {                                                                                             \
        const int off_ = __mul24((IDX), row_bytes) + lane_boff;                                      \
        _Pragma("unroll") for (int hh_ = 0; hh_ < KH; ++hh_) {                                       \
            int oh_ = off_ + 64 * hh_;                                                               \
            if constexpr (PART) oh_ = k_ok[hh_] ? oh_ : (int)0xFFFFFFF0;                             \
            A[hh_] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, oh_, 0, 0);                        \
        }                                                                                            \
    } while (0)
    // pop the lowest offset of the (wave-uniform) mask m: O = its number or -1; olast = the last offset popped
#define TB_POP(O)                                                                                    \
    do {                                                                                             \
        asm volatile("s_ff1_i32_b32 %0, %1" : "=s"(O) : "s"(m));                                     \
        m &= m - 1;                                                                                  \
        olast = max(olast, (O));                                                                     \
    } while (0)
#define TB_INDEX(DST) DST = __builtin_amdgcn_raw_buffer_load_b32(trsrc, lane_i4, tile_boff + olast * (TB_T * 4), 0)

    // one pipeline step: index of the offset five ahead, rows of the offset three ahead, MFMAs on the oldest set
#define TB_STEP(CUR, GSET, IOLD, INEW)                                                               \
    do {                                                                                             \
        int o4_;                                                                                     \
        TB_POP(o4_);                                                                                 \
        TB_INDEX(INEW);                                                                              \
        TB_GATHER(IOLD, GSET);                                                                       \
        _Pragma("unroll") for (int hh_ = 0; hh_ < KH; ++hh_) {                                       \
            const s16x8 a_ = __builtin_elementwise_max(__builtin_bit_cast(s16x8, CUR[hh_]), floor8); \
            const bf16x8 af_ = __builtin_bit_cast(bf16x8, a_);                                       \
            const unsigned short* wo_ = wlane + ((size_t)oq0 * KH + hh_) * (CT * TB_KC);             \
            _Pragma("unroll") for (int nb_ = 0; nb_ < NB; ++nb_) {                                   \
                const bf16x8 bf_ = *(const bf16x8*)(wo_ + nb_ * 16 * TB_KC);                         \
                acc[nb_] = MFMAB(af_, bf_, acc[nb_]);                                                \
            }                                                                                        \
        }                                                                                            \
        oq0 = oq1; oq1 = oq2; oq2 = oq3; oq3 = oq4; oq4 = o4_;                                       \
    } while (0)

    // ---- epilogue pieces ----------------------------------------------------------------------------------------------
    // A lane's NB consecutive bf16 columns of a row travel as NB/2 32-bit words (one 4- or 8-byte access); explicit words
    // and shifts -- element-indexed short vectors made the compiler spill to scratch.  vec_ok: cout % NB == 0 and aligned
    // operands; otherwise every column is its own 2-byte access.
    constexpr int NWD = NB / 2;
    const bool vec_ok = (cout % NB == 0) && ((((uintptr_t)Y | (uintptr_t)residual | (uintptr_t)relu_mask) & (2 * NB - 1)) == 0);
    auto tb_load_ops = [&](const int (&orow)[4], unsigned (&rw)[4][NWD], unsigned (&mw)[4][NWD]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {                                // all operand loads first, one wait
            const long long e = (long long)(orow[j] < 0 ? 0 : orow[j]) * cout + ncol;
#pragma unroll
            for (int w = 0; w < NWD; ++w) { rw[j][w] = 0u; mw[j][w] = 0x3f803f80u; }      // residual 0, mask 1.0
            if (vec_ok && n_ok) {
                if (residual) {
                    if (NB == 2) rw[j][0] = *(const unsigned*)(residual + e);
                    else { const uint2 t = *(const uint2*)(residual + e); rw[j][0] = t.x; rw[j][NWD - 1] = t.y; }
                }
                if (relu_mask) {
                    if (NB == 2) mw[j][0] = *(const unsigned*)(relu_mask + e);
                    else { const uint2 t = *(const uint2*)(relu_mask + e); mw[j][0] = t.x; mw[j][NWD - 1] = t.y; }
                }
            } else if (!vec_ok) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    if (ncol + nb >= cout) continue;
                    if (residual) rw[j][nb >> 1] = (nb & 1) ? (rw[j][nb >> 1] & 0xffffu) | ((unsigned)residual[e + nb] << 16)
                                                            : (rw[j][nb >> 1] & 0xffff0000u) | residual[e + nb];
                    if (relu_mask) mw[j][nb >> 1] = (nb & 1) ? (mw[j][nb >> 1] & 0xffffu) | ((unsigned)relu_mask[e + nb] << 16)
                                                             : (mw[j][nb >> 1] & 0xffff0000u) | relu_mask[e + nb];
                }
            }
        }
    };
    auto tb_store = [&](const int (&orow)[4], const f32x4 (&acc)[NB], const unsigned (&rw)[4][NWD], const unsigned (&mw)[4][NWD]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (orow[j] < 0) continue;
            unsigned ow[NWD];
#pragma unroll
            for (int w = 0; w < NWD; ++w) ow[w] = 0u;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const unsigned rbits = (nb & 1) ? (rw[j][nb >> 1] & 0xffff0000u) : (rw[j][nb >> 1] << 16);
                const unsigned mbits = (nb & 1) ? (mw[j][nb >> 1] & 0xffff0000u) : (mw[j][nb >> 1] << 16);
                const float r = __uint_as_float(rbits);
                float y = acc[nb][j] + (res_last ? 0.f : r);
                if (!(__uint_as_float(mbits) > 0.f)) y = 0.f;
                if (res_last) y += r;
                ow[nb >> 1] |= (unsigned)f32_to_bf16(y) << (16 * (nb & 1));
            }
            unsigned short* yp = Y + (long long)orow[j] * cout + ncol;
            if (vec_ok) {
                if (n_ok) {
                    if (NB == 2) *(unsigned*)yp = ow[0];
                    else *(uint2*)yp = make_uint2(ow[0], ow[NWD - 1]);
                }
            } else {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    if (ncol + nb < cout) yp[nb] = (unsigned short)(ow[nb >> 1] >> (16 * (nb & 1)));
            }
        }
    };
    auto tb_write = [&](const int (&orow)[4], const f32x4 (&acc)[NB]) {
        unsigned rw[4][NWD], mw[4][NWD];
        tb_load_ops(orow, rw, mw);
        tb_store(orow, acc, rw, mw);
    };
    // In-launch K reduction, pipelined over the wave's tiles (see k_conv_ts).  The combiner's loads are ONE round trip where
    // it can be: the output rows are kept from the tile's own pass (not re-read through `perm`), the residual / ReLU-mask
    // operands are requested together with the partial tiles, and R K-chunks of partials are in flight per round -- R = 2
    // between tiles (the gather pipeline holds its registers), R = 4 in the two flush rounds behind the last tile, where the
    // combine is a serial chain at the END of the launch (tools/ablate_tb_locality.py with -DTB_EXP: the combines cost
    // 7-10 us of a 28-30 us launch at levels 1-3 of the cfg-2 scene before this).
    long long q1 = -1, q2 = -1;
    int tk_v = 0;
    int orow_q1[4] = {-1, -1, -1, -1}, orow_q2[4] = {-1, -1, -1, -1};
    auto tb_publish = [&](long long t, const f32x4 (&acc)[NB]) {
        const int sb = (int)((t * n_chunks + chunk) * n_kc + kci) * (TB_T * CT * 4) + lane * 16;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, acc[nb]),
                                                   srsrc, sb + nb * 1024, 0, 16);         // aux 16 = sc1: write-through
        q1 = t;
    };
    auto tb_retire = [&](auto round_c) {
        constexpr int R = decltype(round_c)::value;
        // the ticket of q2 has returned (callers drained the wave); q1's ticket goes out FIRST, so that its round trip runs
        // beside the combine of q2 instead of behind it
        const int ticket = __builtin_amdgcn_readfirstlane(tk_v);
        if (q1 >= 0 && lane == 0)
            tk_v = __hip_atomic_fetch_add(counters + (int)(q1 * n_chunks + chunk), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (q2 >= 0) {
            if (ticket == n_kc - 1) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                    // compiler ordering only
                const int unit = (int)(q2 * n_chunks + chunk);
                if (lane == 0) __hip_atomic_store(counters + unit, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (TB_EXP == 3) goto tb_no_combine;
                int orow2[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) orow2[j] = orow_q2[j];
                unsigned rw[4][NWD], mw[4][NWD];
                tb_load_ops(orow2, rw, mw);
                const int sb = (unit * n_kc) * (TB_T * CT * 4) + lane * 16;
                f32x4 y[NB];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) y[nb] = (f32x4){bcol[nb], bcol[nb], bcol[nb], bcol[nb]};
                for (int k0 = 0; k0 < n_kc; k0 += R) {            // bias + slab[0] + slab[1] + ...: ascending K-chunks
                    f32x4 p[R][NB];
#pragma unroll
                    for (int u = 0; u < R; ++u) {
                        const int k = k0 + u < n_kc ? k0 + u : n_kc - 1;
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            p[u][nb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                srsrc, sb + k * (TB_T * CT * 4) + nb * 1024, 0, 16));
                    }
#pragma unroll
                    for (int u = 0; u < R; ++u) {
                        if (k0 + u < n_kc) {
#pragma unroll
                            for (int nb = 0; nb < NB; ++nb) y[nb] += p[u][nb];
                        }
                    }
                }
                tb_store(orow2, y, rw, mw);
            }
        }
    tb_no_combine:
        q2 = q1;
        q1 = -1;
#pragma unroll
        for (int j = 0; j < 4; ++j) orow_q2[j] = orow_q1[j];
    };
    using R2 = std::integral_constant<int, 2>;
    using R4 = std::integral_constant<int, 4>;

    while (tile_next >= 0) {
        const long long tile = tile_next;
        unsigned m = m_next;
        int orow[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) orow[j] = orow_next[j];
        const int tile_boff = (int)tile * n_off * (TB_T * 4);       // byte offset of the tile's table block (< 2^31: host)
        const int n_steps = __popc(m);
        tile_next = grab();
        if (tile_next >= 0) {
            m_next = __builtin_amdgcn_readfirstlane(tile_mask[tile_next]);
#pragma unroll
            for (int j = 0; j < 4; ++j) orow_next[j] = perm[tile_next * TB_T + 4 * kq + j];
        }
        int oq0, oq1, oq2, oq3, oq4, iq0, iq1, iq2, iqa, iqb, iqc, iqd;
        int olast = 0;
        TB_POP(oq0); TB_INDEX(iq0);
        TB_POP(oq1); TB_INDEX(iq1);
        TB_POP(oq2); TB_INDEX(iq2);
        TB_POP(oq3); TB_INDEX(iqa);
        TB_POP(oq4); TB_INDEX(iqb);
        __builtin_amdgcn_sched_barrier(0);
        i32x4 s0[KH], s1[KH], s2[KH], s3[KH];
        TB_GATHER(iq0, s0);
        TB_GATHER(iq1, s1);
        TB_GATHER(iq2, s2);
        __builtin_amdgcn_sched_barrier(0);

        f32x4 acc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const float b0 = single ? bcol[nb] : 0.f;               // K split: the combiner adds the bias first
            acc[nb] = (f32x4){b0, b0, b0, b0};
        }
        int n_left = n_steps;
        for (; n_left >= 4; n_left -= 4) {
            TB_STEP(s0, s3, iqa, iqc);
            TB_STEP(s1, s0, iqb, iqd);
            TB_STEP(s2, s1, iqc, iqa);
            TB_STEP(s3, s2, iqd, iqb);
        }
        if (n_left >= 1) { TB_STEP(s0, s3, iqa, iqc); }
        if (n_left >= 2) { TB_STEP(s1, s0, iqb, iqd); }
        if (n_left >= 3) { TB_STEP(s2, s1, iqc, iqa); }

        // ---- tile epilogue -------------------------------------------------------------------------------------
        if (FUSED && !single) {             // pipelined hand-off, see k_conv_ts
            if (TB_EXP == 1) { if (acc[0][0] == 123.456f) tb_publish(tile, acc); continue; }
            if (TB_EXP == 2) { tb_publish(tile, acc); continue; }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            tb_retire(R2{});
            tb_publish(tile, acc);
#pragma unroll
            for (int j = 0; j < 4; ++j) orow_q1[j] = orow[j];
            continue;
        }
        if (direct) tb_write(orow, acc);
        else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (orow[j] < 0) continue;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    if (ncol + nb < cout) out_slab[(long long)orow[j] * cout + ncol + nb] = acc[nb][j];
            }
        }
    }
    if (FUSED && !single) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tb_retire(R4{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tb_retire(R4{});
    }
}
#undef TB_STEP
#undef TB_INDEX
#undef TB_POP
#undef TB_GATHER



// ---------------------------------------------------------------------------------------------------------------------
// k_conv_tbs (round 4): the same convolution OFFSET-OUTER with the weights STREAMED, for layers whose weights do not fit one
// workgroup's LDS (Cin >= 64: levels >= 1 of the U-Net, 48 of the 62 tile launches of a cfg-2 step).
//
// k_conv_tb keeps the weights of ALL offsets of one 32-channel K-chunk resident (110 KB) and therefore splits K over
// workgroups: every (tile, column chunk) is computed as n_kc partial tiles that travel through memory (write-through fp32,
// tickets, a combine by the last arriver) -- 30-45 % of a launch at levels 1-3 of the 150 k scene, a serial chain of trips to
// memory behind every wave's last tile, on top of a 110 KB staging front.  Here a workgroup
// owns NW tiles (one per wave) x 64 output columns with the FULL K: the offsets run in the outer loop, the 64-column slice of
// W[o] (Cin x 64 bf16: 8 / 16 / 32 KB at Cin = 64 / 128 / 256) streams through a double-buffered LDS stage -- requested
// when offset o - 1 starts, written when its MFMAs have been issued, one workgroup barrier per offset --, a wave's 16 x 64
// accumulators stay in registers across all offsets, and a row is gathered as KS 16-byte pieces per lane, D offsets ahead.
// No K split, no partial tiles, no tickets, no 110 KB front, one write of Y.  Offsets a tile lacks still take their barrier
// and their (zero-returning, out-of-range) gather instructions but skip their MFMAs (wave-uniform branch without loads).
//
// Reads the SAME packed weight image as k_conv_tb (piece (chunk, K-chunk ks, offset o) = 4 KB) and the same tile structures;
// the fp32 summation order differs (k_conv_tb: offsets inside a K-chunk, then the K-chunks; here: K-chunks inside an offset),
// so results agree with k_conv_tb's to fp32 rounding before the one bf16 rounding (tests/test_gpu_exec.py) -- both are
// deterministic, neither depends on placement.  SCN_TB_STREAM=0 keeps k_conv_tb everywhere.
// ---------------------------------------------------------------------------------------------------------------------
#ifndef TBS_EXP
#define TBS_EXP 0           // developer timing experiments (results invalid): 1 no weight streaming behind slice 0, 2 no row
#endif                      // gathers behind the first D, 3 no barriers in the offset loop, 4 no MFMAs
template <int KS, int N_OFF, int NW, int D>
__global__ __launch_bounds__(NW * 64) void k_conv_tbs(
    const unsigned short* __restrict__ X, long long n_in, int cin, const int* __restrict__ tstab,
    const unsigned* __restrict__ tile_mask, const int* __restrict__ perm, const int* __restrict__ tile_order, long long nt,
    const unsigned short* __restrict__ image, const float* __restrict__ bias, const unsigned short* __restrict__ residual,
    const unsigned short* __restrict__ relu_mask, unsigned short* __restrict__ Y, long long n_out, int cout, int flags,
    int n_chunks, int n_wgb) {
    constexpr int NB = 4, CT = 64, THREADS = NW * 64;
    constexpr int PIECES = KS * CT * 4;                             // 16-byte pieces of one offset's slice
    constexpr int PW = (PIECES + THREADS - 1) / THREADS;            // ... per thread
    extern __shared__ __attribute__((aligned(16))) unsigned short Wb[];     // [2][KS][CT][32] bf16, then int idx[NW][IDXW]
    constexpr int IDXW = (((N_OFF + 1) * TB_T + 63) / 64) * 64;     // row indices of a wave's tile, all offsets + a row of -1
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int i = lane & 15, kq = lane >> 4;
    int* idx_w = (int*)(Wb + (size_t)2 * PIECES * 8) + w * IDXW;
    // workgroups are dealt to XCDs round-robin by block index: the n_chunks column chunks of a tile batch (which gather the
    // same rows) get block indices 8 apart, i.e. one XCD and one L2
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int chunk = jb % n_chunks, wgb = (jb / n_chunks) * 8 + xcd;
    if (wgb >= n_wgb) return;
    const int n0 = chunk * CT;
    const bool relu_in = flags & SCN_F_RELU_IN;
    const bool res_last = flags & SCN_F_RESIDUAL_LAST;
    const int ncol = n0 + NB * i;
    const bool n_ok = ncol < cout;
    float bcol[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) bcol[nb] = (bias && ncol + nb < cout) ? bias[ncol + nb] : 0.f;
    const __amdgpu_buffer_rsrc_t xrsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)(unsigned)(n_in * cin * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t trsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)tstab, 0, (int)(unsigned)(nt * N_OFF * (TB_T * 4)), 0x00020000);
    // the slice of offset o: K-chunk ks of this column chunk is the 4 KB piece ((chunk KS + ks) N_OFF + o) of the image; a
    // thread's pieces p = u THREADS + tid sit at a fixed vector offset, the offset advances the SCALAR offset
    const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(image + (size_t)chunk * KS * N_OFF * (CT * TB_KC)), 0, (int)(unsigned)(KS * N_OFF * (CT * TB_KC * 2)), 0x00020000);
    int wvoff[PW];
#pragma unroll
    for (int u = 0; u < PW; ++u) {
        const int p = u * THREADS + tid;
        wvoff[u] = p < PIECES ? (p / (CT * 4)) * (N_OFF * CT * TB_KC * 2) + (p % (CT * 4)) * 16 : (int)0x7FFFFFF0;
    }
    const int row_bytes = cin * 2, lane_boff = 16 * kq;
    const short relu_floor = relu_in ? (short)0 : (short)-32768;
    const s16x8 floor8 = {relu_floor, relu_floor, relu_floor, relu_floor, relu_floor, relu_floor, relu_floor, relu_floor};
    // LDS image of a slice: the B fragment of (K-chunk ks, column block nb) is 1 KB in LANE order (lane = 16 kq + i holds
    // column 16 nb + i, channels 8 kq .. + 7): one conflict-free ds_read_b128 per MFMA.  The packed image stores a 4 KB piece as
    // [n][kq] (k_conv_tb's order: 16 lanes 64 bytes apart, 4-way bank conflicts -- there the kernel waits for its gathers,
    // here the B reads are the inner loop); the transposition [n = 16 nb + i][kq] -> [nb][kq][i] happens in the staging write.
    const unsigned short* wlane = Wb + (size_t)lane * 8;             // + buf PIECES 8 + (ks 4 + nb) 512
    int wdst[PW];
#pragma unroll
    for (int u = 0; u < PW; ++u) {
        const int p = u * THREADS + tid, r = p % (CT * 4);           // r = n 4 + kq inside the plane
        wdst[u] = (p / (CT * 4)) * (CT * 4) + (r >> 6) * 64 + (r & 3) * 16 + ((r >> 2) & 15);
    }
    constexpr int NWD = NB / 2;
    const bool vec_ok = (cout % NB == 0) && ((((uintptr_t)Y | (uintptr_t)residual | (uintptr_t)relu_mask) & (2 * NB - 1)) == 0);

#define TBS_WLOAD(O, V)                                                                                        \
    _Pragma("unroll") for (int u_ = 0; u_ < PW; ++u_)                                                          \
        V[u_] = __builtin_amdgcn_raw_buffer_load_b128(irsrc, wvoff[u_], (O) * (CT * TB_KC * 2), 0)
#define TBS_WSTORE(BUF, V)                                                                                     \
    _Pragma("unroll") for (int u_ = 0; u_ < PW; ++u_) {                                                        \
        const int p_ = u_ * THREADS + tid;                                                                     \
        if (p_ < PIECES) ((i32x4*)Wb)[(size_t)(BUF) * PIECES + wdst[u_]] = V[u_];                              \
    }
#define TBS_GATHER(O, SLOT)                                                                                    \
    do {                                                                                                       \
        const int off_ = __mul24(idx_w[(O) * TB_T + i], row_bytes) + lane_boff;                                \
        _Pragma("unroll") for (int ks_ = 0; ks_ < KS; ++ks_)                                                   \
            A[SLOT][ks_] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, off_ + 64 * ks_, 0, 0);                \
    } while (0)

    unsigned* um_s = (unsigned*)((int*)(Wb + (size_t)2 * PIECES * 8) + NW * IDXW);      // union of the batch's tile masks
    for (long long bt = wgb; bt * NW < nt; bt += n_wgb) {
        const long long tl = bt * NW + w;
        // consecutive tile IDS (not the LPT order): tiles are cut from rows sorted by offset mask, so the NW tiles of a batch
        // have nearly the same offsets -- the batch walks only the UNION of their masks
        const long long tile = tl < nt ? tl : -1;                   // (wave-uniform)
        unsigned m = 0;
        int orow[4] = {-1, -1, -1, -1};
        if (tile >= 0) {
            m = __builtin_amdgcn_readfirstlane(tile_mask[tile]);
#pragma unroll
            for (int j = 0; j < 4; ++j) orow[j] = perm[tile * TB_T + 4 * kq + j];
        }
        if (tid == 0) *um_s = 0u;
        // row indices of all offsets -> this wave's LDS strip (a tile's block of the table is N_OFF x 64 contiguous bytes: one
        // coalesced 256-byte load per 4 offsets); row N_OFF of the strip is -1: the "no offset left" slot of the prefetch.
        // A wave without a tile gets -1 everywhere: no gather traffic.
        {
            const int tile_boff = tile >= 0 ? (int)tile * N_OFF * (TB_T * 4) : 0;
            int v[IDXW / 64];
#pragma unroll
            for (int t = 0; t < IDXW / 64; ++t)
                v[t] = (tile >= 0 && t * 64 + lane < N_OFF * TB_T)
                           ? __builtin_amdgcn_raw_buffer_load_b32(trsrc, (t * 64 + lane) * 4, tile_boff, 0) : -1;
#pragma unroll
            for (int t = 0; t < IDXW / 64; ++t) idx_w[t * 64 + lane] = v[t];
        }
        __syncthreads();
        if (lane == 0 && m) atomicOr(um_s, m);
        __syncthreads();
        unsigned ua = __builtin_amdgcn_readfirstlane(*um_s), uc = ua;     // offsets left to request / to compute
        const int n_steps = __popc(ua);
        int olast = N_OFF;
        // pop the lowest offset of the (workgroup-uniform) mask U: its number, or N_OFF (the all-"-1" row) when none is left
#define TBS_POP(U, O)                                                                                          \
    do {                                                                                                       \
        int f_;                                                                                                \
        asm volatile("s_ff1_i32_b32 %0, %1" : "=s"(f_) : "s"(U));                                              \
        (O) = f_ < 0 ? N_OFF : f_;                                                                             \
        (U) &= (U) - 1u;                                                                                       \
    } while (0)
        // ---- the first D offsets: weight slice -> register set d, rows -> A[d].  The slices travel D steps ahead like the
        // rows, and in the SAME order: loads return in order, so waiting for the next slice must not wait for younger gathers
        i32x4 A[D][KS], wv[D][PW];
        int oq[D];                                                  // offset of the step whose operands sit in set d
#pragma unroll
        for (int d = 0; d < D; ++d) {
            TBS_POP(ua, oq[d]);
            TBS_WLOAD(oq[d], wv[d]);
            TBS_GATHER(oq[d], d);
        }
        (void)olast;
        TBS_WSTORE(0, wv[0]);                 // (buffer 0: the barrier above freed it)
        f32x4 acc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = (f32x4){bcol[nb], bcol[nb], bcol[nb], bcol[nb]};

        // one step: barrier; MFMAs of this step's offset if the wave's tile has it; the next step's slice -> LDS; the sets just
        // freed take slice and rows of the step D ahead.  SLOT = step % D, BUF = step & 1 (D is even).
#define TBS_STEP(SLOT, BUF)                                                                                    \
    do {                                                                                                       \
        if (TBS_EXP != 3) __syncthreads();                                                                     \
        const int oc_ = oq[SLOT];                                                                              \
        if (TBS_EXP != 4 && ((m >> oc_) & 1u)) {                                                               \
            const unsigned short* wo = wlane + (size_t)(BUF) * (PIECES * 8);                                   \
            _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                                                \
                const s16x8 a_ = __builtin_elementwise_max(__builtin_bit_cast(s16x8, A[SLOT][ks]), floor8);    \
                const bf16x8 af = __builtin_bit_cast(bf16x8, a_);                                              \
                _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) {                                            \
                    const bf16x8 bf = *(const bf16x8*)(wo + (ks * 4 + nb) * 512);                              \
                    acc[nb] = MFMAB(af, bf, acc[nb]);                                                          \
                }                                                                                              \
            }                                                                                                  \
        }                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        if (TBS_EXP != 1) TBS_WSTORE(1 - (BUF), wv[((SLOT) + 1) % D]);                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        TBS_POP(ua, oq[SLOT]);                                                                                 \
        if (TBS_EXP != 1) TBS_WLOAD(oq[SLOT], wv[SLOT]);                                                       \
        if (TBS_EXP != 2) TBS_GATHER(oq[SLOT], SLOT);                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
    } while (0)
        static_assert(D == 2 || D == 4, "the step macro alternates the LDS buffer with the slot parity");
        int n_left = n_steps;
        (void)uc;
        for (; n_left >= D; n_left -= D) {
            TBS_STEP(0, 0);
            TBS_STEP(1, 1);
            if constexpr (D == 4) { TBS_STEP(2, 0); TBS_STEP(3, 1); }
        }
        if (n_left >= 1) TBS_STEP(0, 0);
        if constexpr (D == 4) {
            if (n_left >= 2) TBS_STEP(1, 1);
            if (n_left >= 3) TBS_STEP(2, 0);
        }
#undef TBS_STEP
#undef TBS_POP
        // ---- epilogue: residual / ReLU-backward mask, one bf16 rounding, one write (as k_conv_tb's direct form)
        if (tile >= 0) {
            unsigned rw[4][NWD], mw[4][NWD];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long long e = (long long)(orow[j] < 0 ? 0 : orow[j]) * cout + ncol;
#pragma unroll
                for (int ww = 0; ww < NWD; ++ww) { rw[j][ww] = 0u; mw[j][ww] = 0x3f803f80u; }
                if (vec_ok && n_ok) {
                    if (residual) { const uint2 t = *(const uint2*)(residual + e); rw[j][0] = t.x; rw[j][1] = t.y; }
                    if (relu_mask) { const uint2 t = *(const uint2*)(relu_mask + e); mw[j][0] = t.x; mw[j][1] = t.y; }
                } else if (!vec_ok) {
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        if (ncol + nb >= cout) continue;
                        if (residual) rw[j][nb >> 1] = (nb & 1) ? (rw[j][nb >> 1] & 0xffffu) | ((unsigned)residual[e + nb] << 16)
                                                                : (rw[j][nb >> 1] & 0xffff0000u) | residual[e + nb];
                        if (relu_mask) mw[j][nb >> 1] = (nb & 1) ? (mw[j][nb >> 1] & 0xffffu) | ((unsigned)relu_mask[e + nb] << 16)
                                                                 : (mw[j][nb >> 1] & 0xffff0000u) | relu_mask[e + nb];
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (orow[j] < 0) continue;
                unsigned ow[NWD] = {0u, 0u};
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const unsigned rbits = (nb & 1) ? (rw[j][nb >> 1] & 0xffff0000u) : (rw[j][nb >> 1] << 16);
                    const unsigned mbits = (nb & 1) ? (mw[j][nb >> 1] & 0xffff0000u) : (mw[j][nb >> 1] << 16);
                    const float r = __uint_as_float(rbits);
                    float y = acc[nb][j] + (res_last ? 0.f : r);
                    if (!(__uint_as_float(mbits) > 0.f)) y = 0.f;
                    if (res_last) y += r;
                    ow[nb >> 1] |= (unsigned)f32_to_bf16(y) << (16 * (nb & 1));
                }
                unsigned short* yp = Y + (long long)orow[j] * cout + ncol;
                if (vec_ok) {
                    if (n_ok) *(uint2*)yp = make_uint2(ow[0], ow[1]);
                } else {
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        if (ncol + nb < cout) yp[nb] = (unsigned short)(ow[nb >> 1] >> (16 * (nb & 1)));
                }
            }
        }
        __syncthreads();                     // the next batch restages buffer 0 (read last by offset N_OFF - 1 when that is even)
    }
#undef TBS_GATHER
#undef TBS_WSTORE
#undef TBS_WLOAD
}

// Y = bf16( bias + sum_kc slab[kc] (+ residual, ReLU-backward mask) ), K-chunks added in ascending order.
// V = 4: 16-byte slab reads, 8-byte bf16 accesses (cout % 4 == 0, aligned buffers).
template <int V>
__global__ void k_conv_tb_sum(const float* __restrict__ slabs, int n_kc, long long n_out, int cout,
                              const float* __restrict__ bias, const unsigned short* __restrict__ residual,
                              const unsigned short* __restrict__ relu_mask, unsigned short* __restrict__ Y,
                              int res_last) {
    typedef float vf_t __attribute__((ext_vector_type(V)));
    typedef unsigned short vh_t __attribute__((ext_vector_type(V)));
    const long long total = n_out * cout / V;
    for (long long g = blockIdx.x * (long long)blockDim.x + threadIdx.x; g < total;
         g += (long long)gridDim.x * blockDim.x) {
        const long long e = g * V;
        vf_t y;
        if (bias) y = *(const vf_t*)(bias + e % cout);
        else {
#pragma unroll
            for (int v = 0; v < V; ++v) y[v] = 0.f;
        }
        for (int k = 0; k < n_kc; ++k) y += *(const vf_t*)(slabs + (long long)k * n_out * cout + e);
        vh_t rh, mh;
        if (residual) rh = *(const vh_t*)(residual + e);
        if (relu_mask) mh = *(const vh_t*)(relu_mask + e);
        vh_t out;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const float r = residual ? bf16_to_f32(rh[v]) : 0.f;
            float t = y[v] + (res_last ? 0.f : r);
            if (relu_mask && !(bf16_to_f32(mh[v]) > 0.f)) t = 0.f;
            if (res_last) t += r;
            out[v] = f32_to_bf16(t);
        }
        *(vh_t*)(Y + e) = out;
    }
}

// ---- launch shape, shared by the pack and the convolution ---------------------------------------------------------------
namespace {
struct TbShape { int nb, kh, ct, kc, n_chunks, n_kc; };

TbShape tb_shape(int cin, int cout) {
    const int nb_env = (int)scn::sw(scn::SW_TB_NB).i;      // developer switches
    const int kh_env = (int)scn::sw(scn::SW_TB_KH).i;
    // shapes: 32 columns x 32 channels (small layers), 64 columns x 32 channels, 32 columns x 64 channels -- the last
    // halves the number of K-chunks (no partial sums at Cin = 64) at twice the gather traffic per output
    TbShape s;
    s.kh = (kh_env == 1 || kh_env == 2) ? kh_env : 1;
    if (cin <= 32) s.kh = 1;
    s.nb = (nb_env == 2 || nb_env == 4) ? nb_env : (cout > 32 ? 4 : 2);
    if (s.kh == 2) s.nb = 2;
    s.ct = 16 * s.nb;
    s.kc = TB_KC * s.kh;
    s.n_chunks = (int)cdiv(cout, s.ct);
    s.n_kc = (int)cdiv(cin, s.kc);
    return s;
}
}  // namespace

// image[slice = chunk * n_kc + kci][o][plane h][n = 16 nb + i][kk] = bf16( W[o'][k = kc + 32 h + kk][col = n0 + NB i + nb] )
// (zero outside the layer).  WT: the layer stores [o][col][k] (backward-data); REV: o' = n_off - 1 - o.
//
// Round 4: the pack runs once per step over every layer's master weights (12.4 M fp32 -> two bf16 images each: 50 MB read,
// 50 MB written at cfg 2) and took 87 us -- one 8-element piece per thread meant 64-bit div/mod chains, a per-THREAD job
// search and eight 4-byte loads a cout apart.  Now a UNIT = one (slice, offset, plane) = ct columns x 32 channels is packed
// by ct lanes of a wave, lane = column: the unit's indices are wave-uniform (scalar), the [k][col] source is read as 32
// coalesced rows (ct x 4 B each), the [col][k] source as one 128-byte run per lane, and a lane writes its column's four
// pieces as one 64-byte line.  Same rounding, same image, bit for bit.
struct TbPackJob { const float* W; unsigned short* image; int cin, cout, n_off, wt, rev; TbShape sh; };

__device__ __forceinline__ void tb_pack_unit(const TbPackJob& jb, int unit, int cl) {
    const TbShape& sh = jb.sh;
    int r = unit;                                                    // unit = ((chunk * n_kc + kci) * n_off + o) * kh + h
    const int h = r % sh.kh; r /= sh.kh;
    const int o = r % jb.n_off; r /= jb.n_off;
    const int kci = r % sh.n_kc, chunk = r / sh.n_kc;
    const int wo = jb.rev ? jb.n_off - 1 - o : o;
    const int k0 = kci * sh.kc + 32 * h;
    // lane cl = local column: image position n = 16 nb + i holds column NB i + nb
    const int col = chunk * sh.ct + cl;
    const int n = 16 * (cl % sh.nb) + cl / sh.nb;
    float w[32];
    const bool col_ok = col < jb.cout;
    if (jb.wt) {                                                     // [o][col][k]: 32 consecutive floats
        const float* src = jb.W + ((long long)wo * jb.cout + (col_ok ? col : 0)) * jb.cin + k0;
        if (col_ok && k0 + 32 <= jb.cin && (jb.cin & 3) == 0 && (((uintptr_t)jb.W) & 15) == 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float4 t = ((const float4*)src)[q];
                w[4 * q] = t.x; w[4 * q + 1] = t.y; w[4 * q + 2] = t.z; w[4 * q + 3] = t.w;
            }
        } else {
#pragma unroll
            for (int u = 0; u < 32; ++u) w[u] = (col_ok && k0 + u < jb.cin) ? src[u] : 0.f;
        }
    } else {                                                         // [o][k][col]: a row of ct consecutive floats per k
        const float* src = jb.W + ((long long)wo * jb.cin + k0) * jb.cout + (col_ok ? col : 0);
        if (k0 + 32 <= jb.cin) {
#pragma unroll
            for (int u = 0; u < 32; ++u) w[u] = col_ok ? src[(long long)u * jb.cout] : 0.f;
        } else {
#pragma unroll
            for (int u = 0; u < 32; ++u) w[u] = (col_ok && k0 + u < jb.cin) ? src[(long long)u * jb.cout] : 0.f;
        }
    }
    uint4* dst = (uint4*)jb.image + ((long long)unit * sh.ct + n) * 4;
#pragma unroll
    for (int k8 = 0; k8 < 4; ++k8) {
        uint4 pk;
        pk.x = f32_to_bf16(w[8 * k8 + 0]) | ((unsigned)f32_to_bf16(w[8 * k8 + 1]) << 16);
        pk.y = f32_to_bf16(w[8 * k8 + 2]) | ((unsigned)f32_to_bf16(w[8 * k8 + 3]) << 16);
        pk.z = f32_to_bf16(w[8 * k8 + 4]) | ((unsigned)f32_to_bf16(w[8 * k8 + 5]) << 16);
        pk.w = f32_to_bf16(w[8 * k8 + 6]) | ((unsigned)f32_to_bf16(w[8 * k8 + 7]) << 16);
        dst[k8] = pk;
    }
}

// units per wave: 64 / ct (ct = 32 or 64); a wave's units belong to one layer
__global__ void __launch_bounds__(256) k_tb_pack(TbPackJob jb, int n_units) {
    const int upw = 64 / jb.sh.ct;
    const int wave = (int)((blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63;
    const int unit = wave * upw + lane / jb.sh.ct;
    if (unit < n_units) tb_pack_unit(jb, unit, lane % jb.sh.ct);
}

// Many layers in ONE launch (a U-Net packs the forward and backward-data images of all its convolutions once per step:
// 62 separate pack launches cost 0.37 ms of a 5 ms step).  Descriptors travel by value in the kernel arguments;
// start[j] = first WAVE of job j.
#define SCN_PACK_MAX 80
struct TbPackJobs { TbPackJob job[SCN_PACK_MAX]; int start[SCN_PACK_MAX + 1]; int n_units[SCN_PACK_MAX]; int n; };

__global__ void __launch_bounds__(256) k_tb_pack_many(TbPackJobs jobs) {
    const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6));
    if (wave >= jobs.start[jobs.n]) return;
    int lo = 0, hi = jobs.n - 1;                          // job of this wave: last start <= wave (wave-uniform: scalar loads)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs.start[mid] <= wave) lo = mid; else hi = mid - 1;
    }
    const TbPackJob& jb = jobs.job[lo];
    const int lane = threadIdx.x & 63;
    const int unit = (wave - jobs.start[lo]) * (64 / jb.sh.ct) + lane / jb.sh.ct;
    if (unit < jobs.n_units[lo]) tb_pack_unit(jb, unit, lane % jb.sh.ct);
}

extern "C" int scn_conv_tiles_bf16_pack_many(int n, const float* const* W_host, const int32_t* cin_host,
                                             const int32_t* cout_host, const int32_t* n_off_host,
                                             const int32_t* flags_host, uint16_t* const* image_host, scn_stream_t stream) {
    SCN_REQUIRE(n >= 0 && (n == 0 || (W_host && cin_host && cout_host && n_off_host && flags_host && image_host)));
    for (int base = 0; base < n; base += SCN_PACK_MAX) {
        TbPackJobs jobs;
        jobs.n = n - base < SCN_PACK_MAX ? n - base : SCN_PACK_MAX;
        jobs.start[0] = 0;
        for (int j = 0; j < jobs.n; ++j) {
            const int q = base + j;
            SCN_REQUIRE(W_host[q] && image_host[q] && cin_host[q] >= 1 && cout_host[q] >= 1 && n_off_host[q] >= 1 &&
                        n_off_host[q] <= 27 && (((uintptr_t)image_host[q]) & 15) == 0);
            TbPackJob& jb = jobs.job[j];
            jb.W = W_host[q]; jb.image = image_host[q]; jb.cin = cin_host[q]; jb.cout = cout_host[q]; jb.n_off = n_off_host[q];
            jb.wt = (flags_host[q] & SCN_F_W_TRANSPOSED) ? 1 : 0;
            jb.rev = (flags_host[q] & SCN_F_OFF_REVERSE) ? 1 : 0;
            jb.sh = tb_shape(jb.cin, jb.cout);
            const int64_t units = (int64_t)jb.sh.n_chunks * jb.sh.n_kc * jb.n_off * jb.sh.kh;
            SCN_REQUIRE(units < (1ll << 24));
            jobs.n_units[j] = (int)units;
            jobs.start[j + 1] = jobs.start[j] + (int)cdiv(units, 64 / jb.sh.ct);
        }
        if (jobs.start[jobs.n] == 0) continue;
        hipLaunchKernelGGL(k_tb_pack_many, dim3((unsigned)cdiv(jobs.start[jobs.n], 4)), dim3(256), 0, S(stream), jobs);
        SCN_LAUNCH_CHECK();
    }
    return SCN_OK;
}

extern "C" int64_t scn_conv_tiles_bf16_image_bytes(int cin, int cout, int n_off) {
    if (cin < 1 || cout < 1 || n_off < 1) return -1;
    const TbShape sh = tb_shape(cin, cout);
    return (int64_t)sh.n_chunks * sh.n_kc * n_off * sh.ct * sh.kc * (int64_t)sizeof(uint16_t);
}

extern "C" int scn_conv_tiles_bf16_pack(const float* W, int cin, int cout, int n_off, int flags, uint16_t* image,
                                        scn_stream_t stream) {
    SCN_REQUIRE(W && image && cin >= 1 && cout >= 1 && n_off >= 1 && n_off <= 27);
    SCN_REQUIRE((((uintptr_t)image) & 15) == 0);
    TbPackJob jb;
    jb.W = W; jb.image = image; jb.cin = cin; jb.cout = cout; jb.n_off = n_off;
    jb.wt = (flags & SCN_F_W_TRANSPOSED) ? 1 : 0; jb.rev = (flags & SCN_F_OFF_REVERSE) ? 1 : 0;
    jb.sh = tb_shape(cin, cout);
    const int64_t units = (int64_t)jb.sh.n_chunks * jb.sh.n_kc * n_off * jb.sh.kh;
    SCN_REQUIRE(units < (1ll << 24));
    const int64_t waves = cdiv(units, 64 / jb.sh.ct);
    hipLaunchKernelGGL(k_tb_pack, dim3((unsigned)cdiv(waves, 4)), dim3(256), 0, S(stream), jb, (int)units);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

extern "C" int64_t scn_conv_tiles_bf16_scratch_bytes(int cin, int64_t n_out, int cout) {
    const TbShape sh = tb_shape(cin, cout);
    return 256 + (sh.n_kc > 1 ? (int64_t)sh.n_kc * cdiv(n_out, TB_T) * TB_T * sh.n_chunks * sh.ct * (int64_t)sizeof(float) : 0);
}

extern "C" int64_t scn_conv_tiles_bf16_arrival_counters(int cin, int64_t n_out, int cout) {
    const TbShape sh = tb_shape(cin, cout);
    return sh.n_kc > 1 ? cdiv(n_out, TB_T) * sh.n_chunks : 0;
}

extern "C" int scn_conv_tiles_bf16(const uint16_t* X, int64_t n_in, int cin, const int32_t* tstab,
                                   const uint32_t* tile_mask, const int32_t* perm, const int32_t* tile_order, int n_off,
                                   int64_t n_out, const uint16_t* image, const float* bias, const uint16_t* residual,
                                   const uint16_t* relu_mask, uint16_t* Y, int cout, int flags, void* scratch,
                                   int32_t* arrival, scn_stream_t stream) {
    SCN_REQUIRE(n_off >= 1 && n_off <= 27 && n_out >= 0 && n_in >= 0 && cin >= 8 && cout >= 1);
    SCN_REQUIRE(cin % 8 == 0);                                   // 16-byte row pieces of X (any cout)
    if (n_out == 0) return SCN_OK;
    SCN_REQUIRE(X && tstab && tile_mask && perm && tile_order && image && Y && scratch);
    SCN_REQUIRE((((uintptr_t)X | (uintptr_t)image) & 15) == 0);
    SCN_REQUIRE(n_in < (1ll << 23) && n_in * cin * 2 < (1ll << 32) - (1ll << 24));    // 24-bit rows, 32-bit offsets
    SCN_REQUIRE(n_out < (1ll << 23));                             // the tile table is addressed with 31-bit byte offsets
    const int64_t nt = cdiv(n_out, TB_T);
    const TbShape sh = tb_shape(cin, cout);
    const int nb = sh.nb, kh = sh.kh, ct = sh.ct, kcs = sh.kc, n_chunks = sh.n_chunks, n_kc = sh.n_kc;
    // ---- the offset-outer, weight-streaming kernel (k_conv_tbs above).  Measured on the cfg-2 scene (tools/ablate_tbs_exp.py,
    // profiles/r4_tbs_streaming_experiment.txt): 26.2 / 39.3 / 47.9 us per launch at levels 1 / 2 / 3 against k_conv_tb's
    // 30.0 / 26.9 / 27.7 -- it wins only where one offset's slice is small (Cin = 64), so THAT is what it runs by default:
    // SubM 3^3 layers with two K-chunks.  SCN_TB_STREAM=1: every eligible layer (Cin = 64 / 128 / 256, 3^3 and 2^3 tables);
    // SCN_TB_STREAM=0: none (k_conv_tb everywhere: A/B, cross-check in the tests)
    const scn::SwitchVal ts_sw = scn::sw(scn::SW_TB_STREAM);        // (scn_debug_set: the tests switch it inside one process)
    const int ts_mode = ts_sw.set ? (int)ts_sw.i : -1;
    const bool ts_eligible = n_kc >= 2 && kh == 1 && nb == 4 && cin % 32 == 0 && (n_kc == 2 || n_kc == 4 || n_kc == 8) &&
                             (n_off == 27 || n_off == 8) && !(flags & SCN_F_SPLIT_SUM);
    if (ts_eligible && ts_mode != 0 && (ts_mode == 1 || (n_kc == 2 && n_off == 27))) {
        hipStream_t st = S(stream);
        // waves per workgroup: 8 (one tile each); 4 when the level has too few tiles to put a workgroup on every CU
        const bool small = cdiv(nt, 8) * n_chunks < 224;
        const int nw = small ? 4 : 8;
        int64_t n_wgb = cdiv(nt, nw);
        const int64_t cap = (int64_t)(n_kc == 2 ? 512 : 256) / n_chunks;          // resident workgroups: 2 per CU at Cin = 64
        if (n_wgb > cap) n_wgb = cap < 8 ? 8 : cap;
        const int64_t n_wgb8 = cdiv(n_wgb, 8) * 8;
        dim3 grid((unsigned)(n_wgb8 * n_chunks));
        const size_t lds = (size_t)2 * n_kc * 64 * 32 * sizeof(uint16_t) + (size_t)nw * (((n_off + 1) * 16 + 63) / 64 * 64) * 4 + 16;
#define LAUNCH_TBS(KS_, NO_, NW_, D_)                                                                                  \
    do {                                                                                                            \
        static scn::DeviceOnce attr_set;                                                                               \
        if (attr_set.needed()) {                                                                                            \
            SCN_HIP(hipFuncSetAttribute((const void*)k_conv_tbs<KS_, NO_, NW_, D_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                        160 * 1024));                                                               \
            attr_set.done();                                                                                        \
        }                                                                                                           \
        hipLaunchKernelGGL((k_conv_tbs<KS_, NO_, NW_, D_>), grid, dim3(NW_ * 64), lds, st, X, (long long)n_in, cin, tstab,  \
                           tile_mask, perm, tile_order, (long long)nt, image, bias, residual, relu_mask, Y, (long long)n_out, \
                           cout, flags, n_chunks, (int)n_wgb);                                                      \
    } while (0)
#define PICK_TBS(KS_, D_)                                                                                           \
    do {                                                                                                            \
        if (n_off == 27) { if (nw == 8) LAUNCH_TBS(KS_, 27, 8, D_); else LAUNCH_TBS(KS_, 27, 4, D_); }              \
        else { if (nw == 8) LAUNCH_TBS(KS_, 8, 8, D_); else LAUNCH_TBS(KS_, 8, 4, D_); }                            \
    } while (0)
        if (n_kc == 2) PICK_TBS(2, 4);
        else if (n_kc == 4) PICK_TBS(4, 4);
        else PICK_TBS(8, 2);
#undef PICK_TBS
#undef LAUNCH_TBS
        SCN_LAUNCH_CHECK();
        return SCN_OK;
    }
    float* slabs = (float*)((char*)scratch + 256);
    const size_t lds = (size_t)n_off * ct * kcs * sizeof(uint16_t) + 16;
    int wg_per_cu = (int)((160 * 1024) / lds);
    if (wg_per_cu > 2) wg_per_cu = 2;
    if (wg_per_cu < 1) wg_per_cu = 1;
    int64_t n_tg = ((int64_t)scn::cu_budget() * wg_per_cu) / ((int64_t)n_chunks * n_kc);
    if (n_tg > cdiv(nt, TB_NW)) n_tg = cdiv(nt, TB_NW);
    if (n_tg < 1) n_tg = 1;
    const bool fused = n_kc > 1 && arrival != nullptr && !(flags & SCN_F_SPLIT_SUM) &&
                       nt * n_chunks * n_kc * (int64_t)(TB_T * ct * 4) < (1ll << 31);
    dim3 grid((unsigned)(n_tg * n_chunks * n_kc));
    hipStream_t st = S(stream);
    const bool part = cin % kcs != 0;                              // the last K-chunk has channel groups past Cin
    // XCD-local hand-out (SCN_F_TILE_ORDER_X): possible when the slices of a tile group fall on 8 / n_slices whole groups of
    // XCDs, i.e. 1, 2 or 4 slices; with 8 or more every XCD sees every tile group anyway
    const int n_slices = n_chunks * n_kc;
    const bool x_off = scn::sw(scn::SW_TB_NO_XORDER).set;             // (scn_debug_set: the tests switch it inside one process)
    const int xbins = ((flags & SCN_F_TILE_ORDER_X) && !x_off && (n_slices == 1 || n_slices == 2 || n_slices == 4) &&
                       n_tg >= 8) ? 8 / n_slices : 1;
#define LAUNCH_TB(N, K, FU, PT)                                                                                     \
    do {                                                                                                            \
        static scn::DeviceOnce attr_set;                                                                               \
        if (attr_set.needed()) {                                                                                            \
            SCN_HIP(hipFuncSetAttribute((const void*)k_conv_tb<N, K, FU, PT>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                        160 * 1024));                                                               \
            attr_set.done();                                                                                        \
        }                                                                                                           \
        hipLaunchKernelGGL((k_conv_tb<N, K, FU, PT>), grid, dim3(TB_NW * 64), lds, st, X, (long long)n_in, cin, tstab, \
                           tile_mask, perm, tile_order, n_off, (long long)nt, image, bias, residual, relu_mask, Y, slabs, \
                           (long long)n_out, cout, flags, n_chunks, n_kc, (int*)arrival, xbins);                    \
    } while (0)
#define PICK_TB(N, K)                                                                                               \
    do {                                                                                                            \
        if (fused && part) LAUNCH_TB(N, K, true, true);                                                             \
        else if (fused) LAUNCH_TB(N, K, true, false);                                                               \
        else if (part) LAUNCH_TB(N, K, false, true);                                                                \
        else LAUNCH_TB(N, K, false, false);                                                                         \
    } while (0)
    if (kh == 2) PICK_TB(2, 2);
    else if (nb == 4) PICK_TB(4, 1);
    else PICK_TB(2, 1);
#undef PICK_TB
#undef LAUNCH_TB
    SCN_LAUNCH_CHECK();
    if (n_kc > 1 && !fused) {
        const int rl = (flags & SCN_F_RESIDUAL_LAST) ? 1 : 0;
        const bool v4 = cout % 4 == 0 && ((((uintptr_t)slabs | (uintptr_t)bias) & 15) == 0) &&
                        ((((uintptr_t)Y | (uintptr_t)residual | (uintptr_t)relu_mask) & 7) == 0);
        if (v4)
            hipLaunchKernelGGL(k_conv_tb_sum<4>, dim3(scn::ew_grid(n_out * cout / 4, 256)), dim3(256), 0, st,
                               (const float*)slabs, n_kc, (long long)n_out, cout, bias, residual, relu_mask, Y, rl);
        else
            hipLaunchKernelGGL(k_conv_tb_sum<1>, dim3(scn::ew_grid(n_out * cout, 256)), dim3(256), 0, st,
                               (const float*)slabs, n_kc, (long long)n_out, cout, bias, residual, relu_mask, Y, rl);
        SCN_LAUNCH_CHECK();
    }
    return SCN_OK;
}
