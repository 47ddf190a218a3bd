// wgrad: weight (and bias) gradient of every conv-type layer,   dW[o] = sum_{p in R_o} in(X[in_p])^T . dY[out_p]
//
// Both MFMA operands of this product are "rule-major": A[i = channel of X][k = rule] and B[k = rule][j = channel of dY]
// mean that lane (i, kq) of v_mfma_f32_16x16x4_f32 needs ONE element of the gathered row of rule kq -- element i.  So a
// 16-lane group can read its fragment straight from the row in global memory: with the channel <-> (tile, i) mapping
// channel = T*i + t, the T fragment values a lane needs for its T tiles are T CONSECUTIVE floats of the row, i.e. one
// global_load_dwordx2/x4 per lane, and the 16 lanes of a group read one contiguous 64*T/4-byte piece of the row.
// No LDS staging, no transposition, no workgroup barrier: every wave streams its own rule range with its own software
// pipeline (row indices 2-3 blocks ahead, rows 2 blocks ahead, three named register sets rotating), which is what a
// latency-bound gather wants -- the first, LDS-staged kernel met at a barrier every 32 rules and needed 2-4x the
// instructions per MFMA (measured 35-60 TFLOP/s; this one 50-72, the small rectangular layers 2x faster).
//
// Workgroup = 4 waves.  Wave block = (16 TA) x (16 TB) channels of dW[o].
//   K mode   : the 4 waves take the 4 quarters of the unit's rule range on the SAME block and add their partial blocks
//              through LDS in wave order at the end (channels <= 64).
//   QUAD mode: the 4 waves take the 2 x 2 quadrants of a (32 TA) x (32 TB) block over the whole rule range (the two waves
//              that share an operand read the same rows at about the same time: second reader hits in L1/L2).
// Every workgroup writes its block to a slab; k_wgradd_sum adds the units of an offset in fixed order (bitwise
// reproducible, no float atomics).  Bias gradient: column sums of the dY fragments of the offsets in db_mask, same route.
#include <stdlib.h>

#include <string.h>
#include <vector>

#include "scn_common.h"

#ifndef WD_DEEP_ALL
#define WD_DEEP_ALL 1       // 1: three register sets (rows two blocks ahead) for every wave block size
#endif

using scn::S;
using scn::cdiv;

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

#ifndef WD_EXP
#define WD_EXP 0            // (developer builds, TIMING ONLY -- wrong results) 1: the dY row pieces and their row indices come from
                            // registers, not from memory: the ceiling of any "one gather per rule" form whose dY tile sits in
                            // LDS (profiles/r6_wgrad_one_gather.txt); 2: the dY pieces are read from a small LDS array (ds_read)
#endif
#ifndef WD_TIMELINE
#define WD_TIMELINE 0       // 1 (developer builds): per-wave wall_clock64 stamps of k_wgrad_direct (tools/wd_timeline.py)
#endif
#if WD_TIMELINE
__device__ long long wd_stamps[65536 * 4];      // [wave slot][t0, t_loop, t_end, rules]
extern "C" int scn_debug_wd_stamps(void* dst_host) {
    return hipMemcpyFromSymbol(dst_host, HIP_SYMBOL(wd_stamps), sizeof(wd_stamps)) == hipSuccess ? 0 : 1;
}
#endif

// Several problems on ONE rule list (the weight gradients of one or two residual units: scn_wgrad_bias_rules2 / _rules_n)
// run as one launch over n_prob x n_real VIRTUAL offsets: virtual offset v = problem * n_real + offset, its rules are
// rule_start[v] - problem * p_rules in the shared list.  A single problem has n_real == n_off and p_rules == 0.
#define WD_MAX_PROB 4
struct DPlan {
    long long rule_start[129];          // prefix of rules per (virtual) offset
    int unit_start[129];                // prefix of work units per offset; a unit = `per` consecutive rules of one offset
    long long per;                      // rules per unit (multiple of 64)
    long long p_rules;                  // rules of one problem (multi-problem launches), else 0
    int n_off, cbi, cbj, nbi, nbj;      // workgroup block (channels of X x channels of dY), blocks along Cin / Cout
    int n_real;                         // offsets per problem
};

template <int T>
struct Frag { typedef float type __attribute__((ext_vector_type(T))); };
// T = 3 (48-channel wave blocks: the reference's 48- and 96-channel layers are whole multiples): a row piece is three
// dwords at a 12-byte lane pitch, so the type must not promise the 16-byte alignment a 3-vector has by default
typedef float f32x3_u __attribute__((ext_vector_type(3), aligned(4)));
template <>
struct Frag<3> { typedef f32x3_u type; };

// operand pairs of the problems of a launch (by value in the kernel arguments)
struct WdOps { const void* X[WD_MAX_PROB]; const void* dY[WD_MAX_PROB]; };

// (virtual) offset of a unit: unit_start is non-decreasing, so v = #{v' : unit >= unit_start[v'+1]} -- two ballots cover
// the 128 virtual offsets of a four-problem launch
__device__ __forceinline__ int wd_offset_of(const DPlan& plan, int unit, int lane) {
    const bool lo = lane < plan.n_off && unit >= plan.unit_start[lane + 1];
    const bool hi = lane + 64 < plan.n_off && unit >= plan.unit_start[lane + 65];
    return __builtin_amdgcn_readfirstlane(__popcll(__ballot(lo)) + __popcll(__ballot(hi)));
}

// Row pieces as they sit in the register ring: fp32 storage = the T floats themselves; bf16 storage (HB) = the T packed
// bf16 values as loaded (half the registers, half the gather bytes), widened to fp32 when the block is multiplied:
// element 2j is the low half of word j (<< 16), element 2j+1 the high half (& 0xffff0000) -- exact, one VALU op each.
template <int T, bool PACKED>
struct RawFrag { typedef typename Frag<T>::type type; };
template <>
struct RawFrag<2, true> { typedef unsigned type __attribute__((ext_vector_type(1))); };
template <>
struct RawFrag<4, true> { typedef unsigned type __attribute__((ext_vector_type(2))); };

template <int T, bool PACKED, typename R>
__device__ __forceinline__ typename Frag<T>::type widen(const R& r) {
    typename Frag<T>::type f;
    if constexpr (PACKED) {
#pragma unroll
        for (int j = 0; j < T / 2; ++j) {
            f[2 * j] = __uint_as_float(r[j] << 16);
            f[2 * j + 1] = __uint_as_float(r[j] & 0xffff0000u);
        }
    } else {
        f = r;
    }
    return f;
}

template <int TA, int TB, bool QUAD, bool EDGE, bool IDENT, bool HB = false>
__global__ __launch_bounds__(256) void k_wgrad_direct(const float* __restrict__ X_0, int cin, const float* __restrict__ dY_0,
                                                      int cout, const int* __restrict__ in_rows,
                                                      const int* __restrict__ out_rows, DPlan plan,
                                                      float* __restrict__ slabs, int relu_in,
                                                      float* __restrict__ db_slabs, unsigned db_mask, int cout_pad,
                                                      WdOps more) {
    typedef typename Frag<TA>::type fa_t;
    typedef typename Frag<TB>::type fb_t;
    constexpr bool PACKED = HB && !EDGE;                         // bf16 rows kept packed in the ring
    typedef typename RawFrag<TA, PACKED>::type ra_t;
    typedef typename RawFrag<TB, PACKED>::type rb_t;
    constexpr int ES = HB ? 2 : 4;                               // bytes per stored element
    constexpr int WI = 16 * TA, WJ = 16 * TB;                   // wave block
    constexpr int CBI = QUAD ? 2 * WI : WI, CBJ = QUAD ? 2 * WJ : WJ;
    constexpr int NACC = TA * TB;
    extern __shared__ __attribute__((aligned(16))) float red[];     // K mode: 4 partial blocks + 4 x 64 bias sums

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#if WD_TIMELINE
    const long long wd_t0 = wall_clock64();
#endif
    const int i = lane & 15, kq = lane >> 4;
    const int unit = blockIdx.x;
    // offset of this unit: unit_start is non-decreasing, so o = #{o' : unit >= unit_start[o'+1]} -- one ballot instead
    // of a serial scalar search (every workgroup pays its prologue; with ~1000 short workgroups it adds up)
    const int ov = wd_offset_of(plan, unit, lane);                               // (virtual) offset of this unit
    const int prob = ov / plan.n_real;                                            // which operand pair (scalar)
    const int o = ov - prob * plan.n_real;
    const float* X = prob ? (const float*)more.X[prob] : X_0;
    const float* dY = prob ? (const float*)more.dY[prob] : dY_0;
    const int s_unit = unit - plan.unit_start[ov];
    const int bi = blockIdx.z / plan.nbj, bj = blockIdx.z % plan.nbj;
    const int wi = QUAD ? (wave >> 1) : 0, wj = QUAD ? (wave & 1) : 0;
    const int ci0 = bi * CBI + wi * WI, co0 = bj * CBJ + wj * WJ;       // first channel of the wave block

    const long long p_shift = prob * plan.p_rules;                     // every problem walks the same rule list
    const long long p_lo = plan.rule_start[ov] - p_shift, p_hi = plan.rule_start[ov + 1] - p_shift;
    const long long p0 = p_lo + (long long)s_unit * plan.per;
    const long long p1 = p0 + plan.per < p_hi ? p0 + plan.per : p_hi;
    // this wave's rule range [q0, q1): K mode = a quarter of the unit (multiple of 16 rules), QUAD = the whole unit
    long long q0 = p0, q1 = p1;
    if (!QUAD) {
        const long long quarter = ((p1 - p0 + 63) / 64) * 16;
        q0 = p0 + wave * quarter;
        q1 = q0 + quarter < p1 ? q0 + quarter : p1;
    }
    // rules of this wave, relative to q0: nfull whole blocks of 16 (pipelined, unmasked) + one masked tail block
    const int nrel = __builtin_amdgcn_readfirstlane(q1 > q0 ? (int)(q1 - q0) : 0);
    const int nfull = nrel / 16;
    const int* inq = IDENT ? nullptr : in_rows + q0;
    const int* outq = IDENT ? nullptr : out_rows + q0;
    const int lane_r = 4 * kq;
    const int q0i = (int)q0;                                   // identity list: rule index == row index (< 2^31)

    // channel offsets of this lane inside a row: T consecutive floats starting at ci0 + TA*i (resp. co0 + TB*i)
    const int ca = ci0 + TA * i, cbn = co0 + TB * i;
    bool a_ok[TA], b_ok[TB];
#pragma unroll
    for (int t = 0; t < TA; ++t) a_ok[t] = !EDGE || ca + t < cin;
#pragma unroll
    for (int t = 0; t < TB; ++t) b_ok[t] = !EDGE || cbn + t < cout;
    const bool do_db = db_slabs != nullptr && ((db_mask >> o) & 1u) && bi == 0 && wi == 0;

    f32x4 acc[TA][TB];
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float dbacc[TB];
#pragma unroll
    for (int t = 0; t < TB; ++t) dbacc[t] = 0.f;

    // A block = 16 rules = 4 MFMA steps; lane group kq owns rules qb + 4*kq + s, s = step.
    // Three named register sets (0,1,2) rotate through: indices of block b+3.., rows of block b+2.., MFMAs of block b.
    // The instruction budget matters as much as the latency: a 16x16x4 fp32 MFMA is 32 cycles = 8 VALU slots, and at
    // TA = TB = 2 a block has only 16 of them.  So: the block offset of the index loads is SCALAR (saddr + lane offset +
    // immediate, zero VALU), a row address is ONE v_mad_i64_i32 (row * stride + per-lane base), ReLU is one v_max
    // against 0 or -inf, and rule masking exists only in the peeled tail block.  (First version: ~110 VALU per block,
    // VALU-issue-bound at 1/3 of the MFMA rate.)
    const char* xlane = (const char*)X + (long long)ca * ES;
    const char* ylane = (const char*)dY + (long long)cbn * ES;
    const int xstride = ES * cin, ystride = ES * cout;         // int: row * stride is one v_mad_i64_i32
    const int relu_lo = (relu_in & 1) ? 0 : (int)0x80000000;
    // EDGE with whole fragments (relu_in bit 1, set by the host when Cin % TA == 0, Cout % TB == 0 and rows are 16-byte
    // aligned -- the reference's 48 / 80 / 112-channel layers on 64-wide blocks): a lane's T channels are all inside the
    // layer or all outside, so the row piece is still ONE vector load, from a clamped lane offset, zeroed at use
    const bool evec = EDGE && !HB && (relu_in & 2);
    const char* xlane_e = (const char*)X + (long long)(a_ok[0] ? ca : 0) * ES;
    const char* ylane_e = (const char*)dY + (long long)(b_ok[0] ? cbn : 0) * ES;
    const int last_full = nfull > 0 ? (nfull - 1) * 16 : 0;        // prefetches past the end re-read the last whole block
#define WD_IDX(IN, OUT, QB)                                                                          \
    {                                                                                                \
        const int qs_ = (QB) < last_full ? (QB) : last_full;           /* scalar */                  \
        _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) {                                           \
            IN[s_] = IDENT ? q0i + qs_ + lane_r + s_ : inq[qs_ + lane_r + s_];                       \
            OUT[s_] = (IDENT || WD_EXP) ? q0i + qs_ + lane_r + s_ : outq[qs_ + lane_r + s_];         \
        }                                                                                            \
    }
#define WD_ROWS(A, B, IN, OUT)                                                                       \
    _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) {                                               \
        if (EDGE) {                  /* element loads; channels past the end read channel 0 (zeroed at use) */ \
            if constexpr (!HB) {                                                                     \
                if (evec) {              /* whole fragments in or out: one vector load per row piece */ \
                    A[s_] = *(const ra_t*)(xlane_e + (long long)IN[s_] * xstride);                   \
                    B[s_] = *(const rb_t*)(ylane_e + (long long)OUT[s_] * ystride);                  \
                    continue;                                                                        \
                }                                                                                    \
            }                                                                                        \
            if (HB) {                                                                                \
                const unsigned short* xr_ = (const unsigned short*)X + (long long)IN[s_] * cin;      \
                const unsigned short* yr_ = (const unsigned short*)dY + (long long)OUT[s_] * cout;   \
                _Pragma("unroll") for (int t_ = 0; t_ < TA; ++t_)                                    \
                    A[s_][t_] = __uint_as_float((unsigned)xr_[a_ok[t_] ? ca + t_ : 0] << 16);       \
                _Pragma("unroll") for (int t_ = 0; t_ < TB; ++t_)                                    \
                    B[s_][t_] = __uint_as_float((unsigned)yr_[b_ok[t_] ? cbn + t_ : 0] << 16);       \
            } else {                                                                                 \
                const float* xr_ = X + (long long)IN[s_] * cin;                                      \
                const float* yr_ = dY + (long long)OUT[s_] * cout;                                   \
                _Pragma("unroll") for (int t_ = 0; t_ < TA; ++t_) A[s_][t_] = xr_[a_ok[t_] ? ca + t_ : 0]; \
                _Pragma("unroll") for (int t_ = 0; t_ < TB; ++t_) B[s_][t_] = yr_[b_ok[t_] ? cbn + t_ : 0]; \
            }                                                                                        \
        } else if (WD_EXP == 1) {                                                                    \
            A[s_] = *(const ra_t*)(xlane + (long long)IN[s_] * xstride);                             \
            _Pragma("unroll") for (int t_ = 0; t_ < (int)(sizeof(rb_t) / 4); ++t_)                   \
                ((float*)&B[s_])[t_] = __int_as_float(0x3f800000 | (OUT[s_] & 0xffff));              \
        } else if (WD_EXP == 2) {                                                                    \
            A[s_] = *(const ra_t*)(xlane + (long long)IN[s_] * xstride);                             \
            B[s_] = *(const rb_t*)((const char*)red + (((OUT[s_] & 15) * 16 + i) * (int)sizeof(rb_t)));  \
        } else {                                                                                     \
            A[s_] = *(const ra_t*)(xlane + (long long)IN[s_] * xstride);                             \
            B[s_] = *(const rb_t*)(ylane + (long long)OUT[s_] * ystride);                            \
        }                                                                                            \
    }
    // MASK: rules at or past `nrel` contribute nothing (tail block only)
#define WD_MFMA(A, B, QB, MASK)                                                                      \
    _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) {                                               \
        const bool v_ = !(MASK) || (QB) + lane_r + s_ < nrel;                                        \
        fa_t a_ = widen<TA, PACKED>(A[s_]);                                                          \
        fb_t b_ = widen<TB, PACKED>(B[s_]);                                                          \
        _Pragma("unroll") for (int t_ = 0; t_ < TA; ++t_) {                                          \
            /* ReLU as ONE integer max on the bit pattern (fmaxf costs a canonicalising v_max x,x more; inline asm \
               hides the VALU->MFMA hazard from the compiler): negative floats are negative ints */  \
            float x_ = __int_as_float(max(__float_as_int(a_[t_]), relu_lo));                         \
            if ((MASK) || EDGE) x_ = (v_ && a_ok[t_]) ? x_ : 0.f;                                    \
            a_[t_] = x_;                                                                             \
        }                                                                                            \
        if ((MASK) || EDGE) {                                                                        \
            _Pragma("unroll") for (int t_ = 0; t_ < TB; ++t_) b_[t_] = (v_ && b_ok[t_]) ? b_[t_] : 0.f; \
        }                                                                                            \
        if (do_db) { _Pragma("unroll") for (int t_ = 0; t_ < TB; ++t_) dbacc[t_] += b_[t_]; }         \
        _Pragma("unroll") for (int ta_ = 0; ta_ < TA; ++ta_)                                         \
            _Pragma("unroll") for (int tb_ = 0; tb_ < TB; ++tb_)                                     \
                acc[ta_][tb_] = MFMA16(a_[ta_], b_[tb_], acc[ta_][tb_]);                             \
    }

    int in0[4], out0[4], in1[4], out1[4], in2[4], out2[4];
    ra_t a0[4], a1[4], a2[4];
    rb_t b0[4], b1[4], b2[4];
    // Ring depth: 3 sets (rows two blocks ahead) for the small wave blocks; 2 sets (one block ahead) at TA = TB = 4,
    // where a block is 64 MFMAs = 2048 cycles and the third set would cost the second resident wave per SIMD
    // (64 accumulators + 3 x 32 row registers + indices > 256 registers).
    constexpr bool DEEP = WD_DEEP_ALL || TA * TB < 16;
    if (DEEP && nfull > 0) {
        // (the prologue's issue order is pinned too: the loop header's wait counts are the merge of both ways in)
        WD_IDX(in0, out0, 0);
        WD_IDX(in1, out1, 16);
        __builtin_amdgcn_sched_barrier(0);
        WD_IDX(in2, out2, 32);
        __builtin_amdgcn_sched_barrier(0);
        WD_ROWS(a0, b0, in0, out0);
        WD_IDX(in0, out0, 48);
        __builtin_amdgcn_sched_barrier(0);
        WD_ROWS(a1, b1, in1, out1);
        WD_IDX(in1, out1, 64);
        // steady state, block b (set b%3 holds its rows): queue rows of b+2 (indices arrived), indices of b+5, multiply b.
        // The scheduling barriers keep each phase's address arithmetic in its phase: hoisted to the loop top it made
        // every iteration wait for the newest index loads (s_waitcnt vmcnt(0)) and serialised the pipeline.
        int qb = 0, b = 0;
        for (; b + 3 <= nfull; b += 3, qb += 48) {
            __builtin_amdgcn_sched_barrier(0);
            WD_ROWS(a2, b2, in2, out2);
            WD_IDX(in2, out2, qb + 80);
            WD_MFMA(a0, b0, qb, false);
            __builtin_amdgcn_sched_barrier(0);
            WD_ROWS(a0, b0, in0, out0);
            WD_IDX(in0, out0, qb + 96);
            WD_MFMA(a1, b1, qb + 16, false);
            __builtin_amdgcn_sched_barrier(0);
            WD_ROWS(a1, b1, in1, out1);
            WD_IDX(in1, out1, qb + 112);
            WD_MFMA(a2, b2, qb + 32, false);
        }
        __builtin_amdgcn_sched_barrier(0);
        // remainder: 0, 1 or 2 whole blocks; their rows are already queued (sets 0 and 1)
        if (b < nfull) { WD_MFMA(a0, b0, qb, false); }
        if (b + 1 < nfull) { WD_MFMA(a1, b1, qb + 16, false); }
    }
    if (!DEEP && nfull > 0) {
        WD_IDX(in0, out0, 0);
        __builtin_amdgcn_sched_barrier(0);
        WD_IDX(in1, out1, 16);
        __builtin_amdgcn_sched_barrier(0);
        WD_ROWS(a0, b0, in0, out0);
        WD_IDX(in0, out0, 32);
        // block b: set b%2 holds its rows; queue rows of b+1, indices of b+3, multiply b
        int qb = 0, b = 0;
        for (; b + 2 <= nfull; b += 2, qb += 32) {
            __builtin_amdgcn_sched_barrier(0);
            WD_ROWS(a1, b1, in1, out1);
            WD_IDX(in1, out1, qb + 48);
            WD_MFMA(a0, b0, qb, false);
            __builtin_amdgcn_sched_barrier(0);
            WD_ROWS(a0, b0, in0, out0);
            WD_IDX(in0, out0, qb + 64);
            WD_MFMA(a1, b1, qb + 16, false);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (b < nfull) { WD_MFMA(a0, b0, qb, false); }
    }
    if (nrel > nfull * 16) {        // tail block: clamped indices, masked values, no pipelining (once per wave)
        const int qt = nfull * 16;
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
            int r_ = qt + lane_r + s_;
            r_ = r_ < nrel ? r_ : nrel - 1;
            in2[s_] = IDENT ? q0i + r_ : inq[r_];
            out2[s_] = IDENT ? q0i + r_ : outq[r_];
        }
        WD_ROWS(a2, b2, in2, out2);
        WD_MFMA(a2, b2, qt, true);
    }
#undef WD_IDX
#undef WD_ROWS
#undef WD_MFMA

#if WD_TIMELINE
    const long long wd_t1 = wall_clock64();
#endif
    float* slab = slabs + ((long long)unit * (plan.nbi * plan.nbj) + blockIdx.z) * (CBI * CBJ);
    float* dbr = red + (QUAD ? 0 : 4 * NACC * 4 * 64);                  // [4 waves][64 columns]

    // ---- K mode: add the four waves' partial blocks in wave order.  Every wave stores its whole block to LDS
    // ([wave][element][lane], conflict-free) and then OWNS a quarter of the elements: it adds the four copies of its
    // quarter in wave order and writes those rows of the slab -- the epilogue is spread over the 4 waves and never holds
    // more than one quarter in registers (a first version had wave 0 add everything: 200+ VGPRs, one wave per SIMD).
    // acc[a][b][j] is row TA*(4kq+j)+a, column TB*i+b of the wave block.
    if (!QUAD) {
        float* mine = red + wave * (NACC * 4 * 64);
#pragma unroll
        for (int a = 0; a < TA; ++a)
#pragma unroll
            for (int b = 0; b < TB; ++b)
#pragma unroll
                for (int j = 0; j < 4; ++j) mine[((a * TB + b) * 4 + j) * 64 + lane] = acc[a][b][j];
        __syncthreads();
        // the 4 TA (a, j) pairs in order p = 4 a + j, TA consecutive ones per wave (TA = 4: a = wave; TA = 2: a = wave / 2,
        // j = 2 (wave & 1) + jj -- the owners of rounds 1-2; TA = 3: three pairs that may straddle two a)
#pragma unroll
        for (int jj = 0; jj < TA; ++jj) {
            const int p_own = wave * TA + jj;
            const int a_own = p_own >> 2, jx = p_own & 3;
            fb_t v;
#pragma unroll
            for (int b = 0; b < TB; ++b) {
                const float* src = red + ((a_own * TB + b) * 4 + jx) * 64 + lane;
                v[b] = ((src[0] + src[NACC * 4 * 64]) + src[2 * NACC * 4 * 64]) + src[3 * NACC * 4 * 64];
            }
            const int r = TA * (4 * kq + jx) + a_own;
            *(fb_t*)(slab + r * CBJ + TB * i) = v;
        }
    } else {
#pragma unroll
        for (int a = 0; a < TA; ++a)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = wi * WI + TA * (4 * kq + j) + a;
                fb_t v;
#pragma unroll
                for (int b = 0; b < TB; ++b) v[b] = acc[a][b][j];
                *(fb_t*)(slab + r * CBJ + wj * WJ + TB * i) = v;
            }
    }

    // ---- bias gradient: lanes (i, kq) hold column sums of their own rules -> add the 4 kq groups, then the waves ------
    if (db_slabs != nullptr && ((db_mask >> o) & 1u) && bi == 0) {
#pragma unroll
        for (int t = 0; t < TB; ++t) {
            float v = dbacc[t];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            dbacc[t] = v;
        }
        if (!QUAD) {
            if (kq == 0) {
#pragma unroll
                for (int t = 0; t < TB; ++t) dbr[wave * 64 + TB * i + t] = dbacc[t];
            }
            __syncthreads();
            if (wave == 0 && kq == 0) {
#pragma unroll
                for (int t = 0; t < TB; ++t) {
                    const int c = TB * i + t;
                    const float v = ((dbr[c] + dbr[64 + c]) + dbr[128 + c]) + dbr[192 + c];
                    if (co0 + c < cout_pad) db_slabs[(long long)unit * cout_pad + co0 + c] = v;
                }
            }
        } else if (wi == 0 && kq == 0) {
#pragma unroll
            for (int t = 0; t < TB; ++t)
                if (cbn + t < cout_pad) db_slabs[(long long)unit * cout_pad + cbn + t] = dbacc[t];
        }
    }
#if WD_TIMELINE
    if (lane == 0) {
        const long long slot = (((long long)blockIdx.x + (long long)gridDim.x * blockIdx.z) * 4 + wave) & 65535;
        wd_stamps[slot * 4 + 0] = wd_t0; wd_stamps[slot * 4 + 1] = wd_t1; wd_stamps[slot * 4 + 2] = wall_clock64();
        wd_stamps[slot * 4 + 3] = nrel;
    }
#endif
}


// ---------------------------------------------------------------------------------------------------------------------------
// bf16 STORAGE on bf16 MFMA (BASELINE configs 3-5): v_mfma_f32_16x16x32_bf16 contracts 32 RULES per instruction, so both
// operands are needed rule-minor (A[channel][rule], B[rule][channel] with the rule index inside a lane's 8 elements) while
// the rows arrive channel-minor.  The transposition is done by the LDS: the 32 gathered rows of a step are written
// row-major into a swizzled image (256-byte pitch, 16-byte chunk c of row r at 16 (c ^ ((r & 3) << 2 | (r >> 2) & 3)), the
// dual-use image of cdna_hip_programming.md T10) and read back with ds_read_b64_tr_b16, which hands lane i of a 16-lane
// group column i of four rows -- exactly an operand fragment.  fp32 accumulation; dW and db come back fp32.
// Workgroup = 4 waves = 2 x 2 wave blocks of (16 TA) x (16 TB) channels; two LDS buffers, one barrier per step; the rows of
// step s + 1 and the row indices of step s + 2 are in flight while step s is multiplied.  Same units, slabs and fixed-order
// sum (k_wgradd_sum) as the fp32 kernel.  Bias gradient: one extra MFMA per column tile with a ones-row A operand.
// ---------------------------------------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMAB32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ int wtb_off(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

template <int TA, int TB, bool IDENT>
__global__ __launch_bounds__(256) void k_wgrad_tb(const unsigned short* __restrict__ X_0, int cin,
                                                  const unsigned short* __restrict__ dY_0, int cout,
                                                  const int* __restrict__ in_rows, const int* __restrict__ out_rows,
                                                  DPlan plan, float* __restrict__ slabs, int relu_in,
                                                  float* __restrict__ db_slabs, unsigned db_mask, int cout_pad,
                                                  WdOps more) {
    constexpr int CBI = 32 * TA, CBJ = 32 * TB;                  // workgroup block
    constexpr int PA = CBI / 8, PB = CBJ / 8;                    // 16-byte pieces per gathered row
    constexpr int NPIECE = 32 * (PA + PB);                       // pieces per step (32 rules, both operands)
    constexpr int NLD = (NPIECE + 255) / 256;                    // pieces per thread
    constexpr int IMG = 32 * 256;                                // bytes of one image (32 rows, 256-byte pitch)
    extern __shared__ __attribute__((aligned(16))) char wlds[];  // [buffer 0/1][image A/B][32 rows][256 B]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave >> 1, wj = wave & 1;
    const int unit = blockIdx.x;
    const int ov = wd_offset_of(plan, unit, lane);                               // (virtual) offset: see DPlan
    const int prob = ov / plan.n_real;
    const int o = ov - prob * plan.n_real;
    const unsigned short* X = prob ? (const unsigned short*)more.X[prob] : X_0;
    const unsigned short* dY = prob ? (const unsigned short*)more.dY[prob] : dY_0;
    const int s_unit = unit - plan.unit_start[ov];
    const int bi = blockIdx.z / plan.nbj, bj = blockIdx.z % plan.nbj;
    const int ci0 = bi * CBI, co0 = bj * CBJ;
    const long long p_shift = prob * plan.p_rules;
    const long long p_lo = plan.rule_start[ov] - p_shift, p_hi = plan.rule_start[ov + 1] - p_shift;
    const long long p0 = p_lo + (long long)s_unit * plan.per;
    const long long p1 = p0 + plan.per < p_hi ? p0 + plan.per : p_hi;
    const int nrel = (int)(p1 > p0 ? p1 - p0 : 0);
    const int nsteps = (nrel + 31) / 32;
    const bool do_db = db_slabs != nullptr && ((db_mask >> o) & 1u) && bi == 0 && wi == 0;

    // ---- loader mapping: piece e of a step -> (image, row, chunk) --------------------------------------------------------
    int l_row[NLD], l_ch[NLD], l_lds[NLD];
    bool l_b[NLD], l_ok[NLD];
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
        const int e = tid + 256 * u;
        const bool isb = e >= 32 * PA;
        const int e2 = isb ? e - 32 * PA : e;
        const int P = isb ? PB : PA;
        l_b[u] = isb;
        l_row[u] = e2 / P;
        l_ch[u] = e2 % P;
        const int chan = (isb ? co0 : ci0) + 8 * l_ch[u];
        l_ok[u] = e < NPIECE && chan < (isb ? cout : cin);       // channel counts are multiples of 8: a piece is in or out
        l_lds[u] = (isb ? IMG : 0) + wtb_off(l_row[u] < 32 ? l_row[u] : 0, l_ch[u]);
    }
    auto load_idx = [&](int step, int (&idx)[NLD]) {
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const long long r = (long long)step * 32 + l_row[u];
            int v = -1;
            if (l_ok[u] && r < nrel) v = IDENT ? (int)(p0 + r) : (l_b[u] ? out_rows[p0 + r] : in_rows[p0 + r]);
            idx[u] = v;
        }
    };
    auto load_rows = [&](const int (&idx)[NLD], uint4 (&st)[NLD]) {
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            st[u] = make_uint4(0u, 0u, 0u, 0u);
            if (idx[u] >= 0) {
                const unsigned short* src = l_b[u] ? dY + (long long)idx[u] * cout + co0 + 8 * l_ch[u]
                                                   : X + (long long)idx[u] * cin + ci0 + 8 * l_ch[u];
                st[u] = *(const uint4*)src;
            }
        }
    };
    auto store_rows = [&](int buf, const uint4 (&st)[NLD]) {
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            if (tid + 256 * u >= NPIECE) continue;
            uint4 v = st[u];
            if (relu_in && !l_b[u]) {                               // ReLU on the X operand: a negative bf16 is a negative int16
                s16x8 a = __builtin_bit_cast(s16x8, v);
                a = __builtin_elementwise_max(a, (s16x8){0, 0, 0, 0, 0, 0, 0, 0});
                v = __builtin_bit_cast(uint4, a);
            }
            *(uint4*)(wlds + buf * 2 * IMG + l_lds[u]) = v;
        }
    };

    f32x4 acc[TA][TB];
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 dbacc[TB];
#pragma unroll
    for (int b = 0; b < TB; ++b) dbacc[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    s16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (lane & 15) == 0 ? (short)0x3F80 : (short)0;

    // transposed-read addresses of this lane (T10): lane 4q + p of 16-lane group g supplies row 8g + 4t + q, chunk c0 + (p>>1)
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    int tr_a[2][TA], tr_b[2][TB];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int row = 8 * g + 4 * t + q;
#pragma unroll
        for (int a = 0; a < TA; ++a) tr_a[t][a] = wtb_off(row, 2 * (wi * TA + a) + (pp >> 1)) + 8 * (pp & 1);
#pragma unroll
        for (int b = 0; b < TB; ++b) tr_b[t][b] = IMG + wtb_off(row, 2 * (wj * TB + b) + (pp >> 1)) + 8 * (pp & 1);
    }

    int idx0[NLD], idx1[NLD];
    uint4 st[NLD];
    if (nsteps > 0) {
        load_idx(0, idx0);
        load_idx(1, idx1);
        load_rows(idx0, st);
    }
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        store_rows(buf, st);                                   // rows of step s (requested one step ago)
        if (s + 1 < nsteps) {
            if (s & 1) { load_rows(idx0, st); load_idx(s + 2, idx1); }      // idx0/idx1 alternate: rows s+1, indices s+2
            else { load_rows(idx1, st); load_idx(s + 2, idx0); }
        }
        __syncthreads();
        const char* base = wlds + buf * 2 * IMG;
        bf16x8 fa[TA], fb[TB];
#pragma unroll
        for (int a = 0; a < TA; ++a) {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + tr_a[0][a]));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + tr_a[1][a]));
            fa[a] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
#pragma unroll
        for (int b = 0; b < TB; ++b) {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + tr_b[0][b]));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + tr_b[1][b]));
            fb[b] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
#pragma unroll
        for (int a = 0; a < TA; ++a)
#pragma unroll
            for (int b = 0; b < TB; ++b) acc[a][b] = MFMAB32(fa[a], fb[b], acc[a][b]);
        if (do_db) {
#pragma unroll
            for (int b = 0; b < TB; ++b) dbacc[b] = MFMAB32(__builtin_bit_cast(bf16x8, ones), fb[b], dbacc[b]);
        }
    }

    // ---- partial block of this unit -> slab (row-major CBI x CBJ); D: row 4 kq + reg, column lane & 15 ---------------------
    float* slab = slabs + ((long long)unit * (plan.nbi * plan.nbj) + blockIdx.z) * (CBI * CBJ);
    const int i = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                slab[(16 * (wi * TA + a) + 4 * kq + j) * CBJ + 16 * (wj * TB + b) + i] = acc[a][b][j];
    if (do_db && kq == 0) {
#pragma unroll
        for (int b = 0; b < TB; ++b) {
            const int c = co0 + 16 * (wj * TB + b) + i;
            if (c < cout_pad) db_slabs[(long long)unit * cout_pad + c] = dbacc[b][0];
        }
    }
}

// dW[o][ci][co] = sum over the units of offset o.  A thread owns V consecutive output channels (one 16-byte slab read
// per unit at V = 4) and every G-th unit; the G partial sums of an element meet in LDS and are added in ascending g --
// a fixed association for a given plan (bitwise reproducible).  G is chosen on the host so that small dW (many units,
// few elements) still fill the chip.  Trailing blocks: bias gradient, one column each.
// What the sum of one weight-gradient launch needs (the unit prefix of its plan and its block shape), by value.
struct SumArgs {
    const float* slabs; float* dW; const float* db_slabs; float* db;
    int unit_start[129];
    int n_off, n_real, cbi, cbj, nbi, nbj, cin, cout, cout_pad, main_blocks, G;
    unsigned db_mask;
};

template <int V, typename A>
__device__ __forceinline__ void wgradd_sum_body(const A& a, int block) {
    typedef typename Frag<V>::type vec_t;
    const float* __restrict__ slabs = a.slabs;
    const int cin = a.cin, cout = a.cout, G = a.G;
    if (block >= a.main_blocks) {
        const int cidx = block - a.main_blocks;                      // (problem, column)
        const int prob = cidx / cout, co = cidx - prob * cout;
        float s = 0.f;
        for (int o = 0; o < a.n_real; ++o) {
            if (!((a.db_mask >> o) & 1u)) continue;
            const int ov = prob * a.n_real + o;
            for (int u = a.unit_start[ov] + threadIdx.x; u < a.unit_start[ov + 1]; u += 256)
                s += a.db_slabs[(long long)u * a.cout_pad + co];
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) s += __shfl_down(s, d);
        __shared__ float w[4];
        if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) a.db[cidx] = (w[0] + w[1]) + (w[2] + w[3]);
        return;
    }
    __shared__ __attribute__((aligned(16))) float part[256 * V];
    const int per_block = 256 / G;                                   // element groups per block
    const int el = threadIdx.x % per_block, g = threadIdx.x / per_block;
    const long long total = (long long)a.n_off * cin * (cout / V);       // element groups (cout % V == 0)
    const int cbi = a.cbi, cbj = a.cbj, nblk = a.nbi * a.nbj;
    const long long eg = (long long)block * per_block + el;
    vec_t sum;
#pragma unroll
    for (int v = 0; v < V; ++v) sum[v] = 0.f;
    long long e = 0;
    if (eg < total) {
        e = eg * V;
        const int co = (int)(e % cout);
        const int ci = (int)((e / cout) % cin);
        const int o = (int)(e / ((long long)cin * cout));
        const int blk = (ci / cbi) * a.nbj + co / cbj;
        const int u0 = a.unit_start[o], nu = a.unit_start[o + 1] - u0;
        const long long stride = (long long)nblk * cbi * cbj;
        const float* p = slabs + ((long long)u0 * nblk + blk) * (cbi * cbj) + (ci % cbi) * cbj + co % cbj;
        for (int s = g; s < nu; s += G) {
            const vec_t x = *(const vec_t*)(p + s * stride);
#pragma unroll
            for (int v = 0; v < V; ++v) sum[v] += x[v];
        }
    }
    if (G > 1) {
        *(vec_t*)(part + (g * per_block + el) * V) = sum;
        __syncthreads();
        if (g == 0) {
            for (int k = 1; k < G; ++k) {
                const vec_t x = *(const vec_t*)(part + (k * per_block + el) * V);
#pragma unroll
                for (int v = 0; v < V; ++v) sum[v] += x[v];
            }
        }
    }
    if (g == 0 && eg < total) *(vec_t*)(a.dW + e) = sum;
}

template <int V>
__global__ __launch_bounds__(256) void k_wgradd_sum(SumArgs a) { wgradd_sum_body<V>(a, (int)blockIdx.x); }

// The sums of several weight-gradient launches in ONE launch (deferred sums of a network level, scn_wgrad_defer_begin /
// _flush): block -> (job, block of the job); per job the arithmetic and its order are k_wgradd_sum's.
#define WD_SUM_MANY 6
struct SumJobs { int n; int block_start[WD_SUM_MANY + 1]; SumArgs job[WD_SUM_MANY]; };

__global__ __launch_bounds__(256) void k_wgradd_sum_many(SumJobs jobs) {
    int j = 0;
    while (j + 1 < jobs.n && (int)blockIdx.x >= jobs.block_start[j + 1]) ++j;
    wgradd_sum_body<4>(jobs.job[j], (int)blockIdx.x - jobs.block_start[j]);
}

namespace {
struct Shape { int ta, tb; bool quad; };

Shape pick_shape(int cin, int cout, bool hb = false) {
    Shape s;
    s.ta = cin > 32 ? 4 : 2;
    s.tb = cout > 32 ? 4 : 2;
    s.quad = cin > 64 && cout > 64;
    // the reference's 48- and 96-channel layers (scannet_config/run.py:539-549): whole multiples of a 48-wide wave block
    // (TA = TB = 3) -- 64- / 128-wide blocks execute 1.78x their MFMAs (fp32 rows only: no packed-bf16 ring for T = 3)
    const bool no_t3 = scn::sw(scn::SW_WD_NO_T3).set;              // (scn_debug_set: the tests switch it inside one process)
    if (!hb && !no_t3 && cin % 48 == 0 && cout % 48 == 0 && cin <= 96 && cout <= 96) { s.ta = s.tb = 3; s.quad = false; }
    return s;
}

// bf16 MFMA kernel: tiles of 16 channels per wave along Cin / Cout (workgroup block 32 T x 32 T)
int tb_tiles(int c) { return c <= 32 ? 1 : (c <= 64 ? 2 : 4); }
bool tb_usable(int cin, int cout) {
    const bool off = scn::sw(scn::SW_WGRAD_BF16_MFMA).set && scn::sw(scn::SW_WGRAD_BF16_MFMA).i == 0;
    return !off && cin % 8 == 0 && cout % 8 == 0;
}

int make_dplan(int cin, int cout, const int64_t* prefix_host, int n_off, DPlan& pl, bool hb_mfma = false, bool hb = false) {
    const Shape sh = pick_shape(cin, cout, hb || hb_mfma);
    pl.n_off = n_off;
    pl.n_real = n_off;
    pl.p_rules = 0;
    pl.cbi = 16 * sh.ta * (sh.quad ? 2 : 1);
    pl.cbj = 16 * sh.tb * (sh.quad ? 2 : 1);
    if (hb_mfma) { pl.cbi = 32 * tb_tiles(cin); pl.cbj = 32 * tb_tiles(cout); }
    pl.nbi = (int)cdiv(cin, pl.cbi);
    pl.nbj = (int)cdiv(cout, pl.cbj);
    pl.rule_start[0] = prefix_host[0];
    for (int o = 0; o < n_off; ++o) {
        if (prefix_host[o + 1] < prefix_host[o]) return SCN_EINVAL;
        pl.rule_start[o + 1] = prefix_host[o + 1];
    }
    const int64_t total = prefix_host[n_off] - prefix_host[0];
    const int64_t nblk = (int64_t)pl.nbi * pl.nbj;
    // Unit count: the grid should fill the resident workgroup slots of the chip a whole number of times (R rounds) --
    // 1.05 rounds costs as much as 2.  Slots = 256 CUs x workgroups per CU (registers / LDS of the instantiation).
    // Every offset rounds its unit count up, hence the n_off margin.  R and the slot counts: tools/sweep_wgrad_splits.py.
    int occ = sh.quad ? 2 : (sh.ta == 4 && sh.tb == 4 ? 2 : (sh.ta == 2 && sh.tb == 2 ? 4 : 3));
    int rounds = (sh.quad ? nblk > 1 : (sh.ta == 4 && sh.tb == 4) || (sh.ta == 3 && nblk > 1)) ? 2 : 1;
    const bool tb_big = hb_mfma && tb_tiles(cin) == 4 && tb_tiles(cout) == 4;      // 128 x 128 workgroup blocks
    if (hb_mfma) { occ = 4; rounds = 1; }                                   // 32 KB of LDS, <= 128 registers: 4 per CU
    int64_t target = ((int64_t)scn::cu_budget() * occ * rounds) / nblk - n_off;
    if (scn::sw(scn::SW_WGRAD_SPLITS).set) target = (int)scn::sw(scn::SW_WGRAD_SPLITS).i;       // developer override
    if (target < 1) target = 1;
    const int64_t gran = hb_mfma ? 32 : (sh.quad ? 16 : 64);
    int64_t per = cdiv(cdiv(total, target), gran) * gran;
    if (tb_big && !scn::sw(scn::SW_WGRAD_SPLITS).set && n_off > 0 && total > 0 && per < 1024) {
        // bf16-MFMA kernel, 128 x 128 blocks, SHORT units (a unit writes 64 KB of partial sums for `per` rules): half as many
        // workgroups, and units of EQUAL length inside an offset -- k units per (average) offset.  A `per` just above half
        // an offset's rules makes units of 1 : 0.35 and half again as many of them (tools/sweep_wgrad_tb_units.py: the
        // paired launch at C = 256 jumps from 42.5 to 57.3 us between 110 and 120 target units: per 672 = two equal units
        // of the ~1340 rules of an offset, per 608 = three).  Measured on the cfg-2 scene: 51.8 -> 40.6 us per paired
        // launch at C = 128, 50.4 -> 42.6 at C = 256; long units (600 k voxels) keep all four workgroups per CU.
        const int64_t avg = cdiv(total, n_off), slots = ((int64_t)scn::cu_budget() * 2) / nblk;
        int64_t k = (slots + n_off / 2) / n_off;
        if (k < 1) k = 1;
        for (;; --k) {                           // the largest k whose units still fit the slots in one round
            per = cdiv(cdiv(avg, k), gran) * gran;
            int64_t units = 0;
            for (int o = 0; o < n_off; ++o) units += cdiv(prefix_host[o + 1] - prefix_host[o], per);
            if (units <= slots || k == 1) break;
        }
    }
    const int64_t min_per = hb_mfma ? 256 : (sh.quad ? 128 : 512);         // >= 8 steps per workgroup
    if (per < min_per) per = min_per;
    pl.per = per;
    pl.unit_start[0] = 0;
    for (int o = 0; o < n_off; ++o)
        pl.unit_start[o + 1] = pl.unit_start[o] + (int)cdiv(prefix_host[o + 1] - prefix_host[o], per);
    return SCN_OK;
}
}  // namespace

extern "C" int64_t scn_wgrad_scratch_bytes(int cin, int cout, const int64_t* prefix_host, int n_off) {
    if (!prefix_host || n_off < 1 || n_off > 32 || cin < 1 || cout < 1) return -1;
    // the largest of the plans a call may use: fp32 rows (v = 0), bf16-stored rows on the fp32-MFMA kernel (v = 1: other
    // block shapes than v = 0 where 48-wide blocks apply), bf16-MFMA kernel (v = 2)
    int64_t best = -1;
    for (int v = 0; v < 3; ++v) {
        if (v == 2 && !tb_usable(cin, cout)) continue;
        DPlan pl;
        if (make_dplan(cin, cout, prefix_host, n_off, pl, v == 2, v == 1) != SCN_OK) return -1;
        const int64_t b = (int64_t)pl.unit_start[n_off] *
                              ((int64_t)pl.nbi * pl.nbj * pl.cbi * pl.cbj + (int64_t)pl.nbj * pl.cbj) * (int64_t)sizeof(float) + 512;
        if (b > best) best = b;
    }
    return best;
}

// n_prob problems on one rule list: n_prob x n_off virtual offsets, problem p's rules behind those of problem p - 1 in the
// virtual rule space; dW = [n_prob][n_off][cin][cout], db = [n_prob][cout].
static int multi_problem_plan(int cin, int cout, const int64_t* prefix_host, int n_off, int n_prob, DPlan& pl,
                              bool hb_mfma = false, bool hb = false) {
    if (n_off > 32 || n_prob < 2 || n_prob > WD_MAX_PROB || prefix_host[0] != 0) return SCN_EINVAL;
    int64_t vprefix[129];
    const int64_t P = prefix_host[n_off];
    for (int v = 0; v <= n_prob * n_off; ++v) vprefix[v] = (v / n_off) * P + prefix_host[v % n_off];
    vprefix[n_prob * n_off] = n_prob * P;
    const int rc = make_dplan(cin, cout, vprefix, n_prob * n_off, pl, hb_mfma, hb);
    pl.n_real = n_off;
    pl.p_rules = P;
    return rc;
}

// Deferred sums (scn_wgrad_defer_begin / _flush): per calling thread, the sums of the launches made in between.
namespace {
struct DeferState { bool on = false; std::vector<SumArgs> jobs; std::vector<int> blocks; };
thread_local DeferState g_defer;
}  // namespace

static int wgrad_impl(const float* X, int cin, const float* dY, int cout, const int32_t* in_rows,
                      const int32_t* out_rows, const int64_t* prefix_host, int n_off_real, float* dW, float* db,
                      unsigned db_mask, void* scratch, int flags, scn_stream_t stream, bool hb = false,
                      int n_prob = 1, const void* const* Xs = nullptr, const void* const* dYs = nullptr) {
    SCN_REQUIRE(prefix_host && n_off_real >= 1 && n_off_real <= 32 && cin >= 1 && cout >= 1 && dW && scratch);
    SCN_REQUIRE((in_rows == nullptr) == (out_rows == nullptr));
    SCN_REQUIRE(in_rows || n_off_real == 1);
    // several problems (operand pairs Xs[p], dYs[p]; Xs[0] == X, dYs[0] == dY) on the one rule list
    const bool multi = n_prob > 1;
    SCN_REQUIRE(n_prob >= 1 && n_prob <= WD_MAX_PROB && (!multi || (Xs && dYs && in_rows)));
    WdOps more{};
    uintptr_t all_ptrs = (uintptr_t)X | (uintptr_t)dY;
    for (int q = 0; multi && q < n_prob; ++q) {
        SCN_REQUIRE(Xs[q] && dYs[q]);
        more.X[q] = Xs[q];
        more.dY[q] = dYs[q];
        all_ptrs |= (uintptr_t)Xs[q] | (uintptr_t)dYs[q];
    }
    SCN_REQUIRE(!multi || (Xs[0] == (const void*)X && dYs[0] == (const void*)dY));
    const int n_off = n_prob * n_off_real;                                   // (virtual) offsets of the launch
    const bool mfma16 = hb && tb_usable(cin, cout) && ((all_ptrs & 15) == 0);
    DPlan pl;
    if (multi) SCN_REQUIRE(multi_problem_plan(cin, cout, prefix_host, n_off_real, n_prob, pl, mfma16, hb) == SCN_OK);
    else SCN_REQUIRE(make_dplan(cin, cout, prefix_host, n_off, pl, mfma16, hb) == SCN_OK);
    SCN_REQUIRE(prefix_host[n_off_real] == prefix_host[0] || (X && dY));
    SCN_REQUIRE((all_ptrs & (hb ? 1 : 3)) == 0);
    if (pl.unit_start[n_off] == 0) {
        SCN_HIP(hipMemsetAsync(dW, 0, sizeof(float) * (size_t)n_off * cin * cout, S(stream)));
        if (db) SCN_HIP(hipMemsetAsync(db, 0, sizeof(float) * (size_t)n_prob * cout, S(stream)));
        return SCN_OK;
    }
    const Shape sh = pick_shape(cin, cout, hb);
    // vector row pieces need aligned rows and whole blocks; anything else takes the element-wise (EDGE) instantiation
    const bool edge = (cin % (16 * sh.ta) != 0) || (cout % (16 * sh.tb) != 0) ||
                      ((all_ptrs & (sh.ta == 3 ? 3 : 15)) != 0);
    const bool ident = in_rows == nullptr;
    const int cout_pad = pl.nbj * pl.cbj;
    float* db_slabs = db ? (float*)scratch + (int64_t)pl.unit_start[n_off] * pl.nbi * pl.nbj * pl.cbi * pl.cbj : nullptr;
    dim3 grid((unsigned)pl.unit_start[n_off], 1, (unsigned)(pl.nbi * pl.nbj));
    // bit 1: EDGE blocks whose fragments are whole (vector loads stay possible, see k_wgrad_direct)
    const bool evec = edge && !hb && cin % sh.ta == 0 && cout % sh.tb == 0 && (all_ptrs & 15) == 0 &&
                      (cin * 4) % 16 == 0 && (cout * 4) % 16 == 0 && !scn::sw(scn::SW_WD_NO_EVEC).set;
    const int relu_in = ((flags & SCN_F_RELU_IN) ? 1 : 0) | (evec ? 2 : 0);
    if (mfma16) {
        const int ta = tb_tiles(cin), tb = tb_tiles(cout);
#define LAUNCH_WT(TA_, TB_, I_)                                                                                  \
    hipLaunchKernelGGL((k_wgrad_tb<TA_, TB_, I_>), grid, dim3(256), 4 * 32 * 256, S(stream), (const unsigned short*)X, cin, \
                       (const unsigned short*)dY, cout, in_rows, out_rows, pl, (float*)scratch, relu_in, db_slabs, db_mask, \
                       cout_pad, more)
#define PICK_WT(TA_, TB_) do { if (ident) LAUNCH_WT(TA_, TB_, true); else LAUNCH_WT(TA_, TB_, false); } while (0)
        if (ta == 1 && tb == 1) PICK_WT(1, 1);
        else if (ta == 1 && tb == 2) PICK_WT(1, 2);
        else if (ta == 1 && tb == 4) PICK_WT(1, 4);
        else if (ta == 2 && tb == 1) PICK_WT(2, 1);
        else if (ta == 2 && tb == 2) PICK_WT(2, 2);
        else if (ta == 2 && tb == 4) PICK_WT(2, 4);
        else if (ta == 4 && tb == 1) PICK_WT(4, 1);
        else if (ta == 4 && tb == 2) PICK_WT(4, 2);
        else PICK_WT(4, 4);
#undef PICK_WT
#undef LAUNCH_WT
    } else {
#define LAUNCH_WD(TA_, TB_, Q_, E_, I_, H_)                                                                      \
    do {                                                                                                         \
        const size_t lds_ = ((Q_) ? 4 * 64 : 4 * (TA_) * (TB_) * 4 * 64 + 4 * 64) * sizeof(float);               \
        static scn::DeviceOnce attr_set;                                                                            \
        if (attr_set.needed()) {                                                                                         \
            SCN_HIP(hipFuncSetAttribute((const void*)k_wgrad_direct<TA_, TB_, Q_, E_, I_, H_>,                   \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                \
            attr_set.done();                                                                                     \
        }                                                                                                        \
        hipLaunchKernelGGL((k_wgrad_direct<TA_, TB_, Q_, E_, I_, H_>), grid, dim3(256), lds_, S(stream), X, cin, \
                           dY, cout, in_rows, out_rows, pl, (float*)scratch, relu_in, db_slabs, db_mask,         \
                           cout_pad, more);                                                                      \
    } while (0)
#define PICK_I(TA_, TB_, Q_, E_, H_)                                                                             \
    do { if (ident) LAUNCH_WD(TA_, TB_, Q_, E_, true, H_); else LAUNCH_WD(TA_, TB_, Q_, E_, false, H_); } while (0)
#define PICK_EI(TA_, TB_, Q_)                                                                                    \
    do {                                                                                                         \
        if (hb) { if (edge) PICK_I(TA_, TB_, Q_, true, true); else PICK_I(TA_, TB_, Q_, false, true); }          \
        else { if (edge) PICK_I(TA_, TB_, Q_, true, false); else PICK_I(TA_, TB_, Q_, false, false); }           \
    } while (0)
    if (sh.quad) PICK_EI(4, 4, true);
    else if (sh.ta == 3) { if (ident) LAUNCH_WD(3, 3, false, false, true, false); else LAUNCH_WD(3, 3, false, false, false, false); }
    else if (sh.ta == 4 && sh.tb == 4) PICK_EI(4, 4, false);
    else if (sh.ta == 4) PICK_EI(4, 2, false);
    else if (sh.tb == 4) PICK_EI(2, 4, false);
    else PICK_EI(2, 2, false);
#undef PICK_EI
#undef PICK_I
#undef LAUNCH_WD
    }
    SCN_LAUNCH_CHECK();
    // sum of the units: V output channels per thread, G threads per element group so that ~>= 128k threads run
    const int V = (cout % 4 == 0 && (((uintptr_t)dW | (uintptr_t)scratch) & 15) == 0) ? 4 : 1;
    const int64_t groups = (int64_t)n_off * cin * (cout / V);
    int G = 1;
    while (G < 16 && groups * G < 128 * 1024) G *= 2;
    SumArgs sa;
    sa.slabs = (const float*)scratch; sa.dW = dW; sa.db_slabs = db_slabs; sa.db = db;
    memcpy(sa.unit_start, pl.unit_start, sizeof(sa.unit_start));
    sa.n_off = pl.n_off; sa.n_real = pl.n_real; sa.cbi = pl.cbi; sa.cbj = pl.cbj; sa.nbi = pl.nbi; sa.nbj = pl.nbj;
    sa.cin = cin; sa.cout = cout; sa.cout_pad = cout_pad; sa.main_blocks = (int)cdiv(groups, 256 / G); sa.G = G;
    sa.db_mask = db_mask;
    const int blocks = sa.main_blocks + (db ? n_prob * cout : 0);
    if (g_defer.on && V == 4) {                  // the caller batches the sums of several launches (own scratch per launch)
        g_defer.jobs.push_back(sa);
        g_defer.blocks.push_back(blocks);
        return SCN_OK;
    }
    if (V == 4) hipLaunchKernelGGL(k_wgradd_sum<4>, dim3(blocks), dim3(256), 0, S(stream), sa);
    else hipLaunchKernelGGL(k_wgradd_sum<1>, dim3(blocks), dim3(256), 0, S(stream), sa);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

extern "C" int scn_wgrad_defer_begin(void) {
    g_defer.on = true;                           // (a recorder left open by a failed pass is simply restarted)
    g_defer.jobs.clear();
    g_defer.blocks.clear();
    return SCN_OK;
}

extern "C" int scn_wgrad_defer_flush(scn_stream_t stream) {
    SCN_REQUIRE(g_defer.on);
    g_defer.on = false;
    const size_t n = g_defer.jobs.size();
    for (size_t base = 0; base < n; base += WD_SUM_MANY) {
        SumJobs jobs;
        jobs.n = (int)(n - base < WD_SUM_MANY ? n - base : WD_SUM_MANY);
        jobs.block_start[0] = 0;
        for (int j = 0; j < jobs.n; ++j) {
            jobs.job[j] = g_defer.jobs[base + j];
            jobs.block_start[j + 1] = jobs.block_start[j] + g_defer.blocks[base + j];
        }
        if (jobs.n == 1) hipLaunchKernelGGL(k_wgradd_sum<4>, dim3(jobs.block_start[1]), dim3(256), 0, S(stream), jobs.job[0]);
        else hipLaunchKernelGGL(k_wgradd_sum_many, dim3(jobs.block_start[jobs.n]), dim3(256), 0, S(stream), jobs);
        SCN_LAUNCH_CHECK();
    }
    g_defer.jobs.clear();
    g_defer.blocks.clear();
    return SCN_OK;
}

extern "C" int scn_wgrad_rules(const float* X, int cin, const float* dY, int cout, const int32_t* in_rows,
                               const int32_t* out_rows, const int64_t* prefix_host, int n_off, float* dW, void* scratch,
                               int flags, scn_stream_t stream) {
    return wgrad_impl(X, cin, dY, cout, in_rows, out_rows, prefix_host, n_off, dW, nullptr, 0u, scratch, flags, stream);
}

// bf16 STORAGE of both operands (BASELINE configs 3-5): the rows are gathered as packed bf16 and widened to fp32 in
// registers (exact), the products and sums are the fp32 kernel's (v_mfma_f32_16x16x4_f32) -- dW is fp32.
extern "C" int scn_wgrad_rules_bf16(const uint16_t* X, int cin, const uint16_t* dY, int cout, const int32_t* in_rows,
                                    const int32_t* out_rows, const int64_t* prefix_host, int n_off, float* dW,
                                    void* scratch, int flags, scn_stream_t stream) {
    return wgrad_impl((const float*)X, cin, (const float*)dY, cout, in_rows, out_rows, prefix_host, n_off, dW, nullptr,
                      0u, scratch, flags, stream, true);
}

extern "C" int scn_wgrad_bias_rules(const float* X, int cin, const float* dY, int cout, const int32_t* in_rows,
                                    const int32_t* out_rows, const int64_t* prefix_host, int n_off, float* dW, float* db,
                                    uint32_t db_offsets, void* scratch, int flags, scn_stream_t stream) {
    SCN_REQUIRE(db && db_offsets);
    return wgrad_impl(X, cin, dY, cout, in_rows, out_rows, prefix_host, n_off, dW, db, db_offsets, scratch, flags,
                      stream);
}

// The weight (and bias) gradients of the n_prob = 2 convolutions of a residual unit (or the 4 of two stacked units) in ONE
// launch + one sum: they share the rule list and the channel counts; (Xs[p], dYs[p]) are their operand pairs.
// dW = [n_prob][n_off][cin][cout], db = [n_prob][cout] (NULL: no bias gradients).  Same per-unit arithmetic and fixed-order
// sum as scn_wgrad_bias_rules under this call's plan; what it buys is a launch whose tail and fixed costs are paid once
// for n_prob times the work (tools/wgrad_batch_bound.py: two problems take 0.82-0.85 of two calls).
extern "C" int64_t scn_wgrad_scratch_bytes_n(int cin, int cout, const int64_t* prefix_host, int n_off, int n_prob) {
    if (!prefix_host || n_off < 1 || n_off > 32 || cin < 1 || cout < 1 || n_prob < 2 || n_prob > WD_MAX_PROB) return -1;
    int64_t best = -1;
    for (int v = 0; v < 3; ++v) {                    // fp32 rows; bf16-stored rows on the fp32-MFMA kernel; bf16-MFMA plan
        if (v == 2 && !tb_usable(cin, cout)) continue;
        DPlan pl;
        if (multi_problem_plan(cin, cout, prefix_host, n_off, n_prob, pl, v == 2, v == 1) != SCN_OK) return -1;
        const int64_t b = (int64_t)pl.unit_start[n_prob * n_off] *
                              ((int64_t)pl.nbi * pl.nbj * pl.cbi * pl.cbj + (int64_t)pl.nbj * pl.cbj) * (int64_t)sizeof(float) + 512;
        if (b > best) best = b;
    }
    return best;
}

extern "C" int64_t scn_wgrad_scratch_bytes2(int cin, int cout, const int64_t* prefix_host, int n_off) {
    return scn_wgrad_scratch_bytes_n(cin, cout, prefix_host, n_off, 2);
}

extern "C" int scn_wgrad_bias_rules_n(const float* const* Xs, const float* const* dYs, int n_prob, int cin, int cout,
                                     const int32_t* in_rows, const int32_t* out_rows, const int64_t* prefix_host, int n_off,
                                     float* dW, float* db, uint32_t db_offsets, void* scratch, int flags,
                                     scn_stream_t stream) {
    SCN_REQUIRE(Xs && dYs && n_prob >= 2 && n_prob <= WD_MAX_PROB && in_rows && out_rows);
    SCN_REQUIRE((db != nullptr) == (db_offsets != 0));
    return wgrad_impl(Xs[0], cin, dYs[0], cout, in_rows, out_rows, prefix_host, n_off, dW, db, db_offsets, scratch, flags,
                      stream, false, n_prob, (const void* const*)Xs, (const void* const*)dYs);
}

/* ... for bf16-stored operand pairs (dW, db fp32). */
extern "C" int scn_wgrad_bias_rules_n_bf16(const uint16_t* const* Xs, const uint16_t* const* dYs, int n_prob, int cin,
                                          int cout, const int32_t* in_rows, const int32_t* out_rows,
                                          const int64_t* prefix_host, int n_off, float* dW, float* db, uint32_t db_offsets,
                                          void* scratch, int flags, scn_stream_t stream) {
    SCN_REQUIRE(Xs && dYs && n_prob >= 2 && n_prob <= WD_MAX_PROB && in_rows && out_rows);
    SCN_REQUIRE((db != nullptr) == (db_offsets != 0));
    return wgrad_impl((const float*)Xs[0], cin, (const float*)dYs[0], cout, in_rows, out_rows, prefix_host, n_off, dW, db,
                      db_offsets, scratch, flags, stream, true, n_prob, (const void* const*)Xs, (const void* const*)dYs);
}

extern "C" int scn_wgrad_bias_rules2(const float* X0, const float* dY0, const float* X1, const float* dY1, int cin, int cout,
                                     const int32_t* in_rows, const int32_t* out_rows, const int64_t* prefix_host, int n_off,
                                     float* dW, float* db, uint32_t db_offsets, void* scratch, int flags,
                                     scn_stream_t stream) {
    const float* Xs[2] = {X0, X1};
    const float* dYs[2] = {dY0, dY1};
    return scn_wgrad_bias_rules_n(Xs, dYs, 2, cin, cout, in_rows, out_rows, prefix_host, n_off, dW, db, db_offsets, scratch,
                                 flags, stream);
}

extern "C" int scn_wgrad_bias_rules2_bf16(const uint16_t* X0, const uint16_t* dY0, const uint16_t* X1, const uint16_t* dY1,
                                          int cin, int cout, const int32_t* in_rows, const int32_t* out_rows,
                                          const int64_t* prefix_host, int n_off, float* dW, float* db, uint32_t db_offsets,
                                          void* scratch, int flags, scn_stream_t stream) {
    const uint16_t* Xs[2] = {X0, X1};
    const uint16_t* dYs[2] = {dY0, dY1};
    return scn_wgrad_bias_rules_n_bf16(Xs, dYs, 2, cin, cout, in_rows, out_rows, prefix_host, n_off, dW, db, db_offsets,
                                      scratch, flags, stream);
}

extern "C" int scn_wgrad_bias_rules_bf16(const uint16_t* X, int cin, const uint16_t* dY, int cout, const int32_t* in_rows,
                                         const int32_t* out_rows, const int64_t* prefix_host, int n_off, float* dW,
                                         float* db, uint32_t db_offsets, void* scratch, int flags, scn_stream_t stream) {
    SCN_REQUIRE(db && db_offsets);
    return wgrad_impl((const float*)X, cin, (const float*)dY, cout, in_rows, out_rows, prefix_host, n_off, dW, db,
                      db_offsets, scratch, flags, stream, true);
}
