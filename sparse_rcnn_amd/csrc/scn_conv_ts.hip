// conv_ts: output-stationary sparse convolution over MASK-SORTED row tiles -- the hot kernel of the backbone.
//
//   Y[r] = residual[r] + bias + sum_o in(X[table[o][r]]) . W[o']              (rows visited in tiles of 16 sorted rows)
//
// Design (measurements in profiles/):
//   * A wave owns TPW tiles of 16 output rows x 32 output columns; their accumulators (TPW x 2 x f32x4) stay in
//     registers for the whole kernel -- no scatter, no atomics, no LDS traffic for results.  Two earlier versions that
//     accumulated dense rule tiles into an LDS slab lost to this one: ds_add_f32 costs ~200 cycles per wave-instruction
//     and the lockstep-per-offset variant was bound by barriers and per-tile read-modify-write chains.
//   * Rows are grouped by offset mask (scn_tiles.hip), so a tile visits only the offsets in its tile_mask and most of
//     its 16 rows are live in each: executed/useful MFMA work 1.2-1.4 (3^3) / 1.02 (2^3 stride 2).
//   * The weight slices W[o'][kc..kc+32)[n0..n0+32) of ALL offsets are staged once per K-chunk into LDS (n_off x 4 KB,
//     110 KB for 27 offsets) and shared by every wave of the workgroup; B fragments are ds_read_b128 from an
//     XOR-swizzled [o][n][k] image.  One barrier pair per K-chunk, none inside.
//   * A fragments are gathered straight from HBM/L2 into registers: lane (i, kq) reads 2 x 16 bytes of row
//     tstab[t][o][i]; the index block of a (tile, offset) is one coalesced 64-byte read.
//   * v_mfma_f32_16x16x4_f32, exact fp32.  Step (half, e) contracts channels 16*half + 4*kq + e (kq = lane>>4): a fixed
//     permutation of the summation order shared by A and B so each lane fetches 4 channels per 16-byte access.
//     C/D: acc[j] = D[4*kq + j][lane & 15].
#include <atomic>
#include <stdlib.h>

#include "scn_common.h"

#ifndef TS_EXP
#define TS_EXP 0
#endif
#ifndef TS_DEPTH
#define TS_DEPTH 3          // row gathers this many offsets ahead of the MFMAs (3: indices 5 ahead; 5: indices 7 ahead)
#endif
#ifndef TS_TIMELINE_NOPROG
#define TS_TIMELINE_NOPROG 0
#endif
#ifndef TS_TIMELINE
#define TS_TIMELINE 0       // 1: per-wave wall_clock64 stamps appended to the scratch buffer (tools/ts_timeline.py)
#endif

using scn::S;
using scn::cdiv;

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

static constexpr int TS_KC = 32;        // channels per K-chunk
static constexpr int TS_CT = 32;        // output columns per workgroup
static constexpr int TS_T = 16;         // rows per tile
#ifndef TS_NW_V
#define TS_NW_V 16
#endif
static constexpr int TS_NW = TS_NW_V;   // waves per workgroup (16; -DTS_NW_V=12: the co-residency experiment of DESIGN.md section 4.1)

// LDS image of a weight slice: Ws[o][k][n], n contiguous (the layer's own layout: staging is a straight 16-byte copy),
// columns XOR-ed with 16 on every other group of 4 channels: the B fragment of MFMA step (half, e) is read by lane
// (i, kq) at channel k = 16*half + 4*kq + e, column i (+16): lanes of one 32-lane half differ in kq by 1, so the two kq
// land on opposite halves of the 32 banks -- conflict-free ds_read_b32, immediate offsets only.
__device__ __forceinline__ int ts_ws_off(int o, int k, int n) { return (o * TS_KC + k) * TS_CT + (n ^ (((k >> 2) & 1) << 4)); }

// Work decomposition (one launch per layer, plus a reduction when the layer has more than 32 input channels):
//   workgroup = (tile group tg, column chunk, K-chunk).  It stages ITS weight slice W[o'][kc..kc+32)[n0..n0+32) for all
//   offsets once (n_off x 4 KB), then its 16 waves pull the tiles tg, tg + n_tg, ... from an LDS counter (dynamic: tiles
//   differ in cost by their offset count; interleaved: the mask sort orders tiles by offset set, so contiguous ranges
//   would be badly unbalanced).  The grid is sized to the chip (256 CUs x workgroups that fit in LDS), so there is no
//   tail of half-empty waves and no restaging.
//   With one K-chunk the tile epilogue writes Y (bias, residual, ReLU-backward mask fused).  With several, each
//   K-chunk's workgroups write their partial sums to slab[kc] and k_conv_ts_sum adds the slabs in fixed order.
// FULLK: the fast path (rows gathered through a buffer descriptor, Cin % 4 == 0); PART: Cin is not a multiple of the
// 32-channel K-chunk, the lanes of the last chunk whose channels lie past Cin gather from an out-of-range offset (zeros)
// FUSED: K-chunk partial sums are reduced INSIDE this launch (no k_conv_ts_sum): every wave publishes its 16 x 32 partial
// tile to slab[(tile, chunk)][kc] in fragment layout (two 1 KB write-through stores), takes a ticket on the (tile, chunk)
// arrival counter, and the wave whose ticket is the last one adds the n_kc partials in ascending K-chunk order (all
// through L1-bypassing loads) and runs the epilogue.  Same association as k_conv_ts_sum: the
// two forms give the same bits.  Publish / consume follow MI355X_MICROARCH.md "splitk-seam" and cdna_hip_programming.md
// Guideline 16 R1: payload stored sc1 (write-through, 16 B per lane, whole 128-B lines per instruction), every storing
// wave drains its stores (s_waitcnt vmcnt(0)) before its agent-scope ticket add, the combiner learns it is last from the
// value its own add returned and reads the payload with sc1 loads only; no fence.  The counters are zero on entry and the
// last arriver puts its counter back to zero, so the caller zeroes the array once, not per launch.
// TAIL (round 3; the reference's own channel plan 32-48-64-80-96-112, scannet_config/run.py:539-549): a K-chunk with at most
// 16 valid channels skips the eight MFMAs of its dead half, a column chunk with at most 16 valid columns the eight MFMAs of
// its dead column block -- a 48-channel layer executes 48 x 48, not 64 x 64 (the skipped MFMAs multiplied zeros: same bits).
// The (column chunk, K-chunk) slices then differ in cost, so the host hands every slice a share of the workgroups in
// proportion to its work instead of the uniform modulo mapping.  There are four classes of slices -- full, dead column block
// (last column chunk), dead K half (last K-chunk), both -- and a slice of a class gets g[class] workgroups; workgroups are
// numbered class by class: [full slices][column-tail slices][K-tail slices][the corner slice].
struct TsSlices { int g_full, g_ntail, g_ktail, g_both; };

// CHAIN (round 6): up to four DEPENDENT convolutions over the same tiles (the four SubM 3^3 launches of a level's two
// residual units, forward or backward-data: scn_conv_tiles_chain) as ONE launch of n_roles x wgs workgroups.  A workgroup
// draws a ticket (global atomic: roles are dealt in the order workgroups actually START, so a workgroup of role r exists
// only once every workgroup of the roles before it is running or done -- no assumption about dispatch order, no deadlock),
// role = ticket / wgs, and with it its operands.  Role r's workgroups are placed as role r-1's leave their CUs: they stage
// THEIR weight slice and fetch their first tile's mask and rows while role r-1's tail drains, then wait for
// done[r-1] == wgs (one lane polls, agent-scope acquire, workgroup barrier: MI350X_MICROARCH "Valid forms", consumer side).
// Producer side: every output store of a chained launch is write-through (sc1), every wave drains its stores, workgroup
// barrier, one agent-scope add on done[r].  The last workgroup of the last role puts the words back to zero.
// What it removes per link: the launch front (~4 us) and, on every CU but the last to finish, the 3.3 us of weight staging.
#define TS_MAX_ROLES 4
struct TsRole {
    const float* X; const float* W; const float* bias; const float* residual; const float* relu_mask; float* Y;
    int flags, pad;
};
struct TsChain {
    int n_roles, wgs;
    int exp, pad;           // (developer experiments, SCN_EXP_A: 1 no wait, 2 plain stores, 4 role = blockIdx / wgs, no ticket)
    int* sync;              // [0] ticket, [1 .. n_roles] done counters, [7] spin-limit flag; zero on entry, zero on exit
    TsRole role[TS_MAX_ROLES];
};

template <bool WT, bool VEC, bool VECN, bool FULLK, bool PART, bool FUSED, bool TAIL = false, int SPLIT = 1, bool CHAIN = false>
__global__ __launch_bounds__(TS_NW * 64) void k_conv_ts(
    const float* __restrict__ X_, long long n_in, int cin, const int* __restrict__ tstab,
    const unsigned* __restrict__ tile_mask,
    const int* __restrict__ perm, const int* __restrict__ tile_order, int n_off, long long nt,
    const float* __restrict__ W_, const float* __restrict__ bias_,
    const float* __restrict__ residual_, const float* __restrict__ relu_mask_, float* __restrict__ Y_,
    float* __restrict__ slabs, long long n_out, int cout, int flags_, int n_chunks, int n_kc,
    int* __restrict__ counters, TsSlices slices, TsChain chain) {
    extern __shared__ __attribute__((aligned(16))) float Ws[];           // [n_off][32 k][32 n] swizzled
    constexpr int THREADS = TS_NW * 64;
    const int tid = threadIdx.x, lane = tid & 63;
    int* counter = (int*)(Ws + n_off * TS_KC * TS_CT);                   // [0] tile queue, [1] ticket of a chained launch
    int bid = (int)blockIdx.x, n_bid = (int)gridDim.x, role = 0;
    if (tid == 0) {
        *counter = 0;
        counter[2] = 0;                                                  // PROG: waves whose later weight groups have landed
        if constexpr (CHAIN)
            counter[1] = (chain.exp & 4) ? (int)blockIdx.x : __hip_atomic_fetch_add(chain.sync, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if constexpr (CHAIN) {
        const int ticket = __builtin_amdgcn_readfirstlane(counter[1]);
        role = ticket / chain.wgs;
        bid = ticket - role * chain.wgs;
        n_bid = chain.wgs;
    }
    // operands of this workgroup's role (a plain launch: the kernel arguments)
    const float* __restrict__ X = CHAIN ? chain.role[role].X : X_;
    const float* __restrict__ W = CHAIN ? chain.role[role].W : W_;
    const float* __restrict__ bias = CHAIN ? chain.role[role].bias : bias_;
    const float* __restrict__ residual = CHAIN ? chain.role[role].residual : residual_;
    const float* __restrict__ relu_mask = CHAIN ? chain.role[role].relu_mask : relu_mask_;
    float* __restrict__ Y = CHAIN ? chain.role[role].Y : Y_;
    const int flags = CHAIN ? chain.role[role].flags : flags_;
    int chunk, kci, tg, n_tg;
    if constexpr (TAIL) {
        const int ncf = n_chunks - (slices.g_ntail ? 1 : 0), nkf = n_kc - (slices.g_ktail ? 1 : 0);   // full column / K chunks
        int b = (int)blockIdx.x;
        const int end_full = nkf * ncf * slices.g_full, end_n = end_full + nkf * slices.g_ntail;
        const int end_k = end_n + ncf * slices.g_ktail;
        if (b < end_full) {
            n_tg = slices.g_full;
            const int sl = b / n_tg;
            tg = b - sl * n_tg; kci = sl / ncf; chunk = sl - kci * ncf;
        } else if (b < end_n) {
            b -= end_full; n_tg = slices.g_ntail;
            kci = b / n_tg; tg = b - kci * n_tg; chunk = n_chunks - 1;
        } else if (b < end_k) {
            b -= end_n; n_tg = slices.g_ktail;
            chunk = b / n_tg; tg = b - chunk * n_tg; kci = n_kc - 1;
        } else {
            tg = b - end_k; n_tg = slices.g_both; chunk = n_chunks - 1; kci = n_kc - 1;
        }
    } else {
        chunk = bid % n_chunks;
        kci = (bid / n_chunks) % n_kc;
        tg = bid / (n_chunks * n_kc);
        n_tg = n_bid / (n_chunks * n_kc);
    }
    const int n0 = chunk * TS_CT, kc = kci * TS_KC;
    const bool kh = TAIL && cin - kc <= 16, nh = TAIL && cout - n0 <= 16;     // wave-uniform: dead K half / dead column block
    const bool relu_in = flags & SCN_F_RELU_IN;
    const bool rev = flags & SCN_F_OFF_REVERSE;
    const bool res_last = flags & SCN_F_RESIDUAL_LAST;

    // ---- tile queue: tile_order lists the tiles by offset count descending.  A workgroup owns every n_tg-th entry
    // (an even sample of costs) and its 16 waves pull them from an LDS counter in that order: expensive tiles start
    // first, cheap ones fill the tail (longest-processing-time scheduling).  (A chip-wide queue on global atomics was measured slower: returning global atomics cost more than
    // the imbalance they remove -- DESIGN.md.)
    long long tl_t0 = 0, tl_t1 = 0, tl_c1 = 0, tl_steps = 0, tl_tiles = 0;
    if (TS_TIMELINE) tl_t0 = wall_clock64();
    const int n_tiles = (int)((nt - tg + n_tg - 1) / n_tg);
    auto grab = [&]() -> long long {                    // -> tile id or -1
        int tl = 0;
        if (lane == 0) tl = atomicAdd(counter, 1);
        tl = __builtin_amdgcn_readfirstlane(tl);
        return tl < n_tiles ? tile_order[tg + (long long)tl * n_tg] : -1;
    };
    const int i = lane & 15, kq = lane >> 4;
    // PROG (round 6): PROGRESSIVE staging of the forward slice.  The 27 x 4 KB image is 108 pieces of 1 KB =
    // one LDS-DMA wave-instruction each (global_load_lds_dwordx4: lane-linear destination, the XOR-16 column swizzle applied
    // to the SOURCE address), dealt to the 16 waves in offset order, all in flight at once.  A wave waits only for its pieces
    // of the first g0 offsets before the workgroup barrier; the other pieces land while the first tile's index loads, row
    // gathers and first g0 steps run.  Loads retire in order, so the first row index a wave consumes proves that its own
    // DMA pieces have landed: it then adds one to an LDS word, and the first step that needs an offset >= g0 waits until that
    // word reads TS_NW (LDS serves its requests in order: a wave that read TS_NW reads the landed bytes).  The DMA
    // instructions are inline asm, invisible to hipcc's wait counting: they are the OLDEST vector-memory operations of the
    // wave (every counted load is issued after them), so hipcc's counted waits stay sufficient.
    constexpr bool PROG_OK = !WT && VECN && FULLK && !PART && !TAIL && SPLIT == 1 && !CHAIN && TS_DEPTH == 3 && !TS_TIMELINE_NOPROG;
    const int g0 = PROG_OK ? (flags_ >> 24) & 31 : 0;                   // offsets staged before the barrier; 0: classic staging
    bool w_ready = true, sig_pending = false;                           // wave-uniform
    // first tile: id, mask and output rows are requested before the weight slice is staged
    // (SPLIT > 1, scn_conv_ts_small.inc: tiles are dealt to wave groups by round, nothing to grab)
    long long tile_next = SPLIT == 1 ? grab() : -1;
    unsigned m_next = 0;
    int orow_next[4] = {-1, -1, -1, -1};
    if (PROG_OK && g0 > 0) {
        if (tile_next >= 0) m_next = __builtin_amdgcn_readfirstlane(tile_mask[tile_next]);   // consumed BEFORE the asm loads
        const int wv = tid >> 6, n_pieces = n_off * 4;
        const unsigned lds0 = (unsigned)(uintptr_t)Ws;
        const int p4 = (lane & 7) * 4, kr = lane >> 3;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            int c = wv + TS_NW * j;
            c = c < n_pieces ? c : n_pieces - 1;                        // (a wave with six pieces fetches its last one twice)
            const int o = c >> 2, k = 8 * (c & 3) + kr;
            const int wo = rev ? n_off - 1 - o : o;
            const float* src = W + ((long long)wo * cin + kc + k) * cout + n0 + (p4 ^ (((k >> 2) & 1) << 4));
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)c * 1024u);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
        }
        // pieces j < g0 / 4 hold the offsets below g0 (g0 is 4, 8 or 12)
        if (g0 <= 4) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (g0 <= 8) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        w_ready = false; sig_pending = true;
        if (tile_next >= 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) orow_next[j] = perm[tile_next * TS_T + 4 * kq + j];
        }
    } else {
    if (tile_next >= 0) {
        m_next = tile_mask[tile_next];
#pragma unroll
        for (int j = 0; j < 4; ++j) orow_next[j] = perm[tile_next * TS_T + 4 * kq + j];
    }

    // ---- stage the weight slice: 16-byte global reads, all in flight before the LDS writes ---------------------------
    {
        constexpr int SB = (27 * 256 + THREADS - 1) / THREADS;               // one batch covers 27 offsets
        const int total4 = n_off * TS_KC * (TS_CT / 4);
        for (int base = 0; base < total4; base += THREADS * SB) {
            float4 v[SB];
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int e = base + u * THREADS + tid;
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e < total4) {
                    const int c4 = e & 7, m = (e >> 3) & 31, o = e >> 8;
                    const int wo = rev ? n_off - 1 - o : o;
                    if (WT) {          // layer weight [o][n = m][k]: k contiguous
                        const int k = kc + 4 * c4, ng = n0 + m;
                        if (ng < cout) {
                            const float* src = W + ((long long)wo * cout + ng) * cin + k;
                            if (VEC && k + 3 < cin) v[u] = *(const float4*)src;
                            else {
                                if (k < cin) v[u].x = src[0];
                                if (k + 1 < cin) v[u].y = src[1];
                                if (k + 2 < cin) v[u].z = src[2];
                                if (k + 3 < cin) v[u].w = src[3];
                            }
                        }
                    } else {           // [o][k = m][n]: n contiguous
                        const int kg = kc + m, ng = n0 + 4 * c4;
                        if (kg < cin) {
                            const float* src = W + ((long long)wo * cin + kg) * cout + ng;
                            if (VECN && ng + 3 < cout) v[u] = *(const float4*)src;
                            else {
                                if (ng < cout) v[u].x = src[0];
                                if (ng + 1 < cout) v[u].y = src[1];
                                if (ng + 2 < cout) v[u].z = src[2];
                                if (ng + 3 < cout) v[u].w = src[3];
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int e = base + u * THREADS + tid;
                if (e < total4) {
                    const int c4 = e & 7, m = (e >> 3) & 31, o = e >> 8;
                    if (WT) {          // transpose into [k][n]
                        Ws[ts_ws_off(o, 4 * c4 + 0, m)] = v[u].x;
                        Ws[ts_ws_off(o, 4 * c4 + 1, m)] = v[u].y;
                        Ws[ts_ws_off(o, 4 * c4 + 2, m)] = v[u].z;
                        Ws[ts_ws_off(o, 4 * c4 + 3, m)] = v[u].w;
                    } else {
                        *(float4*)(Ws + ts_ws_off(o, m, 4 * c4)) = v[u];
                    }
                }
            }
        }
    }
    __syncthreads();
    }
    if constexpr (CHAIN) {
        // the role before this one has to be complete: its outputs are this role's X / residual / mask, and its K-split
        // slabs and arrival counters are this role's too.  One lane polls (relaxed, L1-bypassing), then the agent-scope
        // acquire (buffer_inv sc1) and its wait; the barrier holds the other waves until the invalidate is done.
        if (role > 0 && !(chain.exp & 1)) {
            if (tid == 0) {
                const int* done = chain.sync + role;
                int spins = 0;
                while (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < chain.wgs) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1 << 24)) {          // bounded: a lost workgroup must not hang the GPU (flag: results invalid)
                        __hip_atomic_store(chain.sync + 7, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
        }
    }
    if (TS_TIMELINE) { tl_t1 = wall_clock64(); tl_c1 = clock64(); }

    const int ka = kc + 4 * kq;
    const bool k0_ok = ka + 3 < cin, k1_ok = ka + 16 + 3 < cin;
    const int nA = n0 + i, nB = n0 + 16 + i;
    const bool single = !FUSED && n_kc == 1;            // (the host picks FUSED only for n_kc > 1)
    const bool direct = single || FUSED;                // this launch writes Y itself
    const float bA = (direct && bias && nA < cout) ? bias[nA] : 0.f;
    const float bB = (direct && bias && nB < cout) ? bias[nB] : 0.f;
    float* out = direct ? Y : slabs + (long long)kci * n_out * cout;
    const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)slabs, 0, FUSED && !single ? (int)(unsigned)(nt * n_chunks * n_kc * (TS_T * TS_CT * 4)) : 0, 0x00020000);
    // B fragment address of this lane inside an offset's slice (see ts_ws_off): channel 4*kq (+e, +16*half), column i
    const int bofs = (4 * kq) * TS_CT + (i ^ ((kq & 1) << 4));

    // A-row gather of one offset into two named registers (macro, not a lambda: the ping-pong below must keep the
    // two register sets apart by NAME).
    // FULLK: RAW BUFFER loads.  The buffer descriptor covers X exactly, so a row index of -1 (no rule) turns into a byte
    // offset past num_records and the hardware returns zeros: no select on the eight A values, no clamp of the index,
    // and the address is ONE v_mad_i32_i24 (row * row_bytes + lane offset) instead of a 64-bit multiply-add chain.
    // Every VALU instruction counts here: on this chip a VALU op takes ~4 cycles away from the MFMA pipe of the same
    // SIMD (tools/micro/step_skeleton.hip: 137 TFLOP/s with 0, 119 with 16, 99 with 64 dependent VALU ops per step).
    const __amdgpu_buffer_rsrc_t xrsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)(unsigned)(n_in * cin * 4), 0x00020000);
    const int row_bytes = cin * 4, lane_boff = ka * 4;
    // Epilogue operands (residual, ReLU-backward mask) through raw buffer descriptors too (FULLK): a missing operand is a
    // descriptor of zero records (every load returns 0 without touching memory), so the loads are UNCONDITIONAL and can be
    // issued when a tile STARTS -- they are back long before its epilogue, which used to be a dependent round trip per
    // tile (SQ counters, profiles/r2c_sq_counters.txt: the backward-data launches ran 12-14 % more cycles than the forward).
    const bool use_res = direct && residual != nullptr, use_mask = direct && relu_mask != nullptr;
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)residual, 0, use_res ? (int)(unsigned)(n_out * cout * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t mrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)relu_mask, 0, use_mask ? (int)(unsigned)(n_out * cout * 4) : 0, 0x00020000);
    const int out_row_bytes = cout * 4;
    const int colA_b = nA < cout ? nA * 4 : (int)0x7FFFFFF0, colB_b = nB < cout ? nB * 4 : (int)0x7FFFFFF0;
#define TS_LOAD_OPS(OROW, RS, MK)                                                                    \
    do {                                                                                             \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                           \
            const int ro_ = __mul24((OROW)[j_], out_row_bytes);          /* row -1 -> out of range */  \
            RS[j_][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrsrc, ro_ + colA_b, 0, 0)); \
            RS[j_][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrsrc, ro_ + colB_b, 0, 0)); \
            MK[j_][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(mrsrc, ro_ + colA_b, 0, 0)); \
            MK[j_][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(mrsrc, ro_ + colB_b, 0, 0)); \
        }                                                                                            \
    } while (0)
#define TS_GATHER(IDX, A0, A1)                                                                       \
    do {                                                                                             \
        if (TS_EXP >= 1) { A0 = (f32x4){(float)(IDX), 1.f, 2.f, 3.f}; A1 = A0; }                        \
        else if constexpr (FULLK) {                                                                       \
            const int off_ = __mul24((IDX), row_bytes) + lane_boff;      /* -1 -> out of range -> 0 */ \
            int off0_ = off_, off1_ = off_ + 64;                                                     \
            if constexpr (PART) {                                                                    \
                off0_ = k0_ok ? off0_ : (int)0xFFFFFFF0;                                             \
                off1_ = k1_ok ? off1_ : (int)0xFFFFFFF0;                                             \
            }                                                                                        \
            A0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, off0_, 0, 0)); \
            A1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, off1_, 0, 0)); \
        } else {                                                                                     \
            A0 = (f32x4){0.f, 0.f, 0.f, 0.f};                                                        \
            A1 = A0;                                                                                 \
            if ((IDX) >= 0) {                                                                        \
                const float* xp_ = X + (long long)(IDX) * cin + ka;                                  \
                if (VEC) {                                                                           \
                    if (k0_ok) A0 = *(const f32x4*)xp_;                                              \
                    if (k1_ok) A1 = *(const f32x4*)(xp_ + 16);                                       \
                } else {                                                                             \
                    if (ka < cin) A0[0] = xp_[0];                                                    \
                    if (ka + 1 < cin) A0[1] = xp_[1];                                                \
                    if (ka + 2 < cin) A0[2] = xp_[2];                                                \
                    if (ka + 3 < cin) A0[3] = xp_[3];                                                \
                    if (ka + 16 < cin) A1[0] = xp_[16];                                              \
                    if (ka + 17 < cin) A1[1] = xp_[17];                                              \
                    if (ka + 18 < cin) A1[2] = xp_[18];                                              \
                    if (ka + 19 < cin) A1[3] = xp_[19];                                              \
                }                                                                                    \
            }                                                                                        \
        }                                                                                            \
    } while (0)

    // one pipeline step: index of the offset 5 ahead (into INEW), A rows of the offset 3 ahead into (G0, G1) from the
    // index loaded TWO steps earlier (IOLD), MFMAs on (C0, C1).  Four named A register sets and four named index
    // registers rotate with the unrolled loop -- no register moves (a move of a value still in flight forces a wait, and
    // every VALU op costs MFMA time).  The index must not be consumed in the step after its load: loads retire in order,
    // so waiting for a one-step-old index also waited for the row gathers issued right after it (`s_waitcnt vmcnt(0)`
    // in every step: the 3-deep gather pipeline was only one deep).  With ~2 us of gather latency and 512 MFMA cycles per offset a SIMD needs ~9 gathers in
    // flight (4 waves x 3 here); one-ahead prefetch left the kernel latency-bound.
    // ZERO0: (non-FULLK only) the item multiplied now has rows without a rule -> handled in TS_GATHER (zero-filled).
#define TS_STEP(C0, C1, G0, G1, IOLD, INEW)                                                          \
    do {                                                                                             \
        if constexpr (PROG_OK) {                        /* first use of a weight group that may not have landed */ \
            if (!w_ready && oq0 >= g0) { ts_wait_groups(); w_ready = true; }                         \
        }                                                                                            \
        int o4_ = -1;                                                                                \
        if (m) { o4_ = __builtin_ctz(m); m &= m - 1; olast = o4_; }                                  \
        if (TS_EXP >= 2) INEW = olast + i; else                                                      \
        INEW = tb_s[olast * TS_T + i];                  /* scalar base + lane offset: no VALU */     \
        TS_GATHER(IOLD, G0, G1);                                                                     \
        f32x4 a0_ = C0, a1_ = C1;                                                                    \
        if (relu_in) {                                  /* ReLU as one integer max per value */      \
            _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) {                                       \
                a0_[e_] = __int_as_float(max(__float_as_int(a0_[e_]), 0));                           \
                a1_[e_] = __int_as_float(max(__float_as_int(a1_[e_]), 0));                           \
            }                                                                                        \
        }                                                                                            \
        const float* wb_ = Ws + (oq0 * (TS_KC * TS_CT) + bofs);                                      \
        const float* wc_ = Ws + (oq0 * (TS_KC * TS_CT) + (bofs ^ 16));                               \
        float bl_[8], bh_[8];                                                                        \
        /* TS_KH / TS_NH (compile-time per copy of the tile loop): dead K half / dead column block skipped; the order of   \
           the MFMAs that remain is unchanged per accumulator */                                                          \
        _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) {                                           \
            bl_[e_] = wb_[e_ * TS_CT];                                                               \
            if (!TS_NH) bh_[e_] = wc_[e_ * TS_CT];                                                   \
            if (!TS_KH) bl_[4 + e_] = wb_[(16 + e_) * TS_CT];                                        \
            if (!TS_KH && !TS_NH) bh_[4 + e_] = wc_[(16 + e_) * TS_CT];                              \
        }                                                                                            \
        _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) {                                           \
            c0 = MFMA16(a0_[e_], bl_[e_], c0);                                                       \
            if (!TS_NH) c1 = MFMA16(a0_[e_], bh_[e_], c1);                                           \
        }                                                                                            \
        if (!TS_KH) {                                                                                \
            _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) {                                       \
                c0 = MFMA16(a1_[e_], bl_[4 + e_], c0);                                               \
                if (!TS_NH) c1 = MFMA16(a1_[e_], bh_[4 + e_], c1);                                   \
            }                                                                                        \
        }                                                                                            \
        oq0 = oq1; oq1 = oq2; oq2 = oq3; oq3 = oq4;                                                  \
        if (TS_DEPTH == 3) oq4 = o4_; else ts_o4_ = o4_;     /* (depth 5: the caller shifts the longer queue) */ \
    } while (0)

    // PROG: the word the waves count themselves into / wait on
    auto ts_signal = [&]() {
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");               // all but the six youngest (the prologue's row gathers)
        if (lane == 0) atomicAdd(counter + 2, 1);
    };
    auto ts_wait_groups = [&]() {                      // (asm: a volatile read through the generic pointer compiles to flat_load)
        const unsigned fa = (unsigned)(uintptr_t)(counter + 2);
        int f;
        do {
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(f) : "v"(fa) : "memory");
            if (__builtin_amdgcn_readfirstlane(f) >= TS_NW) break;
            __builtin_amdgcn_s_sleep(1);
        } while (true);
    };
    // ---- epilogue pieces --------------------------------------------------------------------------------------------
    // ts_write: 16 lanes write 64 contiguous bytes of a row.  The residual / ReLU-mask operands of all eight outputs are
    // requested first and consumed afterwards (one wait, not eight round trips).
    auto put = [&](float* p, float y) {                 // an output element: write-through in a chained launch (hand-off)
        if constexpr (CHAIN) {
            if (chain.exp & 2) *p = y;
            else __hip_atomic_store(p, y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else *p = y;
    };
    auto ts_write = [&](const int (&orow)[4], const f32x4& c0, const f32x4& c1) {
        float rs[4][2], mk[4][2];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long long off = (long long)(orow[j] < 0 ? 0 : orow[j]) * cout;
            rs[j][0] = rs[j][1] = 0.f;
            mk[j][0] = mk[j][1] = 1.f;
            if (use_res) {
                if (nA < cout) rs[j][0] = residual[off + nA];
                if (nB < cout) rs[j][1] = residual[off + nB];
            }
            if (use_mask) {
                if (nA < cout) mk[j][0] = relu_mask[off + nA];
                if (nB < cout) mk[j][1] = relu_mask[off + nB];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = orow[j];
            if (row < 0) continue;
            const long long off = (long long)row * cout;
            if (nA < cout) {
                float y = c0[j] + (res_last ? 0.f : rs[j][0]);
                if (!(mk[j][0] > 0.f)) y = 0.f;
                if (res_last) y += rs[j][0];
                put(out + off + nA, y);
            }
            if (nB < cout) {
                float y = c1[j] + (res_last ? 0.f : rs[j][1]);
                if (!(mk[j][1] > 0.f)) y = 0.f;
                if (res_last) y += rs[j][1];
                put(out + off + nB, y);
            }
        }
    };
    auto ts_store = [&](const int (&orow)[4], const f32x4& c0, const f32x4& c1, const float (&rs)[4][2],
                        const float (&mk)[4][2]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = orow[j];
            if (row < 0) continue;
            const long long off = (long long)row * cout;
            if (nA < cout) {
                float y = c0[j] + (res_last ? 0.f : rs[j][0]);
                if (use_mask && !(mk[j][0] > 0.f)) y = 0.f;
                if (res_last) y += rs[j][0];
                put(out + off + nA, y);
            }
            if (nB < cout) {
                float y = c1[j] + (res_last ? 0.f : rs[j][1]);
                if (use_mask && !(mk[j][1] > 0.f)) y = 0.f;
                if (res_last) y += rs[j][1];
                put(out + off + nB, y);
            }
        }
    };
    // in-launch K reduction (FUSED): q1 = tile whose partial is published and not yet ticketed, q2 = tile whose ticket is
    // in flight (tk_v holds it in lane 0); -1 = none.  Callers drain the wave's memory operations first.
    long long q1 = -1, q2 = -1;
    int tk_v = 0;
    int ts_o4_ = -1;                                   // (TS_DEPTH 5: the offset a step popped, handed to the longer queue)
    int orow_q1[4] = {-1, -1, -1, -1}, orow_q2[4] = {-1, -1, -1, -1};     // output rows of the tiles q1 / q2 (FULLK)
    auto ts_publish = [&](long long t, const f32x4& c0, const f32x4& c1) {
        const int sb = (int)((t * n_chunks + chunk) * n_kc + kci) * (TS_T * TS_CT * 4) + lane * 16;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, c0), srsrc,
                                               sb, 0, 16);                               // aux 16 = sc1: write-through
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, c1), srsrc,
                                               sb + 1024, 0, 16);
        q1 = t;
    };
    auto ts_retire = [&]() {
        // (a) the ticket of q2 has returned; (b) the stores of q1 are drained: its ticket goes out FIRST, so that its round
        // trip runs beside the combine of q2 instead of behind it (the two flush rounds behind a wave's last tile are a
        // serial chain at the end of the launch); (c) combine q2 if this wave arrived last
        const int ticket = __builtin_amdgcn_readfirstlane(tk_v);
        if (q1 >= 0 && lane == 0)
            tk_v = __hip_atomic_fetch_add(counters + (int)(q1 * n_chunks + chunk), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (q2 >= 0) {
            if (ticket == n_kc - 1) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                    // compiler ordering only
                const int unit = (int)(q2 * n_chunks + chunk);
                if (lane == 0) __hip_atomic_store(counters + unit, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int orow2[4];
                float rs2[4][2], mk2[4][2];
                if constexpr (FULLK) {          // rows kept from the tile's own pass; operands requested with the partials
#pragma unroll
                    for (int j = 0; j < 4; ++j) orow2[j] = orow_q2[j];
                    TS_LOAD_OPS(orow2, rs2, mk2);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) orow2[j] = perm[q2 * TS_T + 4 * kq + j];
                }
                const int sb = (unit * n_kc) * (TS_T * TS_CT * 4) + lane * 16;
                // bias + slab[0] + slab[1] + ... in ascending K-chunk order (the association of k_conv_ts_sum)
                f32x4 y0 = {bA, bA, bA, bA}, y1 = {bB, bB, bB, bB};
                for (int k0 = 0; k0 < n_kc; k0 += 4) {
                    f32x4 p0[4], p1[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int k = k0 + u < n_kc ? k0 + u : n_kc - 1;                  // unconditional loads
                        p0[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srsrc, sb + k * 2048, 0, 16));
                        p1[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srsrc, sb + k * 2048 + 1024, 0, 16));
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (k0 + u < n_kc) { y0 += p0[u]; y1 += p1[u]; }
                    }
                }
                if constexpr (FULLK) ts_store(orow2, y0, y1, rs2, mk2);
                else ts_write(orow2, y0, y1);
            }
        }
        q2 = q1;
        q1 = -1;
#pragma unroll
        for (int j = 0; j < 4; ++j) orow_q2[j] = orow_q1[j];
    };

    // ---- tile loop: the id, mask and output rows of the NEXT tile are fetched while the current one computes --------
    if constexpr (TAIL) {
        if (kh && nh) {
#define TS_KH 1
#define TS_NH 1
if constexpr (SPLIT == 1) {
#include "scn_conv_ts_loop.inc"
        } else {
#include "scn_conv_ts_small.inc"
        }
#undef TS_KH
#undef TS_NH
        } else if (kh) {
#define TS_KH 1
#define TS_NH 0
if constexpr (SPLIT == 1) {
#include "scn_conv_ts_loop.inc"
        } else {
#include "scn_conv_ts_small.inc"
        }
#undef TS_KH
#undef TS_NH
        } else if (nh) {
#define TS_KH 0
#define TS_NH 1
if constexpr (SPLIT == 1) {
#include "scn_conv_ts_loop.inc"
        } else {
#include "scn_conv_ts_small.inc"
        }
#undef TS_KH
#undef TS_NH
        } else {
#define TS_KH 0
#define TS_NH 0
if constexpr (SPLIT == 1) {
#include "scn_conv_ts_loop.inc"
        } else {
#include "scn_conv_ts_small.inc"
        }
#undef TS_KH
#undef TS_NH
        }
    } else {
#define TS_KH 0
#define TS_NH 0
if constexpr (SPLIT == 1) {
#include "scn_conv_ts_loop.inc"
        } else {
#include "scn_conv_ts_small.inc"
        }
#undef TS_KH
#undef TS_NH
    }
    if constexpr (PROG_OK) {
        if (sig_pending) {                   // a wave without a tile still owes its arrival
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) atomicAdd(counter + 2, 1);
        }
    }
    if (FUSED && !single) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ts_retire();                         // combine the tile before last, ticket the last one
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ts_retire();                         // combine the last one
    }
    if constexpr (CHAIN) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every wave: its write-through stores have landed
        __syncthreads();
        if (tid == 0) {
            const int t = __hip_atomic_fetch_add(chain.sync + 1 + role, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (role == chain.n_roles - 1 && t == chain.wgs - 1) {    // the very last workgroup: nobody polls any more
                for (int k = 0; k <= chain.n_roles; ++k)
                    __hip_atomic_store(chain.sync + k, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (TS_TIMELINE && lane == 0) {         // [t0, staged, end, tiles, steps] per wave, after the K-chunk slabs
        long long* d = (long long*)(slabs + (n_kc > 1 ? (long long)n_kc * nt * TS_T * n_chunks * TS_CT : 0)) +
                       ((long long)blockIdx.x * TS_NW + (tid >> 6)) * 8;
        d[0] = tl_t0; d[1] = tl_t1; d[2] = wall_clock64(); d[3] = tl_tiles; d[4] = tl_steps; d[5] = 0;
        d[6] = tl_c1; d[7] = clock64();             // shader-clock stamps: in-kernel clock = d(clock64) / d(wall_clock64) x 100 MHz
    }
}

// Y = bias + sum_kc slab[kc] (+ residual, ReLU-backward mask); K-chunks added in ascending order.  V = 4: 16-byte
// accesses (cout % 4 == 0, aligned buffers) -- the kernel is pure streaming traffic, (n_kc + 1..3) * N * cout * 4 bytes.
template <int V>
__global__ void k_conv_ts_sum(const float* __restrict__ slabs, int n_kc, long long n_out, int cout,
                              const float* __restrict__ bias, const float* __restrict__ residual,
                              const float* __restrict__ relu_mask, float* __restrict__ Y, int res_last) {
    typedef float vec_t __attribute__((ext_vector_type(V)));
    const long long total = n_out * cout / V;
    for (long long g = blockIdx.x * (long long)blockDim.x + threadIdx.x; g < total;
         g += (long long)gridDim.x * blockDim.x) {
        const long long e = g * V;
        vec_t y;
        if (bias) y = *(const vec_t*)(bias + e % cout);
        else {
#pragma unroll
            for (int v = 0; v < V; ++v) y[v] = 0.f;
        }
        for (int k = 0; k < n_kc; ++k) y += *(const vec_t*)(slabs + (long long)k * n_out * cout + e);
        vec_t r, mk;
        if (residual) r = *(const vec_t*)(residual + e);
        if (relu_mask) mk = *(const vec_t*)(relu_mask + e);
        if (residual && !res_last) y += r;
        if (relu_mask) {
#pragma unroll
            for (int v = 0; v < V; ++v) if (!(mk[v] > 0.f)) y[v] = 0.f;
        }
        if (residual && res_last) y += r;
        *(vec_t*)(Y + e) = y;
    }
}

#undef TS_STEP
#undef TS_GATHER
#undef TS_LOAD_OPS

// scratch = [256 reserved bytes] [slabs if n_kc > 1: n_kc partial tiles of 16 x 32 floats per (tile, column chunk)]
// (covers both slab layouts: the fused kernel's fragment-major one and the row-major one of the two-launch form)
static int64_t ts_counter_bytes(int, int) { return 256; }

extern "C" int64_t scn_conv_tiles_scratch_bytes(int cin, int64_t n_out, int cout) {
    const int64_t n_kc = cdiv(cin, TS_KC);
    return ts_counter_bytes(cin, cout) +
           (n_kc > 1 ? n_kc * cdiv(n_out, TS_T) * TS_T * cdiv(cout, TS_CT) * TS_CT * (int64_t)sizeof(float) : 0);
}

static std::atomic<int64_t> g_ts_paths[4];
static std::atomic<int64_t> g_ts_split;            // launches of the four-waves-per-tile loop (scn_conv_ts_small.inc)

extern "C" int64_t scn_conv_tiles_split_count(int reset) { return reset ? g_ts_split.exchange(0) : g_ts_split.load(); }

extern "C" void scn_conv_tiles_path_counts(int64_t out[4], int reset) {
    for (int i = 0; i < 4; ++i) out[i] = reset ? g_ts_paths[i].exchange(0) : g_ts_paths[i].load();
}

extern "C" int64_t scn_conv_tiles_arrival_counters(int cin, int64_t n_out, int cout) {
    return cdiv(cin, TS_KC) > 1 ? cdiv(n_out, TS_T) * cdiv(cout, TS_CT) : 0;
}

namespace scn {
int conv_tiles_stream(const float* X, int64_t n_in, int cin, const int32_t* tstab, const uint32_t* tile_mask,
                      const int32_t* perm, int n_off, int64_t n_out, const float* W, const float* bias, const float* residual,
                      const float* relu_mask, float* Y, int cout, int flags, hipStream_t st, bool* launched);
}

extern "C" int scn_conv_tiles(const float* X, int64_t n_in, int cin, const int32_t* tstab, const uint32_t* tile_mask,
                              const int32_t* perm, const int32_t* tile_order, int n_off, int64_t n_out, const float* W, const float* bias,
                              const float* residual, const float* relu_mask, float* Y, int cout, int flags,
                              void* scratch, int32_t* arrival, scn_stream_t stream) {
    SCN_REQUIRE(n_off >= 1 && n_off <= 27 && n_out >= 0 && n_in >= 0 && cin >= 1 && cout >= 1);
    if (n_out == 0) return SCN_OK;
    SCN_REQUIRE(X && tstab && tile_mask && perm && tile_order && W && Y);
    const int64_t nt = cdiv(n_out, TS_T);
    const int n_chunks = (int)cdiv(cout, TS_CT);
    const int n_kc = (int)cdiv(cin, TS_KC);
    SCN_REQUIRE(scratch);
    if (n_kc > 1 && arrival != nullptr) {        // (the fused mode's callers; the two-launch cross-check keeps k_conv_ts)
        // layers that would split K over workgroups at Cin = 64 / 128: the offset-outer weight-streaming kernel (scn_conv_tss.hip)
        bool launched = false;
        const int rc = scn::conv_tiles_stream(X, n_in, cin, tstab, tile_mask, perm, n_off, n_out, W, bias, residual, relu_mask,
                                              Y, cout, flags, S(stream), &launched);
        if (rc != SCN_OK) return rc;
        if (launched) {
            g_ts_paths[0].fetch_add(1, std::memory_order_relaxed);
            g_ts_paths[2].fetch_add(1, std::memory_order_relaxed);
            return SCN_OK;
        }
    }
    float* slabs = (float*)((char*)scratch + ts_counter_bytes(cin, cout));
    // the in-launch K reduction needs the caller's zeroed arrival counters and 32-bit slab offsets; without them (or when
    // the caller asks for the two-launch form) the partial sums go to row-major slabs and k_conv_ts_sum adds them
    const bool fused = n_kc > 1 && arrival != nullptr && !(flags & SCN_F_SPLIT_SUM) &&
                       nt * n_chunks * n_kc * (int64_t)(TS_T * TS_CT * 4) < (1ll << 31);
    int* counters = (int*)arrival;
    const bool wt = flags & SCN_F_W_TRANSPOSED;
    const bool vec = (cin % 4 == 0) && (((uintptr_t)X & 15) == 0) && (((uintptr_t)W & 15) == 0);
    const bool vecn = (cout % 4 == 0) && (((uintptr_t)W & 15) == 0);
    // the fast path addresses X through a raw buffer descriptor: 32-bit byte offsets, 24-bit row indices
    const bool fullk = vec && n_in < (1ll << 23) && n_in * cin * 4 < (1ll << 32) - (1ll << 24) &&
                       n_out < (1ll << 23) && n_out * cout * 4 < (1ll << 32) - (1ll << 24);
    // Latency-bound levels (scn_conv_ts_small.inc): with fewer (tile, slice) pairs than half the chip's waves a launch is ONE
    // tile's serial chain of up to 27 dependent gathers, so four waves share a tile.  Decided from the tile and slice counts
    // alone: every form of a layer (fused / two-launch K reduction, TAIL / padded slices) takes the same loop.
    // SCN_TS_SPLIT=0: never (A/B; the cross-check in the tests -- the two loops associate an element's sum differently).
    const scn::SwitchVal sp_sw = scn::sw(scn::SW_TS_SPLIT);            // (scn_debug_set: the tests switch it inside one process)
    const int64_t sp_max = scn::sw(scn::SW_TS_SPLIT_MAX).set ? scn::sw(scn::SW_TS_SPLIT_MAX).i : 2048;   // (developer switch)
    const bool split4 = fullk && nt * n_chunks * n_kc <= sp_max && !(sp_sw.set && sp_sw.i == 0);
    const int tiles_per_round = split4 ? TS_NW / 4 : TS_NW;
    const size_t lds = (size_t)n_off * TS_KC * TS_CT * sizeof(float) + 16 + (split4 ? (size_t)2 * (TS_NW / 4) * 3 * 2 * 64 * 16 : 0);   // two buffers of partial tiles
    int wg_per_cu = (int)((160 * 1024) / lds);
    if (wg_per_cu > 2) wg_per_cu = 2;                  // 16 waves each: 2 workgroups fill a CU
    if (wg_per_cu < 1) wg_per_cu = 1;
    // grid sized to the chip: workgroups per (column chunk, K-chunk) slice; a workgroup should see at least ~16 tiles
    int64_t n_tg = ((int64_t)scn::cu_budget() * wg_per_cu) / ((int64_t)n_chunks * n_kc);
    if (n_tg > cdiv(nt, tiles_per_round)) n_tg = cdiv(nt, tiles_per_round);
    if (n_tg < 1) n_tg = 1;
    const bool part = cin % TS_KC != 0;
    // TAIL: a dead K half or a dead column block exists (the reference's 48 / 80 / 112-channel layers): those MFMAs are
    // skipped and the slices get workgroups in proportion to their cost.  Estimated cost of a tile step relative to a
    // full 16-MFMA step: one dead half 0.70, both 0.50 (the per-step fixed work -- index load, gathers, ~15 VALU -- does
    // not shrink with the MFMAs; calibrated with tools/ablate_conv_tail.py on the 48 / 80 / 112-channel layers:
    // profiles/r3_ablate_conv_tail.txt).
    const bool k_tail = cin % TS_KC != 0 && cin % TS_KC <= 16, n_tail = cout % TS_CT != 0 && cout % TS_CT <= 16;
    const bool tail = fullk && (k_tail || n_tail) && !scn::sw(scn::SW_TS_NO_TAIL).set;
    TsSlices slices = {0, 0, 0, 0};
    int64_t grid_x = n_tg * n_chunks * n_kc;
    if (tail) {
        const double w_half = scn::sw(scn::SW_TS_W_HALF).set ? scn::sw(scn::SW_TS_W_HALF).f : 0.70;
        const double w_both = scn::sw(scn::SW_TS_W_BOTH).set ? scn::sw(scn::SW_TS_W_BOTH).f : 0.50;
        const int ncf = n_chunks - (n_tail ? 1 : 0), nkf = n_kc - (k_tail ? 1 : 0);
        const double wsum = (double)nkf * ncf + (n_tail ? nkf * w_half : 0.0) + (k_tail ? ncf * w_half : 0.0) +
                            (n_tail && k_tail ? w_both : 0.0);
        const int64_t total_wg = (int64_t)scn::cu_budget() * wg_per_cu, cap = cdiv(nt, tiles_per_round);   // a workgroup should see at least ~16 tiles
        auto share = [&](double w) { int64_t g = (int64_t)(total_wg * w / wsum); return (int)(g > cap ? cap : (g < 1 ? 1 : g)); };
        slices.g_full = share(1.0);
        slices.g_ntail = n_tail ? share(w_half) : 0;
        slices.g_ktail = k_tail ? share(w_half) : 0;
        slices.g_both = n_tail && k_tail ? share(w_both) : 0;
        grid_x = (int64_t)nkf * ncf * slices.g_full + (int64_t)nkf * slices.g_ntail + (int64_t)ncf * slices.g_ktail + slices.g_both;
    }
    // progressive staging of the forward slice (PROG in the kernel): SCN_TS_PROG = offsets staged before the barrier (4 / 8 / 12;
    // 0 = the classic staging)
    int prog_g0 = 0;
    if (fullk && !wt && vecn && !part && !tail && !split4 && n_off == 27 && cout % TS_CT == 0) {
        const scn::SwitchVal pg = scn::sw(scn::SW_TS_PROG);
        prog_g0 = pg.set ? (int)pg.i : 0;
        if (prog_g0 < 0 || prog_g0 > 12) prog_g0 = 0;
        prog_g0 = (prog_g0 + 3) / 4 * 4;
    }
    const int kflags = (flags & 0xFFFFFF) | (prog_g0 << 24);
    g_ts_paths[fullk ? 0 : 1].fetch_add(1, std::memory_order_relaxed);
    if (n_kc > 1) g_ts_paths[fused ? 2 : 3].fetch_add(1, std::memory_order_relaxed);
    if (split4) g_ts_split.fetch_add(1, std::memory_order_relaxed);
    dim3 grid((unsigned)grid_x);
    hipStream_t st = S(stream);
#define LAUNCH_TS_F(T, V, VN, FK, PT, FU, TL, SP)                                                                   \
    do {                                                                                                            \
        static scn::DeviceOnce attr_set;                                                                               \
        if (attr_set.needed()) {                                                                                            \
            SCN_HIP(hipFuncSetAttribute((const void*)k_conv_ts<T, V, VN, FK, PT, FU, TL, SP>,                       \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                   \
            attr_set.done();                                                                                        \
        }                                                                                                           \
        hipLaunchKernelGGL((k_conv_ts<T, V, VN, FK, PT, FU, TL, SP>), grid, dim3(TS_NW * 64), lds, st, X, (long long)n_in, cin, tstab, tile_mask, \
                           perm, tile_order, n_off, (long long)nt, W, bias, residual, relu_mask, Y, slabs,                      \
                           (long long)n_out, cout, kflags, n_chunks, n_kc, counters, slices, TsChain{});             \
    } while (0)
#define LAUNCH_TS(T, V, VN, FK, PT)                                                                                 \
    do {                                                                                                            \
        if (fused) LAUNCH_TS_F(T, V, VN, FK, PT, true, false, 1); else LAUNCH_TS_F(T, V, VN, FK, PT, false, false, 1); \
    } while (0)
    // the raw-buffer fast path (FULLK), plain or TAIL slices, one tile per wave or four waves per tile
#define LAUNCH_TS_FK(T, VN, PT, TL)                                                                                 \
    do {                                                                                                            \
        if (split4) {                                                                                               \
            if (fused) LAUNCH_TS_F(T, true, VN, true, PT, true, TL, 4); else LAUNCH_TS_F(T, true, VN, true, PT, false, TL, 4); \
        } else {                                                                                                    \
            if (fused) LAUNCH_TS_F(T, true, VN, true, PT, true, TL, 1); else LAUNCH_TS_F(T, true, VN, true, PT, false, TL, 1); \
        }                                                                                                           \
    } while (0)
    if (tail && wt && part) LAUNCH_TS_FK(true, true, true, true);
    else if (tail && wt) LAUNCH_TS_FK(true, true, false, true);
    else if (tail && vecn && part) LAUNCH_TS_FK(false, true, true, true);
    else if (tail && vecn) LAUNCH_TS_FK(false, true, false, true);
    else if (tail && part) LAUNCH_TS_FK(false, false, true, true);
    else if (tail) LAUNCH_TS_FK(false, false, false, true);
    else if (fullk && wt && part) LAUNCH_TS_FK(true, true, true, false);
    else if (fullk && wt) LAUNCH_TS_FK(true, true, false, false);
    else if (fullk && vecn && part) LAUNCH_TS_FK(false, true, true, false);
    else if (fullk && vecn) LAUNCH_TS_FK(false, true, false, false);
    else if (fullk && part) LAUNCH_TS_FK(false, false, true, false);
    else if (fullk) LAUNCH_TS_FK(false, false, false, false);
    else if (wt) LAUNCH_TS(true, false, true, false, false);
    else if (vecn) LAUNCH_TS(false, false, true, false, false);
    else LAUNCH_TS(false, false, false, false, false);
#undef LAUNCH_TS
#undef LAUNCH_TS_FK
#undef LAUNCH_TS_F
    SCN_LAUNCH_CHECK();
    if (fused || (flags & SCN_F_SPLIT_SUM)) return SCN_OK;
    return scn_conv_tiles_finish(cin, n_out, bias, residual, relu_mask, Y, cout, flags, scratch, stream);
}

// ---- chained launch (round 6) -------------------------------------------------------------------------------------------
// Sync words of a chained launch: 8 ints per (device, stream), zero between launches (the kernel's last workgroup resets
// them); allocated once, never freed (a process has a handful of streams).
#include <map>
#include <mutex>
namespace {
std::mutex g_chain_mu;
std::map<std::pair<int, hipStream_t>, int*> g_chain_sync;
int chain_sync_words(hipStream_t st, int** out) {
    int dev = 0;
    SCN_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_chain_mu);
    auto key = std::make_pair(dev, st);
    auto it = g_chain_sync.find(key);
    if (it == g_chain_sync.end()) {
        int* p = nullptr;
        SCN_HIP(hipMalloc(&p, 64));
        SCN_HIP(hipMemsetAsync(p, 0, 64, st));
        it = g_chain_sync.emplace(key, p).first;
    }
    *out = it->second;
    return SCN_OK;
}
std::atomic<int64_t> g_ts_chain_launches, g_ts_chain_roles;
}  // namespace

extern "C" void scn_conv_tiles_chain_counts(int64_t out[2], int reset) {
    out[0] = reset ? g_ts_chain_launches.exchange(0) : g_ts_chain_launches.load();
    out[1] = reset ? g_ts_chain_roles.exchange(0) : g_ts_chain_roles.load();
}

// 1 if scn_conv_tiles_chain would run these shapes as ONE launch (else it runs its roles as separate scn_conv_tiles calls)
static bool ts_chainable(int n_roles, int64_t n_in, int cin, int n_off, int64_t n_out, int cout, const scn_conv_role* roles) {
    if (n_roles < 2 || n_roles > TS_MAX_ROLES) return false;
    if (scn::sw(scn::SW_TS_NO_CHAIN).set && scn::sw(scn::SW_TS_NO_CHAIN).i != 0) return false;
    if (cin % TS_KC != 0 || cout % TS_CT != 0) return false;                       // no PART / TAIL slices
    const int64_t nt = cdiv(n_out, TS_T);
    const int n_chunks = (int)cdiv(cout, TS_CT), n_kc = (int)cdiv(cin, TS_KC);
    if (n_in >= (1ll << 23) || n_in * cin * 4 >= (1ll << 32) - (1ll << 24) || n_out >= (1ll << 23) ||
        n_out * cout * 4 >= (1ll << 32) - (1ll << 24))
        return false;                                                              // FULLK only
    if (n_kc > 1 && nt * n_chunks * n_kc * (int64_t)(TS_T * TS_CT * 4) >= (1ll << 31)) return false;
    const int64_t sp_max = scn::sw(scn::SW_TS_SPLIT_MAX).set ? scn::sw(scn::SW_TS_SPLIT_MAX).i : 2048;
    const scn::SwitchVal sp_sw = scn::sw(scn::SW_TS_SPLIT);
    if (nt * n_chunks * n_kc <= sp_max && !(sp_sw.set && sp_sw.i == 0)) return false;   // the four-waves-per-tile loop
    if ((size_t)n_off * TS_KC * TS_CT * sizeof(float) + 16 > 160 * 1024) return false;
    const int f0 = roles[0].flags & (SCN_F_W_TRANSPOSED | SCN_F_OFF_REVERSE);
    for (int r = 0; r < n_roles; ++r) {
        if ((roles[r].flags & (SCN_F_W_TRANSPOSED | SCN_F_OFF_REVERSE)) != f0) return false;
        if (roles[r].flags & SCN_F_SPLIT_SUM) return false;
        if (!roles[r].X || !roles[r].W || !roles[r].Y) return false;
        if ((((uintptr_t)roles[r].X | (uintptr_t)roles[r].W) & 15) != 0) return false;
    }
    return true;
}

extern "C" int scn_conv_tiles_chain(int n_roles, const scn_conv_role* roles, int64_t n_in, int cin, const int32_t* tstab,
                                    const uint32_t* tile_mask, const int32_t* perm, const int32_t* tile_order, int n_off,
                                    int64_t n_out, int cout, void* scratch, int32_t* arrival, scn_stream_t stream) {
    SCN_REQUIRE(n_roles >= 1 && roles != nullptr);
    SCN_REQUIRE(n_off >= 1 && n_off <= 27 && n_out >= 0 && n_in >= 0 && cin >= 1 && cout >= 1);
    if (n_out == 0) return SCN_OK;
    const int n_kc = (int)cdiv(cin, TS_KC);
    if (!ts_chainable(n_roles, n_in, cin, n_off, n_out, cout, roles) || (n_kc > 1 && arrival == nullptr)) {
        for (int r = 0; r < n_roles; ++r) {
            const int rc = scn_conv_tiles((const float*)roles[r].X, n_in, cin, tstab, tile_mask, perm, tile_order, n_off, n_out,
                                          (const float*)roles[r].W, (const float*)roles[r].bias, (const float*)roles[r].residual,
                                          (const float*)roles[r].relu_mask, (float*)roles[r].Y, cout, roles[r].flags, scratch,
                                          arrival, stream);
            if (rc != SCN_OK) return rc;
        }
        return SCN_OK;
    }
    SCN_REQUIRE(tstab && tile_mask && perm && tile_order && scratch);
    const int64_t nt = cdiv(n_out, TS_T);
    const int n_chunks = (int)cdiv(cout, TS_CT);
    hipStream_t st = S(stream);
    int* sync = nullptr;
    { const int rc = chain_sync_words(st, &sync); if (rc != SCN_OK) return rc; }
    float* slabs = (float*)((char*)scratch + ts_counter_bytes(cin, cout));
    const bool fused = n_kc > 1;
    const bool wt = roles[0].flags & SCN_F_W_TRANSPOSED;
    const size_t lds = (size_t)n_off * TS_KC * TS_CT * sizeof(float) + 16;
    int wg_per_cu = (int)((160 * 1024) / lds);
    if (wg_per_cu > 2) wg_per_cu = 2;
    if (wg_per_cu < 1) wg_per_cu = 1;
    int64_t n_tg = ((int64_t)scn::cu_budget() * wg_per_cu) / ((int64_t)n_chunks * n_kc);
    if (n_tg > cdiv(nt, TS_NW)) n_tg = cdiv(nt, TS_NW);
    if (n_tg < 1) n_tg = 1;
    TsChain chain{};
    chain.n_roles = n_roles;
    chain.exp = scn::sw(scn::SW_EXP_A).set ? (int)scn::sw(scn::SW_EXP_A).i : 0;
    chain.wgs = (int)(n_tg * n_chunks * n_kc);
    chain.sync = sync;
    for (int r = 0; r < n_roles; ++r) {
        chain.role[r].X = (const float*)roles[r].X; chain.role[r].W = (const float*)roles[r].W;
        chain.role[r].bias = (const float*)roles[r].bias; chain.role[r].residual = (const float*)roles[r].residual;
        chain.role[r].relu_mask = (const float*)roles[r].relu_mask; chain.role[r].Y = (float*)roles[r].Y;
        chain.role[r].flags = roles[r].flags;
    }
    g_ts_paths[0].fetch_add(n_roles, std::memory_order_relaxed);
    if (n_kc > 1) g_ts_paths[2].fetch_add(n_roles, std::memory_order_relaxed);
    g_ts_chain_launches.fetch_add(1, std::memory_order_relaxed);
    g_ts_chain_roles.fetch_add(n_roles, std::memory_order_relaxed);
    dim3 grid((unsigned)((int64_t)chain.wgs * n_roles));
    TsSlices slices = {0, 0, 0, 0};
#define LAUNCH_TS_C(T, FU)                                                                                          \
    do {                                                                                                            \
        static scn::DeviceOnce attr_set;                                                                               \
        if (attr_set.needed()) {                                                                                            \
            SCN_HIP(hipFuncSetAttribute((const void*)k_conv_ts<T, true, true, true, false, FU, false, 1, true>,     \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                   \
            attr_set.done();                                                                                        \
        }                                                                                                           \
        hipLaunchKernelGGL((k_conv_ts<T, true, true, true, false, FU, false, 1, true>), grid, dim3(TS_NW * 64), lds, st,        \
                           (const float*)nullptr, (long long)n_in, cin, tstab, tile_mask, perm, tile_order, n_off, (long long)nt, \
                           (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,          \
                           (float*)nullptr, slabs, (long long)n_out, cout, 0, n_chunks, n_kc, (int*)arrival, slices, chain);    \
    } while (0)
    if (wt) { if (fused) LAUNCH_TS_C(true, true); else LAUNCH_TS_C(true, false); }
    else { if (fused) LAUNCH_TS_C(false, true); else LAUNCH_TS_C(false, false); }
#undef LAUNCH_TS_C
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

extern "C" int scn_conv_tiles_finish(int cin, int64_t n_out, const float* bias, const float* residual,
                                     const float* relu_mask, float* Y, int cout, int flags, void* scratch,
                                     scn_stream_t stream) {
    SCN_REQUIRE(cin >= 1 && cout >= 1 && n_out >= 0);
    const int n_kc = (int)cdiv(cin, TS_KC);
    if (n_kc <= 1 || n_out == 0) return SCN_OK;
    SCN_REQUIRE(Y && scratch);
    float* slabs = (float*)((char*)scratch + ts_counter_bytes(cin, cout));
    hipStream_t st = S(stream);
    const bool v4 = cout % 4 == 0 && ((((uintptr_t)slabs | (uintptr_t)Y | (uintptr_t)bias | (uintptr_t)residual |
                                        (uintptr_t)relu_mask) & 15) == 0);
    const int rl = (flags & SCN_F_RESIDUAL_LAST) ? 1 : 0;
    if (v4)
        hipLaunchKernelGGL(k_conv_ts_sum<4>, dim3(scn::ew_grid(n_out * cout / 4, 256)), dim3(256), 0, st,
                           (const float*)slabs, n_kc, (long long)n_out, cout, bias, residual, relu_mask, Y, rl);
    else
        hipLaunchKernelGGL(k_conv_ts_sum<1>, dim3(scn::ew_grid(n_out * cout, 256)), dim3(256), 0, st,
                           (const float*)slabs, n_kc, (long long)n_out, cout, bias, residual, relu_mask, Y, rl);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}
