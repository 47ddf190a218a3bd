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
#include "scn_common.h"

using scn::S;
using scn::cdiv;

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

static constexpr int TS_KC = 32;        // channels per K-chunk
static constexpr int TS_CT = 32;        // output columns per workgroup
static constexpr int TS_T = 16;         // rows per tile

__device__ __forceinline__ int ts_ws_off(int o, int n, int k) {
    return (o * TS_CT + n) * TS_KC + ((((k >> 2) ^ (n & 7)) << 2) | (k & 3));
}

// Work decomposition (one launch per layer):
//   grid = n_chunks x wgs_per_chunk workgroups, sized so every CU holds its share (persistent-style, no tail waves).
//   A workgroup owns column chunk `chunk` and every wgs_per_chunk-th tile.  For each K-chunk (32 input
//   channels) it stages the weight slice once, then its waves pull tiles from an LDS counter (dynamic: tiles differ in
//   cost by their offset count).  A tile's partial sum over one K-chunk is carried through Y between K-chunks (read-add-
//   write by whichever wave pulls the tile; K-chunks are separated by a barrier), so no accumulator outlives a tile and
//   any wave can take any tile.  The first K-chunk adds the bias, the last one residual / ReLU-backward mask.
template <int NW, bool WT, bool VEC, bool VECN, bool FULLK>
__global__ __launch_bounds__(NW * 64) void k_conv_ts(
    const float* __restrict__ X, int cin, const int* __restrict__ tstab, const unsigned* __restrict__ tile_mask,
    const int* __restrict__ perm, int n_off, long long nt, const float* __restrict__ W, const float* __restrict__ bias,
    const float* __restrict__ residual, const float* __restrict__ relu_mask, float* __restrict__ Y, int cout, int flags,
    int n_chunks) {
    extern __shared__ __attribute__((aligned(16))) float Ws[];           // [n_off][32 n][32 k] swizzled, then counter
    constexpr int THREADS = NW * 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int i = lane & 15, kq = lane >> 4;
    const int chunk = blockIdx.x % n_chunks;
    const int wg_in_chunk = blockIdx.x / n_chunks, wgs_per_chunk = gridDim.x / n_chunks;
    const int n0 = chunk * TS_CT;
    const bool relu_in = flags & SCN_F_RELU_IN;
    const bool rev = flags & SCN_F_OFF_REVERSE;
    // debug ablation switches (tools/ablate_conv.py); never set by the product path
    const bool dbg_no_gather = flags & 256, dbg_no_mfma = flags & 512, dbg_no_stage = flags & 1024,
               dbg_no_store = flags & 2048, dbg_no_tiles = flags & 4096;
    int* counter = (int*)(Ws + n_off * TS_CT * TS_KC);
    // tiles are interleaved over the workgroups of a chunk (tile = wg + k * wgs): the mask sort orders tiles by their
    // offset sets, so contiguous ranges would give some workgroups all the 20-offset tiles and others the 5-offset ones
    const int n_tiles = (int)((nt - wg_in_chunk + wgs_per_chunk - 1) / wgs_per_chunk);

    const int nA = n0 + i, nB = n0 + 16 + i;
    const float bA = (bias && nA < cout) ? bias[nA] : 0.f;
    const float bB = (bias && nB < cout) ? bias[nB] : 0.f;

    for (int kc = 0; kc < cin; kc += TS_KC) {
        const bool first_kc = kc == 0, last_kc = kc + TS_KC >= cin;
        // ---- stage W[o'][kc..kc+32)[n0..n0+32) for every offset: 16-byte global reads, SB in flight per thread ----
        if (!dbg_no_stage) {
            constexpr int SB = (27 * 256 + THREADS - 1) / THREADS;           // one batch covers 27 offsets
            const int total4 = n_off * TS_CT * (TS_KC / 4);
            for (int base = 0; base < total4; base += THREADS * SB) {
                float4 v[SB];
#pragma unroll
                for (int u = 0; u < SB; ++u) {
                    const int e = base + u * THREADS + tid;
                    v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (e < total4) {
                        const int c4 = e & 7, m = (e >> 3) & 31, o = e >> 8;
                        const int wo = rev ? n_off - 1 - o : o;
                        if (WT) {          // [o][n = m][k]: k contiguous
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
                        if (WT) {
                            *(float4*)(Ws + ts_ws_off(o, m, 4 * c4)) = v[u];
                        } else {           // transpose into [n][k]
                            Ws[ts_ws_off(o, 4 * c4 + 0, m)] = v[u].x;
                            Ws[ts_ws_off(o, 4 * c4 + 1, m)] = v[u].y;
                            Ws[ts_ws_off(o, 4 * c4 + 2, m)] = v[u].z;
                            Ws[ts_ws_off(o, 4 * c4 + 3, m)] = v[u].w;
                        }
                    }
                }
            }
        }
        if (tid == 0) *counter = 0;
        __syncthreads();

        const int ka = kc + 4 * kq;
        const bool k0_ok = ka + 3 < cin, k1_ok = ka + 16 + 3 < cin;
        constexpr bool fullk = FULLK;     // cin % 32 == 0 and 16-byte aligned rows: every lane's 2 x 16 bytes are in range
        auto gather = [&](int idx, float4& a0, float4& a1) {
            if constexpr (FULLK) {
                // branch-free: rows without a rule read row 0 and are zeroed afterwards, so the loads stay outside
                // divergent control flow and the compiler can keep them in flight across the MFMAs (counted vmcnt)
                const float* xp = X + (long long)(idx < 0 ? 0 : idx) * cin + ka;
                a0 = *(const float4*)xp;
                a1 = *(const float4*)(xp + 16);
                return;
            } else {
            a0 = make_float4(0.f, 0.f, 0.f, 0.f);
            a1 = a0;
            if (idx >= 0 && !dbg_no_gather) {
                const float* xp = X + (long long)idx * cin + ka;
                if (VEC) {
                    if (k0_ok) a0 = *(const float4*)xp;
                    if (k1_ok) a1 = *(const float4*)(xp + 16);
                } else {
                    if (ka < cin) a0.x = xp[0];
                    if (ka + 1 < cin) a0.y = xp[1];
                    if (ka + 2 < cin) a0.z = xp[2];
                    if (ka + 3 < cin) a0.w = xp[3];
                    if (ka + 16 < cin) a1.x = xp[16];
                    if (ka + 17 < cin) a1.y = xp[17];
                    if (ka + 18 < cin) a1.z = xp[18];
                    if (ka + 19 < cin) a1.w = xp[19];
                }
            }
            }
        };

        // ---- tiles of this workgroup, pulled dynamically -----------------------------------------------------------
        for (;;) {
            int tl = 0;
            if (lane == 0) tl = atomicAdd(counter, 1);
            tl = __builtin_amdgcn_readfirstlane(tl);
            if (tl >= n_tiles || dbg_no_tiles) break;
            const long long tile = wg_in_chunk + (long long)tl * wgs_per_chunk;
            unsigned m = tile_mask[tile];
            const int* tb = tstab + tile * n_off * TS_T + i;
            const int rowbase = (int)(tile * TS_T) + 4 * kq;
            int orow[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) orow[j] = perm[rowbase + j];

            // 3-stage pipeline over the set bits of m:  indices 2 offsets ahead, A rows 1 offset ahead, MFMA now.
            // All loads are unconditional (a finished list re-reads its last offset) -- see gather().  The A registers
            // ping-pong between two named sets (loop unrolled by 2): a register move of a prefetched value would force
            // its load to complete and serialise the pipeline.
            int oc = -1, on = -1, idxc = -1, idxn = -1;
            int olast = 0;
            float4 p0, p1, q0, q1;
            if (m) { oc = __builtin_ctz(m); m &= m - 1; olast = oc; }
            idxc = tb[olast * TS_T];
            if (m) { on = __builtin_ctz(m); m &= m - 1; olast = on; }
            idxn = tb[olast * TS_T];
            gather(idxc, p0, p1);

            // partial sums of the previous K-chunks (or the bias)
            f32x4 c0, c1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                c0[j] = bA;
                c1[j] = bB;
                if (!first_kc && orow[j] >= 0) {
                    const long long off = (long long)orow[j] * cout;
                    if (nA < cout) c0[j] = Y[off + nA];
                    if (nB < cout) c1[j] = Y[off + nB];
                }
            }

            auto step = [&](float4& cur0, float4& cur1, float4& nxt0, float4& nxt1) {
                int onn = -1;
                if (m) { onn = __builtin_ctz(m); m &= m - 1; olast = onn; }
                const int idxnn = tb[olast * TS_T];                  // stage I: two offsets ahead
                gather(idxn, nxt0, nxt1);                            // stage G: next offset's A rows

                float4 a0 = cur0, a1 = cur1;                         // stage C: this offset
                if (FULLK && idxc < 0) { a0 = make_float4(0.f, 0.f, 0.f, 0.f); a1 = a0; }
                if (relu_in) {
                    a0.x = fmaxf(a0.x, 0.f); a0.y = fmaxf(a0.y, 0.f); a0.z = fmaxf(a0.z, 0.f); a0.w = fmaxf(a0.w, 0.f);
                    a1.x = fmaxf(a1.x, 0.f); a1.y = fmaxf(a1.y, 0.f); a1.z = fmaxf(a1.z, 0.f); a1.w = fmaxf(a1.w, 0.f);
                }
                const float4 b00 = *(const float4*)(Ws + ts_ws_off(oc, i, 4 * kq));
                const float4 b01 = *(const float4*)(Ws + ts_ws_off(oc, i, 16 + 4 * kq));
                const float4 b10 = *(const float4*)(Ws + ts_ws_off(oc, 16 + i, 4 * kq));
                const float4 b11 = *(const float4*)(Ws + ts_ws_off(oc, 16 + i, 16 + 4 * kq));
                if (dbg_no_mfma) {
                    c0[0] += a0.x + b00.x + a1.y + b01.y; c1[0] += a0.z + b10.z + a1.w + b11.w;
                } else {
                    c0 = MFMA16(a0.x, b00.x, c0);  c1 = MFMA16(a0.x, b10.x, c1);
                    c0 = MFMA16(a0.y, b00.y, c0);  c1 = MFMA16(a0.y, b10.y, c1);
                    c0 = MFMA16(a0.z, b00.z, c0);  c1 = MFMA16(a0.z, b10.z, c1);
                    c0 = MFMA16(a0.w, b00.w, c0);  c1 = MFMA16(a0.w, b10.w, c1);
                    c0 = MFMA16(a1.x, b01.x, c0);  c1 = MFMA16(a1.x, b11.x, c1);
                    c0 = MFMA16(a1.y, b01.y, c0);  c1 = MFMA16(a1.y, b11.y, c1);
                    c0 = MFMA16(a1.z, b01.z, c0);  c1 = MFMA16(a1.z, b11.z, c1);
                    c0 = MFMA16(a1.w, b01.w, c0);  c1 = MFMA16(a1.w, b11.w, c1);
                }
                oc = on; idxc = idxn;
                on = onn; idxn = idxnn;
            };
            while (oc >= 0) {
                step(p0, p1, q0, q1);
                if (oc < 0) break;
                step(q0, q1, p0, p1);
            }
            // ---- tile epilogue: 16 lanes write 64 contiguous bytes of a row ---------------------------------------
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = orow[j];
                if (row < 0 || dbg_no_store) continue;
                const long long off = (long long)row * cout;
                if (nA < cout) {
                    float y = c0[j];
                    if (last_kc) {
                        if (residual) y += residual[off + nA];
                        if (relu_mask && !(relu_mask[off + nA] > 0.f)) y = 0.f;
                    }
                    Y[off + nA] = y;
                }
                if (nB < cout) {
                    float y = c1[j];
                    if (last_kc) {
                        if (residual) y += residual[off + nB];
                        if (relu_mask && !(relu_mask[off + nB] > 0.f)) y = 0.f;
                    }
                    Y[off + nB] = y;
                }
            }
        }
        __syncthreads();      // all tiles of this K-chunk done (Y partials visible to this workgroup, Ws free)
    }
}

extern "C" int scn_conv_tiles(const float* X, int cin, const int32_t* tstab, const uint32_t* tile_mask,
                              const int32_t* perm, int n_off, int64_t n_out, const float* W, const float* bias,
                              const float* residual, const float* relu_mask, float* Y, int cout, int flags,
                              scn_stream_t stream) {
    SCN_REQUIRE(n_off >= 1 && n_off <= 27 && n_out >= 0 && cin >= 1 && cout >= 1);
    if (n_out == 0) return SCN_OK;
    SCN_REQUIRE(X && tstab && tile_mask && perm && W && Y);
    constexpr int NW = 16;
    const int64_t nt = cdiv(n_out, TS_T);
    const int n_chunks = (int)cdiv(cout, TS_CT);
    const size_t lds = (size_t)n_off * TS_CT * TS_KC * sizeof(float) + 16;
    int wg_per_cu = (int)((160 * 1024) / lds);
    if (wg_per_cu > 4) wg_per_cu = 4;
    if (wg_per_cu < 1) wg_per_cu = 1;
    // persistent-style grid: every CU gets its workgroups, split evenly over the column chunks; a workgroup should
    // have at least ~NW tiles
    int64_t wgs_per_chunk = (256 * wg_per_cu) / n_chunks;
    if (wgs_per_chunk < 1) wgs_per_chunk = 1;
    if (wgs_per_chunk > cdiv(nt, NW)) wgs_per_chunk = cdiv(nt, NW);
    const bool wt = flags & SCN_F_W_TRANSPOSED;
    const bool vec = (cin % 4 == 0) && (((uintptr_t)X & 15) == 0) && (((uintptr_t)W & 15) == 0);
    const bool vecn = (cout % 4 == 0) && (((uintptr_t)W & 15) == 0);
    dim3 grid((unsigned)(wgs_per_chunk * n_chunks));
    hipStream_t st = S(stream);
    const bool fullk = vec && (cin % TS_KC == 0) && !(flags & 256);
#define LAUNCH_TS(T, V, VN, FK)                                                                                     \
    do {                                                                                                            \
        SCN_HIP(hipFuncSetAttribute((const void*)k_conv_ts<NW, T, V, VN, FK>,                                       \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                       \
        hipLaunchKernelGGL((k_conv_ts<NW, T, V, VN, FK>), grid, dim3(NW * 64), lds, st, X, cin, tstab, tile_mask,   \
                           perm, n_off, (long long)nt, W, bias, residual, relu_mask, Y, cout, flags, n_chunks);     \
    } while (0)
    if (fullk && wt) LAUNCH_TS(true, true, true, true);
    else if (fullk && vecn) LAUNCH_TS(false, true, true, true);
    else if (fullk) LAUNCH_TS(false, true, false, true);
    else if (wt && vec) LAUNCH_TS(true, true, true, false);
    else if (wt) LAUNCH_TS(true, false, true, false);
    else if (vec && vecn) LAUNCH_TS(false, true, true, false);
    else if (vec) LAUNCH_TS(false, true, false, false);
    else if (vecn) LAUNCH_TS(false, false, true, false);
    else LAUNCH_TS(false, false, false, false);
#undef LAUNCH_TS
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}
