// Tile-sorted rule tables for the output-stationary convolution kernel (scn_conv_ts.hip).
//
// A [n_off][N] rule table walked in tiles of 16 consecutive rows wastes matrix work: on surface-like clouds the rows
// of a tile seldom share their offsets (measured 2.8-3.2x executed/useful at 150k voxels, profiles/r1a).  Rows are
// therefore regrouped by their OFFSET MASK (bit o set <=> table[o][r] >= 0): a stable radix sort of (mask, row) puts
// rows with equal masks next to each other and keeps the original (spatially coherent) order inside a group, which
// brings executed/useful down to 1.2-1.4 for 3^3 submanifold tables and ~1.02 for 2^3/stride-2 child tables.
// The permutation only decides which rows share an MFMA tile; every row still accumulates its own offsets in
// ascending order, so results do not depend on it.
//
// Outputs (all int32, caller-allocated):
//   perm      [NT*16]          sorted position -> original row (-1 padding in the last tile)
//   tstab     [NT][n_off][16]  tile-major table: tstab[t][o][i] = table[o][perm[16t+i]]  (one coalesced 64-B read
//                              per (tile, offset) in the kernel)
//   tile_mask [NT]             OR of the row masks of the tile: the offsets the kernel has to visit
//   tile_order[NT]             tile ids by offset count descending (the kernel's hand-out order)
//
// Both sorts -- rows by mask key, tiles by offset count -- are the hand-written stable LSD radix sort of scn_sort.hip
// (round 1-2a used rocPRIM's radix_sort_pairs, which below ~1 M items is a block sort + 4-6 merge launches, and a
// one-workgroup counting sort for the tile order).
#include <cstring>

#include "scn_common.h"
#include "scn_sort.h"

using scn::S;
using scn::cdiv;

// Sort key = Gray-code rank of the offset mask with its bits PERMUTED: bit position of offset o is key_bit[o].  For 3^3 tables the rare
// offsets (the 8 corners, then the 12 edges) take the most significant positions, faces and the centre the least: rows
// that differ only in common offsets end up adjacent.  Measured executed/useful 1.31 vs 1.38 (natural bit order) at
// 150k voxels.  Other table shapes keep the natural order.
struct KeyBits { unsigned char pos[32]; };

static KeyBits make_key_bits(int n_off) {
    KeyBits kb;
    for (int o = 0; o < 32; ++o) kb.pos[o] = (unsigned char)o;
    if (n_off == 27) {
        int next = 0;
        for (int cls = 0; cls <= 3; ++cls)                 // centre (0), faces (1), edges (2), corners (3): LSB -> MSB
            for (int o = 0; o < 27; ++o) {
                const int dx = o / 9 - 1, dy = (o / 3) % 3 - 1, dz = o % 3 - 1;
                if ((dx != 0) + (dy != 0) + (dz != 0) == cls) kb.pos[o] = (unsigned char)next++;
            }
    }
    return kb;
}

// bin_shift (round 6, bf16 tile kernels only): the key's top bits are the row's BIN, row * 2^lb / n -- rows are numbered in the
// order the points arrive (mesh order), so a row range is a region of the scene: the rows of a tile then come from ONE region
// and the 3^3 neighbours a launch gathers for consecutive tiles meet in L2.  27 mask bits + lb <= 5 bin bits.
__global__ void k_row_masks(const int* __restrict__ table, int n_off, long long n, KeyBits kb, unsigned* __restrict__ key, int lb) {
    for (long long r = blockIdx.x * (long long)blockDim.x + threadIdx.x; r < n;
         r += (long long)gridDim.x * blockDim.x) {
        unsigned m = 0;
        for (int o = 0; o < n_off; ++o) m |= (table[(long long)o * n + r] >= 0 ? 1u : 0u) << kb.pos[o];
        // sort key = RANK of the permuted mask in the reflected Gray code (binary value whose Gray code is m): masks that
        // follow each other in this order differ in few offsets, so the tiles that mix several masks visit fewer offsets
        // than in binary order -- executed/useful 1.276 / 1.157 / 1.173 / 1.253 on the four levels of the cfg-2 scene
        // against 1.311 / 1.175 / 1.186 / 1.267
        m ^= m >> 1; m ^= m >> 2; m ^= m >> 4; m ^= m >> 8; m ^= m >> 16;
        if (lb) m |= (unsigned)((r << lb) / n) << 27;
        key[r] = m;
    }
}

__global__ void k_build_tiles(const int* __restrict__ table, int n_off, long long n, const int* __restrict__ sorted_rows,
                              const unsigned* __restrict__ sorted_key, KeyBits kb, long long nt, int* __restrict__ perm,
                              int* __restrict__ tstab, unsigned* __restrict__ tile_mask, unsigned* __restrict__ tile_cost,
                              unsigned* __restrict__ tile_xkey, int lb) {
    // one thread per (tile, lane i); 16 threads of a tile are adjacent
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < nt * 16;
         e += (long long)gridDim.x * blockDim.x) {
        const long long t = e >> 4;
        const int i = (int)(e & 15);
        const bool ok = e < n;
        const int row = ok ? sorted_rows[e] : -1;
        perm[e] = row;
        const unsigned rank = ok ? (lb ? sorted_key[e] & 0x7FFFFFFu : sorted_key[e]) : 0u;      // (without the bin bits)
        const unsigned key = rank ^ (rank >> 1);            // Gray code of the rank = the permuted mask
        unsigned m = 0;                                     // the real offset mask: undo the key's bit permutation
        for (int o = 0; o < n_off; ++o) {
            const unsigned have = (key >> kb.pos[o]) & 1u;
            m |= have << o;
            tstab[(t * n_off + o) * 16 + i] = have ? table[(long long)o * n + row] : -1;
        }
        // OR over the 16 lanes of the tile (they sit in one quarter of a wave)
        m |= __shfl_xor(m, 1);
        m |= __shfl_xor(m, 2);
        m |= __shfl_xor(m, 4);
        m |= __shfl_xor(m, 8);
        if (i == 0) {
            tile_mask[t] = m;
            tile_cost[t] = 32u - (unsigned)__popc(m);       // sort key of the hand-out order: most offsets first
            // key of the XCD-local hand-out order (scn_tiles_build_x): (spatial bin of the tile's first row, cost) -- rows
            // are numbered in the order the points arrive (mesh order), so a row range is a region of the scene
            if (tile_xkey) tile_xkey[t] = ((unsigned)(((long long)(row < 0 ? 0 : row) * 8) / n) << 6) | (32u - (unsigned)__popc(m));
        }
    }
}

// tile_order: tile ids by offset count (popcount of tile_mask) DESCENDING, equal counts in tile order -- the order in
// which the convolution kernel hands tiles to its waves (longest-processing-time first): a stable sort of the tile ids on
// the 6-bit key 32 - popcount.

static inline int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }

// bin_start[b] = first position of the (bin, cost)-sorted tile list whose bin is >= b (b = 0..8; bin_start[8] = nt)
__global__ void k_xbin_start(const unsigned* __restrict__ xkey_sorted, long long nt, int* __restrict__ bin_start) {
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < nt; t += (long long)gridDim.x * blockDim.x) {
        const int kb = (int)(xkey_sorted[t] >> 6);
        const int kp = t ? (int)(xkey_sorted[t - 1] >> 6) : -1;
        for (int b = kp + 1; b <= kb; ++b) bin_start[b] = (int)t;
        if (t == nt - 1)
            for (int b = kb + 1; b <= 8; ++b) bin_start[b] = (int)nt;
    }
}

extern "C" int64_t scn_tiles_scratch_bytes(int n_off, int64_t n) {
    if (n_off < 1 || n_off > 32 || n < 0) return -1;
    const int64_t nt = cdiv(n, 16);
    return align256(4 * n) * 3 + align256(scn::sort_pairs_scratch_bytes(n)) + align256(4 * nt) * 4 +
           align256(scn::sort_pairs_scratch_bytes(nt)) + 256;
}

extern "C" int scn_tiles_build_x(const int32_t* table, int n_off, int64_t n, int32_t* perm, int32_t* tstab,
                                 uint32_t* tile_mask, int32_t* tile_order, int with_x, void* scratch, scn_stream_t stream) {
    SCN_REQUIRE(n_off >= 1 && n_off <= 32 && n >= 0);
    if (n == 0) return SCN_OK;
    SCN_REQUIRE(table && perm && tstab && tile_mask && tile_order && scratch);
    SCN_REQUIRE(n < 2147483647LL / 32);
    // with_x: bit 0 = also the XCD-local hand-out order; bits 8-10 = log2 of the row bins in the sort key (27-offset tables)
    const int lb = n_off == 27 ? (with_x >> 8) & 7 : 0;
    SCN_REQUIRE(lb <= 5);
    with_x &= 1;
    const int64_t nt = cdiv(n, 16);
    char* p = (char*)scratch;
    unsigned* mask = (unsigned*)p;          p += align256(4 * n);
    unsigned* mask_sorted = (unsigned*)p;   p += align256(4 * n);
    int* rows_sorted = (int*)p;             p += align256(4 * n);
    void* sort_scr = p;                     p += align256(scn::sort_pairs_scratch_bytes(n));
    unsigned* cost = (unsigned*)p;          p += align256(4 * nt);
    unsigned* cost_sorted = (unsigned*)p;   p += align256(4 * nt);
    unsigned* xkey = (unsigned*)p;          p += align256(4 * nt);
    unsigned* xkey_sorted = (unsigned*)p;   p += align256(4 * nt);
    void* sort_scr2 = p;
    hipStream_t st = S(stream);
    const KeyBits kb = make_key_bits(n_off);
    hipLaunchKernelGGL(k_row_masks, dim3(scn::ew_grid(n, 256)), dim3(256), 0, st, table, n_off, (long long)n, kb, mask, lb);
    SCN_LAUNCH_CHECK();
    int rc = scn::sort_pairs(mask, nullptr, n, n_off + lb, mask_sorted, rows_sorted, sort_scr, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_build_tiles, dim3(scn::ew_grid(nt * 16, 256)), dim3(256), 0, st, table, n_off, (long long)n,
                       (const int*)rows_sorted, (const unsigned*)mask_sorted, kb, (long long)nt, perm, tstab, tile_mask,
                       cost, with_x ? xkey : (unsigned*)nullptr, lb);
    SCN_LAUNCH_CHECK();
    rc = scn::sort_pairs(cost, nullptr, nt, 6, cost_sorted, tile_order, sort_scr2, st);
    if (rc) return rc;
    if (with_x) {
        // the XCD-local hand-out order behind the first one: tile ids by (bin, cost) and the nine bin starts
        int32_t* order_x = tile_order + nt;
        rc = scn::sort_pairs(xkey, nullptr, nt, 9, xkey_sorted, order_x, sort_scr2, st);
        if (rc) return rc;
        hipLaunchKernelGGL(k_xbin_start, dim3(scn::ew_grid(nt, 256)), dim3(256), 0, st, (const unsigned*)xkey_sorted,
                           (long long)nt, order_x + nt);
        SCN_LAUNCH_CHECK();
    }
    return SCN_OK;
}

extern "C" int64_t scn_tiles_order_ints(int64_t n, int with_x) {
    if (n < 0) return -1;
    const int64_t nt = cdiv(n, 16);
    return (with_x & 1) ? 2 * nt + 16 : nt;          // (bit 0: the second order; bits 8-10 are the row-bin count of the sort key)
}

extern "C" int scn_tiles_build(const int32_t* table, int n_off, int64_t n, int32_t* perm, int32_t* tstab,
                               uint32_t* tile_mask, int32_t* tile_order, void* scratch, scn_stream_t stream) {
    return scn_tiles_build_x(table, n_off, n, perm, tstab, tile_mask, tile_order, 0, scratch, stream);
}
