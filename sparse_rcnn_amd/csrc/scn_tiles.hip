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
// The radix sort is rocPRIM's device_radix_sort (a plain library primitive on the index-building path); everything
// else is hand-written.
#include <cstring>

#include "scn_common.h"

#include <rocprim/rocprim.hpp>

using scn::S;
using scn::cdiv;

// Sort key = the offset mask with its bits PERMUTED: bit position of offset o is key_bit[o].  For 3^3 tables the rare
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

__global__ void k_row_masks(const int* __restrict__ table, int n_off, long long n, KeyBits kb, unsigned* __restrict__ key,
                            int* __restrict__ iota) {
    for (long long r = blockIdx.x * (long long)blockDim.x + threadIdx.x; r < n;
         r += (long long)gridDim.x * blockDim.x) {
        unsigned m = 0;
        for (int o = 0; o < n_off; ++o) m |= (table[(long long)o * n + r] >= 0 ? 1u : 0u) << kb.pos[o];
        key[r] = m;
        iota[r] = (int)r;
    }
}

__global__ void k_build_tiles(const int* __restrict__ table, int n_off, long long n, const int* __restrict__ sorted_rows,
                              const unsigned* __restrict__ sorted_key, KeyBits kb, long long nt, int* __restrict__ perm,
                              int* __restrict__ tstab, unsigned* __restrict__ tile_mask) {
    // one thread per (tile, lane i); 16 threads of a tile are adjacent
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < nt * 16;
         e += (long long)gridDim.x * blockDim.x) {
        const long long t = e >> 4;
        const int i = (int)(e & 15);
        const bool ok = e < n;
        const int row = ok ? sorted_rows[e] : -1;
        perm[e] = row;
        const unsigned key = ok ? sorted_key[e] : 0u;
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
        if (i == 0) tile_mask[t] = m;
    }
}

// tile_order: tile ids sorted by offset count (popcount of tile_mask) DESCENDING -- the order in which the convolution
// kernel hands tiles to its waves (longest-processing-time first).  Counting sort on 33 bins, one block.
__global__ __launch_bounds__(1024) void k_tile_order(const unsigned* __restrict__ tile_mask, int nt,
                                                     int* __restrict__ tile_order) {
    __shared__ int hist[33], cursor[33];
    if (threadIdx.x < 33) hist[threadIdx.x] = 0;
    __syncthreads();
    for (int t = threadIdx.x; t < nt; t += 1024) atomicAdd(&hist[32 - __popc(tile_mask[t])], 1);   // bin 0 = 32 offsets
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int b = 0; b < 33; ++b) { cursor[b] = run; run += hist[b]; }
    }
    __syncthreads();
    // position inside a bin = number of earlier tiles of the same bin: walk in tile order so the result is deterministic
    for (int base = 0; base < nt; base += 1024) {
        const int t = base + threadIdx.x;
        const int bin = t < nt ? 32 - __popc(tile_mask[t]) : -1;
        // rank among the threads of this pass with the same bin (wave ballot + per-wave counts through LDS)
        __shared__ int wcnt[16][33];
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        for (int b = lane; b < 33; b += 64) wcnt[w][b] = 0;
        __syncthreads();
        int my_rank = 0;
        for (int b = 0; b < 33; ++b) {
            const unsigned long long mb = __ballot(bin == b);
            if (bin == b) my_rank = __popcll(mb & ((1ull << lane) - 1ull));
            if (lane == 0) wcnt[w][b] = __popcll(mb);
        }
        __syncthreads();
        if (bin >= 0) {
            int before = 0;
            for (int k = 0; k < w; ++k) before += wcnt[k][bin];
            tile_order[cursor[bin] + before + my_rank] = t;
        }
        __syncthreads();
        if (threadIdx.x < 33) {
            int tot = 0;
            for (int k = 0; k < 16; ++k) tot += wcnt[k][threadIdx.x];
            cursor[threadIdx.x] += tot;
        }
        __syncthreads();
    }
}

static inline int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }

static size_t sort_temp_bytes(int64_t n, int bits) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (unsigned*)nullptr, (unsigned*)nullptr, (int*)nullptr, (int*)nullptr,
                                    (size_t)n, 0, bits, (hipStream_t)0);
    return bytes;
}

extern "C" int64_t scn_tiles_scratch_bytes(int n_off, int64_t n) {
    if (n_off < 1 || n_off > 32 || n < 0) return -1;
    return align256(4 * n) * 4 + align256((int64_t)sort_temp_bytes(n, n_off)) + 256;
}

extern "C" int scn_tiles_build(const int32_t* table, int n_off, int64_t n, int32_t* perm, int32_t* tstab,
                               uint32_t* tile_mask, int32_t* tile_order, void* scratch, scn_stream_t stream) {
    SCN_REQUIRE(n_off >= 1 && n_off <= 32 && n >= 0);
    if (n == 0) return SCN_OK;
    SCN_REQUIRE(table && perm && tstab && tile_mask && tile_order && scratch);
    SCN_REQUIRE(n < 2147483647LL / 32);
    char* p = (char*)scratch;
    unsigned* mask = (unsigned*)p;          p += align256(4 * n);
    unsigned* mask_sorted = (unsigned*)p;   p += align256(4 * n);
    int* iota = (int*)p;                    p += align256(4 * n);
    int* rows_sorted = (int*)p;             p += align256(4 * n);
    void* temp = p;
    size_t temp_bytes = sort_temp_bytes(n, n_off);
    hipStream_t st = S(stream);
    const KeyBits kb = make_key_bits(n_off);
    hipLaunchKernelGGL(k_row_masks, dim3(scn::ew_grid(n, 256)), dim3(256), 0, st, table, n_off, (long long)n, kb, mask,
                       iota);
    SCN_LAUNCH_CHECK();
    SCN_HIP(rocprim::radix_sort_pairs(temp, temp_bytes, mask, mask_sorted, iota, rows_sorted, (size_t)n, 0, n_off, st));
    const int64_t nt = cdiv(n, 16);
    hipLaunchKernelGGL(k_build_tiles, dim3(scn::ew_grid(nt * 16, 256)), dim3(256), 0, st, table, n_off, (long long)n,
                       (const int*)rows_sorted, (const unsigned*)mask_sorted, kb, (long long)nt, perm, tstab, tile_mask);
    SCN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_tile_order, dim3(1), dim3(1024), 0, st, (const unsigned*)tile_mask, (int)nt, tile_order);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}
