// scn_pyramid_build_ex(..., SCN_PYRAMID_FUSED): the index build of one forward pass WITHOUT host round trips.
//
// What the reference's scn.Metadata builds on the host while InputLayer / SubmanifoldConvolution / Convolution run
// (custom_operations.py:62-86: a new Metadata inside every forward; roi_select_sparse.py:74-84: again per ROI batch) was,
// in rounds 1-3, ~136 launches of 3-8 us on two queues and four host waits (scn_pyramid.hip: every kernel took its row count
// as a launch parameter, so the host had to learn the size of level l + 1 before it could queue level l + 1) -- 0.94 ms for
// 34 MB of algorithmic bytes, 0.45 % of the HBM peak, the launches being their own floor.  Here:
//
//  * every level size stays in DEVICE memory (`dsz`): a kernel reads its row count from the word the numbering kernel
//    wrote, grids are sized by the upper bound N_l <= n_points and surplus workgroups return at once, buffers are placed at
//    upper-bound offsets (the table of a level is still the contiguous [k^3][N_l] the consumers expect: the stride is read
//    from `dsz` too);
//  * the levels run side by side INSIDE a launch instead of one after the other: a launch is a list of jobs (SubM table of
//    level 0 .. L-1, child tables, parents, ...) and a workgroup finds its job from its block index -- the chain of launches
//    is as long as the dependency chain, not as long as chain x levels;
//  * flag / scan / fill of the numbering is one pass (decoupled look-back over the workgroups' counts, tickets for the
//    forward-progress order), coarse sites are numbered WITHOUT a second hash-insert pass: a fine row is the first of its
//    coarse site iff it is the lowest-numbered of the <= 8 children, which it learns from 7 probes of its own level's table;
//    the child table is 8 probes per coarse row (no memset + scatter), the row masks / sort keys / first-pass digit counts /
//    per-offset rule counts come out of the table kernel that has the 27 probe results in registers anyway;
//  * ONE device -> host copy at the end brings every size the host needs (level rows, rule prefixes, range flag).
//
//   launches for L levels, k = 3:  fill | insert0 | number0 | coarsen x (L-1) | tables | 3 mask-sort passes (counts of pass 0
//   come from `tables`: scan, scatter | hist, scan, scatter | hist, scan, scatter; rule scans / fills ride in the first two) |
//   tiles | tile orders  =  14 + (L - 1)   (17 for the benchmark U-Net; rounds 1-3: 136), one host wait.
//
// Round 5: BRICKS.  Every neighbour / sibling / child / parent probe of the build went to an open-addressing table keyed by the
// voxel: 4 M scattered probes for the 27-neighbour tables of the 150 k scene alone, each a miss in the XCD's L2 (100 us of the
// 326 us of kernels).  Now every level also keeps its sites in 4 x 4 x 4 BRICKS: a small directory keyed by (b, x>>2, y>>2, z>>2)
// -- a surface scene has one brick per ~10 rows, so the directory and the bricks (64 row numbers + a 64-bit occupancy word
// each) of a level stay in L2 -- and rows arrive in mesh order, so the probes of a wave meet the same few bricks.  The 27
// neighbours of a voxel lie in at most 8 bricks, its 8 siblings / the 8 children of a coarse site / its parent in ONE.  The
// bricks are filled where the rows are numbered (k_number0, k_coarsen): no launch is added.  A directory that overflows
// (points so sparse that nearly every one has a brick of its own) sets a device word and every consumer takes the voxel
// tables as before -- same structures either way (tests: fused == round-3 builder, SCN_PYRAMID_NO_BRICKS A/B).
//
// Same structures as the step-by-step entry points, bit for bit (tables, compacted rules, perm / tstab / tile_mask / both
// tile orders, parents, child tables, row numbering): tests/test_gpu_parity.py::test_fused_pyramid_build_*.  Only the hash
// tables differ (every level gets the capacity of n_points: its size must not depend on a count the host does not know).
#include <stdlib.h>

#include "scn_common.h"

using scn::S;
using scn::cdiv;

namespace {

constexpr int T = 256;                 // threads per workgroup, every kernel of this file
constexpr int IT = 4;                  // items per thread in the scan / sort workgroups
constexpr int TILE = T * IT;           // items per workgroup: idx = base + it * T + tid  (ballot order == index order)
constexpr int MAXL = SCN_PYRAMID_MAX_LEVELS;
constexpr int MAXB = 1024;             // radix bins (9-bit passes; 10-bit passes where the key carries row-bin bits)
constexpr unsigned F_AGG = 1u << 30, F_INCL = 2u << 30, VMASK = (1u << 30) - 1u;
constexpr int SPIN_MAX = 1 << 22;      // look-back polls before a workgroup gives up (sets DS_ERR; never hangs the GPU)

// the device-resident sizes (int64 words), copied to the host ONCE when everything has been queued
constexpr int DS_BAD = 0, DS_ERR = 1, DS_N = 2 /* + l */, DS_SP = 16 /* + 28 l + o */, DS_CP = 16 + 28 * MAXL /* + 9 l + o */;
constexpr int DS_NOBRICK = DS_CP + 9 * MAXL;      // != 0: a brick directory overflowed (or bricks are switched off): voxel tables
constexpr int DS_LEN = DS_NOBRICK + 1;
constexpr int BRICK_PROBES = 32;       // slots an insert may walk before it gives the directory up

struct KeyBits { unsigned char pos[32]; };

// per-level buffers as byte offsets >> 8 into the workspace (every buffer is 256-byte aligned)
struct LvA {
    uint32_t coords, status;
    uint32_t table, bsums, prefix, key, key_s, rows_s, ktmp, vtmp, counts, totals, perm, tstab, tmask, torder, cost, xkey,
        in_rows, out_rows;
    uint32_t parent, fine_off, child, cbsums, cprefix, ckey, ckey_s, crows_s, ccounts, ctotals, cperm, ctstab, ctmask,
        ctorder, ccost, cin_rows, cout_rows;
};

struct PA {
    char* base;
    long long* dsz;
    int* tickets;                       // [MAXL]  (numbering of level l)
    unsigned long long* keys;           // [n_levels][cap]
    int* hrows;                         // [n_levels][cap]
    unsigned long long* bkeys;          // [n_levels][bcap]      brick directory: keys (b, x>>2, y>>2, z>>2), EMPTY-filled
    unsigned long long* bmask;          // [n_levels][bcap]      occupancy of the 64 sites of a brick, zero-filled
    int* bent;                          // [n_levels][bcap][64]  row numbers, valid where the occupancy bit is set
    long long cap, bound, bcap;
    int no_bricks;                      // developer switch SCN_PYRAMID_NO_BRICKS: voxel tables everywhere (A/B, cross-check)
    int n_levels, k, with_x, nblk;      // nblk = cdiv(bound, TILE)
    int lbins;                          // log2 of the row bins in level 0's SubM sort key (0: mask sort only)
    LvA lv[MAXL];
    KeyBits kb;
};

template <class X>
__device__ __forceinline__ X* at(const PA& a, uint32_t off) { return (X*)(a.base + ((size_t)off << 8)); }

__device__ __forceinline__ unsigned ld_status(const unsigned* p) {
    return __hip_atomic_load(const_cast<unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_status(unsigned* p, unsigned v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ int lookup(const unsigned long long* __restrict__ keys, const int* __restrict__ rows,
                                      unsigned long long mask, unsigned long long key) {
    unsigned long long slot = scn_hash_slot(key, mask);
    for (unsigned long long probe = 0; probe <= mask; ++probe) {
        const unsigned long long k = keys[slot];
        if (k == key) return rows[slot];
        if (k == SCN_EMPTY_KEY) return -1;
        slot = (slot + 1) & mask;
    }
    return -1;
}

__device__ __forceinline__ long long cdiv_dev(long long n) { return (n + TILE - 1) / TILE; }

// ---- bricks ---------------------------------------------------------------------------------------------------------
struct Bricks { unsigned long long* keys; unsigned long long* mask; int* ent; unsigned long long m; };
__device__ __forceinline__ Bricks bricks_of(const PA& a, int l) {
    return Bricks{a.bkeys + (size_t)l * a.bcap, a.bmask + (size_t)l * a.bcap, a.bent + (size_t)l * a.bcap * 64,
                  (unsigned long long)a.bcap - 1ull};
}
__device__ __forceinline__ int brick_local(int x, int y, int z) { return ((x & 3) * 4 + (y & 3)) * 4 + (z & 3); }

// row `row` at site (x, y, z, b): claim / find the brick, set the occupancy bit, store the row.  false: directory full.
__device__ __forceinline__ bool brick_insert(const Bricks& B, int x, int y, int z, int b, int row) {
    const unsigned long long key = scn_pack_key(x >> 2, y >> 2, z >> 2, b);
    unsigned long long slot = scn_hash_slot(key, B.m);
    for (int p = 0; p < BRICK_PROBES; ++p) {
        const unsigned long long prev = atomicCAS(&B.keys[slot], SCN_EMPTY_KEY, key);
        if (prev == SCN_EMPTY_KEY || prev == key) {
            B.ent[slot * 64 + brick_local(x, y, z)] = row;
            atomicOr(&B.mask[slot], 1ull << brick_local(x, y, z));
            return true;
        }
        slot = (slot + 1) & B.m;
    }
    return false;
}

// G brick keys -> directory slots (or -1) and occupancy words; the G first-slot loads are in flight together (as lookup_n)
template <int G>
__device__ __forceinline__ void brick_find_n(const Bricks& B, const unsigned long long (&key)[G], const bool (&want)[G],
                                             int (&slot_out)[G], unsigned long long (&occ)[G]) {
    unsigned long long slot[G], k[G];
#pragma unroll
    for (int j = 0; j < G; ++j) slot[j] = scn_hash_slot(key[j], B.m);
#pragma unroll
    for (int j = 0; j < G; ++j) k[j] = B.keys[slot[j]];
#pragma unroll
    for (int j = 0; j < G; ++j) {
        slot_out[j] = (want[j] && k[j] == key[j]) ? (int)slot[j] : -1;
        if (want[j] && k[j] != key[j] && k[j] != SCN_EMPTY_KEY) {          // a foreign key in the first slot: walk on (rare)
            unsigned long long s2 = (slot[j] + 1) & B.m;
            for (int p = 1; p < BRICK_PROBES; ++p) {
                const unsigned long long kk = B.keys[s2];
                if (kk == key[j]) { slot_out[j] = (int)s2; break; }
                if (kk == SCN_EMPTY_KEY) break;
                s2 = (s2 + 1) & B.m;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < G; ++j) occ[j] = B.mask[slot_out[j] >= 0 ? slot_out[j] : 0];
#pragma unroll
    for (int j = 0; j < G; ++j) occ[j] = slot_out[j] >= 0 ? occ[j] : 0ull;
}

// the 2 x 2 x 2 cell with even corner (x0, y0, z0) lies in one brick: its 8 rows (or -1), o = (dx * 2 + dy) * 2 + dz
__device__ __forceinline__ void brick_cell8(const Bricks& B, int x0, int y0, int z0, int b, bool want, int (&out)[8]) {
    unsigned long long key[1] = {scn_pack_key(x0 >> 2, y0 >> 2, z0 >> 2, b)}, occ[1];
    bool w[1] = {want};
    int s[1];
    brick_find_n<1>(B, key, w, s, occ);
    const int* e = B.ent + (size_t)(s[0] >= 0 ? s[0] : 0) * 64;
    int v[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) v[o] = e[brick_local(x0 + (o >> 2), y0 + ((o >> 1) & 1), z0 + (o & 1))];
#pragma unroll
    for (int o = 0; o < 8; ++o)
        out[o] = ((occ[0] >> brick_local(x0 + (o >> 2), y0 + ((o >> 1) & 1), z0 + (o & 1))) & 1ull) ? v[o] : -1;
}

// Width of a pass of the SubM mask sort at level l.  Round 6: with `lbins` (the bf16 tile kernels' builds) LEVEL 0 sorts by
// (row bin, mask): 27 + 3 bits in three 10-bit passes -- its tiles then take their 16 rows from one region of the scene and the
// 3^3 neighbours consecutive tiles gather meet in L2 (600 k voxels, C = 32: 71 -> 53 us per bf16 launch although the tiles
// visit 10 % more offsets; deeper levels lose 3-7 % and keep the plain mask sort: profiles/r6_bin_tiles.txt).
__device__ __forceinline__ int mask_sort_width(const PA& a, int l) { return (a.lbins && l == 0) ? 10 : 9; }

__device__ __forceinline__ unsigned gray_rank(unsigned m) {      // binary value whose reflected Gray code is m
    m ^= m >> 1; m ^= m >> 2; m ^= m >> 4; m ^= m >> 8; m ^= m >> 16;
    return m;
}

// ---------------------------------------------------------------------------------------------------------------------
// fill: three regions (hash keys = EMPTY, hash rows = INT_MAX, the zero region), 16 bytes per thread and step
// ---------------------------------------------------------------------------------------------------------------------
struct FillA { uint4* p[3]; long long n16[3]; unsigned v[3]; };

__global__ __launch_bounds__(T) void k_fill(FillA a) {
    for (int j = 0; j < 3; ++j) {
        const uint4 v = make_uint4(a.v[j], a.v[j], a.v[j], a.v[j]);
        for (long long i = blockIdx.x * (long long)T + threadIdx.x; i < a.n16[j]; i += (long long)gridDim.x * T) a.p[j][i] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// level 0: points -> voxels.  int64 -> int32 coordinates (range check) + hash insert with atomicMin of the point index
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(T) void k_insert0(const long long* __restrict__ in, long long n, int4* __restrict__ c32,
                                               unsigned long long* __restrict__ keys, int* __restrict__ tmin, long long cap,
                                               int* __restrict__ slot_of, long long* __restrict__ dsz) {
    const unsigned long long mask = (unsigned long long)cap - 1ull;
    int bad = 0;
    for (long long i = blockIdx.x * (long long)T + threadIdx.x; i < n; i += (long long)gridDim.x * T) {
        const long long x = in[4 * i], y = in[4 * i + 1], z = in[4 * i + 2], b = in[4 * i + 3];
        // 16 bits per key field; the batch column stops one short so that no site packs to SCN_EMPTY_KEY
        const bool oob = x < 0 || x > 65535 || y < 0 || y > 65535 || z < 0 || z > 65535 || b < 0 || b > 65534;
        bad |= oob;
        c32[i] = make_int4((int)x, (int)y, (int)z, (int)b);
        int found = -1;
        if (!oob) {
            const unsigned long long key = scn_pack_key((int)x, (int)y, (int)z, (int)b);
            unsigned long long slot = scn_hash_slot(key, mask);
            for (long long probe = 0; probe < cap; ++probe) {
                const unsigned long long prev = atomicCAS(&keys[slot], SCN_EMPTY_KEY, key);
                if (prev == SCN_EMPTY_KEY || prev == key) {
                    atomicMin(&tmin[slot], (int)i);
                    found = (int)slot;
                    break;
                }
                slot = (slot + 1) & mask;
            }
        }
        slot_of[i] = found;
    }
    const unsigned long long m = __ballot(bad != 0);
    if (m && (threadIdx.x & 63) == 0) atomicAdd((unsigned long long*)&dsz[DS_BAD], 1ull);
}

// ---------------------------------------------------------------------------------------------------------------------
// one-pass numbering: flags in item order -> positions, through a decoupled look-back over the workgroups' counts
// ---------------------------------------------------------------------------------------------------------------------
// virtual workgroup id from a ticket: a workgroup that holds ticket v knows that 0 .. v-1 are running or done, so waiting
// for their status words cannot deadlock whatever order the hardware starts workgroups in
__device__ __forceinline__ int take_ticket(int* ticket) {
    __shared__ int s_vb;
    if (threadIdx.x == 0) s_vb = atomicAdd(ticket, 1);
    __syncthreads();
    return s_vb;
}

// flags f[it] of items base + it*T + tid  ->  lp[it] = flagged items of the workgroup before this one, returns the count
__device__ __forceinline__ int block_rank(const bool* f, int* lp) {
    __shared__ int wcnt[IT][T / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int mine[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const unsigned long long m = __ballot(f[it]);
        mine[it] = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wcnt[it][w] = __popcll(m);
    }
    __syncthreads();
    int run = 0;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
#pragma unroll
        for (int k = 0; k < T / 64; ++k) {
            if (k == w) lp[it] = run + mine[it];
            run += wcnt[it][k];
        }
    }
    return run;
}

// the same for the 1024-thread workgroups of the numbering kernels: one item per thread, item = base + tid
constexpr int TT = 1024;               // threads per workgroup of k_number0 / k_coarsen / k_tables (TT == TILE items)
__device__ __forceinline__ int block_rank1(bool f, int* lp) {
    __shared__ int wcnt[TT / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned long long m = __ballot(f);
    if (lane == 0) wcnt[w] = __popcll(m);
    __syncthreads();
    int run = 0, before = 0;
#pragma unroll
    for (int k = 0; k < TT / 64; ++k) {
        before += k < w ? wcnt[k] : 0;
        run += wcnt[k];
    }
    *lp = before + __popcll(m & ((1ull << lane) - 1ull));
    return run;
}

// G probes of one table in flight per thread: the G first-slot key loads are issued together, then the G row loads (a miss
// reads row slot 0 -- one hot line -- so that every load stays unconditional: loads under divergent branches make the
// compiler wait for each), and only keys that met a FOREIGN key in their first slot walk on.  want[j] false: out[j] = -1.
template <int G>
__device__ __forceinline__ void lookup_n(const unsigned long long* __restrict__ keys, const int* __restrict__ rows,
                                         unsigned long long mask, const unsigned long long (&key)[G], const bool (&want)[G],
                                         int (&out)[G]) {
    unsigned long long slot[G], k[G];
#pragma unroll
    for (int j = 0; j < G; ++j) slot[j] = scn_hash_slot(key[j], mask);
#pragma unroll
    for (int j = 0; j < G; ++j) k[j] = keys[slot[j]];
    int r[G];
#pragma unroll
    for (int j = 0; j < G; ++j) r[j] = rows[(want[j] && k[j] == key[j]) ? slot[j] : 0ull];
#pragma unroll
    for (int j = 0; j < G; ++j) {
        out[j] = (want[j] && k[j] == key[j]) ? r[j] : -1;
        if (want[j] && k[j] != key[j] && k[j] != SCN_EMPTY_KEY) {          // collision in the first slot: the rare slow path
            unsigned long long s2 = (slot[j] + 1) & mask;
            for (unsigned long long probe = 0; probe < mask; ++probe) {
                const unsigned long long kk = keys[s2];
                if (kk == key[j]) { out[j] = rows[s2]; break; }
                if (kk == SCN_EMPTY_KEY) break;
                s2 = (s2 + 1) & mask;
            }
        }
    }
}

// exclusive prefix of `agg` over the virtual workgroups 0 .. vb-1 (status: 2 flag bits | 30 value bits, zero = not yet).
// The status word is its own payload (one naturally aligned 4-byte granule, agent-scope store / load: MI355X_MICROARCH
// "Valid forms" R2), so no fence is involved; wave 0 polls 64 predecessors per step.
__device__ __forceinline__ int lookback(unsigned* __restrict__ status, int vb, int agg, long long* __restrict__ dsz) {
    __shared__ int s_excl;
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        if (lane == 0) st_status(&status[vb], (vb == 0 ? F_INCL : F_AGG) | (unsigned)agg);
        int excl = 0, spins = 0;
        for (int p = vb - 1; p >= 0;) {
            const int idx = p - lane;
            const unsigned s = idx >= 0 ? ld_status(&status[idx]) : F_INCL;
            const unsigned long long incl = __ballot((s >> 30) == 2u);
            const unsigned long long wait = __ballot((s >> 30) == 0u);
            const int first = incl ? __ffsll((long long)incl) - 1 : 63;            // lanes 0 .. first are needed
            const unsigned long long need = first >= 63 ? ~0ull : ((2ull << first) - 1ull);
            if (wait & need) {
                if (++spins > SPIN_MAX) {
                    if (lane == 0) dsz[DS_ERR] = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
                continue;
            }
            int v = lane <= first ? (int)(s & VMASK) : 0;
#pragma unroll
            for (int d = 32; d; d >>= 1) v += __shfl_xor(v, d);
            excl += v;
            if (incl) break;
            p -= 64;
        }
        if (lane == 0) {
            if (vb > 0) st_status(&status[vb], F_INCL | (unsigned)(excl + agg));
            s_excl = excl;
        }
    }
    __syncthreads();
    return s_excl;
}

// points -> level-0 rows: the first point of a voxel (atomicMin above) gets the next row, in point order
__global__ __launch_bounds__(TT) void k_number0(PA a, long long n, const int4* __restrict__ c32, const int* __restrict__ slot_of,
                                                int* __restrict__ row_first) {
    const int vb = take_ticket(&a.tickets[0]);
    const long long base = (long long)vb * TILE;
    if (base >= n) return;
    int* __restrict__ hrows = a.hrows;                       // level 0's table: holds the minimum point index per slot
    int4* __restrict__ coords = at<int4>(a, a.lv[0].coords);
    const long long i = base + threadIdx.x;
    const int slot = i < n ? slot_of[i] : -1;
    // (the row number is written over the minimum below while other workgroups still compare: a later point j of the same
    //  voxel reads either the minimum i or the row number, and both are < j -- its flag stays false)
    const bool f = slot >= 0 && hrows[slot] == (int)i;
    int lp;
    const int agg = block_rank1(f, &lp);
    const int excl = lookback(at<unsigned>(a, a.lv[0].status), vb, agg, a.dsz);
    if (f) {
        const int pos = excl + lp;
        hrows[slot] = pos;
        row_first[pos] = (int)i;
        const int4 c = c32[i];
        coords[pos] = c;
        if (!a.no_bricks && !brick_insert(bricks_of(a, 0), c.x, c.y, c.z, c.w, pos)) a.dsz[DS_NOBRICK] = 1;
    }
    if (base + TILE >= n && threadIdx.x == 0) a.dsz[DS_N] = excl + agg;
}

// level l -> l + 1: fine row i is the FIRST row of its coarse site iff no other child of that site has a lower row number;
// the first rows get the coarse rows in ascending fine-row order (= first occurrence scanning the fine rows ascending, the
// canonical order of DESIGN.md section 2) and enter the coarse site into level l + 1's hash table
__global__ __launch_bounds__(TT) void k_coarsen(PA a, int l) {
    const int vb = take_ticket(&a.tickets[l + 1]);
    const long long n = a.dsz[DS_N + l];
    const long long base = (long long)vb * TILE;
    if (base >= n) {
        if (n == 0 && vb == 0 && threadIdx.x == 0) a.dsz[DS_N + l + 1] = 0;
        return;
    }
    const unsigned long long mask = (unsigned long long)a.cap - 1ull;
    const unsigned long long* __restrict__ fkeys = a.keys + (size_t)l * a.cap;
    const int* __restrict__ frows = a.hrows + (size_t)l * a.cap;
    unsigned long long* __restrict__ ckeys = a.keys + (size_t)(l + 1) * a.cap;
    int* __restrict__ crows = a.hrows + (size_t)(l + 1) * a.cap;
    const int4* __restrict__ fine = at<int4>(a, a.lv[l].coords);
    int4* __restrict__ coarse = at<int4>(a, a.lv[l + 1].coords);
    const long long i = base + threadIdx.x;
    const bool live = i < n;
    const int4 c = live ? fine[i] : make_int4(0, 0, 0, 0);
    unsigned long long key[8];
    bool want[8];
    int sib[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        key[o] = scn_pack_key((c.x & ~1) + (o >> 2), (c.y & ~1) + ((o >> 1) & 1), (c.z & ~1) + (o & 1), c.w);
        want[o] = live;
    }
    // (a word the kernels BEFORE this one may have set: the bricks of level l are complete whenever it reads zero)
    const bool use_bricks = !a.no_bricks && a.dsz[DS_NOBRICK] == 0;
    if (use_bricks) brick_cell8(bricks_of(a, l), c.x & ~1, c.y & ~1, c.z & ~1, c.w, live, sib);
    else lookup_n<8>(fkeys, frows, mask, key, want, sib);    // (its own offset answers i: not lower)
    bool f = live;
#pragma unroll
    for (int o = 0; o < 8; ++o) f = f && !(sib[o] >= 0 && sib[o] < (int)i);
    int lp;
    const int agg = block_rank1(f, &lp);
    const int excl = lookback(at<unsigned>(a, a.lv[l + 1].status), vb, agg, a.dsz);
    if (f) {
        const int pos = excl + lp;
        const int4 cc = make_int4(c.x >> 1, c.y >> 1, c.z >> 1, c.w);
        coarse[pos] = cc;
        const unsigned long long ck = scn_pack_key(cc.x, cc.y, cc.z, cc.w);
        unsigned long long slot = scn_hash_slot(ck, mask);
        for (long long probe = 0; probe < a.cap; ++probe) {          // every coarse site is inserted exactly once
            if (atomicCAS(&ckeys[slot], SCN_EMPTY_KEY, ck) == SCN_EMPTY_KEY) {
                crows[slot] = pos;
                break;
            }
            slot = (slot + 1) & mask;
        }
        if (!a.no_bricks && !brick_insert(bricks_of(a, l + 1), cc.x, cc.y, cc.z, cc.w, pos)) a.dsz[DS_NOBRICK] = 1;
    }
    if (base + TILE >= n && threadIdx.x == 0) a.dsz[DS_N + l + 1] = excl + agg;
}

// ---------------------------------------------------------------------------------------------------------------------
// tables: SubM table + sort key + first-pass digit counts + per-offset rule counts of every level, child tables likewise,
// parents / fine offsets, the points' rows -- one launch, job = blockIdx / nblk
// ---------------------------------------------------------------------------------------------------------------------
// 1024 rows per workgroup, one per thread: the probes of a row in groups of G in flight; per offset the workgroup's rule
// count (ballots), per row the sort key, per workgroup the digit counts of the first sort pass
template <int N_OFF, bool SUBM>
__device__ __forceinline__ void table_job(const PA& a, int l, int b) {
    // SUBM: rows of level l, probes of level l's own table at the k^3 offsets; else: rows of level l + 1 (coarse), probes of
    // level l's table at the 8 children
    const int lr = SUBM ? l : l + 1;
    const long long n = a.dsz[DS_N + lr];
    const LvA& L = a.lv[l];
    int* __restrict__ bsums = at<int>(a, SUBM ? L.bsums : L.cbsums);
    int* __restrict__ counts = at<int>(a, SUBM ? L.counts : L.ccounts);
    const long long base = (long long)b * TILE;
    __shared__ int wsum[N_OFF][TT / 64];
    __shared__ int hist[MAXB];
    const int width0 = SUBM ? mask_sort_width(a, l) : 8, bins = 1 << width0;
    if (base >= n) {                                     // surplus workgroup: its rule counts must read zero in the scan
        if (threadIdx.x < N_OFF) bsums[threadIdx.x * a.nblk + b] = 0;
        return;
    }
    for (int d = threadIdx.x; d < bins; d += TT) hist[d] = 0;
    const unsigned long long mask = (unsigned long long)a.cap - 1ull;
    const unsigned long long* __restrict__ keys = a.keys + (size_t)l * a.cap;
    const int* __restrict__ rows = a.hrows + (size_t)l * a.cap;
    const int4* __restrict__ coords = at<int4>(a, a.lv[lr].coords);
    int* __restrict__ table = at<int>(a, SUBM ? L.table : L.child);
    unsigned* __restrict__ key_out = at<unsigned>(a, SUBM ? L.key : L.ckey);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long r = base + threadIdx.x;
    const bool live = r < n;
    const int4 c = live ? coords[r] : make_int4(0, 0, 0, 0);
    unsigned m = 0;
    constexpr int G = SUBM ? 9 : 8;
    const bool use_bricks = !a.no_bricks && a.dsz[DS_NOBRICK] == 0;      // (every brick was filled by the kernels before this one)
    if (use_bricks) {
        const Bricks B = bricks_of(a, l);
        int v[N_OFF];
        if constexpr (SUBM) {
            // the 3 x 3 x 3 neighbourhood of (x, y, z) meets the bricks (bx0 | bx1, by0 | by1, bz0 | bz1): at most 8 directory
            // probes and occupancy words, then 27 loads out of those bricks
            const int bx0 = (c.x - 1) >> 2, by0 = (c.y - 1) >> 2, bz0 = (c.z - 1) >> 2;
            const int bx1 = (c.x + 1) >> 2, by1 = (c.y + 1) >> 2, bz1 = (c.z + 1) >> 2;
            unsigned long long key[8], occ[8];
            bool want[8];
            int bs[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int ix = j >> 2, iy = (j >> 1) & 1, iz = j & 1;
                const int bx = ix ? bx1 : bx0, by = iy ? by1 : by0, bz = iz ? bz1 : bz0;
                want[j] = live && (!ix || bx1 != bx0) && (!iy || by1 != by0) && (!iz || bz1 != bz0) &&
                          (unsigned)bx < 16384u && (unsigned)by < 16384u && (unsigned)bz < 16384u;
                key[j] = scn_pack_key(bx & 16383, by & 16383, bz & 16383, c.w);
            }
            brick_find_n<8>(B, key, want, bs, occ);
            int loc[N_OFF], sl[N_OFF];
            bool ok[N_OFF];
#pragma unroll
            for (int o = 0; o < N_OFF; ++o) {
                const int x = c.x + (o / 9 - 1), y = c.y + ((o / 3) % 3 - 1), z = c.z + (o % 3 - 1);
                const int j = (((x >> 2) != bx0) * 2 + ((y >> 2) != by0)) * 2 + ((z >> 2) != bz0);
                int sj = bs[0];
                unsigned long long oj = occ[0];
#pragma unroll
                for (int q = 1; q < 8; ++q) { sj = j == q ? bs[q] : sj; oj = j == q ? occ[q] : oj; }
                loc[o] = brick_local(x, y, z);
                ok[o] = live && sj >= 0 && ((oj >> loc[o]) & 1ull) && (unsigned)x < 65536u && (unsigned)y < 65536u && (unsigned)z < 65536u;
                sl[o] = sj >= 0 ? sj : 0;
            }
#pragma unroll
            for (int o = 0; o < N_OFF; ++o) v[o] = B.ent[(size_t)sl[o] * 64 + loc[o]];
#pragma unroll
            for (int o = 0; o < N_OFF; ++o) v[o] = ok[o] ? v[o] : -1;
        } else {
            int v8[8];
            brick_cell8(B, 2 * c.x, 2 * c.y, 2 * c.z, c.w, live, v8);          // the 8 children of coarse site c: one brick
#pragma unroll
            for (int o = 0; o < 8; ++o) v[o] = v8[o];
        }
#pragma unroll
        for (int o = 0; o < N_OFF; ++o) {
            if (live) table[(long long)o * n + r] = v[o];
            m |= (v[o] >= 0 ? 1u : 0u) << (SUBM ? a.kb.pos[o] : o);
            const int cnt = __popcll(__ballot(v[o] >= 0));
            if (lane == 0) wsum[o][w] = cnt;
        }
    } else
#pragma unroll
    for (int g = 0; g < N_OFF / G; ++g) {
        unsigned long long key[G];
        bool want[G];
        int v[G];
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int o = g * G + j;
            const int x = SUBM ? c.x + (o / 9 - 1) : 2 * c.x + (o >> 2), y = SUBM ? c.y + ((o / 3) % 3 - 1) : 2 * c.y + ((o >> 1) & 1),
                      z = SUBM ? c.z + (o % 3 - 1) : 2 * c.z + (o & 1);
            want[j] = live && (unsigned)x < 65536u && (unsigned)y < 65536u && (unsigned)z < 65536u;
            key[j] = scn_pack_key(x & 65535, y & 65535, z & 65535, c.w);
        }
        lookup_n<G>(keys, rows, mask, key, want, v);       // (the centre offset of a SubM table finds the row itself)
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int o = g * G + j;
            if (live) table[(long long)o * n + r] = v[j];
            m |= (v[j] >= 0 ? 1u : 0u) << (SUBM ? a.kb.pos[o] : o);
            const int cnt = __popcll(__ballot(v[j] >= 0));
            if (lane == 0) wsum[o][w] = cnt;
        }
    }
    __syncthreads();                                      // (also orders the hist zeroing before the adds)
    if (live) {
        unsigned kk = gray_rank(m);
        if (SUBM && a.lbins && l == 0) kk |= (unsigned)(((long long)r << a.lbins) / n) << 27;     // row bin above the 27 mask bits
        key_out[r] = kk;
        atomicAdd(&hist[kk & (unsigned)(bins - 1)], 1);
    }
    if (threadIdx.x < N_OFF) {
        int s = 0;
#pragma unroll
        for (int k = 0; k < TT / 64; ++k) s += wsum[threadIdx.x][k];
        bsums[threadIdx.x * a.nblk + b] = s;
    }
    __syncthreads();
    for (int d = threadIdx.x; d < bins; d += TT) counts[(long long)d * a.nblk + b] = hist[d];
}

__global__ __launch_bounds__(TT) void k_tables(PA a, const int* __restrict__ slot_of, long long n_points,
                                               int* __restrict__ item_row, int* __restrict__ row_count) {
    const int Lc = a.n_levels;
    const int job = blockIdx.x / a.nblk, b = blockIdx.x % a.nblk;
    if (job < Lc) {
        if (a.k == 3) table_job<27, true>(a, job, b);
        return;
    }
    if (job < 2 * Lc - 1) {
        table_job<8, false>(a, job - Lc, b);
        return;
    }
    if (job < 3 * Lc - 2) {                              // parents and offsets of the fine rows of level l
        const int l = job - (2 * Lc - 1);
        const long long n = a.dsz[DS_N + l];
        const unsigned long long mask = (unsigned long long)a.cap - 1ull;
        const unsigned long long* __restrict__ ckeys = a.keys + (size_t)(l + 1) * a.cap;
        const int* __restrict__ crows = a.hrows + (size_t)(l + 1) * a.cap;
        const int4* __restrict__ fine = at<int4>(a, a.lv[l].coords);
        int* __restrict__ parent = at<int>(a, a.lv[l].parent);
        int* __restrict__ fine_off = at<int>(a, a.lv[l].fine_off);
        const bool use_bricks = !a.no_bricks && a.dsz[DS_NOBRICK] == 0;
        const Bricks B = bricks_of(a, l + 1);
        for (long long i = (long long)b * TT + threadIdx.x; i < n; i += (long long)a.nblk * TT) {
            const int4 c = fine[i];
            if (use_bricks) {                                // (the parent exists: it was numbered from this very row's cell)
                const int px = c.x >> 1, py = c.y >> 1, pz = c.z >> 1;
                unsigned long long key[1] = {scn_pack_key(px >> 2, py >> 2, pz >> 2, c.w)}, occ[1];
                bool w1[1] = {true};
                int s1[1];
                brick_find_n<1>(B, key, w1, s1, occ);
                const int loc = brick_local(px, py, pz);
                parent[i] = (s1[0] >= 0 && ((occ[0] >> loc) & 1ull)) ? B.ent[(size_t)s1[0] * 64 + loc] : -1;
            } else {
                parent[i] = lookup(ckeys, crows, mask, scn_pack_key(c.x >> 1, c.y >> 1, c.z >> 1, c.w));
            }
            fine_off[i] = ((c.x & 1) * 2 + (c.y & 1)) * 2 + (c.z & 1);
        }
        return;
    }
    // the points' rows and the multiplicity of every level-0 row (InputLayer mode 4 divides by it)
    for (long long i = (long long)b * TT + threadIdx.x; i < n_points; i += (long long)a.nblk * TT) {
        const int s = slot_of[i];
        const int r = s >= 0 ? a.hrows[s] : 0;
        item_row[i] = r;
        if (s >= 0) atomicAdd(&row_count[r], 1);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// mask sort, pass p of a stable LSD radix sort (scn_sort.hip's scheme: per-workgroup digit counts -> per-digit scan over the
// workgroups -> stable scatter), all levels in one launch; the rule scans / fills of pass 0 ride along
// ---------------------------------------------------------------------------------------------------------------------
struct SortJob { const unsigned* kin; const int* vin; unsigned* kout; int* vout; int* counts; int* totals; int shift, width; };

__device__ __forceinline__ SortJob mask_sort_job(const PA& a, int l, int pass) {
    const LvA& L = a.lv[l];
    SortJob j;
    j.counts = at<int>(a, L.counts);
    j.totals = at<int>(a, L.totals);
    j.width = mask_sort_width(a, l);
    j.shift = j.width * pass;
    // 27 bits, three passes: key -> key_s -> ktmp -> key_s (the last pass lands in the output)
    j.kin = pass == 0 ? at<unsigned>(a, L.key) : pass == 1 ? at<unsigned>(a, L.key_s) : at<unsigned>(a, L.ktmp);
    j.vin = pass == 0 ? nullptr : pass == 1 ? at<int>(a, L.rows_s) : at<int>(a, L.vtmp);
    j.kout = pass == 1 ? at<unsigned>(a, L.ktmp) : at<unsigned>(a, L.key_s);
    j.vout = pass == 1 ? at<int>(a, L.vtmp) : at<int>(a, L.rows_s);
    return j;
}

__device__ __forceinline__ SortJob child_sort_job(const PA& a, int l) {
    const LvA& L = a.lv[l];
    SortJob j;
    j.counts = at<int>(a, L.ccounts);
    j.totals = at<int>(a, L.ctotals);
    j.shift = 0;
    j.width = 8;
    j.kin = at<unsigned>(a, L.ckey);
    j.vin = nullptr;
    j.kout = at<unsigned>(a, L.ckey_s);
    j.vout = at<int>(a, L.crows_s);
    return j;
}

// workgroup b counts the digits of its TILE items -> counts[digit][b]
__device__ __forceinline__ void hist_job(const SortJob& j, long long n, int b, int nblk) {
    __shared__ int hist[MAXB];
    const long long base = (long long)b * TILE;
    if (base >= n) return;
    const int bins = 1 << j.width;
    for (int d = threadIdx.x; d < bins; d += T) hist[d] = 0;
    __syncthreads();
    const unsigned dm = (unsigned)bins - 1u;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const long long e = base + it * T + threadIdx.x;
        if (e < n) atomicAdd(&hist[(j.kin[e] >> j.shift) & dm], 1);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < bins; d += T) j.counts[(long long)d * nblk + b] = hist[d];
}

// workgroup d: exclusive scan of counts[d][0 .. nb) in place, totals[d] = the digit's item count
__device__ __forceinline__ void digit_scan_job(const SortJob& j, long long n, int d, int nblk) {
    if (d >= (1 << j.width)) return;
    const int nb = (int)cdiv_dev(n);
    __shared__ int wtot[T / 64];
    __shared__ int carry_s;
    int* c = j.counts + (long long)d * nblk;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += T) {
        const int i = base + threadIdx.x;
        const int v = i < nb ? c[i] : 0;
        int x = v;
#pragma unroll
        for (int dl = 1; dl < 64; dl <<= 1) {
            const int y = __shfl_up(x, dl);
            if (lane >= dl) x += y;
        }
        if (lane == 63) wtot[w] = x;
        __syncthreads();
        int woff = 0;
#pragma unroll
        for (int k = 0; k < T / 64; ++k) woff += k < w ? wtot[k] : 0;
        const int carry = carry_s;
        if (i < nb) c[i] = carry + woff + x - v;
        __syncthreads();
        if (threadIdx.x == T - 1) carry_s = carry + woff + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) j.totals[d] = carry_s;
}

// exclusive scan of totals[0 .. bins) into LDS base[] (bins <= 512)
__device__ __forceinline__ void scan_totals(const int* __restrict__ totals, int bins, int* base, int* wtmp) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int carry = 0;
    for (int b0 = 0; b0 < bins; b0 += T) {
        const int d = b0 + threadIdx.x;
        const int v = d < bins ? totals[d] : 0;
        int x = v;
#pragma unroll
        for (int dl = 1; dl < 64; dl <<= 1) {
            const int y = __shfl_up(x, dl);
            if (lane >= dl) x += y;
        }
        if (lane == 63) wtmp[w] = x;
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int k = 0; k < T / 64; ++k) { woff += k < w ? wtmp[k] : 0; tot += wtmp[k]; }
        if (d < bins) base[d] = carry + woff + x - v;
        carry += tot;
        __syncthreads();
    }
}

// lanes of the wave holding the same digit as this lane (all 64 lanes take part; `valid` = this lane holds an item)
__device__ __forceinline__ unsigned long long peers_of(unsigned d, int width, bool valid) {
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 10; ++b) {
        if (b < width) {
            const unsigned long long m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
    }
    return peers;
}

// Stable scatter of workgroup b's items to gbase[digit] + (items of the digit in this workgroup before the item).  gbase (LDS,
// [bins]) = first output position of (digit, this workgroup), prepared by the caller.  A wave owns a CONTIGUOUS run of items
// (64 per load, IT loads), finds the lanes with its digit by one ballot per digit bit and keeps a wave-private running count
// per digit in LDS: item order = (wave, load, lane) = index order, so equal digits keep their input order.
__device__ __forceinline__ void stable_scatter(const unsigned* __restrict__ kin, const int* __restrict__ vin, long long n,
                                               long long base0, int shift, int width, int (*whist)[MAXB], const int* gbase,
                                               unsigned* __restrict__ kout, int* __restrict__ vout) {
    const int bins = 1 << width, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned dm = (unsigned)bins - 1u;
    const long long base = base0 + (long long)w * (64 * IT);
    unsigned key[IT], dig[IT];
    int val[IT], rank[IT];
    bool ok[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const long long e = base + it * 64 + lane;
        ok[it] = e < n;
        key[it] = ok[it] ? kin[e] : 0u;
        val[it] = ok[it] ? (vin ? vin[e] : (int)e) : 0;
        dig[it] = (key[it] >> shift) & dm;
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const unsigned long long peers = peers_of(dig[it], width, ok[it]);
        volatile int* cnt = &whist[w][dig[it]];                     // every lane of the group reads the same count, then
        const int before = *cnt;                                    // its lowest lane advances it (LDS operations of a wave
        rank[it] = before + __popcll(peers & ((1ull << lane) - 1ull));   // complete in program order)
        if (ok[it] && (peers & ((1ull << lane) - 1ull)) == 0) *cnt = before + __popcll(peers);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    for (int d = threadIdx.x; d < bins; d += T) {                   // counts per wave -> exclusive offsets per wave
        int run = 0;
#pragma unroll
        for (int k = 0; k < T / 64; ++k) { const int c = whist[k][d]; whist[k][d] = run; run += c; }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        if (ok[it]) {
            const int pos = gbase[dig[it]] + whist[w][dig[it]] + rank[it];
            kout[pos] = key[it];
            vout[pos] = val[it];
        }
    }
}

__device__ __forceinline__ void scatter_job(const SortJob& j, long long n, int b, int nblk) {
    __shared__ int whist[T / 64][MAXB];
    __shared__ int gbase[MAXB];
    __shared__ int wtmp[T / 64];
    const long long base = (long long)b * TILE;
    if (base >= n) return;
    const int bins = 1 << j.width;
    for (int d = threadIdx.x; d < bins; d += T) {
#pragma unroll
        for (int k = 0; k < T / 64; ++k) whist[k][d] = 0;
    }
    scan_totals(j.totals, bins, gbase, wtmp);                       // (ends with a barrier)
    for (int d = threadIdx.x; d < bins; d += T) gbase[d] += j.counts[(long long)d * nblk + b];
    __syncthreads();
    stable_scatter(j.kin, j.vin, n, base, j.shift, j.width, whist, gbase, j.kout, j.vout);
}

// one workgroup: exclusive scan, in flat (offset-major) order, of the per-(offset, workgroup) rule counts of a table;
// prefix[o] = first rule of offset o (device copy for the kernels, `dsz` copy for the host).  A thread owns a contiguous
// chunk: chunk sums -> one scan over the threads -> the chunk's running prefix.
__device__ __forceinline__ void rule_scan_job(int* __restrict__ bsums, int n_off, int nblk, long long* __restrict__ prefix,
                                              long long* __restrict__ dsz_prefix) {
    __shared__ int wtot[T / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int total = n_off * nblk;
    const int chunk = (total + T - 1) / T;
    const int lo = min(threadIdx.x * chunk, total), hi = min(lo + chunk, total);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += bsums[i];
    int x = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
    if (lane == 63) wtot[w] = x;
    __syncthreads();
    int woff = 0, all = 0;
#pragma unroll
    for (int k = 0; k < T / 64; ++k) { woff += k < w ? wtot[k] : 0; all += wtot[k]; }
    int run = woff + x - s;
    for (int i = lo; i < hi; ++i) {
        const int v = bsums[i];
        bsums[i] = run;
        if (i % nblk == 0) { prefix[i / nblk] = run; dsz_prefix[i / nblk] = run; }
        run += v;
    }
    if (threadIdx.x == 0) { prefix[n_off] = all; dsz_prefix[n_off] = all; }
}

// rules of the offsets seg0 .. seg0 + NSEG - 1 of workgroup b's 1024 rows: the table entries >= 0 in row order -> in_rows /
// out_rows at the scanned positions.  All NSEG x IT entries of a thread are requested up front (one latency, not NSEG).
template <int NSEG>
__device__ __forceinline__ void rule_fill_job(const int* __restrict__ table, long long n, int seg0, int b, int nblk,
                                              const int* __restrict__ offs, int* __restrict__ in_rows,
                                              int* __restrict__ out_rows) {
    const long long base = (long long)b * TILE;
    if (base >= n) return;                               // (uniform over the workgroup)
    int v[NSEG][IT];
#pragma unroll
    for (int sg = 0; sg < NSEG; ++sg) {
        const int* __restrict__ t = table + (long long)(seg0 + sg) * n;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const long long idx = base + it * T + threadIdx.x;
            v[sg][it] = idx < n ? t[idx] : -1;
        }
    }
#pragma unroll
    for (int sg = 0; sg < NSEG; ++sg) {
        bool f[IT];
        int lp[IT];
#pragma unroll
        for (int it = 0; it < IT; ++it) f[it] = v[sg][it] >= 0;
        (void)block_rank(f, lp);
        const int off = offs[(long long)(seg0 + sg) * nblk + b];
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            if (f[it]) {
                in_rows[off + lp[it]] = v[sg][it];
                out_rows[off + lp[it]] = (int)(base + it * T + threadIdx.x);
            }
        }
        __syncthreads();                                  // block_rank's LDS words are reused by the next offset
    }
}

// which: 0 = per-workgroup digit counts (passes 1, 2), 1 = per-digit scans (+ rule scans in pass 0), 2 = scatter (+ the child
// tables' single 8-bit pass and the SubM rule fills in pass 0, the child rule fills in pass 1)
__global__ __launch_bounds__(T) void k_sort(PA a, int pass, int which) {
    const int Lc = a.n_levels, nb = a.nblk;
    int blk = blockIdx.x;
    if (which == 0) {                                                // jobs: level l x nb
        const int l = blk / nb, b = blk % nb;
        hist_job(mask_sort_job(a, l, pass), a.dsz[DS_N + l], b, nb);
        return;
    }
    if (which == 1) {                                                // jobs: level l x 512 digits | child l x 256 | rule scans
        if (blk < Lc * MAXB) {
            const int l = blk / MAXB;
            digit_scan_job(mask_sort_job(a, l, pass), a.dsz[DS_N + l], blk % MAXB, nb);
            return;
        }
        blk -= Lc * MAXB;
        if (pass != 0) return;
        if (blk < (Lc - 1) * 256) {
            const int l = blk / 256;
            digit_scan_job(child_sort_job(a, l), a.dsz[DS_N + l + 1], blk % 256, nb);
            return;
        }
        blk -= (Lc - 1) * 256;
        if (blk < Lc) {
            rule_scan_job(at<int>(a, a.lv[blk].bsums), 27, nb, at<long long>(a, a.lv[blk].prefix), a.dsz + DS_SP + 28 * blk);
            return;
        }
        blk -= Lc;
        if (blk < Lc - 1)
            rule_scan_job(at<int>(a, a.lv[blk].cbsums), 8, nb, at<long long>(a, a.lv[blk].cprefix), a.dsz + DS_CP + 9 * blk);
        return;
    }
    // scatter
    if (blk < Lc * nb) {
        const int l = blk / nb;
        scatter_job(mask_sort_job(a, l, pass), a.dsz[DS_N + l], blk % nb, nb);
        return;
    }
    blk -= Lc * nb;
    if (pass == 0) {
        if (blk < (Lc - 1) * nb) {
            const int l = blk / nb;
            scatter_job(child_sort_job(a, l), a.dsz[DS_N + l + 1], blk % nb, nb);
            return;
        }
        blk -= (Lc - 1) * nb;
        if (blk < Lc * 3 * nb) {                        // a workgroup fills 9 offsets of its 1024 rows
            const int l = blk / (3 * nb), r = blk % (3 * nb);
            rule_fill_job<9>(at<int>(a, a.lv[l].table), a.dsz[DS_N + l], (r / nb) * 9, r % nb, nb, at<int>(a, a.lv[l].bsums),
                             at<int>(a, a.lv[l].in_rows), at<int>(a, a.lv[l].out_rows));
        }
        return;
    }
    if (pass == 1 && blk < (Lc - 1) * nb) {               // child rules: all 8 offsets of a workgroup's 1024 coarse rows
        const int l = blk / nb;
        rule_fill_job<8>(at<int>(a, a.lv[l].child), a.dsz[DS_N + l + 1], 0, blk % nb, nb, at<int>(a, a.lv[l].cbsums),
                         at<int>(a, a.lv[l].cin_rows), at<int>(a, a.lv[l].cout_rows));
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// tiles (scn_tiles.hip's k_build_tiles with the row count read from `dsz`), SubM and child tables of every level in one launch
// ---------------------------------------------------------------------------------------------------------------------
template <int N_OFF, bool SUBM>
__device__ __forceinline__ void tiles_job(const PA& a, int l, int b, int nblk16) {
    const long long n = a.dsz[DS_N + (SUBM ? l : l + 1)];
    const long long nt = (n + 15) / 16;
    const LvA& L = a.lv[l];
    const int* __restrict__ table = at<int>(a, SUBM ? L.table : L.child);
    const int* __restrict__ sorted_rows = at<int>(a, SUBM ? L.rows_s : L.crows_s);
    const unsigned* __restrict__ sorted_key = at<unsigned>(a, SUBM ? L.key_s : L.ckey_s);
    int* __restrict__ perm = at<int>(a, SUBM ? L.perm : L.cperm);
    int* __restrict__ tstab = at<int>(a, SUBM ? L.tstab : L.ctstab);
    unsigned* __restrict__ tile_mask = at<unsigned>(a, SUBM ? L.tmask : L.ctmask);
    unsigned* __restrict__ tile_cost = at<unsigned>(a, SUBM ? L.cost : L.ccost);
    unsigned* __restrict__ tile_xkey = (SUBM && a.with_x) ? at<unsigned>(a, L.xkey) : nullptr;
    // one thread per (tile, lane i); 16 threads of a tile are adjacent
    for (long long e = (long long)b * T + threadIdx.x; e < nt * 16; e += (long long)nblk16 * T) {
        const long long t = e >> 4;
        const int i = (int)(e & 15);
        const bool ok = e < n;
        const int row = ok ? sorted_rows[e] : -1;
        perm[e] = row;
        const unsigned rank = ok ? sorted_key[e] & 0x7FFFFFFu : 0u;      // (without the row-bin bits of a binned key)
        const unsigned key = rank ^ (rank >> 1);            // Gray code of the rank = the (bit-permuted) mask
        unsigned m = 0;
#pragma unroll 1
        for (int o = 0; o < N_OFF; ++o) {
            const unsigned have = (key >> (SUBM ? a.kb.pos[o] : o)) & 1u;
            m |= have << o;
            tstab[(t * N_OFF + o) * 16 + i] = have ? table[(long long)o * n + row] : -1;
        }
        m |= __shfl_xor(m, 1);
        m |= __shfl_xor(m, 2);
        m |= __shfl_xor(m, 4);
        m |= __shfl_xor(m, 8);
        if (i == 0) {
            tile_mask[t] = m;
            tile_cost[t] = 32u - (unsigned)__popc(m);
            if (tile_xkey) tile_xkey[t] = ((unsigned)(((long long)(row < 0 ? 0 : row) * 8) / n) << 6) | (32u - (unsigned)__popc(m));
        }
    }
}

__global__ __launch_bounds__(T) void k_tiles(PA a, int nblk16) {
    const int Lc = a.n_levels;
    const int job = blockIdx.x / nblk16, b = blockIdx.x % nblk16;
    if (job < Lc) {
        if (a.k == 3) tiles_job<27, true>(a, job, b, nblk16);
    } else {
        tiles_job<8, false>(a, job - Lc, b, nblk16);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// tile orders: ONE-pass stable counting sorts of the tile ids (<= 9 key bits, a few thousand tiles) in one launch.  No
// histogram launch, no look-back: every workgroup counts the digits of ALL tiles before its own run itself (nt keys, L2-hot).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void order_job(const unsigned* __restrict__ keys, long long nt, int width, int b,
                                          int* __restrict__ order, int* __restrict__ bin_start) {
    __shared__ int whist[T / 64][MAXB];
    __shared__ int gbase[MAXB];          // digit totals -> exclusive scan -> + items of the digit before this workgroup
    __shared__ int before[MAXB];
    __shared__ int wtmp[T / 64];
    const long long base = (long long)b * TILE;
    if (base >= nt) return;
    const int bins = 1 << width;
    const unsigned dm = (unsigned)bins - 1u;
    for (int d = threadIdx.x; d < bins; d += T) {
        gbase[d] = 0;
        before[d] = 0;
#pragma unroll
        for (int k = 0; k < T / 64; ++k) whist[k][d] = 0;
    }
    __syncthreads();
    // (one LDS atomic per digit GROUP of a wave's 64 keys, not per key: tile costs take ~20 distinct values, and 64 atomics
    //  on one LDS word serialise; a wave's 64 keys lie wholly before `base` or wholly not, TILE being a multiple of 64)
    for (long long e0 = 0; e0 < nt; e0 += 8 * T) {        // 8 independent key loads in flight per thread, then the counting
        unsigned dd[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const long long e = e0 + j * T + threadIdx.x;
            dd[j] = e < nt ? (keys[e] & dm) : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const long long e = e0 + j * T + threadIdx.x;
            const bool ok = dd[j] != 0xFFFFFFFFu;
            const unsigned d = ok ? dd[j] : 0u;
            const unsigned long long peers = peers_of(d, width, ok);
            if (ok && (peers & ((1ull << (threadIdx.x & 63)) - 1ull)) == 0) {
                const int c = __popcll(peers);
                atomicAdd(&gbase[d], c);
                if (e < base) atomicAdd(&before[d], c);
            }
        }
    }
    __syncthreads();
    if (bin_start && b == 0 && threadIdx.x <= 8) {       // XCD-local order: first position of every spatial bin (key >> 6)
        int s = 0;
        for (int d = 0; d < bins; ++d) s += (d >> 6) < threadIdx.x ? gbase[d] : 0;
        bin_start[threadIdx.x] = threadIdx.x == 8 ? (int)nt : s;
    }
    __syncthreads();
    {                                                     // exclusive scan of the totals (bins <= 512 = two rounds of T)
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        int carry = 0;
        for (int b0 = 0; b0 < bins; b0 += T) {
            const int d = b0 + threadIdx.x;
            const int v = d < bins ? gbase[d] : 0;
            int x = v;
#pragma unroll
            for (int dl = 1; dl < 64; dl <<= 1) {
                const int y = __shfl_up(x, dl);
                if (lane >= dl) x += y;
            }
            if (lane == 63) wtmp[w] = x;
            __syncthreads();
            int woff = 0, tot = 0;
#pragma unroll
            for (int k = 0; k < T / 64; ++k) { woff += k < w ? wtmp[k] : 0; tot += wtmp[k]; }
            if (d < bins) gbase[d] = carry + woff + x - v + before[d];
            carry += tot;
            __syncthreads();
        }
    }
    // (keys_out is not needed by anybody: the sorted keys go to a dummy slot of the caller's choosing -- here: nowhere)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long wbase = base + (long long)w * (64 * IT);
    unsigned dig[IT];
    int rank[IT];
    bool ok[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const long long e = wbase + it * 64 + lane;
        ok[it] = e < nt;
        dig[it] = ok[it] ? (keys[e] & dm) : 0u;
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const unsigned long long peers = peers_of(dig[it], width, ok[it]);
        volatile int* cnt = &whist[w][dig[it]];
        const int bf = *cnt;
        rank[it] = bf + __popcll(peers & ((1ull << lane) - 1ull));
        if (ok[it] && (peers & ((1ull << lane) - 1ull)) == 0) *cnt = bf + __popcll(peers);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    for (int d = threadIdx.x; d < bins; d += T) {
        int run = 0;
#pragma unroll
        for (int k = 0; k < T / 64; ++k) { const int c = whist[k][d]; whist[k][d] = run; run += c; }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < IT; ++it)
        if (ok[it]) order[gbase[dig[it]] + whist[w][dig[it]] + rank[it]] = (int)(wbase + it * 64 + lane);
}

__global__ __launch_bounds__(T) void k_orders(PA a, int nblk_t) {
    const int Lc = a.n_levels;
    const int job = blockIdx.x / nblk_t, b = blockIdx.x % nblk_t;
    if (job < Lc) {                                      // SubM tiles by offset count descending (LPT hand-out order)
        if (a.k != 3) return;
        const long long nt = (a.dsz[DS_N + job] + 15) / 16;
        order_job(at<unsigned>(a, a.lv[job].cost), nt, 6, b, at<int>(a, a.lv[job].torder), nullptr);
    } else if (job < 2 * Lc - 1) {
        const int l = job - Lc;
        const long long nt = (a.dsz[DS_N + l + 1] + 15) / 16;
        order_job(at<unsigned>(a, a.lv[l].ccost), nt, 6, b, at<int>(a, a.lv[l].ctorder), nullptr);
    } else {                                             // the XCD-local hand-out order behind the first one + bin starts
        const int l = job - (2 * Lc - 1);
        if (a.k != 3) return;
        const long long nt = (a.dsz[DS_N + l] + 15) / 16;
        int* order_x = at<int>(a, a.lv[l].torder) + nt;
        order_job(at<unsigned>(a, a.lv[l].xkey), nt, 9, b, order_x, order_x + nt);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
struct Bump2 {
    int64_t used = 0;
    int64_t take(int64_t bytes) {
        used = (used + 255) & ~(int64_t)255;
        const int64_t off = used;
        used += bytes > 0 ? bytes : 16;
        return off;
    }
};

struct Plan {
    int64_t keys, hrows, zero, zero_end, dsz, tickets, row_count, status[MAXL], bkeys, bmask, bent, bcap;
    int64_t c32, slot0, item_row, row_first;
    int64_t off[MAXL][40];
    int64_t total, cap, nblk, bsums_ints, cbsums_ints;
};

enum {
    O_COORDS, O_TABLE, O_BSUMS, O_PREFIX, O_KEY, O_KEY_S, O_ROWS_S, O_KTMP, O_VTMP, O_COUNTS, O_TOTALS, O_PERM, O_TSTAB,
    O_TMASK, O_TORDER, O_COST, O_XKEY, O_IN, O_OUT, O_PARENT, O_FINE_OFF, O_CHILD, O_CBSUMS, O_CPREFIX, O_CKEY, O_CKEY_S,
    O_CROWS_S, O_CCOUNTS, O_CTOTALS, O_CPERM, O_CTSTAB, O_CTMASK, O_CTORDER, O_CCOST, O_CIN, O_COUT, O_END
};

Plan make_plan(int64_t n, int n_levels, int k, bool with_x) {
    Plan p{};
    Bump2 w;
    const int64_t cap = scn_hash_capacity(n), nblk = cdiv(n, TILE), nt = cdiv(n, 16);
    const int n_off = k * k * k;
    p.cap = cap; p.nblk = nblk;
    // brick directories: a quarter of the voxel tables' slots (a 4^3 brick of a surface scene holds ~10 rows: load < 0.2; a
    // scene with a brick per row overflows it and the build falls back to the voxel tables).  The keys sit right behind the
    // voxel keys (one fill region), the occupancy words inside the zero region.
    const int64_t bcap = cap / 4 > 1024 ? cap / 4 : 1024;
    p.bcap = bcap;
    p.keys = w.take((int64_t)n_levels * cap * 8 + (int64_t)n_levels * bcap * 8);
    p.bkeys = p.keys + (int64_t)n_levels * cap * 8;
    p.hrows = w.take((int64_t)n_levels * cap * 4);
    // ---- zero region
    p.zero = p.dsz = w.take(DS_LEN * 8);
    p.tickets = w.take(64 * 4);
    p.row_count = w.take(n * 4);
    for (int l = 0; l < n_levels; ++l) p.status[l] = w.take(nblk * 4);
    p.bmask = w.take((int64_t)n_levels * bcap * 8);
    w.used = (w.used + 255) & ~(int64_t)255;
    p.zero_end = w.used;
    p.bent = w.take((int64_t)n_levels * bcap * 64 * 4);          // (never initialised: valid where an occupancy bit is set)
    // ---- the rest
    p.c32 = w.take(n * 16);
    p.slot0 = w.take(n * 4);
    p.item_row = w.take(n * 4);
    p.row_first = w.take(n * 4);
    p.bsums_ints = n_off * nblk;
    p.cbsums_ints = 8 * nblk;
    for (int l = 0; l < n_levels; ++l) {
        int64_t* o = p.off[l];
        o[O_COORDS] = w.take(n * 16);
        if (k == 3) {
            o[O_TABLE] = w.take((int64_t)n_off * n * 4);
            o[O_BSUMS] = w.take(p.bsums_ints * 4);
            o[O_PREFIX] = w.take((n_off + 1) * 8);
            o[O_KEY] = w.take(n * 4); o[O_KEY_S] = w.take(n * 4); o[O_ROWS_S] = w.take(n * 4);
            o[O_KTMP] = w.take(n * 4); o[O_VTMP] = w.take(n * 4);
            o[O_COUNTS] = w.take((int64_t)MAXB * nblk * 4); o[O_TOTALS] = w.take(MAXB * 4);
            o[O_PERM] = w.take(nt * 16 * 4); o[O_TSTAB] = w.take(nt * n_off * 16 * 4); o[O_TMASK] = w.take(nt * 4);
            o[O_TORDER] = w.take(scn_tiles_order_ints(n, with_x ? 1 : 0) * 4);
            o[O_COST] = w.take(nt * 4); o[O_XKEY] = w.take(nt * 4);
            o[O_IN] = w.take((int64_t)n_off * n * 4); o[O_OUT] = w.take((int64_t)n_off * n * 4);
        }
        if (l + 1 < n_levels) {
            o[O_PARENT] = w.take(n * 4); o[O_FINE_OFF] = w.take(n * 4);
            o[O_CHILD] = w.take(8 * n * 4);
            o[O_CBSUMS] = w.take(p.cbsums_ints * 4); o[O_CPREFIX] = w.take(9 * 8);
            o[O_CKEY] = w.take(n * 4); o[O_CKEY_S] = w.take(n * 4); o[O_CROWS_S] = w.take(n * 4);
            o[O_CCOUNTS] = w.take((int64_t)256 * nblk * 4); o[O_CTOTALS] = w.take(256 * 4);
            o[O_CPERM] = w.take(nt * 16 * 4); o[O_CTSTAB] = w.take(nt * 8 * 16 * 4); o[O_CTMASK] = w.take(nt * 4);
            o[O_CTORDER] = w.take(nt * 4); o[O_CCOST] = w.take(nt * 4);
            o[O_CIN] = w.take(n * 4); o[O_COUT] = w.take(n * 4);         // every fine row is exactly one rule
        }
    }
    p.total = (w.used + 255) & ~(int64_t)255;
    return p;
}

KeyBits make_key_bits27() {
    KeyBits kb;
    for (int o = 0; o < 32; ++o) kb.pos[o] = (unsigned char)o;
    int next = 0;
    for (int cls = 0; cls <= 3; ++cls)                 // centre (0), faces (1), edges (2), corners (3): LSB -> MSB
        for (int o = 0; o < 27; ++o) {
            const int dx = o / 9 - 1, dy = (o / 3) % 3 - 1, dz = o % 3 - 1;
            if ((dx != 0) + (dy != 0) + (dz != 0) == cls) kb.pos[o] = (unsigned char)next++;
        }
    return kb;
}

struct HostWords {
    int64_t* words = nullptr;
    hipEvent_t ev = nullptr;
    bool ok = false;
    bool init() {
        if (ok) return true;
        if (hipHostMalloc((void**)&words, sizeof(int64_t) * DS_LEN, hipHostMallocDefault) != hipSuccess) return false;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return false;
        ok = true;
        return true;
    }
};
thread_local HostWords g_hw;        // one per calling thread (main thread, index helper thread)

}  // namespace

namespace scn {

int64_t pyramid2_workspace_bytes(int64_t n_points, int n_levels, int k) {
    return make_plan(n_points > 0 ? n_points : 1, n_levels, k, true).total + 4096;
}

int pyramid2_build(const int64_t* coords, int64_t n_points, int n_levels, int k, void* workspace, int64_t workspace_bytes,
                   int64_t* desc, int flags, scn_stream_t stream) {
    SCN_REQUIRE(coords && workspace && desc && n_points >= 1 && n_levels >= 1 && n_levels <= MAXL && k == 3);
    SCN_REQUIRE((int64_t)27 * n_points < 2147483647LL && ((uintptr_t)workspace & 255) == 0);
    const bool with_x = (flags & SCN_PYRAMID_XCD_ORDER) != 0;
    const Plan p = make_plan(n_points, n_levels, k, with_x);
    SCN_REQUIRE(p.total <= workspace_bytes);
    hipStream_t st = S(stream);
    char* base = (char*)workspace;
    if (!g_hw.init()) return scn::fail(SCN_EHIP, "%spinned host words for the level sizes could not be created", "");
    const int n_off = k * k * k;
    const int64_t n = n_points, nblk = p.nblk;

    PA a{};
    a.base = base;
    a.dsz = (long long*)(base + p.dsz);
    a.tickets = (int*)(base + p.tickets);
    a.keys = (unsigned long long*)(base + p.keys);
    a.hrows = (int*)(base + p.hrows);
    a.bkeys = (unsigned long long*)(base + p.bkeys);
    a.bmask = (unsigned long long*)(base + p.bmask);
    a.bent = (int*)(base + p.bent);
    a.bcap = p.bcap;
    a.no_bricks = scn::sw(scn::SW_PYRAMID_NO_BRICKS).set ? 1 : 0;
    a.cap = p.cap; a.bound = n; a.n_levels = n_levels; a.k = k; a.with_x = with_x ? 1 : 0; a.nblk = (int)nblk;
    a.lbins = (with_x && k == 3 && !(scn::sw(scn::SW_TB_NO_BINS).set && scn::sw(scn::SW_TB_NO_BINS).i != 0)) ? 3 : 0;
    a.kb = make_key_bits27();
    auto q = [](int64_t off) { return (uint32_t)(off >> 8); };
    for (int l = 0; l < n_levels; ++l) {
        const int64_t* o = p.off[l];
        LvA& L = a.lv[l];
        L.coords = q(o[O_COORDS]); L.status = q(p.status[l]);
        L.table = q(o[O_TABLE]); L.bsums = q(o[O_BSUMS]); L.prefix = q(o[O_PREFIX]); L.key = q(o[O_KEY]); L.key_s = q(o[O_KEY_S]);
        L.rows_s = q(o[O_ROWS_S]); L.ktmp = q(o[O_KTMP]); L.vtmp = q(o[O_VTMP]); L.counts = q(o[O_COUNTS]); L.totals = q(o[O_TOTALS]);
        L.perm = q(o[O_PERM]); L.tstab = q(o[O_TSTAB]); L.tmask = q(o[O_TMASK]); L.torder = q(o[O_TORDER]); L.cost = q(o[O_COST]);
        L.xkey = q(o[O_XKEY]); L.in_rows = q(o[O_IN]); L.out_rows = q(o[O_OUT]);
        L.parent = q(o[O_PARENT]); L.fine_off = q(o[O_FINE_OFF]); L.child = q(o[O_CHILD]); L.cbsums = q(o[O_CBSUMS]);
        L.cprefix = q(o[O_CPREFIX]); L.ckey = q(o[O_CKEY]); L.ckey_s = q(o[O_CKEY_S]); L.crows_s = q(o[O_CROWS_S]);
        L.ccounts = q(o[O_CCOUNTS]); L.ctotals = q(o[O_CTOTALS]); L.cperm = q(o[O_CPERM]); L.ctstab = q(o[O_CTSTAB]);
        L.ctmask = q(o[O_CTMASK]); L.ctorder = q(o[O_CTORDER]); L.ccost = q(o[O_CCOST]); L.cin_rows = q(o[O_CIN]); L.cout_rows = q(o[O_COUT]);
    }

    // ---- queue everything; nothing below waits for the device ------------------------------------------------------
    FillA fa{};
    fa.p[0] = (uint4*)(base + p.keys);  fa.n16[0] = ((int64_t)n_levels * p.cap * 8 + (int64_t)n_levels * p.bcap * 8) / 16; fa.v[0] = 0xFFFFFFFFu;
    fa.p[1] = (uint4*)(base + p.hrows); fa.n16[1] = (int64_t)n_levels * p.cap * 4 / 16; fa.v[1] = 0x7FFFFFFFu;
    fa.p[2] = (uint4*)(base + p.zero);  fa.n16[2] = (p.zero_end - p.zero) / 16;          fa.v[2] = 0u;
    hipLaunchKernelGGL(k_fill, dim3(scn::ew_grid(fa.n16[0], T)), dim3(T), 0, st, fa);
    SCN_LAUNCH_CHECK();
    int4* c32 = (int4*)(base + p.c32);
    int* slot0 = (int*)(base + p.slot0);
    hipLaunchKernelGGL(k_insert0, dim3(scn::ew_grid(n, T)), dim3(T), 0, st, (const long long*)coords, (long long)n, c32, a.keys,
                       a.hrows, (long long)p.cap, slot0, a.dsz);
    SCN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_number0, dim3((unsigned)nblk), dim3(TT), 0, st, a, (long long)n, (const int4*)c32, (const int*)slot0,
                       (int*)(base + p.row_first));
    SCN_LAUNCH_CHECK();
    for (int l = 0; l + 1 < n_levels; ++l) {
        hipLaunchKernelGGL(k_coarsen, dim3((unsigned)nblk), dim3(TT), 0, st, a, l);
        SCN_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_tables, dim3((unsigned)((3 * n_levels - 1) * nblk)), dim3(TT), 0, st, a, (const int*)slot0, (long long)n,
                       (int*)(base + p.item_row), (int*)(base + p.row_count));
    SCN_LAUNCH_CHECK();
    {
        const int L = n_levels;
        for (int pass = 0; pass < 3; ++pass) {
            if (pass > 0) {
                hipLaunchKernelGGL(k_sort, dim3((unsigned)(L * nblk)), dim3(T), 0, st, a, pass, 0);
                SCN_LAUNCH_CHECK();
            }
            const int64_t g1 = (int64_t)L * MAXB + (pass == 0 ? (int64_t)(L - 1) * 256 + L + (L - 1) : 0);
            hipLaunchKernelGGL(k_sort, dim3((unsigned)g1), dim3(T), 0, st, a, pass, 1);
            SCN_LAUNCH_CHECK();
            int64_t g2 = (int64_t)L * nblk;
            if (pass == 0) g2 += (int64_t)(L - 1) * nblk + (int64_t)L * 3 * nblk;
            if (pass == 1) g2 += (int64_t)(L - 1) * nblk;
            hipLaunchKernelGGL(k_sort, dim3((unsigned)g2), dim3(T), 0, st, a, pass, 2);
            SCN_LAUNCH_CHECK();
        }
    }
    const int64_t nblk16 = cdiv(cdiv(n, 16) * 16, T);
    hipLaunchKernelGGL(k_tiles, dim3((unsigned)((2 * n_levels - 1) * nblk16)), dim3(T), 0, st, a, (int)nblk16);
    SCN_LAUNCH_CHECK();
    const int64_t nblk_t = cdiv(cdiv(n, 16), TILE);
    hipLaunchKernelGGL(k_orders, dim3((unsigned)(((with_x ? 3 : 2) * n_levels - 1) * nblk_t)), dim3(T), 0, st, a, (int)nblk_t);
    SCN_LAUNCH_CHECK();
    // ---- the one host wait ------------------------------------------------------------------------------------------
    SCN_HIP(hipMemcpyAsync(g_hw.words, a.dsz, sizeof(int64_t) * DS_LEN, hipMemcpyDeviceToHost, st));
    SCN_HIP(hipEventRecord(g_hw.ev, st));
    SCN_HIP(hipEventSynchronize(g_hw.ev));
    const int64_t* hw = g_hw.words;

    for (int i = 0; i < SCN_PYRAMID_DESC_LEN; ++i) desc[i] = 0;
    desc[0] = n_levels; desc[1] = n_points; desc[2] = p.total; desc[3] = hw[DS_BAD];
    desc[4] = p.item_row; desc[5] = p.row_count; desc[6] = p.row_first; desc[7] = p.c32;
    if (hw[DS_ERR]) return scn::fail(SCN_EHIP, "%sindex build: a workgroup gave up waiting for its predecessors (look-back)", "");
    if (hw[DS_BAD]) return scn::fail(SCN_EHASH, "%scoordinates outside [0,65535] in %lld wave(s)", "", (long long)hw[DS_BAD]);
    for (int l = 0; l < n_levels; ++l) {
        int64_t* L = desc + 8 + l * SCN_PYRAMID_LEVEL_STRIDE;
        const int64_t* o = p.off[l];
        const int64_t nl = hw[DS_N + l];
        L[0] = nl; L[1] = p.cap; L[2] = o[O_COORDS]; L[3] = p.keys + (int64_t)l * p.cap * 8; L[4] = p.hrows + (int64_t)l * p.cap * 4;
        if (nl > 0 && k == 3) {
            L[5] = o[O_TABLE]; L[6] = o[O_BSUMS]; L[7] = p.bsums_ints; L[8] = o[O_PREFIX];
            L[9] = o[O_PERM]; L[10] = o[O_TSTAB]; L[11] = o[O_TMASK]; L[12] = o[O_TORDER]; L[13] = cdiv(nl, 16);
            for (int i = 0; i <= n_off; ++i) L[25 + i] = hw[DS_SP + 28 * l + i];
            L[64] = o[O_IN]; L[65] = o[O_OUT];
        }
        if (l + 1 < n_levels && nl > 0) {
            const int64_t nc = hw[DS_N + l + 1];
            L[14] = o[O_PARENT]; L[15] = o[O_FINE_OFF]; L[16] = o[O_CHILD]; L[17] = o[O_CBSUMS]; L[18] = p.cbsums_ints;
            L[19] = o[O_CPREFIX]; L[20] = o[O_CPERM]; L[21] = o[O_CTSTAB]; L[22] = o[O_CTMASK]; L[23] = o[O_CTORDER];
            L[24] = cdiv(nc, 16);
            for (int i = 0; i <= 8; ++i) L[53 + i] = hw[DS_CP + 9 * l + i];
            L[66] = o[O_CIN]; L[67] = o[O_COUT];
        }
    }
    return SCN_OK;
}

}  // namespace scn
