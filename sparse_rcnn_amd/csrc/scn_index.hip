// Index path of libscn_mi355x: voxel hash, first-occurrence row numbering, rule tables, wave-ballot
// compaction of rule tables into (in,out) rule lists, sparse ROI crop indicator.
//
// All of this is HBM/L2-bound integer work (SURVEY.md §8d regime (i)): kernels are grid-strided,
// reads/writes are coalesced along the row index, and the only cross-thread primitives are 64-bit
// atomicCAS / atomicMin on the hash table and wave64 ballot + popcount prefix sums for compaction.
#include "scn_common.h"

namespace scn {
thread_local char g_err[512] = "";
}

using scn::S;
using scn::cdiv;

extern "C" int scn_abi_version(void) { return SCN_ABI_VERSION; }
extern "C" const char* scn_last_error_string(void) { return scn::g_err; }

extern "C" int64_t scn_hash_capacity(int64_t n) {
    int64_t cap = 1024;
    while (cap < 2 * n) cap <<= 1;
    return cap;
}

// ------------------------------------------------------------------------------------------------
// coords int64 -> int32, range check
// ------------------------------------------------------------------------------------------------
__global__ void k_coords_to_i32(const long long* __restrict__ in, long long n4, int* __restrict__ out,
                                int* __restrict__ bad) {
    int local_bad = 0;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4;
         i += (long long)gridDim.x * blockDim.x) {
        long long v = in[i];
        // 16 bits per key field; the batch column stops one short so that no site packs to SCN_EMPTY_KEY
        local_bad += (v < 0 || v > 65535 || ((i & 3) == 3 && v > 65534));
        out[i] = (int)v;
    }
    unsigned long long m = __ballot(local_bad != 0);
    if (m && (threadIdx.x & 63) == 0) atomicAdd(bad, 1);
}

extern "C" int scn_coords_to_i32(const int64_t* coords, int64_t n, int32_t* out, int32_t* scratch1,
                                 int64_t* bad_host, scn_stream_t stream) {
    SCN_REQUIRE(n >= 0 && (n == 0 || (coords && out)) && scratch1);
    SCN_HIP(hipMemsetAsync(scratch1, 0, sizeof(int32_t), S(stream)));
    if (n) {
        hipLaunchKernelGGL(k_coords_to_i32, dim3(scn::ew_grid(n * 4, 256)), dim3(256), 0, S(stream),
                           (const long long*)coords, (long long)n * 4, out, scratch1);
        SCN_LAUNCH_CHECK();
    }
    if (!bad_host) return SCN_OK;          // asynchronous form: the caller reads *scratch1 behind its own event
    int32_t bad = 0;
    SCN_HIP(hipMemcpyAsync(&bad, scratch1, sizeof(bad), hipMemcpyDeviceToHost, S(stream)));
    SCN_HIP(hipStreamSynchronize(S(stream)));
    *bad_host = bad;
    if (bad) return scn::fail(SCN_EHASH, "%scoordinates outside [0,65535] in %lld wave(s)", "", bad);
    return SCN_OK;
}

// ------------------------------------------------------------------------------------------------
// flag-table scan: table[seg][n] int32, flag = value >= 0.  Block = 256 threads x 4 iterations.
// Order inside a block is idx = it*256 + tid, so ballot order == index order.
// ------------------------------------------------------------------------------------------------
static constexpr int SCAN_T = 256;
static constexpr int SCAN_IT = 4;
static constexpr int SCAN_TILE = SCAN_T * SCAN_IT;

__global__ __launch_bounds__(SCAN_T) void k_flag_count(const int* __restrict__ table, long long n,
                                                       int* __restrict__ block_sums) {
    const long long seg = blockIdx.y;
    const long long base = (long long)blockIdx.x * SCAN_TILE;
    const int* t = table + seg * n;
    int cnt = 0;
#pragma unroll
    for (int it = 0; it < SCAN_IT; ++it) {
        long long idx = base + it * SCAN_T + threadIdx.x;
        bool f = idx < n && t[idx] >= 0;
        cnt += __popcll(__ballot(f));
    }
    __shared__ int wsum[SCAN_T / 64];
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[seg * gridDim.x + blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// single block: exclusive scan of block_sums in place; prefix[seg] = offset of the segment's first block.
__global__ __launch_bounds__(1024) void k_scan_blocks(int* __restrict__ block_sums, long long nblocks,
                                                      long long blocks_per_seg, int n_seg,
                                                      long long* __restrict__ prefix) {
    __shared__ int wtot[16];
    __shared__ long long carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (long long base = 0; base < nblocks; base += 1024) {
        long long i = base + threadIdx.x;
        int v = i < nblocks ? block_sums[i] : 0;
        int x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            int y = __shfl_up(x, d);
            if (lane >= d) x += y;
        }
        if (lane == 63) wtot[w] = x;
        __syncthreads();
        int woff = 0;
        for (int k = 0; k < w; ++k) woff += wtot[k];
        long long carry = carry_s;
        long long excl = carry + woff + x - v;
        if (i < nblocks) {
            block_sums[i] = (int)excl;
            if (i % blocks_per_seg == 0) prefix[i / blocks_per_seg] = excl;
        }
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) prefix[n_seg] = carry_s;
}

// position of each flagged element inside its block, via ballot + popcount

__device__ __forceinline__ void block_positions(const int* __restrict__ t, long long n, long long base, int block_off,
                                                int* vals, int* pos, bool* flag) {
    __shared__ int wcnt[SCAN_IT][SCAN_T / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int lp[SCAN_IT];
#pragma unroll
    for (int it = 0; it < SCAN_IT; ++it) {
        long long idx = base + it * SCAN_T + threadIdx.x;
        int v = idx < n ? t[idx] : -1;
        vals[it] = v;
        flag[it] = v >= 0;
        unsigned long long m = __ballot(flag[it]);
        lp[it] = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wcnt[it][w] = __popcll(m);
    }
    __syncthreads();
    int run = block_off;
#pragma unroll
    for (int it = 0; it < SCAN_IT; ++it) {
#pragma unroll
        for (int k = 0; k < SCAN_T / 64; ++k) {
            if (k == w) pos[it] = run + lp[it];
            run += wcnt[it][k];
        }
    }
}

__global__ __launch_bounds__(SCAN_T) void k_rules_fill(const int* __restrict__ table, long long n,
                                                       const int* __restrict__ block_offs, int* __restrict__ in_rows,
                                                       int* __restrict__ out_rows, int* __restrict__ seg_of) {
    const long long seg = blockIdx.y;
    const long long base = (long long)blockIdx.x * SCAN_TILE;
    int vals[SCAN_IT], pos[SCAN_IT];
    bool flag[SCAN_IT];
    block_positions(table + seg * n, n, base, block_offs[seg * gridDim.x + blockIdx.x], vals, pos, flag);
#pragma unroll
    for (int it = 0; it < SCAN_IT; ++it) {
        if (flag[it]) {
            in_rows[pos[it]] = vals[it];
            out_rows[pos[it]] = (int)(base + it * SCAN_T + threadIdx.x);
            if (seg_of) seg_of[pos[it]] = (int)seg;
        }
    }
}

extern "C" int64_t scn_rules_blocks(int n_off, int64_t n_out) {
    return (int64_t)n_off * (cdiv(n_out, SCAN_TILE) > 0 ? cdiv(n_out, SCAN_TILE) : 1);
}

static int scan_launch(const int32_t* table, int n_seg, int64_t n, int32_t* block_sums, int64_t* prefix,
                       hipStream_t st) {
    int64_t bps = cdiv(n, SCAN_TILE);
    if (bps < 1) bps = 1;
    SCN_REQUIRE(bps < 2147483647 && n_seg < 65536);
    hipLaunchKernelGGL(k_flag_count, dim3((unsigned)bps, (unsigned)n_seg), dim3(SCAN_T), 0, st, table, (long long)n,
                       block_sums);
    SCN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, st, block_sums, (long long)(bps * n_seg),
                       (long long)bps, n_seg, (long long*)prefix);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

extern "C" int scn_rules_scan(const int32_t* table, int n_off, int64_t n_out, int32_t* block_sums, int64_t* prefix,
                              int64_t* prefix_host, scn_stream_t stream) {
    SCN_REQUIRE(n_off >= 1 && n_out >= 0 && block_sums && prefix && (n_out == 0 || table));
    SCN_REQUIRE((int64_t)n_off * n_out < 2147483647LL);
    int rc = scan_launch(table, n_off, n_out, block_sums, prefix, S(stream));
    if (rc) return rc;
    if (!prefix_host) return SCN_OK;       // asynchronous form: the caller copies `prefix` back behind its own event
    SCN_HIP(hipMemcpyAsync(prefix_host, prefix, sizeof(int64_t) * (n_off + 1), hipMemcpyDeviceToHost, S(stream)));
    SCN_HIP(hipStreamSynchronize(S(stream)));
    return SCN_OK;
}

extern "C" int scn_rules_fill(const int32_t* table, int n_off, int64_t n_out, const int32_t* block_sums,
                              int32_t* in_rows, int32_t* out_rows, int32_t* seg_of, scn_stream_t stream) {
    SCN_REQUIRE(n_off >= 1 && n_out >= 0 && block_sums);
    if (n_out == 0) return SCN_OK;
    SCN_REQUIRE(table && in_rows && out_rows);
    int64_t bps = cdiv(n_out, SCAN_TILE);
    hipLaunchKernelGGL(k_rules_fill, dim3((unsigned)bps, (unsigned)n_off), dim3(SCAN_T), 0, S(stream), table,
                       (long long)n_out, block_sums, in_rows, out_rows, seg_of);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

// ------------------------------------------------------------------------------------------------
// dedup: hash insert with atomicMin of the item index (deterministic first occurrence)
// ------------------------------------------------------------------------------------------------
__global__ void k_table_init(unsigned long long* __restrict__ keys, int* __restrict__ rows, long long cap) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < cap;
         i += (long long)gridDim.x * blockDim.x) {
        keys[i] = SCN_EMPTY_KEY;
        rows[i] = 0x7FFFFFFF;
    }
}

// How a fine coordinate becomes its coarse site: x >> shift (size = stride = 2^shift: the reference's 2^3/2 layers) or, with
// dx > 0, (x / dx, y / dy, z / dz) -- any filter_size = filter_stride the reference's `get_downsampler(stride=...)` can ask for
// (module_factory.py:221-241: an int or one entry per axis).  Coordinates are non-negative.
struct Coarsen { int shift, dx, dy, dz; };
__device__ __forceinline__ int4 scn_coarse(int4 c, Coarsen cs) {
    if (cs.dx == 0) return make_int4(c.x >> cs.shift, c.y >> cs.shift, c.z >> cs.shift, c.w);
    return make_int4(c.x / cs.dx, c.y / cs.dy, c.z / cs.dz, c.w);
}

__global__ void k_hash_insert_min(const int4* __restrict__ coords, long long n, Coarsen cs,
                                  unsigned long long* __restrict__ keys, int* __restrict__ tmin, long long cap,
                                  int* __restrict__ slot_of) {
    const unsigned long long mask = (unsigned long long)cap - 1ull;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const int4 c = scn_coarse(coords[i], cs);
        unsigned long long key = scn_pack_key(c.x, c.y, c.z, c.w);
        unsigned long long slot = scn_hash_slot(key, mask);
        int found = -1;
        for (long long probe = 0; probe < cap; ++probe) {
            unsigned long long prev = atomicCAS(&keys[slot], SCN_EMPTY_KEY, key);
            if (prev == SCN_EMPTY_KEY || prev == key) {
                atomicMin(&tmin[slot], (int)i);
                found = (int)slot;
                break;
            }
            slot = (slot + 1) & mask;
        }
        slot_of[i] = found;
    }
}

// first[i] = i if item i is the first occurrence of its key, else -1  (a [1][n] flag table)
__global__ void k_flag_first(const int* __restrict__ slot_of, const int* __restrict__ tmin, long long n,
                             int* __restrict__ first) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        int s = slot_of[i];
        first[i] = (s >= 0 && tmin[s] == (int)i) ? (int)i : -1;
    }
}

__global__ __launch_bounds__(SCAN_T) void k_assign_rows(const int* __restrict__ first, long long n,
                                                        const int* __restrict__ block_offs,
                                                        const int* __restrict__ slot_of,
                                                        const int4* __restrict__ coords, Coarsen cs,
                                                        int* __restrict__ table_rows, int* __restrict__ row_first,
                                                        int4* __restrict__ row_coords) {
    const long long base = (long long)blockIdx.x * SCAN_TILE;
    int vals[SCAN_IT], pos[SCAN_IT];
    bool flag[SCAN_IT];
    block_positions(first, n, base, block_offs[blockIdx.x], vals, pos, flag);
#pragma unroll
    for (int it = 0; it < SCAN_IT; ++it) {
        if (flag[it]) {
            int i = vals[it];
            table_rows[slot_of[i]] = pos[it];
            if (row_first) row_first[pos[it]] = i;
            row_coords[pos[it]] = scn_coarse(coords[i], cs);
        }
    }
}

__global__ void k_item_rows(const int* __restrict__ slot_of, const int* __restrict__ table_rows, long long n,
                            int* __restrict__ item_row, int* __restrict__ row_count) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        int r = table_rows[slot_of[i]];
        item_row[i] = r;
        if (row_count) atomicAdd(&row_count[r], 1);
    }
}

static inline int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }

extern "C" int64_t scn_dedup_scratch_bytes(int64_t n) {
    int64_t blocks = scn_rules_blocks(1, n);
    return align256(4 * n) * 2 + align256(4 * blocks) + 256;
}

static int dedup_impl(const int32_t* coords, int64_t n, Coarsen cs, uint64_t* table_keys,
                      int32_t* table_rows, int64_t cap, int32_t* item_row, int32_t* row_count,
                      int32_t* row_first, int32_t* row_coords, void* scratch, int64_t* n_rows_host,
                      int64_t* n_rows_dev, scn_stream_t stream) {
    SCN_REQUIRE(n >= 0 && cs.shift >= 0 && cs.shift < 16 && table_keys && table_rows && (n_rows_host || n_rows_dev) &&
                scratch);
    SCN_REQUIRE(cs.dx == 0 || (cs.dx >= 1 && cs.dy >= 1 && cs.dz >= 1));
    SCN_REQUIRE(cap >= 2 * n && (cap & (cap - 1)) == 0);
    SCN_REQUIRE(n < 2147483647LL);
    hipStream_t st = S(stream);
    hipLaunchKernelGGL(k_table_init, dim3(scn::ew_grid(cap, 256)), dim3(256), 0, st,
                       (unsigned long long*)table_keys, table_rows, (long long)cap);
    SCN_LAUNCH_CHECK();
    if (n == 0) {
        if (n_rows_host) *n_rows_host = 0;
        if (n_rows_dev) SCN_HIP(hipMemsetAsync(n_rows_dev, 0, sizeof(int64_t), st));
        return SCN_OK;
    }
    SCN_REQUIRE(coords && item_row && row_coords);
    char* p = (char*)scratch;
    int* slot_of = (int*)p;                 p += align256(4 * n);
    int* first = (int*)p;                   p += align256(4 * n);
    int64_t blocks = scn_rules_blocks(1, n);
    int* block_sums = (int*)p;              p += align256(4 * blocks);
    long long* prefix = (long long*)p;      // [2]
    const int g = scn::ew_grid(n, 256);
    hipLaunchKernelGGL(k_hash_insert_min, dim3(g), dim3(256), 0, st, (const int4*)coords, (long long)n, cs,
                       (unsigned long long*)table_keys, table_rows, (long long)cap, slot_of);
    SCN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_flag_first, dim3(g), dim3(256), 0, st, slot_of, table_rows, (long long)n, first);
    SCN_LAUNCH_CHECK();
    int rc = scan_launch(first, 1, n, block_sums, (int64_t*)prefix, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_assign_rows, dim3((unsigned)blocks), dim3(SCAN_T), 0, st, first, (long long)n, block_sums,
                       slot_of, (const int4*)coords, cs, table_rows, row_first, (int4*)row_coords);
    SCN_LAUNCH_CHECK();
    if (row_count) SCN_HIP(hipMemsetAsync(row_count, 0, sizeof(int32_t) * n, st));
    hipLaunchKernelGGL(k_item_rows, dim3(g), dim3(256), 0, st, slot_of, table_rows, (long long)n, item_row, row_count);
    SCN_LAUNCH_CHECK();
    if (n_rows_dev)
        SCN_HIP(hipMemcpyAsync(n_rows_dev, prefix + 1, sizeof(int64_t), hipMemcpyDeviceToDevice, st));
    if (n_rows_host) {
        long long host_prefix[2] = {0, 0};
        SCN_HIP(hipMemcpyAsync(host_prefix, prefix, sizeof(host_prefix), hipMemcpyDeviceToHost, st));
        SCN_HIP(hipStreamSynchronize(st));
        *n_rows_host = host_prefix[1];
    }
    return SCN_OK;
}

extern "C" int scn_dedup_build(const int32_t* coords, int64_t n, int shift, uint64_t* table_keys,
                               int32_t* table_rows, int64_t cap, int32_t* item_row, int32_t* row_count,
                               int32_t* row_first, int32_t* row_coords, void* scratch, int64_t* n_rows_host,
                               scn_stream_t stream) {
    SCN_REQUIRE(n_rows_host);
    return dedup_impl(coords, n, Coarsen{shift, 0, 0, 0}, table_keys, table_rows, cap, item_row, row_count, row_first, row_coords,
                      scratch, n_rows_host, nullptr, stream);
}

extern "C" int scn_dedup_launch(const int32_t* coords, int64_t n, int shift, uint64_t* table_keys,
                                int32_t* table_rows, int64_t cap, int32_t* item_row, int32_t* row_count,
                                int32_t* row_first, int32_t* row_coords, void* scratch, int64_t* n_rows_dev,
                                scn_stream_t stream) {
    SCN_REQUIRE(n_rows_dev);
    return dedup_impl(coords, n, Coarsen{shift, 0, 0, 0}, table_keys, table_rows, cap, item_row, row_count, row_first, row_coords,
                      scratch, nullptr, n_rows_dev, stream);
}

extern "C" int scn_dedup_launch_div(const int32_t* coords, int64_t n, int sx, int sy, int sz, uint64_t* table_keys,
                                    int32_t* table_rows, int64_t cap, int32_t* item_row, int32_t* row_count,
                                    int32_t* row_first, int32_t* row_coords, void* scratch, int64_t* n_rows_dev,
                                    scn_stream_t stream) {
    SCN_REQUIRE(n_rows_dev && sx >= 1 && sy >= 1 && sz >= 1);
    return dedup_impl(coords, n, Coarsen{0, sx, sy, sz}, table_keys, table_rows, cap, item_row, row_count, row_first, row_coords,
                      scratch, nullptr, n_rows_dev, stream);
}

// ------------------------------------------------------------------------------------------------
// rule tables
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int hash_lookup(const unsigned long long* __restrict__ keys,
                                           const int* __restrict__ rows, unsigned long long mask,
                                           unsigned long long key) {
    unsigned long long slot = scn_hash_slot(key, mask);
    for (unsigned long long probe = 0; probe <= mask; ++probe) {
        unsigned long long k = keys[slot];
        if (k == key) return rows[slot];
        if (k == SCN_EMPTY_KEY) return -1;
        slot = (slot + 1) & mask;
    }
    return -1;
}

__global__ void k_subm_table(const int4* __restrict__ coords, long long n, const unsigned long long* __restrict__ keys,
                             const int* __restrict__ rows, long long cap, int k, int* __restrict__ table) {
    const unsigned long long mask = (unsigned long long)cap - 1ull;
    const int h = k / 2;
    const long long total = (long long)k * k * k * n;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int o = (int)(idx / n);
        const long long r = idx - (long long)o * n;
        const int dz = o % k - h, dy = (o / k) % k - h, dx = o / (k * k) - h;
        int4 c = coords[r];
        int x = c.x + dx, y = c.y + dy, z = c.z + dz;
        int res = -1;
        if ((unsigned)x < 65536u && (unsigned)y < 65536u && (unsigned)z < 65536u)
            res = (dx == 0 && dy == 0 && dz == 0) ? (int)r : hash_lookup(keys, rows, mask, scn_pack_key(x, y, z, c.w));
        table[idx] = res;
    }
}

extern "C" int scn_subm_table(const int32_t* coords, int64_t n, const uint64_t* table_keys,
                              const int32_t* table_rows, int64_t cap, int k, int32_t* table, scn_stream_t stream) {
    SCN_REQUIRE(n >= 0 && k >= 1 && (k & 1) && k <= 7);
    if (n == 0) return SCN_OK;
    SCN_REQUIRE(coords && table_keys && table_rows && table && (cap & (cap - 1)) == 0);
    SCN_REQUIRE((int64_t)k * k * k * n < 2147483647LL);
    hipLaunchKernelGGL(k_subm_table, dim3(scn::ew_grid((int64_t)k * k * k * n, 256)), dim3(256), 0, S(stream),
                       (const int4*)coords, (long long)n, (const unsigned long long*)table_keys, table_rows,
                       (long long)cap, k, table);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

__global__ void k_child_table(const int4* __restrict__ fine, const int* __restrict__ parent, long long n,
                              long long n_coarse, int* __restrict__ child, int* __restrict__ fine_off) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        int4 c = fine[i];
        int o = ((c.x & 1) * 2 + (c.y & 1)) * 2 + (c.z & 1);
        child[(long long)o * n_coarse + parent[i]] = (int)i;
        fine_off[i] = o;
    }
}

// size = stride = (sx, sy, sz): child[o][coarse row] = fine row, o = ((x % sx) sy + y % sy) sz + z % sz (the 2^3 numbering above
// for sx = sy = sz = 2)
__global__ void k_child_table_div(const int4* __restrict__ fine, const int* __restrict__ parent, long long n,
                                  long long n_coarse, int sx, int sy, int sz, int* __restrict__ child,
                                  int* __restrict__ fine_off) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const int4 c = fine[i];
        const int o = ((c.x % sx) * sy + c.y % sy) * sz + c.z % sz;
        child[(long long)o * n_coarse + parent[i]] = (int)i;
        fine_off[i] = o;
    }
}

extern "C" int scn_child_table_div(const int32_t* fine_coords, const int32_t* parent, int64_t n_fine, int64_t n_coarse,
                                   int sx, int sy, int sz, int32_t* child, int32_t* fine_off, scn_stream_t stream) {
    SCN_REQUIRE(n_fine >= 0 && n_coarse >= 0 && sx >= 1 && sy >= 1 && sz >= 1);   // (n_coarse > n_fine: an existing, larger grid)
    SCN_REQUIRE((int64_t)sx * sy * sz <= 4096);
    if (n_fine == 0) return SCN_OK;
    SCN_REQUIRE(fine_coords && parent && child && fine_off);
    SCN_HIP(hipMemsetAsync(child, 0xFF, sizeof(int32_t) * (size_t)sx * sy * sz * n_coarse, S(stream)));
    hipLaunchKernelGGL(k_child_table_div, dim3(scn::ew_grid(n_fine, 256)), dim3(256), 0, S(stream),
                       (const int4*)fine_coords, parent, (long long)n_fine, (long long)n_coarse, sx, sy, sz, child, fine_off);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

// A size = stride Convolution INTO a grid that already exists (another path of the network reached the same spatial size on this
// Metadata; SparseConvNet keys its grids by spatial size): parent[i] = the existing grid's row of floor(fine[i] / stride), -1 and a
// count when that site is not in the grid.
__global__ void k_parent_lookup_div(const int4* __restrict__ fine, long long n, int sx, int sy, int sz,
                                    const unsigned long long* __restrict__ keys, const int* __restrict__ rows, long long cap,
                                    int* __restrict__ parent, unsigned long long* __restrict__ n_missing) {
    const unsigned long long mask = (unsigned long long)cap - 1ull;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const int4 c = fine[i];
        const int r = hash_lookup(keys, rows, mask, scn_pack_key(c.x / sx, c.y / sy, c.z / sz, c.w));
        parent[i] = r;
        if (r < 0) atomicAdd(n_missing, 1ull);
    }
}

extern "C" int scn_parent_lookup_div(const int32_t* fine_coords, int64_t n_fine, int sx, int sy, int sz,
                                     const uint64_t* table_keys, const int32_t* table_rows, int64_t cap, int32_t* parent,
                                     int64_t* n_missing_dev, scn_stream_t stream) {
    SCN_REQUIRE(n_fine >= 0 && sx >= 1 && sy >= 1 && sz >= 1 && n_missing_dev);
    SCN_HIP(hipMemsetAsync(n_missing_dev, 0, sizeof(int64_t), S(stream)));
    if (n_fine == 0) return SCN_OK;
    SCN_REQUIRE(fine_coords && table_keys && table_rows && parent && cap > 0 && (cap & (cap - 1)) == 0);
    hipLaunchKernelGGL(k_parent_lookup_div, dim3(scn::ew_grid(n_fine, 256)), dim3(256), 0, S(stream),
                       (const int4*)fine_coords, (long long)n_fine, sx, sy, sz, (const unsigned long long*)table_keys, table_rows,
                       (long long)cap, parent, (unsigned long long*)n_missing_dev);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

extern "C" int scn_child_table(const int32_t* fine_coords, const int32_t* parent, int64_t n_fine, int64_t n_coarse,
                               int32_t* child, int32_t* fine_off, scn_stream_t stream) {
    SCN_REQUIRE(n_fine >= 0 && n_coarse >= 0 && n_coarse <= n_fine);
    if (n_fine == 0) return SCN_OK;
    SCN_REQUIRE(fine_coords && parent && child && fine_off);
    SCN_HIP(hipMemsetAsync(child, 0xFF, sizeof(int32_t) * 8 * n_coarse, S(stream)));
    hipLaunchKernelGGL(k_child_table, dim3(scn::ew_grid(n_fine, 256)), dim3(256), 0, S(stream),
                       (const int4*)fine_coords, parent, (long long)n_fine, (long long)n_coarse, child, fine_off);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

// ------------------------------------------------------------------------------------------------
// sparse ROI crop
// ------------------------------------------------------------------------------------------------
__global__ void k_roi_boxes(const float* __restrict__ boxes, const int* __restrict__ sample, int bb,
                            const int* __restrict__ size3, const float* __restrict__ resize3, int* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= bb) return;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float lo = boxes[i * 6 + d], hi = boxes[i * 6 + 3 + d];
        if (resize3) { lo = lo / resize3[d]; hi = hi / resize3[d]; }       // Divider: correctly rounded fp32 division
        int a = (int)floorf(lo);
        int b = (int)ceilf(hi);
        if (size3) {
            int s = size3[d];
            a = min(max(a, 0), s - 1);
            b = min(max(b, 1), s);
        }
        out[i * 8 + d] = a;
        out[i * 8 + 4 + d] = b;
    }
    out[i * 8 + 3] = sample[i];
    out[i * 8 + 7] = sample[i] + 1;
}

extern "C" int scn_roi_boxes(const float* boxes, const int32_t* box_sample, int bb,
                             const int32_t* spatial_size3_or_null, const float* resize3_or_null, int32_t* out,
                             scn_stream_t stream) {
    SCN_REQUIRE(bb >= 0);
    if (bb == 0) return SCN_OK;
    SCN_REQUIRE(boxes && box_sample && out);
    hipLaunchKernelGGL(k_roi_boxes, dim3((bb + 63) / 64), dim3(64), 0, S(stream), boxes, box_sample, bb,
                       spatial_size3_or_null, resize3_or_null, out);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

__global__ void k_roi_coords(const int4* __restrict__ coords, const int* __restrict__ src, const int* __restrict__ box,
                             long long m, long long* __restrict__ out) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < m;
         i += (long long)gridDim.x * blockDim.x) {
        int4 c = coords[src[i]];
        out[i * 4 + 0] = c.x;
        out[i * 4 + 1] = c.y;
        out[i * 4 + 2] = c.z;
        out[i * 4 + 3] = box[i];
    }
}

extern "C" int scn_roi_coords(const int32_t* coords, const int32_t* src_row, const int32_t* box_of, int64_t m,
                              int64_t* out_coords, scn_stream_t stream) {
    SCN_REQUIRE(m >= 0);
    if (m == 0) return SCN_OK;
    SCN_REQUIRE(coords && src_row && box_of && out_coords);
    hipLaunchKernelGGL(k_roi_coords, dim3(scn::ew_grid(m, 256)), dim3(256), 0, S(stream), (const int4*)coords, src_row,
                       box_of, (long long)m, (long long*)out_coords);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}
