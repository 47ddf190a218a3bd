// Sparse ROI crop without any [boxes x points] object (hard part H5).
//
// The reference materialises a BB x N bool matrix and boolean-mask-gathers an expanded BB x N x C view
// (ndsis/modules/roi_select_sparse.py:125-180).  Its RESULT is a list: for box 0, 1, ... the points inside it in ascending
// point row -- (box-major, ascending point row).  Here that list is produced by count -> scan -> scatter:
//
//   k_roi_count : a wave owns a UNIT of 256 consecutive points (4 per lane, kept in registers).  It reduces the unit's
//                 bounding box and sample range once, then walks the boxes: a box whose sample interval or extent misses
//                 the unit's bounds is skipped by a wave-uniform branch (points arrive sample-major and mesh-ordered, so
//                 a unit is spatially compact and a box only meets its candidate units); for the others the four
//                 inside tests are one ballot + popcount each.  cnt[box][unit] (int32, box-major) is the only
//                 intermediate: BB x N / 256 counters, 43 k integers at 64 boxes x 172 k points.
//   k_roi_scan_box / k_roi_prefix: exclusive scan of every box's counters (one workgroup per box) and of the box totals:
//                 prefix[box] = first output row of the box (the CSR the mask-head epilogue consumes), prefix[box] +
//                 cnt[box][unit] = start of the (box, unit) run in the output.
//   k_roi_fill  : the same walk; lanes write src_row / box_of and the int64 (x, y, z, box) coordinate rows of
//                 select_coords at  start(box, unit) + rank inside the unit  (ballot prefix popcount).
//
// Bytes: coords are read twice (2 x 16 N), M rows of (4 + 4 + 32) bytes are written; no term in BB x N.
// The dense bool matrix the reference's roi_cut also returns is rebuilt from the list only when a caller asks for it
// (scn_roi_inside).
#include "scn_common.h"

using scn::S;
using scn::cdiv;

static constexpr int ROI_UNIT = 256;            // points per wave
static constexpr int ROI_WAVES = 4;             // waves per workgroup
static constexpr int ROI_BOX_TILE = 1024;       // boxes staged in LDS at a time (32 KB)

struct RoiUnit {
    int4 p[4];
    bool valid[4];                              // point index < n
    int lo[4], hi[4];                           // bounds over the unit's valid points: x, y, z, sample (inclusive)
};

__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = min(v, __shfl_xor(v, d));
    return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, __shfl_xor(v, d));
    return v;
}

__device__ __forceinline__ void roi_load_unit(const int4* __restrict__ coords, long long n, long long base, int lane,
                                              RoiUnit& u) {
    int lo[4] = {0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF}, hi[4] = {-0x7FFFFFFF, -0x7FFFFFFF, -0x7FFFFFFF, -0x7FFFFFFF};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const long long j = base + q * 64 + lane;
        u.valid[q] = j < n;
        u.p[q] = j < n ? coords[j] : make_int4(0, 0, 0, 0);
        if (j < n) {
            lo[0] = min(lo[0], u.p[q].x); hi[0] = max(hi[0], u.p[q].x);
            lo[1] = min(lo[1], u.p[q].y); hi[1] = max(hi[1], u.p[q].y);
            lo[2] = min(lo[2], u.p[q].z); hi[2] = max(hi[2], u.p[q].z);
            lo[3] = min(lo[3], u.p[q].w); hi[3] = max(hi[3], u.p[q].w);
        }
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        u.lo[d] = __builtin_amdgcn_readfirstlane(wave_min(lo[d]));
        u.hi[d] = __builtin_amdgcn_readfirstlane(wave_max(hi[d]));
    }
}

// box = (start x, y, z, sample ; stop x, y, z, sample + 1), half-open as in get_inside_indicator (:157-167)
__device__ __forceinline__ bool roi_box_meets_unit(const int* b, const RoiUnit& u) {
    return b[0] <= u.hi[0] && b[4] > u.lo[0] && b[1] <= u.hi[1] && b[5] > u.lo[1] && b[2] <= u.hi[2] && b[6] > u.lo[2] &&
           b[3] <= u.hi[3] && b[7] > u.lo[3];
}

__device__ __forceinline__ bool roi_inside(const int4& c, const int* b) {
    return c.x >= b[0] && c.x < b[4] && c.y >= b[1] && c.y < b[5] && c.z >= b[2] && c.z < b[6] && c.w >= b[3] && c.w < b[7];
}

// FILL = false: cnt[box][unit] = points of the unit inside the box.
// FILL = true : cnt holds the exclusive scan; write the selection rows.
template <bool FILL>
__global__ __launch_bounds__(ROI_WAVES * 64) void k_roi_walk(const int4* __restrict__ coords, long long n,
                                                             const int* __restrict__ boxes, int bb, long long n_units,
                                                             int* __restrict__ cnt, int* __restrict__ src_row,
                                                             int* __restrict__ box_of, long long* __restrict__ out_coords,
                                                             const long long* __restrict__ prefix) {
    __shared__ int sbox[ROI_BOX_TILE * 8];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long unit = (long long)blockIdx.x * ROI_WAVES + w;
    const bool live = unit < n_units;                       // wave-uniform
    RoiUnit u;
    if (live) roi_load_unit(coords, n, unit * ROI_UNIT, lane, u);
    for (int b0 = 0; b0 < bb; b0 += ROI_BOX_TILE) {
        const int nb = min(ROI_BOX_TILE, bb - b0);
        __syncthreads();                                    // previous tile fully consumed
        for (int e = threadIdx.x; e < nb * 8; e += ROI_WAVES * 64) sbox[e] = boxes[(long long)b0 * 8 + e];
        __syncthreads();
        if (!live) continue;
        for (int k = 0; k < nb; ++k) {
            const int* b = sbox + k * 8;                    // wave-uniform address: LDS broadcast reads
            const long long slot = (long long)(b0 + k) * n_units + unit;
            if (!roi_box_meets_unit(b, u)) {                // wave-uniform skip
                if (!FILL && lane == 0) cnt[slot] = 0;
                continue;
            }
            if (!FILL) {
                int c = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) c += __popcll(__ballot(u.valid[q] && roi_inside(u.p[q], b)));
                if (lane == 0) cnt[slot] = c;
            } else {
                long long run = prefix[b0 + k] + cnt[slot]; // first output row of this (box, unit) run
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bool in = u.valid[q] && roi_inside(u.p[q], b);
                    const unsigned long long m = __ballot(in);
                    if (in) {
                        const long long r = run + __popcll(m & ((1ull << lane) - 1ull));
                        src_row[r] = (int)(unit * ROI_UNIT + q * 64 + lane);
                        box_of[r] = b0 + k;
                        if (out_coords) {
                            long long* oc = out_coords + r * 4;
                            oc[0] = u.p[q].x; oc[1] = u.p[q].y; oc[2] = u.p[q].z; oc[3] = b0 + k;
                        }
                    }
                    run += __popcll(m);
                }
            }
        }
    }
}

// Exclusive scan of cnt[box][0 .. n_units) in place, one workgroup per box, and the box totals.  (The scan shared with the
// rulebooks -- one workgroup over all boxes x units -- took 60 of the crop's 120 us at 64 boxes x 674 units.)
__global__ __launch_bounds__(256) void k_roi_scan_box(int* __restrict__ cnt, long long n_units, int* __restrict__ total) {
    __shared__ int wsum[4];
    __shared__ int carry_s;
    int* row = cnt + (long long)blockIdx.x * n_units;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (long long base = 0; base < n_units; base += 256) {
        const long long i = base + threadIdx.x;
        const int v = i < n_units ? row[i] : 0;
        int x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int y = __shfl_up(x, d);
            if (lane >= d) x += y;
        }
        if (lane == 63) wsum[w] = x;
        __syncthreads();
        int off = carry_s;
        for (int k = 0; k < w; ++k) off += wsum[k];
        if (i < n_units) row[i] = off + x - v;
        __syncthreads();
        if (threadIdx.x == 255) carry_s = off + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) total[blockIdx.x] = carry_s;
}

// prefix[b] = rows selected by the boxes before b (one workgroup; bb < 65536), prefix[bb] = M
__global__ __launch_bounds__(1024) void k_roi_prefix(const int* __restrict__ total, int bb, long long* __restrict__ prefix) {
    __shared__ long long wtot[16];
    __shared__ long long carry_s;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < bb; base += 1024) {
        const int i = base + threadIdx.x;
        const long long v = i < bb ? total[i] : 0;
        long long x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const long long y = __shfl_up(x, d);
            if (lane >= d) x += y;
        }
        if (lane == 63) wtot[w] = x;
        __syncthreads();
        long long off = carry_s;
        for (int k = 0; k < w; ++k) off += wtot[k];
        if (i < bb) prefix[i] = off + x - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = off + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) prefix[bb] = carry_s;
}

extern "C" int64_t scn_roi_units(int64_t n) { return n > 0 ? cdiv(n, ROI_UNIT) : 1; }

extern "C" int scn_roi_count(const int32_t* coords, int64_t n, const int32_t* boxes, int bb, int32_t* unit_offsets,
                             int64_t* prefix, int64_t* prefix_host, scn_stream_t stream) {
    SCN_REQUIRE(n >= 0 && bb >= 0 && bb < 65536 && prefix);
    hipStream_t st = S(stream);
    if (n == 0 || bb == 0) {
        SCN_HIP(hipMemsetAsync(prefix, 0, sizeof(int64_t) * (bb + 1), st));
    } else {
        SCN_REQUIRE(coords && boxes && unit_offsets);
        SCN_REQUIRE(n < 2147483647LL);
        const int64_t n_units = scn_roi_units(n);
        SCN_REQUIRE(n_units * bb < 2147483647LL);
        hipLaunchKernelGGL(k_roi_walk<false>, dim3((unsigned)cdiv(n_units, ROI_WAVES)), dim3(ROI_WAVES * 64), 0, st,
                           (const int4*)coords, (long long)n, boxes, bb, (long long)n_units, unit_offsets, (int*)nullptr,
                           (int*)nullptr, (long long*)nullptr, (const long long*)nullptr);
        SCN_LAUNCH_CHECK();
        int* totals = unit_offsets + n_units * bb;           // bb box totals behind the counters
        hipLaunchKernelGGL(k_roi_scan_box, dim3((unsigned)bb), dim3(256), 0, st, unit_offsets, (long long)n_units, totals);
        SCN_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_roi_prefix, dim3(1), dim3(1024), 0, st, (const int*)totals, bb, (long long*)prefix);
        SCN_LAUNCH_CHECK();
    }
    if (!prefix_host) return SCN_OK;
    SCN_HIP(hipMemcpyAsync(prefix_host, prefix, sizeof(int64_t) * (bb + 1), hipMemcpyDeviceToHost, st));
    SCN_HIP(hipStreamSynchronize(st));
    SCN_REQUIRE(prefix_host[bb] < 2147483647LL);            // the selection is indexed with int32 rows
    return SCN_OK;
}

extern "C" int scn_roi_fill(const int32_t* coords, int64_t n, const int32_t* boxes, int bb, const int32_t* unit_offsets,
                            const int64_t* prefix, int32_t* src_row, int32_t* box_of, int64_t* out_coords,
                            scn_stream_t stream) {
    SCN_REQUIRE(n >= 0 && bb >= 0 && bb < 65536);
    if (n == 0 || bb == 0) return SCN_OK;
    SCN_REQUIRE(coords && boxes && unit_offsets && prefix && src_row && box_of);
    const int64_t n_units = scn_roi_units(n);
    hipLaunchKernelGGL(k_roi_walk<true>, dim3((unsigned)cdiv(n_units, ROI_WAVES)), dim3(ROI_WAVES * 64), 0, S(stream),
                       (const int4*)coords, (long long)n, boxes, bb, (long long)n_units, (int*)unit_offsets, src_row,
                       box_of, (long long*)out_coords, (const long long*)prefix);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

__global__ void k_roi_inside(const int* __restrict__ src_row, const int* __restrict__ box_of, long long m, long long n,
                             unsigned char* __restrict__ inside) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < m; i += (long long)gridDim.x * blockDim.x)
        inside[(long long)box_of[i] * n + src_row[i]] = 1;
}

extern "C" int scn_roi_inside(const int32_t* src_row, const int32_t* box_of, int64_t m, int64_t n, int bb,
                              uint8_t* inside_u8, scn_stream_t stream) {
    SCN_REQUIRE(m >= 0 && n >= 0 && bb >= 0);
    if (n == 0 || bb == 0) return SCN_OK;
    SCN_REQUIRE(inside_u8);
    SCN_HIP(hipMemsetAsync(inside_u8, 0, (size_t)bb * (size_t)n, S(stream)));
    if (m == 0) return SCN_OK;
    SCN_REQUIRE(src_row && box_of);
    hipLaunchKernelGGL(k_roi_inside, dim3(scn::ew_grid(m, 256)), dim3(256), 0, S(stream), src_row, box_of, (long long)m,
                       (long long)n, inside_u8);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}
