// Stable LSD radix sort of (uint32 key, int32 value) pairs for the index build (scn_tiles.hip: rows by offset mask, tiles
// by offset count).  Hand-written for the sizes of this path -- 3 k ... 600 k pairs, 6 ... 27 significant key bits --
// where a library sort is a chain of 5-8 short launches (rocPRIM picks a block sort + merge passes below ~1 M items:
// 0.57 ms per step of the cfg-2 build, profiles/r2_kernel_stats.csv).
//
//   passes      = ceil(bits / 9), digit width = ceil(bits / passes)  (<= 512 bins: 27-bit masks take 3 passes, the 8-bit
//                 masks of a 2^3 child table and the 6-bit tile costs one)
//   n <= 4096   : ONE launch, one workgroup: every pass runs out of LDS (k_rs_block)
//   otherwise   : per pass  k_rs_hist   workgroup b counts the digits of its 1024 items -> counts[digit][b]
//                           k_rs_scan   workgroup d scans counts[d][*] in place, totals[d] = the digit's item count
//                           k_rs_scatter workgroup b ranks its items (stable) and writes them to their final positions
//
// Stable ranking inside a workgroup: a wave owns a CONTIGUOUS run of items (64 per load, 4 loads), finds the lanes that
// hold the same digit with one ballot per digit bit, and keeps a wave-private running count per digit in LDS; the
// lowest lane of a digit group advances the count.  Item order == (workgroup, wave, load, lane), so equal keys keep their
// input order -- the result is THE stable sort, bit-identical to any other stable sort (tests compare with numpy's).
#include "scn_common.h"
#include "scn_sort.h"

using scn::S;
using scn::cdiv;

namespace {

constexpr int RS_T = 256;             // threads per workgroup (multi-workgroup form)
constexpr int RS_IT = 4;              // items per thread
constexpr int RS_TILE = RS_T * RS_IT;  // items per workgroup
constexpr int RS_MAXB = 512;          // bins
constexpr int RS_SMALL = 4096;        // one-workgroup form up to here (1024 threads x 4 items)

struct PassPlan { int n_pass, width[4], shift[4]; };

PassPlan plan_of(int bits) {
    PassPlan p{};
    if (bits < 1) bits = 1;
    p.n_pass = (bits + 8) / 9;
    int w = (bits + p.n_pass - 1) / p.n_pass, s = 0;
    for (int i = 0; i < p.n_pass; ++i) {
        p.shift[i] = s;
        p.width[i] = (s + w <= bits) ? w : bits - s;
        s += p.width[i];
    }
    return p;
}

// lanes of the wave holding the same digit as this lane (all 64 lanes take part; `valid` = this lane holds an item)
__device__ __forceinline__ unsigned long long peers_of(unsigned d, int width, bool valid) {
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 9; ++b) {
        if (b < width) {
            const unsigned long long m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
    }
    return peers;
}

__global__ __launch_bounds__(RS_T) void k_rs_hist(const unsigned* __restrict__ keys, long long n, int shift, int width,
                                                  int* __restrict__ counts, int nblk) {
    __shared__ int hist[RS_MAXB];
    const int bins = 1 << width;
    for (int d = threadIdx.x; d < bins; d += RS_T) hist[d] = 0;
    __syncthreads();
    const long long base = (long long)blockIdx.x * RS_TILE;
    const unsigned dm = (unsigned)bins - 1u;
#pragma unroll
    for (int it = 0; it < RS_IT; ++it) {
        const long long e = base + it * RS_T + threadIdx.x;
        if (e < n) atomicAdd(&hist[(keys[e] >> shift) & dm], 1);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < bins; d += RS_T) counts[(long long)d * nblk + blockIdx.x] = hist[d];
}

// workgroup d: exclusive scan of counts[d][0..nblk) in place; totals[d] = sum
__global__ __launch_bounds__(RS_T) void k_rs_scan(int* __restrict__ counts, int nblk, int* __restrict__ totals) {
    __shared__ int wtot[RS_T / 64];
    __shared__ int carry_s;
    int* c = counts + (long long)blockIdx.x * nblk;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < nblk; base += RS_T) {
        const int i = base + threadIdx.x;
        const int v = i < nblk ? c[i] : 0;
        int x = v;
#pragma unroll
        for (int dlt = 1; dlt < 64; dlt <<= 1) {
            const int y = __shfl_up(x, dlt);
            if (lane >= dlt) x += y;
        }
        if (lane == 63) wtot[w] = x;
        __syncthreads();
        int woff = 0;
#pragma unroll
        for (int k = 0; k < RS_T / 64; ++k) woff += k < w ? wtot[k] : 0;
        const int carry = carry_s;
        if (i < nblk) c[i] = carry + woff + x - v;
        __syncthreads();
        if (threadIdx.x == RS_T - 1) carry_s = carry + woff + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry_s;
}

// exclusive scan of totals[0..bins) into LDS base[] (bins <= 512, any workgroup size that is a multiple of 64)
__device__ __forceinline__ void scan_totals(const int* __restrict__ totals, int bins, int* base, int* wtmp) {
    const int T = blockDim.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = T >> 6;
    int carry = 0;
    for (int b0 = 0; b0 < bins; b0 += T) {
        const int d = b0 + threadIdx.x;
        const int v = d < bins ? totals[d] : 0;
        int x = v;
#pragma unroll
        for (int dlt = 1; dlt < 64; dlt <<= 1) {
            const int y = __shfl_up(x, dlt);
            if (lane >= dlt) x += y;
        }
        if (lane == 63) wtmp[w] = x;
        __syncthreads();
        int woff = 0, tot = 0;
        for (int k = 0; k < nw; ++k) { woff += k < w ? wtmp[k] : 0; tot += wtmp[k]; }
        if (d < bins) base[d] = carry + woff + x - v;
        carry += tot;
        __syncthreads();
    }
}

__global__ __launch_bounds__(RS_T) void k_rs_scatter(const unsigned* __restrict__ keys, const int* __restrict__ vals,
                                                     long long n, int shift, int width, const int* __restrict__ counts,
                                                     int nblk, const int* __restrict__ totals,
                                                     unsigned* __restrict__ keys_out, int* __restrict__ vals_out) {
    __shared__ int whist[RS_T / 64][RS_MAXB];       // running count per (wave, digit); then the wave's offset
    __shared__ int gbase[RS_MAXB];                  // first output position of (digit, this workgroup)
    __shared__ int wtmp[RS_T / 64];
    const int bins = 1 << width, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned dm = (unsigned)bins - 1u;
    for (int d = threadIdx.x; d < bins; d += RS_T) {
#pragma unroll
        for (int k = 0; k < RS_T / 64; ++k) whist[k][d] = 0;
    }
    scan_totals(totals, bins, gbase, wtmp);         // (ends with a barrier)
    for (int d = threadIdx.x; d < bins; d += RS_T) gbase[d] += counts[(long long)d * nblk + blockIdx.x];
    // a wave owns items [base + w*256, +256): load it*64 + lane
    const long long base = (long long)blockIdx.x * RS_TILE + (long long)w * (64 * RS_IT);
    unsigned key[RS_IT], dig[RS_IT];
    int val[RS_IT], rank[RS_IT];
    bool ok[RS_IT];
#pragma unroll
    for (int it = 0; it < RS_IT; ++it) {
        const long long e = base + it * 64 + lane;
        ok[it] = e < n;
        key[it] = ok[it] ? keys[e] : 0u;
        val[it] = ok[it] ? (vals ? vals[e] : (int)e) : 0;
        dig[it] = (key[it] >> shift) & dm;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < RS_IT; ++it) {
        const unsigned long long peers = peers_of(dig[it], width, ok[it]);
        volatile int* cnt = &whist[w][dig[it]];                     // every lane of the group reads the same count,
        const int before = *cnt;                                    // then its lowest lane advances it (LDS operations of
        rank[it] = before + __popcll(peers & ((1ull << lane) - 1ull));   // a wave complete in program order)
        if (ok[it] && (peers & ((1ull << lane) - 1ull)) == 0) *cnt = before + __popcll(peers);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    for (int d = threadIdx.x; d < bins; d += RS_T) {               // counts per wave -> exclusive offsets per wave
        int run = 0;
#pragma unroll
        for (int k = 0; k < RS_T / 64; ++k) { const int c = whist[k][d]; whist[k][d] = run; run += c; }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < RS_IT; ++it) {
        if (ok[it]) {
            const int pos = gbase[dig[it]] + whist[w][dig[it]] + rank[it];
            keys_out[pos] = key[it];
            vals_out[pos] = val[it];
        }
    }
}

// n <= 4096: one workgroup of 1024 threads, all passes in LDS
__global__ __launch_bounds__(1024) void k_rs_block(const unsigned* __restrict__ keys, const int* __restrict__ vals, int n,
                                                   PassPlan plan, unsigned* __restrict__ keys_out,
                                                   int* __restrict__ vals_out) {
    __shared__ unsigned short whist[16][RS_MAXB];   // 16 KB (counts <= 4096)
    __shared__ int gbase[RS_MAXB];
    __shared__ int wtmp[16];
    __shared__ unsigned kbuf[RS_SMALL];
    __shared__ int vbuf[RS_SMALL];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int base = w * (64 * RS_IT);
    unsigned key[RS_IT];
    int val[RS_IT];
    bool ok[RS_IT];
#pragma unroll
    for (int it = 0; it < RS_IT; ++it) {
        const int e = base + it * 64 + lane;
        ok[it] = e < n;
        key[it] = ok[it] ? keys[e] : 0u;
        val[it] = ok[it] ? (vals ? vals[e] : e) : 0;
    }
    for (int p = 0; p < plan.n_pass; ++p) {
        const int width = plan.width[p], shift = plan.shift[p], bins = 1 << width;
        const unsigned dm = (unsigned)bins - 1u;
        for (int d = threadIdx.x; d < bins; d += 1024) {
#pragma unroll
            for (int k = 0; k < 16; ++k) whist[k][d] = 0;
        }
        __syncthreads();
        unsigned dig[RS_IT];
        int rank[RS_IT];
#pragma unroll
        for (int it = 0; it < RS_IT; ++it) {
            dig[it] = (key[it] >> shift) & dm;
            const unsigned long long peers = peers_of(dig[it], width, ok[it]);
            volatile unsigned short* cnt = &whist[w][dig[it]];
            const int before = *cnt;
            rank[it] = before + __popcll(peers & ((1ull << lane) - 1ull));
            if (ok[it] && (peers & ((1ull << lane) - 1ull)) == 0) *cnt = (unsigned short)(before + __popcll(peers));
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
        for (int d = threadIdx.x; d < bins; d += 1024) {           // per-wave offsets; gbase[d] = the digit's total
            int run = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) { const int c = whist[k][d]; whist[k][d] = (unsigned short)run; run += c; }
            gbase[d] = run;
        }
        __syncthreads();
        // exclusive scan of the digit totals (bins <= 512 <= 1024 threads: one round)
        {
            const int d = threadIdx.x;
            const int v = d < bins ? gbase[d] : 0;
            int x = v;
#pragma unroll
            for (int dlt = 1; dlt < 64; dlt <<= 1) {
                const int y = __shfl_up(x, dlt);
                if (lane >= dlt) x += y;
            }
            if (lane == 63) wtmp[w] = x;
            __syncthreads();
            int woff = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) woff += k < w ? wtmp[k] : 0;
            if (d < bins) gbase[d] = woff + x - v;
            __syncthreads();
        }
#pragma unroll
        for (int it = 0; it < RS_IT; ++it) {
            if (ok[it]) {
                const int pos = gbase[dig[it]] + whist[w][dig[it]] + rank[it];
                kbuf[pos] = key[it];
                vbuf[pos] = val[it];
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < RS_IT; ++it) {
            const int e = base + it * 64 + lane;
            if (ok[it]) { key[it] = kbuf[e]; val[it] = vbuf[e]; }
        }
        __syncthreads();
    }
#pragma unroll
    for (int it = 0; it < RS_IT; ++it) {
        const int e = base + it * 64 + lane;
        if (ok[it]) { keys_out[e] = key[it]; vals_out[e] = val[it]; }
    }
}

inline int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }

}  // namespace

namespace scn {

int64_t sort_pairs_scratch_bytes(int64_t n) {
    if (n <= RS_SMALL) return 256;
    const int64_t nblk = cdiv(n, RS_TILE);
    return 2 * align256(4 * n) + align256(4 * RS_MAXB * nblk) + align256(4 * RS_MAXB) + 256;
}

int sort_pairs(const uint32_t* keys, const int32_t* vals, int64_t n, int bits, uint32_t* keys_out, int32_t* vals_out,
               void* scratch, hipStream_t st) {
    SCN_REQUIRE(n >= 0 && n < 2147483647LL && bits >= 1 && bits <= 32);
    if (n == 0) return SCN_OK;
    SCN_REQUIRE(keys && keys_out && vals_out && scratch);
    SCN_REQUIRE(keys != keys_out && vals != vals_out);
    PassPlan plan = plan_of(bits);
    if (n <= RS_SMALL) {
        hipLaunchKernelGGL(k_rs_block, dim3(1), dim3(1024), 0, st, keys, vals, (int)n, plan, keys_out, vals_out);
        SCN_LAUNCH_CHECK();
        return SCN_OK;
    }
    const int nblk = (int)cdiv(n, RS_TILE);
    char* p = (char*)scratch;
    unsigned* ktmp = (unsigned*)p;  p += align256(4 * n);
    int* vtmp = (int*)p;            p += align256(4 * n);
    int* counts = (int*)p;          p += align256(4 * (int64_t)RS_MAXB * nblk);
    int* totals = (int*)p;
    const unsigned* ksrc = keys;
    const int* vsrc = vals;
    for (int i = 0; i < plan.n_pass; ++i) {
        // the last pass must land in the caller's output: destinations alternate backwards from there
        const bool to_out = ((plan.n_pass - 1 - i) & 1) == 0;
        unsigned* kdst = to_out ? keys_out : ktmp;
        int* vdst = to_out ? vals_out : vtmp;
        const int bins = 1 << plan.width[i];
        hipLaunchKernelGGL(k_rs_hist, dim3(nblk), dim3(RS_T), 0, st, ksrc, (long long)n, plan.shift[i], plan.width[i],
                           counts, nblk);
        SCN_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_rs_scan, dim3(bins), dim3(RS_T), 0, st, counts, nblk, totals);
        SCN_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_rs_scatter, dim3(nblk), dim3(RS_T), 0, st, ksrc, vsrc, (long long)n, plan.shift[i],
                           plan.width[i], (const int*)counts, nblk, (const int*)totals, kdst, vdst);
        SCN_LAUNCH_CHECK();
        ksrc = kdst;
        vsrc = vdst;
    }
    return SCN_OK;
}

}  // namespace scn

extern "C" int64_t scn_sort_pairs_scratch_bytes(int64_t n) { return n < 0 ? -1 : scn::sort_pairs_scratch_bytes(n); }

extern "C" int scn_sort_pairs(const uint32_t* keys, const int32_t* vals, int64_t n, int bits, uint32_t* keys_out,
                              int32_t* vals_out, void* scratch, scn_stream_t stream) {
    return scn::sort_pairs(keys, vals, n, bits, keys_out, vals_out, scratch, S(stream));
}
