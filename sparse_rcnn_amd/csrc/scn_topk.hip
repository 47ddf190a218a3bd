// Exact top-k of a score field with the boxes gathered along (scn_topk_boxes): what ProposalSelector runs before its NMS
// (ndsis/modules/proposal_selector.py:60-75: torch.topk(rpn_score, num_keep_pre_nms) then rpn_bbox[batch, indices]).
// torch.topk on a [1, 524 288] field is a segmented merge sort: 19 launches, 165 us of the cfg 3-rpn step.  Here: a radix
// SELECT -- two 11-bit histogram passes over an order-preserving 32-bit key find the 22-bit bucket the k-th score lies in,
// one pass compacts everything above the bucket ("sure", < k elements) and the bucket itself ("ties"), and one workgroup per
// scene sorts those candidates (bitonic in LDS, key descending, index ascending) and writes scores, indices and boxes.
// Four launches; every launch boundary is the coherence point (no tickets, no fences): the workgroups of a launch find the
// thresholds of the launches before them from the global histograms themselves (2048 bins, one block scan).
// Order: descending score; equal scores by ascending index (torch.topk leaves the order of ties open); NaN counts as the
// largest value, as in torch.  A bucket with more than TK_TIE_CAP members (a score field that is mostly ONE constant, and
// the k-th score is that constant) takes a slow path inside the last kernel: that workgroup alone selects the remaining
// elements by four more radix passes over (low key bits, index).  Same result, ~0.2 ms; not seen in the detection step.
// Scratch: per scene a TkState (zero before the first call -- every call leaves it zero again) and the candidate list.
#include "scn_common.h"

using scn::S;

namespace {
constexpr int TK_BINS = 2048;
constexpr int TK_MAX_K = 2048;
constexpr int TK_TIE_CAP = 6144;                   // TK_MAX_K + TK_TIE_CAP = 8192 candidates = 64 KB of LDS in the sort
constexpr int TK_CAND = TK_MAX_K + TK_TIE_CAP;

struct TkState {
    unsigned hist1[TK_BINS];     // key >> 21
    unsigned hist2[TK_BINS];     // (key >> 10) & 2047 of the elements whose key >> 21 is the threshold bin of pass 1
    unsigned n_sure, n_tie;      // cursors of the compaction
    unsigned cut[4];             // (t1 << 11 | t2, above2, pop2, -) as the compaction's workgroup 0 found them
};

__device__ __forceinline__ unsigned tk_key(float f) {          // ascending unsigned order == ascending float order, NaN on top
    unsigned u = __float_as_uint(f);
    if (f != f) return 0xFFFFFFFFu;
    if (f == 0.f) u = 0u;                                      // -0 == +0: a tie, ordered by index like every other
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// The bin in which the suffix count (from the top bin down) reaches `need`: t, the count ABOVE it, its own population.
// h: the histogram in LDS (TK_BINS or fewer bins, `bins` a multiple of blockDim.x or smaller); part: blockDim.x words of LDS;
// res: 3 words of LDS.  Every thread returns the same values.  Needs sum(h) >= need >= 1.
__device__ __forceinline__ void tk_threshold(const unsigned* h, int bins, unsigned need, unsigned* part, unsigned* res,
                                             unsigned& t, unsigned& above, unsigned& pop) {
    const int nt = blockDim.x, tid = threadIdx.x;
    const int per = (bins + nt - 1) / nt;
    const int lo = tid * per, hi = min(bins, lo + per);
    unsigned mine = 0;
    for (int b = lo; b < hi; ++b) mine += h[b];
    part[tid] = mine;
    __syncthreads();
    for (int d = 1; d < nt; d <<= 1) {                          // inclusive suffix sums over the threads
        const unsigned add = (tid + d < nt) ? part[tid + d] : 0u;
        __syncthreads();
        part[tid] += add;
        __syncthreads();
    }
    unsigned acc = part[tid] - mine;                           // everything in the bins above this thread's range
    if (acc < need && acc + mine >= need) {
        for (int b = hi - 1; b >= lo; --b) {
            const unsigned c = h[b];
            if (acc + c >= need) { res[0] = (unsigned)b; res[1] = acc; res[2] = c; break; }
            acc += c;
        }
    }
    __syncthreads();
    t = res[0]; above = res[1]; pop = res[2];
    __syncthreads();
}

struct TkChunk { long long lo, hi; };
__device__ __forceinline__ TkChunk tk_chunk(long long n) {      // the contiguous share of this workgroup (x) of a scene (y)
    const long long per = (n + gridDim.x - 1) / gridDim.x;
    const long long lo = (long long)blockIdx.x * per;
    return TkChunk{lo, lo + per < n ? lo + per : n};
}

// thresholds of pass 1 (and pass 2) from the global histograms, by every workgroup that needs them
struct TkCut { unsigned t1, above1, pop1, t2, above2, pop2; };
__device__ __forceinline__ void tk_cut1(const TkState* st, int k, unsigned* h, unsigned* part, unsigned* res, TkCut& c) {
    for (int b = threadIdx.x; b < TK_BINS; b += blockDim.x) h[b] = st->hist1[b];
    __syncthreads();
    tk_threshold(h, TK_BINS, (unsigned)k, part, res, c.t1, c.above1, c.pop1);
}
__device__ __forceinline__ void tk_cut2(const TkState* st, int k, unsigned* h, unsigned* part, unsigned* res, TkCut& c) {
    for (int b = threadIdx.x; b < TK_BINS; b += blockDim.x) h[b] = st->hist2[b];
    __syncthreads();
    unsigned a2;
    tk_threshold(h, TK_BINS, (unsigned)k - c.above1, part, res, c.t2, a2, c.pop2);
    c.above2 = c.above1 + a2;
}

// A workgroup walks its share in tiles of 2048 elements, eight per thread, all eight loads in flight before the first use
// (a loop around one load + one LDS atomic runs one memory round trip per element: 7.4 us for 2 MB, measured).
constexpr int TK_UNROLL = 8;
template <class F>
__device__ __forceinline__ void tk_for_each_key(const float* __restrict__ x, TkChunk ch, F&& f) {
    for (long long base = ch.lo; base < ch.hi; base += 256 * TK_UNROLL) {
        float vals[TK_UNROLL];
#pragma unroll
        for (int j = 0; j < TK_UNROLL; ++j) {
            const long long i = base + j * 256 + threadIdx.x;
            vals[j] = i < ch.hi ? x[i] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < TK_UNROLL; ++j) {
            const long long i = base + j * 256 + threadIdx.x;
            if (i < ch.hi) f(tk_key(vals[j]), i);
        }
    }
}

template <int PASS>
__global__ __launch_bounds__(256) void k_topk_hist(const float* __restrict__ scores, long long n, int k, TkState* __restrict__ st) {
    __shared__ unsigned h[TK_BINS];
    __shared__ unsigned part[256];
    __shared__ unsigned res[3];
    TkState* s = st + blockIdx.y;
    const float* x = scores + (long long)blockIdx.y * n;
    unsigned t1 = 0;
    if (PASS == 2) {
        TkCut c;
        tk_cut1(s, k, h, part, res, c);
        t1 = c.t1;
    }
    for (int b = threadIdx.x; b < TK_BINS; b += 256) h[b] = 0;
    __syncthreads();
    const TkChunk ch = tk_chunk(n);
    tk_for_each_key(x, ch, [&](unsigned key, long long) {
        if (PASS == 1) atomicAdd(&h[key >> 21], 1u);
        else if ((key >> 21) == t1) atomicAdd(&h[(key >> 10) & 2047u], 1u);
    });
    __syncthreads();
    unsigned* g = PASS == 1 ? s->hist1 : s->hist2;
    for (int b = threadIdx.x; b < TK_BINS; b += 256)
        if (h[b]) atomicAdd(&g[b], h[b]);
}

// cand[scene][0 .. TK_MAX_K): sure (buckets above the threshold's), cand[scene][TK_MAX_K ..): ties (the threshold's bucket), as
// (key, index) pairs in arrival order.  A workgroup parks what it finds in LDS (its share holds a handful of candidates;
// TK_STAGE of them per list and tile round, flushed when full), one global atomic per flush and list hands out the slots.
constexpr int TK_STAGE = 2048;                                 // >= one tile round: a flush per round can never overflow
__global__ __launch_bounds__(256) void k_topk_compact(const float* __restrict__ scores, long long n, int k,
                                                      TkState* __restrict__ st, uint2* __restrict__ cand) {
    __shared__ unsigned h[TK_BINS];
    __shared__ unsigned part[256];
    __shared__ unsigned res[3];
    __shared__ unsigned cnt[2], base[2];
    __shared__ uint2 stage[2][TK_STAGE];
    TkState* s = st + blockIdx.y;
    const float* x = scores + (long long)blockIdx.y * n;
    uint2* out = cand + (size_t)blockIdx.y * TK_CAND;
    TkCut c;
    tk_cut1(s, k, h, part, res, c);
    tk_cut2(s, k, h, part, res, c);
    const unsigned P = (c.t1 << 11) | c.t2;
    const bool ties = c.pop2 <= (unsigned)TK_TIE_CAP;          // a larger bucket: the last kernel selects from the field itself
    if (blockIdx.x == 0 && threadIdx.x == 0) { s->cut[0] = P; s->cut[1] = c.above2; s->cut[2] = c.pop2; }
    const TkChunk ch = tk_chunk(n);
    for (long long lo = ch.lo; lo < ch.hi; lo += 256 * TK_UNROLL) {         // one tile round at a time (<= 2048 finds)
        if (threadIdx.x < 2) cnt[threadIdx.x] = 0;
        __syncthreads();
        const long long hi = lo + 256 * TK_UNROLL < ch.hi ? lo + 256 * TK_UNROLL : ch.hi;
        tk_for_each_key(x, TkChunk{lo, hi}, [&](unsigned key, long long i) {
            const unsigned top = key >> 10;
            if (top > P) stage[0][atomicAdd(&cnt[0], 1u)] = make_uint2(key, (unsigned)i);
            else if (top == P && ties) stage[1][atomicAdd(&cnt[1], 1u)] = make_uint2(key, (unsigned)i);
        });
        __syncthreads();
        if (threadIdx.x < 2 && cnt[threadIdx.x])
            base[threadIdx.x] = atomicAdd(threadIdx.x == 0 ? &s->n_sure : &s->n_tie, cnt[threadIdx.x]);
        __syncthreads();
        for (unsigned j = threadIdx.x; j < cnt[0]; j += 256) out[base[0] + j] = stage[0][j];
        for (unsigned j = threadIdx.x; j < cnt[1]; j += 256) out[TK_MAX_K + base[1] + j] = stage[1][j];
        __syncthreads();
    }
}

__device__ __forceinline__ unsigned long long tk_composite(unsigned key, unsigned idx) {       // descending = key desc, index asc
    return ((unsigned long long)key << 32) | (unsigned long long)(0xFFFFFFFFu - idx);
}

// One workgroup per scene: candidates -> LDS, bitonic sort, the first k out; the state goes back to zero.
__global__ __launch_bounds__(1024) void k_topk_final(const float* __restrict__ scores, const float* __restrict__ boxes,
                                                     long long n, int k, TkState* __restrict__ st,
                                                     const uint2* __restrict__ cand, float* __restrict__ out_scores,
                                                     long long* __restrict__ out_index, float* __restrict__ out_boxes) {
    extern __shared__ unsigned long long v[];                   // TK_CAND composites
    __shared__ unsigned h[TK_BINS];
    __shared__ unsigned part[1024];
    __shared__ unsigned res[3];
    __shared__ unsigned cursor;
    TkState* s = st + blockIdx.x;
    const float* x = scores + (long long)blockIdx.x * n;
    const uint2* in = cand + (size_t)blockIdx.x * TK_CAND;
    const int tid = threadIdx.x;
    TkCut c;                                                    // as the compaction found it (a launch boundary ago)
    c.t1 = s->cut[0] >> 11; c.t2 = s->cut[0] & 2047u; c.above2 = s->cut[1]; c.pop2 = s->cut[2];
    const unsigned n_sure = c.above2;                           // (== s->n_sure)
    unsigned total;
    for (unsigned i = tid; i < n_sure; i += 1024) v[i] = tk_composite(in[i].x, in[i].y);
    if (c.pop2 <= (unsigned)TK_TIE_CAP) {
        for (unsigned i = tid; i < c.pop2; i += 1024) v[n_sure + i] = tk_composite(in[TK_MAX_K + i].x, in[TK_MAX_K + i].y);
        total = n_sure + c.pop2;
    } else {
        // The bucket is too large to list: select its (k - n_sure) largest (low key bits, ~index) composites from the field.
        // Radix select over the 42 remaining bits, 11 per pass (10 in the first: the key's low bits), then collect.
        const unsigned P = (c.t1 << 11) | c.t2;
        unsigned need = (unsigned)k - n_sure;
        unsigned long long prefix = 0;                          // the decided high bits of the 42-bit remainder
        int decided = 0;                                        // how many of the 42 bits are decided
        const int widths[4] = {10, 11, 11, 10};
        for (int pass = 0; pass < 4; ++pass) {
            const int w = widths[pass], bins = 1 << w, shift = 42 - decided - w;
            for (int b = tid; b < bins; b += 1024) h[b] = 0;
            __syncthreads();
            for (long long i = tid; i < n; i += 1024) {
                const unsigned key = tk_key(x[i]);
                if ((key >> 10) != P) continue;
                const unsigned long long rem = ((unsigned long long)(key & 1023u) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
                if (decided && (rem >> (42 - decided)) != prefix) continue;
                atomicAdd(&h[(unsigned)(rem >> shift) & (unsigned)(bins - 1)], 1u);
            }
            __syncthreads();
            unsigned t, above, pop;
            tk_threshold(h, bins, need, part, res, t, above, pop);
            need -= above;
            prefix = (prefix << w) | t;
            decided += w;
        }
        // prefix = the 42-bit remainder of the LAST element to take (need == 1 now): take every bucket element >= it
        if (tid == 0) cursor = 0;
        __syncthreads();
        for (long long i = tid; i < n; i += 1024) {
            const unsigned key = tk_key(x[i]);
            if ((key >> 10) != P) continue;
            const unsigned long long rem = ((unsigned long long)(key & 1023u) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
            if (rem >= prefix) v[n_sure + atomicAdd(&cursor, 1u)] = tk_composite(key, (unsigned)i);
        }
        __syncthreads();
        total = (unsigned)k;
    }
    unsigned m = 1;
    while (m < total) m <<= 1;
    for (unsigned i = total + tid; i < m; i += 1024) v[i] = 0ull;         // below every real composite
    __syncthreads();
    // Bitonic sort, descending.  Pair t of a round is (i, i + stride), i = (t / stride) 2 stride + t % stride: with
    // stride <= 64 the 64 pairs of a wave's t-range lie inside ITS 128 elements, so those rounds need no workgroup
    // barrier (a wave's LDS operations complete in order) -- 10 barriers instead of 66 at 2048 candidates.
    auto round = [&](unsigned size, unsigned stride) {
        for (unsigned t = tid; t < (m >> 1); t += 1024) {
            const unsigned i = ((t / stride) * stride * 2) + (t % stride), j = i + stride;
            const bool desc = ((i & size) == 0);
            const unsigned long long a = v[i], b = v[j];
            if ((a < b) == desc) { v[i] = b; v[j] = a; }
        }
    };
    for (unsigned size = 2; size <= m; size <<= 1) {
        unsigned stride = size >> 1;
        for (; stride > 64; stride >>= 1) { round(size, stride); __syncthreads(); }
        for (; stride > 0; stride >>= 1) { round(size, stride); __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); }
        __syncthreads();
    }
    for (int i = tid; i < k; i += 1024) {
        const unsigned idx = 0xFFFFFFFFu - (unsigned)(v[i] & 0xFFFFFFFFull);
        const size_t o = (size_t)blockIdx.x * k + i;
        out_scores[o] = x[idx];
        out_index[o] = (long long)idx;
        if (boxes) {
            const float* bsrc = boxes + ((size_t)blockIdx.x * n + idx) * 6;
#pragma unroll
            for (int d = 0; d < 6; ++d) out_boxes[o * 6 + d] = bsrc[d];
        }
    }
    // leave the state as it was found: zero
    for (int b = tid; b < TK_BINS; b += 1024) { s->hist1[b] = 0; s->hist2[b] = 0; }
    if (tid == 0) { s->n_sure = 0; s->n_tie = 0; s->cut[0] = s->cut[1] = s->cut[2] = 0; }
}
}  // namespace

extern "C" int64_t scn_topk_scratch_bytes(int batch) {
    if (batch <= 0) return 0;
    return (int64_t)batch * ((int64_t)sizeof(TkState) + (int64_t)TK_CAND * (int64_t)sizeof(uint2));
}

extern "C" int scn_topk_boxes(const float* scores, const float* boxes, int batch, int64_t n, int k, float* out_scores,
                              int64_t* out_index, float* out_boxes, void* scratch, scn_stream_t stream) {
    SCN_REQUIRE(batch >= 0 && n >= 0 && k >= 0 && k <= TK_MAX_K && (int64_t)k <= n && n < 4294967295LL);
    if (batch == 0 || k == 0) return SCN_OK;
    SCN_REQUIRE(scores && out_scores && out_index && scratch && (!boxes || out_boxes));
    TkState* st = (TkState*)scratch;
    uint2* cand = (uint2*)((char*)scratch + (size_t)batch * sizeof(TkState));
    const int g = (int)std::min<int64_t>(256, (n + 2047) / 2048);        // (64 / 128 / 192 workgroups: 45 / 41 / 42 us per call against 42)
    hipLaunchKernelGGL(k_topk_hist<1>, dim3(g, batch), dim3(256), 0, S(stream), scores, (long long)n, k, st);
    SCN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_topk_hist<2>, dim3(g, batch), dim3(256), 0, S(stream), scores, (long long)n, k, st);
    SCN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_topk_compact, dim3(g, batch), dim3(256), 0, S(stream), scores, (long long)n, k, st, cand);
    SCN_LAUNCH_CHECK();
    static scn::DeviceOnce attr;
    if (attr.needed()) {
        SCN_HIP(hipFuncSetAttribute((const void*)k_topk_final, hipFuncAttributeMaxDynamicSharedMemorySize, TK_CAND * 8));
        attr.done();
    }
    hipLaunchKernelGGL(k_topk_final, dim3(batch), dim3(1024), (size_t)TK_CAND * 8, S(stream), scores, boxes, (long long)n, k, st,
                       cand, out_scores, (long long*)out_index, out_boxes);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}
