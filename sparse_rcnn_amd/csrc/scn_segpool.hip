// Per-sample pooling of a feature slab: mean / sum / max over the active rows of every sample (SURVEY.md §8f N1 remainder).
// Serves the reference's SparseGlobalPool / split_batch (ndsis/modules/custom_operations.py:24-59; reached by the sparse class
// network, model.py:507-512, module_factory.py:655-657), which builds a [samples, rows] bool mask on the HOST from
// get_spatial_locations() and boolean-indexes the slab once per sample.  Here: the sample of a row is column 3 of the
// grid's int32 coordinates, already in HBM; one streaming pass over the slab, HBM-bound.
//
// Sums accumulate in fp64 through atomics (a block first reduces its run of rows of one sample in registers, so a slab
// whose rows are grouped by sample issues one atomic per (block, column)); the maximum goes through an order-preserving
// integer encoding and atomicMax.  Both are independent of arrival order to within fp64 rounding of the sum.
#include "scn_common.h"

using scn::S;
using scn::cdiv;

namespace {

constexpr int SP_ROWS = 256;        // rows per block
constexpr int SP_TY = 4;            // row lanes per block (256 threads = 4 x 64 columns)

__device__ __forceinline__ int sp_enc(float f) {
    int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ __forceinline__ float sp_dec(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

// op: 0 mean, 1 sum, 2 max
template <int OP>
__global__ __launch_bounds__(256) void k_segpool_acc(const float* __restrict__ X, const int4* __restrict__ coords,
                                                     long long n, int c, int n_samples, double* __restrict__ acc,
                                                     int* __restrict__ imax, int* __restrict__ cnt,
                                                     int* __restrict__ unsorted) {
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (long long r0 = (long long)blockIdx.x * SP_ROWS; r0 < n; r0 += (long long)gridDim.x * SP_ROWS) {
        const long long r1 = r0 + SP_ROWS < n ? r0 + SP_ROWS : n;
        // row counts and the grouped-by-sample check: one thread per row
        for (long long r = r0 + threadIdx.x; r < r1; r += 256) {
            const int b = coords[r].w;
            if ((unsigned)b < (unsigned)n_samples) atomicAdd(&cnt[b], 1);
            if (r > 0 && coords[r - 1].w > b) atomicOr(unsorted, 1);
        }
        for (int cb = 0; cb < c; cb += 64) {
            const int col = cb + tx;
            if (col >= c) continue;
            int cur = -1;
            double s = 0.0;
            float m = 0.f;
            for (long long r = r0 + ty; r < r1; r += SP_TY) {
                const int b = coords[r].w;
                if ((unsigned)b >= (unsigned)n_samples) continue;
                const float v = X[r * c + col];
                if (b != cur) {
                    if (cur >= 0) {
                        if (OP == 2) atomicMax(&imax[(long long)cur * c + col], sp_enc(m));
                        else atomicAdd(&acc[(long long)cur * c + col], s);
                    }
                    cur = b; s = 0.0; m = v;
                }
                if (OP == 2) m = fmaxf(m, v); else s += (double)v;
            }
            if (cur >= 0) {
                if (OP == 2) atomicMax(&imax[(long long)cur * c + col], sp_enc(m));
                else atomicAdd(&acc[(long long)cur * c + col], s);
            }
        }
    }
}

__global__ void k_segpool_init(double* acc, int* imax, int* cnt, int* unsorted, long long bc, int n_samples, int op) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i < bc) {
        if (op == 2) imax[i] = (int)0x80000000; else acc[i] = 0.0;       // encoded -inf side / zero
    }
    if (i < n_samples) cnt[i] = 0;
    if (i == 0) *unsorted = 0;
}

__global__ void k_segpool_final(const double* __restrict__ acc, const int* __restrict__ imax, const int* __restrict__ cnt,
                                long long bc, int c, int op, float* __restrict__ Y) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= bc) return;
    const int n = cnt[i / c];
    float y = 0.f;                                 // a sample without rows pools to zeros (custom_operations.py:53-54)
    if (n > 0) y = op == 2 ? sp_dec(imax[i]) : (op == 0 ? (float)(acc[i] / (double)n) : (float)acc[i]);
    Y[i] = y;
}

__global__ void k_segpool_ties(const float* __restrict__ X, const float* __restrict__ Y, const int4* __restrict__ coords,
                               long long n, int c, int n_samples, int* __restrict__ ties) {
    const long long total = n * c;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / c;
        const int col = (int)(i - r * c), b = coords[r].w;
        if ((unsigned)b < (unsigned)n_samples && X[i] == Y[(long long)b * c + col]) atomicAdd(&ties[(long long)b * c + col], 1);
    }
}

__global__ void k_segpool_bwd(const float* __restrict__ X, const float* __restrict__ Y, const float* __restrict__ dY,
                              const int4* __restrict__ coords, long long n, int c, int n_samples, int op,
                              const int* __restrict__ cnt, const int* __restrict__ ties, float* __restrict__ dX) {
    const long long total = n * c;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / c;
        const int col = (int)(i - r * c), b = coords[r].w;
        float g = 0.f;
        if ((unsigned)b < (unsigned)n_samples) {
            const long long j = (long long)b * c + col;
            if (op == 0) g = dY[j] / (float)cnt[b];
            else if (op == 1) g = dY[j];
            else if (X[i] == Y[j]) g = dY[j] / (float)ties[j];      // torch.amax: evenly among equal maxima
        }
        dX[i] = g;
    }
}

}  // namespace

extern "C" int64_t scn_segment_pool_scratch_bytes(int n_samples, int c) {
    return (int64_t)n_samples * c * (int64_t)sizeof(double) + 256;
}

extern "C" int scn_sample_counts(const int32_t* coords, int64_t n, int n_samples, int32_t* cnt, int32_t* unsorted,
                                 scn_stream_t stream) {
    SCN_REQUIRE(n >= 0 && n_samples >= 0 && cnt && unsorted && (n == 0 || coords));
    hipStream_t st = S(stream);
    hipLaunchKernelGGL(k_segpool_init, dim3((unsigned)cdiv(n_samples > 0 ? n_samples : 1, 256)), dim3(256), 0, st,
                       (double*)nullptr, (int*)nullptr, cnt, unsorted, 0ll, n_samples, 0);
    if (n > 0 && n_samples > 0)
        hipLaunchKernelGGL(k_segpool_acc<0>, dim3((unsigned)(cdiv(n, SP_ROWS) > 2048 ? 2048 : cdiv(n, SP_ROWS))), dim3(256), 0, st,
                           (const float*)nullptr, (const int4*)coords, (long long)n, 0, n_samples, (double*)nullptr,
                           (int*)nullptr, cnt, unsorted);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

extern "C" int scn_segment_pool_fwd(const float* X, const int32_t* coords, int64_t n, int c, int n_samples, int op,
                                    float* Y, int32_t* cnt, int32_t* unsorted, void* scratch, scn_stream_t stream) {
    SCN_REQUIRE(n >= 0 && c >= 1 && n_samples >= 0 && op >= 0 && op <= 2);
    if (n_samples == 0) return SCN_OK;
    SCN_REQUIRE(Y && cnt && unsorted && scratch && (n == 0 || (X && coords)));
    hipStream_t st = S(stream);
    const long long bc = (long long)n_samples * c;
    double* acc = (double*)scratch;
    int* imax = (int*)scratch;
    hipLaunchKernelGGL(k_segpool_init, dim3((unsigned)cdiv(bc > n_samples ? bc : n_samples, 256)), dim3(256), 0, st, acc, imax,
                       cnt, unsorted, bc, n_samples, op);
    if (n > 0) {
        const unsigned g = (unsigned)(cdiv(n, SP_ROWS) > 2048 ? 2048 : cdiv(n, SP_ROWS));
        const int4* c4 = (const int4*)coords;
        if (op == 0) hipLaunchKernelGGL(k_segpool_acc<0>, dim3(g), dim3(256), 0, st, X, c4, (long long)n, c, n_samples, acc, imax, cnt, unsorted);
        else if (op == 1) hipLaunchKernelGGL(k_segpool_acc<1>, dim3(g), dim3(256), 0, st, X, c4, (long long)n, c, n_samples, acc, imax, cnt, unsorted);
        else hipLaunchKernelGGL(k_segpool_acc<2>, dim3(g), dim3(256), 0, st, X, c4, (long long)n, c, n_samples, acc, imax, cnt, unsorted);
    }
    hipLaunchKernelGGL(k_segpool_final, dim3((unsigned)cdiv(bc, 256)), dim3(256), 0, st, (const double*)acc, (const int*)imax,
                       (const int*)cnt, bc, c, op, Y);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

extern "C" int scn_segment_pool_bwd(const float* X, const float* Y, const float* dY, const int32_t* coords, int64_t n, int c,
                                    int n_samples, int op, const int32_t* cnt, float* dX, void* scratch,
                                    scn_stream_t stream) {
    SCN_REQUIRE(n >= 0 && c >= 1 && n_samples >= 0 && op >= 0 && op <= 2);
    if (n == 0) return SCN_OK;
    SCN_REQUIRE(dY && coords && cnt && dX && (op != 2 || (X && Y && scratch)));
    hipStream_t st = S(stream);
    int* ties = (int*)scratch;
    const int g = scn::ew_grid(n * c, 256);
    if (op == 2) {
        SCN_HIP(hipMemsetAsync(ties, 0, (size_t)n_samples * c * sizeof(int), st));
        hipLaunchKernelGGL(k_segpool_ties, dim3(g), dim3(256), 0, st, X, Y, (const int4*)coords, (long long)n, c, n_samples, ties);
    }
    hipLaunchKernelGGL(k_segpool_bwd, dim3(g), dim3(256), 0, st, X, Y, dY, (const int4*)coords, (long long)n, c, n_samples, op,
                       (const int*)cnt, (const int*)ties, dX);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}
