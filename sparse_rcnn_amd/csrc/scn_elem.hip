// HBM-bound feature kernels of libscn_mi355x: ReLU / AddTable, BatchNorm(Leaky)ReLU, InputLayer / OutputLayer
// feature movement, SparseToDense.  All are pure streaming or row-gather kernels (SURVEY.md §8d regime (i)):
// 16-byte vector accesses where alignment allows, grid-stride loops, fp64 accumulation for reductions so results
// do not depend on arrival order.
#include "scn_common.h"

using scn::S;
using scn::cdiv;

// ------------------------------------------------------------------------------------------------
// ReLU / add
// ------------------------------------------------------------------------------------------------
template <int OP>   // 0 relu fwd (a), 1 relu bwd (a = x, b = dy), 2 add
__global__ void k_ew(const float* __restrict__ a, const float* __restrict__ b, long long count, float* __restrict__ y,
                     int vec) {
    const long long tid = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    if (vec) {
        const long long n4 = count >> 2;
        for (long long i = tid; i < n4; i += stride) {
            float4 u = ((const float4*)a)[i], r;
            if (OP == 0) {
                r = make_float4(fmaxf(u.x, 0.f), fmaxf(u.y, 0.f), fmaxf(u.z, 0.f), fmaxf(u.w, 0.f));
            } else {
                float4 w = ((const float4*)b)[i];
                if (OP == 1) r = make_float4(u.x > 0.f ? w.x : 0.f, u.y > 0.f ? w.y : 0.f, u.z > 0.f ? w.z : 0.f,
                                             u.w > 0.f ? w.w : 0.f);
                else r = make_float4(u.x + w.x, u.y + w.y, u.z + w.z, u.w + w.w);
            }
            ((float4*)y)[i] = r;
        }
        for (long long i = (n4 << 2) + tid; i < count; i += stride) {
            float u = a[i];
            y[i] = OP == 0 ? fmaxf(u, 0.f) : (OP == 1 ? (u > 0.f ? b[i] : 0.f) : u + b[i]);
        }
    } else {
        for (long long i = tid; i < count; i += stride) {
            float u = a[i];
            y[i] = OP == 0 ? fmaxf(u, 0.f) : (OP == 1 ? (u > 0.f ? b[i] : 0.f) : u + b[i]);
        }
    }
}

template <int OP>
static int ew_launch(const float* a, const float* b, int64_t count, float* y, scn_stream_t stream) {
    SCN_REQUIRE(count >= 0);
    if (count == 0) return SCN_OK;
    SCN_REQUIRE(a && y && (OP == 0 || b));
    int vec = (((uintptr_t)a | (uintptr_t)y | (uintptr_t)b) & 15) == 0;
    hipLaunchKernelGGL(k_ew<OP>, dim3(scn::ew_grid(cdiv(count, 4), 256)), dim3(256), 0, S(stream), a, b,
                       (long long)count, y, vec);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

extern "C" int scn_relu_fwd(const float* X, int64_t count, float* Y, scn_stream_t stream) {
    return ew_launch<0>(X, nullptr, count, Y, stream);
}
extern "C" int scn_relu_bwd(const float* X, const float* dY, int64_t count, float* dX, scn_stream_t stream) {
    return ew_launch<1>(X, dY, count, dX, stream);
}
extern "C" int scn_add(const float* A, const float* B, int64_t count, float* Y, scn_stream_t stream) {
    return ew_launch<2>(A, B, count, Y, stream);
}

// ------------------------------------------------------------------------------------------------
// row gathers / scatters.  One thread per (row, 4-channel group) when c % 4 == 0, else per element.
// ------------------------------------------------------------------------------------------------
__global__ void k_gather_rows(const float* __restrict__ X, const int* __restrict__ rows, long long m, int c,
                              float* __restrict__ Y, int vec) {
    const long long tid = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    if (vec) {
        const int c4 = c >> 2;
        for (long long i = tid; i < m * c4; i += stride) {
            long long r = i / c4;
            int g = (int)(i - r * c4);
            ((float4*)Y)[i] = ((const float4*)(X + (long long)rows[r] * c))[g];
        }
    } else {
        for (long long i = tid; i < m * c; i += stride) {
            long long r = i / c;
            Y[i] = X[(long long)rows[r] * c + (i - r * c)];
        }
    }
}

extern "C" int scn_gather_rows(const float* X, const int32_t* rows, int64_t m, int c, float* Y, scn_stream_t stream) {
    SCN_REQUIRE(m >= 0 && c >= 1);
    if (m == 0) return SCN_OK;
    SCN_REQUIRE(X && rows && Y);
    int vec = (c % 4 == 0) && ((((uintptr_t)X | (uintptr_t)Y) & 15) == 0);
    hipLaunchKernelGGL(k_gather_rows, dim3(scn::ew_grid(m * (vec ? c / 4 : c), 256)), dim3(256), 0, S(stream), X, rows,
                       (long long)m, c, Y, vec);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

// acc64[row][ch] += v[item][ch]  (fp64 atomics: the sum of a handful of fp32 values is exact in fp64, so the
// rounded result does not depend on arrival order)
__global__ void k_scatter_add64(const float* __restrict__ V, const int* __restrict__ item_row, long long n_items, int c,
                                double* __restrict__ acc) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n_items * c;
         i += (long long)gridDim.x * blockDim.x) {
        long long it = i / c;
        int ch = (int)(i - it * c);
        atomicAdd(&acc[(long long)item_row[it] * c + ch], (double)V[i]);
    }
}

__global__ void k_finish64(const double* __restrict__ acc, const int* __restrict__ row_count, long long n_rows, int c,
                           int mean, float* __restrict__ Y) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n_rows * c;
         i += (long long)gridDim.x * blockDim.x) {
        double v = acc[i];
        if (mean) {
            int cnt = row_count[i / c];
            v /= (double)(cnt > 0 ? cnt : 1);
        }
        Y[i] = (float)v;
    }
}

__global__ void k_row_last(const int* __restrict__ item_row, long long n_items, int* __restrict__ row_last) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n_items;
         i += (long long)gridDim.x * blockDim.x)
        atomicMax(&row_last[item_row[i]], (int)i);
}

extern "C" int scn_segment_sum(const float* dY, const int32_t* item_row, int64_t n_items, int64_t n_rows, int c,
                               float* dX, double* acc64, scn_stream_t stream) {
    SCN_REQUIRE(n_items >= 0 && n_rows >= 0 && c >= 1);
    if (n_rows == 0) return SCN_OK;
    SCN_REQUIRE(dX && acc64);
    SCN_HIP(hipMemsetAsync(acc64, 0, sizeof(double) * n_rows * c, S(stream)));
    if (n_items) {
        SCN_REQUIRE(dY && item_row);
        hipLaunchKernelGGL(k_scatter_add64, dim3(scn::ew_grid(n_items * c, 256)), dim3(256), 0, S(stream), dY, item_row,
                           (long long)n_items, c, acc64);
        SCN_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_finish64, dim3(scn::ew_grid(n_rows * c, 256)), dim3(256), 0, S(stream), (const double*)acc64,
                       (const int*)nullptr, (long long)n_rows, c, 0, dX);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

extern "C" int scn_input_fwd(const float* feats, const int32_t* item_row, const int32_t* row_count,
                             const int32_t* row_first, int64_t n_items, int64_t n_rows, int c, int mode, float* Y,
                             double* acc64, int32_t* row_last, scn_stream_t stream) {
    SCN_REQUIRE(n_items >= 0 && n_rows >= 0 && c >= 1 && mode >= 0 && mode <= 4);
    if (n_rows == 0) return SCN_OK;
    SCN_REQUIRE(feats && item_row && Y);
    hipStream_t st = S(stream);
    if (mode == 3 || mode == 4) {
        SCN_REQUIRE(acc64 && (mode == 3 || row_count));
        SCN_HIP(hipMemsetAsync(acc64, 0, sizeof(double) * n_rows * c, st));
        hipLaunchKernelGGL(k_scatter_add64, dim3(scn::ew_grid(n_items * c, 256)), dim3(256), 0, st, feats, item_row,
                           (long long)n_items, c, acc64);
        SCN_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_finish64, dim3(scn::ew_grid(n_rows * c, 256)), dim3(256), 0, st, (const double*)acc64,
                           row_count, (long long)n_rows, c, mode == 4, Y);
        SCN_LAUNCH_CHECK();
        return SCN_OK;
    }
    const int32_t* src = row_first;                 // modes 0 and 2: the first (for mode 0: only) item of each row
    if (mode == 1) {
        SCN_REQUIRE(row_last);
        SCN_HIP(hipMemsetAsync(row_last, 0xFF, sizeof(int32_t) * n_rows, st));
        hipLaunchKernelGGL(k_row_last, dim3(scn::ew_grid(n_items, 256)), dim3(256), 0, st, item_row, (long long)n_items,
                           row_last);
        SCN_LAUNCH_CHECK();
        src = row_last;
    }
    SCN_REQUIRE(src);
    return scn_gather_rows(feats, src, n_rows, c, Y, stream);
}

__global__ void k_input_bwd(const float* __restrict__ dY, const int* __restrict__ item_row,
                            const int* __restrict__ row_count, const int* __restrict__ winner, long long n_items, int c,
                            int mode, float* __restrict__ dF) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n_items * c;
         i += (long long)gridDim.x * blockDim.x) {
        long long it = i / c;
        int ch = (int)(i - it * c);
        int r = item_row[it];
        float g = dY[(long long)r * c + ch];
        if (mode == 4) g /= (float)row_count[r];
        else if (mode == 1 || mode == 2) g = (winner[r] == (int)it) ? g : 0.f;
        dF[i] = g;
    }
}

extern "C" int scn_input_bwd(const float* dY, const int32_t* item_row, const int32_t* row_count,
                             const int32_t* row_first, const int32_t* row_last, int64_t n_items, int c, int mode,
                             float* dfeats, scn_stream_t stream) {
    SCN_REQUIRE(n_items >= 0 && c >= 1 && mode >= 0 && mode <= 4);
    if (n_items == 0) return SCN_OK;
    SCN_REQUIRE(dY && item_row && dfeats);
    const int32_t* winner = mode == 1 ? row_last : row_first;
    SCN_REQUIRE(mode != 4 || row_count);
    SCN_REQUIRE((mode != 1 && mode != 2) || winner);
    hipLaunchKernelGGL(k_input_bwd, dim3(scn::ew_grid(n_items * c, 256)), dim3(256), 0, S(stream), dY, item_row,
                       row_count, winner, (long long)n_items, c, mode, dfeats);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

// ------------------------------------------------------------------------------------------------
// SparseToDense: out[b][ch][x][y][z]
// ------------------------------------------------------------------------------------------------
template <bool BWD>
__global__ void k_s2d(const float* __restrict__ src, const int4* __restrict__ coords, long long n, int c, long long sx,
                      long long sy, long long sz, float* __restrict__ dst) {
    const long long vol = sx * sy * sz;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n * c;
         i += (long long)gridDim.x * blockDim.x) {
        // channel-major inside a row so that, for a fixed channel, neighbouring rows hit nearby z
        long long r = i % n;
        int ch = (int)(i / n);
        int4 p = coords[r];
        long long d = ((long long)p.w * c + ch) * vol + ((long long)p.x * sy + p.y) * sz + p.z;
        if (BWD) dst[r * c + ch] = src[d];
        else dst[d] = src[r * c + ch];
    }
}

extern "C" int scn_sparse_to_dense_fwd(const float* X, const int32_t* coords, int64_t n, int c,
                                       const int64_t* size3_host, float* out, scn_stream_t stream) {
    SCN_REQUIRE(n >= 0 && c >= 1 && size3_host);
    if (n == 0) return SCN_OK;
    SCN_REQUIRE(X && coords && out);
    hipLaunchKernelGGL(k_s2d<false>, dim3(scn::ew_grid(n * c, 256)), dim3(256), 0, S(stream), X, (const int4*)coords,
                       (long long)n, c, (long long)size3_host[0], (long long)size3_host[1], (long long)size3_host[2],
                       out);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

extern "C" int scn_sparse_to_dense_bwd(const float* dOut, const int32_t* coords, int64_t n, int c,
                                       const int64_t* size3_host, float* dX, scn_stream_t stream) {
    SCN_REQUIRE(n >= 0 && c >= 1 && size3_host);
    if (n == 0) return SCN_OK;
    SCN_REQUIRE(dOut && coords && dX);
    hipLaunchKernelGGL(k_s2d<true>, dim3(scn::ew_grid(n * c, 256)), dim3(256), 0, S(stream), dOut, (const int4*)coords,
                       (long long)n, c, (long long)size3_host[0], (long long)size3_host[1], (long long)size3_host[2],
                       dX);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

// ------------------------------------------------------------------------------------------------
// BatchNorm(Leaky)ReLU.  Statistics in fp64 (column sums of x and x^2), two-stage, fixed order.
// ------------------------------------------------------------------------------------------------
static constexpr int BN_BLOCKS = 256;

// partial[blk][2][c] doubles: sum, sum of squares (or for bwd: sum g, sum g*xhat)
template <bool BWD>
__global__ __launch_bounds__(256) void k_bn_partial(const float* __restrict__ X, const float* __restrict__ dY,
                                                    long long n, int c, const float* __restrict__ mean,
                                                    const float* __restrict__ var, float eps,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    float leak, double* __restrict__ partial) {
    const long long rows_per_block = (n + gridDim.x - 1) / gridDim.x;
    const long long r_lo = blockIdx.x * rows_per_block;
    long long r_hi = r_lo + rows_per_block;
    if (r_hi > n) r_hi = n;
    __shared__ double red0[256], red1[256];
    for (int c0 = 0; c0 < c; c0 += 256) {
        const int width = min(256, c - c0);
        const int rows_par = 256 / width;
        const int col = threadIdx.x % width, rsub = threadIdx.x / width;
        double s0 = 0.0, s1 = 0.0;
        if (rsub < rows_par) {
            float mu = 0.f, is = 0.f, ga = 0.f, be = 0.f;
            if (BWD) {
                mu = mean[c0 + col];
                is = rsqrtf(var[c0 + col] + eps);
                ga = gamma[c0 + col];
                be = beta[c0 + col];
            }
            for (long long r = r_lo + rsub; r < r_hi; r += rows_par) {
                float x = X[r * c + c0 + col];
                if (BWD) {
                    float xh = (x - mu) * is;
                    float pre = xh * ga + be;
                    float g = dY[r * c + c0 + col] * (pre > 0.f ? 1.f : leak);
                    s0 += (double)g;
                    s1 += (double)g * (double)xh;
                } else {
                    s0 += (double)x;
                    s1 += (double)x * (double)x;
                }
            }
        }
        red0[threadIdx.x] = s0;
        red1[threadIdx.x] = s1;
        __syncthreads();
        if (threadIdx.x < width) {
            double t0 = 0.0, t1 = 0.0;
            for (int k = 0; k < rows_par; ++k) {
                t0 += red0[k * width + threadIdx.x];
                t1 += red1[k * width + threadIdx.x];
            }
            partial[((long long)blockIdx.x * 2 + 0) * c + c0 + threadIdx.x] = t0;
            partial[((long long)blockIdx.x * 2 + 1) * c + c0 + threadIdx.x] = t1;
        }
        __syncthreads();
    }
}

__global__ void k_bn_stats_final(const double* __restrict__ partial, int nblk, int c, long long n,
                                 float* __restrict__ mean, float* __restrict__ var) {
    int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= c) return;
    double s0 = 0.0, s1 = 0.0;
    for (int b = 0; b < nblk; ++b) {
        s0 += partial[((long long)b * 2 + 0) * c + col];
        s1 += partial[((long long)b * 2 + 1) * c + col];
    }
    double mu = n > 0 ? s0 / (double)n : 0.0;
    double v = n > 0 ? s1 / (double)n - mu * mu : 0.0;
    mean[col] = (float)mu;
    var[col] = (float)(v > 0.0 ? v : 0.0);
}

// SyncBN pieces (batch statistics over all ranks; module_factory.py:92-102: the reference's BatchNorm sees the whole batch
// as one feature matrix): the per-channel sums leave the library as float64 [2][c] so that the caller can all-reduce them.
__global__ void k_bn_sums_final(const double* __restrict__ partial, int nblk, int c, double* __restrict__ sums) {
    int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= c) return;
    double s0 = 0.0, s1 = 0.0;
    for (int b = 0; b < nblk; ++b) {
        s0 += partial[((long long)b * 2 + 0) * c + col];
        s1 += partial[((long long)b * 2 + 1) * c + col];
    }
    sums[col] = s0;
    sums[c + col] = s1;
}

extern "C" int64_t scn_bn_scratch_bytes(int c) { return (int64_t)sizeof(double) * (BN_BLOCKS * 2 + 2) * c; }

extern "C" int scn_bn_stats(const float* X, int64_t n, int c, float* mean, float* var_biased, void* scratch,
                            scn_stream_t stream) {
    SCN_REQUIRE(n >= 0 && c >= 1 && mean && var_biased && scratch);
    SCN_REQUIRE(n == 0 || X);
    hipLaunchKernelGGL(k_bn_partial<false>, dim3(BN_BLOCKS), dim3(256), 0, S(stream), X, (const float*)nullptr,
                       (long long)n, c, (const float*)nullptr, (const float*)nullptr, 0.f, (const float*)nullptr,
                       (const float*)nullptr, 0.f, (double*)scratch);
    SCN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_bn_stats_final, dim3((c + 255) / 256), dim3(256), 0, S(stream), (const double*)scratch,
                       BN_BLOCKS, c, (long long)n, mean, var_biased);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

extern "C" int scn_bn_sums(const float* X, int64_t n, int c, double* sums, void* scratch, scn_stream_t stream) {
    SCN_REQUIRE(n >= 0 && c >= 1 && sums && scratch);
    SCN_REQUIRE(n == 0 || X);
    hipLaunchKernelGGL(k_bn_partial<false>, dim3(BN_BLOCKS), dim3(256), 0, S(stream), X, (const float*)nullptr,
                       (long long)n, c, (const float*)nullptr, (const float*)nullptr, 0.f, (const float*)nullptr,
                       (const float*)nullptr, 0.f, (double*)scratch);
    SCN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_bn_sums_final, dim3((c + 255) / 256), dim3(256), 0, S(stream), (const double*)scratch, BN_BLOCKS,
                       c, sums);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

__global__ void k_bn_fwd(const float* __restrict__ X, long long n, int c, const float* __restrict__ mean,
                         const float* __restrict__ var, float eps, const float* __restrict__ gamma,
                         const float* __restrict__ beta, float leak, float* __restrict__ Y) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n * c;
         i += (long long)gridDim.x * blockDim.x) {
        int ch = (int)(i % c);
        float y = (X[i] - mean[ch]) * rsqrtf(var[ch] + eps) * gamma[ch] + beta[ch];
        Y[i] = y > 0.f ? y : y * leak;
    }
}

extern "C" int scn_bn_fwd(const float* X, int64_t n, int c, const float* mean, const float* var, float eps,
                          const float* gamma, const float* beta, float leak, float* Y, scn_stream_t stream) {
    SCN_REQUIRE(n >= 0 && c >= 1);
    if (n == 0) return SCN_OK;
    SCN_REQUIRE(X && mean && var && gamma && beta && Y);
    hipLaunchKernelGGL(k_bn_fwd, dim3(scn::ew_grid(n * c, 256)), dim3(256), 0, S(stream), X, (long long)n, c, mean, var,
                       eps, gamma, beta, leak, Y);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

__global__ void k_bn_bwd_final(const double* __restrict__ partial, int nblk, int c, float* __restrict__ dgamma,
                               float* __restrict__ dbeta, double* __restrict__ sums /* [2][c] */) {
    int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= c) return;
    double s0 = 0.0, s1 = 0.0;
    for (int b = 0; b < nblk; ++b) {
        s0 += partial[((long long)b * 2 + 0) * c + col];
        s1 += partial[((long long)b * 2 + 1) * c + col];
    }
    dbeta[col] = (float)s0;
    dgamma[col] = (float)s1;
    sums[col] = s0;
    sums[c + col] = s1;
}

__global__ void k_bn_bwd_dx(const float* __restrict__ X, const float* __restrict__ dY, long long n, int c,
                            const float* __restrict__ mean, const float* __restrict__ var, float eps,
                            const float* __restrict__ gamma, const float* __restrict__ beta, float leak, int training,
                            const double* __restrict__ sums, float* __restrict__ dX, long long n_stat) {
    const double inv_n = n_stat > 0 ? 1.0 / (double)n_stat : 0.0;   // rows the statistics were taken over (all ranks')
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n * c;
         i += (long long)gridDim.x * blockDim.x) {
        int ch = (int)(i % c);
        float is = rsqrtf(var[ch] + eps);
        float xh = (X[i] - mean[ch]) * is;
        float pre = xh * gamma[ch] + beta[ch];
        float g = dY[i] * (pre > 0.f ? 1.f : leak);
        float dx;
        if (training) dx = gamma[ch] * is * (g - (float)(sums[ch] * inv_n) - xh * (float)(sums[c + ch] * inv_n));
        else dx = gamma[ch] * is * g;
        dX[i] = dx;
    }
}

extern "C" int scn_bn_bwd(const float* X, const float* dY, int64_t n, int c, const float* mean, const float* var,
                          float eps, const float* gamma, const float* beta, float leak, int training, float* dX,
                          float* dgamma, float* dbeta, void* scratch, scn_stream_t stream) {
    SCN_REQUIRE(n >= 0 && c >= 1 && mean && var && gamma && beta && dgamma && dbeta && scratch);
    SCN_REQUIRE(n == 0 || (X && dY && dX));
    double* partial = (double*)scratch;                       // [BN_BLOCKS][2][c]
    double* sums = partial + (long long)BN_BLOCKS * 2 * c;    // [2][c]
    hipLaunchKernelGGL(k_bn_partial<true>, dim3(BN_BLOCKS), dim3(256), 0, S(stream), X, dY, (long long)n, c, mean, var,
                       eps, gamma, beta, leak, partial);
    SCN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_bn_bwd_final, dim3((c + 255) / 256), dim3(256), 0, S(stream), (const double*)partial, BN_BLOCKS,
                       c, dgamma, dbeta, sums);
    SCN_LAUNCH_CHECK();
    if (n) {
        hipLaunchKernelGGL(k_bn_bwd_dx, dim3(scn::ew_grid(n * c, 256)), dim3(256), 0, S(stream), X, dY, (long long)n, c,
                           mean, var, eps, gamma, beta, leak, training, (const double*)sums, dX, (long long)n);
        SCN_LAUNCH_CHECK();
    }
    return SCN_OK;
}

// scn_bn_bwd in two halves with the reduction exposed: local (sum g, sum g x^) -> caller all-reduces -> apply.
extern "C" int scn_bn_bwd_reduce(const float* X, const float* dY, int64_t n, int c, const float* mean, const float* var,
                                 float eps, const float* gamma, const float* beta, float leak, float* dgamma,
                                 float* dbeta, double* sums, void* scratch, scn_stream_t stream) {
    SCN_REQUIRE(n >= 0 && c >= 1 && mean && var && gamma && beta && dgamma && dbeta && sums && scratch);
    SCN_REQUIRE(n == 0 || (X && dY));
    hipLaunchKernelGGL(k_bn_partial<true>, dim3(BN_BLOCKS), dim3(256), 0, S(stream), X, dY, (long long)n, c, mean, var,
                       eps, gamma, beta, leak, (double*)scratch);
    SCN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_bn_bwd_final, dim3((c + 255) / 256), dim3(256), 0, S(stream), (const double*)scratch, BN_BLOCKS,
                       c, dgamma, dbeta, sums);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

extern "C" int scn_bn_bwd_apply(const float* X, const float* dY, int64_t n, int c, const float* mean, const float* var,
                                float eps, const float* gamma, const float* beta, float leak, const double* sums,
                                int64_t n_stat, float* dX, scn_stream_t stream) {
    SCN_REQUIRE(n >= 0 && c >= 1 && n_stat >= n && mean && var && gamma && beta && sums);
    if (n == 0) return SCN_OK;
    SCN_REQUIRE(X && dY && dX);
    hipLaunchKernelGGL(k_bn_bwd_dx, dim3(scn::ew_grid(n * c, 256)), dim3(256), 0, S(stream), X, dY, (long long)n, c, mean,
                       var, eps, gamma, beta, leak, 1, sums, dX, (long long)n_stat);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

// ------------------------------------------------------------------------------------------------
// MaxPooling / AveragePooling, pool_size = pool_stride = 2 (module_factory.py:315-354), on the strided rulebook.
//   max: Y[c] = max(0, max over existing children)   (output zero-initialised, as the upstream CPU path does)
//   avg: Y[c] = (sum over existing children) / 8     (inactive children count as zeros, like the dense twin)
// ------------------------------------------------------------------------------------------------
__global__ void k_pool_fwd(const float* __restrict__ X, const int* __restrict__ child, long long n_coarse, int c,
                           int avg, float* __restrict__ Y) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n_coarse * c;
         i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / c;
        const int ch = (int)(i - r * c);
        float acc = 0.f;
        const int n_off = (avg >> 8) ? (avg >> 8) : 8;                  // `average` carries the pool volume above bit 8 (0: 2^3)
        const bool mean = avg & 1;
        for (int o = 0; o < n_off; ++o) {
            const int f = child[(long long)o * n_coarse + r];
            if (f >= 0) {
                const float v = X[(long long)f * c + ch];
                acc = mean ? acc + v : fmaxf(acc, v);
            }
        }
        Y[i] = mean ? acc * (1.0f / (float)n_off) : acc;
    }
}

__global__ void k_pool_bwd(const float* __restrict__ X, const float* __restrict__ Y, const float* __restrict__ dY,
                           const int* __restrict__ parent, long long n_fine, int c, int avg, float* __restrict__ dX) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n_fine * c;
         i += (long long)gridDim.x * blockDim.x) {
        const long long f = i / c;
        const int ch = (int)(i - f * c);
        const long long o = (long long)parent[f] * c + ch;
        const int n_off = (avg >> 8) ? (avg >> 8) : 8;
        dX[i] = (avg & 1) ? dY[o] * (1.0f / (float)n_off) : (X[i] == Y[o] ? dY[o] : 0.f);
    }
}

extern "C" int scn_pool_fwd(const float* X, const int32_t* child, int64_t n_coarse, int c, int average, float* Y,
                            scn_stream_t stream) {
    SCN_REQUIRE(n_coarse >= 0 && c >= 1);
    if (n_coarse == 0) return SCN_OK;
    SCN_REQUIRE(X && child && Y);
    hipLaunchKernelGGL(k_pool_fwd, dim3(scn::ew_grid(n_coarse * c, 256)), dim3(256), 0, S(stream), X, child,
                       (long long)n_coarse, c, average, Y);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

extern "C" int scn_pool_bwd(const float* X, const float* Y, const float* dY, const int32_t* parent, int64_t n_fine, int c,
                            int average, float* dX, scn_stream_t stream) {
    SCN_REQUIRE(n_fine >= 0 && c >= 1);
    if (n_fine == 0) return SCN_OK;
    SCN_REQUIRE(X && Y && dY && parent && dX);
    hipLaunchKernelGGL(k_pool_bwd, dim3(scn::ew_grid(n_fine * c, 256)), dim3(256), 0, S(stream), X, Y, dY, parent,
                       (long long)n_fine, c, average, dX);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

// ------------------------------------------------------------------------------------------------
// Mask-head epilogue on the device: consumers of the ROI selection in CSR form.
//   rows m = 0..M-1 of the crop are box-major; box_of[m] = box, src_point[m] = point row in the batch.
// ------------------------------------------------------------------------------------------------
// SparseMaskPredictor.forward (model.py:859-882): per sample a dense [boxes, points] mask, sigmoid of the score of the
// box's class at the points inside the box, 0 elsewhere and for invalid classes.  out is pre-zeroed; row_base[box] =
// offset of the box's row in `out` minus the first point row of its sample.
__global__ void k_mask_scatter(const float* __restrict__ scores, long long m, int k, const int* __restrict__ src_point,
                               const int* __restrict__ box_of, const long long* __restrict__ class_of_box,
                               int num_valid, const long long* __restrict__ row_base, float* __restrict__ out) {
    for (long long r = blockIdx.x * (long long)blockDim.x + threadIdx.x; r < m; r += (long long)gridDim.x * blockDim.x) {
        const int box = box_of[r];
        const long long cls = class_of_box[box];
        const bool valid = cls >= 0 && (num_valid == 0 || cls < num_valid) && cls < k;
        if (!valid) continue;
        const float x = scores[r * k + cls];
        out[row_base[box] + src_point[r]] = 1.f / (1.f + __expf(-x));
    }
}

extern "C" int scn_mask_scatter(const float* scores, int64_t m, int k, const int32_t* src_point, const int32_t* box_of,
                                const int64_t* class_of_box, int num_valid, const int64_t* row_base, float* out,
                                scn_stream_t stream) {
    SCN_REQUIRE(m >= 0 && k >= 1 && num_valid >= 0);
    if (m == 0) return SCN_OK;
    SCN_REQUIRE(scores && src_point && box_of && class_of_box && row_base && out);
    hipLaunchKernelGGL(k_mask_scatter, dim3(scn::ew_grid(m, 256)), dim3(256), 0, S(stream), scores, (long long)m, k,
                       src_point, box_of, (const long long*)class_of_box, num_valid, (const long long*)row_base, out);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

// SparseMaskLossSelector (model.py:1157-1227): pred[r] = scores[r][label[box]]; gt[r] = gt_flat[gt_base[box] + point]
// (gt_base[box] = offset of the associated ground-truth mask row minus the sample's first point row).  Boxes with
// label < 0 or gt_base < 0 are not kept: their rows get pred = gt = 0 and keep_row = 0.
__global__ void k_mask_gather(const float* __restrict__ scores, long long m, int k, const int* __restrict__ src_point,
                              const int* __restrict__ box_of, const long long* __restrict__ label_of_box,
                              const long long* __restrict__ gt_base, const float* __restrict__ gt_flat,
                              float* __restrict__ pred, float* __restrict__ gt, unsigned char* __restrict__ keep_row) {
    for (long long r = blockIdx.x * (long long)blockDim.x + threadIdx.x; r < m; r += (long long)gridDim.x * blockDim.x) {
        const int box = box_of[r];
        const long long lab = label_of_box[box], gb = gt_base[box];
        const bool keep = lab >= 0 && lab < k && gb >= 0;
        pred[r] = keep ? scores[r * k + lab] : 0.f;
        gt[r] = keep ? gt_flat[gb + src_point[r]] : 0.f;
        if (keep_row) keep_row[r] = keep ? 1 : 0;
    }
}

extern "C" int scn_mask_gather(const float* scores, int64_t m, int k, const int32_t* src_point, const int32_t* box_of,
                               const int64_t* label_of_box, const int64_t* gt_base, const float* gt_flat, float* pred,
                               float* gt, uint8_t* keep_row, scn_stream_t stream) {
    SCN_REQUIRE(m >= 0 && k >= 1);
    if (m == 0) return SCN_OK;
    SCN_REQUIRE(scores && src_point && box_of && label_of_box && gt_base && gt_flat && pred && gt);
    hipLaunchKernelGGL(k_mask_gather, dim3(scn::ew_grid(m, 256)), dim3(256), 0, S(stream), scores, (long long)m, k,
                       src_point, box_of, (const long long*)label_of_box, (const long long*)gt_base, gt_flat, pred, gt,
                       keep_row);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

// backward of pred w.r.t. scores: dscores[r][c] = (c == label[box]) ? dpred[r] : 0   (whole rows written)
__global__ void k_mask_gather_bwd(const float* __restrict__ dpred, long long m, int k, const int* __restrict__ box_of,
                                  const long long* __restrict__ label_of_box, const long long* __restrict__ gt_base,
                                  float* __restrict__ dscores) {
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < m * k; e += (long long)gridDim.x * blockDim.x) {
        const long long r = e / k;
        const int c = (int)(e - r * k);
        const int box = box_of[r];
        const long long lab = label_of_box[box];
        dscores[e] = (lab == c && gt_base[box] >= 0) ? dpred[r] : 0.f;
    }
}

extern "C" int scn_mask_gather_bwd(const float* dpred, int64_t m, int k, const int32_t* box_of,
                                   const int64_t* label_of_box, const int64_t* gt_base, float* dscores,
                                   scn_stream_t stream) {
    SCN_REQUIRE(m >= 0 && k >= 1);
    if (m == 0) return SCN_OK;
    SCN_REQUIRE(dpred && box_of && label_of_box && gt_base && dscores);
    hipLaunchKernelGGL(k_mask_gather_bwd, dim3(scn::ew_grid(m * k, 256)), dim3(256), 0, S(stream), dpred, (long long)m, k,
                       box_of, (const long long*)label_of_box, (const long long*)gt_base, dscores);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

// ------------------------------------------------------------------------------------------------
// Greedy non-maximum suppression of score-sorted 3-D boxes (ndsis/utils/bbox.py:713-759
// non_maximum_supression, IoU as bbox.py:205-242 / 598-620).  The reference sweeps the columns of an N x N threshold
// matrix with N tiny kernels; here one workgroup owns a scene: thread i keeps box i in registers, step j broadcasts
// box j and -- if j is still alive -- every later box tests its IoU against it.  keep[j] is final when step j starts
// (only earlier boxes can clear it), so one barrier per step is enough.  Arithmetic uses the non-contracted
// round-to-nearest intrinsics in the reference's operation order: the comparison `overlap > threshold` is bit-exact.
// ------------------------------------------------------------------------------------------------
static constexpr int NMS_THREADS = 1024;
static constexpr int NMS_MAX_PER_THREAD = 8;

__device__ __forceinline__ float nms_volume(const float* b) {          // size.prod(-1), left to right
    return __fmul_rn(__fmul_rn(__fsub_rn(b[3], b[0]), __fsub_rn(b[4], b[1])), __fsub_rn(b[5], b[2]));
}

__global__ __launch_bounds__(NMS_THREADS) void k_nms(const float* __restrict__ boxes, int n, float thr,
                                                     unsigned char* __restrict__ keep_out) {
    __shared__ float cur[8];             // box j: start xyz, stop xyz, volume, alive
    extern __shared__ unsigned char alive[];                 // [n]
    const float* B = boxes + (long long)blockIdx.x * n * 6;
    unsigned char* K = keep_out + (long long)blockIdx.x * n;
    float mine[NMS_MAX_PER_THREAD][7];
    for (int q = 0; q < NMS_MAX_PER_THREAD; ++q) {
        const int i = threadIdx.x + q * NMS_THREADS;
        if (i < n) {
            for (int d = 0; d < 6; ++d) mine[q][d] = B[i * 6 + d];
            mine[q][6] = nms_volume(mine[q]);
            alive[i] = 1;
        }
    }
    __syncthreads();
    for (int j = 0; j < n; ++j) {
        const int owner = j % NMS_THREADS, oq = j / NMS_THREADS;
        if ((int)threadIdx.x == owner) {
#pragma unroll
            for (int q = 0; q < NMS_MAX_PER_THREAD; ++q)
                if (q == oq) {
                    for (int d = 0; d < 7; ++d) cur[d] = mine[q][d];
                    cur[7] = alive[j] ? 1.f : 0.f;
                }
        }
        __syncthreads();
        if (cur[7] != 0.f) {
#pragma unroll
            for (int q = 0; q < NMS_MAX_PER_THREAD; ++q) {
                const int i = threadIdx.x + q * NMS_THREADS;
                if (i > j && i < n && alive[i]) {
                    float inter = 1.f;                       // prod over the dims of clamp(min_end - max_start, 0)
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        const float lo = fmaxf(cur[d], mine[q][d]), hi = fminf(cur[3 + d], mine[q][3 + d]);
                        const float e = fmaxf(__fsub_rn(hi, lo), 0.f);
                        inter = d == 0 ? e : __fmul_rn(inter, e);
                    }
                    const float uni = __fsub_rn(__fadd_rn(cur[6], mine[q][6]), inter);
                    if (__fdiv_rn(inter, uni) > thr) alive[i] = 0;       // NaN (0/0) compares false, as in torch
                }
            }
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < n; i += NMS_THREADS) K[i] = alive[i];
}

extern "C" int scn_nms(const float* boxes, int batch, int n, float overlap_threshold, uint8_t* keep,
                       scn_stream_t stream) {
    SCN_REQUIRE(batch >= 0 && n >= 0 && n <= NMS_THREADS * NMS_MAX_PER_THREAD);
    if (batch == 0 || n == 0) return SCN_OK;
    SCN_REQUIRE(boxes && keep);
    hipLaunchKernelGGL(k_nms, dim3(batch), dim3(NMS_THREADS), (size_t)n, S(stream), boxes, n, overlap_threshold, keep);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

// ---- round 5: the same greedy NMS as a bit matrix + one serial scan (scn_nms_bits).  k_nms above walks the n boxes with two
// workgroup barriers per box: 0.60 ms for ONE scene of 1024 boxes (profiles/r5_kernel_stats_cfg3rpn_bf16.csv) -- 7 % of a
// detection + mask step.  Here (a) k_nms_matrix: every (suppressor j, 64 candidates i > j) word of the upper triangle is one
// thread's 64 IoU tests, one wave per 64 x 64 block, spread over the chip (same non-contracted arithmetic, same operation
// order: the comparison `overlap > threshold` is bit-exact and symmetric in (i, j)); (b) k_nms_resolve: one workgroup per
// scene stages 256 rows of the matrix into LDS at a time and ONE wave walks them -- lane l holds word l of the `removed`
// set; box j is alive iff its bit is clear, and then its row is OR-ed in.  keep[] equals k_nms's bit for bit (tests).
static constexpr size_t NMSB_LDS_BYTES = 128 * 1024;   // matrix rows staged per round: what fits here
static constexpr int NMSB_MAX_N = 4096;    // 64 words: one per lane

__global__ __launch_bounds__(64) void k_nms_matrix(const float* __restrict__ boxes, int n, float thr,
                                                   unsigned long long* __restrict__ M) {
    const int cb = blockIdx.x, rb = blockIdx.y, nw = (n + 63) >> 6;
    if (cb < rb) return;                                     // lower triangle: never read
    const float* B = boxes + (long long)blockIdx.z * n * 6;
    unsigned long long* Ms = M + (long long)blockIdx.z * n * nw;
    __shared__ float cand[64][7];
    const int t = threadIdx.x;
    {
        const int i = cb * 64 + t;
        if (i < n) {
            float b[6];
            for (int d = 0; d < 6; ++d) b[d] = B[i * 6 + d];
            for (int d = 0; d < 6; ++d) cand[t][d] = b[d];
            cand[t][6] = nms_volume(b);
        }
    }
    __syncthreads();
    const int j = rb * 64 + t;
    if (j >= n) return;
    float cur[7];
    for (int d = 0; d < 6; ++d) cur[d] = B[j * 6 + d];
    cur[6] = nms_volume(cur);
    unsigned long long bits = 0;
    const int lim = min(64, n - cb * 64);
    for (int b = 0; b < lim; ++b) {
        const int i = cb * 64 + b;
        if (i <= j) continue;
        float inter = 1.f;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float lo = fmaxf(cur[d], cand[b][d]), hi = fminf(cur[3 + d], cand[b][3 + d]);
            const float e = fmaxf(__fsub_rn(hi, lo), 0.f);
            inter = d == 0 ? e : __fmul_rn(inter, e);
        }
        const float uni = __fsub_rn(__fadd_rn(cur[6], cand[b][6]), inter);
        if (__fdiv_rn(inter, uni) > thr) bits |= 1ull << b;          // NaN (0/0) compares false, as in torch
    }
    Ms[(long long)j * nw + cb] = bits;
}

__global__ __launch_bounds__(1024) void k_nms_resolve(const unsigned long long* __restrict__ M, int n, int rows_per_round,
                                                     unsigned char* __restrict__ keep_out) {
    extern __shared__ unsigned long long rows[];              // [rows_per_round][nw]
    const int nw = (n + 63) >> 6, lane = threadIdx.x & 63;
    const unsigned long long* Ms = M + (long long)blockIdx.x * n * nw;
    unsigned char* K = keep_out + (long long)blockIdx.x * n;
    unsigned long long removed = 0;                           // wave 0, lane l: word l
    for (int r0 = 0; r0 < n; r0 += rows_per_round) {
        const int nr = min(rows_per_round, n - r0);
        for (int e0 = threadIdx.x; e0 < nr * nw; e0 += 4 * blockDim.x) {          // four independent loads in flight per thread
            unsigned long long v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = e0 + q * blockDim.x;
                const int r = e / nw, w = e - r * nw;
                const bool want = e < nr * nw && w >= ((r0 + r) >> 6);                      // (lower triangle: not written)
                v[q] = Ms[want ? (long long)(r0 + r) * nw + w : (long long)r0 * nw + nw - 1];
                v[q] = want ? v[q] : 0ull;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = e0 + q * blockDim.x;
                if (e < nr * nw) rows[e] = v[q];
            }
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            // A block of 64 rows only tests ONE word of the removed set (word w = j >> 6), and that word of the block's 64 rows is
            // ONE LDS read (lane l <-> row l of the block).  The serial chain then runs on scalars, 64 unrolled steps of
            // (read lane u -- a constant lane --, test bit u, conditional OR); afterwards the rows of the surviving boxes
            // are OR-ed into the full-width set (lane l = word l), sixteen independent LDS reads in flight at a time.
            // (The first forms re-read a vector per box inside the chain: ~190 cycles per box, 69 us for 1024 boxes.  Now, by
            //  leaving a part out: chain 16 us, row ORs 21 us, staging + the rest ~8 us.)
            for (int rb = 0; rb < nr; rb += 64) {
                const int w = (r0 + rb) >> 6;
                const int nb = min(64, nr - rb);
                const unsigned long long rw = lane < nb ? rows[(rb + lane) * nw + w] : 0ull;
                unsigned cur_lo = __builtin_amdgcn_readlane((unsigned)removed, w);
                unsigned cur_hi = __builtin_amdgcn_readlane((unsigned)(removed >> 32), w);
#pragma unroll
                for (int u = 0; u < 32; ++u) {                                   // rows 0..31 of the block: bit u of the low half
                    const unsigned lo = __builtin_amdgcn_readlane((unsigned)rw, u);
                    const unsigned hi = __builtin_amdgcn_readlane((unsigned)(rw >> 32), u);
                    const bool alive = !((cur_lo >> u) & 1u);
                    cur_lo |= alive ? lo : 0u;
                    cur_hi |= alive ? hi : 0u;
                }
#pragma unroll
                for (int u = 0; u < 32; ++u) {                                   // rows 32..63: the upper triangle has no low bits
                    const unsigned hi = __builtin_amdgcn_readlane((unsigned)(rw >> 32), 32 + u);
                    const bool alive = !((cur_hi >> u) & 1u);
                    cur_hi |= alive ? hi : 0u;
                }
                const unsigned long long cur = ((unsigned long long)cur_hi << 32) | cur_lo;
                const unsigned long long kept = ~cur & (nb == 64 ? ~0ull : ((1ull << nb) - 1ull));     // wave-uniform
                for (int r = 0; r < nb; r += 16) {
                    unsigned long long row[16];
                    // (unconditional reads at clamped addresses, the selection afterwards: a predicated read is a branch per row)
                    const unsigned long long* src = rows + (rb + r) * nw + min(lane, nw - 1);
                    const int last = (nb - 1 - r) * nw;
#pragma unroll
                    for (int u = 0; u < 16; ++u) row[u] = src[min(u * nw, last)];
                    unsigned long long acc[4] = {0ull, 0ull, 0ull, 0ull};          // (four chains: one wave, nothing hides a dependency)
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const unsigned long long m = 0ull - ((kept >> (r + u)) & 1ull);      // scalar: all ones for a surviving row
                        acc[u & 3] = (row[u] & m) | acc[u & 3];
                    }
                    removed |= (acc[0] | acc[1]) | (acc[2] | acc[3]);
                }
            }
        }
        __syncthreads();
    }
    // keep[] bytes: wave 0 parks the removed set in LDS, the workgroup writes one byte per thread and step
    if (threadIdx.x < 64 && lane < nw) rows[lane] = removed;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) K[i] = (unsigned char)(((rows[i >> 6] >> (i & 63)) & 1ull) ? 0 : 1);
}

extern "C" int64_t scn_nms_scratch_bytes(int batch, int n) {
    if (batch <= 0 || n <= 0) return 0;
    return (int64_t)batch * n * ((n + 63) / 64) * 8;
}

extern "C" int scn_nms_bits(const float* boxes, int batch, int n, float overlap_threshold, uint8_t* keep, void* scratch,
                            scn_stream_t stream) {
    SCN_REQUIRE(batch >= 0 && n >= 0 && n <= NMSB_MAX_N);
    if (batch == 0 || n == 0) return SCN_OK;
    SCN_REQUIRE(boxes && keep && scratch);
    const int nw = (n + 63) / 64;
    hipLaunchKernelGGL(k_nms_matrix, dim3(nw, nw, batch), dim3(64), 0, S(stream), boxes, n, overlap_threshold,
                       (unsigned long long*)scratch);
    SCN_LAUNCH_CHECK();
    // rows staged per round: as many as 128 KB of LDS hold (a multiple of 64: the walk goes by blocks of 64 rows) -- 1024
    // boxes are ONE round (the four rounds of 256 rows cost a load round trip + two barriers each)
    int rpr = (int)((NMSB_LDS_BYTES / ((size_t)nw * 8)) / 64 * 64);
    rpr = std::max(64, std::min(rpr, (n + 63) / 64 * 64));
    const size_t lds = (size_t)rpr * nw * 8;
    static scn::DeviceOnce attr;
    if (attr.needed()) {
        SCN_HIP(hipFuncSetAttribute((const void*)k_nms_resolve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)NMSB_LDS_BYTES));
        attr.done();
    }
    hipLaunchKernelGGL(k_nms_resolve, dim3(batch), dim3(1024), lds, S(stream), (const unsigned long long*)scratch, n, rpr, keep);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

// ------------------------------------------------------------------------------------------------
// Voxelisation front-end on the device (ndsis/data/sparse_augmentation.py:81-126 augment_coords,
// :42-47 fix_cut_out, :50-78 random_cut_out; ndsis/data/data.py:95-98 collate).  The random numbers of the reference
// (distortion matrix, sub-pixel offset, cut-out start) are inputs; the deterministic core is:
//   aug = points @ R;  shift = -min(aug) + offset;  discrete = trunc(aug + shift);  keep rows inside the cut-out.
// `points @ R` is evaluated as fma(z, R2j, fma(y, R1j, x * R0j)) -- the association torch's CPU matmul uses for K = 3
// (checked on 10^6 points: identical bits), so the truncation sees the same fp32 values as the reference.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_vox_project(const float* __restrict__ P, long long n, float r00, float r01,
                                                     float r02, float r10, float r11, float r12, float r20, float r21,
                                                     float r22, float* __restrict__ aug, float* __restrict__ part) {
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float x = P[3 * i], y = P[3 * i + 1], z = P[3 * i + 2];
        const float a[3] = {fmaf(z, r20, fmaf(y, r10, __fmul_rn(x, r00))), fmaf(z, r21, fmaf(y, r11, __fmul_rn(x, r01))),
                            fmaf(z, r22, fmaf(y, r12, __fmul_rn(x, r02)))};
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            aug[3 * i + d] = a[d];
            mn[d] = fminf(mn[d], a[d]);
            mx[d] = fmaxf(mx[d], a[d]);
        }
    }
    __shared__ float red[4][6];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            mn[d] = fminf(mn[d], __shfl_xor(mn[d], o));
            mx[d] = fmaxf(mx[d], __shfl_xor(mx[d], o));
        }
    }
    if ((threadIdx.x & 63) == 0)
        for (int d = 0; d < 3; ++d) { red[threadIdx.x >> 6][d] = mn[d]; red[threadIdx.x >> 6][3 + d] = mx[d]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int d = threadIdx.x;
        float v = red[0][d];
        for (int w = 1; w < 4; ++w) v = d < 3 ? fminf(v, red[w][d]) : fmaxf(v, red[w][d]);
        part[blockIdx.x * 6 + d] = v;
    }
}

// shift[d] = -min[d] + offset[d];  out[3..5] = max(aug)[d]  (one block)
__global__ void k_vox_shift(const float* __restrict__ part, int blocks, float o0, float o1, float o2,
                            float* __restrict__ shift_max) {
    const int d = threadIdx.x;
    if (d >= 6) return;
    float v = part[d];
    for (int b = 1; b < blocks; ++b) v = d < 3 ? fminf(v, part[b * 6 + d]) : fmaxf(v, part[b * 6 + d]);
    const float off = d == 0 ? o0 : (d == 1 ? o1 : o2);
    shift_max[d] = d < 3 ? __fadd_rn(-v, off) : v;
}

extern "C" int64_t scn_vox_scratch_bytes(int64_t n) { return (int64_t)scn::ew_grid(n, 256) * 6 * sizeof(float) + 256; }

extern "C" int scn_vox_project(const float* points, int64_t n, const float* rot_and_scale_host,
                               const float* offset_host, float* aug, float* shift_max, void* scratch,
                               scn_stream_t stream) {
    SCN_REQUIRE(n >= 1 && points && rot_and_scale_host && offset_host && aug && shift_max && scratch);
    const float* r = rot_and_scale_host;
    const int blocks = scn::ew_grid(n, 256);
    hipLaunchKernelGGL(k_vox_project, dim3(blocks), dim3(256), 0, S(stream), points, (long long)n, r[0], r[1], r[2], r[3],
                       r[4], r[5], r[6], r[7], r[8], aug, (float*)scratch);
    SCN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_vox_shift, dim3(1), dim3(64), 0, S(stream), (const float*)scratch, blocks, offset_host[0],
                       offset_host[1], offset_host[2], shift_max);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

// discrete = trunc(aug + shift) (torch's .long() of a float);  table[i] = i if 0 <= discrete - test_start < size on every
// axis else -1 (size_ok = 0: everything is inside)
__global__ void k_vox_discretize(const float* __restrict__ aug, long long n, const float* __restrict__ shift,
                                 int t0, int t1, int t2, int s0, int s1, int s2, int size_ok,
                                 int* __restrict__ discrete, int* __restrict__ table) {
    const float sh[3] = {shift[0], shift[1], shift[2]};
    const int ts[3] = {t0, t1, t2}, sz[3] = {s0, s1, s2};
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        bool inside = true;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int v = (int)__fadd_rn(aug[3 * i + d], sh[d]);
            discrete[3 * i + d] = v;
            if (size_ok) inside = inside && v - ts[d] >= 0 && v - ts[d] < sz[d];
        }
        table[i] = inside ? (int)i : -1;
    }
}

extern "C" int scn_vox_discretize(const float* aug, int64_t n, const float* shift, const int32_t* test_start_host,
                                  const int32_t* size_host, int32_t* discrete, int32_t* table, scn_stream_t stream) {
    SCN_REQUIRE(n >= 1 && n < 2147483647LL && aug && shift && discrete && table);
    SCN_REQUIRE((test_start_host == nullptr) == (size_host == nullptr));
    const int32_t zero[3] = {0, 0, 0};
    const int32_t* t = test_start_host ? test_start_host : zero;
    const int32_t* z = size_host ? size_host : zero;
    hipLaunchKernelGGL(k_vox_discretize, dim3(scn::ew_grid(n, 256)), dim3(256), 0, S(stream), aug, (long long)n, shift,
                       t[0], t[1], t[2], z[0], z[1], z[2], size_host ? 1 : 0, discrete, table);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

// out[j] = (discrete[rows[j]] - start, batch_index)  as int64 [m][4]  (data.py:95-98: the batch index is the 4th column)
__global__ void k_vox_gather(const int* __restrict__ discrete, const int* __restrict__ rows, long long m, int s0, int s1,
                             int s2, long long batch_index, long long* __restrict__ out) {
    for (long long j = blockIdx.x * (long long)blockDim.x + threadIdx.x; j < m; j += (long long)gridDim.x * blockDim.x) {
        const long long r = rows[j];
        out[4 * j + 0] = discrete[3 * r + 0] - s0;
        out[4 * j + 1] = discrete[3 * r + 1] - s1;
        out[4 * j + 2] = discrete[3 * r + 2] - s2;
        out[4 * j + 3] = batch_index;
    }
}

extern "C" int scn_vox_gather(const int32_t* discrete, const int32_t* rows, int64_t m, const int32_t* start_host,
                              int64_t batch_index, int64_t* out, scn_stream_t stream) {
    SCN_REQUIRE(m >= 0 && start_host);
    if (m == 0) return SCN_OK;
    SCN_REQUIRE(discrete && rows && out);
    hipLaunchKernelGGL(k_vox_gather, dim3(scn::ew_grid(m, 256)), dim3(256), 0, S(stream), discrete, rows, (long long)m,
                       start_host[0], start_host[1], start_host[2], (long long)batch_index, (long long*)out);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

// ---- channel padding of parameters, many tensors per launch ------------------------------------------------------------------
// The mask network's 23-channel level runs on slabs padded to 24 columns (16-byte rows: the vector kernels, bf16 storage), so
// its six layers see zero-padded copies of their logical [fv][nIn][nOut] weights and [nOut] biases -- through
// torch.nn.functional.pad that was a fill + a copy per tensor forward and a slice copy backward: ~50 launches of ~4.5 us per
// detection + mask step for 0.3 MB.  One launch pads all of them (forward), one slices all gradients back (backward).
// A tensor is [fv][rows][cols]; up to two row segments of the source map to row ranges of the destination (the two joined parts
// of a NetworkInNetwork over a JoinTable are padded separately); everything else of the destination is zero.
#define SCN_PAD_MAX 40
struct PadJob { const float* src; float* dst; int fv, src_rows, src_cols, dst_rows, dst_cols; int seg[6]; };   // seg: (src row, count, dst row) x 2
struct PadJobs { PadJob job[SCN_PAD_MAX]; int start[SCN_PAD_MAX + 1]; int n; int backward; };

__global__ __launch_bounds__(256) void k_pad_many(PadJobs jobs) {
    int j = 0;
    while (j + 1 < jobs.n && (int)blockIdx.x >= jobs.start[j + 1]) ++j;              // (block-uniform: scalar)
    const PadJob& jb = jobs.job[j];
    const int e = ((int)blockIdx.x - jobs.start[j]) * 256 + (int)threadIdx.x;
    if (!jobs.backward) {                                            // dst <- pad(src)
        if (e >= jb.fv * jb.dst_rows * jb.dst_cols) return;
        const int c = e % jb.dst_cols, r = (e / jb.dst_cols) % jb.dst_rows, o = e / (jb.dst_cols * jb.dst_rows);
        int sr = -1;
        if (r >= jb.seg[2] && r < jb.seg[2] + jb.seg[1]) sr = jb.seg[0] + r - jb.seg[2];
        else if (r >= jb.seg[5] && r < jb.seg[5] + jb.seg[4]) sr = jb.seg[3] + r - jb.seg[5];
        jb.dst[e] = (sr >= 0 && c < jb.src_cols) ? jb.src[((long long)o * jb.src_rows + sr) * jb.src_cols + c] : 0.f;
    } else {                                                         // src-shaped gradient <- slice(dst-shaped gradient)
        if (e >= jb.fv * jb.src_rows * jb.src_cols) return;
        const int c = e % jb.src_cols, r = (e / jb.src_cols) % jb.src_rows, o = e / (jb.src_cols * jb.src_rows);
        int dr = -1;
        if (r >= jb.seg[0] && r < jb.seg[0] + jb.seg[1]) dr = jb.seg[2] + r - jb.seg[0];
        else if (r >= jb.seg[3] && r < jb.seg[3] + jb.seg[4]) dr = jb.seg[5] + r - jb.seg[3];
        // (here `src` is the padded gradient -- NULL: no gradient arrived, zeros -- and `dst` the logical one)
        jb.dst[e] = (dr >= 0 && jb.src) ? jb.src[((long long)o * jb.dst_rows + dr) * jb.dst_cols + c] : 0.f;
    }
}

extern "C" int scn_pad_params_many(int n, const float* const* in_host, float* const* out_host, const int32_t* desc_host,
                                   int backward, scn_stream_t stream) {
    SCN_REQUIRE(n >= 0 && (n == 0 || (in_host && out_host && desc_host)));
    for (int base = 0; base < n; base += SCN_PAD_MAX) {
        PadJobs jobs;
        jobs.n = n - base < SCN_PAD_MAX ? n - base : SCN_PAD_MAX;
        jobs.backward = backward ? 1 : 0;
        jobs.start[0] = 0;
        for (int j = 0; j < jobs.n; ++j) {
            const int32_t* d = desc_host + (size_t)(base + j) * 11;
            PadJob& jb = jobs.job[j];
            jb.src = in_host[base + j]; jb.dst = out_host[base + j];
            jb.fv = d[0]; jb.src_rows = d[1]; jb.src_cols = d[2]; jb.dst_rows = d[3]; jb.dst_cols = d[4];
            for (int k = 0; k < 6; ++k) jb.seg[k] = d[5 + k];
            SCN_REQUIRE(jb.dst && (jb.src || backward));
            SCN_REQUIRE(jb.fv >= 1 && jb.src_rows >= 1 && jb.src_cols >= 1 && jb.dst_rows >= jb.src_rows && jb.dst_cols >= jb.src_cols);
            SCN_REQUIRE((int64_t)jb.fv * jb.dst_rows * jb.dst_cols < (1ll << 30));
            // the segments lie inside both tensors and do not overlap in the destination
            SCN_REQUIRE(jb.seg[1] >= 0 && jb.seg[4] >= 0 && jb.seg[0] >= 0 && jb.seg[3] >= 0 && jb.seg[2] >= 0 && jb.seg[5] >= 0);
            SCN_REQUIRE(jb.seg[0] + jb.seg[1] <= jb.src_rows && jb.seg[3] + jb.seg[4] <= jb.src_rows);
            SCN_REQUIRE(jb.seg[2] + jb.seg[1] <= jb.dst_rows && jb.seg[5] + jb.seg[4] <= jb.dst_rows);
            SCN_REQUIRE(jb.seg[4] == 0 || jb.seg[5] >= jb.seg[2] + jb.seg[1] || jb.seg[2] >= jb.seg[5] + jb.seg[4]);
            const int64_t elems = backward ? (int64_t)jb.fv * jb.src_rows * jb.src_cols : (int64_t)jb.fv * jb.dst_rows * jb.dst_cols;
            jobs.start[j + 1] = jobs.start[j] + (int)((elems + 255) / 256);
        }
        hipLaunchKernelGGL(k_pad_many, dim3((unsigned)jobs.start[jobs.n]), dim3(256), 0, S(stream), jobs);
        SCN_LAUNCH_CHECK();
    }
    return SCN_OK;
}
