// bf16-STORAGE forms of the HBM-bound feature kernels (BASELINE configs 3-5): OutputLayer gather / segment sum, the ROI
// feature gather (the same row gather), Max/AveragePooling, SparseToDense, AddTable, and the storage casts.  Slabs are
// uint16 bf16 bit patterns; values are widened exactly, the arithmetic is the fp32 kernels' (scn_elem.hip), and every
// result is rounded to bf16 ONCE (round-to-nearest-even, the rounding of torch's .to(torch.bfloat16)).  Half the bytes of
// the fp32 forms: these kernels are pure streaming / row-gather passes, bound by HBM.
#include "scn_common.h"

using scn::S;
using scn::cdiv;

namespace {

__device__ __forceinline__ float bw(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ unsigned short bn(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }

typedef unsigned short us;

// ---- casts ------------------------------------------------------------------------------------------------------
__global__ void k_cast_f2b(const float* __restrict__ x, long long count, us* __restrict__ y, int vec) {
    const long long tid = blockIdx.x * (long long)blockDim.x + threadIdx.x, stride = (long long)gridDim.x * blockDim.x;
    if (vec) {
        const long long n4 = count >> 2;
        for (long long i = tid; i < n4; i += stride) {
            const float4 v = ((const float4*)x)[i];
            uint2 o;
            o.x = (unsigned)bn(v.x) | ((unsigned)bn(v.y) << 16);
            o.y = (unsigned)bn(v.z) | ((unsigned)bn(v.w) << 16);
            ((uint2*)y)[i] = o;
        }
        for (long long i = (n4 << 2) + tid; i < count; i += stride) y[i] = bn(x[i]);
    } else {
        for (long long i = tid; i < count; i += stride) y[i] = bn(x[i]);
    }
}

__global__ void k_cast_b2f(const us* __restrict__ x, long long count, float* __restrict__ y, int vec) {
    const long long tid = blockIdx.x * (long long)blockDim.x + threadIdx.x, stride = (long long)gridDim.x * blockDim.x;
    if (vec) {
        const long long n4 = count >> 2;
        for (long long i = tid; i < n4; i += stride) {
            const uint2 v = ((const uint2*)x)[i];
            ((float4*)y)[i] = make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u),
                                          __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u));
        }
        for (long long i = (n4 << 2) + tid; i < count; i += stride) y[i] = bw(x[i]);
    } else {
        for (long long i = tid; i < count; i += stride) y[i] = bw(x[i]);
    }
}

__global__ void k_add_b(const us* __restrict__ a, const us* __restrict__ b, long long count, us* __restrict__ y) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < count; i += (long long)gridDim.x * blockDim.x)
        y[i] = bn(bw(a[i]) + bw(b[i]));
}

// ---- row gather (OutputLayer, ROI feature gather): a byte copy of 2c-byte rows ------------------------------------
__global__ void k_gather_rows_b(const us* __restrict__ X, const int* __restrict__ rows, long long m, int c,
                                us* __restrict__ Y, int vec) {
    const long long tid = blockIdx.x * (long long)blockDim.x + threadIdx.x, stride = (long long)gridDim.x * blockDim.x;
    if (vec) {                                        // 16-byte pieces: c % 8 == 0
        const int c8 = c >> 3;
        for (long long i = tid; i < m * c8; i += stride) {
            const long long r = i / c8;
            const int g = (int)(i - r * c8);
            ((uint4*)Y)[i] = ((const uint4*)(X + (long long)rows[r] * c))[g];
        }
    } else {
        for (long long i = tid; i < m * c; i += stride) {
            const long long r = i / c;
            Y[i] = X[(long long)rows[r] * c + (i - r * c)];
        }
    }
}

__global__ void k_scatter_add64_b(const us* __restrict__ V, const int* __restrict__ item_row, long long n_items, int c,
                                  double* __restrict__ acc) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n_items * c;
         i += (long long)gridDim.x * blockDim.x) {
        const long long it = i / c;
        atomicAdd(&acc[(long long)item_row[it] * c + (int)(i - it * c)], (double)bw(V[i]));
    }
}

__global__ void k_finish64_b(const double* __restrict__ acc, long long count, us* __restrict__ Y) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < count; i += (long long)gridDim.x * blockDim.x)
        Y[i] = bn((float)acc[i]);
}

// ---- pooling 2^3 / 2 ----------------------------------------------------------------------------------------------
__global__ void k_pool_fwd_b(const us* __restrict__ X, const int* __restrict__ child, long long n_coarse, int c, int avg,
                             us* __restrict__ Y) {
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
                const float v = bw(X[(long long)f * c + ch]);
                acc = mean ? acc + v : fmaxf(acc, v);
            }
        }
        Y[i] = bn(mean ? acc * (1.0f / (float)n_off) : acc);
    }
}

__global__ void k_pool_bwd_b(const us* __restrict__ X, const us* __restrict__ Y, const us* __restrict__ dY,
                             const int* __restrict__ parent, long long n_fine, int c, int avg, us* __restrict__ dX) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n_fine * c;
         i += (long long)gridDim.x * blockDim.x) {
        const long long f = i / c;
        const long long o = (long long)parent[f] * c + (int)(i - f * c);
        // max: a stored maximum IS one of the stored inputs (a maximum is not rounded), so the bit patterns compare
        const int n_off = (avg >> 8) ? (avg >> 8) : 8;
        dX[i] = (avg & 1) ? bn(bw(dY[o]) * (1.0f / (float)n_off)) : (X[i] == Y[o] ? dY[o] : (us)0);
    }
}

// ---- SparseToDense ------------------------------------------------------------------------------------------------
template <bool BWD>
__global__ void k_s2d_b(const us* __restrict__ src, const int4* __restrict__ coords, long long n, int c, long long sx,
                        long long sy, long long sz, us* __restrict__ dst) {
    const long long vol = sx * sy * sz;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n * c; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i % n;
        const int ch = (int)(i / n);
        const int4 p = coords[r];
        const long long d = ((long long)p.w * c + ch) * vol + ((long long)p.x * sy + p.y) * sz + p.z;
        if (BWD) dst[r * c + ch] = src[d];
        else dst[d] = src[r * c + ch];
    }
}

}  // namespace

extern "C" int scn_cast_f32_to_bf16(const float* X, int64_t count, uint16_t* Y, scn_stream_t stream) {
    SCN_REQUIRE(count >= 0);
    if (count == 0) return SCN_OK;
    SCN_REQUIRE(X && Y);
    const int vec = (((uintptr_t)X & 15) == 0) && (((uintptr_t)Y & 7) == 0);
    hipLaunchKernelGGL(k_cast_f2b, dim3(scn::ew_grid(vec ? count / 4 + 1 : count, 256)), dim3(256), 0, S(stream), X,
                       (long long)count, (us*)Y, vec);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

extern "C" int scn_cast_bf16_to_f32(const uint16_t* X, int64_t count, float* Y, scn_stream_t stream) {
    SCN_REQUIRE(count >= 0);
    if (count == 0) return SCN_OK;
    SCN_REQUIRE(X && Y);
    const int vec = (((uintptr_t)Y & 15) == 0) && (((uintptr_t)X & 7) == 0);
    hipLaunchKernelGGL(k_cast_b2f, dim3(scn::ew_grid(vec ? count / 4 + 1 : count, 256)), dim3(256), 0, S(stream), (const us*)X,
                       (long long)count, Y, vec);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

extern "C" int scn_add_bf16(const uint16_t* A, const uint16_t* B, int64_t count, uint16_t* Y, scn_stream_t stream) {
    SCN_REQUIRE(count >= 0);
    if (count == 0) return SCN_OK;
    SCN_REQUIRE(A && B && Y);
    hipLaunchKernelGGL(k_add_b, dim3(scn::ew_grid(count, 256)), dim3(256), 0, S(stream), (const us*)A, (const us*)B,
                       (long long)count, (us*)Y);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

extern "C" int scn_gather_rows_bf16(const uint16_t* X, const int32_t* rows, int64_t m, int c, uint16_t* Y,
                                    scn_stream_t stream) {
    SCN_REQUIRE(m >= 0 && c >= 1);
    if (m == 0) return SCN_OK;
    SCN_REQUIRE(X && rows && Y);
    const int vec = (c % 8 == 0) && ((((uintptr_t)X | (uintptr_t)Y) & 15) == 0);
    hipLaunchKernelGGL(k_gather_rows_b, dim3(scn::ew_grid(m * (vec ? c / 8 : c), 256)), dim3(256), 0, S(stream), (const us*)X,
                       rows, (long long)m, c, (us*)Y, vec);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

extern "C" int scn_segment_sum_bf16(const uint16_t* dY, const int32_t* item_row, int64_t n_items, int64_t n_rows, int c,
                                    uint16_t* dX, double* acc64, scn_stream_t stream) {
    SCN_REQUIRE(n_items >= 0 && n_rows >= 0 && c >= 1);
    if (n_rows == 0) return SCN_OK;
    SCN_REQUIRE(dX && acc64);
    SCN_HIP(hipMemsetAsync(acc64, 0, sizeof(double) * n_rows * c, S(stream)));
    if (n_items) {
        SCN_REQUIRE(dY && item_row);
        hipLaunchKernelGGL(k_scatter_add64_b, dim3(scn::ew_grid(n_items * c, 256)), dim3(256), 0, S(stream), (const us*)dY,
                           item_row, (long long)n_items, c, acc64);
        SCN_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_finish64_b, dim3(scn::ew_grid(n_rows * c, 256)), dim3(256), 0, S(stream), (const double*)acc64,
                       (long long)(n_rows * c), (us*)dX);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

extern "C" int scn_pool_fwd_bf16(const uint16_t* X, const int32_t* child, int64_t n_coarse, int c, int average,
                                 uint16_t* Y, scn_stream_t stream) {
    SCN_REQUIRE(n_coarse >= 0 && c >= 1);
    if (n_coarse == 0) return SCN_OK;
    SCN_REQUIRE(X && child && Y);
    hipLaunchKernelGGL(k_pool_fwd_b, dim3(scn::ew_grid(n_coarse * c, 256)), dim3(256), 0, S(stream), (const us*)X, child,
                       (long long)n_coarse, c, average, (us*)Y);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

extern "C" int scn_pool_bwd_bf16(const uint16_t* X, const uint16_t* Y, const uint16_t* dY, const int32_t* parent,
                                 int64_t n_fine, int c, int average, uint16_t* dX, scn_stream_t stream) {
    SCN_REQUIRE(n_fine >= 0 && c >= 1);
    if (n_fine == 0) return SCN_OK;
    SCN_REQUIRE(X && Y && dY && parent && dX);
    hipLaunchKernelGGL(k_pool_bwd_b, dim3(scn::ew_grid(n_fine * c, 256)), dim3(256), 0, S(stream), (const us*)X, (const us*)Y,
                       (const us*)dY, parent, (long long)n_fine, c, average, (us*)dX);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

extern "C" int scn_sparse_to_dense_fwd_bf16(const uint16_t* X, const int32_t* coords, int64_t n, int c,
                                            const int64_t* size3_host, uint16_t* out, scn_stream_t stream) {
    SCN_REQUIRE(n >= 0 && c >= 1 && size3_host);
    if (n == 0) return SCN_OK;
    SCN_REQUIRE(X && coords && out);
    hipLaunchKernelGGL(k_s2d_b<false>, dim3(scn::ew_grid(n * c, 256)), dim3(256), 0, S(stream), (const us*)X,
                       (const int4*)coords, (long long)n, c, (long long)size3_host[0], (long long)size3_host[1],
                       (long long)size3_host[2], (us*)out);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

extern "C" int scn_sparse_to_dense_bwd_bf16(const uint16_t* dOut, const int32_t* coords, int64_t n, int c,
                                            const int64_t* size3_host, uint16_t* dX, scn_stream_t stream) {
    SCN_REQUIRE(n >= 0 && c >= 1 && size3_host);
    if (n == 0) return SCN_OK;
    SCN_REQUIRE(dOut && coords && dX);
    hipLaunchKernelGGL(k_s2d_b<true>, dim3(scn::ew_grid(n * c, 256)), dim3(256), 0, S(stream), (const us*)dOut,
                       (const int4*)coords, (long long)n, c, (long long)size3_host[0], (long long)size3_host[1],
                       (long long)size3_host[2], (us*)dX);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}
