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

// ------------------------------------------------------------------------------------------------
// Dilation gather (round 5; rpn.DenseRpn): the first dense 3^3 same-convolution behind scn.SparseToDense
// (module_factory.py:581-611 `get_dilation_network`) reads a volume in which only the N active sites of the sparse level are
// non-zero (2.3 % at the stride-8 level of the cfg-2 scene).  conv3d(pad 1):  h[c] = b + sum_o x[c + d_o] . W[o],  o = (a*3+b)*3+c',
// d_o = (a-1, b-1, c'-1): a cell sums over its ACTIVE neighbours only.  So the layer is ONE row GEMM on the active rows,
//   P[r][o][:] = X[r] . W[o]        ([N, 27 * Cout]; 1.3 GFLOP instead of the volume's 58),
// followed by this gather:  h[c] = b + sum_o P[map[c + d_o]][o][:]  in ascending o (deterministic), map = cell -> active row | -1.
// Backward: dP[r][o][:] = dh[cell(r) - d_o] (a pure gather), db = column sums of dh (scn_colsum), dX / dW through the GEMM.
// T = float (fp32 slabs) or unsigned short (bf16-stored P / h; fp32 accumulation, one round-to-nearest-even per output).
// ------------------------------------------------------------------------------------------------
template <typename TT_> __device__ __forceinline__ float dg_ld(const TT_* p);
template <> __device__ __forceinline__ float dg_ld<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float dg_ld<us>(const us* p) { return bw(*p); }
template <typename TT_> __device__ __forceinline__ void dg_st(TT_* p, float v);
template <> __device__ __forceinline__ void dg_st<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void dg_st<us>(us* p, float v) { *p = bn(v); }

template <typename TT_>
__global__ __launch_bounds__(256) void k_dilate_fwd(const TT_* __restrict__ P, const int* __restrict__ map, int B, int X, int Y,
                                                    int Z, int c, const float* __restrict__ bias, TT_* __restrict__ out) {
    const long long cells = (long long)B * X * Y * Z, total = cells * c;
    for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total; e += (long long)gridDim.x * 256ll) {
        const long long cell = e / c;
        const int ch = (int)(e - cell * c);
        int z = (int)(cell % Z);
        long long t = cell / Z;
        int y = (int)(t % Y);
        t /= Y;
        int x = (int)(t % X);
        const long long b = t / X;
        float acc = bias ? bias[ch] : 0.f;
#pragma unroll 1
        for (int o = 0; o < 27; ++o) {
            const int nx = x + o / 9 - 1, ny = y + (o / 3) % 3 - 1, nz = z + o % 3 - 1;
            if ((unsigned)nx >= (unsigned)X || (unsigned)ny >= (unsigned)Y || (unsigned)nz >= (unsigned)Z) continue;
            const int r = map[((b * X + nx) * Y + ny) * Z + nz];
            if (r >= 0) acc += dg_ld<TT_>(P + ((long long)r * 27 + o) * c + ch);
        }
        dg_st<TT_>(out + e, acc);
    }
}

// the same, four channels per thread (c % 4 == 0, fewer than 2^31 cells): the 27 map entries of the thread's cell are requested
// together (independent loads, 32-bit index arithmetic), then the hits -- most cells of a 2 % occupied volume have none -- are
// accumulated in ascending offset order
template <typename TT_>
__global__ __launch_bounds__(256) void k_dilate_fwd4(const TT_* __restrict__ P, const int* __restrict__ map, int B, int X, int Y,
                                                     int Z, int c, const float* __restrict__ bias, TT_* __restrict__ out) {
    const int c4 = c >> 2;
    const long long total = (long long)B * X * Y * Z * c4;
    for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total; e += (long long)gridDim.x * 256ll) {
        const int cell = (int)(e / c4), ch = ((int)(e - (long long)cell * c4)) << 2;
        const int z = cell % Z, t1 = cell / Z, y = t1 % Y, t2 = t1 / Y, x = t2 % X, b = t2 / X;
        int r[27];
#pragma unroll
        for (int o = 0; o < 27; ++o) {
            const int nx = x + o / 9 - 1, ny = y + (o / 3) % 3 - 1, nz = z + o % 3 - 1;
            const bool in = (unsigned)nx < (unsigned)X && (unsigned)ny < (unsigned)Y && (unsigned)nz < (unsigned)Z;
            r[o] = map[in ? ((b * X + nx) * Y + ny) * Z + nz : cell];          // (unconditional load; masked below)
            r[o] = in ? r[o] : -1;
        }
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        if (bias) { a0 = bias[ch]; a1 = bias[ch + 1]; a2 = bias[ch + 2]; a3 = bias[ch + 3]; }
#pragma unroll
        for (int o = 0; o < 27; ++o) {
            if (r[o] >= 0) {
                const TT_* q = P + ((long long)r[o] * 27 + o) * c + ch;
                a0 += dg_ld<TT_>(q); a1 += dg_ld<TT_>(q + 1); a2 += dg_ld<TT_>(q + 2); a3 += dg_ld<TT_>(q + 3);
            }
        }
        TT_* w = out + (long long)cell * c + ch;
        dg_st<TT_>(w, a0); dg_st<TT_>(w + 1, a1); dg_st<TT_>(w + 2, a2); dg_st<TT_>(w + 3, a3);
    }
}

template <typename TT_>
__global__ __launch_bounds__(256) void k_dilate_bwd(const TT_* __restrict__ dOut, const long long* __restrict__ cell_of_row,
                                                    long long n, int X, int Y, int Z, int c, TT_* __restrict__ dP) {
    const long long total = n * 27 * c;
    for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total; e += (long long)gridDim.x * 256ll) {
        const int ch = (int)(e % c);
        const long long ro = e / c;
        const int o = (int)(ro % 27);
        const long long r = ro / 27;
        const long long cell = cell_of_row[r];
        int z = (int)(cell % Z);
        long long t = cell / Z;
        int y = (int)(t % Y);
        t /= Y;
        int x = (int)(t % X);
        const long long b = t / X;
        // row r feeds cell c' through offset o iff c' + d_o = cell(r)
        const int cx = x - (o / 9 - 1), cy = y - ((o / 3) % 3 - 1), cz = z - (o % 3 - 1);
        float v = 0.f;
        if ((unsigned)cx < (unsigned)X && (unsigned)cy < (unsigned)Y && (unsigned)cz < (unsigned)Z)
            v = dg_ld<TT_>(dOut + (((b * X + cx) * Y + cy) * Z + cz) * c + ch);
        dg_st<TT_>(dP + e, v);
    }
}

// cell of every active row of a sparse level in the channels-last volume [B X Y Z] and the inverse map (-1 on empty cells):
// what SparseToDense's scatter and the dilation gather above index with -- one memset + one launch for torch's seven.
__global__ void k_cell_map(const int4* __restrict__ coords, long long n, int X, int Y, int Z, int batch,
                           long long* __restrict__ cell_of_row, int* __restrict__ map, int* __restrict__ bad) {
    for (long long r = blockIdx.x * (long long)blockDim.x + threadIdx.x; r < n; r += (long long)gridDim.x * blockDim.x) {
        const int4 c = coords[r];                                         // (x, y, z, sample)
        if ((unsigned)c.x >= (unsigned)X || (unsigned)c.y >= (unsigned)Y || (unsigned)c.z >= (unsigned)Z || (unsigned)c.w >= (unsigned)batch) {
            atomicAdd(bad, 1);
            cell_of_row[r] = 0;
            continue;
        }
        const long long cell = (((long long)c.w * X + c.x) * Y + c.y) * Z + c.z;
        cell_of_row[r] = cell;
        map[cell] = (int)r;
    }
}

extern "C" int scn_cell_map(const int32_t* coords, int64_t n, int batch, const int64_t* size3_host, int64_t* cell_of_row,
                            int32_t* map, int32_t* n_outside_dev, scn_stream_t stream) {
    SCN_REQUIRE(batch >= 0 && n >= 0 && size3_host && n_outside_dev);
    const int X = (int)size3_host[0], Y = (int)size3_host[1], Z = (int)size3_host[2];
    const int64_t cells = (int64_t)batch * X * Y * Z;
    SCN_REQUIRE(X > 0 && Y > 0 && Z > 0 && cells < (1ll << 31) && n < (1ll << 31));
    SCN_HIP(hipMemsetAsync(n_outside_dev, 0, sizeof(int32_t), S(stream)));
    if (cells == 0) return SCN_OK;
    SCN_REQUIRE(map && (n == 0 || (coords && cell_of_row)));
    SCN_HIP(hipMemsetAsync(map, 0xFF, sizeof(int32_t) * (size_t)cells, S(stream)));
    if (n == 0) return SCN_OK;
    hipLaunchKernelGGL(k_cell_map, dim3(scn::ew_grid(n, 256)), dim3(256), 0, S(stream), (const int4*)coords, (long long)n, X, Y, Z,
                       batch, (long long*)cell_of_row, map, n_outside_dev);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

extern "C" int scn_dilate_gather_fwd(const void* P, const int32_t* map, int batch, const int64_t* size3_host, int c, int bf16,
                                     const float* bias, void* out, scn_stream_t stream) {
    SCN_REQUIRE(batch >= 0 && size3_host && c >= 1);
    const int X = (int)size3_host[0], Y = (int)size3_host[1], Z = (int)size3_host[2];
    const int64_t total = (int64_t)batch * X * Y * Z * c;
    if (total == 0) return SCN_OK;
    SCN_REQUIRE(P && map && out && X > 0 && Y > 0 && Z > 0);
    const bool v4 = c % 4 == 0 && (int64_t)batch * X * Y * Z < (1ll << 31);
    const int grid = (int)cdiv(v4 ? total / 4 : total, 256) < 65536 * 16 ? (int)cdiv(v4 ? total / 4 : total, 256) : 65536 * 16;
    if (v4 && bf16)
        hipLaunchKernelGGL(k_dilate_fwd4<us>, dim3(grid), dim3(256), 0, S(stream), (const us*)P, map, batch, X, Y, Z, c, bias, (us*)out);
    else if (v4)
        hipLaunchKernelGGL(k_dilate_fwd4<float>, dim3(grid), dim3(256), 0, S(stream), (const float*)P, map, batch, X, Y, Z, c, bias,
                           (float*)out);
    else if (bf16)
        hipLaunchKernelGGL(k_dilate_fwd<us>, dim3(scn::ew_grid(total, 256)), dim3(256), 0, S(stream), (const us*)P, map, batch, X,
                           Y, Z, c, bias, (us*)out);
    else
        hipLaunchKernelGGL(k_dilate_fwd<float>, dim3(scn::ew_grid(total, 256)), dim3(256), 0, S(stream), (const float*)P, map,
                           batch, X, Y, Z, c, bias, (float*)out);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}

extern "C" int scn_dilate_gather_bwd(const void* dOut, const int64_t* cell_of_row, int64_t n, const int64_t* size3_host, int c,
                                     int bf16, void* dP, scn_stream_t stream) {
    SCN_REQUIRE(n >= 0 && size3_host && c >= 1);
    if (n == 0) return SCN_OK;
    const int X = (int)size3_host[0], Y = (int)size3_host[1], Z = (int)size3_host[2];
    SCN_REQUIRE(dOut && cell_of_row && dP && X > 0 && Y > 0 && Z > 0);
    const int64_t total = n * 27 * c;
    if (bf16)
        hipLaunchKernelGGL(k_dilate_bwd<us>, dim3(scn::ew_grid(total, 256)), dim3(256), 0, S(stream), (const us*)dOut,
                           (const long long*)cell_of_row, (long long)n, X, Y, Z, c, (us*)dP);
    else
        hipLaunchKernelGGL(k_dilate_bwd<float>, dim3(scn::ew_grid(total, 256)), dim3(256), 0, S(stream), (const float*)dOut,
                           (const long long*)cell_of_row, (long long)n, X, Y, Z, c, (float*)dP);
    SCN_LAUNCH_CHECK();
    return SCN_OK;
}
