// CPU restatement of the SparseConvNet CPU path for the benchmark U-Net -- TEST INFRASTRUCTURE / TIMED CPU BASELINE ONLY.
//
// Only tests/, __graft_entry__ and bench.py's cpu_baseline leg may build, load or call this file.  The product
// (sparse_rcnn_amd) never does.  It is NOT the SparseConvNet binary: that library is not in /root/reference, not
// installed and not pinned (parity unpinned); this file restates its published CPU algorithm
// [UPSTREAM-SCN, SURVEY.md §2.1 / Appendix B] for the layer list the reference's factory code builds
// (ndsis/modules/module_factory.py:127-183 residual units, :221-271 down/up-samplers, :513-578 levels;
// custom_container.py:70-83 skip reunite), and is checked against oracle/scn_oracle.py in tests/test_cpu_baseline.py:
//
//   Metadata      per-batch hash map (b,x,y,z) -> row; InputLayer mode 4 rows by first occurrence, mean of duplicates
//   rulebooks     SubM 3^3: per active site 27 probes -> (in,out) pairs per offset; Convolution 2^3/2: coarse sites by
//                 first occurrence, pairs per child offset; Deconvolution re-uses the encoder's rulebook, roles swapped
//   conv-type op  output = bias; for each kernel offset: gather rows -> contiguous buffer -> sgemm -> scatter-add
//                 (backward: dX through W^T on the swapped pairs, dW = gathered X^T . gathered dY, db = column sums)
//   glue          ReLU / AddTable / JoinTable / NetworkInNetwork as separate passes over the slab, as upstream
//
// fp32 throughout (InputLayer mean accumulates in double, as the HIP path and the Python oracle do).  OpenMP over row
// blocks inside an offset (output rows of one offset are distinct, so blocks never collide); offsets ascending, bias
// first -- the summation order of SURVEY Appendix B.  Build: g++ -O3 -march=native -fopenmp -shared -fPIC.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// phase timers (printed when SCN_CPU_PROFILE is set): where a step spends its wall time
double g_t[8];
const char* g_tn[8] = {"input+hash", "rulebooks", "conv_fwd", "conv_bwd", "glue fwd", "glue bwd", "alloc", "other"};
struct Tick {
    int k; double t0;
    explicit Tick(int k_) : k(k_), t0(now()) {}
    ~Tick() { g_t[k] += now() - t0; }
    static double now() {
#ifdef _OPENMP
        return omp_get_wtime();
#else
        return 0.0;
#endif
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// hash map: packed key -> row (open addressing, linear probing)
// ---------------------------------------------------------------------------------------------------------------------
struct HashGrid {
    std::vector<uint64_t> keys;
    std::vector<int32_t> rows;
    uint64_t mask = 0;
    static constexpr uint64_t EMPTY = ~0ull;
    static uint64_t pack(int64_t x, int64_t y, int64_t z, int64_t b) {
        return ((uint64_t)b << 48) | ((uint64_t)x << 32) | ((uint64_t)y << 16) | (uint64_t)z;
    }
    void init(int64_t n) {
        uint64_t cap = 1024;
        while (cap < (uint64_t)(2 * n)) cap <<= 1;
        keys.assign(cap, EMPTY);
        rows.assign(cap, -1);
        mask = cap - 1;
    }
    static uint64_t slot_of(uint64_t key, uint64_t mask) {
        uint64_t h = key * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29;
        return h & mask;
    }
    // returns the row of key, inserting `next` if absent (*inserted tells which)
    int32_t find_or_insert(uint64_t key, int32_t next, bool* inserted) {
        uint64_t s = slot_of(key, mask);
        for (;;) {
            if (keys[s] == key) { *inserted = false; return rows[s]; }
            if (keys[s] == EMPTY) { keys[s] = key; rows[s] = next; *inserted = true; return next; }
            s = (s + 1) & mask;
        }
    }
    int32_t find(uint64_t key) const {
        uint64_t s = slot_of(key, mask);
        for (;;) {
            if (keys[s] == key) return rows[s];
            if (keys[s] == EMPTY) return -1;
            s = (s + 1) & mask;
        }
    }
};

struct Rules {                       // per offset: pairs (in, out), out ascending
    int n_off = 0;
    std::vector<std::vector<int32_t>> in, out;
};

struct Level {
    std::vector<int32_t> coords;     // [n][4] x,y,z,b
    int64_t n = 0;
    HashGrid grid;
    Rules subm;                      // 3^3
    Rules down;                      // 2^3/2 to the next level (in = fine rows, out = coarse rows)
};

// ---------------------------------------------------------------------------------------------------------------------
// sgemm micro kernels (row-major).  C[M x N] (+)= A[M x K] . B[K x N]
// ---------------------------------------------------------------------------------------------------------------------
#if defined(__AVX512F__)
constexpr int VW = 16;
#else
constexpr int VW = 8;
#endif
typedef float vf __attribute__((vector_size(VW * 4)));                              // register type
typedef float vfu __attribute__((vector_size(VW * 4), aligned(4), may_alias));      // unaligned view of memory

static inline vf vload(const float* p) { return *(const vfu*)p; }
static inline void vstore(float* p, vf v) { *(vfu*)p = v; }
static inline vf vsplat(float a) { return a - (vf){}; }

// Register tile of MR rows x NRV vectors of C, accumulated over k.  TN = false: C[m x n] += A[m x k] . B[k x n];
// TN = true: C[k x n] += A^T . B with A[m x k], B[m x n] (the weight gradient of a block of gathered rows) -- the tile
// then covers MR rows of C = MR columns of A and the reduction runs over the m gathered rows.
template <int MR, int NRV, bool TN>
static inline void tile(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int red) {
    vf acc[MR][NRV];
    for (int r = 0; r < MR; ++r)
        for (int v = 0; v < NRV; ++v) acc[r][v] = vload(C + (size_t)r * ldc + v * VW);
    for (int p = 0; p < red; ++p) {
        vf b[NRV];
        for (int v = 0; v < NRV; ++v) b[v] = vload(B + (size_t)p * ldb + v * VW);
        for (int r = 0; r < MR; ++r) {
            const vf a = vsplat(TN ? A[(size_t)p * lda + r] : A[(size_t)r * lda + p]);
            for (int v = 0; v < NRV; ++v) acc[r][v] += a * b[v];
        }
    }
    for (int r = 0; r < MR; ++r)
        for (int v = 0; v < NRV; ++v) vstore(C + (size_t)r * ldc + v * VW, acc[r][v]);
}

// one column panel of NRV vectors: rows of C in tiles of MR, then single rows
template <int MR, int NRV, bool TN>
static inline void panel(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int rows, int red) {
    int i = 0;
    for (; i + MR <= rows; i += MR) tile<MR, NRV, TN>(TN ? A + i : A + (size_t)i * lda, lda, B, ldb, C + (size_t)i * ldc, ldc, red);
    for (; i < rows; ++i) tile<1, NRV, TN>(TN ? A + i : A + (size_t)i * lda, lda, B, ldb, C + (size_t)i * ldc, ldc, red);
}

// C (rows x n) += op(A) . B, row-major; the n columns are cut into panels of 4, 2, 1 vectors and a scalar tail, so that
// 32-channel layers (half a 64-float panel on AVX-512) still run on full register tiles (16 accumulators each)
template <bool TN>
static void gemm_acc(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int rows, int n, int red) {
    int j = 0;
    // tile shapes picked by measurement on an AVX-512 Xeon (tools-free micro-benchmark, one core, GFLOP/s at 64 / 256
    // channels): NN 6x2 vectors 135 / 99 (4x4: 51 / 46, 8x2: 54 / 44); TN 6x4 129 / 113 (4x4: 132 / 91)
    if (TN)
        for (; j + 4 * VW <= n; j += 4 * VW) panel<6, 4, TN>(A, lda, B + j, ldb, C + j, ldc, rows, red);
    for (; j + 2 * VW <= n; j += 2 * VW) panel<6, 2, TN>(A, lda, B + j, ldb, C + j, ldc, rows, red);
    if (j + VW <= n) { panel<12, 1, TN>(A, lda, B + j, ldb, C + j, ldc, rows, red); j += VW; }
    if (j < n)
        for (int i = 0; i < rows; ++i)
            for (int p = 0; p < red; ++p) {
                const float a = TN ? A[(size_t)p * lda + i] : A[(size_t)i * lda + p];
                for (int jj = j; jj < n; ++jj) C[(size_t)i * ldc + jj] += a * B[(size_t)p * ldb + jj];
            }
}

// C[m x n] += A[m x k] . B[k x n]
static void gemm_nn_acc(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int m, int n, int k) {
    gemm_acc<false>(A, lda, B, ldb, C, ldc, m, n, k);
}
// C[k x n] += A^T . B with A[m x k], B[m x n]
static void gemm_tn_acc(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int m, int n, int k) {
    gemm_acc<true>(A, lda, B, ldb, C, ldc, k, n, m);
}

constexpr int BLK = 128;             // gathered rows per sgemm call

// Feature slabs live in one arena that persists across steps (a training loop's caching allocator): a fresh
// std::vector per slab costs an mmap plus a page fault per 4 KB, measured 5 of 7.8 s per step on 8 threads.
struct Arena {
    std::vector<std::unique_ptr<float[]>> chunks;
    std::vector<size_t> cap;
    size_t cur = 0, used = 0;
    float* take(size_t n) {
        n = (n + 15) & ~(size_t)15;
        while (cur < chunks.size() && used + n > cap[cur]) { ++cur; used = 0; }
        if (cur == chunks.size()) {
            const size_t c = std::max<size_t>(n, (size_t)64 << 20);
            chunks.emplace_back(new float[c]);
            cap.push_back(c);
            used = 0;
        }
        float* p = chunks[cur].get() + used;
        used += n;
        return p;
    }
    void reset() { cur = 0; used = 0; }
};
Arena g_arena;

struct View {                        // minimal vector-like view of arena memory
    float* p = nullptr;
    size_t n = 0;
    float& operator[](size_t i) { return p[i]; }
    const float& operator[](size_t i) const { return p[i]; }
    float* data() { return p; }
    const float* data() const { return p; }
    size_t size() const { return n; }
    bool empty() const { return p == nullptr; }
    void clear() { p = nullptr; n = 0; }
};

struct Slab {
    int64_t n = 0;
    int c = 0;
    View v, g;                       // values, gradient (taken from the arena on first use, zero-filled)
    float* grad() {
        if (g.empty()) {
            g.n = (size_t)n * c;
            g.p = g_arena.take(g.n);
            float* q = g.p;
            const int64_t tot = (int64_t)g.n;
#pragma omp parallel for schedule(static)
            for (int64_t e = 0; e < tot; ++e) q[e] = 0.f;
        }
        return g.p;
    }
};

// Y = bias; for each offset: gather -> sgemm -> scatter-add            (SubM / Convolution / Deconvolution forward)
static void conv_fwd(const Slab& X, const Rules& R, bool swapped, const float* W, const float* b, Slab& Y) {
    Tick tk(2);
    const int cin = X.c, cout = Y.c;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < Y.n; ++r) std::memcpy(&Y.v[(size_t)r * cout], b, sizeof(float) * cout);
    for (int o = 0; o < R.n_off; ++o) {
        const std::vector<int32_t>& in = swapped ? R.out[o] : R.in[o];
        const std::vector<int32_t>& out = swapped ? R.in[o] : R.out[o];
        const int64_t P = (int64_t)in.size();
        const float* Wo = W + (size_t)o * cin * cout;
#pragma omp parallel
        {
            std::vector<float> A((size_t)BLK * cin), C((size_t)BLK * cout);
#pragma omp for schedule(static)
            for (int64_t p0 = 0; p0 < P; p0 += BLK) {
                const int m = (int)std::min<int64_t>(BLK, P - p0);
                for (int i = 0; i < m; ++i) std::memcpy(&A[(size_t)i * cin], &X.v[(size_t)in[p0 + i] * cin], sizeof(float) * cin);
                std::fill(C.begin(), C.begin() + (size_t)m * cout, 0.f);
                gemm_nn_acc(A.data(), cin, Wo, cout, C.data(), cout, m, cout, cin);
                for (int i = 0; i < m; ++i) {
                    float* y = &Y.v[(size_t)out[p0 + i] * cout];
                    const float* c = &C[(size_t)i * cout];
                    for (int j = 0; j < cout; ++j) y[j] += c[j];
                }
            }
        }
    }
}

// dX += dY . W^T on the swapped pairs; dW[o] = X_g^T . dY_g; db = column sums of dY
static void conv_bwd(Slab& X, const Rules& R, bool swapped, const float* W, Slab& Y, float* dW, float* db) {
    Tick tk(3);
    const int cin = X.c, cout = Y.c;
    const float* dY = Y.grad();
    float* dX = X.grad();
    for (int j = 0; j < cout; ++j) db[j] = 0.f;
    for (int64_t r = 0; r < Y.n; ++r)
        for (int j = 0; j < cout; ++j) db[j] += dY[(size_t)r * cout + j];
    std::vector<float> WT((size_t)cin * cout);
#ifdef _OPENMP
    const int nth = omp_get_max_threads();
#else
    const int nth = 1;
#endif
    std::vector<float> dWp((size_t)nth * cin * cout);
    for (int o = 0; o < R.n_off; ++o) {
        const std::vector<int32_t>& in = swapped ? R.out[o] : R.in[o];
        const std::vector<int32_t>& out = swapped ? R.in[o] : R.out[o];
        const int64_t P = (int64_t)in.size();
        const float* Wo = W + (size_t)o * cin * cout;
        for (int k = 0; k < cin; ++k)
            for (int j = 0; j < cout; ++j) WT[(size_t)j * cin + k] = Wo[(size_t)k * cout + j];
        std::fill(dWp.begin(), dWp.end(), 0.f);
#pragma omp parallel
        {
#ifdef _OPENMP
            float* dWt = &dWp[(size_t)omp_get_thread_num() * cin * cout];
#else
            float* dWt = dWp.data();
#endif
            std::vector<float> A((size_t)BLK * cin), G((size_t)BLK * cout), C((size_t)BLK * cin);
#pragma omp for schedule(static)
            for (int64_t p0 = 0; p0 < P; p0 += BLK) {
                const int m = (int)std::min<int64_t>(BLK, P - p0);
                for (int i = 0; i < m; ++i) {
                    std::memcpy(&A[(size_t)i * cin], &X.v[(size_t)in[p0 + i] * cin], sizeof(float) * cin);
                    std::memcpy(&G[(size_t)i * cout], &dY[(size_t)out[p0 + i] * cout], sizeof(float) * cout);
                }
                std::fill(C.begin(), C.begin() + (size_t)m * cin, 0.f);
                gemm_nn_acc(G.data(), cout, WT.data(), cin, C.data(), cin, m, cin, cout);
                for (int i = 0; i < m; ++i) {                       // input rows of one offset are distinct
                    float* x = &dX[(size_t)in[p0 + i] * cin];
                    const float* c = &C[(size_t)i * cin];
                    for (int k = 0; k < cin; ++k) x[k] += c[k];
                }
                gemm_tn_acc(A.data(), cin, G.data(), cout, dWt, cout, m, cout, cin);
            }
        }
        float* dWo = dW + (size_t)o * cin * cout;
        for (size_t e = 0; e < (size_t)cin * cout; ++e) {
            float s = 0.f;
            for (int t = 0; t < nth; ++t) s += dWp[(size_t)t * cin * cout + e];
            dWo[e] = s;
        }
    }
}

static Rules identity_rules(int64_t n) {
    Rules r;
    r.n_off = 1;
    r.in.resize(1);
    r.out.resize(1);
    r.in[0].resize(n);
    for (int64_t i = 0; i < n; ++i) r.in[0][i] = (int32_t)i;
    r.out[0] = r.in[0];
    return r;
}

// ---------------------------------------------------------------------------------------------------------------------
// index structures
// ---------------------------------------------------------------------------------------------------------------------
static void build_subm(Level& L) {
    const int64_t n = L.n;
    L.subm.n_off = 27;
    L.subm.in.assign(27, {});
    L.subm.out.assign(27, {});
#ifdef _OPENMP
    const int nth = omp_get_max_threads();
#else
    const int nth = 1;
#endif
    std::vector<std::vector<int32_t>> tin((size_t)nth * 27), tout((size_t)nth * 27);
#pragma omp parallel
    {
#ifdef _OPENMP
        const int t = omp_get_thread_num();
#else
        const int t = 0;
#endif
        const int64_t lo = n * t / nth, hi = n * (t + 1) / nth;            // contiguous row ranges keep `out` ascending
        for (int64_t r = lo; r < hi; ++r) {
            const int32_t* c = &L.coords[(size_t)r * 4];
            for (int o = 0; o < 27; ++o) {
                const int dx = o / 9 - 1, dy = (o / 3) % 3 - 1, dz = o % 3 - 1;
                const int x = c[0] + dx, y = c[1] + dy, z = c[2] + dz;
                if ((unsigned)x >= 65536u || (unsigned)y >= 65536u || (unsigned)z >= 65536u) continue;
                const int32_t q = o == 13 ? (int32_t)r : L.grid.find(HashGrid::pack(x, y, z, c[3]));
                if (q >= 0) { tin[(size_t)t * 27 + o].push_back(q); tout[(size_t)t * 27 + o].push_back((int32_t)r); }
            }
        }
    }
    for (int o = 0; o < 27; ++o)
        for (int t = 0; t < nth; ++t) {
            L.subm.in[o].insert(L.subm.in[o].end(), tin[(size_t)t * 27 + o].begin(), tin[(size_t)t * 27 + o].end());
            L.subm.out[o].insert(L.subm.out[o].end(), tout[(size_t)t * 27 + o].begin(), tout[(size_t)t * 27 + o].end());
        }
}

static void build_down(Level& F, Level& Cn) {       // coarse sites by first occurrence over fine rows ascending
    Cn.grid.init(F.n);
    Cn.coords.clear();
    std::vector<int32_t> parent(F.n);
    int32_t next = 0;
    for (int64_t r = 0; r < F.n; ++r) {
        const int32_t* c = &F.coords[(size_t)r * 4];
        bool ins;
        const int32_t row = Cn.grid.find_or_insert(HashGrid::pack(c[0] >> 1, c[1] >> 1, c[2] >> 1, c[3]), next, &ins);
        if (ins) {
            Cn.coords.insert(Cn.coords.end(), {c[0] >> 1, c[1] >> 1, c[2] >> 1, c[3]});
            ++next;
        }
        parent[r] = row;
    }
    Cn.n = next;
    std::vector<int32_t> child((size_t)8 * next, -1);
    for (int64_t r = 0; r < F.n; ++r) {
        const int32_t* c = &F.coords[(size_t)r * 4];
        const int o = ((c[0] & 1) * 2 + (c[1] & 1)) * 2 + (c[2] & 1);
        child[(size_t)o * next + parent[r]] = (int32_t)r;
    }
    F.down.n_off = 8;
    F.down.in.assign(8, {});
    F.down.out.assign(8, {});
    for (int o = 0; o < 8; ++o)
        for (int32_t q = 0; q < next; ++q)
            if (child[(size_t)o * next + q] >= 0) { F.down.in[o].push_back(child[(size_t)o * next + q]); F.down.out[o].push_back(q); }
}

struct Param { const float* w; const float* b; float* dw; float* db; };

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// C entry point: one full step (rulebooks + forward + backward) of the A12 U-Net on one batch
// ---------------------------------------------------------------------------------------------------------------------
extern "C" int scn_cpu_unet_step(const int64_t* coords, int64_t n_pts, const float* feats, int cin, const int* channels,
                                 int n_levels, const float* params, const float* dY_or_null, float* out_or_null,
                                 float* grads_or_null, float* dfeats_or_null, int64_t* n_active, int64_t* n_rules_level0,
                                 int threads) {
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
    if (n_levels < 1 || n_levels > 8 || cin < 1 || n_pts < 0) return 1;
    for (double& t : g_t) t = 0.0;
    const double t_begin = Tick::now();
    std::unique_ptr<Tick> ph(new Tick(0));
    // ---- InputLayer mode 4 (custom_operations.py:67-83): rows by first occurrence, mean of duplicates ----------------
    std::vector<Level> Lv(n_levels);
    Level& L0 = Lv[0];
    L0.grid.init(n_pts);
    std::vector<int32_t> prow(n_pts);
    int32_t next = 0;
    for (int64_t i = 0; i < n_pts; ++i) {
        const int64_t* c = coords + i * 4;
        if (c[0] < 0 || c[0] > 65535 || c[1] < 0 || c[1] > 65535 || c[2] < 0 || c[2] > 65535 || c[3] < 0 || c[3] > 65534) return 3;
        bool ins;
        prow[i] = L0.grid.find_or_insert(HashGrid::pack(c[0], c[1], c[2], c[3]), next, &ins);
        if (ins) {
            L0.coords.insert(L0.coords.end(), {(int32_t)c[0], (int32_t)c[1], (int32_t)c[2], (int32_t)c[3]});
            ++next;
        }
    }
    L0.n = next;
    *n_active = next;
    std::vector<int32_t> count(next, 0);
    std::vector<double> acc((size_t)next * cin, 0.0);
    for (int64_t i = 0; i < n_pts; ++i) {
        ++count[prow[i]];
        for (int k = 0; k < cin; ++k) acc[(size_t)prow[i] * cin + k] += feats[(size_t)i * cin + k];
    }
    // ---- rulebooks ------------------------------------------------------------------------------------------------------
    ph.reset(new Tick(1));
    for (int l = 0; l < n_levels; ++l) {
        if (l + 1 < n_levels) build_down(Lv[l], Lv[l + 1]);
        build_subm(Lv[l]);
    }
    if (n_rules_level0) {
        int64_t t = 0;
        for (int o = 0; o < 27; ++o) t += (int64_t)L0.subm.in[o].size();
        *n_rules_level0 = t;
    }
    ph.reset();
    // ---- parameters in the order of oracle.scn_oracle.unet_param_shapes -------------------------------------------------
    std::vector<Param> P;
    {
        size_t off = 0;
        auto take = [&](size_t nw, size_t nb) {
            Param p;
            p.w = params + off; p.dw = grads_or_null ? grads_or_null + off : nullptr; off += nw;
            p.b = params + off; p.db = grads_or_null ? grads_or_null + off : nullptr; off += nb;
            P.push_back(p);
        };
        for (int l = 0; l < n_levels; ++l) {
            const int c = channels[l];
            if (l == 0) take((size_t)cin * c, c); else take((size_t)8 * channels[l - 1] * c, c);
            for (int u = 0; u < 4; ++u) take((size_t)27 * c * c, c);
        }
        for (int l = n_levels - 2; l >= 0; --l) {
            const int c = channels[l], cup = channels[l + 1];
            take((size_t)8 * cup * c, c);
            take((size_t)2 * c * c, c);
            for (int u = 0; u < 4; ++u) take((size_t)27 * c * c, c);
        }
    }
    std::vector<std::vector<float>> scratch_dw;       // when the caller wants no gradients, backward still computes them
    auto dw_of = [&](const Param& p, size_t nw) -> float* {
        if (p.dw) return p.dw;
        scratch_dw.emplace_back(nw);
        return scratch_dw.back().data();
    };
    auto db_of = [&](const Param& p, size_t nb) -> float* {
        if (p.db) return p.db;
        scratch_dw.emplace_back(nb);
        return scratch_dw.back().data();
    };

    // ---- tape ------------------------------------------------------------------------------------------------------------
    g_arena.reset();
    enum Kind { CONV, RELU, ADD, CAT };
    struct Op { Kind k; int a, b, y; const Rules* R; bool swapped; int param; };
    std::vector<std::unique_ptr<Slab>> S;
    std::vector<Op> tape;
    std::vector<Rules> ident(n_levels);
    for (int l = 0; l < n_levels; ++l) ident[l] = identity_rules(Lv[l].n);
    auto new_slab = [&](int64_t n, int c) { Tick tk(6); S.emplace_back(new Slab); S.back()->n = n; S.back()->c = c; S.back()->v.n = (size_t)n * c; S.back()->v.p = g_arena.take((size_t)n * c); return (int)S.size() - 1; };
    auto conv = [&](int x, const Rules* R, bool swapped, int param, int64_t n_out, int cout) {
        const int y = new_slab(n_out, cout);
        conv_fwd(*S[x], *R, swapped, P[param].w, P[param].b, *S[y]);
        tape.push_back({CONV, x, -1, y, R, swapped, param});
        return y;
    };
    auto relu = [&](int x) {
        const int y = new_slab(S[x]->n, S[x]->c);
        Tick tk(4);
        const size_t tot = S[x]->v.size();
        const float* a = S[x]->v.data();
        float* o = S[y]->v.data();
#pragma omp parallel for schedule(static)
        for (int64_t e = 0; e < (int64_t)tot; ++e) o[e] = a[e] > 0.f ? a[e] : 0.f;
        tape.push_back({RELU, x, -1, y, nullptr, false, -1});
        return y;
    };
    auto add = [&](int a, int b) {
        const int y = new_slab(S[a]->n, S[a]->c);
        Tick tk(4);
        const size_t tot = S[a]->v.size();
#pragma omp parallel for schedule(static)
        for (int64_t e = 0; e < (int64_t)tot; ++e) S[y]->v[e] = S[a]->v[e] + S[b]->v[e];
        tape.push_back({ADD, a, b, y, nullptr, false, -1});
        return y;
    };
    auto cat = [&](int a, int b) {
        const int ca = S[a]->c, cb = S[b]->c;
        const int y = new_slab(S[a]->n, ca + cb);
        Tick tk(4);
        for (int64_t r = 0; r < S[a]->n; ++r) {
            std::memcpy(&S[y]->v[(size_t)r * (ca + cb)], &S[a]->v[(size_t)r * ca], sizeof(float) * ca);
            std::memcpy(&S[y]->v[(size_t)r * (ca + cb) + ca], &S[b]->v[(size_t)r * cb], sizeof(float) * cb);
        }
        tape.push_back({CAT, a, b, y, nullptr, false, -1});
        return y;
    };
    int pi = 0;
    auto residual_units = [&](int x, int l) {
        const int c = channels[l];
        for (int u = 0; u < 2; ++u) {
            int y = conv(relu(x), &Lv[l].subm, false, pi++, Lv[l].n, c);
            y = conv(relu(y), &Lv[l].subm, false, pi++, Lv[l].n, c);
            x = add(x, y);
        }
        return x;
    };
    int x = new_slab(L0.n, cin);
    for (int64_t r = 0; r < L0.n; ++r)
        for (int k = 0; k < cin; ++k) S[x]->v[(size_t)r * cin + k] = (float)(acc[(size_t)r * cin + k] / count[r]);
    const int x_in = x;
    std::vector<int> skips;
    for (int l = 0; l < n_levels; ++l) {
        if (l == 0) x = conv(x, &ident[0], false, pi++, L0.n, channels[0]);
        else x = conv(x, &Lv[l - 1].down, false, pi++, Lv[l].n, channels[l]);
        x = residual_units(x, l);
        skips.push_back(x);
    }
    for (int l = n_levels - 2; l >= 0; --l) {
        const int up = conv(relu(x), &Lv[l].down, true, pi++, Lv[l].n, channels[l]);
        x = conv(cat(up, skips[l]), &ident[l], false, pi++, Lv[l].n, channels[l]);
        x = residual_units(x, l);
    }
    if (out_or_null) std::memcpy(out_or_null, S[x]->v.data(), sizeof(float) * S[x]->v.size());

    // ---- backward --------------------------------------------------------------------------------------------------------
    {
        float* g = S[x]->grad();
        const size_t tot = S[x]->v.size();
        if (dY_or_null) std::memcpy(g, dY_or_null, sizeof(float) * tot);
        else std::fill(g, g + tot, 1.f);
    }
    for (int t = (int)tape.size() - 1; t >= 0; --t) {
        const Op& op = tape[t];
        Slab& Y = *S[op.y];
        if (Y.g.empty()) continue;
        switch (op.k) {
        case CONV: {
            Slab& X = *S[op.a];
            const size_t nw = (size_t)op.R->n_off * X.c * Y.c;
            conv_bwd(X, *op.R, op.swapped, P[op.param].w, Y, dw_of(P[op.param], nw), db_of(P[op.param], Y.c));
            break;
        }
        case RELU: {
            Tick tk(5);
            Slab& X = *S[op.a];
            float* gx = X.grad();
            const size_t tot = X.v.size();
#pragma omp parallel for schedule(static)
            for (int64_t e = 0; e < (int64_t)tot; ++e) gx[e] += X.v[e] > 0.f ? Y.g[e] : 0.f;
            break;
        }
        case ADD: {
            Tick tk(5);
            float* ga = S[op.a]->grad();
            float* gb = S[op.b]->grad();
            const size_t tot = Y.v.size();
#pragma omp parallel for schedule(static)
            for (int64_t e = 0; e < (int64_t)tot; ++e) { ga[e] += Y.g[e]; gb[e] += Y.g[e]; }
            break;
        }
        case CAT: {
            Tick tk(5);
            Slab &A = *S[op.a], &B = *S[op.b];
            float* ga = A.grad();
            float* gb = B.grad();
            const int ca = A.c, cb = B.c;
            for (int64_t r = 0; r < A.n; ++r) {
                for (int k = 0; k < ca; ++k) ga[(size_t)r * ca + k] += Y.g[(size_t)r * (ca + cb) + k];
                for (int k = 0; k < cb; ++k) gb[(size_t)r * cb + k] += Y.g[(size_t)r * (ca + cb) + ca + k];
            }
            break;
        }
        }
        Y.g.clear();
    }
    if (dfeats_or_null) {                      // InputLayer backward, mode 4: dF[i] = dX[row(i)] / multiplicity
        const float* gx = S[x_in]->grad();
        for (int64_t i = 0; i < n_pts; ++i)
            for (int k = 0; k < cin; ++k)
                dfeats_or_null[(size_t)i * cin + k] = gx[(size_t)prow[i] * cin + k] / (float)count[prow[i]];
    }
    if (std::getenv("SCN_CPU_PROFILE")) {
        double tot = Tick::now() - t_begin, acc = 0.0;
        for (int k = 0; k < 7; ++k) { std::fprintf(stderr, "[cpu] %-12s %7.3f s\n", g_tn[k], g_t[k]); acc += g_t[k]; }
        std::fprintf(stderr, "[cpu] %-12s %7.3f s   total %7.3f s\n", g_tn[7], tot - acc, tot);
    }
    return 0;
}

extern "C" int scn_cpu_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

extern "C" const char* scn_cpu_isa(void) {
#if defined(__AVX512F__)
    return "avx512";
#elif defined(__AVX2__)
    return "avx2";
#else
    return "generic";
#endif
}
