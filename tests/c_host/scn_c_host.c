/* A host written in C against include/scn_mi355x.h only -- no Python, no torch: what a maintainer of a compiled
 * caller would link.  Builds the index structures of a small two-sample scene with one call (scn_pyramid_build), runs
 * the hot kernel (scn_conv_tiles: SubmanifoldConvolution 3^3, 32 -> 32 channels, bias, input ReLU) and its
 * backward-data launch, and checks everything against a brute-force restatement written here:
 *   rows      : first-occurrence numbering of the distinct (x,y,z,sample) sites            -- exact
 *   neighbours: table[o][r] = row of site(r) + delta_o, o = ((dx+1)*3 + (dy+1))*3 + (dz+1)  -- exact
 *   features  : Y[r] = b + sum_o relu(X[table[o][r]]) . W[o]                               -- 1e-4 (fp32)
 *   adjoint   : <conv(X), G> == <X, conv^T(G)> with the transposed / reversed launch        -- 1e-4
 * Exit code 0 = all checks passed.  Built by __graft_entry__.build(); run by tests/test_gpu_c_host.py. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "scn_mi355x.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
#define CHECK_SCN(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "scn error %d (%s) at %s:%d\n", r_, scn_last_error_string(), __FILE__, __LINE__); return 3; } } while (0)

enum { G = 16, B = 2, NPTS = 1500, C = 32, NOFF = 27 };

static uint32_t lcg_state = 12345u;
static uint32_t lcg(void) { lcg_state = lcg_state * 1664525u + 1013904223u; return lcg_state >> 8; }
static float frand(void) { return (float)(lcg() % 20001) / 10000.0f - 1.0f; }

int main(void) {
    if (scn_abi_version() != SCN_ABI_VERSION || SCN_ABI_VERSION != 5) { fprintf(stderr, "ABI version\n"); return 1; }
    /* ---- scene: points with duplicates, two samples ------------------------------------------------ */
    static int64_t coords[NPTS][4];
    for (int p = 0; p < NPTS; ++p) {
        if (p >= 100 && lcg() % 5 == 0) { memcpy(coords[p], coords[lcg() % p], sizeof coords[p]); continue; }
        coords[p][0] = lcg() % G; coords[p][1] = lcg() % G; coords[p][2] = lcg() % (G / 2); coords[p][3] = p < NPTS / 2 ? 0 : 1;
    }
    /* brute force: first-occurrence rows */
    static int grid[B][G][G][G];
    memset(grid, 0xff, sizeof grid);
    static int site[NPTS][4];
    static int item_row_ref[NPTS];
    int n0 = 0;
    for (int p = 0; p < NPTS; ++p) {
        int* cell = &grid[coords[p][3]][coords[p][0]][coords[p][1]][coords[p][2]];
        if (*cell < 0) { *cell = n0; for (int d = 0; d < 4; ++d) site[n0][d] = (int)coords[p][d]; ++n0; }
        item_row_ref[p] = *cell;
    }

    /* ---- device: one call builds every index structure -------------------------------------------- */
    int64_t* d_coords; void* ws;
    const int64_t ws_bytes = scn_pyramid_workspace_bytes(NPTS, 1, 3);
    if (ws_bytes <= 0) { fprintf(stderr, "workspace bytes\n"); return 1; }
    CHECK_HIP(hipMalloc((void**)&d_coords, sizeof coords));
    CHECK_HIP(hipMalloc(&ws, (size_t)ws_bytes));
    CHECK_HIP(hipMemcpy(d_coords, coords, sizeof coords, hipMemcpyHostToDevice));
    static int64_t desc[SCN_PYRAMID_DESC_LEN];
    CHECK_SCN(scn_pyramid_build(d_coords, NPTS, 1, 3, ws, ws_bytes, desc, NULL));
    const int64_t* L = desc + 8;
    const int64_t n = L[0], nt = L[13];
    int fails = 0;
    if (n != n0) { fprintf(stderr, "row count %lld != %d\n", (long long)n, n0); return 4; }
    if (desc[3] != 0) { fprintf(stderr, "out-of-range coordinates reported\n"); return 4; }
    char* w8 = (char*)ws;
    static int32_t item_row[NPTS], lv_coords[NPTS][4];
    int32_t* table = (int32_t*)malloc(sizeof(int32_t) * NOFF * n);
    CHECK_HIP(hipMemcpy(item_row, w8 + desc[4], sizeof(int32_t) * NPTS, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(lv_coords, w8 + L[2], sizeof(int32_t) * 4 * n, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(table, w8 + L[5], sizeof(int32_t) * NOFF * n, hipMemcpyDeviceToHost));
    for (int p = 0; p < NPTS; ++p) fails += item_row[p] != item_row_ref[p];
    for (int r = 0; r < n; ++r) for (int d = 0; d < 4; ++d) fails += lv_coords[r][d] != site[r][d];
    if (fails) { fprintf(stderr, "row numbering differs in %d places\n", fails); return 4; }
    int64_t n_rules = 0;
    for (int o = 0; o < NOFF; ++o) {
        const int dx = o / 9 - 1, dy = o / 3 % 3 - 1, dz = o % 3 - 1;
        for (int r = 0; r < n; ++r) {
            const int x = site[r][0] + dx, y = site[r][1] + dy, z = site[r][2] + dz;
            const int exp = (x < 0 || y < 0 || z < 0 || x >= G || y >= G || z >= G) ? -1 : grid[site[r][3]][x][y][z];
            fails += table[(int64_t)o * n + r] != exp;
            n_rules += exp >= 0;
        }
    }
    if (fails) { fprintf(stderr, "neighbour table differs in %d entries\n", fails); return 5; }
    if (L[25 + NOFF] != n_rules) { fprintf(stderr, "rule count %lld != %lld\n", (long long)L[25 + NOFF], (long long)n_rules); return 5; }

    /* ---- the hot kernel --------------------------------------------------------------------------- */
    float* X = (float*)malloc(sizeof(float) * n * C); float* Gy = (float*)malloc(sizeof(float) * n * C);
    float* W = (float*)malloc(sizeof(float) * NOFF * C * C); float bias[C];
    for (int64_t e = 0; e < n * C; ++e) { X[e] = frand(); Gy[e] = frand(); }
    for (int e = 0; e < NOFF * C * C; ++e) W[e] = frand() * 0.05f;
    for (int c = 0; c < C; ++c) bias[c] = frand() * 0.5f;
    float *dX, *dG, *dW, *dB, *dY, *dXg; void* scratch;
    const int64_t sbytes = scn_conv_tiles_scratch_bytes(C, n, C);
    CHECK_HIP(hipMalloc((void**)&dX, sizeof(float) * n * C)); CHECK_HIP(hipMalloc((void**)&dG, sizeof(float) * n * C));
    CHECK_HIP(hipMalloc((void**)&dY, sizeof(float) * n * C)); CHECK_HIP(hipMalloc((void**)&dXg, sizeof(float) * n * C));
    CHECK_HIP(hipMalloc((void**)&dW, sizeof(float) * NOFF * C * C)); CHECK_HIP(hipMalloc((void**)&dB, sizeof bias));
    CHECK_HIP(hipMalloc(&scratch, (size_t)(sbytes > 0 ? sbytes : 256)));
    CHECK_HIP(hipMemcpy(dX, X, sizeof(float) * n * C, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(dG, Gy, sizeof(float) * n * C, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(dW, W, sizeof(float) * NOFF * C * C, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(dB, bias, sizeof bias, hipMemcpyHostToDevice));
    const int32_t* perm = (const int32_t*)(w8 + L[9]); const int32_t* tstab = (const int32_t*)(w8 + L[10]);
    const uint32_t* tmask = (const uint32_t*)(w8 + L[11]); const int32_t* torder = (const int32_t*)(w8 + L[12]);
    (void)nt;
    /* zeroed arrival counters: layers with more than 32 input channels add their K-chunk partial sums inside the launch */
    int32_t* arrival = NULL;
    const int64_t n_arr = scn_conv_tiles_arrival_counters(C, n, C);
    CHECK_HIP(hipMalloc((void**)&arrival, sizeof(int32_t) * (size_t)(n_arr > 0 ? n_arr : 1)));
    CHECK_HIP(hipMemset(arrival, 0, sizeof(int32_t) * (size_t)(n_arr > 0 ? n_arr : 1)));
    CHECK_SCN(scn_conv_tiles(dX, n, C, tstab, tmask, perm, torder, NOFF, n, dW, dB, NULL, NULL, dY, C, SCN_F_RELU_IN, scratch,
                             arrival, NULL));
    /* backward-data of the same layer without the ReLU: dX = sum_o G[table[26-o]] . W[o]^T */
    CHECK_SCN(scn_conv_tiles(dG, n, C, tstab, tmask, perm, torder, NOFF, n, dW, NULL, NULL, NULL, dXg, C,
                             SCN_F_W_TRANSPOSED | SCN_F_OFF_REVERSE, scratch, arrival, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    float* Y = (float*)malloc(sizeof(float) * n * C); float* Xg = (float*)malloc(sizeof(float) * n * C);
    CHECK_HIP(hipMemcpy(Y, dY, sizeof(float) * n * C, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(Xg, dXg, sizeof(float) * n * C, hipMemcpyDeviceToHost));
    double max_err = 0, scale = 1, lhs = 0, rhs = 0;
    for (int r = 0; r < n; ++r) {
        double acc[C], lin[C];
        for (int c = 0; c < C; ++c) { acc[c] = bias[c]; lin[c] = 0; }
        for (int o = 0; o < NOFF; ++o) {
            const int i = table[(int64_t)o * n + r];
            if (i < 0) continue;
            for (int k = 0; k < C; ++k) {
                const double x = X[(int64_t)i * C + k], xr = x > 0 ? x : 0;
                const float* w = W + ((int64_t)o * C + k) * C;
                for (int c = 0; c < C; ++c) { acc[c] += xr * w[c]; lin[c] += x * w[c]; }
            }
        }
        for (int c = 0; c < C; ++c) {
            const double e = fabs(acc[c] - Y[(int64_t)r * C + c]);
            if (e > max_err) max_err = e;
            if (fabs(acc[c]) > scale) scale = fabs(acc[c]);
            lhs += lin[c] * Gy[(int64_t)r * C + c];                 /* <conv_linear(X), G> */
        }
    }
    for (int64_t e = 0; e < n * C; ++e) rhs += (double)X[e] * Xg[e]; /* <X, conv^T(G)> */
    const double rel = max_err / scale, adj = fabs(lhs - rhs) / (fabs(lhs) > 1 ? fabs(lhs) : 1);
    printf("c host: points %d rows %lld rules %lld tiles %lld  max|dY| %.2e (rel %.2e)  adjoint %.2e\n", NPTS, (long long)n,
           (long long)n_rules, (long long)nt, max_err, rel, adj);
    if (!(rel <= 1e-4) || !(adj <= 1e-4)) { fprintf(stderr, "feature check failed\n"); return 6; }
    printf("c host: ok\n");
    return 0;
}
