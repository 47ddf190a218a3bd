// Shared host-side helpers for libscn_mi355x (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/scn_mi355x.h"

namespace scn {

extern thread_local char g_err[512];

inline int fail(int code, const char* fmt, const char* a = "", long long b = 0, long long c = 0) {
    snprintf(g_err, sizeof(g_err), fmt, a, b, c);
    return code;
}

#define SCN_HIP(expr)                                                                               \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            snprintf(scn::g_err, sizeof(scn::g_err), "%s failed: %s (%s:%d)", #expr,                \
                     hipGetErrorString(e_), __FILE__, __LINE__);                                    \
            return SCN_EHIP;                                                                        \
        }                                                                                           \
    } while (0)

#define SCN_LAUNCH_CHECK()                                                                          \
    do {                                                                                            \
        hipError_t e_ = hipGetLastError();                                                          \
        if (e_ != hipSuccess) {                                                                     \
            snprintf(scn::g_err, sizeof(scn::g_err), "kernel launch failed: %s (%s:%d)",            \
                     hipGetErrorString(e_), __FILE__, __LINE__);                                    \
            return SCN_EHIP;                                                                        \
        }                                                                                           \
    } while (0)

#define SCN_REQUIRE(cond)                                                                           \
    do {                                                                                            \
        if (!(cond)) {                                                                              \
            snprintf(scn::g_err, sizeof(scn::g_err), "%s: requirement failed: %s", __func__, #cond); \
            return SCN_EINVAL;                                                                      \
        }                                                                                           \
    } while (0)

// Developer switches (scn_debug.hip): the environment is read once at first use; later changes go through scn_debug_set().
enum Switch {
    SW_TS_SPLIT, SW_TS_SPLIT_MAX, SW_TS_NO_TAIL, SW_TS_W_HALF, SW_TS_W_BOTH, SW_TB_NB, SW_TB_KH, SW_TB_STREAM,
    SW_TB_NO_XORDER, SW_TS_STREAM, SW_TSS_NW, SW_EXEC_DEFER_SUMS, SW_PYRAMID_V1, SW_PYRAMID_ONE_STREAM, SW_WD_NO_T3,
    SW_WGRAD_BF16_MFMA, SW_WGRAD_SPLITS, SW_WD_NO_EVEC, SW_PYRAMID_NO_BRICKS, SW_CU_BUDGET, SW_EXP_A, SW_EXP_B, SW_TS_NO_CHAIN, SW_TS_PROG, SW_TB_NO_BINS, SW_COUNT
};
struct SwitchVal { bool set = false; long long i = 0; double f = 0.0; };
SwitchVal sw(Switch s);
// CUs the matrix kernels size their grids to (SCN_CU_BUDGET, default 256 = the chip): a grid of one workgroup per CU starts some
// of its workgroups late whenever another stream's kernel (an RCCL collective, the index build) holds a CU.
inline int cu_budget() { const SwitchVal v = sw(SW_CU_BUDGET); return (v.set && v.i >= 32 && v.i <= 256) ? (int)v.i : 256; }

// "Once per device" guard of a hipFuncSetAttribute call site (a process-wide `static bool` is neither per device nor
// safe against a second launching thread).  One bit per device ordinal; the bit is set AFTER the attribute call, so two threads
// may both make the (idempotent) call but neither launches before it has been made on its device.
struct DeviceOnce {
    std::atomic<uint64_t> mask{0};
    static int cur() { int d = 0; return (hipGetDevice(&d) == hipSuccess && d >= 0 && d < 64) ? d : -1; }
    bool needed() const { const int d = cur(); return d < 0 || !(mask.load(std::memory_order_acquire) & (1ull << d)); }
    void done() { const int d = cur(); if (d >= 0) mask.fetch_or(1ull << d, std::memory_order_release); }
};

inline hipStream_t S(scn_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Grid for HBM-bound elementwise kernels: cap and grid-stride (cdna_hip_programming.md Guideline 11).
inline int ew_grid(int64_t work_items, int block) {
    int64_t g = cdiv(work_items, block);
    if (g > 256 * 8) g = 256 * 8;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace scn

// ---- device-side key packing / hashing (shared by index kernels) ----
#define SCN_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull

__device__ __forceinline__ unsigned long long scn_pack_key(int x, int y, int z, int b) {
    return ((unsigned long long)(unsigned)b << 48) | ((unsigned long long)(unsigned)x << 32) |
           ((unsigned long long)(unsigned)y << 16) | (unsigned long long)(unsigned)z;
}

// Slot of a key: a full 64-bit finaliser (the one of MurmurHash3).  Rounds 1-3 used `key * golden; h ^= h >> 29`, whose low
// log2(cap) bits do not depend on the batch field (bits 48-63 of the key reach bits >= 19 of h only): the voxels that several
// samples share -- the overlapping boxes of an ROI batch keep their absolute scene coordinates, roi_select_sparse.py:136-149 --
// all started their probe at ONE slot and formed clusters as long as the number of samples (index build of a 64-box ROI
// batch: 2.6 ms against 0.9 ms for as many points of one scene; 0.86 ms with this function).  Slot layout only: no result
// depends on it.
// Measured and NOT kept (round 4): a locality-preserving layout -- the 16 sites of a 4 x 4 (y, z) column in one 128-byte line
// of the key array, slot = hash(column) * 16 + (y & 3) * 4 + (z & 3), so that the 27 neighbour probes of a row touch ~7 lines
// instead of 27 (every probe of a scattered table is an L2 miss served by the Infinity Cache).  Surfaces fill whole columns,
// linear probing then builds runs across lines, and the 19 of 27 probes that MISS walk them: table kernel 98 -> 336 us,
// coarse-site numbering 33 -> 100 us at 150 k voxels.
__device__ __forceinline__ unsigned long long scn_hash_slot(unsigned long long key, unsigned long long mask) {
    unsigned long long h = key;
    h ^= h >> 33;
    h *= 0xff51afd7ed558ccdull;
    h ^= h >> 33;
    h *= 0xc4ceb9fe1a85ec53ull;
    h ^= h >> 33;
    return h & mask;
}
