// scn_pyramid_build: every index structure of one forward pass of an n-level U-Net in ONE call.
//
// InputLayer rules (voxel hash, first-occurrence rows) + per level the SubM neighbour table, its rule scan and its
// mask-sorted tiles + between levels the coarse-site numbering, child table, rule scan and tiles -- the same kernels, in
// the same order, as the step-by-step entry points of scn_index.hip / scn_tiles.hip queue them from Python
// (sparse_rcnn_amd/metadata.py), so the results are bit-identical.  What changes is who drives them: ~35 C calls and
// ~100 tensor allocations of the Python path become one call that carves its buffers out of a caller-provided workspace
// and spends its life inside the HIP runtime, i.e. WITHOUT the interpreter lock.  A helper thread can therefore build the
// structures of the next batch while the main thread queues the matrix kernels of the current one (the Python-driven
// prefetch fought the main thread for the GIL and gained nothing); tools/bound_no_index.py: 7.18 -> 6.08 ms/step if the
// index build were free.
//
// The call waits for the device once per level size (events behind the count copies, which are queued AHEAD of the SubM work
// of the level above: the GPU keeps working while the host learns a size) and once for the rule-list sizes, after which the
// compacted rule lists are queued too; all of them on the caller's stream only.
#include <stdlib.h>

#include "scn_common.h"

using scn::S;
using scn::cdiv;

namespace {
// Pinned host words for the row counts that travel back during a build, and events that mark their arrival: the D2H copy of
// a level's coarse-row count is queued right after the numbering kernels, BEFORE the SubM table / scan / tile kernels of the
// level above are queued, and the host waits for the copy's event only -- it learns the count while that work still runs
// and queues the next level behind it, so the GPU no longer idles through a host round trip per level (round 2: the copy
// sat behind the SubM work and the host waited for the whole stream, five times per build).
// Round 3 also splits the build over TWO queues: the caller's stream carries the chain every level depends on (number the
// coarse sites of level l + 1 -> child table / rule scan / tiles of the strided rulebook l -> number level l + 2 ...), a
// library-owned side stream the SubM work of every level (neighbour table, rule scan, mask sort, tiles: ~30 launches per
// level that nothing on the main chain waits for).  ~130 launches of a few microseconds each used to run back to back; now
// the two chains (~55 and ~75 launches) run side by side: 1.06 -> 0.90 ms per build alone on an idle GPU, 1.50 -> 1.18 for
// six levels (tools/index_ab.py).  The side stream is ordered behind the event that marks a level's grid complete, and the
// caller's stream is ordered behind the side stream before the call returns (the caller still sees ONE stream-ordered
// build).  It is an OPTION (scn_pyramid_build_ex, SCN_PYRAMID_TWO_QUEUES): next to another batch's matrix kernels -- the
// pipelined prefetch of bench.py -- the second queue of tiny high-priority kernels costs the step ~0.1 ms (5.83 vs 5.71 ms,
// three alternating runs), so only the inline builds ask for it.  One side stream PER LEVEL and rulebook kind (up to 16
// queues) was measured too: no faster alone (1.27 vs 1.25 ms) and the pipelined step lost 0.4 ms.
// SCN_PYRAMID_ONE_STREAM=1: everything on the caller's stream whatever the flag (same bits either way).
struct HostSlots {
    int64_t* words = nullptr;                       // [SCN_PYRAMID_MAX_LEVELS + 2]
    hipEvent_t ev[SCN_PYRAMID_MAX_LEVELS + 2] = {};  // count copies
    hipEvent_t grid_ev[SCN_PYRAMID_MAX_LEVELS + 1] = {};   // grid of level l complete on the caller's stream
    static constexpr int NSIDE = 1;
    hipEvent_t join_ev[NSIDE] = {};
    hipStream_t side[NSIDE] = {};                           // the SubM work of every level
    bool ok = false;
    bool init() {
        if (ok) return true;
        if (hipHostMalloc((void**)&words, sizeof(int64_t) * (SCN_PYRAMID_MAX_LEVELS + 2), hipHostMallocDefault) != hipSuccess)
            return false;
        for (auto& e : ev)
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return false;
        for (auto& e : grid_ev)
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return false;
        for (auto& e : join_ev)
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return false;
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) return false;
        for (auto& q : side)
            if (hipStreamCreateWithPriority(&q, hipStreamNonBlocking, hi) != hipSuccess) return false;
        ok = true;
        return true;
    }
};
thread_local HostSlots g_slots;                     // one per calling thread (main thread, index helper thread)

struct Bump {
    char* base;
    int64_t used, cap;
    bool ok = true;
    void* take(int64_t bytes, int64_t* off) {
        used = (used + 255) & ~(int64_t)255;
        *off = used;
        used += bytes;
        if (used > cap) ok = false;
        return base + *off;
    }
};

inline int64_t nt_of(int64_t n) { return cdiv(n, 16); }

// bytes the builder takes for a level whose row count is bounded by n (and whose parent level by n_parent)
int64_t level_bytes(int64_t n, int n_off, bool has_next) {
    const int64_t pad = 256;
    int64_t b = 0;
    const int64_t nt = nt_of(n);
    b += (int64_t)n_off * n * 4 + pad;                                  // table
    b += scn_rules_blocks(n_off, n) * 4 + pad + (n_off + 1) * 8 + pad;  // scan
    b += 2 * ((int64_t)n_off * n * 4 + pad);                            // compacted rules (at most one per table entry)
    b += nt * 16 * 4 + pad + nt * n_off * 16 * 4 + pad + nt * 4 + pad + (2 * nt + 16) * 4 + pad + scn_tiles_scratch_bytes(n_off, n) + pad;
    if (has_next) {                                                     // numbering of the coarse sites + strided rulebook
        const int64_t cap = scn_hash_capacity(n);
        b += cap * 8 + pad + cap * 4 + pad + n * 4 + pad + n * 16 + pad + scn_dedup_scratch_bytes(n) + pad + 8 + pad;
        const int64_t ntc = nt_of(n);
        b += 8 * n * 4 + pad + n * 4 + pad;                             // child (<= n coarse rows), fine_off
        b += scn_rules_blocks(8, n) * 4 + pad + 9 * 8 + pad + 2 * (n * 4 + pad);      // every fine row is one rule
        b += ntc * 16 * 4 + pad + ntc * 8 * 16 * 4 + pad + ntc * 4 + pad + ntc * 4 + pad + scn_tiles_scratch_bytes(8, n) + pad;
    }
    return b;
}
}  // namespace

namespace scn {
int64_t pyramid2_workspace_bytes(int64_t n_points, int n_levels, int k);
int pyramid2_build(const int64_t* coords, int64_t n_points, int n_levels, int k, void* workspace, int64_t workspace_bytes,
                   int64_t* desc, int flags, scn_stream_t stream);
}  // namespace scn

static int64_t v1_workspace_bytes(int64_t n_points, int n_levels, int k) {
    const int64_t n = n_points > 0 ? n_points : 1, pad = 256;
    const int64_t cap = scn_hash_capacity(n);
    int64_t b = n * 16 + pad + 4 + pad;                                                     // int32 coords, range flag
    b += cap * 8 + pad + cap * 4 + pad + 3 * (n * 4 + pad) + n * 16 + pad + scn_dedup_scratch_bytes(n) + pad + 8 + pad;
    for (int l = 0; l < n_levels; ++l) b += level_bytes(n, k * k * k, l + 1 < n_levels);   // every level is bounded by n
    return b + 4096;
}

extern "C" int64_t scn_pyramid_workspace_bytes(int64_t n_points, int n_levels, int k) {
    if (n_points < 0 || n_levels < 1 || n_levels > SCN_PYRAMID_MAX_LEVELS || k < 1 || k > 3 || k % 2 == 0) return -1;
    const int64_t v1 = v1_workspace_bytes(n_points, n_levels, k);
    if (k != 3) return v1;
    // (the fused builder places every level at its upper bound and keeps its sort buffers per level: a little more)
    const int64_t v2 = scn::pyramid2_workspace_bytes(n_points, n_levels, k);
    return v2 > v1 ? v2 : v1;
}

extern "C" int scn_pyramid_build(const int64_t* coords, int64_t n_points, int n_levels, int k, void* workspace,
                                 int64_t workspace_bytes, int64_t* desc, scn_stream_t stream) {
    return scn_pyramid_build_ex(coords, n_points, n_levels, k, workspace, workspace_bytes, desc, 0, stream);
}

extern "C" int scn_pyramid_build_ex(const int64_t* coords, int64_t n_points, int n_levels, int k, void* workspace,
                                    int64_t workspace_bytes, int64_t* desc, int flags, scn_stream_t stream) {
    SCN_REQUIRE(coords && workspace && desc && n_points >= 1 && n_levels >= 1 && n_levels <= SCN_PYRAMID_MAX_LEVELS);
    SCN_REQUIRE(k == 1 || k == 3);
    SCN_REQUIRE(((uintptr_t)workspace & 255) == 0);
    // SCN_PYRAMID_FUSED: the build without host round trips (scn_pyramid2.hip); SCN_PYRAMID_V1=1 keeps this file's builder
    if ((flags & SCN_PYRAMID_FUSED) && k == 3 && !scn::sw(scn::SW_PYRAMID_V1).set)
        return scn::pyramid2_build(coords, n_points, n_levels, k, workspace, workspace_bytes, desc, flags, stream);
    const int n_off = k * k * k;
    hipStream_t st = S(stream);
    Bump ws{(char*)workspace, 0, workspace_bytes};
    for (int i = 0; i < SCN_PYRAMID_DESC_LEN; ++i) desc[i] = 0;
    desc[0] = n_levels;
    desc[1] = n_points;
    int rc;

    // ---- InputLayer: int64 -> int32 coordinates (range check), first-occurrence numbering ------------------------
    int64_t off;
    int32_t* c32 = (int32_t*)ws.take(n_points * 16, &off);        desc[7] = off;
    int32_t* flag = (int32_t*)ws.take(4, &off);
    const int64_t cap0 = scn_hash_capacity(n_points);
    uint64_t* keys = (uint64_t*)ws.take(cap0 * 8, &off);           const int64_t off_keys0 = off;
    int32_t* hrows = (int32_t*)ws.take(cap0 * 4, &off);            const int64_t off_hrows0 = off;
    int32_t* item_row = (int32_t*)ws.take(n_points * 4, &off);     desc[4] = off;
    int32_t* row_count = (int32_t*)ws.take(n_points * 4, &off);    desc[5] = off;
    int32_t* row_first = (int32_t*)ws.take(n_points * 4, &off);    desc[6] = off;
    int32_t* row_coords = (int32_t*)ws.take(n_points * 16, &off);  const int64_t off_coords0 = off;
    void* dscr = ws.take(scn_dedup_scratch_bytes(n_points), &off);
    int64_t* cnt_dev = (int64_t*)ws.take(8, &off);
    SCN_REQUIRE(ws.ok);
    if ((rc = scn_coords_to_i32(coords, n_points, c32, flag, nullptr, stream))) return rc;
    if ((rc = scn_dedup_launch(c32, n_points, 0, keys, hrows, cap0, item_row, row_count, row_first, row_coords, dscr,
                               cnt_dev, stream)))
        return rc;
    if (!g_slots.init()) return scn::fail(SCN_EHIP, "%spinned host words / events for the level sizes could not be created", "");
    HostSlots& hs = g_slots;
    hs.words[0] = 0; hs.words[1] = 0;
    SCN_HIP(hipMemcpyAsync(&hs.words[0], cnt_dev, 8, hipMemcpyDeviceToHost, st));
    SCN_HIP(hipMemcpyAsync(&hs.words[1], flag, 4, hipMemcpyDeviceToHost, st));
    SCN_HIP(hipEventRecord(hs.ev[0], st));
    SCN_HIP(hipEventSynchronize(hs.ev[0]));
    const int64_t n_rows = hs.words[0];
    const int32_t bad = (int32_t)hs.words[1];
    desc[3] = bad;
    const bool two = (flags & SCN_PYRAMID_TWO_QUEUES) && !scn::sw(scn::SW_PYRAMID_ONE_STREAM).set;
    bool side_used[HostSlots::NSIDE] = {};
    // an error return while side streams still work on the caller's workspace must not hand that workspace back
    struct SideGuard {
        hipStream_t* s; const bool* used; bool armed = true;
        ~SideGuard() {
            if (armed)
                for (int q = 0; q < HostSlots::NSIDE; ++q)
                    if (used[q]) (void)hipStreamSynchronize(s[q]);
        }
    } side_guard{hs.side, side_used};
    if (two) SCN_HIP(hipEventRecord(hs.grid_ev[0], st));            // level 0's grid (row coordinates, hash) is complete
    if (bad) return scn::fail(SCN_EHASH, "%scoordinates outside [0,65535] in %lld wave(s)", "", (long long)bad);

    // ---- levels --------------------------------------------------------------------------------------------------
    const int32_t* lv_coords = row_coords;
    int64_t lv_off_coords = off_coords0, lv_off_keys = off_keys0, lv_off_hrows = off_hrows0, lv_cap = cap0;
    const uint64_t* lv_keys = keys;
    const int32_t* lv_hrows = hrows;
    int64_t n = n_rows;
    int64_t* prefix_dev[SCN_PYRAMID_MAX_LEVELS][2] = {};
    for (int l = 0; l < n_levels; ++l) {
        int64_t* L = desc + 8 + l * SCN_PYRAMID_LEVEL_STRIDE;
        const bool has_next = l + 1 < n_levels && n > 0;
        L[0] = n; L[1] = lv_cap; L[2] = lv_off_coords; L[3] = lv_off_keys; L[4] = lv_off_hrows;
        // numbering of the coarse sites of level l+1 first: its row count travels back while the SubM work is queued
        uint64_t* nkeys = nullptr; int32_t *nhrows = nullptr, *parent = nullptr, *ncoords = nullptr;
        int64_t ncap = 0, off_nkeys = 0, off_nhrows = 0, off_ncoords = 0;
        int64_t* ncnt = nullptr;
        if (has_next) {
            ncap = scn_hash_capacity(n);
            nkeys = (uint64_t*)ws.take(ncap * 8, &off_nkeys);
            nhrows = (int32_t*)ws.take(ncap * 4, &off_nhrows);
            parent = (int32_t*)ws.take(n * 4, &off);           L[14] = off;
            ncoords = (int32_t*)ws.take(n * 16, &off_ncoords);
            void* s2 = ws.take(scn_dedup_scratch_bytes(n), &off);
            ncnt = (int64_t*)ws.take(8, &off);
            SCN_REQUIRE(ws.ok);
            if ((rc = scn_dedup_launch(lv_coords, n, 1, nkeys, nhrows, ncap, parent, nullptr, nullptr, ncoords, s2, ncnt,
                                       stream)))
                return rc;
            // the count starts its way back now; the SubM work of this level is queued behind it
            SCN_HIP(hipMemcpyAsync(&hs.words[2 + l], ncnt, 8, hipMemcpyDeviceToHost, st));
            SCN_HIP(hipEventRecord(hs.ev[1 + l], st));
            if (two) SCN_HIP(hipEventRecord(hs.grid_ev[l + 1], st));    // level l + 1's grid is complete behind this point
        }
        if (n > 0 && k > 1) {
            const int64_t nt = nt_of(n);
            int32_t* table = (int32_t*)ws.take((int64_t)n_off * n * 4, &off);            L[5] = off;
            const int64_t blocks = scn_rules_blocks(n_off, n);
            int32_t* bsums = (int32_t*)ws.take(blocks * 4, &off);                        L[6] = off; L[7] = blocks;
            int64_t* prefix = (int64_t*)ws.take((n_off + 1) * 8, &off);                  L[8] = off;
            int32_t* perm = (int32_t*)ws.take(nt * 16 * 4, &off);                        L[9] = off;
            int32_t* tstab = (int32_t*)ws.take(nt * n_off * 16 * 4, &off);               L[10] = off;
            uint32_t* tmask = (uint32_t*)ws.take(nt * 4, &off);                          L[11] = off;
            // (with SCN_PYRAMID_XCD_ORDER the XCD-local hand-out order and its bin starts sit behind the first order)
            const bool with_x = (flags & SCN_PYRAMID_XCD_ORDER) != 0;
            int32_t* torder = (int32_t*)ws.take(scn_tiles_order_ints(n, with_x ? 1 : 0) * 4, &off);   L[12] = off; L[13] = nt;
            void* tscr = ws.take(scn_tiles_scratch_bytes(n_off, n), &off);
            SCN_REQUIRE(ws.ok);
            scn_stream_t sub = stream;
            if (two) {
                SCN_HIP(hipStreamWaitEvent(hs.side[0], hs.grid_ev[l], 0));
                side_used[0] = true;
                sub = (scn_stream_t)hs.side[0];
            }
            if ((rc = scn_subm_table(lv_coords, n, lv_keys, lv_hrows, lv_cap, k, table, sub))) return rc;
            if ((rc = scn_rules_scan(table, n_off, n, bsums, prefix, nullptr, sub))) return rc;
            // (round 6: with the XCD-local order, level 0's tiles are sorted by (row bin, mask): 8 row bins -- as the fused build does)
            const int lb0 = (with_x && l == 0 && n_off == 27 && !(scn::sw(scn::SW_TB_NO_BINS).set && scn::sw(scn::SW_TB_NO_BINS).i != 0)) ? 3 : 0;
            if ((rc = scn_tiles_build_x(table, n_off, n, perm, tstab, tmask, torder, (with_x ? 1 : 0) | (lb0 << 8), tscr, sub))) return rc;
            prefix_dev[l][0] = prefix;
        }
        if (!has_next) {
            if (l + 1 < n_levels)                       // an empty level: everything below it is empty too
                for (int m = l + 1; m < n_levels; ++m) desc[8 + m * SCN_PYRAMID_LEVEL_STRIDE] = 0;
            break;
        }
        SCN_HIP(hipEventSynchronize(hs.ev[1 + l]));
        const int64_t nc = hs.words[2 + l];
        {
            const int64_t ntc = nt_of(nc);
            int32_t* fine_off = (int32_t*)ws.take(n * 4, &off);                          L[15] = off;
            int32_t* child = (int32_t*)ws.take(8 * nc * 4, &off);                        L[16] = off;
            const int64_t blocks = scn_rules_blocks(8, nc);
            int32_t* bsums = (int32_t*)ws.take(blocks * 4, &off);                        L[17] = off; L[18] = blocks;
            int64_t* prefix = (int64_t*)ws.take(9 * 8, &off);                            L[19] = off;
            int32_t* perm = (int32_t*)ws.take(ntc * 16 * 4, &off);                       L[20] = off;
            int32_t* tstab = (int32_t*)ws.take(ntc * 8 * 16 * 4, &off);                  L[21] = off;
            uint32_t* tmask = (uint32_t*)ws.take(ntc * 4, &off);                         L[22] = off;
            int32_t* torder = (int32_t*)ws.take(ntc * 4, &off);                          L[23] = off; L[24] = ntc;
            void* tscr = ws.take(scn_tiles_scratch_bytes(8, nc), &off);
            SCN_REQUIRE(ws.ok);
            scn_stream_t cs = stream;          // (the strided rulebook stays on the main chain)
            if ((rc = scn_child_table(lv_coords, parent, n, nc, child, fine_off, cs))) return rc;
            if ((rc = scn_rules_scan(child, 8, nc, bsums, prefix, nullptr, cs))) return rc;
            if ((rc = scn_tiles_build(child, 8, nc, perm, tstab, tmask, torder, tscr, cs))) return rc;
            prefix_dev[l][1] = prefix;
        }
        lv_coords = ncoords; lv_keys = nkeys; lv_hrows = nhrows; lv_cap = ncap;
        lv_off_coords = off_ncoords; lv_off_keys = off_nkeys; lv_off_hrows = off_nhrows;
        n = nc;
    }
    for (int q = 0; q < HostSlots::NSIDE; ++q)                // from here on the caller's stream sees the side work too
        if (side_used[q]) {
            SCN_HIP(hipEventRecord(hs.join_ev[q], hs.side[q]));
            SCN_HIP(hipStreamWaitEvent(st, hs.join_ev[q], 0));
        }
    side_guard.armed = false;
    // ---- rule-list sizes (consumed by the weight-gradient / rule-list GEMMs) -----------------------------------------
    for (int l = 0; l < n_levels; ++l) {
        int64_t* L = desc + 8 + l * SCN_PYRAMID_LEVEL_STRIDE;
        if (prefix_dev[l][0])
            SCN_HIP(hipMemcpyAsync(L + 25, prefix_dev[l][0], (n_off + 1) * 8, hipMemcpyDeviceToHost, st));
        if (prefix_dev[l][1]) SCN_HIP(hipMemcpyAsync(L + 53, prefix_dev[l][1], 9 * 8, hipMemcpyDeviceToHost, st));
    }
    SCN_HIP(hipStreamSynchronize(st));
    // ---- compacted rule lists: sizes are known now; no further wait (the caller orders its stream behind this one) ------
    for (int l = 0; l < n_levels; ++l) {
        int64_t* L = desc + 8 + l * SCN_PYRAMID_LEVEL_STRIDE;
        char* base = (char*)workspace;
        if (prefix_dev[l][0]) {
            const int64_t P = L[25 + n_off];
            int32_t* in_rows = (int32_t*)ws.take(P * 4, &off);      L[64] = off;
            int32_t* out_rows = (int32_t*)ws.take(P * 4, &off);     L[65] = off;
            SCN_REQUIRE(ws.ok);
            if ((rc = scn_rules_fill((const int32_t*)(base + L[5]), n_off, L[0], (const int32_t*)(base + L[6]), in_rows,
                                     out_rows, nullptr, stream)))
                return rc;
        }
        if (prefix_dev[l][1]) {
            const int64_t P = L[53 + 8], nc = desc[8 + (l + 1) * SCN_PYRAMID_LEVEL_STRIDE];
            int32_t* in_rows = (int32_t*)ws.take(P * 4, &off);      L[66] = off;
            int32_t* out_rows = (int32_t*)ws.take(P * 4, &off);     L[67] = off;
            SCN_REQUIRE(ws.ok);
            if ((rc = scn_rules_fill((const int32_t*)(base + L[16]), 8, nc, (const int32_t*)(base + L[17]), in_rows,
                                     out_rows, nullptr, stream)))
                return rc;
        }
    }
    desc[2] = ws.used;
    return SCN_OK;
}
