// Step executor: one C call walks the launch plan of a whole network pass (include/scn_mi355x.h, "Step executor").
//
// Why: the reference drives the scn surface layer by layer from Python; this package's autograd functions did the same.  A
// backbone step is ~330 launches, a detection + mask step ~760, and every launch cost ~13 us of host time of which only
// 4-5 us are hipLaunchKernel -- the bf16-storage steps (BASELINE configs 3-5) were bound by the interpreter, not by the
// GPU.  Here the per-layer calls are made from a flat op list: no interpreter, no allocator, no autograd
// node per layer.  Every op IS one of the library's own entry points with the arguments the layer-by-layer path passes, so
// the two paths produce the same bits (tests/test_gpu_exec.py).
#include <stdlib.h>
#include <vector>

#include <atomic>
#include <mutex>
#include <vector>

#include "scn_common.h"

using scn::cdiv;

namespace {

// Scratch of a pass: [shared region: the largest need of any op that is not a deferred weight gradient][one region per
// weight-gradient op of the pass].  With deferred sums (below) a weight-gradient launch keeps its unit slabs until the
// batched sum at the end of the pass, so it cannot share them with the next launch.
struct Ctx {
    const scn_exec_level* levels;
    int n_levels;
    void* const* bufs;
    const void* const* params;
    void* const* grads;
    void* scratch;
    int32_t* arrival;
    scn_stream_t stream;
};

inline int64_t align256(int64_t b) { return (b + 255) & ~(int64_t)255; }

// scratch bytes of a weight-gradient op (0: not one)
int64_t wgrad_scratch_of(const scn_exec_op& o, const scn_exec_level& L) {
    switch (o.op) {
    case SCN_OP_WGRAD_SUBM: return scn_wgrad_scratch_bytes(o.cin, o.cout, L.prefix_host, 27);
    case SCN_OP_WGRAD2_SUBM: return scn_wgrad_scratch_bytes2(o.cin, o.cout, L.prefix_host, 27);
    case SCN_OP_WGRAD_DOWN:
    case SCN_OP_WGRAD_UP: return scn_wgrad_scratch_bytes(o.cin, o.cout, L.c_prefix_host, 8);
    case SCN_OP_WGRAD_IDENT: {
        const int64_t ident[2] = {0, (o.flags & SCN_XF_COARSE_ROWS) ? L.n_coarse : L.n};
        return scn_wgrad_scratch_bytes(o.cin, o.cout, ident, 1);
    }
    default: return 0;
    }
}

inline bool bf(const scn_exec_op& o) { return (o.flags & SCN_XF_BF16) != 0; }
inline int call_flags(const scn_exec_op& o) { return o.flags & 0xffff; }
inline int64_t rows_of(const scn_exec_op& o, const scn_exec_level& L) {
    return (o.flags & SCN_XF_COARSE_ROWS) ? L.n_coarse : L.n;
}

template <typename T>
inline T* B(const Ctx& c, int id) { return id < 0 ? nullptr : (T*)c.bufs[id]; }
template <typename T>
inline const T* P(const Ctx& c, int id) { return id < 0 ? nullptr : (const T*)c.params[id]; }
template <typename T>
inline T* G(const Ctx& c, int id) { return id < 0 ? nullptr : (T*)c.grads[id]; }

// own_scratch: the op's own scratch region (a weight gradient whose sum is deferred), NULL: the pass's shared region
int run_op(const Ctx& c0, const scn_exec_op& o, void* own_scratch = nullptr) {
    SCN_REQUIRE(o.level >= 0 && o.level < c0.n_levels);
    const scn_exec_level& L = c0.levels[o.level];
    Ctx c = c0;
    if (own_scratch) c.scratch = own_scratch;
    const int fl = call_flags(o);
    const bool h = bf(o);
    scn_stream_t st = c.stream;
    typedef uint16_t u16;
    switch (o.op) {
    case SCN_OP_GEMM_IDENT: {
        const int64_t n = rows_of(o, L);
        if (h) return scn_gemm_table_bf16(B<u16>(c, o.x), n, o.cin, nullptr, 1, n, P<float>(c, o.w), P<float>(c, o.b),
                                          B<u16>(c, o.r), B<u16>(c, o.m), B<u16>(c, o.y), o.cout, fl, st);
        return scn_gemm_table(B<float>(c, o.x), n, o.cin, nullptr, 1, n, P<float>(c, o.w), P<float>(c, o.b), B<float>(c, o.r),
                              B<float>(c, o.m), B<float>(c, o.y), o.cout, fl, st);
    }
    case SCN_OP_CONV_SUBM:
    case SCN_OP_CONV_CHILD: {
        const bool child = o.op == SCN_OP_CONV_CHILD;
        const int64_t n_in = L.n, n_out = child ? L.n_coarse : L.n;
        const int n_off = child ? 8 : 27;
        const int32_t* tstab = child ? L.c_tstab : L.tstab;
        const uint32_t* tmask = child ? L.c_tile_mask : L.tile_mask;
        const int32_t* perm = child ? L.c_perm : L.perm;
        const int32_t* order = child ? L.c_tile_order : L.tile_order;
        if (h) {
            int32_t* arr = scn_conv_tiles_bf16_arrival_counters(o.cin, n_out, o.cout) ? c.arrival : nullptr;
            const int flx = fl | ((!child && (L.flags & SCN_XL_TILE_ORDER_X)) ? SCN_F_TILE_ORDER_X : 0);
            return scn_conv_tiles_bf16(B<u16>(c, o.x), n_in, o.cin, tstab, tmask, perm, order, n_off, n_out, P<u16>(c, o.w),
                                       P<float>(c, o.b), B<u16>(c, o.r), B<u16>(c, o.m), B<u16>(c, o.y), o.cout, flx, c.scratch,
                                       arr, st);
        }
        int32_t* arr = o.cin > 32 ? c.arrival : nullptr;
        return scn_conv_tiles(B<float>(c, o.x), n_in, o.cin, tstab, tmask, perm, order, n_off, n_out, P<float>(c, o.w),
                              P<float>(c, o.b), B<float>(c, o.r), B<float>(c, o.m), B<float>(c, o.y), o.cout, fl, c.scratch, arr,
                              st);
    }
    case SCN_OP_RULES_CHILD: {       // coarse rows -> fine rows: the strided rules with their roles swapped
        if (h) return scn_gemm_rules_bf16(B<u16>(c, o.x), o.cin, L.c_out_rows, L.c_in_rows, L.c_prefix_host, 8, P<float>(c, o.w),
                                          P<float>(c, o.b), B<u16>(c, o.m), B<u16>(c, o.y), o.cout, fl, st);
        return scn_gemm_rules(B<float>(c, o.x), o.cin, L.c_out_rows, L.c_in_rows, L.c_prefix_host, 8, P<float>(c, o.w),
                              P<float>(c, o.b), B<float>(c, o.m), B<float>(c, o.y), o.cout, fl, st);
    }
    case SCN_OP_ROWS2: {
        const int64_t n = rows_of(o, L);
        if (o.y1 >= 0)               // two destinations: backward-data of the NiN over a JoinTable
            return scn_gemm_rows2(B<void>(c, o.x), o.cin, nullptr, 0, n, P<float>(c, o.w), nullptr, nullptr, nullptr,
                                  B<void>(c, o.y), o.cout, B<void>(c, o.y1), o.c1, fl, h ? 1 : 0, st);
        return scn_gemm_rows2(B<void>(c, o.x), o.cin, B<void>(c, o.x1), o.c1, n, P<float>(c, o.w), P<float>(c, o.b), nullptr,
                              nullptr, B<void>(c, o.y), o.cout, nullptr, 0, fl, h ? 1 : 0, st);
    }
    case SCN_OP_WGRAD_SUBM: {
        const uint32_t dbm = o.b >= 0 ? (1u << 13) : 0u;
        if (h) return scn_wgrad_bias_rules_bf16(B<u16>(c, o.x), o.cin, B<u16>(c, o.y), o.cout, L.in_rows, L.out_rows, L.prefix_host,
                                                27, G<float>(c, o.w), G<float>(c, o.b), dbm, c.scratch, fl, st);
        return scn_wgrad_bias_rules(B<float>(c, o.x), o.cin, B<float>(c, o.y), o.cout, L.in_rows, L.out_rows, L.prefix_host, 27,
                                    G<float>(c, o.w), G<float>(c, o.b), dbm, c.scratch, fl, st);
    }
    case SCN_OP_WGRAD2_SUBM: {
        const uint32_t dbm = o.b >= 0 ? (1u << 13) : 0u;
        if (h) return scn_wgrad_bias_rules2_bf16(B<u16>(c, o.x), B<u16>(c, o.y), B<u16>(c, o.x1), B<u16>(c, o.y1), o.cin, o.cout,
                                                 L.in_rows, L.out_rows, L.prefix_host, 27, G<float>(c, o.w), G<float>(c, o.b), dbm,
                                                 c.scratch, fl, st);
        return scn_wgrad_bias_rules2(B<float>(c, o.x), B<float>(c, o.y), B<float>(c, o.x1), B<float>(c, o.y1), o.cin, o.cout,
                                     L.in_rows, L.out_rows, L.prefix_host, 27, G<float>(c, o.w), G<float>(c, o.b), dbm, c.scratch,
                                     fl, st);
    }
    case SCN_OP_WGRAD_DOWN: {        // Convolution: X fine rows, dY coarse rows
        if (h) return scn_wgrad_rules_bf16(B<u16>(c, o.x), o.cin, B<u16>(c, o.y), o.cout, L.c_in_rows, L.c_out_rows,
                                           L.c_prefix_host, 8, G<float>(c, o.w), c.scratch, fl, st);
        return scn_wgrad_rules(B<float>(c, o.x), o.cin, B<float>(c, o.y), o.cout, L.c_in_rows, L.c_out_rows, L.c_prefix_host, 8,
                               G<float>(c, o.w), c.scratch, fl, st);
    }
    case SCN_OP_WGRAD_UP: {          // Deconvolution: X coarse rows, dY fine rows; every fine row occurs in exactly one list
        if (o.b < 0) {
            if (h) return scn_wgrad_rules_bf16(B<u16>(c, o.x), o.cin, B<u16>(c, o.y), o.cout, L.c_out_rows, L.c_in_rows,
                                               L.c_prefix_host, 8, G<float>(c, o.w), c.scratch, fl, st);
            return scn_wgrad_rules(B<float>(c, o.x), o.cin, B<float>(c, o.y), o.cout, L.c_out_rows, L.c_in_rows, L.c_prefix_host,
                                   8, G<float>(c, o.w), c.scratch, fl, st);
        }
        if (h) return scn_wgrad_bias_rules_bf16(B<u16>(c, o.x), o.cin, B<u16>(c, o.y), o.cout, L.c_out_rows, L.c_in_rows,
                                                L.c_prefix_host, 8, G<float>(c, o.w), G<float>(c, o.b), 0xFFu, c.scratch, fl, st);
        return scn_wgrad_bias_rules(B<float>(c, o.x), o.cin, B<float>(c, o.y), o.cout, L.c_out_rows, L.c_in_rows, L.c_prefix_host,
                                    8, G<float>(c, o.w), G<float>(c, o.b), 0xFFu, c.scratch, fl, st);
    }
    case SCN_OP_WGRAD_IDENT: {
        const int64_t ident[2] = {0, rows_of(o, L)};
        float* dW = G<float>(c, o.w);
        SCN_REQUIRE(dW != nullptr);
        dW += o.aux;
        if (o.b < 0) {
            if (h) return scn_wgrad_rules_bf16(B<u16>(c, o.x), o.cin, B<u16>(c, o.y), o.cout, nullptr, nullptr, ident, 1, dW,
                                               c.scratch, fl, st);
            return scn_wgrad_rules(B<float>(c, o.x), o.cin, B<float>(c, o.y), o.cout, nullptr, nullptr, ident, 1, dW, c.scratch, fl,
                                   st);
        }
        if (h) return scn_wgrad_bias_rules_bf16(B<u16>(c, o.x), o.cin, B<u16>(c, o.y), o.cout, nullptr, nullptr, ident, 1, dW,
                                                G<float>(c, o.b), 1u, c.scratch, fl, st);
        return scn_wgrad_bias_rules(B<float>(c, o.x), o.cin, B<float>(c, o.y), o.cout, nullptr, nullptr, ident, 1, dW,
                                    G<float>(c, o.b), 1u, c.scratch, fl, st);
    }
    case SCN_OP_COLSUM: {
        const int64_t n = rows_of(o, L);
        if (h) return scn_colsum_bf16(B<u16>(c, o.x), n, o.cin, G<float>(c, o.b), c.scratch, st);
        return scn_colsum(B<float>(c, o.x), n, o.cin, G<float>(c, o.b), c.scratch, st);
    }
    case SCN_OP_ADD: {
        const int64_t count = rows_of(o, L) * o.cin;
        if (h) return scn_add_bf16(B<u16>(c, o.x), B<u16>(c, o.x1), count, B<u16>(c, o.y), st);
        return scn_add(B<float>(c, o.x), B<float>(c, o.x1), count, B<float>(c, o.y), st);
    }
    case SCN_OP_CAST: {
        const int64_t count = rows_of(o, L) * o.cin;
        if (h) return scn_cast_f32_to_bf16(B<float>(c, o.x), count, B<u16>(c, o.y), st);
        return scn_cast_bf16_to_f32(B<u16>(c, o.x), count, B<float>(c, o.y), st);
    }
    default:
        return scn::fail(SCN_EINVAL, "scn_exec_run: unknown op %s%lld", "", o.op);
    }
}

}  // namespace

extern "C" int64_t scn_exec_struct_bytes(int which) {
    return which == 0 ? (int64_t)sizeof(scn_exec_op) : (which == 1 ? (int64_t)sizeof(scn_exec_level) : -1);
}

namespace {
// Needs of a pass: the largest scratch of an op that is NOT a weight gradient (shared region), the sum of the
// weight-gradient regions, the largest weight-gradient region, the arrival counters.
int pass_needs(const scn_exec_op* ops, int n_ops, const scn_exec_level* levels, int n_levels, int64_t* shared, int64_t* wg_sum,
               int64_t* wg_max, int* n_wg, int64_t* arrival) {
    int64_t sb = 256, ac = 0, ws = 0, wm = 0;
    int nw = 0;
    for (int i = 0; i < n_ops; ++i) {
        const scn_exec_op& o = ops[i];
        SCN_REQUIRE(o.level >= 0 && o.level < n_levels);
        const scn_exec_level& L = levels[o.level];
        int64_t s = 0, a = 0;
        switch (o.op) {
        case SCN_OP_CONV_SUBM:
        case SCN_OP_CONV_CHILD: {
            const int64_t n_out = o.op == SCN_OP_CONV_CHILD ? L.n_coarse : L.n;
            if (bf(o)) {
                s = scn_conv_tiles_bf16_scratch_bytes(o.cin, n_out, o.cout);
                a = scn_conv_tiles_bf16_arrival_counters(o.cin, n_out, o.cout);
            } else {
                s = scn_conv_tiles_scratch_bytes(o.cin, n_out, o.cout);
                a = scn_conv_tiles_arrival_counters(o.cin, n_out, o.cout);
            }
            break;
        }
        case SCN_OP_COLSUM: s = (int64_t)SCN_COLSUM_BLOCKS * o.cin * (int64_t)sizeof(float); break;
        default: {
            const int64_t w = wgrad_scratch_of(o, L);
            if (w < 0) return scn::fail(SCN_EINVAL, "scn_exec: op %s%lld has bad arguments", "", i);
            if (w > 0) { ws += align256(w); ++nw; if (w > wm) wm = w; }
            break;
        }
        }
        if (s < 0) return scn::fail(SCN_EINVAL, "scn_exec: op %s%lld has bad arguments", "", i);
        if (s > sb) sb = s;
        if (a > ac) ac = a;
    }
    *shared = sb; *wg_sum = ws; *wg_max = wm; *n_wg = nw; *arrival = ac;
    return SCN_OK;
}
}  // namespace

extern "C" int scn_exec_requirements(const scn_exec_op* ops, int n_ops, const scn_exec_level* levels, int n_levels,
                                     int64_t* scratch_bytes, int64_t* arrival_counters) {
    SCN_REQUIRE(n_ops >= 0 && (n_ops == 0 || ops) && levels && n_levels >= 1 && scratch_bytes && arrival_counters);
    int64_t shared, wg_sum, wg_max, ac;
    int n_wg;
    const int rc = pass_needs(ops, n_ops, levels, n_levels, &shared, &wg_sum, &wg_max, &n_wg, &ac);
    if (rc != SCN_OK) return rc;
    // shared region + one region per weight-gradient op (deferred sums); never less than the largest single op
    int64_t sb = align256(shared) + wg_sum;
    if (wg_max > sb) sb = wg_max;
    *scratch_bytes = sb;
    *arrival_counters = ac;
    return SCN_OK;
}

namespace {
// Events for the fork / join of the side stream: created once per calling thread, reused by every call.
struct EventPool {
    static constexpr int N = 32;
    hipEvent_t ev[N] = {};
    int next = 0;
    bool ok = false;
    bool init() {
        if (ok) return true;
        for (auto& e : ev)
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return false;
        return ok = true;
    }
    hipEvent_t take() { hipEvent_t e = ev[next]; next = (next + 1) % N; return e; }
};
thread_local EventPool g_events;

// ---- launch timing inside a pass (scn_exec_timing_enable / _collect): bench.py samples the dominant kernel with HIP events;
// with the executor on, a level's launches are inside ONE C call, so the call itself brackets its tile-convolution ops.
// Process-wide (the backward pass of a node runs on autograd's thread), guarded by a mutex; events are created on demand and
// kept for the life of the process.
struct TimingRec { hipEvent_t a, b; int64_t info[7]; };
struct Timing {
    std::mutex mu;
    std::atomic<bool> on{false}, all{false};        // read outside the mutex on every op of a pass
    std::vector<hipEvent_t> pool;
    size_t used = 0;
    std::vector<TimingRec> recs;
    hipEvent_t take() {
        if (used == pool.size()) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess) return nullptr;
            pool.push_back(e);
        }
        return pool[used++];
    }
};
Timing g_timing;

inline bool is_wgrad(int op) {
    return op == SCN_OP_WGRAD_SUBM || op == SCN_OP_WGRAD2_SUBM || op == SCN_OP_WGRAD_DOWN || op == SCN_OP_WGRAD_UP ||
           op == SCN_OP_WGRAD_IDENT;
}
inline bool is_leaf(int op) {
    return op == SCN_OP_WGRAD_SUBM || op == SCN_OP_WGRAD2_SUBM || op == SCN_OP_WGRAD_DOWN || op == SCN_OP_WGRAD_UP ||
           op == SCN_OP_WGRAD_IDENT || op == SCN_OP_COLSUM;
}
}  // namespace

extern "C" int scn_exec_run_streams(const scn_exec_op* ops, int n_ops, const scn_exec_level* levels, int n_levels,
                                    void* const* bufs, const void* const* params, void* const* grads, void* scratch,
                                    int64_t scratch_bytes, int32_t* arrival, scn_stream_t stream, scn_stream_t side_stream,
                                    void* side_scratch, int64_t side_scratch_bytes) {
    SCN_REQUIRE(n_ops >= 0 && (n_ops == 0 || ops) && levels && n_levels >= 1 && bufs && scratch && scratch_bytes >= 256);
    const bool fork = side_stream != nullptr && side_stream != stream && side_scratch != nullptr && side_scratch_bytes >= scratch_bytes;
    if (fork && !g_events.init()) return scn::fail(SCN_EHIP, "%sevents for the side stream could not be created", "");
    // Deferred sums: the unit sums of the pass's weight-gradient launches run as ONE batched launch at its end (7 us
    // launches of a few hundred workgroups each, 27 per backbone step; SCN_EXEC_DEFER_SUMS=0: one sum per launch).  Needs
    // the per-op scratch regions scn_exec_requirements counts; a caller that sized the scratch otherwise keeps the old form.
    const bool defer_env = !(scn::sw(scn::SW_EXEC_DEFER_SUMS).set && scn::sw(scn::SW_EXEC_DEFER_SUMS).i == 0);
    // (the weight-gradient needs are computed once per pass and only for passes that have such ops: backward passes)
    int n_wg = 0;
    for (int i = 0; i < n_ops; ++i) n_wg += is_wgrad(ops[i].op) ? 1 : 0;
    std::vector<int64_t> own(defer_env && !fork && n_wg >= 2 ? n_ops : 0, 0);      // offset of an op's own region, 0: none
    bool defer = false;
    if (!own.empty()) {
        int64_t shared, wg_sum, wg_max, ac;
        int nw;
        const int rc_needs = pass_needs(ops, n_ops, levels, n_levels, &shared, &wg_sum, &wg_max, &nw, &ac);
        if (rc_needs != SCN_OK) return rc_needs;
        if (align256(shared) + wg_sum <= scratch_bytes) {
            int64_t cursor = align256(shared);
            for (int i = 0; i < n_ops; ++i) {
                const int64_t w = is_wgrad(ops[i].op) ? wgrad_scratch_of(ops[i], levels[ops[i].level]) : 0;
                if (w > 0) { own[i] = cursor; cursor += align256(w); }
            }
            defer = true;
        }
    }
    Ctx c{levels, n_levels, bufs, params, grads, scratch, arrival, stream};
    Ctx cs{levels, n_levels, bufs, params, grads, side_scratch, arrival, side_stream};
    if (defer) SCN_REQUIRE(scn_wgrad_defer_begin() == SCN_OK);
    bool side_used = false;
    for (int i = 0; i < n_ops; ++i) {
        int rc;
        if (fork && is_leaf(ops[i].op)) {
            // a parameter-gradient op is a leaf of the pass: it reads slabs the main stream has produced and nothing reads
            // its result before the pass ends -- it runs beside the backward-data chain, behind an event of the main stream
            hipEvent_t e = g_events.take();
            SCN_HIP(hipEventRecord(e, scn::S(stream)));
            SCN_HIP(hipStreamWaitEvent(scn::S(side_stream), e, 0));
            rc = run_op(cs, ops[i]);
            side_used = true;
        } else if (g_timing.on && (g_timing.all || ops[i].op == SCN_OP_CONV_SUBM || ops[i].op == SCN_OP_CONV_CHILD)) {
            std::lock_guard<std::mutex> lock(g_timing.mu);
            const scn_exec_op& o = ops[i];
            const scn_exec_level& L = levels[o.level];
            const bool child = o.op == SCN_OP_CONV_CHILD;
            TimingRec r{};
            r.a = g_timing.take();
            r.b = g_timing.take();
            if (r.a && r.b) SCN_HIP(hipEventRecord(r.a, scn::S(stream)));
            rc = run_op(c, ops[i], defer && own[i] ? (char*)scratch + own[i] : nullptr);
            if (r.a && r.b) {
                SCN_HIP(hipEventRecord(r.b, scn::S(stream)));
                // op, bf16, cin, cout, rows in, rows out, rules (a child table holds every fine row once)
                r.info[0] = o.op; r.info[1] = bf(o) ? 1 : 0; r.info[2] = o.cin; r.info[3] = o.cout;
                r.info[4] = L.n; r.info[5] = child ? L.n_coarse : L.n;
                r.info[6] = child ? L.n : (L.prefix_host ? L.prefix_host[27] - L.prefix_host[0] : 0);
                g_timing.recs.push_back(r);
            }
        } else {
            rc = run_op(c, ops[i], defer && own[i] ? (char*)scratch + own[i] : nullptr);
        }
        if (rc != SCN_OK) {
            char inner[400];
            snprintf(inner, sizeof(inner), "%s", scn::g_err);
            snprintf(scn::g_err, sizeof(scn::g_err), "scn_exec_run: op %d (kind %d, level %d): %s", i, ops[i].op, ops[i].level,
                     inner);
            if (defer) (void)scn_wgrad_defer_flush(stream);
            return rc;
        }
    }
    if (defer) {
        const int rc = scn_wgrad_defer_flush(stream);
        if (rc != SCN_OK) return rc;
    }
    if (side_used) {                  // join: whatever follows on the main stream sees the gradients (and may reuse the slabs)
        hipEvent_t e = g_events.take();
        SCN_HIP(hipEventRecord(e, scn::S(side_stream)));
        SCN_HIP(hipStreamWaitEvent(scn::S(stream), e, 0));
    }
    return SCN_OK;
}

extern "C" int scn_exec_timing_enable(int on) {
    std::lock_guard<std::mutex> lock(g_timing.mu);
    g_timing.on = on != 0;
    g_timing.all = on == 2;          // 2: every op of a pass (tools/exec_launch_table.py); deferred unit sums are not inside any op
    return SCN_OK;
}

extern "C" int64_t scn_exec_timing_collect(float* ms, int64_t* info, int64_t cap) {
    std::lock_guard<std::mutex> lock(g_timing.mu);
    int64_t n = 0;
    size_t taken = 0;
    // Only the records handed out (or unreadable) leave the list: a caller with a smaller buffer calls again and gets the
    // rest.  The event pool is recycled once no record names an event any more.
    for (; taken < g_timing.recs.size() && n < cap; ++taken) {
        const TimingRec& r = g_timing.recs[taken];
        float t = 0.f;
        if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) continue;
        ms[n] = t;
        for (int j = 0; j < 7; ++j) info[7 * n + j] = r.info[j];
        ++n;
    }
    g_timing.recs.erase(g_timing.recs.begin(), g_timing.recs.begin() + (ptrdiff_t)taken);
    if (g_timing.recs.empty()) g_timing.used = 0;
    return n;
}

extern "C" int scn_exec_run(const scn_exec_op* ops, int n_ops, const scn_exec_level* levels, int n_levels, void* const* bufs,
                            const void* const* params, void* const* grads, void* scratch, int64_t scratch_bytes,
                            int32_t* arrival, scn_stream_t stream) {
    return scn_exec_run_streams(ops, n_ops, levels, n_levels, bufs, params, grads, scratch, scratch_bytes, arrival, stream,
                                nullptr, nullptr, 0);
}
