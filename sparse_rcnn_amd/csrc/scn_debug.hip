// Developer switches of libscn_mi355x.  Rounds 1-4 read them with getenv() on every launch of the product
// path (~300 environment scans per step, behaviour that followed ambient variables).  Now: the environment is read ONCE, the
// first time any switch is asked for, into a table; afterwards a switch changes only through scn_debug_set() -- the tests'
// A/B runs inside one process (tests/: `_lib.debug_switch`) -- and a launch reads an int from the table.
#include <stdlib.h>

#include <mutex>

#include "scn_common.h"

namespace scn {
namespace {
struct Entry { const char* name; SwitchVal v; };
Entry g_sw[SW_COUNT] = {
    {"SCN_TS_SPLIT", {}},        {"SCN_TS_SPLIT_MAX", {}},  {"SCN_TS_NO_TAIL", {}},       {"SCN_TS_W_HALF", {}},
    {"SCN_TS_W_BOTH", {}},       {"SCN_TB_NB", {}},         {"SCN_TB_KH", {}},            {"SCN_TB_STREAM", {}},
    {"SCN_TB_NO_XORDER", {}},    {"SCN_TS_STREAM", {}},     {"SCN_TSS_NW", {}},           {"SCN_EXEC_DEFER_SUMS", {}},
    {"SCN_PYRAMID_V1", {}},      {"SCN_PYRAMID_ONE_STREAM", {}}, {"SCN_WD_NO_T3", {}},    {"SCN_WGRAD_BF16_MFMA", {}},
    {"SCN_WGRAD_SPLITS", {}},    {"SCN_WD_NO_EVEC", {}},    {"SCN_PYRAMID_NO_BRICKS", {}},            {"SCN_CU_BUDGET", {}},
    {"SCN_EXP_A", {}},           {"SCN_EXP_B", {}},         {"SCN_TS_NO_CHAIN", {}},
    {"SCN_TS_PROG", {}},         {"SCN_TB_NO_BINS", {}},
};
std::once_flag g_once;
std::mutex g_mu;

void assign(SwitchVal& v, const char* s) {
    v.set = s != nullptr;
    v.i = s ? atoll(s) : 0;
    v.f = s ? atof(s) : 0.0;
}
void load_env() {
    for (int k = 0; k < SW_COUNT; ++k) assign(g_sw[k].v, getenv(g_sw[k].name));
}
}  // namespace

SwitchVal sw(Switch s) {
    std::call_once(g_once, load_env);
    std::lock_guard<std::mutex> lock(g_mu);          // (scn_debug_set writes under the same lock: no torn {set, i, f})
    return g_sw[s].v;
}
}  // namespace scn

extern "C" int scn_debug_set(const char* name, const char* value) {
    SCN_REQUIRE(name != nullptr);
    std::call_once(scn::g_once, scn::load_env);
    std::lock_guard<std::mutex> lock(scn::g_mu);
    for (int k = 0; k < scn::SW_COUNT; ++k)
        if (strcmp(scn::g_sw[k].name, name) == 0) {
            scn::assign(scn::g_sw[k].v, value);
            return SCN_OK;
        }
    return scn::fail(SCN_EINVAL, "scn_debug_set: no switch named %s", name);
}

extern "C" int scn_debug_get(const char* name, int* is_set, int64_t* value) {
    SCN_REQUIRE(name != nullptr);
    for (int k = 0; k < scn::SW_COUNT; ++k)
        if (strcmp(scn::g_sw[k].name, name) == 0) {
            const scn::SwitchVal v = scn::sw((scn::Switch)k);
            if (is_set) *is_set = v.set ? 1 : 0;
            if (value) *value = v.i;
            return SCN_OK;
        }
    return scn::fail(SCN_EINVAL, "scn_debug_get: no switch named %s", name);
}
