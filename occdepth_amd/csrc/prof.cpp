// Launch-time kernel timing with HIP events recorded on the launch stream.
// bench.py turns it on around the timed region; the per-tag totals feed the
// `roofline` object (achieved = algorithmic flops or bytes / measured time).
#include <map>
#include <set>
#include <utility>
#include <mutex>
#include <string>
#include <vector>
#include <cstring>
#include "common.h"

namespace {
struct Rec {
    hipEvent_t e0, e1;
    std::string tag;
    double flops, bytes;
};
std::mutex g_mu;
bool g_on = false;
std::string g_tag;
std::string g_tag_kind;      // the kernel kind the current tag was consumed by first: a tag describes ONE kind of launch
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;

hipEvent_t get_event() {
    if (!g_pool.empty()) {
        hipEvent_t e = g_pool.back();
        g_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace

namespace occd {
ProfScope::ProfScope(const char* kind, hipStream_t s, double flops, double bytes) : slot(-1), stream(s) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_on) return;
    Rec r;
    r.e0 = get_event();
    r.e1 = get_event();
    // A tag is the geometry string of the operator that set it.  Kernels launched later by operators that set no tag
    // used to inherit it ("dwconv2d_nchw:32>22 k333 ... @256x256x32": a depthwise launch labelled with the head
    // convolution's geometry); now the tag belongs to the first kind that uses it, other kinds are reported bare.
    bool tagged = !g_tag.empty();
    if (tagged) {
        if (g_tag_kind.empty()) g_tag_kind = kind;
        else if (g_tag_kind != kind) tagged = false;
    }
    r.tag = tagged ? std::string(kind) + ":" + g_tag : std::string(kind);
    r.flops = flops;
    r.bytes = bytes;
    (void)hipEventRecord(r.e0, s);
    g_recs.push_back(r);
    slot = (int)g_recs.size() - 1;
}
ProfScope::~ProfScope() {
    std::lock_guard<std::mutex> lk(g_mu);
    if (slot < 0 || slot >= (int)g_recs.size()) return;
    (void)hipEventRecord(g_recs[slot].e1, stream);
}
}  // namespace occd

namespace occd {
int ensure_big_lds(const void* kernel) {
    static std::mutex mu;
    static std::set<std::pair<int, const void*>> done;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return OCCD_ELAUNCH;
    std::lock_guard<std::mutex> lk(mu);
    if (done.count({dev, kernel})) return OCCD_OK;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return OCCD_ELAUNCH;
    done.insert({dev, kernel});
    return OCCD_OK;
}
}  // namespace occd

extern "C" int occd_prof_enable(int32_t on) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_on = on != 0;
    return OCCD_OK;
}

extern "C" int occd_prof_set_tag(const char* tag) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_tag = tag ? tag : "";
    g_tag_kind.clear();
    return OCCD_OK;
}

extern "C" int occd_prof_report(occd_prof_row* rows, int32_t max_rows) {
    std::lock_guard<std::mutex> lk(g_mu);
    std::map<std::string, occd_prof_row> agg;
    for (auto& r : g_recs) {
        float ms = 0.f;
        if (hipEventSynchronize(r.e1) != hipSuccess) return OCCD_ELAUNCH;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) return OCCD_ELAUNCH;
        auto it = agg.find(r.tag);
        if (it == agg.end()) {
            occd_prof_row row;
            memset(&row, 0, sizeof(row));
            strncpy(row.tag, r.tag.c_str(), sizeof(row.tag) - 1);
            it = agg.insert({r.tag, row}).first;
        }
        it->second.launches += 1;
        it->second.ms += ms;
        it->second.flops += r.flops;
        it->second.bytes += r.bytes;
        g_pool.push_back(r.e0);
        g_pool.push_back(r.e1);
    }
    g_recs.clear();
    int n = 0;
    for (auto& kv : agg) {
        if (rows && n < max_rows) rows[n] = kv.second;
        ++n;
    }
    return n;
}

extern "C" int occd_abi_version(void) { return 14; }

extern "C" const char* occd_strerror(int code) {
    switch (code) {
        case OCCD_OK: return "ok";
        case OCCD_EINVAL: return "invalid argument or unsupported geometry";
        case OCCD_ELAUNCH: return "HIP launch/runtime error";
        case OCCD_ENOMEM: return "tile does not fit in LDS";
        default: return "unknown error";
    }
}
