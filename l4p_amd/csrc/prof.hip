#include "prof.hpp"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "common.hpp"

bool g_prof_on = false;

namespace {
struct KnobDef {
    const char* name;
    const char* env;
    int dflt;
};
const KnobDef kKnobs[KNOB_NUM] = {{"conv_halo", "L4P_CONV_HALO", 1}, {"gemm_4w", "L4P_GEMM_4W", 0},
                                   {"maskdot_mfma", "L4P_MASKDOT_MFMA", 1},
                                   {"conv_ups", "L4P_CONV_UPS", 0},
                                   {"ln_tracks", "L4P_LN_TRACKS", 1},
                                   {"ln_rows16", "L4P_LN_ROWS16", 1},
                                   {"attn64", "L4P_ATTN64", 1},
                                   {"gemm_skinny", "L4P_GEMM_SKINNY", 1},
                                   {"readout_wide", "L4P_READOUT_WIDE", 1},
                                   {"track_deep", "L4P_TRACK_DEEP", 1},
                                   {"probe_kernels", "", 0}};
std::atomic<int> g_knob[KNOB_NUM];
std::once_flag g_knob_once;
#ifdef L4P_PROBE_KERNELS
constexpr bool kProbeKernels = true;
#else
constexpr bool kProbeKernels = false;
#endif
// knobs that select a measured-and-not-adopted kernel: they exist only in a PROBES=1 build of the library
bool probe_only(int i) { return i == KNOB_GEMM_4W || i == KNOB_CONV_UPS; }
void knobs_init() {
    for (int i = 0; i < KNOB_NUM; ++i) {
        const char* e = kKnobs[i].env[0] ? getenv(kKnobs[i].env) : nullptr;
        int v = e ? atoi(e) : kKnobs[i].dflt;
        if (i == KNOB_PROBE_KERNELS) v = kProbeKernels ? 1 : 0;  // read-only: what this build contains
        if (probe_only(i) && !kProbeKernels) v = 0;
        g_knob[i].store(v, std::memory_order_relaxed);
    }
}
}  // namespace
int knob(int id) {
    std::call_once(g_knob_once, knobs_init);
    return g_knob[id].load(std::memory_order_relaxed);
}

namespace {
struct Pair {
    hipEvent_t a, b;
    int cls;
    char tag[64];
};
std::mutex g_mu;
std::vector<Pair> g_pairs;      // recorded this session
std::vector<hipEvent_t> g_pool; // reusable events
const char* kNames[PROF_NUM] = {"gemm", "conv3d", "attention", "layernorm", "elementwise", "track", "preprocess", "gemm_small"};

hipEvent_t get_event() {
    if (!g_pool.empty()) {
        hipEvent_t e = g_pool.back();
        g_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace

void prof_begin(int cls, hipStream_t stream, const char* tag) {
    std::lock_guard<std::mutex> lk(g_mu);
    Pair p{get_event(), get_event(), cls, {0}};
    if (tag) snprintf(p.tag, sizeof p.tag, "%s", tag);
    (void)hipEventRecord(p.a, stream);
    g_pairs.push_back(p);
}
void prof_end(int cls, hipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (size_t i = g_pairs.size(); i-- > 0;)
        if (g_pairs[i].cls == cls) {
            (void)hipEventRecord(g_pairs[i].b, stream);
            return;
        }
}

extern "C" {
int l4p_set_knob(const char* name, int value) {
    std::call_once(g_knob_once, knobs_init);
    for (int i = 0; i < KNOB_NUM; ++i)
        if (name && !strcmp(name, kKnobs[i].name)) {
            if (i == KNOB_PROBE_KERNELS || (probe_only(i) && !kProbeKernels && value != 0)) {
                l4p_set_error("l4p_set_knob: '%s' %s", name, i == KNOB_PROBE_KERNELS ? "is read-only (1 in a PROBES=1 build of the library)"
                                                                                       : "selects a kernel this build does not contain (make PROBES=1)");
                return L4P_E_INVALID;
            }
            g_knob[i].store(value, std::memory_order_relaxed);
            return L4P_OK;
        }
    l4p_set_error("l4p_set_knob: unknown knob '%s'", name ? name : "(null)");
    return L4P_E_INVALID;
}
int l4p_get_knob(const char* name) {
    for (int i = 0; i < KNOB_NUM; ++i)
        if (name && !strcmp(name, kKnobs[i].name)) return knob(i);
    return -1;
}

int l4p_prof_enable(int on) {
    g_prof_on = on != 0;
    return 0;
}
int l4p_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& p : g_pairs) {
        g_pool.push_back(p.a);
        g_pool.push_back(p.b);
    }
    g_pairs.clear();
    return 0;
}
int l4p_prof_num_classes(void) { return PROF_NUM; }
const char* l4p_prof_class_name(int cls) { return cls >= 0 && cls < PROF_NUM ? kNames[cls] : ""; }
// Sum of event-pair durations of one class since the last reset.  The caller must have synchronised
// the stream(s) the kernels ran on.
int l4p_prof_read(int cls, double* total_ms, long long* count) {
    std::lock_guard<std::mutex> lk(g_mu);
    double t = 0;
    long long n = 0;
    for (auto& p : g_pairs) {
        if (p.cls != cls) continue;
        float ms = 0.f;
        hipError_t e = hipEventElapsedTime(&ms, p.a, p.b);
        if (e != hipSuccess) {
            l4p_set_error("l4p_prof_read: %s (stream not synchronised?)", hipGetErrorString(e));
            return L4P_E_HIP;
        }
        t += ms;
        ++n;
    }
    if (total_ms) *total_ms = t;
    if (count) *count = n;
    return 0;
}
// Per-(class, tag) table of the event pairs since the last reset, as text lines "class<TAB>tag<TAB>count<TAB>total_ms",
// sorted by total time.  Returns the number of bytes needed (including the terminator); writes at most cap bytes.
long long l4p_prof_detail(char* buf, long long cap) {
    std::lock_guard<std::mutex> lk(g_mu);
    std::map<std::string, std::pair<long long, double>> agg;
    for (auto& p : g_pairs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) != hipSuccess) continue;
        auto& e = agg[std::string(kNames[p.cls]) + "\t" + p.tag];
        e.first++;
        e.second += ms;
    }
    std::vector<std::pair<double, std::string>> rows;
    for (auto& kv : agg) {
        char line[192];
        snprintf(line, sizeof line, "%s\t%lld\t%.4f\n", kv.first.c_str(), kv.second.first, kv.second.second);
        rows.emplace_back(-kv.second.second, line);
    }
    std::sort(rows.begin(), rows.end());
    std::string out;
    for (auto& r : rows) out += r.second;
    if (buf && cap > 0) snprintf(buf, (size_t)cap, "%s", out.c_str());
    return (long long)out.size() + 1;
}
}
