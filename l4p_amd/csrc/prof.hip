#include "prof.hpp"

#include <mutex>
#include <vector>

#include "common.hpp"

bool g_prof_on = false;

namespace {
struct Pair {
    hipEvent_t a, b;
    int cls;
};
std::mutex g_mu;
std::vector<Pair> g_pairs;      // recorded this session
std::vector<hipEvent_t> g_pool; // reusable events
const char* kNames[PROF_NUM] = {"gemm", "conv3d", "attention", "layernorm", "elementwise", "track"};

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

void prof_begin(int cls, hipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    Pair p{get_event(), get_event(), cls};
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
}
