// Opt-in kernel timing: HIP events recorded on the launch stream around each main kernel, so
// that a caller (bench.py) can measure per-kernel durations live inside its timed region.
// Off by default; when off, TFC_PROF_* cost one relaxed load.  The reference has nothing
// comparable (its only timer is the epoch wall clock, eigenn/model/utils.py:4-35).
#include "tfc_common.h"

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

namespace {

struct Span {
    int name;
    hipEvent_t start, stop;
};

std::atomic<int> g_enabled{0};
std::mutex g_mu;
std::vector<std::string> g_names;
std::vector<Span> g_spans;
std::vector<std::pair<hipEvent_t, hipEvent_t>> g_pool;
constexpr size_t MAX_SPANS = 1u << 18;

int name_id(const char* name) {
    for (size_t i = 0; i < g_names.size(); ++i)
        if (g_names[i] == name) return (int)i;
    g_names.emplace_back(name);
    return (int)g_names.size() - 1;
}

}  // namespace

// TFC_TRACE_LAUNCHES=1 (debugging aid): every wrapped launch is named on stderr before it is issued and the stream
// is synchronised behind it (outside graph captures), so that the last name printed before a GPU fault is the
// kernel that faulted.
static bool trace_on() {
    static const bool on = getenv("TFC_TRACE_LAUNCHES") && atoi(getenv("TFC_TRACE_LAUNCHES")) != 0;
    return on;
}

int tfc_prof_begin(const char* name, hipStream_t stream) {
    if (trace_on()) {
        fprintf(stderr, "[tfc] %s\n", name);
        fflush(stderr);
    }
    const int mode = g_enabled.load(std::memory_order_relaxed);
    if (!mode) return -1;
    if (mode == 2 && strncmp(name, "tfc_conv_", 9) != 0) return -1;  // convolution kernels only
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_spans.size() >= MAX_SPANS) return -1;
    Span s;
    s.name = name_id(name);
    if (!g_pool.empty()) {
        s.start = g_pool.back().first;
        s.stop = g_pool.back().second;
        g_pool.pop_back();
    } else {
        if (hipEventCreate(&s.start) != hipSuccess) return -1;
        if (hipEventCreate(&s.stop) != hipSuccess) {
            (void)hipEventDestroy(s.start);
            return -1;
        }
    }
    (void)hipEventRecord(s.start, stream);
    g_spans.push_back(s);
    return (int)g_spans.size() - 1;
}

void tfc_prof_end(int idx, hipStream_t stream) {
    if (trace_on()) {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &st) == hipSuccess && st == hipStreamCaptureStatusNone)
            (void)hipStreamSynchronize(stream);
    }
    if (idx < 0) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if ((size_t)idx < g_spans.size()) (void)hipEventRecord(g_spans[idx].stop, stream);
}

extern "C" int tfc_profile_enable(int on) {
    g_enabled.store(on == 2 ? 2 : (on ? 1 : 0));
    return TFC_OK;
}

// Drops all recorded spans (their events return to the pool).
extern "C" int tfc_profile_reset(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& s : g_spans) g_pool.emplace_back(s.start, s.stop);
    g_spans.clear();
    return TFC_OK;
}

// Synchronises on the recorded events.  Writes one line per kernel name,
// "name count total_ms\n", into buf (NUL terminated); returns the number of names.
extern "C" int tfc_profile_report(char* buf, int64_t buf_len) {
    if (!buf || buf_len <= 0) return tfc_set_error(TFC_EINVAL, "tfc_profile_report: null buffer");
    std::lock_guard<std::mutex> lk(g_mu);
    std::vector<double> tot(g_names.size(), 0.0);
    std::vector<long> cnt(g_names.size(), 0);
    for (auto& s : g_spans) {
        if (hipEventSynchronize(s.stop) != hipSuccess) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s.start, s.stop) != hipSuccess) continue;
        tot[s.name] += ms;
        cnt[s.name] += 1;
    }
    std::string out;
    int n = 0;
    for (size_t i = 0; i < g_names.size(); ++i) {
        if (!cnt[i]) continue;
        char line[256];
        snprintf(line, sizeof(line), "%s %ld %.6f\n", g_names[i].c_str(), cnt[i], tot[i]);
        out += line;
        ++n;
    }
    if ((int64_t)out.size() + 1 > buf_len) return tfc_set_error(TFC_EINVAL, "tfc_profile_report: buffer too small");
    std::memcpy(buf, out.c_str(), out.size() + 1);
    return n;
}
