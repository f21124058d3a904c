// HIP-event timing of one kernel class (see bmp_kernels.h).  Diagnostic only.
#include <vector>
#include "bmp_kernels.h"

namespace {
struct Rec { hipEvent_t a, b; double flops, bytes; int key; };
int g_cls = 0;
std::vector<Rec> g_recs;
}  // namespace

BmpProfScope::BmpProfScope(int kclass, double flops, double bytes, hipStream_t s, int kid) : slot(-1), st(s) {
    if (g_cls == 0 || (kclass != g_cls && g_cls != BMP_KCLS_ALL)) return;
    Rec r;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    r.flops = flops; r.bytes = bytes; r.key = kclass * 16 + kid;
    (void)hipEventRecord(r.a, st);
    g_recs.push_back(r);
    slot = (int)g_recs.size() - 1;
}

BmpProfScope::~BmpProfScope() {
    if (slot >= 0) (void)hipEventRecord(g_recs[slot].b, st);
}

extern "C" int bmp_prof_start(int kclass) {
    for (auto& r : g_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    g_recs.clear();
    g_cls = kclass;
    return 0;
}

// Waits for the recorded launches, returns their count; sums go to out[0] = total ms,
// out[1] = total flops, out[2] = total bytes (as given by the launchers: executed, all rows).
extern "C" int bmp_prof_stop(double* out) {
    double ms = 0, fl = 0, by = 0;
    int n = 0;
    for (auto& r : g_recs) {
        float t = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) {
            ms += t; fl += r.flops; by += r.bytes; ++n;
        }
        (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
    }
    g_recs.clear();
    g_cls = 0;
    out[0] = ms; out[1] = fl; out[2] = by;
    return n;
}

// As bmp_prof_stop, per kernel: waits for the recorded launches and aggregates them by key = class * 16 + kernel id
// (bmp_kernels.h) into key / count / ms / flops / bytes [cap]; returns the number of distinct keys (<= cap).
extern "C" int bmp_prof_collect(int* key, int* count, double* ms, double* flops, double* bytes, int cap) {
    int n = 0;
    for (auto& r : g_recs) {
        float t = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) {
            int k = 0;
            while (k < n && key[k] != r.key) ++k;
            if (k == n && n < cap) { key[n] = r.key; count[n] = 0; ms[n] = flops[n] = bytes[n] = 0.0; ++n; }
            if (k < n) { ++count[k]; ms[k] += t; flops[k] += r.flops; bytes[k] += r.bytes; }
        }
        (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
    }
    g_recs.clear();
    g_cls = 0;
    return n;
}

// A stream of the lowest priority the device offers, for launches that run BESIDE a dependent chain (the weight-gradient
// GEMMs of the backward: bmp/functional.py _on_side): their workgroups take the CUs the chain's tile kernels leave idle in
// their last round and give way to the chain otherwise.  The caller destroys it with bmp_stream_destroy.
extern "C" int bmp_stream_create_low(hipStream_t* out) {
    if (!out) return (int)hipErrorInvalidValue;
    int least = 0, greatest = 0;
    hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (e != hipSuccess) return (int)e;
    return (int)hipStreamCreateWithPriority(out, hipStreamNonBlocking, least);
}
extern "C" int bmp_stream_destroy(hipStream_t st) { return st ? (int)hipStreamDestroy(st) : 0; }
