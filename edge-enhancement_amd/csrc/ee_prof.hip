// ee_prof.hip - library bookkeeping: version / error strings and the optional HIP-event timing of
// kernel launches (bench.py reads these to compute roofline.achieved live, on the launch stream).
#include <mutex>
#include <vector>

#include "ee_common.hpp"

namespace {

struct Pair {
    hipEvent_t start, stop;
};
struct Family {
    std::vector<Pair> pending;
    std::vector<Pair> pool;
    double total_ms = 0.0, work = 0.0;
    int64_t launches = 0;
};
Family g_fam[EE_K_COUNT];
bool g_on = false;
std::mutex g_mu;
char g_devname[256];

bool capturing(hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return st != hipStreamCaptureStatusNone;
}

int drain(Family &f) {
    for (Pair &p : f.pending) {
        hipError_t e = hipEventSynchronize(p.stop);
        if (e != hipSuccess) return static_cast<int>(e);
        float ms = 0.0f;
        e = hipEventElapsedTime(&ms, p.start, p.stop);
        if (e != hipSuccess) return static_cast<int>(e);
        f.total_ms += ms;
        f.launches += 1;
        f.pool.push_back(p);
    }
    f.pending.clear();
    return EE_OK;
}

}  // namespace

namespace ee {

ProfScope::ProfScope(int kernel_id, hipStream_t s, double work) : id(kernel_id), stream(s), slot(nullptr) {
    if (!g_on || capturing(s)) return;
    std::lock_guard<std::mutex> lk(g_mu);
    Family &f = g_fam[id];
    Pair p;
    if (!f.pool.empty()) {
        p = f.pool.back();
        f.pool.pop_back();
    } else {
        if (hipEventCreate(&p.start) != hipSuccess || hipEventCreate(&p.stop) != hipSuccess) return;
    }
    (void)hipEventRecord(p.start, s);
    f.pending.push_back(p);
    f.work += work;
    slot = &f;
}

ProfScope::~ProfScope() {
    if (!slot) return;
    std::lock_guard<std::mutex> lk(g_mu);
    Family &f = *static_cast<Family *>(slot);
    (void)hipEventRecord(f.pending.back().stop, stream);
}

}  // namespace ee

EE_API int ee_abi_version(void) { return EEADV_ABI_VERSION; }

EE_API const char *ee_strerror(int code) {
    switch (code) {
        case EE_OK: return "ok";
        case EE_ERR_NULL: return "eeadv: required pointer is NULL";
        case EE_ERR_SHAPE: return "eeadv: size or shape argument out of range";
        case EE_ERR_UNSUPPORTED: return "eeadv: configuration not supported by this build";
        case EE_ERR_ALIGN: return "eeadv: pointer not aligned to its element size";
        default: return code > 0 ? hipGetErrorString(static_cast<hipError_t>(code)) : "eeadv: unknown error";
    }
}

EE_API const char *ee_device_name(void) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    snprintf(g_devname, sizeof(g_devname), "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return g_devname;
}

EE_API int ee_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_on = on != 0;
    return EE_OK;
}

EE_API int ee_prof_mark_empty(void *stream) {
    ee::ProfScope scope(EE_K_EMPTY, ee::as_stream(stream));
    return EE_OK;
}

EE_API int ee_prof_read(int kernel_id, double *total_ms, int64_t *launches) {
    if (kernel_id < 0 || kernel_id >= EE_K_COUNT) return EE_ERR_SHAPE;
    if (!total_ms || !launches) return EE_ERR_NULL;
    std::lock_guard<std::mutex> lk(g_mu);
    Family &f = g_fam[kernel_id];
    int rc = drain(f);
    *total_ms = f.total_ms;
    *launches = f.launches;
    return rc;
}

EE_API int ee_prof_read_work(int kernel_id, double *work) {
    if (kernel_id < 0 || kernel_id >= EE_K_COUNT) return EE_ERR_SHAPE;
    if (!work) return EE_ERR_NULL;
    std::lock_guard<std::mutex> lk(g_mu);
    *work = g_fam[kernel_id].work;
    return EE_OK;
}

EE_API int ee_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (Family &f : g_fam) {
        int rc = drain(f);
        if (rc != EE_OK) return rc;
        f.total_ms = f.work = 0.0;
        f.launches = 0;
    }
    return EE_OK;
}
