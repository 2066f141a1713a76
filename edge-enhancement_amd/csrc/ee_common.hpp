// ee_common.hpp - shared device/host helpers for libeeadv (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "eeadv.h"

#define EE_API extern "C" __attribute__((visibility("default")))

namespace ee {

constexpr int kWave = 64;         // CDNA wavefront
constexpr int kBlock = 256;       // 4 waves per workgroup
constexpr int kMaxGrid = 256 * 8; // 256 CUs x 8 workgroups: cap, then grid-stride

static inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// ---- XCD-aware workgroup numbering for kernels whose workgroups are (image group, result-channel block) pairs ------------------------
// The 8 XCDs of a gfx950 take the workgroups of a 1-D grid round-robin, and each has an L2 of its own: what two workgroups share is fetched
// once only if they sit on the same XCD.  weights_local: the channel block runs fastest (an XCD sees the filter slabs of the blocks
// congruent to it, every image's input goes to min(ncb, 8) XCDs); otherwise an image group's ncb workgroups are numbered 8 apart (one XCD
// fetches that input once, every XCD fetches all filters).  The host picks the cheaper one from the two footprints (xcd_weights_local) and
// sizes the grid with xcd_grid; a workgroup whose decode returns false (the padding of the second scheme) exits at once.
static inline bool xcd_weights_local(double in_bytes, double w_bytes, int ncb) {
    const int dup = ncb < 8 ? ncb : 8;
    return in_bytes * dup + w_bytes * (8 / dup) < in_bytes + 8.0 * w_bytes;
}
static inline unsigned xcd_grid(int nimg, int ncb, bool weights_local) {
    return static_cast<unsigned>(weights_local ? nimg : (nimg + 7) / 8 * 8) * static_cast<unsigned>(ncb);
}
__device__ __forceinline__ bool xcd_decode(int id, int ncb, int nimg, bool weights_local, int &bx, int &by) {
    if (weights_local) {
        bx = id / ncb, by = id - bx * ncb;
        return true;
    }
    const int g = id / (8 * ncb), r = id - g * (8 * ncb);
    by = r >> 3, bx = g * 8 + (r & 7);
    return bx < nimg;
}

// compute units of the current device (persistent kernels size their grids by it); 256 on MI355X, which is also the answer when the query fails
static inline int device_cus() {
    static int cus[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
    if (cus[dev] == 0) {
        int n = 0;
        cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return cus[dev];
}
static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline bool aligned4(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 3u) == 0; }

// ---- timing hooks (ee_prof.hip) ---------------------------------------------------------------
struct ProfScope {
    int id;
    hipStream_t stream;
    void *slot;
    ProfScope(int kernel_id, hipStream_t s, double work = 0.0);
    ~ProfScope();
};

// hipGetLastError() is thread-sticky across ALL runtime calls (torch's included): clear it before a launch so
// that launch_status() reports this launch and not somebody else's stale error.
#define EE_LAUNCH(...)                    \
    do {                                  \
        (void)hipGetLastError();          \
        hipLaunchKernelGGL(__VA_ARGS__);  \
    } while (0)

static inline int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? EE_OK : static_cast<int>(e);
}

// ---- device arithmetic with the reference's (torch) semantics ---------------------------------
// torch.sign: sign(+-0) = 0, sign(NaN) = 0
__device__ __forceinline__ float sgn(float g) { return static_cast<float>((g > 0.0f) - (g < 0.0f)); }
// torch.max / torch.min (binary): NaN in either operand propagates
__device__ __forceinline__ float tmax(float a, float b) { return (a != a || b != b) ? (a + b) : (a > b ? a : b); }
__device__ __forceinline__ float tmin(float a, float b) { return (a != a || b != b) ? (a + b) : (a < b ? a : b); }
// torch.clamp(v, lo, hi); NaN propagates
__device__ __forceinline__ float tclamp(float v, float lo, float hi) {
    float r = v < lo ? lo : v;
    r = r > hi ? hi : r;
    return (v != v) ? v : r;
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ---- Philox4x32-10 ----------------------------------------------------------------------------
struct Philox {
    uint32_t k0, k1;
    __device__ Philox(uint64_t seed) : k0(static_cast<uint32_t>(seed)), k1(static_cast<uint32_t>(seed >> 32)) {}
    __device__ uint4 operator()(uint64_t ctr, uint32_t stream_id = 0) const {
        uint32_t c0 = static_cast<uint32_t>(ctr), c1 = static_cast<uint32_t>(ctr >> 32), c2 = stream_id, c3 = 0;
        uint32_t a = k0, b = k1;
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
            const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
            const uint32_t n0 = hi1 ^ c1 ^ a, n1 = lo1, n2 = hi0 ^ c3 ^ b, n3 = lo0;
            c0 = n0; c1 = n1; c2 = n2; c3 = n3;
            a += 0x9E3779B9u;
            b += 0xBB67AE85u;
        }
        return make_uint4(c0, c1, c2, c3);
    }
};
// 24 random bits -> [0,1), as torch's uniform_ does for float
__device__ __forceinline__ float u01(uint32_t r) { return static_cast<float>(r >> 8) * (1.0f / 16777216.0f); }

}  // namespace ee
