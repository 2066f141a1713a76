// What does a "last arriver finalizes, the others wait on a flag" barrier cost inside a kernel of the convolution kernels' shape
// (200 workgroups x 512 lanes, one per CU)?  The question behind fusing BatchNorm's batch statistics into a convolution's epilogue:
// the separate BatchNorm launch it would replace costs 5 - 8 us.
//   hipcc --offload-arch=gfx950 -O3 scripts/native/group_barrier.hip -o scripts/native/group_barrier.bin && scripts/native/group_barrier.bin
// MODE 0: every workgroup writes its 32-channel x 256-pixel tile (the epilogue of today)
// MODE 1: + per-channel partial moments, ticket per channel group (P workgroups), the last arriver merges the P partials of its 32
//           channels and raises the group's flag; the others spin on the flag (bounded), then everybody applies a * y + b and writes
// MODE 2: the same with ONE group spanning the whole grid (a full grid barrier)
// The counters reset themselves (last departer), so launches can follow each other as in a replayed graph.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>

#define CHECK(x)                                                                    \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            printf("%s -> %s\n", #x, hipGetErrorString(e_));                        \
            exit(1);                                                                \
        }                                                                           \
    } while (0)

constexpr int NT = 512, CO = 32, PX = 256;
constexpr int SPIN_LIMIT = 1 << 20;

struct Sync {
    unsigned arrive, depart, flag, error;
};

template <int MODE, bool FENCE>
__global__ __launch_bounds__(NT) void epilogue(const float *__restrict__ in, float *__restrict__ out, float *__restrict__ partial,
                                               float *__restrict__ ab, Sync *__restrict__ sync, int ngroups, int P, int spin_work) {
    __shared__ float red[CO][17];
    __shared__ unsigned last_s;
    const int group = MODE == 2 ? 0 : blockIdx.x % ngroups, member = MODE == 2 ? blockIdx.x : blockIdx.x / ngroups;
    const int members = MODE == 2 ? gridDim.x : P;
    const int co = threadIdx.x >> 4, tl = threadIdx.x & 15;  // (channel, 16-pixel tile column) as in the output transform
    // the workgroup's tile: 32 channels x 256 pixels, 16 floats per lane (kept in registers like the accumulators' transform output)
    float v[16];
    const float *src = in + (static_cast<size_t>(blockIdx.x) * CO + co) * PX + tl * 16;
#pragma unroll
    for (int i = 0; i < 16; i += 4) *reinterpret_cast<float4 *>(v + i) = *reinterpret_cast<const float4 *>(src + i);
    float fake = 0.0f;
    for (int i = 0; i < spin_work; ++i) fake = fake * 1.0001f + v[i & 15];  // stand-in for the kernel body in front of the epilogue
    float a = 1.0f, b = fake * 1e-30f;
    if (MODE != 0) {
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += v[i];
        for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 16);
        const float mean = s * (1.0f / PX);
        float m2 = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) m2 += (v[i] - mean) * (v[i] - mean);
        for (int o = 8; o > 0; o >>= 1) m2 += __shfl_xor(m2, o, 16);
        const int chan = (MODE == 2 ? blockIdx.x % ngroups : group) * CO + co;
        const int slot = MODE == 2 ? blockIdx.x / ngroups : member;
        if (tl == 0) {
            if (FENCE) {
                partial[(static_cast<size_t>(chan) * P + slot) * 2] = mean;
                partial[(static_cast<size_t>(chan) * P + slot) * 2 + 1] = m2;
            } else {  // write-through stores at agent scope: visible to the other XCDs without flushing this XCD's L2
                __hip_atomic_store(&partial[(static_cast<size_t>(chan) * P + slot) * 2], mean, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&partial[(static_cast<size_t>(chan) * P + slot) * 2 + 1], m2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (FENCE) __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) last_s = atomicAdd(&sync[group].arrive, 1u) == static_cast<unsigned>(members - 1);
        __syncthreads();
        if (last_s) {
            if (FENCE) __threadfence();
            // merge the P partials of each channel in slot order (Chan): 16 lanes per channel, then a fixed-order combine
            const int nch = MODE == 2 ? ngroups * CO : CO;
            for (int c0 = 0; c0 < nch; c0 += CO) {
                const int ch = (MODE == 2 ? 0 : group * CO) + c0 + co;
                float n = 0.0f, mu = 0.0f, M2 = 0.0f;
                for (int p = tl; p < P; p += 16) {
                    const float pm = __hip_atomic_load(&partial[(static_cast<size_t>(ch) * P + p) * 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const float p2 = __hip_atomic_load(&partial[(static_cast<size_t>(ch) * P + p) * 2 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const float nn = n + PX, dl = pm - mu;
                    mu += dl * (PX / nn);
                    M2 += p2 + dl * dl * (n * PX / nn);
                    n = nn;
                }
                red[co][tl] = mu;
                __syncthreads();
                // lane 0 of the channel combines the 16 lane results in order (counts differ by at most PX; fine for a timing stand-in)
                if (tl == 0) {
                    float m = 0.0f;
                    for (int k = 0; k < 16; ++k) m += red[co][k];
                    __hip_atomic_store(&ab[ch * 2], 1.0f / sqrtf(M2 / n + 1e-5f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&ab[ch * 2 + 1], -m * (1.0f / 16.0f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __syncthreads();
            }
            if (FENCE) __threadfence();
            if (threadIdx.x == 0) __hip_atomic_store(&sync[group].flag, 1u, FENCE ? __ATOMIC_RELEASE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (threadIdx.x == 0) {
                int it = 0;
                while (__hip_atomic_load(&sync[group].flag, FENCE ? __ATOMIC_ACQUIRE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && ++it < SPIN_LIMIT) __builtin_amdgcn_s_sleep(1);
                if (it >= SPIN_LIMIT) atomicAdd(&sync[group].error, 1u);
            }
            __syncthreads();
        }
        if (FENCE) __threadfence();
        a = __hip_atomic_load(&ab[chan * 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        b += __hip_atomic_load(&ab[chan * 2 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // self-reset: the last workgroup to leave clears the group's words for the next launch
        __syncthreads();
        if (threadIdx.x == 0) {
            if (atomicAdd(&sync[group].depart, 1u) == static_cast<unsigned>(members - 1)) {
                __hip_atomic_store(&sync[group].arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&sync[group].depart, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&sync[group].flag, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    float *dst = out + (static_cast<size_t>(blockIdx.x) * CO + co) * PX + tl * 16;
#pragma unroll
    for (int i = 0; i < 16; i += 4) {
        float4 o = make_float4(fmaxf(a * v[i] + b, 0.0f), fmaxf(a * v[i + 1] + b, 0.0f), fmaxf(a * v[i + 2] + b, 0.0f), fmaxf(a * v[i + 3] + b, 0.0f));
        *reinterpret_cast<float4 *>(dst + i) = o;
    }
}

template <int MODE, bool FENCE>
float run(const float *in, float *out, float *partial, float *ab, Sync *sync, int grid, int ngroups, int P, int work, int reps) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((epilogue<MODE, FENCE>), dim3(grid), dim3(NT), 0, 0, in, out, partial, ab, sync, ngroups, P, work);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((epilogue<MODE, FENCE>), dim3(grid), dim3(NT), 0, 0, in, out, partial, ab, sync, ngroups, P, work);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return 1e3f * ms / reps;
}

int main() {
    const int reps = 500;
    for (int cfg = 0; cfg < 3; ++cfg) {
        // (grid, channel groups, workgroups per group): 16x16 layer1 (100 images x 2 blocks), 8x8 layer2 (50 x 4), 4x4 layer3 (25 x 8)
        const int ngroups = cfg == 0 ? 2 : cfg == 1 ? 4 : 8, P = cfg == 0 ? 100 : cfg == 1 ? 50 : 25, grid = ngroups * P;
        float *in, *out, *partial, *ab;
        Sync *sync;
        CHECK(hipMalloc(&in, sizeof(float) * grid * CO * PX));
        CHECK(hipMalloc(&out, sizeof(float) * grid * CO * PX));
        CHECK(hipMalloc(&partial, sizeof(float) * 2 * ngroups * CO * grid));
        CHECK(hipMalloc(&ab, sizeof(float) * 2 * ngroups * CO));
        CHECK(hipMalloc(&sync, sizeof(Sync) * 16));
        CHECK(hipMemset(sync, 0, sizeof(Sync) * 16));
        {
            float *h = (float *)malloc(sizeof(float) * grid * CO * PX);
            for (int w = 0; w < grid; ++w)
                for (int c = 0; c < CO; ++c)
                    for (int i = 0; i < PX; ++i) h[(static_cast<size_t>(w) * CO + c) * PX + i] = 1.0f + (w % ngroups) * CO + c + 0.01f * (w / ngroups);
            CHECK(hipMemcpy(in, h, sizeof(float) * grid * CO * PX, hipMemcpyHostToDevice));
            free(h);
        }
        for (int work : {0, 4000}) {
            const float t0 = run<0, false>(in, out, partial, ab, sync, grid, ngroups, P, work, reps);
            const float t1f = run<1, true>(in, out, partial, ab, sync, grid, ngroups, P, work, reps);
            const float t1 = run<1, false>(in, out, partial, ab, sync, grid, ngroups, P, work, reps);
            float hab[2 * 8 * CO];
            CHECK(hipMemcpy(hab, ab, sizeof(float) * 2 * ngroups * CO, hipMemcpyDeviceToHost));
            int bad = 0;  // channel ch of workgroup w holds the constant 1 + ch + 0.01 w ... see the fill below: mean over slots known
            for (int ch = 0; ch < ngroups * CO; ++ch) {
                double want = 0.0;
                for (int p = 0; p < P; ++p) want += 1.0 + ch + 0.01 * p;
                want /= P;
                // the stand-in combine sums the 16 lane means / 16: exact only if every lane saw the same number of slots; compare loosely
                if (!(fabs(-hab[2 * ch + 1] - want) < 0.05 * want + 0.1)) ++bad;
            }
            const float t2 = run<2, false>(in, out, partial, ab, sync, grid, ngroups, grid / ngroups, work, reps);
            printf("   [with __threadfence: %6.2f us]  merged means wrong in %d of %d channels\n", t1f, bad, ngroups * CO);
            Sync h[16];
            CHECK(hipMemcpy(h, sync, sizeof(h), hipMemcpyDeviceToHost));
            unsigned err = 0;
            for (int g = 0; g < 16; ++g) err += h[g].error + h[g].arrive + h[g].depart + h[g].flag;
            printf("grid %3d = %d groups x %3d, body %4d iterations: plain epilogue %6.2f us | group barrier %6.2f us (+%5.2f) | grid barrier %6.2f us (+%5.2f) | leftover/err %u\n",
                   grid, ngroups, P, work, t0, t1, t1 - t0, t2, t2 - t0, err);
        }
        CHECK(hipFree(in));
        CHECK(hipFree(out));
        CHECK(hipFree(partial));
        CHECK(hipFree(ab));
        CHECK(hipFree(sync));
    }
    return 0;
}
