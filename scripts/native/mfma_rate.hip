// How long does a chain of dependent v_mfma_f32_32x32x2_f32 take on this chip, per SIMD, at the occupancies of the convolution
// kernels?  hipcc --offload-arch=gfx950 -O3 scripts/native/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CHAINS>
__global__ void chain32(float *out, int n, float a, float b) {
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = f32x16{0};
    float x = a + threadIdx.x, y = b;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[c], 0, 0, 0);
    }
    float s = 0;
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CHAINS>
__global__ void chain16(float *out, int n, float a, float b) {
    f32x4 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = f32x4{0};
    float x = a + threadIdx.x, y = b;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[c], 0, 0, 0);
    }
    float s = 0;
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 4; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the convolution's inner loop in stages: MODE 1 operands from LDS (ring, lookahead 3), 2 + one LDS write per multiply,
// 3 + one workgroup barrier per 72 multiplies, 4 + 9 float4 global loads per 72 multiplies
template <int MODE>
__global__ void chain_lds(float *out, int n, float a, float b, const float *g) {
    __shared__ float buf[2][6400];
    for (int i = threadIdx.x; i < 6400; i += blockDim.x) { buf[0][i] = a + i; buf[1][i] = b - i; }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x16 acc = f32x16{0};
    float4 gv[9], gw[9];
    for (int j = 0; j < 9; ++j) gv[j] = gw[j] = make_float4(0, 0, 0, 0);
    for (int it = 0; it < n / 72; ++it) {
        const float *cur = buf[it & 1];
        float *nxt = buf[(it + 1) & 1];
        float ra[4], rb[4];
#pragma unroll
        for (int q = 0; q < 3; ++q) { ra[q] = cur[lane + q * 66]; rb[q] = cur[3000 + lane + q * 10]; }
#pragma unroll
        for (int q = 0; q < 72; ++q) {
            if (q + 3 < 72) { ra[(q + 3) & 3] = cur[lane + ((q + 3) % 40) * 66]; rb[(q + 3) & 3] = cur[3000 + lane + ((q + 3) % 40) * 10]; }
            if (MODE >= 2 && q < 43) nxt[threadIdx.x + q * 64] = (MODE >= 4) ? ((float *)gv)[q % 36] : a;
            if (MODE == 5 && q == 0) {
#pragma unroll
                for (int j = 0; j < 9; ++j) gw[j] = *reinterpret_cast<const float4 *>(g + (size_t)(blockIdx.x * 256 + threadIdx.x) * 36 + it * 9216 % 65536 + 4 * j);
            }
            if (MODE == 5 && q == 71) {
#pragma unroll
                for (int j = 0; j < 9; ++j) gv[j] = gw[j];
            }
            if (MODE == 6 && q == 0) {  // coalesced: consecutive lanes read consecutive 16-B pieces
#pragma unroll
                for (int j = 0; j < 9; ++j) gw[j] = *reinterpret_cast<const float4 *>(g + (size_t)blockIdx.x * 9216 + it * 9216 % 65536 + 4 * (threadIdx.x + 256 * j));
            }
            if (MODE == 6 && q == 71) {
#pragma unroll
                for (int j = 0; j < 9; ++j) gv[j] = gw[j];
            }
            if (MODE == 4 && q == 44) {
#pragma unroll
                for (int j = 0; j < 9; ++j) gv[j] = *reinterpret_cast<const float4 *>(g + (size_t)(blockIdx.x * 256 + threadIdx.x) * 36 + it * 9216 % 65536 + 4 * j);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[q & 3], rb[q & 3], acc, 0, 0, 0);
        }
        if (MODE >= 3) __syncthreads();
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class K>
float run(K k, int grid, int block, int n, float *out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, 0, out, n, 1.0f, 2.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, 0, out, n, 1.0f, 2.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 100.0f;  // us per launch
}

int main() {
    float *out; hipMalloc(&out, 4 << 20);
    const int n = 576;
    printf("%-44s %8s %10s %10s\n", "config (576 x 32x32x2 per wave chain)", "us", "clk/MFMA*", "TFLOP/s");
    struct { const char *name; int grid, block; } cfg[] = {{"200 WG x 4 waves (1 wave/SIMD, 200 CUs)", 200, 256}, {"256 WG x 4 waves", 256, 256},
        {"200 WG x 8 waves (2 waves/SIMD)", 200, 512}, {"512 WG x 4 waves", 512, 256}, {"1024 WG x 4 waves", 1024, 256}};
    for (auto &c : cfg) {
        float us = run(chain32<1>, c.grid, c.block, n, out);
        double mf = (double)c.grid * (c.block / 64) * n;
        printf("%-44s %8.1f %10.1f %10.1f\n", c.name, us, us * 2400.0 / n, mf * 4096 / us / 1e6);
    }
    float us = run(chain32<2>, 256, 256, n / 2, out);
    printf("%-44s %8.1f  (two independent chains per wave, same MFMA count)\n", "256 WG x 4 waves, 2 chains", us);
    us = run(chain16<4>, 256, 256, n / 2, out);   // 16x16x4: 1024 FMA each: n/2 * 4 chains = 2 n instr = same FMAs as n 32x32x2
    printf("%-44s %8.1f  (16x16x4, four chains, same FMA count)\n", "256 WG x 4 waves, 16x16x4", us);
    printf("* clocks per MFMA assuming 2.4 GHz and one wave per SIMD\n");
    float *g; hipMalloc(&g, 64 << 20); hipMemset(g, 0, 64 << 20);
    auto run2 = [&](auto k, const char *name) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(200), dim3(256), 0, 0, out, n, 1.0f, 2.0f, g); hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k, dim3(200), dim3(256), 0, 0, out, n, 1.0f, 2.0f, g);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-60s %8.1f us\n", name, ms * 100.0f);
    };
    run2(chain_lds<1>, "200 WG x 4 waves: operands from LDS (lookahead 3)");
    run2(chain_lds<2>, "  + one LDS write per multiply (43 of 72)");
    run2(chain_lds<3>, "  + one barrier per 72 multiplies");
    run2(chain_lds<4>, "  + 9 float4 global loads per 72 multiplies (28 multiplies ahead)");
    run2(chain_lds<5>, "    same loads issued a whole round ahead (2nd register set)");
    run2(chain_lds<6>, "    ... and coalesced (lane-contiguous 16-B pieces)");
    return 0;
}
