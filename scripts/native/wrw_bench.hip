// Where does ee_wrw.hip's product kernel spend a chunk?  Builds the product's source file as is (or with -DEE_WRW_SKIP=mask: phase skipping,
// results then wrong by construction) and times ee_wrw3x3_f32 on the four layer shapes of ResNet-18 at 64x64 inputs, batch 100 and 200.
//   for m in 0 1 2 4 8 16 31; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Iedge-enhancement_amd/csrc -DEE_WRW_SKIP=$m \
//       scripts/native/wrw_bench.hip -o scripts/native/wrw_bench_$m.bin; done
#include "../../edge-enhancement_amd/csrc/ee_wrw.hip"
#include <stdio.h>
#include <stdlib.h>
#include <vector>

namespace ee {  // ee_prof.hip's hooks, not linked here
ProfScope::ProfScope(int, hipStream_t, double) : id(0), stream(nullptr), slot(nullptr) {}
ProfScope::~ProfScope() {}
}  // namespace ee

#define CHECK(x)                                                \
    do {                                                        \
        hipError_t e_ = (x);                                    \
        if (e_ != hipSuccess) {                                 \
            printf("%s -> %s\n", #x, hipGetErrorString(e_));    \
            exit(1);                                            \
        }                                                       \
    } while (0)

int main() {
    const int shapes[4][2] = {{64, 16}, {128, 8}, {256, 4}, {512, 2}};
    printf("EE_WRW_SKIP = %d; us per ee_wrw3x3_f32 (product kernel + sum kernel), 20 back-to-back calls\n", EE_WRW_SKIP);
    for (int B : {100, 200}) {
        for (auto &sh : shapes) {
            const int C = sh[0], H = sh[1];
            const size_t n = static_cast<size_t>(B) * C * H * H;
            float *x, *dy, *dw, *ws;
            CHECK(hipMalloc(&x, n * 4));
            CHECK(hipMalloc(&dy, n * 4));
            CHECK(hipMalloc(&dw, static_cast<size_t>(9) * C * C * 4));
            const long long wsn = ee_wrw3x3_workspace_floats(B, C, C, H);
            CHECK(hipMalloc(&ws, (wsn > 0 ? wsn : 4) * 4));
            std::vector<float> h(n);
            for (size_t i = 0; i < n; ++i) h[i] = static_cast<float>((i * 2654435761u >> 8) & 1023) / 512.0f - 1.0f;
            CHECK(hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice));
            CHECK(hipMemcpy(dy, h.data(), n * 4, hipMemcpyHostToDevice));
            hipEvent_t a, b;
            CHECK(hipEventCreate(&a));
            CHECK(hipEventCreate(&b));
            for (int i = 0; i < 5; ++i)
                if (ee_wrw3x3_f32(x, dy, dw, ws, B, C, C, H, nullptr) != 0) { printf("launch failed\n"); return 1; }
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(a, nullptr));
            for (int i = 0; i < 20; ++i) ee_wrw3x3_f32(x, dy, dw, ws, B, C, C, H, nullptr);
            CHECK(hipEventRecord(b, nullptr));
            CHECK(hipEventSynchronize(b));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, a, b));
            printf("  B %3d  %3d ch %2dx%-2d  %7.2f\n", B, C, H, H, ms * 1000 / 20);
            CHECK(hipFree(x)); CHECK(hipFree(dy)); CHECK(hipFree(dw)); CHECK(hipFree(ws));
        }
    }
    return 0;
}
