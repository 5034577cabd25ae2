// Micro-benchmark: sustained rate of v_mfma_f32_16x16x4_f32 on this device (no memory traffic), by waves per CU.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_peak.hip -o scripts/micro/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) out[0] = s;
}

int main() {
    float* out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    int clk = 0;
    hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("CUs %d, clock attr %d kHz\n", cus, clk);
    const int iters = 4000;
    for (int wg_per_cu = 1; wg_per_cu <= 3; ++wg_per_cu) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((mfma_loop<24>), dim3(cus * wg_per_cu), dim3(256), 0, 0, out, iters, 1.0f, 2.0f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double flops = (double)cus * wg_per_cu * 4 * iters * 24 * 2048.0;
            if (rep == 2) printf("waves/SIMD %d: %.3f ms  %.1f TFLOP/s\n", wg_per_cu, ms, flops / ms * 1e-9);
        }
    }
    // long run: sustained clocks
    hipEventRecord(e0);
    for (int k = 0; k < 20; ++k) hipLaunchKernelGGL((mfma_loop<24>), dim3(cus * 2), dim3(256), 0, 0, out, iters * 4, 1.0f, 2.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("sustained (20 launches, 2 waves/SIMD): %.1f ms  %.1f TFLOP/s\n", ms, (double)cus * 2 * 4 * iters * 4 * 24 * 2048.0 * 20 / ms * 1e-9);
    return 0;
}
