// Micro-benchmark of the pointwise (1x1) filter gradient of the wide DS-CNN layers through the library's own launcher
// (tcr::launch_conv_wgrad -> pw_wgrad_lds_kernel): dW[276][276] = sum over 4096 x 65 positions, x raw + BN affine + ReLU on the fly.
// Prints the time per launch, the f32-MFMA rate and the core clock the kernel ran at (s_memtime cycles per constant-rate tick), so that
// a launch-geometry or kernel change can be judged in seconds without the whole training step.
// Build (after tc-resnet_amd/build.py):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I tc-resnet_amd/csrc scripts/micro/pw_wgrad_bench.cpp -o scripts/micro/pw_wgrad_bench \
//         -L tc-resnet_amd/lib -ltcresnet_hip -Wl,-rpath,'$ORIGIN/../../tc-resnet_amd/lib'
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "kernels.h"

__global__ void clock_probe(long long* out, int spin) {
    const long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    long long c = c0;
    while (c - c0 < spin) c = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[0] = c - c0; out[1] = wall_clock64() - w0; }
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 4096, C = argc > 2 ? atoi(argv[2]) : 276, P = 65, PP = P + 2 * tcr::kHalo;
    if (argc > 3) tcr_tune(TCR_TUNE_PW_WGRAD, atoi(argv[3]));      // 1: the DMA-staged kernel
    const size_t n = (size_t)B * C * PP;
    std::vector<float> hx(n), hd(n), hs(C), hb(C);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (size_t i = 0; i < n; ++i) { hx[i] = rnd(); hd[i] = ((i % PP) >= (size_t)tcr::kHalo && (i % PP) < (size_t)(tcr::kHalo + P)) ? rnd() : 0.f; }
    for (int c = 0; c < C; ++c) { hs[c] = 1.0f + 0.01f * (c % 7); hb[c] = 0.05f * ((c % 5) - 2); }
    float *x, *d, *dw, *scr, *sc, *sh;
    long long* clk;
    const size_t scratch = (size_t)256 * 288 * 288 * 4;
    hipMalloc(&x, n * 4); hipMalloc(&d, n * 4); hipMalloc(&dw, (size_t)C * C * 4); hipMalloc(&scr, scratch); hipMalloc(&sc, C * 4); hipMalloc(&sh, C * 4);
    hipMalloc(&clk, 16);
    hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(d, hd.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(sc, hs.data(), C * 4, hipMemcpyHostToDevice); hipMemcpy(sh, hb.data(), C * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&]() { return tcr::launch_conv_wgrad(1, 1, 0, x, d, dw, scr, B, C, C, PP, P, PP, nullptr, sc, sh, false); };
    for (int i = 0; i < 20; ++i) if (run() != 0) { printf("launch failed: %s\n", tcr_last_error()); return 1; }
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) run();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / 20, tf = 2.0 * C * C * (double)B * P / (us * 1e-6) * 1e-12;
        printf("pointwise filter gradient %d x %d, batch %d: %.1f us per launch (kernel + slab reduction), %.1f TFLOP/s = %.1f %% of 157.3\n", C, C, B, us, tf, tf / 1.573);
    }
    // clock while the kernel is running on the other CUs' neighbours: a one-wave probe launched right behind it on another stream
    hipStream_t s2;
    hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    for (int i = 0; i < 10; ++i) run();
    hipLaunchKernelGGL(clock_probe, dim3(1), dim3(64), 0, s2, clk, 400000);
    for (int i = 0; i < 10; ++i) run();
    hipDeviceSynchronize();
    long long hc[2];
    hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
    printf("core clock under this kernel: %.0f MHz (%lld cycles in %lld ticks of 100 MHz)\n", (double)hc[0] / ((double)hc[1] / 100.0), hc[0], hc[1]);
    hipLaunchKernelGGL(clock_probe, dim3(1), dim3(64), 0, s2, clk, 400000);
    hipDeviceSynchronize();
    hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
    printf("core clock, probe alone:      %.0f MHz\n", (double)hc[0] / ((double)hc[1] / 100.0));
    std::vector<float> hw((size_t)C * C);
    hipMemcpy(hw.data(), dw, hw.size() * 4, hipMemcpyDeviceToHost);
    double cs = 0;
    for (float v : hw) cs += (double)v;
    printf("checksum %.6f  dw[0] %.6f dw[last] %.6f\n", cs, hw[0], hw.back());
    return 0;
}
