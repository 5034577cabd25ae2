// What a fork costs on the forking stream: a chain of short kernels on stream A, after each of them a kernel on stream B that
// depends on it -- (0) no fork at all, (1) hipEventRecord(A) + hipStreamWaitEvent(B), (2) the kernel itself bumps a counter in
// signal memory and B waits for it with hipStreamWaitValue32 (no packet on A), (3) hipStreamWriteValue32 on A instead of the event.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/fork_gap.hip -o scripts/micro/fork_gap && scripts/micro/fork_gap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void work(float* p, int iters, unsigned* sig) {
    float v = p[threadIdx.x];
    for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
    p[threadIdx.x] = v;
    if (sig && threadIdx.x == 0 && blockIdx.x == 0) {
        __threadfence_system();
        __hip_atomic_fetch_add(sig, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

int main(int argc, char** argv) {
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    hipStream_t A, B;
    CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    float *pa, *pb;
    CK(hipMalloc(&pa, 4096)); CK(hipMalloc(&pb, 4096));
    CK(hipMemset(pa, 0, 4096)); CK(hipMemset(pb, 0, 4096));
    unsigned* sig = nullptr;
    CK(hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory));
    CK(hipMemset(sig, 0, 8));
    hipEvent_t ev, e0, e1;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int N = 40, iters = argc > 1 ? atoi(argv[1]) : 400;
    unsigned expect = 0;
    for (int mode = 0; mode < 4; ++mode) {
        if (mode >= 2 && !can) break;
        double best = 1e30, best_all = 1e30;
        for (int rep = 0; rep < 8; ++rep) {
            CK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            CK(hipEventRecord(e0, A));
            for (int i = 0; i < N; ++i) {
                hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, A, pa, iters, mode == 2 ? sig : nullptr);
                if (mode == 1) { CK(hipEventRecord(ev, A)); CK(hipStreamWaitEvent(B, ev, 0)); }
                if (mode == 2) { ++expect; CK(hipStreamWaitValue32(B, sig, expect, hipStreamWaitValueGte, 0xffffffffu)); }
                if (mode == 3) { ++expect; CK(hipStreamWriteValue32(A, sig, expect, 0)); CK(hipStreamWaitValue32(B, sig, expect, hipStreamWaitValueGte, 0xffffffffu)); }
                if (mode != 0) hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, B, pb, iters, nullptr);
            }
            CK(hipEventRecord(e1, A));
            CK(hipEventSynchronize(e1));
            CK(hipDeviceSynchronize());
            auto t1 = std::chrono::steady_clock::now();
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
            const double all = std::chrono::duration<double, std::milli>(t1 - t0).count();
            if (all < best_all) best_all = all;
        }
        printf("mode %d (%s): stream A %.1f us per kernel, everything done after %.1f us per kernel\n", mode,
               mode == 0 ? "no fork" : mode == 1 ? "event record + wait" : mode == 2 ? "kernel-written counter + hipStreamWaitValue32" : "hipStreamWriteValue32 + hipStreamWaitValue32", best * 1e3 / N, best_all * 1e3 / N);
    }
    unsigned h = 0;
    CK(hipMemcpy(&h, sig, 4, hipMemcpyDeviceToHost));
    printf("counter %u (expected %u)\n", h, expect);
    return 0;
}
