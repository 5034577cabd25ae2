// TEST INFRASTRUCTURE ONLY -- a host-side stand-in for <hip/hip_runtime.h>.
//
// The build container has no GPU, and GPU minutes on the MI355X pool are scarce, so the
// kernel sources under tc-resnet_amd/csrc are ALSO compiled with plain clang++ against this
// header (`-I tests/emu` ahead of the ROCm include path) into `tests/emu/_build/libtcr_emu.so`.
// Every workgroup is run as a set of cooperative fibers on one OS thread, so `__syncthreads`,
// wave-64 shuffles and the f32 MFMA builtins behave as on gfx950 (lane/fragment layouts per
// /opt/skills/guides/cdna_hip_programming.md section 3).  This catches indexing, layout and
// synchronisation bugs on the CPU *before* a gpurun call.
//
// It is NOT a product path, NOT a fallback and NOT the oracle: the product loader
// (tc-resnet_amd/_lib.py) only ever opens the hipcc-built gfx950 library and fails loudly
// without it; only tests/ load the emulator build.
#pragma once
#include <chrono>
#include <ucontext.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define TCR_HOST_EMULATION 1

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define TCR_WAVES_PER_SIMD(n)           // (a register-budget attribute of the gfx950 compiler: nothing to emulate)
#define __constant__ static const

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_emu { unsigned x, y, z; };
struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct int2 { int x, y; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }

typedef void* hipStream_t;
typedef void* hipEvent_t;
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1 };
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipPeekAtLastError() { return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 0 };
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 4; return hipSuccess; }   // tiny "device": exercises the grid-shaping policies
static inline const char* hipGetErrorString(hipError_t) { return "emulated"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }    // (the emulator's dynamic LDS is one fixed 160 KB buffer)
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
#define hipEventDisableTiming 0x2
#define hipEventDisableSystemFence 0x20000000
#define hipStreamNonBlocking 0x1
// (distinct handles, never dereferenced: the host code takes its multi-stream branches; launches still run in issue order)
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { static char ids[64]; static int n = 0; *s = &ids[n++ & 63]; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
#define hipErrorNotReady 600
enum hipStreamCaptureStatus { hipStreamCaptureStatusNone = 0, hipStreamCaptureStatusActive = 1 };
static inline hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus* st) { *st = hipStreamCaptureStatusNone; return hipSuccess; }
static inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }      // (launches are synchronous: nothing ever overtakes)
static inline long long wall_clock64() { return (long long)(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() / 10); }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }

namespace emu {

constexpr int WAVE = 64;
constexpr size_t STACK_BYTES = 512 * 1024;

enum State { READY = 0, WAIT_BLOCK = 1, WAIT_WAVE = 2, DONE = 3 };

// Fiber switch: the callee-saved registers and the stack pointer, nothing else.  (glibc's swapcontext also saves and restores the signal
// mask -- two rt_sigprocmask system calls per switch, and every emulated wave-level operation is a switch: a third of the CPU suite's
// time was spent in the kernel.)  x86-64 System V only; other hosts keep ucontext.
#if defined(__x86_64__) && !defined(TCR_EMU_UCONTEXT)
#define TCR_EMU_ASM_SWITCH 1
extern "C" void tcr_emu_switch(void** save_sp, void* load_sp);
__asm__(R"(
    .text
    .weak tcr_emu_switch
    .type tcr_emu_switch,@function
tcr_emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size tcr_emu_switch, .-tcr_emu_switch
)");
#else
#define TCR_EMU_ASM_SWITCH 0
#endif

struct Fiber {
#if TCR_EMU_ASM_SWITCH
    void* sp = nullptr;
#else
    ucontext_t ctx;
#endif
    char* stack = nullptr;
    State state = DONE;
    dim3 tid;
};

struct Runtime {
    std::vector<Fiber> fibers;
#if TCR_EMU_ASM_SWITCH
    void* main_sp = nullptr;
#else
    ucontext_t main_ctx;
#endif
    int cur = -1;
    int nthreads = 0;
    const std::function<void()>* body = nullptr;
    uint64_t xchg[16][WAVE][4];   // per-wave exchange slots (up to 4 x 64-bit words per lane)
};

inline Runtime& rt() { static Runtime r; return r; }

inline dim3& tidx() { static dim3 v; return v; }
inline dim3& bidx() { static dim3 v; return v; }
inline dim3& bdim() { static dim3 v; return v; }
inline dim3& gdim() { static dim3 v; return v; }

inline void fiber_entry() {
    Runtime& r = rt();
    (*r.body)();
    r.fibers[r.cur].state = DONE;
#if TCR_EMU_ASM_SWITCH
    tcr_emu_switch(&r.fibers[r.cur].sp, r.main_sp);
    __builtin_trap();                       // (a finished fiber is never resumed)
#else
    swapcontext(&r.fibers[r.cur].ctx, &r.main_ctx);
#endif
}

inline void yield_as(State s) {
    Runtime& r = rt();
    int me = r.cur;
    r.fibers[me].state = s;
#if TCR_EMU_ASM_SWITCH
    tcr_emu_switch(&r.fibers[me].sp, r.main_sp);
#else
    swapcontext(&r.fibers[me].ctx, &r.main_ctx);
#endif
}

inline void block_barrier() { yield_as(WAIT_BLOCK); }
inline void wave_barrier() { yield_as(WAIT_WAVE); }

inline int lane_id() { return rt().cur & (WAVE - 1); }
inline int wave_id() { return rt().cur / WAVE; }

inline void run_block() {
    Runtime& r = rt();
    const int n = r.nthreads;
    const dim3 bd = bdim();
    if ((int)r.fibers.size() < n) r.fibers.resize(n);
    for (int i = 0; i < n; ++i) {
        Fiber& f = r.fibers[i];
        if (!f.stack) f.stack = (char*)malloc(STACK_BYTES);
#if TCR_EMU_ASM_SWITCH
        // initial frame: six zeroed callee-saved registers, the entry point as the return address, a null return address for the entry
        // (16-byte aligned top - 64: the entry point starts with rsp = 8 mod 16, as after a call)
        uintptr_t top = (reinterpret_cast<uintptr_t>(f.stack) + STACK_BYTES) & ~uintptr_t(15);
        void** frame = reinterpret_cast<void**>(top - 64);
        for (int k = 0; k < 8; ++k) frame[k] = nullptr;
        frame[6] = reinterpret_cast<void*>(&fiber_entry);
        f.sp = frame;
#else
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = STACK_BYTES;
        f.ctx.uc_link = &r.main_ctx;
        makecontext(&f.ctx, (void (*)())fiber_entry, 0);
#endif
        f.state = READY;
        f.tid = dim3(i % bd.x, (i / bd.x) % bd.y, i / (bd.x * bd.y));
    }
    int done = 0;
    while (done < n) {
        bool progressed = false;
        for (int i = 0; i < n; ++i) {
            Fiber& f = r.fibers[i];
            if (f.state != READY) continue;
            r.cur = i;
            tidx() = f.tid;
#if TCR_EMU_ASM_SWITCH
            tcr_emu_switch(&r.main_sp, f.sp);
#else
            swapcontext(&r.main_ctx, &f.ctx);
#endif
            progressed = true;
            if (f.state == DONE) ++done;
        }
        // release wave barriers
        const int nw = (n + WAVE - 1) / WAVE;
        for (int w = 0; w < nw; ++w) {
            int waiting = 0, alive = 0;
            for (int l = w * WAVE; l < std::min(n, (w + 1) * WAVE); ++l) {
                if (r.fibers[l].state != DONE) ++alive;
                if (r.fibers[l].state == WAIT_WAVE) ++waiting;
            }
            if (alive && waiting == alive) {
                for (int l = w * WAVE; l < std::min(n, (w + 1) * WAVE); ++l)
                    if (r.fibers[l].state == WAIT_WAVE) r.fibers[l].state = READY;
                progressed = true;
            }
        }
        // release the block barrier
        int waiting = 0, alive = 0;
        for (int i = 0; i < n; ++i) {
            if (r.fibers[i].state != DONE) ++alive;
            if (r.fibers[i].state == WAIT_BLOCK) ++waiting;
        }
        if (alive && waiting == alive) {
            for (int i = 0; i < n; ++i) r.fibers[i].state = READY;
            progressed = true;
        }
        if (!progressed && done < n) {
            fprintf(stderr, "[hip-emu] deadlock: divergent barrier/shuffle in block (%u,%u,%u)\n", bidx().x, bidx().y, bidx().z);
            abort();
        }
    }
    r.cur = -1;
}

inline void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    Runtime& r = rt();
    r.body = &body;
    r.nthreads = (int)(block.x * block.y * block.z);
    if (r.nthreads > 1024 || r.nthreads <= 0) { fprintf(stderr, "[hip-emu] bad block size %d\n", r.nthreads); abort(); }
    gdim() = grid;
    bdim() = block;
    for (unsigned z = 0; z < grid.z; ++z)
        for (unsigned y = 0; y < grid.y; ++y)
            for (unsigned x = 0; x < grid.x; ++x) {
                bidx() = dim3(x, y, z);
                run_block();
            }
}

template <class T>
inline T shfl(T v, int src) {
    static_assert(sizeof(T) <= 8, "shfl payload");
    Runtime& r = rt();
    const int w = wave_id(), l = lane_id();
    uint64_t bits = 0;
    memcpy(&bits, &v, sizeof(T));
    r.xchg[w][l][0] = bits;
    wave_barrier();
    uint64_t o = r.xchg[w][src & (WAVE - 1)][0];
    wave_barrier();
    T out;
    memcpy(&out, &o, sizeof(T));
    return out;
}

typedef float f32x4_emu __attribute__((ext_vector_type(4)));
typedef float f32x16_emu __attribute__((ext_vector_type(16)));

// v_mfma_f32_16x16x4_f32: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; D: col=l&15, row=(l>>4)*4+reg.
// Numerically a k-ordered fmaf chain (guide section 3, "FP32-input MFMA").
inline f32x4_emu mfma_16x16x4(float a, float b, f32x4_emu c) {
    Runtime& r = rt();
    const int w = wave_id(), l = lane_id();
    float ab[2] = {a, b};
    memcpy(&r.xchg[w][l][0], ab, 8);
    wave_barrier();
    const int col = l & 15;
    f32x4_emu d = c;
    for (int reg = 0; reg < 4; ++reg) {
        const int row = (l >> 4) * 4 + reg;
        float acc = c[reg];
        for (int k = 0; k < 4; ++k) {
            float av[2], bv[2];
            memcpy(av, &r.xchg[w][row + 16 * k][0], 8);
            memcpy(bv, &r.xchg[w][col + 16 * k][0], 8);
            acc = fmaf(av[0], bv[1], acc);
        }
        d[reg] = acc;
    }
    wave_barrier();
    return d;
}

// v_mfma_f32_32x32x2_f32: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]; D: col=l&31, row=(reg&3)+8*(reg>>2)+4*(l>>5).
inline f32x16_emu mfma_32x32x2(float a, float b, f32x16_emu c) {
    Runtime& r = rt();
    const int w = wave_id(), l = lane_id();
    float ab[2] = {a, b};
    memcpy(&r.xchg[w][l][0], ab, 8);
    wave_barrier();
    const int col = l & 31;
    f32x16_emu d = c;
    for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5);
        float acc = c[reg];
        for (int k = 0; k < 2; ++k) {
            float av[2], bv[2];
            memcpy(av, &r.xchg[w][row + 32 * k][0], 8);
            memcpy(bv, &r.xchg[w][col + 32 * k][0], 8);
            acc = fmaf(av[0], bv[1], acc);
        }
        d[reg] = acc;
    }
    wave_barrier();
    return d;
}

}  // namespace emu

#define threadIdx (emu::tidx())
#define blockIdx (emu::bidx())
#define blockDim (emu::bdim())
#define gridDim (emu::gdim())
static const int warpSize = 64;

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    emu::launch(dim3(grid), dim3(block), [&]() { (kernel)(__VA_ARGS__); })

static inline void __syncthreads() { emu::block_barrier(); }
template <class T> static inline T __shfl(T v, int src, int width = 64) {
    int l = emu::lane_id();
    int base = l & ~(width - 1);
    return emu::shfl(v, base + (src & (width - 1)));
}
template <class T> static inline T __shfl_xor(T v, int mask, int width = 64) {
    (void)width;
    return emu::shfl(v, emu::lane_id() ^ mask);
}
template <class T> static inline T __shfl_down(T v, unsigned d, int width = 64) {
    int l = emu::lane_id();
    int src = ((l & (width - 1)) + (int)d < width) ? l + (int)d : l;
    return emu::shfl(v, src);
}
template <class T> static inline T __shfl_up(T v, unsigned d, int width = 64) {
    int l = emu::lane_id();
    int src = ((l & (width - 1)) >= (int)d) ? l - (int)d : l;
    return emu::shfl(v, src);
}
// v_mov_b32 with a DPP operand, for the row-local controls the kernels use (all lanes active, bound_ctrl: a lane without a source
// would read 0 -- none of these controls leaves one): quad_perm (0x00-0xFF), row_mirror (0x140), row_half_mirror (0x141).
static inline int __builtin_amdgcn_update_dpp_emu(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    (void)old; (void)row_mask; (void)bank_mask; (void)bound_ctrl;
    const int l = emu::lane_id();
    int from = l;
    if (ctrl >= 0 && ctrl <= 0xFF) from = (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3);
    else if (ctrl == 0x140) from = (l & ~15) | (15 - (l & 15));
    else if (ctrl == 0x141) from = (l & ~7) | (7 - (l & 7));
    else { fprintf(stderr, "[hip-emu] DPP control 0x%x is not emulated\n", ctrl); abort(); }
    return emu::shfl(src, from);
}
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) __builtin_amdgcn_update_dpp_emu(old, src, ctrl, rm, bm, bc)
static inline void __builtin_amdgcn_wave_barrier_emu() { emu::wave_barrier(); }
#define __builtin_amdgcn_wave_barrier() __builtin_amdgcn_wave_barrier_emu()
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(mask, size, id) ((void)0)
#define __builtin_amdgcn_sched_barrier(mask) ((void)0)
#define __builtin_amdgcn_s_sleep(n) ((void)0)
static inline unsigned __brev(unsigned v) {
    unsigned r = 0;
    for (int i = 0; i < 32; ++i) r |= ((v >> i) & 1u) << (31 - i);
    return r;
}
static inline int __builtin_amdgcn_readfirstlane_emu(int v) { return emu::shfl(v, 0); }
#define __builtin_amdgcn_readfirstlane(v) __builtin_amdgcn_readfirstlane_emu(v)
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) emu::mfma_16x16x4((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) emu::mfma_32x32x2((a), (b), (c))

static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
static inline void __threadfence() {}

static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
static inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
static inline float min(float a, float b) { return fminf(a, b); }
static inline float max(float a, float b) { return fmaxf(a, b); }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
// round-to-nearest single operations that the compiler must not contract into an fma
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
static inline float __frcp_rn(float x) { return 1.0f / x; }
static inline float __fdividef(float a, float b) { return a / b; }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((uint64_t)a * b) >> 32); }
