// Shared host/device helpers for the gfx950 TC-ResNet kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstddef>
#include <cstdint>
#include <cstdio>

#include "../../include/tcresnet_hip.h"
#include <gfx950_isa.h>      // raw-ISA idioms (inline asm, LDS base, cross-lane moves); tests/emu supplies a host stand-in under the same name

namespace tcr {

// Pins a kernel to n waves per SIMD (the register allocator's budget: 512 / n VGPRs + AGPRs).  An attribute of the gfx950 compiler;
// a host build of the kernel sources (tests/emu) pre-defines the macro as empty.
#ifndef TCR_WAVES_PER_SIMD
#define TCR_WAVES_PER_SIMD(n) __attribute__((amdgpu_waves_per_eu(n, n)))
#endif

constexpr int kHalo = TCR_HALO;
constexpr int kWave = 64;

// ---- error plumbing (thread-local message behind tcr_last_error) --------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);      // hipGetLastError -> TCR_OK / TCR_ERR_HIP
// the library's internal streams (0: filter gradients / shortcut branch, 1: classifier gradients, 2 / 3: tcr_internal_stream), chosen on first
// use so that they do not share a hardware queue with `caller` (the stream of that first call) or each other; nullptr on failure
hipStream_t shared_stream(int idx, hipStream_t caller);
// flags of the library's fork / join events (stream-to-stream ordering on ONE device, never inspected by the host)
unsigned internal_event_flags();
int tune_get(int knob);                  // process-wide tuning knobs (tcr_tune)
int device_cus();                        // compute units of the current device (cached per device; 256 on MI355X)

#define TCR_REQUIRE(cond, ...)                 \
    do {                                       \
        if (!(cond)) {                         \
            ::tcr::set_error(__VA_ARGS__);     \
            return TCR_ERR_ARG;                \
        }                                      \
    } while (0)

#define TCR_TRY(expr)                          \
    do {                                       \
        int _s = (expr);                       \
        if (_s != TCR_OK) return _s;           \
    } while (0)

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int64_t round_up64(int64_t a, int64_t b) { return ceil_div64(a, b) * b; }

// TF "SAME" padding rule (extra padding on the high side).
inline void same_pad(int len, int k, int stride, int* out, int* lo, int* hi) {
    *out = (len + stride - 1) / stride;
    int total = (*out - 1) * stride + k - len;
    if (total < 0) total = 0;
    *lo = total / 2;
    *hi = total - *lo;
}

// ---- device helpers -------------------------------------------------------------------------
// Ordering point between LDS phases that stay INSIDE one wavefront: a wave's DS operations execute in
// order, so no s_barrier is needed -- only the compiler must not move LDS accesses across the point.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// (dyn_lds(), opaque_zero(): gfx950_isa.h)

// n / d for 0 <= n < 2^22, 1 <= d: float multiply + one-step fix-up (a 32-bit integer division by a run-time value is
// ~25 dependent VALU instructions on this ISA).  inv_d = 1.0f / d.
__device__ __forceinline__ int fast_div(int n, int d, float inv_d) {
    int qt = (int)(((float)n + 0.5f) * inv_d);
    qt += (qt + 1) * d <= n ? 1 : (qt * d > n ? -1 : 0);
    return qt;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

// Sum over the 16 lanes of a DPP row (lanes 16 k .. 16 k + 15; every lane of the row gets the total): four v_add_f32 with
// DPP operands -- quad xor 1, quad xor 2, row_half_mirror, row_mirror -- on the VALU, no LDS-pipe round trip (a bpermute per
// step).  After steps 1-2 all lanes of a quad agree, so the mirrors pair every quad with the one an xor 4 / xor 8 would:
// the result is bitwise the xor-butterfly's (same pairs, commutative adds).
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
    return v;
}

// Stateless counter-based uniform in [0,1): two rounds of a 32-bit mixer over (seed, index).
// Mirrored bit-for-bit in tests (numpy) so that dropout masks are reproducible on the host.
__host__ __device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU;
    x ^= x >> 15; x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ float uniform01(uint64_t seed, uint64_t index) {
    uint32_t lo = (uint32_t)index, hi = (uint32_t)(index >> 32);
    uint32_t s0 = (uint32_t)seed, s1 = (uint32_t)(seed >> 32);
    uint32_t h = mix32(lo ^ mix32(hi ^ mix32(s0 ^ mix32(s1 + 0x9e3779b9U))));
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}

}  // namespace tcr
