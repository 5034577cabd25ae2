// TEST INFRASTRUCTURE: host stand-in of tc-resnet_amd/csrc/gfx950_isa.h (same names, plain C++), found first on the emulator build's
// include path (-I tests/emu).  Never part of the product build.
#pragma once
#include <hip/hip_runtime.h>

namespace tcr {

inline char* dyn_lds() {
    alignas(16) static char buf[160 * 1024];
    return buf;
}

__device__ __forceinline__ int opaque_zero() { return 0; }
__device__ __forceinline__ int opaque_zero_after(float) { return 0; }

typedef float v2 __attribute__((ext_vector_type(2)));
typedef float v4 __attribute__((ext_vector_type(4)));


// (host stand-in: the DMA copy done at once by the calling lane)
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    memcpy(static_cast<char*>(lds_wave_base) + 16 * (threadIdx.x & 63), gsrc, 16);
}

__device__ __forceinline__ void wait_dma() {}

// (host stand-in of the buffer-descriptor loads: base + lane offset + uniform offset, in bytes)
typedef const char* buf_rsrc;
__device__ __forceinline__ buf_rsrc make_rsrc(const void* base) { return static_cast<const char*>(base); }
__device__ __forceinline__ float buf_load_f32(buf_rsrc r, unsigned lane_off_bytes, unsigned uniform_off_bytes) {
    float v;
    memcpy(&v, r + lane_off_bytes + uniform_off_bytes, 4);
    return v;
}

// a - i b = (a.x + b.y, a.y - b.x)
__device__ __forceinline__ v2 c_submi(v2 a, v2 b) {
    return (v2){a.x + b.y, a.y - b.x};
}
// a + i b = (a.x - b.y, a.y + b.x)
__device__ __forceinline__ v2 c_addmi(v2 a, v2 b) {
    return (v2){a.x - b.y, a.y + b.x};
}
// a + conj(b), a - conj(b)
__device__ __forceinline__ v2 c_addc(v2 a, v2 b) {
    return (v2){a.x + b.x, a.y - b.y};
}
__device__ __forceinline__ v2 c_subc(v2 a, v2 b) {
    return (v2){a.x - b.x, a.y + b.y};
}
// a * b:  t = a.y * (b.y, b.x);  r = (fma(a.x, b.x, -t.x), fma(a.x, b.y, t.y))
__device__ __forceinline__ v2 c_mul(v2 a, v2 b) {
    const float tx = a.y * b.y, ty = a.y * b.x;
    return (v2){fmaf(a.x, b.x, -tx), fmaf(a.x, b.y, ty)};
}
// a * b with a wave-uniform constant b (scalar register pair)
__device__ __forceinline__ v2 c_mulk(v2 a, v2 b) {
    return c_mul(a, b);
}
// a * conj(b):  r = (fma(a.x, b.x, t.x), fma(-a.x, b.y, t.y))
__device__ __forceinline__ v2 c_mulc(v2 a, v2 b) {
    const float tx = a.y * b.y, ty = a.y * b.x;
    return (v2){fmaf(a.x, b.x, tx), fmaf(-a.x, b.y, ty)};
}

// Cross-lane moves of the real-FFT split (no LDS round trip):
//   row_swap: the odd 16-lane rows of `a` trade places with the even rows of `b` (v_permlane16_swap_b32) -- a frame's two
//             256-point units sit in adjacent rows, so one swap per register pair hands every lane E[k] and O[k] of ITS bins;
//   lane_gather: value of an arbitrary lane (ds_bpermute_b32: the LDS crossbar, no memory, no bank conflicts).
__device__ __forceinline__ void row_swap(float& a, float& b, int lane) {
    const float ax = __shfl_xor(a, 16), bx = __shfl_xor(b, 16);
    const bool odd = (lane >> 4) & 1;
    const float na = odd ? bx : a, nb = odd ? b : ax;
    a = na;
    b = nb;
}
__device__ __forceinline__ float lane_gather(float v, int src_lane) {
    return __shfl(v, src_lane);
}

// ln(x) for normal positive x (the mel energies are >= 1e-12): v_log_f32 (1 ulp) * ln 2 -- two instructions against the ~15 of the
// library call's denormal / special-value handling; 2e-7 relative, far inside the 1e-4 budget of the MFCCs.
__device__ __forceinline__ float fast_log(float x) {
    return logf(x);
}

__device__ __forceinline__ void pk_sq_pair(v2 A, v2 C, float& p_lo, float& p_hi) {
    const v2 X = A + C, Y = A - C;
    p_lo = fmaf(X.x, X.x, X.y * X.y);
    p_hi = fmaf(Y.x, Y.x, Y.y * Y.y);
}

}  // namespace tcr
