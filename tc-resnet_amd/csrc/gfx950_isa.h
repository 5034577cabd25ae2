// The raw-ISA idioms of the gfx950 kernels in ONE place: inline assembly (packed-FP32 VOP3P forms with op_sel / neg modifiers the
// compiler expands to 2-4 instructions, an opaque zero), the dynamic-LDS base, cross-lane moves (v_permlane16_swap, ds_bpermute) and
// v_log_f32.  Product sources include it as <gfx950_isa.h>; the host-side emulator build of the same sources (tests/emu: test
// infrastructure) puts its own plain-C++ stand-in of this header first on the include path, so no kernel source carries an emulation branch.
#pragma once
#include <hip/hip_runtime.h>

namespace tcr {

// Base of the dynamic LDS region (launch-time size, third hipLaunchKernelGGL argument).
__device__ __forceinline__ char* dyn_lds() {
    extern __shared__ __attribute__((aligned(16))) char tcr_dyn_lds[];
    return tcr_dyn_lds;
}

// A zero the optimiser cannot see through.  Adding it to a table pointer keeps loop-invariant table
// loads INSIDE the loop (they hit L1/K$) instead of being hoisted into dozens of long-lived VGPRs.
__device__ __forceinline__ int opaque_zero() {
    int z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    return z;
}

// The same zero, produced only after `dep` exists: pins loads whose address uses it behind the computation of `dep`.
__device__ __forceinline__ int opaque_zero_after(float dep) {
    int z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z) : "v"(dep));
    return z;
}

typedef float v2 __attribute__((ext_vector_type(2)));
typedef float v4 __attribute__((ext_vector_type(4)));


// 16 bytes per lane, global -> LDS by the DMA path (global_load_lds_dwordx4): lane l's 16 bytes land at lds_wave_base + 16 l --
// `lds_wave_base` must be wave-uniform, the global address is per lane.  Asynchronous: counted on vmcnt; a barrier built from
// __syncthreads() drains it (its fence waits vmcnt(0)).
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(reinterpret_cast<const __attribute__((address_space(1))) void*>(reinterpret_cast<uintptr_t>(gsrc)),
                                     reinterpret_cast<__attribute__((address_space(3))) void*>(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(lds_wave_base))), 16, 0, 0);
}

// Buffer-descriptor loads: `buffer_load_dword v, v_offset, s[rsrc], s_offset offen` -- the address is a wave-uniform 128-bit descriptor
// (base pointer in scalar registers) + a per-lane 32-bit byte offset + a wave-uniform 32-bit byte offset, i.e. NO 64-bit vector address
// arithmetic in front of a load whose lane part never changes and whose running part is uniform (flat / global loads of that shape
// compile to a v_lshl_add_u64 per load once the zero-extension of the lane offset has been hoisted out of the loop).  The base must be
// provably wave-uniform (kernel arguments / blockIdx-derived scalars), else the compiler wraps every load in a waterfall loop.  The
// compiler counts these loads on vmcnt like plain ones.  Bounds: 2 GB behind `base` (reads past it return 0).
typedef __amdgpu_buffer_rsrc_t buf_rsrc;
__device__ __forceinline__ buf_rsrc make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ float buf_load_f32(buf_rsrc r, unsigned lane_off_bytes, unsigned uniform_off_bytes) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)lane_off_bytes, (int)uniform_off_bytes, 0));
}

// waits for this wave's outstanding glds16 copies (and any other vector-memory load): s_waitcnt vmcnt(0)
__device__ __forceinline__ void wait_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// a - i b = (a.x + b.y, a.y - b.x)
__device__ __forceinline__ v2 c_submi(v2 a, v2 b) {
    v2 d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// a + i b = (a.x - b.y, a.y + b.x)
__device__ __forceinline__ v2 c_addmi(v2 a, v2 b) {
    v2 d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// a + conj(b), a - conj(b)
__device__ __forceinline__ v2 c_addc(v2 a, v2 b) {
    v2 d;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ v2 c_subc(v2 a, v2 b) {
    v2 d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// a * b:  t = a.y * (b.y, b.x);  r = (fma(a.x, b.x, -t.x), fma(a.x, b.y, t.y))
__device__ __forceinline__ v2 c_mul(v2 a, v2 b) {
    v2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
    return r;
}
// a * b with a wave-uniform constant b (scalar register pair)
__device__ __forceinline__ v2 c_mulk(v2 a, v2 b) {
    v2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "s"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]" : "=v"(r) : "v"(a), "s"(b), "v"(t));
    return r;
}
// a * conj(b):  r = (fma(a.x, b.x, t.x), fma(-a.x, b.y, t.y))
__device__ __forceinline__ v2 c_mulc(v2 a, v2 b) {
    v2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
    return r;
}

// Cross-lane moves of the real-FFT split (no LDS round trip):
//   row_swap: the odd 16-lane rows of `a` trade places with the even rows of `b` (v_permlane16_swap_b32) -- a frame's two
//             256-point units sit in adjacent rows, so one swap per register pair hands every lane E[k] and O[k] of ITS bins;
//   lane_gather: value of an arbitrary lane (ds_bpermute_b32: the LDS crossbar, no memory, no bank conflicts).
__device__ __forceinline__ void row_swap(float& a, float& b, int lane) {
    (void)lane;
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}
__device__ __forceinline__ float lane_gather(float v, int src_lane) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v)));
}

// ln(x) for normal positive x (the mel energies are >= 1e-12): v_log_f32 (1 ulp) * ln 2 -- two instructions against the ~15 of the
// library call's denormal / special-value handling; 2e-7 relative, far inside the 1e-4 budget of the MFCCs.
__device__ __forceinline__ float fast_log(float x) {
    return __builtin_amdgcn_logf(x) * 0.69314718055994530942f;
}

// (X.x, Y.x) and (X.y, Y.y) of X = A + C, Y = A - C packed side by side: both squared magnitudes in one pk_mul + one pk_fma
__device__ __forceinline__ void pk_sq_pair(v2 A, v2 C, float& p_lo, float& p_hi) {
    v2 xs, ys;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[0,1]" : "=v"(xs) : "v"(A), "v"(C));
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,1] neg_hi:[0,1]" : "=v"(ys) : "v"(A), "v"(C));
    const v2 p = __builtin_elementwise_fma(xs, xs, ys * ys);
    p_lo = p.x;
    p_hi = p.y;
}

}  // namespace tcr
