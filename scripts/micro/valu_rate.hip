// Micro-benchmark: issue rate of the VALU / LDS instruction kinds the front-end kernels are built from, by waves per SIMD, on this device
// (no memory traffic).  What it answers: how many cycles a SIMD spends per wave-instruction of each kind -- i.e. what "VALU busy" means
// for a kernel made of v_pk_*_f32 -- and whether kinds overlap (VALU beside LDS reads / stores, ds_bpermute, v_permlane16_swap).
// Build: hipcc --offload-arch=gfx950 -O3 scripts/micro/valu_rate.hip -o scripts/micro/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2 __attribute__((ext_vector_type(2)));

enum Kind { K_FMA = 0, K_ADD, K_PK_FMA, K_PK_ADD, K_PK_MUL, K_PK_ADD_SEL, K_CMUL, K_BPERM, K_SWAP16, K_LDS_R64, K_LDS_W64, K_MIX_PK_R64, K_MIX_PK_W64, K_MIX_PK_BPERM,
            K_DEP_PK_FMA, K_DEP_FMA, K_PK_FMA3, K_FMA3, K_PK_FMA_C, K_FMA_C, K_PK_ADD3, K_MUL, K_PK_FMA_SGPR, K_COUNT };
static const char* kNames[K_COUNT] = {"v_fma_f32", "v_add_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_pk_add_f32 op_sel/neg", "complex mul (pk_mul+pk_fma)",
                                      "ds_bpermute_b32", "v_permlane16_swap", "ds_read_b64", "ds_write_b64", "8 pk_fma + 8 ds_read_b64", "8 pk_fma + 8 ds_write_b64",
                                      "8 pk_fma + 8 ds_bpermute", "v_pk_fma_f32 dependent chain", "v_fma_f32 dependent chain", "v_pk_fma_f32 d=a*b+d (3 regs)", "v_fma_f32 d=a*b+d (3 regs)", "pk fma (compiler)", "fma (compiler)", "v_pk_add_f32 d=a+b (3 regs)", "v_mul_f32", "v_pk_fma_f32 d=a*s+d (sgpr b)"};

template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, float s0) {
    __shared__ v2 lds[256 * 17];
    v2 r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = (v2){s0 + i + threadIdx.x, s0 - i};
    const v2 c = (v2){1.0000001f, 0.9999999f};
    v2 a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = (v2){1.0f + 1e-7f * (i + (threadIdx.x & 3)), 1.0f - 1e-7f * i};
    const int lane = threadIdx.x & 63;
    v2* my = lds + threadIdx.x * 17;
#pragma unroll
    for (int i = 0; i < 16; ++i) my[i] = r[i];
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (KIND == K_FMA) { asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(r[i].x) : "v"(c.x)); }
            if (KIND == K_ADD) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i].x) : "v"(c.x)); }
            if (KIND == K_PK_FMA) { asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(r[i]) : "v"(c)); }
            if (KIND == K_PK_ADD) { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(c)); }
            if (KIND == K_PK_MUL) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(c)); }
            if (KIND == K_PK_ADD_SEL) { asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "+v"(r[i]) : "v"(c)); }
            if (KIND == K_CMUL) {
                v2 t;
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(r[i]), "v"(c));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]" : "+v"(r[i]) : "v"(c), "v"(t));
            }
            if (KIND == K_BPERM) { r[i].x = __int_as_float(__builtin_amdgcn_ds_bpermute(((lane + 17) & 63) << 2, __float_as_int(r[i].x))); }
            if (KIND == K_SWAP16) {
                const auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(r[i].x), __float_as_uint(r[i].y), false, false);
                r[i].x = __uint_as_float(q[0]); r[i].y = __uint_as_float(q[1]);
            }
            if (KIND == K_LDS_R64) { asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r[i]) : "v"((unsigned)(threadIdx.x * 17 * 8)), "n"(i * 8)); }
            if (KIND == K_LDS_W64) { asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"((unsigned)(threadIdx.x * 17 * 8)), "v"(r[i]), "n"(i * 8)); }
            if (KIND == K_MIX_PK_R64) {
                if (i < 8) { asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r[i]) : "v"((unsigned)(threadIdx.x * 17 * 8)), "n"(i * 8)); }
                else { asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(r[i]) : "v"(c)); }
            }
            if (KIND == K_MIX_PK_W64) {
                if (i < 8) { asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"((unsigned)(threadIdx.x * 17 * 8)), "v"(r[i]), "n"(i * 8)); }
                else { asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(r[i]) : "v"(c)); }
            }
            if (KIND == K_MIX_PK_BPERM) {
                if (i < 8) { r[i].x = __int_as_float(__builtin_amdgcn_ds_bpermute(((lane + 17) & 63) << 2, __float_as_int(r[i].x))); }
                else { asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(r[i]) : "v"(c)); }
            }
            if (KIND == K_PK_FMA3) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a[i]), "v"(c)); }
            if (KIND == K_FMA3) { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[i].x) : "v"(a[i].x), "v"(c.x)); }
            if (KIND == K_PK_FMA_C) { r[i] = __builtin_elementwise_fma(a[i], c, r[i]); }
            if (KIND == K_FMA_C) { r[i].x = __builtin_fmaf(a[i].x, c.x, r[i].x); }
            if (KIND == K_PK_ADD3) { asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r[i]) : "v"(a[i]), "v"(r[(i + 5) & 15])); }
            if (KIND == K_MUL) { asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i].x) : "v"(c.x)); }
            if (KIND == K_PK_FMA_SGPR) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a[i]), "s"(c)); }
            if (KIND == K_DEP_PK_FMA) { asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(r[0]) : "v"(c)); }
            if (KIND == K_DEP_FMA) { asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(r[0].x) : "v"(c.x)); }
        }
        if (KIND == K_LDS_R64 || KIND == K_MIX_PK_R64) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += r[i].x + r[i].y + a[i].x;
    if (s == 123.456f) out[0] = s + my[3].x;
}

template <int KIND>
static void run(float* out, int cus, double ghz) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    printf("%-30s", kNames[KIND]);
    for (int wg = 1; wg <= 4; ++wg) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((rate_kernel<KIND>), dim3(cus * wg), dim3(256), 0, 0, out, iters, 1.0f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best) best = ms;
        }
        // cycles a SIMD spends per wave-instruction of the loop body (16 per iteration and wave; wg waves per SIMD)
        const double cyc = best * 1e-3 * ghz * 1e9 / ((double)iters * 16 * wg);
        printf("  %dw: %6.2f cyc/inst", wg, cyc);
    }
    printf("\n");
}

int main() {
    float* out;
    hipMalloc(&out, 4);
    int cus = 0, clk = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    const double ghz = clk * 1e-6;
    {   // clocks ramp with load: ~1.5 s of work before anything is timed (a fresh box reports rates 2x off otherwise)
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        float total = 0.f;
        while (total < 1500.f) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((rate_kernel<K_PK_FMA3>), dim3(cus * 2), dim3(256), 0, 0, out, 200000, 1.0f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            total += ms;
        }
    }
    printf("CUs %d, clock %.2f GHz; cycles per wave-instruction and SIMD at 1..4 waves per SIMD (lower = faster; 'cmul' counts TWO instructions per slot)\n", cus, ghz);
    run<K_FMA>(out, cus, ghz); run<K_ADD>(out, cus, ghz); run<K_PK_FMA>(out, cus, ghz); run<K_PK_ADD>(out, cus, ghz); run<K_PK_MUL>(out, cus, ghz);
    run<K_PK_ADD_SEL>(out, cus, ghz); run<K_CMUL>(out, cus, ghz); run<K_BPERM>(out, cus, ghz); run<K_SWAP16>(out, cus, ghz);
    run<K_LDS_R64>(out, cus, ghz); run<K_LDS_W64>(out, cus, ghz); run<K_MIX_PK_R64>(out, cus, ghz); run<K_MIX_PK_W64>(out, cus, ghz);
    run<K_MIX_PK_BPERM>(out, cus, ghz); run<K_DEP_PK_FMA>(out, cus, ghz); run<K_DEP_FMA>(out, cus, ghz);
    run<K_PK_FMA3>(out, cus, ghz); run<K_FMA3>(out, cus, ghz); run<K_PK_FMA_C>(out, cus, ghz); run<K_FMA_C>(out, cus, ghz); run<K_PK_ADD3>(out, cus, ghz); run<K_MUL>(out, cus, ghz); run<K_PK_FMA_SGPR>(out, cus, ghz);
    return 0;
}
