// Dense contractions on the gfx950 matrix cores with the exact-f32 MFMA
// (v_mfma_f32_16x16x4_f32: f32 in, f32 accumulate, bitwise an fmaf chain):
//   * 1x1 "down" shortcut convolutions                    (audio_nets/tc_resnet.py:30-32)
//   * weight gradients of every convolution -- a [K*Cin] x [positions] x [Cout] contraction over the
//     whole batch (tf.gradients of tf.nn.conv2d wrt the filter, helper/trainer.py:205-211).
//
// Fragment layout of the 16x16x4 f32 MFMA (wave64): A[i = lane & 15][k = lane >> 4],
// B[k = lane >> 4][j = lane & 15], D[row = 4 * (lane >> 4) + reg][col = lane & 15].
#include "kernels.h"

namespace tcr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// D[row = co][col = position] = sum_ci W[ci][co] * x[b][ci][t * stride]
template <int MT, int EPI>
__global__ __launch_bounds__(256) void conv1x1_mfma_kernel(const Conv1x1Args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int pos0 = (blockIdx.x * 4 + wave) * 64;
    const int cot0 = blockIdx.y * MT;

    f32x4 acc[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[m][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const float* xb[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int p = min(pos0 + nt * 16 + r, a.npos - 1);
        const int n = p / a.tout, t = p - n * a.tout;
        xb[nt] = a.x + (size_t)n * a.cin * a.tpi + kHalo + t * a.stride;
    }
    for (int ci0 = 0; ci0 < a.cin; ci0 += 4) {
        const int ci = ci0 + q;
        const bool civ = ci < a.cin;
        float af[MT], bf[4];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int co = (cot0 + m) * 16 + r;
            af[m] = (civ && co < a.cout) ? a.w[(size_t)ci * a.cout + co] : 0.f;
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) bf[nt] = civ ? xb[nt][(size_t)ci * a.tpi] : 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m], bf[nt], acc[m][nt], 0, 0, 0);
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int p = pos0 + nt * 16 + r;
        if (p >= a.npos) continue;
        const int n = p / a.tout, t = p - n * a.tout;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int co = (cot0 + m) * 16 + q * 4 + reg;
                if (co >= a.cout) continue;
                float v = acc[m][nt][reg];
                if (EPI == MF_AFFINE) {
                    v = fmaf(v, a.scale[co], a.shift[co]);
                    if (a.relu) v = fmaxf(v, 0.f);
                }
                float* o = a.y + ((size_t)n * a.cout + co) * a.tpo + kHalo + t;
                o[0] = v;
                if (t == 0) { o[-4] = 0.f; o[-3] = 0.f; o[-2] = 0.f; o[-1] = 0.f; }
                if (t == a.tout - 1) { o[1] = 0.f; o[2] = 0.f; o[3] = 0.f; o[4] = 0.f; }
            }
    }
}

int launch_conv1x1(const Conv1x1Args& a, int epi, hipStream_t s) {
    const int tiles = ceil_div(a.cout, 16);
    const int mt = tiles >= 3 ? 3 : tiles;
    const dim3 grid(ceil_div(a.npos, 256), ceil_div(tiles, mt));
#define TCR_L1(MT_)                                                                                         \
    if (epi == MF_RAW) hipLaunchKernelGGL((conv1x1_mfma_kernel<MT_, MF_RAW>), grid, dim3(256), 0, s, a);    \
    else hipLaunchKernelGGL((conv1x1_mfma_kernel<MT_, MF_AFFINE>), grid, dim3(256), 0, s, a)
    if (mt == 1) { TCR_L1(1); }
    else if (mt == 2) { TCR_L1(2); }
    else { TCR_L1(3); }
#undef TCR_L1
    return check_launch("conv1x1_mfma_kernel");
}

// ---------------------------------------------------------------------------------------------
// Weight gradient: dW[j][ci][co] = sum_{b,t} x[b][ci][t*S + j - pad_lo] * dy[b][co][t].
// Per wave: one 16-channel ci tile, all K taps, NCO co tiles; the reduction index (positions) is the
// MFMA k dimension, 4 positions per instruction.  Workgroups split the position range (split-K);
// the 4 waves of a workgroup are combined through LDS and every workgroup writes one partial
// [K][16][Cout] slab that a second kernel sums in a fixed order (bitwise reproducible).
// ---------------------------------------------------------------------------------------------
struct WgradArgs {
    const float* x;         // [B][Cin][Tpi]   (input activation of the conv, zero halo)
    const float* dy;        // [B][Cout][Tpo]  (gradient wrt the conv output)
    float* partial;         // [nchunk][K][Cin_pad][Cout_pad]
    int npos;               // B * Tout
    int cin, cout, cin_pad, cout_pad;   // cout = width of this output-channel slice
    int cout_all, co_base;              // full channel count of dy / first channel of the slice
    int tpi, tout, tpo, stride;
    int xoff;               // HALO - pad_lo
    int pos_per_block;      // multiple of 16
};

template <int K, int NCO>
__global__ __launch_bounds__(256) void conv_wgrad_mfma_kernel(const WgradArgs a) {
    __shared__ float s_acc[K * 16 * NCO * 16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int ci = blockIdx.y * 16 + r;
    const bool civ = ci < a.cin;
    const int cic = civ ? ci : a.cin - 1;

    f32x4 acc[K][NCO];
#pragma unroll
    for (int j = 0; j < K; ++j)
#pragma unroll
        for (int m = 0; m < NCO; ++m) acc[j][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int blk0 = blockIdx.x * a.pos_per_block;
    const int blk1 = min(blk0 + a.pos_per_block, a.npos);
    // the 4 waves interleave 4-position steps of the block's range
    for (int p4 = blk0 + wave * 4; p4 < blk1; p4 += 16) {
        const int p = p4 + q;
        const bool pv = p < blk1;
        const int pc = pv ? p : blk1 - 1;
        const int n = pc / a.tout, t = pc - n * a.tout;
        const float* xr = a.x + ((size_t)n * a.cin + cic) * a.tpi + t * a.stride + a.xoff;
        const float* dr = a.dy + ((size_t)n * a.cout_all + a.co_base) * a.tpo + kHalo + t;
        float bf[NCO];
#pragma unroll
        for (int m = 0; m < NCO; ++m) {
            const int co = m * 16 + r;
            bf[m] = (pv && co < a.cout) ? dr[(size_t)co * a.tpo] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const float af = (pv && civ) ? xr[j] : 0.f;
#pragma unroll
            for (int m = 0; m < NCO; ++m) acc[j][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf[m], acc[j][m], 0, 0, 0);
        }
    }
    // combine the 4 waves in LDS (fixed order), then write the slab
    for (int wv = 0; wv < 4; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int j = 0; j < K; ++j)
#pragma unroll
                for (int m = 0; m < NCO; ++m)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int row = q * 4 + reg;        // ci within the tile
                        const int idx = ((j * 16 + row) * NCO + m) * 16 + r;
                        if (wv == 0) s_acc[idx] = acc[j][m][reg];
                        else s_acc[idx] += acc[j][m][reg];
                    }
        }
        __syncthreads();
    }
    float* dst = a.partial + (size_t)blockIdx.x * K * a.cin_pad * a.cout_pad;
    for (int i = threadIdx.x; i < K * 16 * NCO * 16; i += 256) {
        const int col = i % (NCO * 16);
        const int row = (i / (NCO * 16)) % 16;
        const int j = i / (NCO * 16 * 16);
        const int cig = blockIdx.y * 16 + row;
        if (cig < a.cin_pad && col < a.cout_pad) dst[((size_t)j * a.cin_pad + cig) * a.cout_pad + col] = s_acc[i];
    }
}

// dw[j][ci][co] = sum_chunk partial[chunk][j][ci][co]   (fixed order)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                           int nchunk, int k, int cin, int cout, int cin_pad, int cout_pad,
                                                           int cout_all, int co_base) {
    const int total = k * cin * cout;
    const size_t slab = (size_t)k * cin_pad * cout_pad;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int co = i % cout;
        const int r = i / cout;
        const int ci = r % cin;
        const int j = r / cin;
        const size_t off = ((size_t)j * cin_pad + ci) * cout_pad + co;
        float s = 0.f;
        for (int c = 0; c < nchunk; ++c) s += partial[(size_t)c * slab + off];
        dw[((size_t)j * cin + ci) * cout_all + co_base + co] = s;
    }
}

int wgrad_chunks(int npos) {
    int n = ceil_div(npos, 1024);       // >= 1024 positions per workgroup
    if (n > 256) n = 256;
    if (n < 1) n = 1;
    return n;
}

size_t wgrad_partial_floats(int k, int cin, int cout, int npos) {
    const int cin_pad = ceil_div(cin, 16) * 16;
    const int cs = cout > 80 ? 80 : cout;
    const int cout_pad = ceil_div(cs, 16) * 16;
    return (size_t)wgrad_chunks(npos) * k * cin_pad * cout_pad;
}

template <int K>
static int launch_wgrad_k(const WgradArgs& a, int nco, dim3 grid, hipStream_t s) {
    switch (nco) {
        case 1: hipLaunchKernelGGL((conv_wgrad_mfma_kernel<K, 1>), grid, dim3(256), 0, s, a); break;
        case 2: hipLaunchKernelGGL((conv_wgrad_mfma_kernel<K, 2>), grid, dim3(256), 0, s, a); break;
        case 3: hipLaunchKernelGGL((conv_wgrad_mfma_kernel<K, 3>), grid, dim3(256), 0, s, a); break;
        case 4: hipLaunchKernelGGL((conv_wgrad_mfma_kernel<K, 4>), grid, dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL((conv_wgrad_mfma_kernel<K, 5>), grid, dim3(256), 0, s, a); break;
    }
    return check_launch("conv_wgrad_mfma_kernel");
}

// dw: [K][Cin][Cout]; scratch: wgrad_partial_floats(...) floats.  Output channels are processed in
// slices of at most 80 (5 MFMA column tiles per wave).
int launch_conv_wgrad(int k, int stride, int pad_lo, const float* x, const float* dy, float* dw, float* scratch,
                      int batch, int cin, int cout, int tpi, int tout, int tpo, hipStream_t s) {
    if (k != 9 && k != 3 && k != 1) { set_error("conv wgrad: kernel %dx1 has no gfx950 instantiation", k); return TCR_ERR_ARG; }
    for (int co_base = 0; co_base < cout; co_base += 80) {
        WgradArgs a;
        a.x = x; a.dy = dy; a.partial = scratch;
        a.npos = batch * tout;
        a.cin = cin;
        a.cout = (cout - co_base) > 80 ? 80 : (cout - co_base);
        a.cout_all = cout; a.co_base = co_base;
        a.cin_pad = ceil_div(cin, 16) * 16;
        a.cout_pad = ceil_div(a.cout, 16) * 16;
        a.tpi = tpi; a.tout = tout; a.tpo = tpo; a.stride = stride;
        a.xoff = kHalo - pad_lo;
        const int nchunk = wgrad_chunks(a.npos);
        a.pos_per_block = ceil_div(ceil_div(a.npos, nchunk), 16) * 16;
        const dim3 grid(ceil_div(a.npos, a.pos_per_block), a.cin_pad / 16);
        const int nco = a.cout_pad / 16;
        int rc;
        if (k == 9) rc = launch_wgrad_k<9>(a, nco, grid, s);
        else if (k == 3) rc = launch_wgrad_k<3>(a, nco, grid, s);
        else rc = launch_wgrad_k<1>(a, nco, grid, s);
        TCR_TRY(rc);
        const int total = k * cin * a.cout;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, s, (const float*)scratch, dw,
                           (int)grid.x, k, cin, a.cout, a.cin_pad, a.cout_pad, cout, co_base);
        TCR_TRY(check_launch("wgrad_reduce_kernel"));
    }
    return TCR_OK;
}

}  // namespace tcr
