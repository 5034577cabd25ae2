// Kernels of the generic 2-D layer-graph engine (net2d.cpp): the other model families behind the reference's factory --
// ResNet2D8(/Pool) (audio_nets/tc_resnet.py:14-15,73-99), Res8/15(/Narrow) (audio_nets/res.py:6-123) and the KWSModel
// architectures (audio_nets/kws.py:15-63) -- SURVEY 8(f) #4.
//
// Layout: planar per utterance, plane = H*W floats at offset kHalo inside a row of H*W + 2*kHalo (the same row convention
// as the 1-D path, so the BatchNorm kernels of bn.hip apply unchanged with "time" = the flattened plane position).
//
// Convolutions (any kernel / stride / dilation, SAME or VALID, optional bias + ReLU; fully connected layers are VALID
// convolutions whose kernel covers the whole plane -- a TF `reshape(-1, H*W*C) @ W` has exactly the HWIO weight layout)
// run as implicit GEMMs on the exact-f32 16x16x4 MFMA: D[channel][position] with the reduction over (tap, channel quad);
// the data gradient is the same kernel in gather form over dy (stride divisibility as a predicate), the filter gradient is
// a [Cin] x [positions] x [Cout] contraction per tap, split over utterance chunks and summed in a fixed order.
#include "kernels.h"
#include "net2d.h"

namespace tcr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- conv2d forward (DGRAD = false) / data gradient (DGRAD = true) ------------------------------------------------------
// forward:  y[n][co][oh][ow]  = act(bias[co] + sum_{i,j,ci} x[n][ci][oh*sh + i*dh - pt][ow*sw + j*dw - pl] * W[i][j][ci][co])
// dgrad:    dx[n][ci][ih][iw] += sum_{i,j,co} dy[n][co][(ih + pt - i*dh) / sh][(iw + pl - j*dw) / sw] * W[i][j][ci][co]   (exact divisions only)
// In both, "M" = the output channel dimension of the launch, "KC" = the reduced channel dimension.
template <int MT, bool DGRAD>
__global__ __launch_bounds__(256) void conv2d_mfma_kernel(const Conv2dArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int M = DGRAD ? a.cin : a.cout, KC = DGRAD ? a.cout : a.cin;
    const int PH = DGRAD ? a.h : a.oh, PW = DGRAD ? a.w : a.ow;         // plane the launch's positions enumerate
    const int SH = DGRAD ? a.oh : a.h, SW = DGRAD ? a.ow : a.w;         // plane the operand is gathered from
    const int spp = DGRAD ? a.ppo : a.ppi, dpp = DGRAD ? a.ppi : a.ppo;
    const int plane = PH * PW;
    const int npos = a.batch * plane;
    const int pos0 = (blockIdx.x * 4 + wave) * 32;
    if (pos0 >= npos) return;
    const int cot0 = blockIdx.y * MT;

    int pn[2], py[2], px[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int p = min(pos0 + nt * 16 + r, npos - 1);
        pn[nt] = p / plane;
        const int rem = p - pn[nt] * plane;
        py[nt] = rem / PW;
        px[nt] = rem - py[nt] * PW;
    }
    f32x4 acc[MT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[m][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int mrow[MT];
    bool mval[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int ch = (cot0 + m) * 16 + r;
        mval[m] = ch < M;
        mrow[m] = mval[m] ? ch : 0;
    }
    const int c4n = (KC + 3) >> 2;
    for (int i = 0; i < a.kh; ++i)
        for (int j = 0; j < a.kw; ++j) {
            // operand position of this lane's two columns for tap (i, j)
            int xo[2];
            bool xv[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                int sy, sx;
                bool ok;
                if (!DGRAD) {
                    sy = py[nt] * a.sh + i * a.dh - a.pt;
                    sx = px[nt] * a.sw + j * a.dw - a.pl;
                    ok = sy >= 0 && sy < SH && sx >= 0 && sx < SW;
                } else {
                    const int ty = py[nt] + a.pt - i * a.dh, tx = px[nt] + a.pl - j * a.dw;
                    sy = ty / a.sh;
                    sx = tx / a.sw;
                    ok = ty >= 0 && tx >= 0 && sy * a.sh == ty && sx * a.sw == tx && sy < SH && sx < SW;
                }
                xv[nt] = ok;
                xo[nt] = ok ? kHalo + sy * SW + sx : kHalo;
            }
            const float* wt = a.wgt + (size_t)(i * a.kw + j) * a.cin * a.cout;
            for (int c4 = 0; c4 < c4n; ++c4) {
                const int kc = c4 * 4 + q;
                const bool kv = kc < KC;
                const int kcc = kv ? kc : 0;
                float af[MT], bf[2];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const float wv = DGRAD ? wt[(size_t)mrow[m] * a.cout + kcc] : wt[(size_t)kcc * a.cout + mrow[m]];
                    af[m] = (kv && mval[m]) ? wv : 0.f;
                }
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const float xvl = a.x[((size_t)pn[nt] * KC + kcc) * spp + xo[nt]];
                    bf[nt] = (kv && xv[nt]) ? xvl : 0.f;
                }
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m], bf[0], acc[m][0], 0, 0, 0);
                    acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m], bf[1], acc[m][1], 0, 0, 0);
                }
            }
        }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int p = pos0 + nt * 16 + r;
        if (p >= npos) continue;
        float* yb = a.y + (size_t)pn[nt] * M * dpp + kHalo + py[nt] * PW + px[nt];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int ch = (cot0 + m) * 16 + q * 4 + reg;
                if (ch >= M) continue;
                float v = acc[m][nt][reg];
                float* o = yb + (size_t)ch * dpp;
                if (DGRAD) {
                    o[0] += v;                              // gradient buffers accumulate (zeroed at the start of backward)
                } else {
                    if (a.bias) v += a.bias[ch];
                    if (a.relu) v = fmaxf(v, 0.f);
                    o[0] = v;
                }
            }
    }
}

template <bool DGRAD>
static int launch_conv2d_t(const Conv2dArgs& a, hipStream_t s) {
    const int M = DGRAD ? a.cin : a.cout;
    const int plane = DGRAD ? a.h * a.w : a.oh * a.ow;
    const int64_t npos = (int64_t)a.batch * plane;
    if (npos >= ((int64_t)1 << 31) - 256) { set_error("conv2d: %lld positions exceed the launch geometry", (long long)npos); return TCR_ERR_ARG; }
    const int tiles = ceil_div(M, 16);
    const int mt = tiles >= 3 ? 3 : tiles;
    const dim3 grid((unsigned)ceil_div64(npos, 128), ceil_div(tiles, mt));
    if (mt == 1) hipLaunchKernelGGL((conv2d_mfma_kernel<1, DGRAD>), grid, dim3(256), 0, s, a);
    else if (mt == 2) hipLaunchKernelGGL((conv2d_mfma_kernel<2, DGRAD>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((conv2d_mfma_kernel<3, DGRAD>), grid, dim3(256), 0, s, a);
    return check_launch("conv2d_mfma_kernel");
}

int launch_conv2d_fwd(const Conv2dArgs& a, hipStream_t s) { return launch_conv2d_t<false>(a, s); }
int launch_conv2d_dgrad(const Conv2dArgs& a, hipStream_t s) { return launch_conv2d_t<true>(a, s); }

// ---- conv2d filter gradient ---------------------------------------------------------------------------------------------
// partial[chunk][tap][ci][co] = sum over the chunk's utterances and all output positions of x[ci][in(pos, tap)] * dy[co][pos].
// One wave per (tap, 16 x 16 tile of [ci][co]); the MFMA k dimension holds 4 consecutive output positions.
__global__ __launch_bounds__(256) void conv2d_wgrad_kernel(const Conv2dArgs a, float* __restrict__ partial, const int utt_per_block,
                                                           const int cin_pad, const int cout_pad) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int tap = blockIdx.y;
    const int ti = tap / a.kw, tj = tap - ti * a.kw;
    const int ncot = cout_pad / 16;
    const int tile = blockIdx.z * 4 + wave;
    if (tile >= (cin_pad / 16) * ncot) return;
    const int cit = tile / ncot, cot = tile - cit * ncot;
    const int ci = cit * 16 + r, co = cot * 16 + r;
    const bool civ = ci < a.cin, cov = co < a.cout;
    const int cic = civ ? ci : 0, coc = cov ? co : 0;
    const int oplane = a.oh * a.ow;
    const float inv_ow = 1.0f / (float)a.ow;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int n0 = blockIdx.x * utt_per_block, n1 = min(n0 + utt_per_block, a.batch);
    for (int n = n0; n < n1; ++n) {
        const float* xr = a.x + ((size_t)n * a.cin + cic) * a.ppi + kHalo;
        const float* dr = a.dy + ((size_t)n * a.cout + coc) * a.ppo + kHalo;
        for (int p0 = 0; p0 < oplane; p0 += 4) {
            const int p = p0 + q;
            const bool pv = p < oplane;
            const int pc = pv ? p : 0;
            const int oy = oplane < (1 << 22) ? fast_div(pc, a.ow, inv_ow) : pc / a.ow, ox = pc - oy * a.ow;
            const int sy = oy * a.sh + ti * a.dh - a.pt, sx = ox * a.sw + tj * a.dw - a.pl;
            const bool sv = pv && sy >= 0 && sy < a.h && sx >= 0 && sx < a.w;
            const float xv = xr[sv ? sy * a.w + sx : 0];
            const float dv = dr[pc];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32((sv && civ) ? xv : 0.f, (pv && cov) ? dv : 0.f, acc, 0, 0, 0);
        }
    }
    float* dst = partial + (((size_t)blockIdx.x * a.kh * a.kw + tap) * cin_pad + cit * 16) * cout_pad + cot * 16;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) dst[(size_t)(q * 4 + reg) * cout_pad + r] = acc[reg];
}

int conv2d_wgrad_chunks(int batch) {
    int n = ceil_div(batch, 8);
    if (n > 64) n = 64;
    return n < 1 ? 1 : n;
}

size_t conv2d_wgrad_partial_floats(int kh, int kw, int cin, int cout, int batch) {
    return (size_t)conv2d_wgrad_chunks(batch) * kh * kw * (ceil_div(cin, 16) * 16) * (ceil_div(cout, 16) * 16);
}

int launch_conv2d_wgrad(const Conv2dArgs& a, float* dw, float* scratch, hipStream_t s) {
    const int cin_pad = ceil_div(a.cin, 16) * 16, cout_pad = ceil_div(a.cout, 16) * 16;
    const int upb = ceil_div(a.batch, conv2d_wgrad_chunks(a.batch));
    const int nchunk = ceil_div(a.batch, upb);
    const int tiles = (cin_pad / 16) * (cout_pad / 16);
    const int taps = a.kh * a.kw;
    if (taps > 65535) { set_error("conv2d wgrad: %d taps exceed the launch geometry", taps); return TCR_ERR_ARG; }
    hipLaunchKernelGGL(conv2d_wgrad_kernel, dim3(nchunk, taps, ceil_div(tiles, 4)), dim3(256), 0, s, a, scratch, upb, cin_pad, cout_pad);
    TCR_TRY(check_launch("conv2d_wgrad_kernel"));
    return launch_wgrad_reduce(scratch, dw, nchunk, taps, a.cin, a.cout, cin_pad, cout_pad, a.cout, 0, s);
}

// ---- per-channel sum over batch x plane (bias gradient), fixed order --------------------------------------------------------
__global__ __launch_bounds__(256) void chan_sum2d_kernel(const float* __restrict__ g, float* __restrict__ out, int batch, int c, int plane, int pp) {
    __shared__ double s_part[256];
    const int ch = blockIdx.x;
    double sacc = 0.0;
    const int64_t total = (int64_t)batch * plane;
    for (int64_t i = threadIdx.x; i < total; i += 256) {
        const int n = (int)(i / plane), p = (int)(i - (int64_t)n * plane);
        sacc += (double)g[((size_t)n * c + ch) * pp + kHalo + p];
    }
    s_part[threadIdx.x] = sacc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < 256; ++i) t += s_part[i];
        out[ch] = (float)t;
    }
}

int launch_chan_sum2d(const float* g, float* out, int batch, int c, int plane, int pp, hipStream_t s) {
    hipLaunchKernelGGL(chan_sum2d_kernel, dim3(c), dim3(256), 0, s, g, out, batch, c, plane, pp);
    return check_launch("chan_sum2d_kernel");
}

// ---- pooling ------------------------------------------------------------------------------------------------------------
// One thread per output element.  TF semantics: VALID windows lie inside the plane; SAME windows are clipped to it -- the
// average divides by the number of in-plane elements, the maximum ignores the padding.  max: argmax (first in scan order)
// is stored for the backward pass.
__global__ __launch_bounds__(256) void pool2d_fwd_kernel(const Pool2dArgs a) {
    const int64_t total = (int64_t)a.planes * a.oh * a.ow;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int op = (int)(idx % (a.oh * a.ow));
        const int64_t pl = idx / (a.oh * a.ow);
        const int oy = op / a.ow, ox = op - oy * a.ow;
        const float* src = a.x + (size_t)pl * a.ppi + kHalo;
        const int y0 = oy * a.sh - a.pt, x0 = ox * a.sw - a.pl;
        float best = -3.4e38f, sum = 0.f;
        int arg = -1, cnt = 0;
        for (int i = 0; i < a.kh; ++i) {
            const int y = y0 + i;
            if (y < 0 || y >= a.h) continue;
            for (int j = 0; j < a.kw; ++j) {
                const int x = x0 + j;
                if (x < 0 || x >= a.w) continue;
                const float v = src[y * a.w + x];
                sum += v;
                ++cnt;
                if (v > best) { best = v; arg = y * a.w + x; }
            }
        }
        a.y[(size_t)pl * a.ppo + kHalo + op] = a.is_max ? best : sum / (float)(cnt > 0 ? cnt : 1);
        if (a.is_max && a.argmax) a.argmax[(size_t)pl * a.oh * a.ow + op] = arg;
    }
}

// dx[in] += sum over the windows that contain `in` of dy[out] / count (avg) or dy[out] * [argmax(out) == in] (max)
__global__ __launch_bounds__(256) void pool2d_bwd_kernel(const Pool2dArgs a) {
    const int64_t total = (int64_t)a.planes * a.h * a.w;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int ip = (int)(idx % (a.h * a.w));
        const int64_t pl = idx / (a.h * a.w);
        const int y = ip / a.w, x = ip - y * a.w;
        const float* dy = a.dy + (size_t)pl * a.ppo + kHalo;
        float g = 0.f;
        // outputs oy with oy*sh - pt <= y <= oy*sh - pt + kh - 1
        const int oy_hi = min((y + a.pt) / a.sh, a.oh - 1), ox_hi = min((x + a.pl) / a.sw, a.ow - 1);
        for (int oy = oy_hi; oy >= 0 && oy * a.sh - a.pt + a.kh - 1 >= y; --oy)
            for (int ox = ox_hi; ox >= 0 && ox * a.sw - a.pl + a.kw - 1 >= x; --ox) {
                const int op = oy * a.ow + ox;
                if (a.is_max) {
                    if (a.argmax[(size_t)pl * a.oh * a.ow + op] == ip) g += dy[op];
                } else {
                    const int y0 = max(oy * a.sh - a.pt, 0), y1 = min(oy * a.sh - a.pt + a.kh, a.h);
                    const int x0 = max(ox * a.sw - a.pl, 0), x1 = min(ox * a.sw - a.pl + a.kw, a.w);
                    g += dy[op] / (float)((y1 - y0) * (x1 - x0));
                }
            }
        a.dx[(size_t)pl * a.ppi + kHalo + ip] += g;
    }
}

static int grid1d(int64_t n) {
    const int64_t b = ceil_div64(n, 256);
    return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

int launch_pool2d_fwd(const Pool2dArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(pool2d_fwd_kernel, dim3(grid1d((int64_t)a.planes * a.oh * a.ow)), dim3(256), 0, s, a);
    return check_launch("pool2d_fwd_kernel");
}

int launch_pool2d_bwd(const Pool2dArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(pool2d_bwd_kernel, dim3(grid1d((int64_t)a.planes * a.h * a.w)), dim3(256), 0, s, a);
    return check_launch("pool2d_bwd_kernel");
}

// ---- element-wise -------------------------------------------------------------------------------------------------------
// MODE 0: out = [relu](a + b)                    (residual add)
// MODE 1: g = [out > 0] ? g : 0  in place         (ReLU backward on a complete gradient buffer)
// MODE 2: dst += g                                (fan a gradient out to an input)
// MODE 3: out = x * mask / keep                   (tf.nn.dropout: div(x, keep_prob) * floor(keep_prob + uniform))
// MODE 4: dst += g * mask / keep
// Rows are [planes][pp]; only the interior (kHalo .. kHalo + plane) is touched.  The dropout mask of element e of sample
// n is a pure function of (seed, node, sample_offset + n, e): a sharded batch draws the same mask.
template <int MODE>
__global__ __launch_bounds__(256) void eltwise2d_kernel(const Elt2dArgs a) {
    const int64_t total = (int64_t)a.planes * a.plane;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t pl = idx / a.plane;
        const int p = (int)(idx - pl * a.plane);
        const size_t o = (size_t)pl * a.pp + kHalo + p;
        if (MODE == 0) {
            float v = a.a[o] + a.b[o];
            if (a.relu) v = fmaxf(v, 0.f);
            a.out[o] = v;
        } else if (MODE == 1) {
            if (!(a.a[o] > 0.f)) a.out[o] = 0.f;
        } else if (MODE == 2) {
            a.out[o] += a.a[o];
        } else {
            const int64_t n = pl / a.c;
            const uint64_t e = (uint64_t)(pl - n * a.c) * (uint64_t)a.plane + (uint64_t)p;
            const uint64_t index = ((uint64_t)(a.sample_offset + n) * (uint64_t)a.c * (uint64_t)a.plane + e);
            const float keep = uniform01(a.seed + 0x9e3779b97f4a7c15ULL * (uint64_t)(a.node + 1), index) < a.keep_prob ? 1.0f : 0.0f;
            if (MODE == 3) a.out[o] = (a.a[o] / a.keep_prob) * keep;
            else a.out[o] += (a.a[o] / a.keep_prob) * keep;
        }
    }
}

int launch_eltwise2d(int mode, const Elt2dArgs& a, hipStream_t s) {
    const dim3 grid(grid1d((int64_t)a.planes * a.plane));
    switch (mode) {
        case 0: hipLaunchKernelGGL((eltwise2d_kernel<0>), grid, dim3(256), 0, s, a); break;
        case 1: hipLaunchKernelGGL((eltwise2d_kernel<1>), grid, dim3(256), 0, s, a); break;
        case 2: hipLaunchKernelGGL((eltwise2d_kernel<2>), grid, dim3(256), 0, s, a); break;
        case 3: hipLaunchKernelGGL((eltwise2d_kernel<3>), grid, dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL((eltwise2d_kernel<4>), grid, dim3(256), 0, s, a); break;
    }
    return check_launch("eltwise2d_kernel");
}

// ---- SVDF pieces (audio_nets/kws.py:490-680): per-channel time filter over the whole plane, rank-group sum --------------------------------
// time filter: y[b][c] = sum_p x[b][c][p] * w[c][p]  (tf.matmul(activations_time [F, B, T], weights_time [F, T, 1]), kws.py:604-612)
__global__ __launch_bounds__(256) void tfilt_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y,
                                                        int64_t rows, int c, int plane, int ppi, int ppo) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows; i += (int64_t)gridDim.x * 256) {
        const int ch = (int)(i % c);
        const float* xr = x + i * ppi + kHalo;
        const float* wr = w + (size_t)ch * plane;
        float s = 0.f;
        for (int p = 0; p < plane; ++p) s = fmaf(xr[p], wr[p], s);
        y[i * ppo + kHalo] = s;
    }
}
// dx[b][c][p] += dy[b][c] * w[c][p]
__global__ __launch_bounds__(256) void tfilt_dx_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx,
                                                       int64_t rows, int c, int plane, int ppi, int ppo) {
    const int64_t total = rows * plane;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / plane;
        const int p = (int)(i - r * plane);
        dx[r * ppi + kHalo + p] += dy[r * ppo + kHalo] * w[(size_t)(r % c) * plane + p];
    }
}
// dw[c][p] = sum_b dy[b][c] * x[b][c][p]  (fixed order over the batch: reproducible)
__global__ __launch_bounds__(256) void tfilt_dw_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dw,
                                                       int batch, int c, int plane, int ppi, int ppo) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= c * plane) return;
    const int ch = i / plane, p = i - ch * plane;
    float s = 0.f;
    for (int b = 0; b < batch; ++b) {
        const int64_t r = (int64_t)b * c + ch;
        s = fmaf(dy[r * ppo + kHalo], x[r * ppi + kHalo + p], s);
    }
    dw[i] = s;
}
// group sum: y[b][u] = [relu](sum_r x[b][u * g + r] + bias[u])  (reshape [units, rank, batch] + reduce_sum + bias_add + relu, kws.py:613-628)
__global__ __launch_bounds__(256) void gsum_fwd_kernel(const float* __restrict__ x, const float* __restrict__ bias, float* __restrict__ y,
                                                       int64_t rows, int units, int group, int pp, int relu) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / units;
        const int u = (int)(i - b * units);
        const float* xr = x + ((int64_t)b * units * group + (int64_t)u * group) * pp + kHalo;
        float s = 0.f;
        for (int r = 0; r < group; ++r) s += xr[(size_t)r * pp];
        if (bias) s += bias[u];
        y[i * pp + kHalo] = relu ? fmaxf(s, 0.f) : s;
    }
}
// dx[b][u * g + r] += dy[b][u]
__global__ __launch_bounds__(256) void gsum_dx_kernel(const float* __restrict__ dy, float* __restrict__ dx, int64_t rows_in, int group, int pp) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows_in; i += (int64_t)gridDim.x * 256)
        dx[i * pp + kHalo] += dy[(i / group) * pp + kHalo];
}

int launch_tfilt_fwd(const float* x, const float* w, float* y, int batch, int c, int plane, hipStream_t s) {
    hipLaunchKernelGGL(tfilt_fwd_kernel, dim3(grid1d((int64_t)batch * c)), dim3(256), 0, s, x, w, y, (int64_t)batch * c, c, plane,
                       plane + 2 * kHalo, 1 + 2 * kHalo);
    return check_launch("tfilt_fwd_kernel");
}
int launch_tfilt_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, int batch, int c, int plane, hipStream_t s) {
    if (dx) hipLaunchKernelGGL(tfilt_dx_kernel, dim3(grid1d((int64_t)batch * c * plane)), dim3(256), 0, s, dy, w, dx, (int64_t)batch * c, c, plane,
                               plane + 2 * kHalo, 1 + 2 * kHalo);
    hipLaunchKernelGGL(tfilt_dw_kernel, dim3(ceil_div(c * plane, 256)), dim3(256), 0, s, dy, x, dw, batch, c, plane, plane + 2 * kHalo, 1 + 2 * kHalo);
    return check_launch("tfilt_bwd_kernel");
}
int launch_gsum_fwd(const float* x, const float* bias, float* y, int batch, int units, int group, int relu, hipStream_t s) {
    hipLaunchKernelGGL(gsum_fwd_kernel, dim3(grid1d((int64_t)batch * units)), dim3(256), 0, s, x, bias, y, (int64_t)batch * units, units, group,
                       1 + 2 * kHalo, relu);
    return check_launch("gsum_fwd_kernel");
}
int launch_gsum_dx(const float* dy, float* dx, int batch, int units, int group, hipStream_t s) {
    hipLaunchKernelGGL(gsum_dx_kernel, dim3(grid1d((int64_t)batch * units * group)), dim3(256), 0, s, dy, dx, (int64_t)batch * units * group, group,
                       1 + 2 * kHalo);
    return check_launch("gsum_dx_kernel");
}

// ---- head: logits rows -> softmax / cross-entropy / dlogits -----------------------------------------------------------------
// z [B][NC][pp] with one interior element per row (the logits node is 1 x 1 spatially).
__global__ __launch_bounds__(64) void head2d_kernel(const float* __restrict__ z, const float* __restrict__ labels, float* __restrict__ logits,
                                                    float* __restrict__ probs, float* __restrict__ dz, float* __restrict__ loss_utt,
                                                    int batch, int nc, int pp, float inv_global_batch, float label_smoothing) {
    const int n = blockIdx.x * 64 + threadIdx.x;
    if (n >= batch) return;
    const float* zr = z + (size_t)n * nc * pp + kHalo;
    float mx = zr[0];
    for (int k = 1; k < nc; ++k) mx = fmaxf(mx, zr[(size_t)k * pp]);
    float se = 0.f;
    for (int k = 0; k < nc; ++k) se += expf(zr[(size_t)k * pp] - mx);
    const float lse = logf(se);
    float loss = 0.f;
    for (int k = 0; k < nc; ++k) {
        const float v = zr[(size_t)k * pp];
        const float p = expf(v - mx) / se;
        logits[(size_t)n * nc + k] = v;
        probs[(size_t)n * nc + k] = p;
        if (labels) {
            float y = labels[(size_t)n * nc + k];
            if (label_smoothing > 0.f) y = y * (1.0f - label_smoothing) + label_smoothing / (float)nc;
            loss -= y * ((v - mx) - lse);
            if (dz) dz[((size_t)n * nc + k) * pp + kHalo] = (p - y) * inv_global_batch;
        }
    }
    if (loss_utt) loss_utt[n] = loss;
}

int launch_head2d(const float* z, const float* labels, float* logits, float* probs, float* dz, float* loss_utt, int batch, int nc, int pp,
                  float inv_global_batch, float label_smoothing, hipStream_t s) {
    hipLaunchKernelGGL(head2d_kernel, dim3(ceil_div(batch, 64)), dim3(64), 0, s, z, labels, logits, probs, dz, loss_utt, batch, nc, pp,
                       inv_global_batch, label_smoothing);
    return check_launch("head2d_kernel");
}

// ---- input re-layout: front-end planar features [B][F][Tp] -> one [T x F] plane per utterance ----------------------------------
__global__ __launch_bounds__(256) void features_to_plane_kernel(const float* __restrict__ feat, float* __restrict__ out, int batch, int t, int f,
                                                                int tp, int pp) {
    const int64_t total = (int64_t)batch * pp;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int n = (int)(idx / pp), o = (int)(idx - (int64_t)n * pp) - kHalo;
        float v = 0.f;
        if (o >= 0 && o < t * f) {
            const int tt = o / f, ff = o - tt * f;
            v = feat[((size_t)n * f + ff) * tp + kHalo + tt];
        }
        out[idx] = v;
    }
}

int launch_features_to_plane(const float* feat, float* out, int batch, int t, int f, hipStream_t s) {
    const int tp = t + 2 * kHalo, pp = t * f + 2 * kHalo;
    hipLaunchKernelGGL(features_to_plane_kernel, dim3(grid1d((int64_t)batch * pp)), dim3(256), 0, s, feat, out, batch, t, f, tp, pp);
    return check_launch("features_to_plane_kernel");
}

}  // namespace tcr
