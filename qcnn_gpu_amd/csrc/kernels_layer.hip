// Layer-level gfx950 kernels: one kernel per member function of the reference's layer API, on the
// reference's own buffer layouts.  They back the qcnn_op_* entry points and the layered (tap)
// mode; the production pass is the pair of fused kernels in kernels_fused.hip.
#include <algorithm>
#include "qcnn_internal.h"
#include <stdlib.h>

namespace qcnn {

Requant make_requant(int blu, int mul, int shift)
{
    Requant r;
    r.blu = blu;
    r.mul = mul;
    r.shift = shift;
    r.bias = (mul != 0 && shift >= 1) ? (int)((1u << (shift - 1)) / (unsigned)mul) : 0;
    r.cmul = (int)((unsigned)r.bias * (unsigned)mul);
    r.bluf = (float)blu;
    const int k = 24 - shift;
    r.mulk = k >= 0 ? (int)((unsigned)mul << k) : 0;
    return r;
}

// The arithmetic of CHW2CHW_VECT_C_QUANT_BLU (inference/mat.cu:286-291) on one value.
__device__ __forceinline__ int requant_ref(float t, const Requant &q)
{
    if (t > q.bluf) return 127;
    if (t < 0.f) return 0;
    return (int)((unsigned)((int)t + q.bias) * (unsigned)q.mul) >> q.shift;
}

// InputLayer::ppro / HW2HW_VECT_C_PPRO (inference/cnn.cu:445-468).  One vec4 per thread; lanes
// 1..3 are written as zero (the reference leaves them uninitialised and relies on zero weights).
__global__ __launch_bounds__(256) void k_ppro(const uint8_t *__restrict__ x, uint32_t *__restrict__ x_ppro,
                                              size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        x_ppro[i] = (uint32_t)(uint8_t)(x[i] - 128);
}

// CovLayer::ConvForward_static (inference/cnn.cu:145-160) as a direct convolution on the
// NCHW_VECT_C layout: each thread owns one output pixel and KB output channels, one v_dot4 per
// (vec4 of input channels, tap, output channel).  u = float(acc) + b; with relu != 0 negative results
// are stored as 0 (CovLayer::activate, inference/cnn.cu:161-166, which the dynamic pass runs on u).
template <int KS, int KB>
__global__ __launch_bounds__(256) void k_conv_static(const int *__restrict__ x, const int *__restrict__ w,
                                                     const float *__restrict__ b, float *__restrict__ u, int H,
                                                     int W, int C4, int K, int relu)
{
    constexpr int P = (KS - 1) / 2;
    const int xo = blockIdx.x * 64 + (threadIdx.x & 63);
    const int yo = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int k0 = blockIdx.z * KB;
    if (xo >= W || yo >= H) return;
    int acc[KB];
#pragma unroll
    for (int i = 0; i < KB; ++i) acc[i] = 0;
    for (int c4 = 0; c4 < C4; ++c4) {
#pragma unroll
        for (int r = 0; r < KS; ++r) {
            const int yy = yo + r - P;
            if (yy < 0 || yy >= H) continue;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int xx = xo + s - P;
                if (xx < 0 || xx >= W) continue;
                const int v = x[((size_t)c4 * H + yy) * W + xx];
#pragma unroll
                for (int i = 0; i < KB; ++i)
                    acc[i] = __builtin_amdgcn_sdot4(v, w[(((size_t)(k0 + i) * C4 + c4) * KS + r) * KS + s], acc[i],
                                                    false);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < KB; ++i) {
        float v = (float)acc[i] + b[k0 + i];
        if (relu && v < 0.f) v = 0.f;
        u[((size_t)(k0 + i) * H + yo) * W + xo] = v;
    }
}

template <int KS>
static hipError_t conv_dispatch(const void *x, const void *w, const void *b, void *u, int H, int W, int cin,
                                int cout, int relu, hipStream_t s)
{
    const int C4 = (cin + 3) / 4;
    dim3 block(256);
    if (cout % 8 == 0) {
        dim3 grid((W + 63) / 64, (H + 3) / 4, cout / 8);
        hipLaunchKernelGGL((k_conv_static<KS, 8>), grid, block, 0, s, (const int *)x, (const int *)w,
                           (const float *)b, (float *)u, H, W, C4, cout, relu);
    } else {
        dim3 grid((W + 63) / 64, (H + 3) / 4, cout);
        hipLaunchKernelGGL((k_conv_static<KS, 1>), grid, block, 0, s, (const int *)x, (const int *)w,
                           (const float *)b, (float *)u, H, W, C4, cout, relu);
    }
    return hipGetLastError();
}

// QCNN_CONV_DIRECT=1 keeps every layer-level convolution on the direct v_dot4 kernel (A/B timing, tests).
bool conv_direct_only()
{
    static const bool v = [] {
        const char *e = getenv("QCNN_CONV_DIRECT");
        return e && atoi(e) != 0;
    }();
    return v;
}

hipError_t launch_conv_static(const void *x, const void *w, const void *b, void *u, int H, int W, int cin,
                              int cout, int ks, int relu, hipStream_t s)
{
    if (!conv_direct_only() && conv_mfma_supported(cin, cout, ks))
        return launch_conv_mfma(x, w, b, u, nullptr, nullptr, nullptr, 1, relu, H, W, cin, cout, ks, s);
    switch (ks) {
    case 1: return conv_dispatch<1>(x, w, b, u, H, W, cin, cout, relu, s);
    case 3: return conv_dispatch<3>(x, w, b, u, H, W, cin, cout, relu, s);
    case 5: return conv_dispatch<5>(x, w, b, u, H, W, cin, cout, relu, s);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_ppro(const void *x, void *x_ppro, size_t n, hipStream_t s)
{
    const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_ppro, dim3(grid), dim3(256), 0, s, (const uint8_t *)x, (uint32_t *)x_ppro, n);
    return hipGetLastError();
}

// CHW2CHW_VECT_C_QUANT_BLU (inference/mat.cu:262-303): one thread per (pixel, group of 4
// channels); a full group is written as one coalesced dword, a ragged tail group byte by byte
// (only the lanes the reference writes).
__global__ __launch_bounds__(256) void k_quant_blu(const float *__restrict__ u, int8_t *__restrict__ out, size_t hw,
                                                   int C, Requant q)
{
    const size_t groups = (size_t)(C + 3) / 4;
    const size_t total = hw * groups;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const size_t grp = idx / hw, p = idx - grp * hw;
        const int c0 = (int)grp * 4;
        if (c0 + 4 <= C) {
            uint32_t pk = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                pk |= (uint32_t)(uint8_t)requant_ref(u[(size_t)(c0 + i) * hw + p], q) << (8 * i);
            reinterpret_cast<uint32_t *>(out)[grp * hw + p] = pk;
        } else {
            for (int i = 0; c0 + i < C; ++i)
                out[(grp * hw + p) * 4 + i] = (int8_t)requant_ref(u[(size_t)(c0 + i) * hw + p], q);
        }
    }
}

hipError_t launch_quant_blu(const void *u, void *out, size_t hw, int channels, const Requant &rq, hipStream_t s)
{
    const size_t total = hw * (size_t)((channels + 3) / 4);
    if (total == 0) return hipSuccess;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(k_quant_blu, dim3(grid), dim3(256), 0, s, (const float *)u, (int8_t *)out, hw, channels, rq);
    return hipGetLastError();
}

// applyRes_GPU_y (inference/cnn.cu:507-523).
__device__ __forceinline__ uint8_t apply_res_one(uint8_t x, float res, int mul, int shift)
{
    int temp = (int)res;
    temp = (int)((unsigned)temp * (unsigned)mul + (1u << (shift - 1))) >> shift;
    const short rec = (short)((int)x + temp);
    return rec > 255 ? (uint8_t)255 : rec < 0 ? (uint8_t)0 : (uint8_t)rec;
}

__global__ __launch_bounds__(256) void k_apply_res_y(const uint8_t *__restrict__ x, uint8_t *__restrict__ rec,
                                                     const float *__restrict__ res, int mul, int shift, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        rec[i] = apply_res_one(x[i], res[i], mul, shift);
}

hipError_t launch_apply_res_y(const void *x, void *rec, const void *res, int mul, int shift, size_t n,
                              hipStream_t s)
{
    if (n == 0) return hipSuccess;
    const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_apply_res_y, dim3(grid), dim3(256), 0, s, (const uint8_t *)x, (uint8_t *)rec,
                       (const float *)res, mul, shift, n);
    return hipGetLastError();
}

// Integer sum of squared differences (the numerator of vrcnn_data::psnr, inference/yuv_data.cpp:92-93).
__global__ __launch_bounds__(256) void k_sse(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b, size_t n,
                                             unsigned long long *sum)
{
    unsigned long long s = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int d = (int)a[i] - (int)b[i];
        s += (unsigned)(d * d);
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    __shared__ unsigned long long part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(sum, part[0] + part[1] + part[2] + part[3]);
}

hipError_t launch_sse(const void *a, const void *b, size_t n, unsigned long long *dev_sum, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(dev_sum, 0, sizeof(unsigned long long), s);
    if (e != hipSuccess || n == 0) return e;
    const int grid = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(k_sse, dim3(grid), dim3(256), 0, s, (const uint8_t *)a, (const uint8_t *)b, n, dev_sum);
    return hipGetLastError();
}

// The same without clearing the sum first (the caller cleared it; several launches may add to it).
hipError_t launch_sse_acc(const void *a, const void *b, size_t n, unsigned long long *dev_sum, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    const int grid = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(k_sse, dim3(grid), dim3(256), 0, s, (const uint8_t *)a, (const uint8_t *)b, n, dev_sum);
    return hipGetLastError();
}

// Both sums of a clip chunk in one launch: frame f = blockIdx.y; sum_before[f] += |in_f - ori_f|^2 and
// sum_after[f] += |out_f - ori_f|^2 (either sum array may be null).  Frames are n bytes apart in all three.
__global__ __launch_bounds__(256) void k_sse_clip(const uint8_t *__restrict__ in, const uint8_t *__restrict__ out,
                                                  const uint8_t *__restrict__ ori, size_t n, unsigned long long *sum_before,
                                                  unsigned long long *sum_after)
{
    const size_t f = blockIdx.y;
    const uint8_t *a = in + f * n, *c = out + f * n, *o = ori + f * n;
    unsigned long long sb = 0, sa = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int ov = (int)o[i];
        const int d0 = (int)a[i] - ov, d1 = (int)c[i] - ov;
        sb += (unsigned)(d0 * d0);
        sa += (unsigned)(d1 * d1);
    }
    for (int off = 32; off > 0; off >>= 1) {
        sb += __shfl_down(sb, off, 64);
        sa += __shfl_down(sa, off, 64);
    }
    __shared__ unsigned long long part[2][4];
    if ((threadIdx.x & 63) == 0) {
        part[0][threadIdx.x >> 6] = sb;
        part[1][threadIdx.x >> 6] = sa;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (sum_before) atomicAdd(sum_before + f, part[0][0] + part[0][1] + part[0][2] + part[0][3]);
        if (sum_after) atomicAdd(sum_after + f, part[1][0] + part[1][1] + part[1][2] + part[1][3]);
    }
}

hipError_t launch_sse_clip(const void *in, const void *out, const void *ori, size_t n, int nframes,
                           unsigned long long *sum_before, unsigned long long *sum_after, hipStream_t s)
{
    if (n == 0 || nframes < 1) return hipSuccess;
    const int gx = (int)((n + 255) / 256 < 128 ? (n + 255) / 256 : 128);
    hipLaunchKernelGGL(k_sse_clip, dim3(gx, nframes), dim3(256), 0, s, (const uint8_t *)in, (const uint8_t *)out,
                       (const uint8_t *)ori, n, sum_before, sum_after);
    return hipGetLastError();
}

// [H][W][48] internal channel order -> Conc1.conc layout [12][H][W][4] (test/tap helper).
__global__ __launch_bounds__(256) void k_conc_to_ref(const int8_t *__restrict__ in, int8_t *__restrict__ out, int H,
                                                     int W)
{
    const size_t hw = (size_t)H * W, total = hw * 48;
    const int wp = conc_wp(W);
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const size_t p = idx / 48;
        const int pos = (int)(idx - p * 48);
        const int ch = refch_of_pos(pos);
        const size_t y = p / W, x = p - y * W;
        out[((size_t)(ch >> 2) * hw + p) * 4 + (ch & 3)] = in[((y + 2) * wp + x + 2) * 48 + pos];
    }
}

// conc_internal: the bordered [H+4+..][W+4][48] image of the fused path; conc_ref: Conc1.conc, [12][H][W][4]
hipError_t launch_conc_to_ref(const void *conc_internal, void *conc_ref, int H, int W, hipStream_t s)
{
    if ((size_t)H * W == 0) return hipSuccess;
    hipLaunchKernelGGL(k_conc_to_ref, dim3(4096), dim3(256), 0, s, (const int8_t *)conc_internal, (int8_t *)conc_ref, H, W);
    return hipGetLastError();
}

// Zeroes the border of `nframes` conc1 images (rows 0, 1, H+2, H+3 and, between them, the four border pixels that
// end one row and start the next): what kernel A leaves unwritten and kernel B reads as the zero padding of C3.
__global__ __launch_bounds__(256) void k_conc_zero_border(int8_t *conc, int H, int W, size_t frame_bytes)
{
    typedef int v4 __attribute__((ext_vector_type(4)));
    const int wp = conc_wp(W);
    int8_t *img = conc + (size_t)blockIdx.y * frame_bytes;
    const size_t row16 = (size_t)wp * 3;                // 16-byte pieces per row
    const size_t full = 2 * row16 + 6;                  // rows 0, 1 and the two border pixels that start row 2
    const size_t seams = (size_t)(H - 1) * 12;          // 4 pixels = 12 pieces after each of the rows 2 .. H
    const size_t tail = 6 + 2 * row16;                  // the two pixels that end row H+1, then rows H+2, H+3
    const size_t total = full + seams + tail;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        size_t piece;
        if (i < full) piece = i;
        else if (i < full + seams) {
            const size_t k = (i - full) / 12, r = (i - full) - k * 12;
            piece = (2 + k) * row16 + (size_t)(W + 2) * 3 + r;
        } else
            piece = (size_t)(H + 1) * row16 + (size_t)(W + 2) * 3 + (i - full - seams);
        reinterpret_cast<v4 *>(img)[piece] = v4{0, 0, 0, 0};
    }
}

hipError_t launch_conc_zero_border(void *conc_internal, int nframes, int H, int W, hipStream_t s)
{
    if (nframes < 1) return hipSuccess;
    const size_t total = 4 * (size_t)conc_wp(W) * 3 + 12 + (size_t)(H - 1) * 12;
    const unsigned gx = (unsigned)std::min<size_t>((total + 255) / 256, 1024);
    hipLaunchKernelGGL(k_conc_zero_border, dim3(gx, nframes), dim3(256), 0, s, (int8_t *)conc_internal, H, W, conc_bytes(H, W));
    return hipGetLastError();
}

}  // namespace qcnn
