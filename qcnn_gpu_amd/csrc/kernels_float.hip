// The float32 configuration of the reference (FLOAT_CONFIG, inference/mat.cuh:89-101): activations and
// weights float32 NCHW, the network of qvrcnn::forward at inference/qvrcnn.cu:244-285.  Kernels:
//   k_ppro_float      FLOAT_PPRO, inference/cnn.cu:470-484            x_ppro = ((float)x - 128) / 255
//   k_conv_f32        CovLayer::ConvForward, inference/cnn.cu:129-144 (cudnnConvolutionForward +
//                     cudnnAddTensor of b) with CovLayer::activate (inference/cnn.cu:161-166) on request
//   k_apply_res_float applyRes_GPU, float arm, inference/cnn.cu:487-502
//
// The convolution is an implicit GEMM D[channel][pixel] = W[channel][k] * X[k][pixel] on the f32-input
// matrix instruction v_mfma_f32_16x16x4_f32 (exact float32: every product is rounded once and added to
// the accumulator in k order, no wider intermediate).  A 256-thread workgroup owns an 8 x 32 output
// tile and walks the input channels in chunks of 16: the chunk's planes (with halo, zero outside the
// frame) and the chunk's weights, read in file order w[K][C][R][S] (contiguous per output channel) and
// scattered into lane-ordered A fragments, are staged in LDS; a wave owns two tile rows (four 16-pixel strips) and every output
// channel.  The k index of one instruction is four input channels at one tap, so that a B fragment is
// one dword per lane read from four planes, 16 consecutive pixels each (the plane stride is padded to
// 16 mod 32 dwords: 64 distinct banks).  The single-channel first layer packs four taps into one
// instruction instead (25 taps = 7 steps, zero weights on the rest).
#include "qcnn_internal.h"

namespace qcnn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct FConvArgs {
    const float *x;  // float NCHW [cin][H][W]
    const float *w;  // float [cout][cin][ks][ks]
    const float *b;  // float [cout]
    float *u;        // float NCHW [cout][H][W]
    int relu;
    int H, W, cout;
    int tiles_x, ntiles;
};

namespace cf {
constexpr int TH = 8, TW = 32;
}

template <int KS, int CIN>
struct FGeom {
    static constexpr int P = (KS - 1) / 2;
    static constexpr int RH = cf::TH + KS - 1, RW = cf::TW + KS - 1;
    static constexpr int CC = CIN == 1 ? 1 : 16;                         // input channels per chunk
    static constexpr int NCHUNK = CIN / CC;
    static constexpr int PS = ((RH * RW + 15) / 32) * 32 + 16;           // plane stride, dwords: 16 mod 32
    static constexpr int STEPS = CIN == 1 ? (KS * KS + 3) / 4 : KS * KS * (CC / 4);  // MFMA k-steps per chunk
    static_assert(CIN == 1 || CIN % 16 == 0, "input channels come in chunks of 16");
    static_assert(PS >= RH * RW && PS % 32 == 16, "plane stride");
};

// Workgroup -> tile map.  Workgroups are dealt to the 8 XCDs round robin (blockIdx.x & 7) and each XCD has an
// L2 of its own: XCD x takes the x-th eighth of the tiles (row-major), so the halo that neighbouring tiles
// share is fetched from HBM once per XCD band, not once per tile.  The grid is 8 * ceil(ntiles / 8)
// workgroups; returns -1 for the few without a tile.
__device__ __forceinline__ int xcd_tile(int ntiles)
{
    const int per = (ntiles + 7) >> 3;
    const int t = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    return ((int)(blockIdx.x >> 3) < per && t < ntiles) ? t : -1;
}

template <int KS, int CIN, int MT>
__global__ __launch_bounds__(256) void k_conv_f32(const FConvArgs a)
{
    using G = FGeom<KS, CIN>;
    constexpr int TH = cf::TH, TW = cf::TW, RH = G::RH, RW = G::RW, PS = G::PS, CC = G::CC, STEPS = G::STEPS;
    __shared__ float sX[CC * PS];
    __shared__ float sA[STEPS * MT * 65];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = a.H, W = a.W;
    const size_t HW = (size_t)H * W;
    const int tile = xcd_tile(a.ntiles);
    if (tile < 0) return;
    const int ty0 = (tile / a.tiles_x) * TH, tx0 = (tile % a.tiles_x) * TW;
    const int kq = lane >> 4, px = lane & 15;

    f32x4 acc[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // strip t of this wave: tile row 2*wave + (t >> 1), columns 16*(t & 1) .. +15
    const int xlane = (2 * wave) * RW + px + (CIN == 1 ? 0 : kq * PS);

    constexpr int FS = 65;  // fragment stride in dwords: consecutive fragments start one bank apart
    for (int chunk = 0; chunk < G::NCHUNK; ++chunk) {
        const int c0 = chunk * CC;
        // All global loads of the chunk are issued before the first LDS store, so their latencies overlap.
        // input planes of the chunk with halo; cudnn pads with zeros (inference/cnn.cu:44-49)
        constexpr int NX = (CC * RH * RW + 255) / 256;
        float xv[NX];
#pragma unroll
        for (int it = 0; it < NX; ++it) {
            const int idx = it * 256 + tid;
            const int c = idx / (RH * RW), rem = idx % (RH * RW);
            const int ry = rem / RW, rx = rem % RW;
            const int gy = ty0 - G::P + ry, gx = tx0 - G::P + rx;
            xv[it] = 0.f;
            if (idx < CC * RH * RW && gy >= 0 && gy < H && gx >= 0 && gx < W)
                xv[it] = a.x[(size_t)(c0 + c) * HW + (size_t)gy * W + gx];
        }
        // Weights of the chunk: for every output channel the CC * KS * KS floats w[ch][c0 ..][.][.] are
        // contiguous in the file order, so consecutive threads read consecutive addresses; each value goes to
        // lane (ch & 15) + 16 k of A fragment (step, m): lane l holds W[channel 16m + (l & 15)][k = l >> 4].
        constexpr int PER_CH = CIN == 1 ? KS * KS : CC * KS * KS;
        constexpr int NW = (16 * MT * PER_CH + 255) / 256;
        float wv[NW];
#pragma unroll
        for (int it = 0; it < NW; ++it) {
            const int idx = it * 256 + tid;
            const int ch = idx / PER_CH, j = idx % PER_CH;
            wv[it] = 0.f;
            if (idx < 16 * MT * PER_CH && ch < a.cout) wv[it] = a.w[((size_t)ch * CIN + c0) * KS * KS + j];
        }
        if (chunk) __syncthreads();  // every wave is done with the previous chunk's tiles
#pragma unroll
        for (int it = 0; it < NX; ++it) {
            const int idx = it * 256 + tid;
            const int c = idx / (RH * RW), rem = idx % (RH * RW);
            if (idx < CC * RH * RW) sX[c * PS + rem] = xv[it];
        }
#pragma unroll
        for (int it = 0; it < NW; ++it) {
            const int idx = it * 256 + tid;
            const int ch = idx / PER_CH, j = idx % PER_CH;
            if (idx < 16 * MT * PER_CH) {
                int step, k;
                if constexpr (CIN == 1) {
                    step = j >> 2, k = j & 3;  // tap j = 4 step + k
                } else {
                    const int c = j / (KS * KS), rs = j % (KS * KS);
                    step = (c >> 2) * (KS * KS) + rs, k = c & 3;
                }
                sA[(step * MT + (ch >> 4)) * FS + (ch & 15) + 16 * k] = wv[it];
            }
        }
        if constexpr (CIN == 1) {
            // taps 25 .. 27 of the last step: zero weights
            if (tid < 16 * MT * (4 * STEPS - KS * KS)) {
                const int ch = tid / (4 * STEPS - KS * KS), tap = KS * KS + tid % (4 * STEPS - KS * KS);
                sA[((tap >> 2) * MT + (ch >> 4)) * FS + (ch & 15) + 16 * (tap & 3)] = 0.f;
            }
        }
        __syncthreads();

        if constexpr (CIN == 1) {
#pragma unroll
            for (int step = 0; step < STEPS; ++step) {
                int tap = 4 * step + kq;
                if (tap >= KS * KS) tap = 0;  // zero weight: any valid address
                const int off = xlane + (tap / KS) * RW + tap % KS;
                float af[MT], bf[4];
#pragma unroll
                for (int m = 0; m < MT; ++m) af[m] = sA[(step * MT + m) * FS + lane];
#pragma unroll
                for (int t = 0; t < 4; ++t) bf[t] = sX[off + (t >> 1) * RW + 16 * (t & 1)];
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m], bf[t], acc[m][t], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int r = 0; r < KS; ++r)
#pragma unroll
                for (int s = 0; s < KS; ++s)
#pragma unroll
                    for (int cg = 0; cg < CC / 4; ++cg) {
                        const int step = cg * (KS * KS) + r * KS + s;
                        const int off = xlane + 4 * cg * PS + r * RW + s;
                        float af[MT], bf[4];
#pragma unroll
                        for (int m = 0; m < MT; ++m) af[m] = sA[(step * MT + m) * FS + lane];
#pragma unroll
                        for (int t = 0; t < 4; ++t) bf[t] = sX[off + (t >> 1) * RW + 16 * (t & 1)];
#pragma unroll
                        for (int m = 0; m < MT; ++m)
#pragma unroll
                            for (int t = 0; t < 4; ++t)
                                acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m], bf[t], acc[m][t], 0, 0, 0);
                    }
        }
    }

    // D: lane l holds pixel l & 15, channels 4*(l >> 4) + v of the 16-channel tile
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int gy = ty0 + 2 * wave + (t >> 1), gx = tx0 + 16 * (t & 1) + px;
            if (gy >= H || gx >= W) continue;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int ch = 16 * m + 4 * kq + v;
                if (ch < a.cout) {
                    float val = acc[m][t][v] + a.b[ch];  // cudnnAddTensor, inference/cnn.cu:141
                    if (a.relu) val = val > 0.f ? val : (val < 0.f ? 0.f : val);  // NaN propagates (cnn.cu:57)
                    a.u[(size_t)ch * HW + (size_t)gy * W + gx] = val;
                }
            }
        }
}

// The single-output-channel layer (C4): a matrix instruction would use one of its 16 rows, so this is a
// plain multiply-add kernel.  One output pixel per thread on the same 8 x 32 tile and 16-channel chunks;
// the weight addresses are uniform over the wave, so they travel through the scalar unit.
template <int KS, int CIN>
__global__ __launch_bounds__(256) void k_conv_f32_one(const FConvArgs a)
{
    using G = FGeom<KS, CIN>;
    constexpr int RH = G::RH, RW = G::RW, PS = G::PS, CC = G::CC;
    __shared__ float sX[CC * PS];
    const int tid = threadIdx.x;
    const int H = a.H, W = a.W;
    const size_t HW = (size_t)H * W;
    const int tile = xcd_tile(a.ntiles);
    if (tile < 0) return;
    const int ty0 = (tile / a.tiles_x) * cf::TH, tx0 = (tile % a.tiles_x) * cf::TW;
    const int py = tid >> 5, px = tid & 31;
    const float *__restrict__ w = a.w;
    float acc = 0.f;
    for (int chunk = 0; chunk < G::NCHUNK; ++chunk) {
        const int c0 = chunk * CC;
        constexpr int NX = (CC * RH * RW + 255) / 256;
        float xv[NX];
#pragma unroll
        for (int it = 0; it < NX; ++it) {
            const int idx = it * 256 + tid;
            const int c = idx / (RH * RW), rem = idx % (RH * RW);
            const int gy = ty0 - G::P + rem / RW, gx = tx0 - G::P + rem % RW;
            xv[it] = 0.f;
            if (idx < CC * RH * RW && gy >= 0 && gy < H && gx >= 0 && gx < W)
                xv[it] = a.x[(size_t)(c0 + c) * HW + (size_t)gy * W + gx];
        }
        if (chunk) __syncthreads();
#pragma unroll
        for (int it = 0; it < NX; ++it) {
            const int idx = it * 256 + tid;
            if (idx < CC * RH * RW) sX[(idx / (RH * RW)) * PS + idx % (RH * RW)] = xv[it];
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < CC; ++c)
#pragma unroll
            for (int r = 0; r < KS; ++r)
#pragma unroll
                for (int s = 0; s < KS; ++s)
                    acc = __builtin_fmaf(w[((c0 + c) * KS + r) * KS + s], sX[c * PS + (py + r) * RW + px + s], acc);
    }
    const int gy = ty0 + py, gx = tx0 + px;
    if (gy < H && gx < W) {
        float val = acc + a.b[0];
        if (a.relu) val = val > 0.f ? val : (val < 0.f ? 0.f : val);
        a.u[(size_t)gy * W + gx] = val;
    }
}

__global__ void k_ppro_float(const uint8_t *x, float *xf, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        xf[i] = __fdiv_rn((float)x[i] - 128.f, 255.f);
}

// rec_int = x + res * 255 + 0.5 (inference/cnn.cu:496): the product and the first sum are float32 (the
// device compiler of the reference contracts them into one fused multiply-add by default), the 0.5 is a
// double constant, the conversion to short truncates; then the clamp of :498-500.
__global__ void k_apply_res_float(const uint8_t *x, const float *res, uint8_t *rec, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double t = (double)__fmaf_rn(res[i], 255.f, (float)x[i]) + 0.5;
        int r;
        if (!(t > -32768.0)) r = -32768;  // also NaN
        else if (t > 32767.0) r = 32767;
        else r = (int)t;
        rec[i] = (uint8_t)(r > 255 ? 255 : r < 0 ? 0 : r);
    }
}

static inline int grid_for(size_t n)
{
    const size_t g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : g > 4096 ? 4096 : g);
}

hipError_t launch_ppro_float(const void *x, void *xf, size_t n, hipStream_t s)
{
    hipLaunchKernelGGL(k_ppro_float, dim3(grid_for(n)), dim3(256), 0, s, (const uint8_t *)x, (float *)xf, n);
    return hipGetLastError();
}

hipError_t launch_apply_res_float(const void *x, const void *res, void *rec, size_t n, hipStream_t s)
{
    hipLaunchKernelGGL(k_apply_res_float, dim3(grid_for(n)), dim3(256), 0, s, (const uint8_t *)x, (const float *)res,
                       (uint8_t *)rec, n);
    return hipGetLastError();
}

bool conv_float_supported(int cin, int cout, int ks)
{
    if (cout < 1 || cout > 64) return false;
    return (ks == 5 && cin == 1) || ((ks == 1 || ks == 3 || ks == 5) && (cin == 48 || cin == 64));
}

// Channel tiles (of 16) one launch may hold: bounded by the 64 KiB of LDS a workgroup declares statically
// (fragments of the chunk's weights + the chunk's planes).
template <int KS, int CIN>
constexpr int max_mt()
{
    using G = FGeom<KS, CIN>;
    int mt = 4;
    while (mt > 1 && (G::STEPS * mt * 65 + G::CC * G::PS) * 4 > 64 * 1024) --mt;
    return mt;
}

template <int KS, int CIN, int MT>
static void conv_f32_slice(const FConvArgs &a, int tiles, hipStream_t s)
{
    if constexpr (MT <= max_mt<KS, CIN>())
        hipLaunchKernelGGL((k_conv_f32<KS, CIN, MT>), dim3(tiles), dim3(256), 0, s, a);
}

// Output channels are covered in slices of up to 16 * max_mt channels, one launch each.
template <int KS, int CIN>
static hipError_t conv_f32_launch(const FConvArgs &a0, hipStream_t s)
{
    constexpr int MAXC = 16 * max_mt<KS, CIN>();
    const int tiles = 8 * ((a0.ntiles + 7) / 8);  // grid: see xcd_tile()
    const size_t HW = (size_t)a0.H * a0.W;
    if constexpr (CIN != 1) {
        if (a0.cout == 1) {
            hipLaunchKernelGGL((k_conv_f32_one<KS, CIN>), dim3(tiles), dim3(256), 0, s, a0);
            return hipGetLastError();
        }
    }
    for (int k0 = 0; k0 < a0.cout; k0 += MAXC) {
        FConvArgs a = a0;
        a.cout = a0.cout - k0 < MAXC ? a0.cout - k0 : MAXC;
        a.w = a0.w + (size_t)k0 * CIN * KS * KS;
        a.b = a0.b + k0;
        a.u = a0.u + (size_t)k0 * HW;
        switch ((a.cout + 15) / 16) {
        case 1: conv_f32_slice<KS, CIN, 1>(a, tiles, s); break;
        case 2: conv_f32_slice<KS, CIN, 2>(a, tiles, s); break;
        case 3: conv_f32_slice<KS, CIN, 3>(a, tiles, s); break;
        default: conv_f32_slice<KS, CIN, 4>(a, tiles, s); break;
        }
        hipError_t he = hipGetLastError();
        if (he != hipSuccess) return he;
    }
    return hipSuccess;
}

// u = conv(x, w) + b, optional ReLU; hipErrorInvalidValue for a shape conv_float_supported() rejects
hipError_t launch_conv_float(const void *x, const void *w, const void *b, void *u, int H, int W, int cin, int cout,
                             int ks, int relu, hipStream_t s)
{
    if (!conv_float_supported(cin, cout, ks)) return hipErrorInvalidValue;
    FConvArgs a;
    a.x = (const float *)x;
    a.w = (const float *)w;
    a.b = (const float *)b;
    a.u = (float *)u;
    a.relu = relu;
    a.H = H;
    a.W = W;
    a.cout = cout;
    a.tiles_x = (W + cf::TW - 1) / cf::TW;
    a.ntiles = a.tiles_x * ((H + cf::TH - 1) / cf::TH);
    if (cin == 1) return conv_f32_launch<5, 1>(a, s);
    if (cin == 48) return ks == 1 ? conv_f32_launch<1, 48>(a, s) : ks == 3 ? conv_f32_launch<3, 48>(a, s) : conv_f32_launch<5, 48>(a, s);
    return ks == 1 ? conv_f32_launch<1, 64>(a, s) : ks == 3 ? conv_f32_launch<3, 64>(a, s) : conv_f32_launch<5, 64>(a, s);
}

}  // namespace qcnn
