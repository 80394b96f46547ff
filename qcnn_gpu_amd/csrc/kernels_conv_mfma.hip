// Layer-level convolution on the int8 matrix cores, for the layer API, the tap mode and the
// dynamic-quantisation pass: CovLayer::ConvForward[_static] (inference/cnn.cu:129-160) on the reference's
// own layouts -- input int8 NCHW_VECT_C [C/4][H][W][4], weights int8 [K][C/4][ks][ks][4] as the model
// file holds them, output float32 NCHW -- with the passes the reference runs afterwards folded into
// the epilogue on request: cudnnAddTensor of the bias (always), CovLayer::activate (ReLU), findMax
// (largest |u|, inference/mat.cu:28-96) and CHW2CHW_VECT_C (divide-quantise + relayout,
// inference/mat.cu:197-236).
//
// Implicit GEMM D[channel][pixel] = W[channel][k] * X[k][pixel] with mfma_i32_16x16x64_i8: weights are the
// A operand, gathered once per workgroup from the file layout into lane-ordered fragments in LDS;
// activations are the B operand, read from an LDS tile whose pixels hold their channels contiguously
// (one 16-byte read per lane and fragment).  Persistent 256-thread workgroups walk 8 x 32 output tiles;
// wave w owns rows 2w, 2w+1 of the tile (four 16-pixel groups) and all output channels.
//
// The single-channel first layer (x_ppro, 4 bytes per pixel with the sample in lane 0) packs four
// horizontally adjacent taps into one 16-byte k-group: kernel row r is covered by two k-groups
// (columns 0..3 and 4..7, columns >= 5 with zero weights), 10 k-groups = 3 MFMA steps.
#include "fused_common.h"

namespace qcnn {

namespace cm {
constexpr int TH = 8, TW = 32;
}

struct ConvArgs {
    const int8_t *x;        // int8 NCHW_VECT_C
    const int8_t *w;        // file order
    const float *b;         // bias (b or b_adj), float[cout]
    float *u;               // float NCHW out, or null
    int8_t *q;              // int8 NCHW_VECT_C out (divide-quantised), or null
    unsigned *maxbits;      // largest |u| as float bits (atomic max), or null
    const int *divisor_dev; // divisor of the quantiser in device memory, or null: use `divisor`
    int divisor;
    int relu;
    int H, W, cout;
    int tiles_x, ntiles;
};

template <int KS, int CIN>
struct ConvGeom {
    static constexpr int P = (KS - 1) / 2;
    static constexpr int C4 = (CIN + 3) / 4;                     // vec4 groups of the input
    static constexpr int PXB = CIN == 1 ? 4 : CIN == 64 ? 80 : CIN;  // bytes per pixel in the LDS tile
    static constexpr int RH = cm::TH + KS - 1;
    static constexpr int RW = CIN == 1 ? cm::TW + 8 : cm::TW + KS - 1;
    static constexpr int NSTEPS = CIN == 1 ? 3 : KS * KS;        // MFMA K-steps
    static constexpr int SX_BYTES = RH * RW * PXB + 64;          // + slack: k-group 3 of a 48-byte pixel
    static_assert(CIN == 1 || CIN == 48 || CIN == 64, "input channel counts of the VRCNN layers");
    static_assert(CIN != 1 || KS == 5, "the packed first layer is 5x5");
};

__device__ __forceinline__ int quant_div_val(float t, float half, float d)
{
    if (t < 0.f) {
        t = __fdiv_rn(t - half, d);
        return t < -128.f ? -128 : (int)t;
    }
    t = __fdiv_rn(t + half, d);
    return t > 127.f ? 127 : (int)t;
}

template <int KS, int CIN, int KT>
__global__ __launch_bounds__(256, 2) void k_conv_mfma(const ConvArgs a)
{
    using G = ConvGeom<KS, CIN>;
    constexpr int TH = cm::TH, TW = cm::TW;
    __shared__ __attribute__((aligned(16))) unsigned char sA[G::NSTEPS * KT * 1024];
    __shared__ __attribute__((aligned(16))) unsigned char sX[G::SX_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int g = lane >> 4, col = lane & 15;
    const int H = a.H, W = a.W;
    const int *w32 = reinterpret_cast<const int *>(a.w);
    const int *x32 = reinterpret_cast<const int *>(a.x);

    // ---- A fragments, once per workgroup ------------------------------------------------------------
    for (int item = tid; item < G::NSTEPS * KT * 64; item += 256) {
        const int ln = item & 63, f = item >> 6;
        const int kt = f % KT, step = f / KT;
        const int k = 16 * kt + (ln & 15), gg = ln >> 4;
        v4i frag = {0, 0, 0, 0};
        if (k < a.cout) {
            if constexpr (CIN == 1) {
                const int gi = 4 * step + gg;
                if (gi < 10) {
                    const int r = gi >> 1, s0 = 4 * (gi & 1);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (s0 + j < 5) frag[j] = w32[(k * 5 + r) * 5 + s0 + j];
                }
            } else {
                const int r = step / KS, s = step % KS;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c4 = 4 * gg + j;
                    if (c4 < G::C4) frag[j] = w32[((k * G::C4 + c4) * KS + r) * KS + s];
                }
            }
        }
        *reinterpret_cast<v4i *>(sA + (size_t)f * 1024 + ln * 16) = frag;
    }
    if (tid < 16) *reinterpret_cast<int *>(sX + G::RH * G::RW * G::PXB + 4 * tid) = 0;  // the slack stays finite

    float bias[KT][4];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ch = 16 * kt + 4 * g + i;
            bias[kt][i] = ch < a.cout ? a.b[ch] : 0.f;
        }
    int divisor = a.divisor;
    if (a.q && a.divisor_dev) divisor = *a.divisor_dev;
    const float dvf = (float)divisor, half = (float)(divisor >> 1);
    float amax = 0.f;

    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
        const int y0 = ty * TH, x0 = tx * TW;
        __syncthreads();  // the previous tile's readers are done (and the A fragments are complete)
        // ---- input region -> LDS, pixel-major ------------------------------------------------------
        for (int idx = tid; idx < G::C4 * G::RH * G::RW; idx += 256) {
            const int rx = idx % G::RW, t2 = idx / G::RW;
            const int ry = t2 % G::RH, c4 = t2 / G::RH;
            const int gy = y0 + ry - G::P, gx = x0 + rx - G::P;
            int v = 0;
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = x32[((size_t)c4 * H + gy) * W + gx];
            *reinterpret_cast<int *>(sX + (ry * G::RW + rx) * G::PXB + 4 * c4) = v;
        }
        __syncthreads();
        // ---- MFMA ----------------------------------------------------------------------------------------
        v4i acc[4][KT];
#pragma unroll
        for (int pt = 0; pt < 4; ++pt)
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) acc[pt][kt] = v4i{0, 0, 0, 0};
#pragma unroll
        for (int step = 0; step < G::NSTEPS; ++step) {
            v4i af[KT];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
                af[kt] = *reinterpret_cast<const v4i *>(sA + (size_t)(step * KT + kt) * 1024 + lane * 16);
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) {
                const int row = 2 * wv + (pt >> 1), px = 16 * (pt & 1) + col;
                v4i bf;
                if constexpr (CIN == 1) {
                    int gi = 4 * step + g;
                    if (gi >= 10) gi = 0;  // zero weights: any readable address
                    const int *src = reinterpret_cast<const int *>(sX + ((row + (gi >> 1)) * G::RW + px + 4 * (gi & 1)) * 4);
                    bf = v4i{src[0], src[1], src[2], src[3]};
                } else {
                    const int r = step / KS, s = step % KS;
                    bf = *reinterpret_cast<const v4i *>(sX + ((row + r) * G::RW + px + s) * G::PXB + 16 * g);
                }
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) acc[pt][kt] = mfma16(af[kt], bf, acc[pt][kt]);
            }
        }
        // ---- epilogue ------------------------------------------------------------------------------------
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            const int gy = y0 + 2 * wv + (pt >> 1), gx = x0 + 16 * (pt & 1) + col;
            if (gy >= H || gx >= W) continue;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                const int ch0 = 16 * kt + 4 * g;
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float t = (float)acc[pt][kt][i] + bias[kt][i];
                    if (a.relu && t < 0.f) t = 0.f;
                    v[i] = t;
                }
                if (ch0 + 4 <= a.cout) {
                    amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
                    if (a.u) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) a.u[((size_t)(ch0 + i) * H + gy) * W + gx] = v[i];
                    }
                    if (a.q) {
                        unsigned pk = 0;
#pragma unroll
                        for (int i = 0; i < 4; ++i) pk |= (unsigned)(unsigned char)quant_div_val(v[i], half, dvf) << (8 * i);
                        reinterpret_cast<unsigned *>(a.q)[((size_t)(ch0 >> 2) * H + gy) * W + gx] = pk;
                    }
                } else {
                    for (int i = 0; ch0 + i < a.cout; ++i) {  // ragged last group (C4: one channel)
                        amax = fmaxf(amax, fabsf(v[i]));
                        if (a.u) a.u[((size_t)(ch0 + i) * H + gy) * W + gx] = v[i];
                        if (a.q) a.q[(((size_t)(ch0 >> 2) * H + gy) * W + gx) * 4 + i] = (int8_t)quant_div_val(v[i], half, dvf);
                    }
                }
            }
        }
    }
    if (a.maxbits) {
        for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_down(amax, off, 64));
        if (lane == 0 && amax > 0.f) atomicMax(a.maxbits, __float_as_uint(amax));
    }
}

template <int KS, int CIN, int KT>
static hipError_t conv_mfma_launch(const ConvArgs &a0, int ncu, hipStream_t s)
{
    ConvArgs a = a0;
    a.tiles_x = (a.W + cm::TW - 1) / cm::TW;
    a.ntiles = a.tiles_x * ((a.H + cm::TH - 1) / cm::TH);
    const int grid = a.ntiles < 2 * ncu ? a.ntiles : 2 * ncu;
    hipLaunchKernelGGL((k_conv_mfma<KS, CIN, KT>), dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError();
}

bool conv_mfma_supported(int cin, int cout, int ks)
{
    return (ks == 5 && cin == 1 && cout == 64) || (ks == 3 && cin == 64 && cout == 32) ||
           (ks == 5 && cin == 64 && cout == 16) || (ks == 3 && cin == 48 && cout == 16) ||
           (ks == 1 && cin == 48 && cout == 32) || (ks == 3 && cin == 48 && cout == 1);
}

// u / q / maxbits / divisor_dev may be null (see ConvArgs); returns hipErrorInvalidValue for a shape
// conv_mfma_supported() rejects.
hipError_t launch_conv_mfma(const void *x, const void *w, const void *b, void *u, void *q, unsigned *maxbits,
                            const int *divisor_dev, int divisor, int relu, int H, int W, int cin, int cout, int ks,
                            hipStream_t s)
{
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
            ncu = n;
        else
            ncu = 256;
    }
    ConvArgs a;
    a.x = (const int8_t *)x;
    a.w = (const int8_t *)w;
    a.b = (const float *)b;
    a.u = (float *)u;
    a.q = (int8_t *)q;
    a.maxbits = maxbits;
    a.divisor_dev = divisor_dev;
    a.divisor = divisor;
    a.relu = relu;
    a.H = H;
    a.W = W;
    a.cout = cout;
    a.tiles_x = a.ntiles = 0;
    if (ks == 5 && cin == 1 && cout == 64) return conv_mfma_launch<5, 1, 4>(a, ncu, s);
    if (ks == 3 && cin == 64 && cout == 32) return conv_mfma_launch<3, 64, 2>(a, ncu, s);
    if (ks == 5 && cin == 64 && cout == 16) return conv_mfma_launch<5, 64, 1>(a, ncu, s);
    if (ks == 3 && cin == 48 && cout == 16) return conv_mfma_launch<3, 48, 1>(a, ncu, s);
    if (ks == 1 && cin == 48 && cout == 32) return conv_mfma_launch<1, 48, 2>(a, ncu, s);
    if (ks == 3 && cin == 48 && cout == 1) return conv_mfma_launch<3, 48, 1>(a, ncu, s);
    return hipErrorInvalidValue;
}

}  // namespace qcnn
