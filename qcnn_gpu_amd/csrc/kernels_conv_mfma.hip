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
// a wave owns one 16-pixel strip of four rows and all output channels.
//
// The single-channel first layer (x_ppro, 4 bytes per pixel with the sample in lane 0) packs four
// horizontally adjacent taps into one 16-byte k-group: kernel row r is covered by two k-groups
// (columns 0..3 and 4..7, columns >= 5 with zero weights), 10 k-groups = 3 MFMA steps.
#include <atomic>
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
    static constexpr int RW = CIN == 1 ? cm::TW + 8 : cm::TW + KS - 1;  // region columns held in LDS
    static constexpr int X0 = P;                                 // the region starts X0 columns left of the tile
    static constexpr int NSTEPS = CIN == 1 ? 3 : KS * KS;        // MFMA K-steps
    static constexpr int SX_BYTES = RH * RW * PXB + 64;          // + slack: k-group 3 of a 48-byte pixel
    static_assert(CIN == 1 || CIN == 48 || CIN == 64, "input channel counts of the VRCNN layers");
    static_assert(CIN != 1 || KS == 5, "the packed first layer is 5x5");
};

// Correctly rounded float32 division by a divisor that is uniform over the kernel.  The compiler's expansion
// of an IEEE division is: scale, reciprocal, two refinement steps of the reciprocal, quotient, and two
// residual corrections, the last one fused with the unscaling.  The refined reciprocal depends on the
// divisor only and is computed once (div_recip); with operands that need no scaling (divisors are integers
// >= 1, dividends below 2^32 in magnitude) the remaining five instructions give the same quotient.
__device__ __forceinline__ float div_recip(float d)
{
    const float y0 = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, y0, 1.f);
    return __builtin_fmaf(e, y0, y0);
}
__device__ __forceinline__ float div_rn_by(float x, float d, float y)
{
    const float q0 = x * y;
    const float r0 = __builtin_fmaf(-d, q0, x);
    const float q1 = __builtin_fmaf(r0, y, q0);
    const float r1 = __builtin_fmaf(-d, q1, x);
    return __builtin_fmaf(r1, y, q1);
}

// CHW2CHW_VECT_C of one value (inference/mat.cu:211-226): (t -+ divisor/2) / divisor in float32, clipped, truncated
__device__ __forceinline__ int quant_div_val(float t, float half, float d, float y)
{
    if (t < 0.f) {
        t = div_rn_by(t - half, d, y);
        return t < -128.f ? -128 : (int)t;
    }
    t = div_rn_by(t + half, d, y);
    return t > 127.f ? 127 : (int)t;
}

// Workgroup barrier that orders LDS traffic only: global loads (the prefetched tile) and the stores of
// the epilogue stay in flight across it, which __syncthreads() would wait for.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// VEC (four horizontally adjacent pixels of one vec4 group as one 16-byte load, with the region widened to
// whole groups) measured slower than dword loads -- 133 against 112 us on the 5x5 64->16 layer at 1080p --
// and is not instantiated.
// MINB: waves per SIMD (= resident 256-thread workgroups per CU) the register allocation must leave room for.  These kernels wait on the fetch of
// their next tile, so a third resident workgroup pays for itself where the float image is not written (the
// passes of qcnn_forward); with the float stores it costs more in spills than it hides.
template <int KS, int CIN, int KT, bool VEC, int MINB>
__global__ __launch_bounds__(256, MINB) void k_conv_mfma(const ConvArgs a)
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
    const float dvf = (float)divisor, half = (float)(divisor >> 1), dvy = div_recip(dvf);
    float amax = 0.f;

    // The input region of a tile travels global -> registers -> LDS; the loads of the next tile are issued
    // before the MFMAs of the current one, so their latency is covered by compute instead of being paid
    // once per tile by every wave.  A thread keeps one column group of the region for a fixed set of rows
    // and vec4 groups, so the addresses of its loads and LDS stores differ by constants only.
    constexpr int CW = VEC ? 4 : 1;                        // pixels per load
    constexpr int NCOL = G::RW / CW;                       // loads per region row and vec4 group
    constexpr int RPP = (256 / NCOL) < G::RH ? (256 / NCOL) : G::RH;  // region rows covered by one pass
    constexpr int NCHUNK = (G::RH + RPP - 1) / RPP;
    constexpr int NGRP = (256 / (NCOL * RPP)) < G::C4 ? (256 / (NCOL * RPP)) : G::C4;  // thread groups sharing the vec4 groups
    constexpr int NC4 = (G::C4 + NGRP - 1) / NGRP;
    constexpr int NLOAD = NC4 * NCHUNK;
    const int fgrp = tid / (NCOL * RPP), fin = tid % (NCOL * RPP);
    const int fcx = fin % NCOL, fry = fin / NCOL;          // fgrp >= NGRP: the thread has no part in the copy
    const size_t plane = (size_t)H * W;
    typedef int prevec __attribute__((ext_vector_type(CW)));
    prevec pre[NLOAD];
    auto fetch = [&](int tile) {
        const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
        const int gx = tx * TW + CW * fcx - G::X0;
        const bool colok = fgrp < NGRP && gx >= 0 && gx < W;  // VEC: gx and W are multiples of 4
#pragma unroll
        for (int ch = 0; ch < NCHUNK; ++ch) {
            const int ry = fry + ch * RPP;
            const int gy = ty * TH + ry - G::P;
            const bool rowok = colok && ry < G::RH && gy >= 0 && gy < H;
            const int *src = x32 + ((size_t)gy * W + gx);
#pragma unroll
            for (int j = 0; j < NC4; ++j) {
                const int c4 = fgrp + j * NGRP;
                prevec v = {};
#ifndef CM_SKIP_FETCH
                if (rowok && c4 < G::C4) v = *reinterpret_cast<const prevec *>(src + c4 * plane);
#endif
                pre[ch * NC4 + j] = v;
            }
        }
    };
    // Workgroup -> tile map.  Workgroups are dealt to the 8 XCDs round robin (blockIdx.x & 7), each XCD with an
    // L2 of its own: XCD x walks the x-th eighth of the tiles (row-major), so that the halo rows and columns
    // neighbouring tiles share are fetched from HBM once per XCD band instead of once per tile.
#ifdef CM_NO_XCD_MAP
    const bool xcd_map = false;
#else
    const bool xcd_map = (gridDim.x & 7) == 0;
#endif
    const int per_xcd = (a.ntiles + 7) >> 3;
    auto tile_at = [&](int it) -> int {  // the it-th tile of this workgroup, -1 when it has none left
        if (!xcd_map) {
            const int t = (int)blockIdx.x + it * (int)gridDim.x;
            return t < a.ntiles ? t : -1;
        }
        const int loc = (int)(blockIdx.x >> 3) + it * (int)(gridDim.x >> 3);
        const int t = (int)(blockIdx.x & 7) * per_xcd + loc;
        return (loc < per_xcd && t < a.ntiles) ? t : -1;
    };
    int tile = tile_at(0);
    if (tile >= 0) fetch(tile);

    for (int it = 0; tile >= 0; ++it) {
        const int next_tile = tile_at(it + 1);
        const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
        const int y0 = ty * TH, x0 = tx * TW;
        lds_barrier();  // the previous tile's readers are done (and the A fragments are complete)
        // ---- input region -> LDS, pixel-major ------------------------------------------------------
#pragma unroll
        for (int ch = 0; ch < NCHUNK; ++ch) {
            const int ry = fry + ch * RPP;
            if (fgrp < NGRP && ry < G::RH) {
                unsigned char *dst = sX + (ry * G::RW + CW * fcx) * G::PXB;
#pragma unroll
                for (int j = 0; j < NC4; ++j) {
                    const int c4 = fgrp + j * NGRP;
                    if (c4 < G::C4) {
#pragma unroll
                        for (int px = 0; px < CW; ++px) {
                            int v;
                            v = pre[ch * NC4 + j][px];
                            *reinterpret_cast<int *>(dst + px * G::PXB + 4 * c4) = v;
                        }
                    }
                }
            }
        }
        lds_barrier();
        if (next_tile >= 0) fetch(next_tile);
        // ---- MFMA ----------------------------------------------------------------------------------------
        // A wave owns one 16-pixel strip and four rows of the tile.  For one kernel column s it keeps the A
        // fragments of all kernel rows in registers and walks the 4 + KS - 1 input rows once: the B fragment
        // of input row yy serves every output row yy - r it contributes to, so a fragment read from LDS
        // feeds up to KS * KT MFMAs.
        const int strip = wv & 1, rbase = 4 * (wv >> 1);
        v4i acc[4][KT];
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) acc[o][kt] = v4i{0, 0, 0, 0};
        if constexpr (CIN == 1) {
#pragma unroll
            for (int step = 0; step < G::NSTEPS; ++step) {
                v4i af[KT];
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
                    af[kt] = *reinterpret_cast<const v4i *>(sA + (size_t)(step * KT + kt) * 1024 + lane * 16);
                int gi = 4 * step + g;
                if (gi >= 10) gi = 0;  // zero weights: any readable address
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const int *src = reinterpret_cast<const int *>(
                        sX + ((rbase + o + (gi >> 1)) * G::RW + 16 * strip + col + 4 * (gi & 1)) * 4);  // X0 == P: column px + s
                    const v4i bf = v4i{src[0], src[1], src[2], src[3]};
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) acc[o][kt] = mfma16(af[kt], bf, acc[o][kt]);
                }
            }
        } else {
#pragma unroll 1
            for (int sft = 0; sft < KS; ++sft) {
                v4i af[KS][KT];
#pragma unroll
                for (int r = 0; r < KS; ++r)
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt)
                        af[r][kt] = *reinterpret_cast<const v4i *>(sA + (size_t)((r * KS + sft) * KT + kt) * 1024 + lane * 16);
                const unsigned char *bp = sX + ((size_t)rbase * G::RW + G::X0 - G::P + 16 * strip + col + sft) * G::PXB + 16 * g;
#pragma unroll
                for (int yy = 0; yy < 4 + KS - 1; ++yy) {
                    const v4i bf = *reinterpret_cast<const v4i *>(bp + (size_t)yy * G::RW * G::PXB);
#pragma unroll
                    for (int r = 0; r < KS; ++r) {
                        const int o = yy - r;
                        if (o >= 0 && o < 4) {
#pragma unroll
                            for (int kt = 0; kt < KT; ++kt) {
#ifndef CM_SKIP_MFMA
                                acc[o][kt] = mfma16(af[r][kt], bf, acc[o][kt]);
#else
                                acc[o][kt] += af[r][kt] ^ bf;
#endif
                            }
                        }
                    }
                }
            }
        }
        // ---- epilogue ------------------------------------------------------------------------------------
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            const int gy = y0 + rbase + pt, gx = x0 + 16 * strip + col;
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
#ifndef CM_SKIP_STORE
                    if (a.u) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) a.u[((size_t)(ch0 + i) * H + gy) * W + gx] = v[i];
                    }
#endif
                    if (a.q) {
                        unsigned pk = 0;
#pragma unroll
                        for (int i = 0; i < 4; ++i) pk |= (unsigned)(unsigned char)quant_div_val(v[i], half, dvf, dvy) << (8 * i);
                        reinterpret_cast<unsigned *>(a.q)[((size_t)(ch0 >> 2) * H + gy) * W + gx] = pk;
                    }
                } else {
                    for (int i = 0; ch0 + i < a.cout; ++i) {  // ragged last group (C4: one channel)
                        amax = fmaxf(amax, fabsf(v[i]));
                        if (a.u) a.u[((size_t)(ch0 + i) * H + gy) * W + gx] = v[i];
                        if (a.q) a.q[(((size_t)(ch0 >> 2) * H + gy) * W + gx) * 4 + i] = (int8_t)quant_div_val(v[i], half, dvf, dvy);
                    }
                }
            }
        }
        tile = next_tile;
    }
    if (a.maxbits) {
        for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_down(amax, off, 64));
        if (lane == 0 && amax > 0.f) atomicMax(a.maxbits, __float_as_uint(amax));
    }
}

// ---- two branches on one input ------------------------------------------------------------------------
// The layer kernels above are bound by fetching their input tiles, and the two branches of a VRCNN layer
// (C2_1 || C2_2 on C1.v, C3_1 || C3_2 on Conc1.conc) read the same input: this kernel stages the tile once
// (with the halo of the larger kernel) and runs both convolutions on it, each with its own weights, bias,
// outputs, maximum and divisor.  Results are those of two k_conv_mfma launches.
struct ConvBranch {
    const int8_t *w;
    const float *b;
    float *u;
    int8_t *q;
    unsigned *maxbits;
    const int *divisor_dev;
    int divisor;
    int cout;
};
struct ConvPairArgs {
    const int8_t *x;
    ConvBranch br[2];
    int relu;
    int H, W;
    int tiles_x, ntiles;
};

// A fragments of one branch, file order -> lane order, into LDS
template <int KS, int CIN, int KT>
__device__ __forceinline__ void pair_load_a(unsigned char *sA, const ConvBranch &br, int tid)
{
    constexpr int C4 = (CIN + 3) / 4;
    const int *w32 = reinterpret_cast<const int *>(br.w);
    for (int item = tid; item < KS * KS * KT * 64; item += 256) {
        const int ln = item & 63, f = item >> 6;
        const int kt = f % KT, step = f / KT;
        const int k = 16 * kt + (ln & 15), gg = ln >> 4;
        const int r = step / KS, s = step % KS;
        v4i frag = {0, 0, 0, 0};
        if (k < br.cout) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c4 = 4 * gg + j;
                if (c4 < C4) frag[j] = w32[((k * C4 + c4) * KS + r) * KS + s];
            }
        }
        *reinterpret_cast<v4i *>(sA + (size_t)f * 1024 + ln * 16) = frag;
    }
}

// One branch on the staged tile: MFMAs of this wave's 4 rows x 16 pixels, then the epilogue.  D = rows /
// columns by which this branch's window starts inside the region (the region has the larger kernel's halo).
template <int KS, int KT, int KSM, int CIN>
__device__ __forceinline__ void pair_branch(const unsigned char *sA, const unsigned char *sX, const ConvBranch &br, int relu,
                                            int H, int W, int y0, int x0, int lane, int wv, float dvf, float half, float dvy, float &amax)
{
    using G = ConvGeom<KSM, CIN>;
    constexpr int D = (KSM - KS) / 2;
    const int g = lane >> 4, col = lane & 15;
    const int strip = wv & 1, rbase = 4 * (wv >> 1);
    v4i acc[4][KT];
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) acc[o][kt] = v4i{0, 0, 0, 0};
#pragma unroll
    for (int sft = 0; sft < KS; ++sft) {
        v4i af[KS][KT];
#pragma unroll
        for (int r = 0; r < KS; ++r)
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
                af[r][kt] = *reinterpret_cast<const v4i *>(sA + (size_t)((r * KS + sft) * KT + kt) * 1024 + lane * 16);
        const unsigned char *bp = sX + ((size_t)(rbase + D) * G::RW + D + 16 * strip + col + sft) * G::PXB + 16 * g;
#pragma unroll
        for (int yy = 0; yy < 4 + KS - 1; ++yy) {
            const v4i bf = *reinterpret_cast<const v4i *>(bp + (size_t)yy * G::RW * G::PXB);
#pragma unroll
            for (int r = 0; r < KS; ++r) {
                const int o = yy - r;
                if (o >= 0 && o < 4) {
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) acc[o][kt] = mfma16(af[r][kt], bf, acc[o][kt]);
                }
            }
        }
    }
    float bias[KT][4];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ch = 16 * kt + 4 * g + i;
            bias[kt][i] = ch < br.cout ? br.b[ch] : 0.f;
        }
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {
        const int gy = y0 + rbase + pt, gx = x0 + 16 * strip + col;
        if (gy >= H || gx >= W) continue;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const int ch0 = 16 * kt + 4 * g;
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float t = (float)acc[pt][kt][i] + bias[kt][i];
                if (relu && t < 0.f) t = 0.f;
                v[i] = t;
            }
            if (ch0 + 4 <= br.cout) {
                amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
                if (br.u) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) br.u[((size_t)(ch0 + i) * H + gy) * W + gx] = v[i];
                }
                if (br.q) {
                    unsigned pk = 0;
#pragma unroll
                    for (int i = 0; i < 4; ++i) pk |= (unsigned)(unsigned char)quant_div_val(v[i], half, dvf, dvy) << (8 * i);
                    reinterpret_cast<unsigned *>(br.q)[((size_t)(ch0 >> 2) * H + gy) * W + gx] = pk;
                }
            } else {
                for (int i = 0; ch0 + i < br.cout; ++i) {
                    amax = fmaxf(amax, fabsf(v[i]));
                    if (br.u) br.u[((size_t)(ch0 + i) * H + gy) * W + gx] = v[i];
                    if (br.q) br.q[(((size_t)(ch0 >> 2) * H + gy) * W + gx) * 4 + i] = (int8_t)quant_div_val(v[i], half, dvf, dvy);
                }
            }
        }
    }
}

template <int CIN, int KSA, int KTA, int KSB, int KTB, int MINB>
__global__ __launch_bounds__(256, MINB) void k_conv_mfma_pair(const ConvPairArgs a)
{
    constexpr int KSM = KSA > KSB ? KSA : KSB;
    using G = ConvGeom<KSM, CIN>;
    constexpr int TH = cm::TH, TW = cm::TW;
    static_assert(CIN == 48 || CIN == 64, "the paired layers read 48 or 64 channels");
    __shared__ __attribute__((aligned(16))) unsigned char sAa[KSA * KSA * KTA * 1024];
    __shared__ __attribute__((aligned(16))) unsigned char sAb[KSB * KSB * KTB * 1024];
    __shared__ __attribute__((aligned(16))) unsigned char sX[G::SX_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int H = a.H, W = a.W;
    const int *x32 = reinterpret_cast<const int *>(a.x);
    pair_load_a<KSA, CIN, KTA>(sAa, a.br[0], tid);
    pair_load_a<KSB, CIN, KTB>(sAb, a.br[1], tid);
    if (tid < 16) *reinterpret_cast<int *>(sX + G::RH * G::RW * G::PXB + 4 * tid) = 0;  // the slack stays finite

    float dvf[2], half[2], dvy[2], amax[2] = {0.f, 0.f};
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        int divisor = a.br[b].divisor;
        if (a.br[b].q && a.br[b].divisor_dev) divisor = *a.br[b].divisor_dev;
        dvf[b] = (float)divisor, half[b] = (float)(divisor >> 1), dvy[b] = div_recip(dvf[b]);
    }

    // tile fetch: global -> registers (issued one tile ahead) -> LDS, as in k_conv_mfma
    constexpr int NCOL = G::RW;
    constexpr int RPP = (256 / NCOL) < G::RH ? (256 / NCOL) : G::RH;
    constexpr int NCHUNK = (G::RH + RPP - 1) / RPP;
    constexpr int NGRP = (256 / (NCOL * RPP)) < G::C4 ? (256 / (NCOL * RPP)) : G::C4;
    constexpr int NC4 = (G::C4 + NGRP - 1) / NGRP;
    const int fgrp = tid / (NCOL * RPP), fin = tid % (NCOL * RPP);
    const int fcx = fin % NCOL, fry = fin / NCOL;
    const size_t plane = (size_t)H * W;
    int pre[NC4 * NCHUNK];
    auto fetch = [&](int tile) {
        const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
        const int gx = tx * TW + fcx - G::X0;
        const bool colok = fgrp < NGRP && gx >= 0 && gx < W;
#pragma unroll
        for (int ch = 0; ch < NCHUNK; ++ch) {
            const int ry = fry + ch * RPP;
            const int gy = ty * TH + ry - G::P;
            const bool rowok = colok && ry < G::RH && gy >= 0 && gy < H;
            const int *src = x32 + ((size_t)gy * W + gx);
#pragma unroll
            for (int j = 0; j < NC4; ++j) {
                const int c4 = fgrp + j * NGRP;
                int v = 0;
                if (rowok && c4 < G::C4) v = src[c4 * plane];
                pre[ch * NC4 + j] = v;
            }
        }
    };
    const bool xcd_map = (gridDim.x & 7) == 0;
    const int per_xcd = (a.ntiles + 7) >> 3;
    auto tile_at = [&](int it) -> int {
        if (!xcd_map) {
            const int t = (int)blockIdx.x + it * (int)gridDim.x;
            return t < a.ntiles ? t : -1;
        }
        const int loc = (int)(blockIdx.x >> 3) + it * (int)(gridDim.x >> 3);
        const int t = (int)(blockIdx.x & 7) * per_xcd + loc;
        return (loc < per_xcd && t < a.ntiles) ? t : -1;
    };
    int tile = tile_at(0);
    if (tile >= 0) fetch(tile);
    for (int it = 0; tile >= 0; ++it) {
        const int next_tile = tile_at(it + 1);
        const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
        const int y0 = ty * TH, x0 = tx * TW;
        lds_barrier();
#pragma unroll
        for (int ch = 0; ch < NCHUNK; ++ch) {
            const int ry = fry + ch * RPP;
            if (fgrp < NGRP && ry < G::RH) {
                unsigned char *dst = sX + (ry * G::RW + fcx) * G::PXB;
#pragma unroll
                for (int j = 0; j < NC4; ++j) {
                    const int c4 = fgrp + j * NGRP;
                    if (c4 < G::C4) *reinterpret_cast<int *>(dst + 4 * c4) = pre[ch * NC4 + j];
                }
            }
        }
        lds_barrier();
        if (next_tile >= 0) fetch(next_tile);
        pair_branch<KSA, KTA, KSM, CIN>(sAa, sX, a.br[0], a.relu, H, W, y0, x0, lane, wv, dvf[0], half[0], dvy[0], amax[0]);
        pair_branch<KSB, KTB, KSM, CIN>(sAb, sX, a.br[1], a.relu, H, W, y0, x0, lane, wv, dvf[1], half[1], dvy[1], amax[1]);
        tile = next_tile;
    }
#pragma unroll
    for (int b = 0; b < 2; ++b)
        if (a.br[b].maxbits) {
            float m = amax[b];
            for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off, 64));
            if (lane == 0 && m > 0.f) atomicMax(a.br[b].maxbits, __float_as_uint(m));
        }
}

template <int CIN, int KSA, int KTA, int KSB, int KTB, int MINB>
static hipError_t conv_pair_launch(ConvPairArgs a, int ncu, hipStream_t s)
{
    a.tiles_x = (a.W + cm::TW - 1) / cm::TW;
    a.ntiles = a.tiles_x * ((a.H + cm::TH - 1) / cm::TH);
    // resident workgroups per CU of this instantiation: the same on every GPU of a node, so one cached value
    // serves all of them; atomic because one host thread per GPU may get here at the same time
    static std::atomic<int> per_cu_cache{0};
    int per_cu = per_cu_cache.load(std::memory_order_relaxed);
    if (per_cu == 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_conv_mfma_pair<CIN, KSA, KTA, KSB, KTB, MINB>, 256, 0) != hipSuccess || n < 1)
            n = 1;
        per_cu = n;
        per_cu_cache.store(n, std::memory_order_relaxed);
    }
    const int grid = a.ntiles < per_cu * ncu ? a.ntiles : per_cu * ncu;
    hipLaunchKernelGGL((k_conv_mfma_pair<CIN, KSA, KTA, KSB, KTB, MINB>), dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError();
}

// compute units of the current device (asked every time: the threads of a multi-GPU host drive different devices)
static int device_cus()
{
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
        return n;
    return 256;
}

// Branch pairs of the network: (3x3 64->32 || 5x5 64->16) and (3x3 48->16 || 1x1 48->32).
bool conv_pair_supported(int cin, int cout_a, int ks_a, int cout_b, int ks_b)
{
    return (cin == 64 && ks_a == 3 && cout_a == 32 && ks_b == 5 && cout_b == 16) ||
           (cin == 48 && ks_a == 3 && cout_a == 16 && ks_b == 1 && cout_b == 32);
}

// Both branches of a layer in one launch; per branch: w, b, u / q / maxbits / divisor_dev as in launch_conv_mfma.
hipError_t launch_conv_mfma_pair(const void *x, int cin, int H, int W, int relu, const ConvPairBranch &ba, const ConvPairBranch &bb,
                                 hipStream_t s)
{
    if (!conv_pair_supported(cin, ba.cout, ba.ks, bb.cout, bb.ks)) return hipErrorInvalidValue;
    ConvPairArgs a;
    a.x = (const int8_t *)x;
    const ConvPairBranch *src[2] = {&ba, &bb};
    for (int i = 0; i < 2; ++i) {
        a.br[i].w = (const int8_t *)src[i]->w;
        a.br[i].b = (const float *)src[i]->b;
        a.br[i].u = (float *)src[i]->u;
        a.br[i].q = (int8_t *)src[i]->q;
        a.br[i].maxbits = src[i]->maxbits;
        a.br[i].divisor_dev = src[i]->divisor_dev;
        a.br[i].divisor = src[i]->divisor;
        a.br[i].cout = src[i]->cout;
    }
    a.relu = relu;
    a.H = H;
    a.W = W;
    a.tiles_x = a.ntiles = 0;
    if (cin == 64) return conv_pair_launch<64, 3, 2, 5, 1, 2>(a, device_cus(), s);
    if (a.br[0].u || a.br[1].u) return conv_pair_launch<48, 3, 1, 1, 2, 2>(a, device_cus(), s);
    return conv_pair_launch<48, 3, 1, 1, 2, 3>(a, device_cus(), s);
}

template <int KS, int CIN, int KT, bool VEC, int MINB = 2>
static hipError_t conv_mfma_launch_v(const ConvArgs &a0, int ncu, hipStream_t s)
{
    ConvArgs a = a0;
    a.tiles_x = (a.W + cm::TW - 1) / cm::TW;
    a.ntiles = a.tiles_x * ((a.H + cm::TH - 1) / cm::TH);
    // persistent grid: as many workgroups as the device keeps resident (registers and LDS of this instantiation)
    static std::atomic<int> per_cu_cache{0};
    int per_cu = per_cu_cache.load(std::memory_order_relaxed);
    if (per_cu == 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_conv_mfma<KS, CIN, KT, VEC, MINB>, 256, 0) != hipSuccess || n < 1) n = 2;
        per_cu = n;
        per_cu_cache.store(n, std::memory_order_relaxed);
    }
    const int grid = a.ntiles < per_cu * ncu ? a.ntiles : per_cu * ncu;
    hipLaunchKernelGGL((k_conv_mfma<KS, CIN, KT, VEC, MINB>), dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError();
}

template <int KS, int CIN, int KT>
static hipError_t conv_mfma_launch(const ConvArgs &a, int ncu, hipStream_t s)
{
    return conv_mfma_launch_v<KS, CIN, KT, false>(a, ncu, s);
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
    const int ncu = device_cus();
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
    if (ks == 5 && cin == 1 && cout == 64)
        return a.u ? conv_mfma_launch_v<5, 1, 4, false, 2>(a, ncu, s) : conv_mfma_launch_v<5, 1, 4, false, 3>(a, ncu, s);
    if (ks == 3 && cin == 64 && cout == 32) return conv_mfma_launch<3, 64, 2>(a, ncu, s);
    if (ks == 5 && cin == 64 && cout == 16) return conv_mfma_launch<5, 64, 1>(a, ncu, s);
    if (ks == 3 && cin == 48 && cout == 16) return conv_mfma_launch<3, 48, 1>(a, ncu, s);
    if (ks == 1 && cin == 48 && cout == 32) return conv_mfma_launch<1, 48, 2>(a, ncu, s);
    if (ks == 3 && cin == 48 && cout == 1) return conv_mfma_launch<3, 48, 1>(a, ncu, s);
    return hipErrorInvalidValue;
}

}  // namespace qcnn
