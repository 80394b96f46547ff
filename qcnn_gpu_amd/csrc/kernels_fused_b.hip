// Kernel B of the fused forward pass (see fused_common.h for the overall design).
#include "fused_common.h"

namespace qcnn {

// =========================================================================================
// Kernel B
// =========================================================================================
template <bool EXACT>
__global__ __launch_bounds__(256, 2) void k_c3c4(const FusedArgs a)
{
    using namespace kb;
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
    unsigned char *sIn = smem;               // [INR][INC][48] conc1 tile (internal channel order)
    unsigned char *sC3 = smem + SIN_BYTES;   // [C3R][C3C][48] conc2 tile (internal channel order)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const int H = a.H, W = a.W;
    const int c = lane & 15, g = lane >> 4;
    const int gk = g == 3 ? 0 : g;  // k-group 3 multiplies zero weights; it re-reads chunk 0
    STAMP_INIT();
    STAMP();

    // All 14 A fragments of this kernel (C3_1: 9, C3_2: 2, C4: 3) are requested up front; they are
    // first used after the tile load below, which covers their latency.
    v4i f31[3][3], f32[2], f4[3];
    {
        const v4i *w31 = reinterpret_cast<const v4i *>(a.wpack + PackedOffsets::C3W31) + lane;
        const v4i *w32 = reinterpret_cast<const v4i *>(a.wpack + PackedOffsets::C3W32) + lane;
        const v4i *w4 = reinterpret_cast<const v4i *>(a.wpack + PackedOffsets::C4W) + lane;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
#pragma unroll
            for (int r = 0; r < 3; ++r) f31[s][r] = w31[(s * 3 + r) * 64];
            f4[s] = w4[s * 64];
        }
        f32[0] = w32[0];
        f32[1] = w32[64];
    }

    // The residual's Y bytes of this lane's 9 output pixels (C4 phase: wave = 9 rows x 16 columns)
    // are requested now, with clamped addresses and no branches; they are used at the very end.
    unsigned char ypix[9];
    const float b4 = a.bias_f[BIAS_OFF[5]];  // C4 bias, also requested early
    {
        const int gx = x0 + (wv & 1) * 16 + c;
        const int gxc = gx < W ? gx : W - 1;
#pragma unroll
        for (int y = 0; y < 9; ++y) {
            const int gy = y0 + (wv >> 1) * 9 + y;
            ypix[y] = a.y[(size_t)(gy < H ? gy : H - 1) * W + gxc];
        }
    }

    // ---- phase 0: conc1 tile, rows y0-2 .. y0+TH+1, cols x0-2 .. x0+TW+1, zero outside ----
    {
        constexpr int PIECES_ROW = INC * 3;           // 16-byte pieces per tile row
        constexpr int PIECES = INR * PIECES_ROW;
        constexpr int ITER = (PIECES + 255) / 256;
        v4i v[ITER];
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int p = it * 256 + tid;
            const int row = p / PIECES_ROW, rem = p - row * PIECES_ROW;
            const int px = rem / 3, ch = rem - px * 3;
            const int gy = y0 - 2 + row, gx = x0 - 2 + px;
            v[it] = v4i{0, 0, 0, 0};
            if (p < PIECES && gy >= 0 && gy < H && gx >= 0 && gx < W)
                v[it] = *reinterpret_cast<const v4i *>(a.conc1 + ((size_t)gy * W + gx) * 48 + ch * 16);
        }
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int p = it * 256 + tid;
            if (p < PIECES) *reinterpret_cast<v4i *>(sIn + p * 16) = v[it];
        }
    }
    STAMP();
    __syncthreads();
    STAMP();

    // ---- phase 1: C3_1 (3x3, 48 -> 16) and C3_2 (1x1, 48 -> 32) on the C3R x C3C region.
    // Wave = 10 region rows x 16 columns x 48 channels. ----
    {
        const int cb = (wv & 1) * 16, R0 = (wv >> 1) * 10;
        v4i acc31[10], acc32[2][10];
        {
            v4i b0 = {0, 0, 0, 0}, b1 = {0, 0, 0, 0}, b2 = {0, 0, 0, 0};
            if constexpr (!EXACT) {
                b0 = *reinterpret_cast<const v4i *>(a.bias_i + BIAS_OFF[3] + 4 * g);
                b1 = *reinterpret_cast<const v4i *>(a.bias_i + BIAS_OFF[4] + 4 * g);
                b2 = *reinterpret_cast<const v4i *>(a.bias_i + BIAS_OFF[4] + 16 + 4 * g);
            }
#pragma unroll
            for (int y = 0; y < 10; ++y) {
                acc31[y] = b0;
                acc32[0][y] = b1;
                acc32[1][y] = b2;
            }
        }
        const unsigned char *rowbase = sIn + (R0 * INC + cb + c) * PX + gk * 16;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
#pragma unroll
            for (int yy = 0; yy < 12; ++yy) {
                const v4i B = *reinterpret_cast<const v4i *>(rowbase + (yy * INC + s) * PX);
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int y = yy - r;
                    if (y >= 0 && y < 10) {
                        acc31[y] = mfma16(f31[s][r], B, acc31[y]);
                        if (s == 1 && r == 1) {
                            acc32[0][y] = mfma16(f32[0], B, acc32[0][y]);
                            acc32[1][y] = mfma16(f32[1], B, acc32[1][y]);
                        }
                    }
                }
            }
        }
        STAMP();
        mfma_drain();
        const Requant q31 = a.rq[3], q32 = a.rq[4];
        float bf[12];
        if constexpr (EXACT) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                bf[i] = a.bias_f[BIAS_OFF[3] + 4 * g + i];
                bf[4 + i] = a.bias_f[BIAS_OFF[4] + 4 * g + i];
                bf[8 + i] = a.bias_f[BIAS_OFF[4] + 16 + 4 * g + i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 12; ++i) bf[i] = 0.f;
        }
        const int jj = cb + c;
        const int gx = x0 - 1 + jj;
#pragma unroll
        for (int y = 0; y < 10; ++y) {
            const int i = R0 + y;
            const int gy = y0 - 1 + i;
            unsigned d0 = quant4<EXACT>(acc31[y][0], acc31[y][1], acc31[y][2], acc31[y][3], bf[0], bf[1], bf[2], bf[3], q31);
            unsigned d1 = quant4<EXACT>(acc32[0][y][0], acc32[0][y][1], acc32[0][y][2], acc32[0][y][3], bf[4], bf[5], bf[6],
                                        bf[7], q32);
            unsigned d2 = quant4<EXACT>(acc32[1][y][0], acc32[1][y][1], acc32[1][y][2], acc32[1][y][3], bf[8], bf[9], bf[10],
                                        bf[11], q32);
            if (!(gy >= 0 && gy < H && gx >= 0 && gx < W)) d0 = d1 = d2 = 0;  // zero padding of the C4 input
            unsigned *dst = reinterpret_cast<unsigned *>(sC3 + (i * C3C + jj) * PX + 12 * g);
            dst[0] = d0;
            dst[1] = d1;
            dst[2] = d2;
        }
    }
    STAMP();
    __syncthreads();
    STAMP();

    // ---- phase 2: C4 (3x3, 48 -> 1) + applyRes_y.  The three kernel rows r are the MFMA rows
    // 0..2: D_yy[r][pixel] = sum_s W4[r][s] . X[yy][pixel+s], and out[y] = D_y[0] + D_{y+1}[1] +
    // D_{y+2}[2] stays inside one lane.  Wave = 9 output rows x 16 columns. ----
    {
        const int cb = (wv & 1) * 16, O0 = (wv >> 1) * 9;
        const unsigned char *rowbase = sC3 + (O0 * C3C + cb + c) * PX + gk * 16;
        const int ox = cb + c;
        const int gx = x0 + ox;
        // 11 input rows x 3 kernel columns: the three MFMAs of one input row form a dependent chain,
        // so the chains of different rows are issued column by column
        v4i D[11];
#pragma unroll
        for (int yy = 0; yy < 11; ++yy) D[yy] = v4i{0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < 3; ++s) {
#pragma unroll
            for (int yy = 0; yy < 11; ++yy) {
                const v4i B = *reinterpret_cast<const v4i *>(rowbase + (yy * C3C + s) * PX);
                D[yy] = mfma16(f4[s], B, D[yy]);
            }
        }
        if (g == 0 && ox < TW && gx < W) {
            const int mul = a.rq[5].mul, shift = a.rq[5].shift;
#pragma unroll
            for (int y = 0; y < 9; ++y) {
                const int gy = y0 + O0 + y;
                if (gy < H) {
                    const int acc = D[y][0] + D[y + 1][1] + D[y + 2][2];
                    const float u = (float)acc + b4;  // C4.u, inference/cnn.cu:148-155
                    int temp = (int)u;                // applyRes_GPU_y, inference/cnn.cu:515-520
                    temp = (int)((unsigned)temp * (unsigned)mul + (1u << (shift - 1))) >> shift;
                    const short rec = (short)((int)ypix[y] + temp);
                    a.rec[(size_t)gy * W + gx] = rec > 255 ? (uint8_t)255 : rec < 0 ? (uint8_t)0 : (uint8_t)rec;
                }
            }
        }
    }
    STAMP();
}

hipError_t launch_fused_c3c4(const FusedArgs &a, bool exact, hipStream_t s)
{
    dim3 grid((a.W + kb::TW - 1) / kb::TW, (a.H + kb::TH - 1) / kb::TH);
    if (exact) hipLaunchKernelGGL(k_c3c4<true>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_c3c4<false>, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace qcnn
