// The production forward pass of qvrcnn::forward_blu (inference/qvrcnn.cu:168-242) as two fused
// gfx950 kernels built on the int8 MFMA instructions:
//
//   k_c1c2 : Y tile -> I1.ppro -> C1 conv+bias -> ReLU/clip/requant -> (C2_1 || C2_2) conv+bias ->
//            ReLU/clip/requant -> concat, written once to HBM as [H][W][48] int8
//   k_c3c4 : conc1 tile -> (C3_1 || C3_2) conv+bias -> ReLU/clip/requant -> concat (LDS only) ->
//            C4 conv+bias -> applyRes_y residual add + clamp -> I1.x_rec
//
// Every convolution is an implicit GEMM D[channel][pixel] = W[channel][k] * X[k][pixel]: the
// weights are the MFMA A operand (pre-packed on the host in lane order), the activations the B
// operand, read from an LDS tile whose pixels hold their channels contiguously, so a B fragment
// is one 16-byte LDS read per lane.  A wave keeps the B fragment of one (input row, column
// shift) in registers and reuses it for every kernel row r that maps it to one of the wave's
// output rows, and keeps the A fragments of one kernel column in registers while it sweeps the
// input rows; that is what holds LDS and L1 traffic per MFMA low.
//
// Two instantiations of each kernel exist.  EXACT=false is used when the host has proved, from
// the loaded weights, that no accumulator or bias can reach 2^24: then the reference's float32
// pipeline (float(acc) + float(b), float compares, (int) truncation) is identical to integer
// arithmetic and the bias is folded into the MFMA accumulator.  EXACT=true replays the float32
// pipeline literally and is bit-exact for every possible model.
#include "qcnn_internal.h"

namespace qcnn {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__device__ __forceinline__ v4i mfma16(v4i a, v4i b, v4i c)
{
    return __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ v16i mfma32(v4i a, v4i b, v16i c)
{
    return __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
}

// Integer requantiser, valid when the host-side proof holds (see engine.hip, prove_integer_path):
// s is acc + bias, exactly what the reference's float u holds.  Both clip arms are one v_med3
// (requant(blu) == 127 is part of the proof), and mul/cmul are pre-shifted on the host so that
// the quotient ((v + bias) * mul) >> shift is a whole byte (byte 2 or 3) of the 32-bit product:
// the shift is then done by the byte-permute that packs four channels into a dword.
// The clamp is inline asm (no integer med3 builtin exists and the compiler only forms one from
// constant bounds).  hipcc does not pad the MFMA-result -> VALU-read hazard for instructions
// inside an asm statement, so every epilogue that applies med3i to accumulators first calls
// mfma_drain(): 20 wait states, more than the 11 an 8-pass MFMA needs, fenced so that nothing is
// scheduled across it.
__device__ __forceinline__ void mfma_drain()
{
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ int med3i(int x, int hi)
{
    int r;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(x), "v"(hi));
    return r;
}
__device__ __forceinline__ unsigned requant_prod(int s, const Requant &q)
{
    return __umul24((unsigned)med3i(s, q.blu), (unsigned)q.mulk) + (unsigned)q.cmulk;
}
// Literal float32 pipeline of ConvForward_static + CHW2CHW_VECT_C_QUANT_BLU.
__device__ __forceinline__ int requant_float(int acc, float bf, const Requant &q)
{
    const float t = (float)acc + bf;
    if (t > q.bluf) return 127;
    if (t < 0.f) return 0;
    return (int)((unsigned)((int)t + q.bias) * (unsigned)q.mul) >> q.shift;
}

__device__ __forceinline__ unsigned pack4(int a, int b, int c, int d)
{
    // low bytes of a,b,c,d -> one dword (the store-to-char truncation of the reference)
    const unsigned lo = __builtin_amdgcn_perm((unsigned)b, (unsigned)a, 0x0c0c0400u);
    const unsigned hi = __builtin_amdgcn_perm((unsigned)d, (unsigned)c, 0x0c0c0400u);
    return __builtin_amdgcn_perm(hi, lo, 0x05040100u);
}

// Requantises four consecutive channels (accumulators a0..a3, float biases only used by the
// EXACT variant) and packs them into one dword.
template <bool EXACT>
__device__ __forceinline__ unsigned quant4(int a0, int a1, int a2, int a3, float b0, float b1, float b2, float b3,
                                           const Requant &q)
{
    if constexpr (EXACT) {
        return pack4(requant_float(a0, b0, q), requant_float(a1, b1, q), requant_float(a2, b2, q),
                     requant_float(a3, b3, q));
    } else {
        const unsigned lo = __builtin_amdgcn_perm(requant_prod(a1, q), requant_prod(a0, q), q.sel2);
        const unsigned hi = __builtin_amdgcn_perm(requant_prod(a3, q), requant_prod(a2, q), q.sel2);
        return __builtin_amdgcn_perm(hi, lo, 0x05040100u);
    }
}

// Byte offset of 16-byte channel chunk `chunk` (0..3) inside the 64-byte pixel at LDS column
// `col`: the chunk position is rotated by col/2.  With it the 16-lane groups of a ds_read_b128
// B-fragment read (16 consecutive pixels x 4 chunks) hit 16 distinct 16-byte bank slots, and the
// 8-lane groups of the ds_write_b128 that fills the tile (8 consecutive pixels, one chunk) hit 8
// distinct slots of the 128-byte write window.
__device__ __forceinline__ int swz(int chunk, int col) { return ((chunk + (col >> 1)) & 3) * 16; }

// A fragments of one kernel column s of the C2 stage: C2_2 rows r = 0..4 and, for the three inner
// columns, C2_1 rows r' = 0..2 x two 16-channel tiles.
struct C2Weights {
    v4i f22[5];
    v4i f21[3][2];
};
__device__ __forceinline__ void load_c2_weights(const v4i *w22, const v4i *w21, int s, C2Weights &o)
{
#pragma unroll
    for (int r = 0; r < 5; ++r) o.f22[r] = w22[(s * 5 + r) * 64];
    if (s >= 1 && s <= 3) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            o.f21[r][0] = w21[(((s - 1) * 3 + r) * 2 + 0) * 64];
            o.f21[r][1] = w21[(((s - 1) * 3 + r) * 2 + 1) * 64];
        }
    }
}

// =========================================================================================
// Kernel A
// =========================================================================================
template <bool EXACT>
__global__ __launch_bounds__(256, 2) void k_c1c2(const FusedArgs a)
{
    using namespace ka;
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
    unsigned char *sC1 = smem;                       // [C1R][C1C][64] int8, chunk-swizzled
    unsigned char *sY16 = smem + SC1_BYTES;          // [Y16R][C1C][16] im2col image for C1
    unsigned char *sY = smem + SC1_BYTES + SY16_BYTES;  // [YR][YC] int8 (x - 128), 0 outside the frame

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const int H = a.H, W = a.W;
    // C2 weights of kernel column 0: requested now, consumed after the C1 phase
    const v4i *w22 = reinterpret_cast<const v4i *>(a.wpack + PackedOffsets::C2W22) + lane;
    const v4i *w21 = reinterpret_cast<const v4i *>(a.wpack + PackedOffsets::C2W21) + lane;
    C2Weights w0;
    load_c2_weights(w22, w21, 0, w0);

    // ---- phase 0: Y tile, rows y0-4 .. y0+TH+4, cols x0-4 .. x0+TW+3; I1.ppro is the ^0x80 ----
    for (int idx = tid; idx < YR * YC; idx += 256) {
        const int row = idx / YC, col = idx - row * YC;
        const int gy = y0 - 4 + row, gx = x0 - 4 + col;
        unsigned char v = 0;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = a.y[(size_t)gy * W + gx] ^ 0x80;
        sY[idx] = v;
    }
    __syncthreads();

    // ---- phase 1: im2col image.  Entry (rho, j) = 16 bytes: for the three Y rows rho..rho+2 the
    // bytes at tile columns j..j+3 (12 bytes), then the three bytes at column j+4, then 0.  A C1
    // output at region (i, j) reads entries (i, j) and (i+3, j): 5 rows x 5 columns of taps. ----
    {
        const unsigned *sYd = reinterpret_cast<const unsigned *>(sY);
        for (int e = tid; e < Y16R * C1C; e += 256) {
            const int rho = e / C1C, j = e - rho * C1C;
            const int sh = j & 3;
            unsigned lo4[3], b5[3];
#pragma unroll
            for (int rr = 0; rr < 3; ++rr) {
                const int d = (rho + rr) * (YC / 4) + (j >> 2);
                const unsigned lo = sYd[d], hi = sYd[d + 1];
                lo4[rr] = __builtin_amdgcn_alignbyte(hi, lo, sh);
                b5[rr] = (hi >> (8 * sh)) & 0xffu;
            }
            v4i o;
            o.x = (int)lo4[0];
            o.y = (int)lo4[1];
            o.z = (int)lo4[2];
            o.w = (int)(b5[0] | (b5[1] << 8) | (b5[2] << 16));
            *reinterpret_cast<v4i *>(sY16 + e * 16) = o;
        }
    }
    __syncthreads();

    // ---- phase 2: C1 (5x5, 1 -> 64) on the C1R x C1C region, 32x32x32 MFMA, K = 25 taps padded
    // to 32.  Lane (pixel = lane&31, h = lane>>5) ends up with channels 16h..16h+15 of each
    // 32-channel tile in its 16 accumulator registers (the host permutes the weight rows), i.e.
    // one 16-byte chunk of the pixel. ----
    {
        const v4i *wp = reinterpret_cast<const v4i *>(a.wpack + PackedOffsets::C1W);
        const v4i A0 = wp[lane], A1 = wp[64 + lane];
        const int h = lane >> 5;
        const Requant q = a.rq[0];
        v16i bias0 = {}, bias1 = {};
        float bf0[16], bf1[16];
        if constexpr (!EXACT) {
            bias0 = *reinterpret_cast<const v16i *>(a.bias_i + 16 * h);
            bias1 = *reinterpret_cast<const v16i *>(a.bias_i + 32 + 16 * h);
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                bf0[i] = a.bias_f[16 * h + i];
                bf1[i] = a.bias_f[32 + 16 * h + i];
            }
        }
        constexpr int NPIX = C1R * C1C;
        constexpr int NT = (NPIX + 31) / 32;
        for (int t = wv; t < NT; t += 4) {
            const int qi = t * 32 + (lane & 31);
            const int qc = qi < NPIX ? qi : NPIX - 1;
            const int i = qc / C1C, j = qc - i * C1C;
            const v4i B = *reinterpret_cast<const v4i *>(sY16 + ((i + 3 * h) * C1C + j) * 16);
            v16i d0 = mfma32(A0, B, bias0);
            v16i d1 = mfma32(A1, B, bias1);
            mfma_drain();
            const int gy = y0 - 2 + i, gx = x0 - 2 + j;
            const bool inside = gy >= 0 && gy < H && gx >= 0 && gx < W;
            v4i o0, o1;
            if constexpr (!EXACT) {
                o0.x = quant4<false>(d0[0], d0[1], d0[2], d0[3], 0.f, 0.f, 0.f, 0.f, q);
                o0.y = quant4<false>(d0[4], d0[5], d0[6], d0[7], 0.f, 0.f, 0.f, 0.f, q);
                o0.z = quant4<false>(d0[8], d0[9], d0[10], d0[11], 0.f, 0.f, 0.f, 0.f, q);
                o0.w = quant4<false>(d0[12], d0[13], d0[14], d0[15], 0.f, 0.f, 0.f, 0.f, q);
                o1.x = quant4<false>(d1[0], d1[1], d1[2], d1[3], 0.f, 0.f, 0.f, 0.f, q);
                o1.y = quant4<false>(d1[4], d1[5], d1[6], d1[7], 0.f, 0.f, 0.f, 0.f, q);
                o1.z = quant4<false>(d1[8], d1[9], d1[10], d1[11], 0.f, 0.f, 0.f, 0.f, q);
                o1.w = quant4<false>(d1[12], d1[13], d1[14], d1[15], 0.f, 0.f, 0.f, 0.f, q);
            } else {
                o0.x = quant4<true>(d0[0], d0[1], d0[2], d0[3], bf0[0], bf0[1], bf0[2], bf0[3], q);
                o0.y = quant4<true>(d0[4], d0[5], d0[6], d0[7], bf0[4], bf0[5], bf0[6], bf0[7], q);
                o0.z = quant4<true>(d0[8], d0[9], d0[10], d0[11], bf0[8], bf0[9], bf0[10], bf0[11], q);
                o0.w = quant4<true>(d0[12], d0[13], d0[14], d0[15], bf0[12], bf0[13], bf0[14], bf0[15], q);
                o1.x = quant4<true>(d1[0], d1[1], d1[2], d1[3], bf1[0], bf1[1], bf1[2], bf1[3], q);
                o1.y = quant4<true>(d1[4], d1[5], d1[6], d1[7], bf1[4], bf1[5], bf1[6], bf1[7], q);
                o1.z = quant4<true>(d1[8], d1[9], d1[10], d1[11], bf1[8], bf1[9], bf1[10], bf1[11], q);
                o1.w = quant4<true>(d1[12], d1[13], d1[14], d1[15], bf1[12], bf1[13], bf1[14], bf1[15], q);
            }
            if (!inside) {  // zero padding of the C2 input: C1 output does not exist outside the frame
                o0 = v4i{0, 0, 0, 0};
                o1 = v4i{0, 0, 0, 0};
            }
            if (qi < NPIX) {
                unsigned char *px = sC1 + (i * C1C + j) * 64;
                *reinterpret_cast<v4i *>(px + swz(h, j)) = o0;      // channels 16h .. 16h+15
                *reinterpret_cast<v4i *>(px + swz(2 + h, j)) = o1;  // channels 32+16h .. 32+16h+15
            }
        }
    }
    __syncthreads();

    // ---- phase 3: C2_1 (3x3, 64 -> 32) and C2_2 (5x5, 64 -> 16) on the TH x TW tile, 16x16x64
    // MFMA.  Wave = 8 output rows x 16 columns x 48 channels.  LDS row = output row + r, LDS
    // column = output column + s (the 3x3 taps are r-1, s-1 of the 5x5 window). ----
    {
        const int c = lane & 15, g = lane >> 4;
        const int cb = (wv & 1) * 16, R0 = (wv >> 1) * 8;
        v4i acc21[2][8], acc22[8];
        {
            v4i b0 = {0, 0, 0, 0}, b1 = {0, 0, 0, 0}, b2 = {0, 0, 0, 0};
            if constexpr (!EXACT) {
                b0 = *reinterpret_cast<const v4i *>(a.bias_i + BIAS_OFF[1] + 4 * g);
                b1 = *reinterpret_cast<const v4i *>(a.bias_i + BIAS_OFF[1] + 16 + 4 * g);
                b2 = *reinterpret_cast<const v4i *>(a.bias_i + BIAS_OFF[2] + 4 * g);
            }
#pragma unroll
            for (int y = 0; y < 8; ++y) {
                acc21[0][y] = b0;
                acc21[1][y] = b1;
                acc22[y] = b2;
            }
        }
        const unsigned char *rowbase = sC1 + (R0 * C1C + cb + c) * 64;
        // The A fragments of kernel column s+1 are requested from L2 before the MFMAs of column s
        // are issued (the sched_barrier pins the requests there), so their latency is covered by
        // ~40-90 MFMAs instead of being paid fragment by fragment.
        C2Weights cur = w0, nxt;
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            const bool inner_s = s >= 1 && s <= 3;
            if (s < 4) load_c2_weights(w22, w21, s + 1, nxt);
            __builtin_amdgcn_sched_barrier(0);
            const unsigned char *colbase = rowbase + s * 64 + swz(g, cb + c + s);
#pragma unroll
            for (int yy = 0; yy < 12; ++yy) {
                const v4i B = *reinterpret_cast<const v4i *>(colbase + yy * (C1C * 64));
#pragma unroll
                for (int r = 0; r < 5; ++r) {
                    const int y = yy - r;
                    if (y >= 0 && y < 8) {
                        acc22[y] = mfma16(cur.f22[r], B, acc22[y]);
                        if (inner_s && r >= 1 && r <= 3) {
                            acc21[0][y] = mfma16(cur.f21[r - 1][0], B, acc21[0][y]);
                            acc21[1][y] = mfma16(cur.f21[r - 1][1], B, acc21[1][y]);
                        }
                    }
                }
            }
            if (s < 4) cur = nxt;
        }
        // epilogue: lane (pixel c, g) owns channels 4g..4g+3 of each 16-channel tile = 12
        // consecutive bytes at position 12g of the pixel in the internal channel order.
        mfma_drain();
        const Requant q21 = a.rq[1], q22 = a.rq[2];
        float bf[12];
        if constexpr (EXACT) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                bf[i] = a.bias_f[BIAS_OFF[1] + 4 * g + i];
                bf[4 + i] = a.bias_f[BIAS_OFF[1] + 16 + 4 * g + i];
                bf[8 + i] = a.bias_f[BIAS_OFF[2] + 4 * g + i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 12; ++i) bf[i] = 0.f;
        }
        const int gx = x0 + cb + c;
#pragma unroll
        for (int y = 0; y < 8; ++y) {
            const int gy = y0 + R0 + y;
            const unsigned d0 = quant4<EXACT>(acc21[0][y][0], acc21[0][y][1], acc21[0][y][2], acc21[0][y][3], bf[0], bf[1],
                                              bf[2], bf[3], q21);
            const unsigned d1 = quant4<EXACT>(acc21[1][y][0], acc21[1][y][1], acc21[1][y][2], acc21[1][y][3], bf[4], bf[5],
                                              bf[6], bf[7], q21);
            const unsigned d2 = quant4<EXACT>(acc22[y][0], acc22[y][1], acc22[y][2], acc22[y][3], bf[8], bf[9], bf[10],
                                              bf[11], q22);
            if (gy < H && gx < W) {
                unsigned *dst = reinterpret_cast<unsigned *>(a.conc1 + ((size_t)gy * W + gx) * 48 + 12 * g);
                dst[0] = d0;
                dst[1] = d1;
                dst[2] = d2;
            }
        }
    }
}

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
    __syncthreads();

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
    __syncthreads();

    // ---- phase 2: C4 (3x3, 48 -> 1) + applyRes_y.  The three kernel rows r are the MFMA rows
    // 0..2: D_yy[r][pixel] = sum_s W4[r][s] . X[yy][pixel+s], and out[y] = D_y[0] + D_{y+1}[1] +
    // D_{y+2}[2] stays inside one lane.  Wave = 9 output rows x 16 columns. ----
    {
        const int cb = (wv & 1) * 16, O0 = (wv >> 1) * 9;
        const unsigned char *rowbase = sC3 + (O0 * C3C + cb + c) * PX + gk * 16;
        int out[9];
#pragma unroll
        for (int y = 0; y < 9; ++y) out[y] = 0;
#pragma unroll
        for (int yy = 0; yy < 11; ++yy) {
            v4i D = {0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const v4i B = *reinterpret_cast<const v4i *>(rowbase + (yy * C3C + s) * PX);
                D = mfma16(f4[s], B, D);
            }
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int y = yy - r;
                if (y >= 0 && y < 9) out[y] += D[r];
            }
        }
        const int ox = cb + c;
        const int gx = x0 + ox;
        if (g == 0 && ox < TW && gx < W) {
            const float b4 = a.bias_f[BIAS_OFF[5]];
            const int mul = a.rq[5].mul, shift = a.rq[5].shift;
#pragma unroll
            for (int y = 0; y < 9; ++y) {
                const int gy = y0 + O0 + y;
                if (gy < H) {
                    const size_t p = (size_t)gy * W + gx;
                    const float u = (float)out[y] + b4;  // C4.u, inference/cnn.cu:148-155
                    int temp = (int)u;                   // applyRes_GPU_y, inference/cnn.cu:515-520
                    temp = (int)((unsigned)temp * (unsigned)mul + (1u << (shift - 1))) >> shift;
                    const short rec = (short)((int)a.y[p] + temp);
                    a.rec[p] = rec > 255 ? (uint8_t)255 : rec < 0 ? (uint8_t)0 : (uint8_t)rec;
                }
            }
        }
    }
}

hipError_t launch_fused_c1c2(const FusedArgs &a, bool exact, hipStream_t s)
{
    dim3 grid((a.W + ka::TW - 1) / ka::TW, (a.H + ka::TH - 1) / ka::TH);
    if (exact) hipLaunchKernelGGL(k_c1c2<true>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_c1c2<false>, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_fused_c3c4(const FusedArgs &a, bool exact, hipStream_t s)
{
    dim3 grid((a.W + kb::TW - 1) / kb::TW, (a.H + kb::TH - 1) / kb::TH);
    if (exact) hipLaunchKernelGGL(k_c3c4<true>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_c3c4<false>, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace qcnn
