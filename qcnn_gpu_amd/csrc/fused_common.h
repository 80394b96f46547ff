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
#pragma once
#include "qcnn_internal.h"
#include <utility>
#include <type_traits>

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
// The clamp is written as the median identity max(min(x, 0), min(max(x, 0), hi)), which the
// compiler selects as one v_med3_i32 (a plain min(max(x, 0), hi) with a non-constant bound becomes
// two instructions).
__device__ __forceinline__ int med3i(int x, int hi)
{
    return max(min(x, 0), min(max(x, 0), hi));
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

// Requantises two consecutive channels (accumulators a0, a1; float biases only used by the EXACT
// variant) into the low half of a dword; join2 makes a dword of four channels from two halves.
template <bool EXACT>
__device__ __forceinline__ unsigned quant2(int a0, int a1, float b0, float b1, const Requant &q)
{
    if constexpr (EXACT)
        return __builtin_amdgcn_perm((unsigned)requant_float(a1, b1, q), (unsigned)requant_float(a0, b0, q), 0x0c0c0400u);
    else
        return __builtin_amdgcn_perm(requant_prod(a1, q), requant_prod(a0, q), q.sel2);
}
__device__ __forceinline__ unsigned join2(unsigned hi, unsigned lo) { return __builtin_amdgcn_perm(hi, lo, 0x05040100u); }

// Requantises four consecutive channels and packs them into one dword (the store-to-char
// truncation of the reference).
template <bool EXACT>
__device__ __forceinline__ unsigned quant4(int a0, int a1, int a2, int a3, float b0, float b1, float b2, float b3,
                                           const Requant &q)
{
    return join2(quant2<EXACT>(a2, a3, b2, b3, q), quant2<EXACT>(a0, a1, b0, b1, q));
}

// Calls f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, N-1>) in order: a loop
// whose index is a constant expression in the body.
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>)
{
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    static_for_impl(static_cast<F &&>(f), std::make_integer_sequence<int, N>{});
}

// The same over LO .. HI-1.
template <int LO, typename F, int... I>
__device__ __forceinline__ void static_range_impl(F &&f, std::integer_sequence<int, I...>)
{
    (f(std::integral_constant<int, LO + I>{}), ...);
}
template <int LO, int HI, typename F>
__device__ __forceinline__ void static_range(F &&f)
{
    if constexpr (HI > LO) static_range_impl<LO>(static_cast<F &&>(f), std::make_integer_sequence<int, HI - LO>{});
}

// Byte offset of 16-byte channel chunk `chunk` (0..3) inside the 64-byte pixel at LDS column
// `col`: the chunk position is rotated by col/2.  With it the 16-lane groups of a ds_read_b128
// B-fragment read (16 consecutive pixels x 4 chunks) hit 16 distinct 16-byte bank slots, and the
// 8-lane groups of the ds_write_b128 that fills the tile (8 consecutive pixels, one chunk) hit 8
// distinct slots of the 128-byte write window.
__device__ __forceinline__ int swz(int chunk, int col) { return ((chunk + (col >> 1)) & 3) * 16; }


// Walks the tiles of one persistent workgroup.  Round n hands workgroup `first` (of `stride`) the
// tile n * stride + (first + n * rot) % stride of the ntiles = nframes * tiles_y * tiles_x tiles
// of a launch; the position inside a round rotates by `rot` per round (0: fixed).  The frame /
// tile row / tile column of the tile are tracked with carries, so a step costs a handful of scalar
// adds instead of three integer divisions.
struct TilePos {
    int f, ty, tx;
    bool valid;
};
struct TileWalker {
    int tiles_x, tiles_y, ntiles, stride, rot;
    int t, r;                    // current tile, its position inside its round
    int d[2][3];                 // (frames, tile rows, tile columns) of the two possible steps
    TilePos pos;
    __device__ __forceinline__ void split(int delta, int *o) const
    {
        const int tpf = tiles_x * tiles_y;
        o[0] = delta / tpf;
        const int rem = delta - o[0] * tpf;
        o[1] = rem / tiles_x;
        o[2] = rem - o[1] * tiles_x;
    }
    __device__ __forceinline__ void init(int tiles_x_, int tiles_y_, int nframes, int first, int stride_, int rot_)
    {
        tiles_x = tiles_x_, tiles_y = tiles_y_, stride = stride_;
        ntiles = tiles_x * tiles_y * nframes;
        rot = rot_ % stride;
        t = first, r = first;
        int o[3];
        split(first, o);
        pos.f = o[0], pos.ty = o[1], pos.tx = o[2];
        pos.valid = first < ntiles;
        split(stride + rot, d[0]);  // next round, position moved up by rot
        split(rot, d[1]);           // next round, position wrapped: (stride + rot) - stride
    }
    __device__ __forceinline__ void next()  // to the tile of the next round
    {
        r += rot;
        const int w = r >= stride ? 1 : 0;
        if (w) r -= stride;
        t += w ? rot : stride + rot;
        pos.tx += d[w][2];
        if (pos.tx >= tiles_x) {
            pos.tx -= tiles_x;
            ++pos.ty;
        }
        pos.ty += d[w][1];
        if (pos.ty >= tiles_y) {
            pos.ty -= tiles_y;
            ++pos.f;
        }
        pos.f += d[w][0];
        pos.valid = t < ntiles;
    }
};

// Diagnostic build only (-DQCNN_STAMPS): per-wave cycle stamps of workgroup 0, written to a debug
// buffer that nothing else reads.  Never compiled into the shipped library.
#ifdef QCNN_STAMPS
#define QCNN_STAMP_SLOTS 128
__device__ __forceinline__ void stamp(unsigned long long *dbg, int blk, int wv, int &idx)
{
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    if (dbg && (int)(blockIdx.x + blockIdx.y * gridDim.x) == blk && (threadIdx.x & 63) == 0 && idx < QCNN_STAMP_SLOTS)
        dbg[wv * QCNN_STAMP_SLOTS + idx] = t;
    ++idx;
    __builtin_amdgcn_sched_barrier(0);
}
#ifndef STAMP_ROW
#define STAMP_ROW wv
#endif
#define STAMP() stamp(a.dbg, a.dbg_block, STAMP_ROW, stamp_idx)
#define STAMP_INIT() int stamp_idx = 0
#else
#define STAMP() do { } while (0)
#define STAMP_INIT() do { } while (0)
#endif

}  // namespace qcnn
