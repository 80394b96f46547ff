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
// Each kernel exists in an integer and in a float-pipeline instantiation.  The integer one is used when the
// host has proved, from the loaded weights, that no accumulator or bias can reach 2^24: then the reference's
// float32 pipeline (float(acc) + float(b), float compares, (int) truncation) is identical to integer
// arithmetic and the bias is folded into the MFMA accumulator.  The float-pipeline one (EXACT) replays the
// float32 pipeline literally and is bit-exact for every possible model.  Kernel A has a third instantiation
// for models whose C2 accumulator bounds alone fail the proof (requant12_wide): integers with one clamp more.
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

// Literal float32 pipeline of ConvForward_static + CHW2CHW_VECT_C_QUANT_BLU.
__device__ __forceinline__ int requant_float(int acc, float bf, const Requant &q)
{
    const float t = (float)acc + bf;
    if (t > q.bluf) return 127;
    if (t < 0.f) return 0;
    return (int)((unsigned)((int)t + q.bias) * (unsigned)q.mul) >> q.shift;
}

// Integer requantiser of four consecutive channels, valid when the host-side proof holds (see
// engine.hip, prove_integer_path).  s = acc + bias + q.bias: what the reference's float u holds plus
// the requantiser's rounding offset, which the host folds into the bias the accumulators start
// from; it fits 24 signed bits.  mul is pre-shifted on the host by 24 - shift, so the quotient
// (s * mul) >> shift is byte 3 of t = s * mulk.  Two instructions per channel:
//   t = saturate_i32(s * mulk)               v_mad_i32_i24 t, s, mulk, 0 clamp  (one VGPR operand)
//       u > blu : t >= requant(blu) << 24 = 127 << 24, and saturation keeps byte 3 at 127;
//       u < 0   : t < q.bias * mulk <= 2^23, so byte 3 is 0 or negative
//   byte j of the dword = max(sext(t[31:24]), 0)   v_max_i32 with sub-dword source and destination
// which is both clip arms of the reference and its store-to-char truncation.  A sub-dword write
// must not be followed directly by a vector instruction that reads the register (one wait state):
// inside the block the next channel's multiply separates the writes, an s_nop ends it.
__device__ __forceinline__ unsigned requant4_int(int s0, int s1, int s2, int s3, const Requant &q)
{
    unsigned p;
    int t;
    asm("v_mad_i32_i24 %1, %2, %6, 0 clamp\n\t"
        "v_max_i32_sdwa %0, sext(%1), 0 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n\t"
        "v_mad_i32_i24 %1, %3, %6, 0 clamp\n\t"
        "v_max_i32_sdwa %0, sext(%1), 0 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD\n\t"
        "v_mad_i32_i24 %1, %4, %6, 0 clamp\n\t"
        "v_max_i32_sdwa %0, sext(%1), 0 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD\n\t"
        "v_mad_i32_i24 %1, %5, %6, 0 clamp\n\t"
        "v_max_i32_sdwa %0, sext(%1), 0 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD\n\t"
        "s_nop 0"
        : "=&v"(p), "=&v"(t)
        : "v"(s0), "v"(s1), "v"(s2), "v"(s3), "s"(q.mulk));
    return p;
}

// The same for three dwords at once (the 12 channels a lane owns in the C2 and C3 epilogues; the first
// two dwords share a requantiser): the three dependency chains are interleaved, so no instruction
// waits for the one before it.
__device__ __forceinline__ void requant12_int(const v4i &a, const v4i &b, const v4i &c, const Requant &qab, const Requant &qc,
                                              unsigned &pa, unsigned &pb, unsigned &pc)
{
    int ta, tb, tc;
    asm("v_mad_i32_i24 %3, %6, %18, 0 clamp\n\t"
        "v_mad_i32_i24 %4, %10, %18, 0 clamp\n\t"
        "v_mad_i32_i24 %5, %14, %19, 0 clamp\n\t"
        "v_max_i32_sdwa %0, sext(%3), 0 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n\t"
        "v_max_i32_sdwa %1, sext(%4), 0 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n\t"
        "v_max_i32_sdwa %2, sext(%5), 0 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n\t"
        "v_mad_i32_i24 %3, %7, %18, 0 clamp\n\t"
        "v_mad_i32_i24 %4, %11, %18, 0 clamp\n\t"
        "v_mad_i32_i24 %5, %15, %19, 0 clamp\n\t"
        "v_max_i32_sdwa %0, sext(%3), 0 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD\n\t"
        "v_max_i32_sdwa %1, sext(%4), 0 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD\n\t"
        "v_max_i32_sdwa %2, sext(%5), 0 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD\n\t"
        "v_mad_i32_i24 %3, %8, %18, 0 clamp\n\t"
        "v_mad_i32_i24 %4, %12, %18, 0 clamp\n\t"
        "v_mad_i32_i24 %5, %16, %19, 0 clamp\n\t"
        "v_max_i32_sdwa %0, sext(%3), 0 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD\n\t"
        "v_max_i32_sdwa %1, sext(%4), 0 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD\n\t"
        "v_max_i32_sdwa %2, sext(%5), 0 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD\n\t"
        "v_mad_i32_i24 %3, %9, %18, 0 clamp\n\t"
        "v_mad_i32_i24 %4, %13, %18, 0 clamp\n\t"
        "v_mad_i32_i24 %5, %17, %19, 0 clamp\n\t"
        "v_max_i32_sdwa %0, sext(%3), 0 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD\n\t"
        "v_max_i32_sdwa %1, sext(%4), 0 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD\n\t"
        "v_max_i32_sdwa %2, sext(%5), 0 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD\n\t"
        "s_nop 0"
        : "=&v"(pa), "=&v"(pb), "=&v"(pc), "=&v"(ta), "=&v"(tb), "=&v"(tc)
        : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w), "v"(c.x), "v"(c.y), "v"(c.z), "v"(c.w),
          "s"(qab.mulk), "s"(qc.mulk));
}

// The same for two dwords with one requantiser (the 8 C2_1 channels a lane owns when its wave runs C2_1 alone).
__device__ __forceinline__ void requant8_int(const v4i &a, const v4i &b, const Requant &q, unsigned &pa, unsigned &pb)
{
    int ta, tb;
#define QCNN_REQ8_STEP(A, B, SEL, UNUSED)                                                                            \
    "v_mad_i32_i24 %2, " A ", %12, 0 clamp\n\t"                                                                       \
    "v_mad_i32_i24 %3, " B ", %12, 0 clamp\n\t"                                                                       \
    "v_max_i32_sdwa %0, sext(%2), 0 dst_sel:" SEL " dst_unused:" UNUSED " src0_sel:BYTE_3 src1_sel:DWORD\n\t"         \
    "v_max_i32_sdwa %1, sext(%3), 0 dst_sel:" SEL " dst_unused:" UNUSED " src0_sel:BYTE_3 src1_sel:DWORD\n\t"
    asm(QCNN_REQ8_STEP("%4", "%8", "BYTE_0", "UNUSED_PAD")
        QCNN_REQ8_STEP("%5", "%9", "BYTE_1", "UNUSED_PRESERVE")
        QCNN_REQ8_STEP("%6", "%10", "BYTE_2", "UNUSED_PRESERVE")
        QCNN_REQ8_STEP("%7", "%11", "BYTE_3", "UNUSED_PRESERVE")
        "s_nop 0"
        : "=&v"(pa), "=&v"(pb), "=&v"(ta), "=&v"(tb)
        : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w), "s"(q.mulk));
#undef QCNN_REQ8_STEP
}

// requant12_int for models whose C2 accumulators may leave the signed 24-bit range (the load-time proof of the
// two-instruction form fails on the accumulator bounds only): s is first clamped to [-1, blu + q.bias], three
// instructions per channel.  Every s above blu + q.bias requantises to 127 like blu itself and every negative s
// to 0, and an accumulator beyond 2^24, where the reference's conversion to float rounds, is further than 2^23
// from both thresholds whatever the rounding did (biases are at most 2^23 in magnitude on this path), so the
// result equals the reference's float32 pipeline.  lim = blu + q.bias.
__device__ __forceinline__ void requant12_wide(const v4i &a, const v4i &b, const v4i &c, const Requant &qab, const Requant &qc,
                                               int limab, int limc, unsigned &pa, unsigned &pb, unsigned &pc)
{
    int ta, tb, tc;
#define QCNN_WIDE_STEP(A, B, C, SEL, UNUSED)                                                                     \
    "v_med3_i32 %3, " A ", -1, %20\n\t"                                                                          \
    "v_med3_i32 %4, " B ", -1, %20\n\t"                                                                          \
    "v_med3_i32 %5, " C ", -1, %21\n\t"                                                                          \
    "v_mad_i32_i24 %3, %3, %18, 0 clamp\n\t"                                                                     \
    "v_mad_i32_i24 %4, %4, %18, 0 clamp\n\t"                                                                     \
    "v_mad_i32_i24 %5, %5, %19, 0 clamp\n\t"                                                                     \
    "v_max_i32_sdwa %0, sext(%3), 0 dst_sel:" SEL " dst_unused:" UNUSED " src0_sel:BYTE_3 src1_sel:DWORD\n\t"    \
    "v_max_i32_sdwa %1, sext(%4), 0 dst_sel:" SEL " dst_unused:" UNUSED " src0_sel:BYTE_3 src1_sel:DWORD\n\t"    \
    "v_max_i32_sdwa %2, sext(%5), 0 dst_sel:" SEL " dst_unused:" UNUSED " src0_sel:BYTE_3 src1_sel:DWORD\n\t"
    asm(QCNN_WIDE_STEP("%6", "%10", "%14", "BYTE_0", "UNUSED_PAD")
        QCNN_WIDE_STEP("%7", "%11", "%15", "BYTE_1", "UNUSED_PRESERVE")
        QCNN_WIDE_STEP("%8", "%12", "%16", "BYTE_2", "UNUSED_PRESERVE")
        QCNN_WIDE_STEP("%9", "%13", "%17", "BYTE_3", "UNUSED_PRESERVE")
        "s_nop 0"
        : "=&v"(pa), "=&v"(pb), "=&v"(pc), "=&v"(ta), "=&v"(tb), "=&v"(tc)
        : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w), "v"(c.x), "v"(c.y), "v"(c.z), "v"(c.w),
          "s"(qab.mulk), "s"(qc.mulk), "s"(limab), "s"(limc));
#undef QCNN_WIDE_STEP
}

// requant12_int in four pieces, one per channel index I of the three dwords, so that a caller can place
// matrix instructions between them (two vector instructions per MFMA of the same wave issue almost for
// free).  Piece 0 starts the dwords, the others insert their byte; LAST ends with the wait state a
// sub-dword write needs before a vector instruction reads the register.  The accumulators must have
// been complete for the latency of their last MFMA: inline assembly gets no hazard wait states from
// the compiler.
template <int I, bool LAST>
__device__ __forceinline__ void requant12_piece(int a, int b, int c, const Requant &qab, const Requant &qc, unsigned &pa,
                                                unsigned &pb, unsigned &pc)
{
    int ta, tb, tc;
    if constexpr (I == 0) {
        asm("v_mad_i32_i24 %3, %6, %9, 0 clamp\n\t"
            "v_mad_i32_i24 %4, %7, %9, 0 clamp\n\t"
            "v_mad_i32_i24 %5, %8, %10, 0 clamp\n\t"
            "v_max_i32_sdwa %0, sext(%3), 0 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n\t"
            "v_max_i32_sdwa %1, sext(%4), 0 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n\t"
            "v_max_i32_sdwa %2, sext(%5), 0 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD"
            : "=&v"(pa), "=&v"(pb), "=&v"(pc), "=&v"(ta), "=&v"(tb), "=&v"(tc)
            : "v"(a), "v"(b), "v"(c), "s"(qab.mulk), "s"(qc.mulk));
    } else {
#define QCNN_REQ_PIECE(SEL, TAIL)                                                                                          \
    asm("v_mad_i32_i24 %3, %6, %9, 0 clamp\n\t"                                                                             \
        "v_mad_i32_i24 %4, %7, %9, 0 clamp\n\t"                                                                             \
        "v_mad_i32_i24 %5, %8, %10, 0 clamp\n\t"                                                                            \
        "v_max_i32_sdwa %0, sext(%3), 0 dst_sel:" SEL " dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD\n\t"     \
        "v_max_i32_sdwa %1, sext(%4), 0 dst_sel:" SEL " dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD\n\t"     \
        "v_max_i32_sdwa %2, sext(%5), 0 dst_sel:" SEL " dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD" TAIL     \
        : "+v"(pa), "+v"(pb), "+v"(pc), "=&v"(ta), "=&v"(tb), "=&v"(tc)                                                    \
        : "v"(a), "v"(b), "v"(c), "s"(qab.mulk), "s"(qc.mulk))
        if constexpr (I == 1) QCNN_REQ_PIECE("BYTE_1", "");
        else if constexpr (I == 2) QCNN_REQ_PIECE("BYTE_2", "");
        else if constexpr (LAST) QCNN_REQ_PIECE("BYTE_3", "\n\ts_nop 0");
        else QCNN_REQ_PIECE("BYTE_3", "");
#undef QCNN_REQ_PIECE
    }
}

// One value of the integer requantiser: its two instructions, into byte I of `p` (I = 0 starts the dword).  A caller
// that places one of these behind every MFMA hides the requantiser in the matrix pipe's shadow.  The same rules as
// for requant12_piece hold: the accumulator complete for the latency of its MFMA, and a wait state (any
// instruction) between the write of byte 3 and a vector instruction that reads the dword.
template <int I>
__device__ __forceinline__ void requant1(int a, int mulk, unsigned &p)
{
    int t;
    if constexpr (I == 0) {
        asm("v_mad_i32_i24 %1, %2, %3, 0 clamp\n\t"
            "v_max_i32_sdwa %0, sext(%1), 0 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD"
            : "=&v"(p), "=&v"(t)
            : "v"(a), "s"(mulk));
    } else {
#define QCNN_REQ_ONE(SEL)                                                                                             \
    asm("v_mad_i32_i24 %1, %2, %3, 0 clamp\n\t"                                                                       \
        "v_max_i32_sdwa %0, sext(%1), 0 dst_sel:" SEL " dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD"    \
        : "+v"(p), "=&v"(t)                                                                                           \
        : "v"(a), "s"(mulk))
        if constexpr (I == 1) QCNN_REQ_ONE("BYTE_1");
        else if constexpr (I == 2) QCNN_REQ_ONE("BYTE_2");
        else QCNN_REQ_ONE("BYTE_3");
#undef QCNN_REQ_ONE
    }
}

// The same for the 16 channels of one 32x32 accumulator (four dwords, one requantiser).
__device__ __forceinline__ v4i requant16_int(const v16i &d, const Requant &q)
{
    unsigned p0, p1, p2, p3;
    int t0, t1, t2, t3;
    asm("v_mad_i32_i24 %4, %8, %24, 0 clamp\n\t"
        "v_mad_i32_i24 %5, %12, %24, 0 clamp\n\t"
        "v_mad_i32_i24 %6, %16, %24, 0 clamp\n\t"
        "v_mad_i32_i24 %7, %20, %24, 0 clamp\n\t"
        "v_max_i32_sdwa %0, sext(%4), 0 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n\t"
        "v_max_i32_sdwa %1, sext(%5), 0 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n\t"
        "v_max_i32_sdwa %2, sext(%6), 0 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n\t"
        "v_max_i32_sdwa %3, sext(%7), 0 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n\t"
        "v_mad_i32_i24 %4, %9, %24, 0 clamp\n\t"
        "v_mad_i32_i24 %5, %13, %24, 0 clamp\n\t"
        "v_mad_i32_i24 %6, %17, %24, 0 clamp\n\t"
        "v_mad_i32_i24 %7, %21, %24, 0 clamp\n\t"
        "v_max_i32_sdwa %0, sext(%4), 0 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD\n\t"
        "v_max_i32_sdwa %1, sext(%5), 0 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD\n\t"
        "v_max_i32_sdwa %2, sext(%6), 0 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD\n\t"
        "v_max_i32_sdwa %3, sext(%7), 0 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD\n\t"
        "v_mad_i32_i24 %4, %10, %24, 0 clamp\n\t"
        "v_mad_i32_i24 %5, %14, %24, 0 clamp\n\t"
        "v_mad_i32_i24 %6, %18, %24, 0 clamp\n\t"
        "v_mad_i32_i24 %7, %22, %24, 0 clamp\n\t"
        "v_max_i32_sdwa %0, sext(%4), 0 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD\n\t"
        "v_max_i32_sdwa %1, sext(%5), 0 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD\n\t"
        "v_max_i32_sdwa %2, sext(%6), 0 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD\n\t"
        "v_max_i32_sdwa %3, sext(%7), 0 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD\n\t"
        "v_mad_i32_i24 %4, %11, %24, 0 clamp\n\t"
        "v_mad_i32_i24 %5, %15, %24, 0 clamp\n\t"
        "v_mad_i32_i24 %6, %19, %24, 0 clamp\n\t"
        "v_mad_i32_i24 %7, %23, %24, 0 clamp\n\t"
        "v_max_i32_sdwa %0, sext(%4), 0 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD\n\t"
        "v_max_i32_sdwa %1, sext(%5), 0 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD\n\t"
        "v_max_i32_sdwa %2, sext(%6), 0 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD\n\t"
        "v_max_i32_sdwa %3, sext(%7), 0 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD\n\t"
        "s_nop 0"
        : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]), "v"(d[4]), "v"(d[5]), "v"(d[6]), "v"(d[7]), "v"(d[8]), "v"(d[9]), "v"(d[10]),
          "v"(d[11]), "v"(d[12]), "v"(d[13]), "v"(d[14]), "v"(d[15]), "s"(q.mulk));
    return v4i{(int)p0, (int)p1, (int)p2, (int)p3};
}

// Requantises four consecutive channels (accumulators a0..a3; the float biases are only used by
// the EXACT variant) and packs them into one dword.
template <bool EXACT>
__device__ __forceinline__ unsigned quant4(int a0, int a1, int a2, int a3, float b0, float b1, float b2, float b3,
                                           const Requant &q)
{
    if constexpr (EXACT) {
        const unsigned lo = __builtin_amdgcn_perm((unsigned)requant_float(a1, b1, q), (unsigned)requant_float(a0, b0, q), 0x0c0c0400u);
        const unsigned hi = __builtin_amdgcn_perm((unsigned)requant_float(a3, b3, q), (unsigned)requant_float(a2, b2, q), 0x0c0c0400u);
        return __builtin_amdgcn_perm(hi, lo, 0x05040100u);
    } else {
        return requant4_int(a0, a1, a2, a3, q);
    }
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
