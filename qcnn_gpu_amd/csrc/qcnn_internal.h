// Internal declarations shared by the engine (host) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "qcnn.h"

namespace qcnn {

// Requantiser constants of one conv layer, derived once from (blu, mul, shift) the way
// CHW2CHW_VECT_C_QUANT_BLU does (inference/mat.cu:268): bias = (1 << (shift-1)) / mul.
struct Requant {
    int blu;
    int mul;
    int shift;
    int bias;   // (1 << (shift-1)) / mul
    int cmul;   // bias * mul  (so that (v + bias) * mul == v * mul + cmul, mod 2^32)
    float bluf; // (float)blu, the value the reference's float compare sees
    // integer fast path: mul shifted left by 24 - shift, so that the quotient is byte 3 of the 32-bit
    // value (v + bias) * mulk (0 when shift > 24: the fast path is then not used)
    int mulk;
};

// ---- fused kernel geometry -------------------------------------------------------------
// Kernel A: Y -> C1 -> (C2_1 || C2_2) -> conc1 in HBM.  Persistent 768-thread workgroups walk
// 20 x 32 output tiles (the tallest that two C1 tiles, the weights and the Y tiles leave room for in 160 KiB
// of LDS); four producer waves fill a double-buffered C1 tile, eight consumer waves
// run the C2 stage on the other buffer.
#ifndef QCNN_A_TH
#define QCNN_A_TH 20
#endif
namespace ka {
constexpr int TH = QCNN_A_TH, TW = 32;        // TH: a multiple of 4 (four producers, two row blocks of consumers)
static_assert(TH % 4 == 0 && TH >= 8, "tile height");
constexpr int C1R = TH + 4, C1C = TW + 4;  // C1 output region held in LDS (halo 2)
constexpr int PROD_ROWS = C1R / 4;         // C1 region rows per producer wave (cold tile: all 20 rows computed)
constexpr int WARM_ROWS = TH / 4;          // the same for a warm tile (rows 4..19; rows 0..3 copied from the tile above)
constexpr int YROWS = PROD_ROWS + 4;       // Y rows a producer needs for its C1 rows (halo 2 + 2)
constexpr int YWR = YROWS + 1, YWC = TW + 8;  // private Y tile of one producer (+ 1 slack row)
constexpr int YWS = YWC + 8;               // row stride of one LDS copy of it: 4 bytes lead-in, 4 tail
// bytes of one copy; a producer keeps 4, shifted by 0..3 bytes.  Pixel j of a 32-pixel group reads copy j & 3 at dword
// (j >> 2) of its row: with the copies 8 dwords apart modulo the 64 banks the 32 lanes hit 32 different banks.
constexpr int ycopy_bytes(int raw)
{
    int n = (raw + 15) / 16 * 16;
    while (n % 256 != 32 && n % 256 != 224) n += 16;
    return n;
}
constexpr int YCOPY = ycopy_bytes(YWR * YWS);
constexpr int SC1_BYTES = C1R * C1C * 64;
constexpr int SYW_BYTES = 4 * YCOPY;
constexpr int SW_BYTES = 43 * 1024;        // the 25 + 18 C2 A fragments
constexpr int FLAG_BYTES = 16;             // hand-off counters of an experiment build of kernel A (unused otherwise)
constexpr int LDS_BYTES = 2 * SC1_BYTES + 4 * SYW_BYTES + SW_BYTES + FLAG_BYTES;
static_assert(SYW_BYTES % 16 == 0 && YWS % 4 == 0, "private Y tiles are read as aligned dwords");
static_assert(LDS_BYTES <= 160 * 1024, "kernel A exceeds the 160 KiB of LDS of one CU");
}  // namespace ka

// Kernel B: conc1 -> (C3_1 || C3_2) -> C4 -> residual add, one 18 x 30 output tile per workgroup.
namespace kb {
constexpr int TH = 18, TW = 30;
constexpr int C3R = TH + 2, C3C = 32;   // C3 output region held in LDS (halo 1)
constexpr int INR = TH + 4, INC = 34;   // conc1 input region (halo 2)
constexpr int PX = 48;                  // bytes per pixel of the conc2 tile (and of conc1 in HBM)
constexpr int SIN_BYTES = INR * INC * PX + 64;    // conc1 tile, dense (+ slack: k-group 3 reads 16 bytes past a pixel)
constexpr int SC3_BYTES = C3R * C3C * PX + 256;  // + slack: edge lanes read 2 pixels past a row
constexpr int LDS_BYTES = 2 * SIN_BYTES + 2 * SC3_BYTES;  // both tiles double-buffered
static_assert(LDS_BYTES <= 160 * 1024, "kernel B exceeds the 160 KiB of LDS of one CU");
}  // namespace kb

// Kernel B, streaming form: every wave walks a strip of 14 output columns down a band of rows, see
// kernels_fused_s.hip.  QCNN_S_SHARED = 0 (shipped): four independent waves per workgroup with a ring of conc1
// rows each (18 columns per strip).  1 (experiment, measured slower): the QCNN_S_WAVES waves of a workgroup walk
// neighbouring strips and share one ring (each wave fetches its share of a row, the four columns two
// neighbouring strips have in common are fetched once; one barrier per R rows): 14 % fewer bytes from HBM, yet
// 29.1 against 26.8 us per 1080p frame with 4 or 12 waves per workgroup (28.4 with the barriers taken out), 35.5
// with 6 (two waves more on two of the four SIMDs).
#ifndef QCNN_S_SHARED
#define QCNN_S_SHARED 0
#endif
namespace ks {
constexpr bool SHARED = QCNN_S_SHARED != 0;
constexpr int SW = 14;                        // output columns of a strip (C3 region 16, conc1 18 columns)
constexpr int CW = SW + 4;                    // conc1 columns a strip reads (halo 2)
constexpr int R = 6;                          // conc1 rows per step of the unrolled row loop
#ifndef QCNN_S_WAVES
#define QCNN_S_WAVES 4
#endif
constexpr int WAVES = SHARED ? QCNN_S_WAVES : 4;  // waves per workgroup
constexpr int RING_COLS = SHARED ? WAVES * SW + 4 : CW;   // conc1 columns of one ring row
constexpr int ROW_BYTES = RING_COLS * 48;     // one ring row: contiguous bytes of the HBM image (864 / 4224)
constexpr int ROW_PIECES = ROW_BYTES / 16 / (SHARED ? WAVES : 1);  // 16-byte pieces of a row one wave fetches (54 / 44)
constexpr int GROUPS = SHARED ? 3 : 2;        // groups of R row slots in a ring
constexpr int GROUP_BYTES = R * ROW_BYTES;        // R rows
constexpr int RING_BYTES = GROUPS * GROUP_BYTES + 64;  // one conc1 ring (+ slack: a fragment read runs 16 bytes past its pixel)
constexpr int YSLOT = SW * 4;                     // Y pixels of one output row of a strip, a dword each
constexpr int YRING_BYTES = 2 * R * YSLOT;        // Y ring of one wave: 2 R slots
constexpr int WG_LDS = (SHARED ? RING_BYTES : WAVES * RING_BYTES) + WAVES * YRING_BYTES;
static_assert(ROW_BYTES % (16 * (SHARED ? WAVES : 1)) == 0 && ROW_PIECES <= 64, "a wave fetches its share of a row with one instruction");
static_assert((12 / WAVES) * WG_LDS <= 160 * 1024, "twelve waves per CU");
}  // namespace ks

// The conc1 hand-off image in HBM: int8 [H + 4][W + 4][48] per frame, the frame at rows 2..H+1, columns
// 2..W+1, inside a border of zeros that kernel A never writes (the zero padding of C3), plus 1 KiB of
// slack that kernel B may read (and ignores) past the last row: a strip at the right edge runs up to
// 14 pixels into the next row.
__host__ __device__ inline int conc_wp(int W) { return W + 4; }
__host__ __device__ inline size_t conc_bytes(int H, int W) { return (size_t)(H + 4) * conc_wp(W) * 48 + 1024; }

// Offsets (bytes) of the packed MFMA A-operand fragments inside one device buffer.
// One fragment = 64 lanes x 16 bytes = 1 KiB, stored in lane order.
struct PackedOffsets {
    static constexpr int C1W = 0;                       // 2 fragments (32x32x32) [channel tile]
    static constexpr int C2W22 = C1W + 2 * 1024;        // 25 fragments [s][r]
    static constexpr int C2W21 = C2W22 + 25 * 1024;     // 18 fragments [s'][r'][t]
    static constexpr int C3W31 = C2W21 + 18 * 1024;     // 9 fragments [s][r]
    static constexpr int C3W32 = C3W31 + 9 * 1024;      // 2 fragments [t]
    static constexpr int C4W = C3W32 + 2 * 1024;        // 3 fragments [s]
    static constexpr int C4W9 = C4W + 3 * 1024;         // 1 fragment: all nine taps as MFMA rows 4s + r
    static constexpr int TOTAL = C4W9 + 1024;
};

// Byte offset of frame f's conc1 image.  -DQCNN_ABL_ONECONC (a measurement build whose results are garbage) puts every
// frame of a launch into the first image, which then stays in the 256 MiB last-level cache between its writer and its reader.
#ifdef QCNN_ABL_ONECONC
#define QCNN_CONC_OFF(a, f) ((size_t)0)
#else
#define QCNN_CONC_OFF(a, f) ((size_t)(f) * (a).conc_stride)
#endif

// Arguments of the fused kernels (passed by value).
struct FusedArgs {
    const uint8_t *y;        // I1.x       uint8 [H][W]
    uint8_t *rec;            // I1.x_rec   uint8 [H][W]
    int8_t *conc1;           // [nframes] images of conc_bytes(H, W), int8, internal channel order (see pos_of_refch())
    int nframes;             // frames of this launch; frame f lives at y/rec + f*frame_stride
    size_t frame_stride;     // bytes between consecutive frames of y and of rec
    size_t conc_stride;      // bytes between consecutive frames of conc1
    const int8_t *wpack;     // packed fragments, PackedOffsets
    const int *bias_i;       // int32 biases, 6 layers concatenated: 64,32,16,16,32,1 (=161); layers 0..4 with the
                             // requantiser's rounding offset (Requant::bias) added: the integer path starts its
                             // accumulators from them
    const float *bias_f;     // same, converted to float as CovLayer::load_static_para does
    int H, W;
    int ncu;                 // compute units of the device (grid size of the persistent kernel)
    Requant rq[QCNN_NUM_LAYERS];
    unsigned long long *dbg; // cycle stamps of a diagnostic build, else null
    int dbg_block;           // linear workgroup index that writes the stamps
};

// Offsets of each layer's biases inside bias_i / bias_f.
constexpr int BIAS_OFF[QCNN_NUM_LAYERS] = {0, 64, 96, 112, 128, 160};

// Internal channel order of the 48-channel images conc1 / conc2: the byte at position `pos` of a
// pixel holds reference channel refch(pos).  It makes every lane of a 16x16 MFMA output tile own
// 12 consecutive bytes.  pos = 12*g + 4*t + i  <->  channel 16*t + 4*g + i.
__host__ __device__ inline int refch_of_pos(int pos)
{
    const int g = pos / 12, t = (pos % 12) / 4, i = pos % 4;
    return 16 * t + 4 * g + i;
}
__host__ __device__ inline int pos_of_refch(int ch)
{
    const int t = ch / 16, g = (ch % 16) / 4, i = ch % 4;
    return 12 * g + 4 * t + i;
}

// Kernel B reads its B fragments from LDS tiles with dense 48-byte pixels.  A ds_read_b128 is
// served in groups of 16 lanes that mix 8 lanes of one MFMA k-group with 8 lanes of the next, and
// it is conflict-free when the group covers 16 distinct 16-byte bank slots.  Pixel p, chunk k sits
// at slot 3p + k: with k-groups (0,1,2,3) reading chunks (0,2,1,3) the two k-groups of a lane group
// are 2 slots apart, and with the lanes 4..11 of a k-group on even pixels and the others on odd
// pixels each half covers one parity class of slots.
__host__ __device__ inline int kb_chunk_of_kgroup(int g) { return g == 1 ? 2 : g == 2 ? 1 : g; }
__host__ __device__ inline int kb_pixel_of_lane(int c)  // c = lane & 15 -> column inside the wave's 16-pixel tile
{
    return (c >= 4 && c < 12) ? 2 * (c - 4) : c < 4 ? 2 * c + 1 : 2 * (c - 12) + 9;
}

// ---- dynamic-quantisation pass ----------------------------------------------------------------
// step_parameters of inference/qvrcnn.cuh:11-15: stepw kept in descending, stepy in ascending order.
struct DynSteps {
    int stepw[4];
    int stepy[4];
};
// Device-resident bookkeeping of one pass of qvrcnn::forward (what the reference keeps in host members:
// steps, CovLayer::step_w / step_y / max_u, and the device arrays CovLayer::b / b_adj).
struct DynState {
    DynSteps steps;
    int step_w[QCNN_NUM_LAYERS];        // from the model file (CovLayer::load_para, inference/cnn.cu:78)
    int step_y[QCNN_NUM_LAYERS];
    int max_u[QCNN_NUM_LAYERS];
    unsigned maxbits[QCNN_NUM_LAYERS];  // largest |u| of each layer as float bits
    float b[161];                       // biases as loaded, layers concatenated (BIAS_OFF)
    float b_adj[161];                   // adjustBasic output
};

// ---- kernel launchers (defined in the .hip files) ----------------------------------------
hipError_t launch_ppro(const void *x, void *x_ppro, size_t n, hipStream_t s);
// relu != 0 applies CovLayer::activate (inference/cnn.cu:161-166) to the result before it is stored
hipError_t launch_conv_static(const void *x, const void *w, const void *b, void *u, int H, int W, int cin,
                              int cout, int ks, int relu, hipStream_t s);
// The same convolution on the matrix cores (kernels_conv_mfma.hip) for the layer shapes of the network,
// with optional fused passes: u (float NCHW), q (divide-quantised int8 NCHW_VECT_C), maxbits (largest |u|).
bool conv_mfma_supported(int cin, int cout, int ks);
bool conv_direct_only();  // QCNN_CONV_DIRECT=1: keep the layer-level convolutions on the v_dot4 kernel
hipError_t launch_conv_mfma(const void *x, const void *w, const void *b, void *u, void *q, unsigned *maxbits,
                            const int *divisor_dev, int divisor, int relu, int H, int W, int cin, int cout, int ks,
                            hipStream_t s);
// Both branches of a layer on one staged input tile (kernels_conv_mfma.hip): the results of two launch_conv_mfma calls.
struct ConvPairBranch {
    const void *w, *b;
    void *u, *q;
    unsigned *maxbits;
    const int *divisor_dev;
    int divisor, cout, ks;
};
bool conv_pair_supported(int cin, int cout_a, int ks_a, int cout_b, int ks_b);
hipError_t launch_conv_mfma_pair(const void *x, int cin, int H, int W, int relu, const ConvPairBranch &ba, const ConvPairBranch &bb,
                                 hipStream_t s);
hipError_t launch_absmax(const void *u, size_t n, unsigned *maxbits, hipStream_t s);
hipError_t launch_relu(void *u, size_t n, hipStream_t s);
hipError_t launch_quant_div(const void *u, void *out, size_t hw, int channels, int divisor, const int *divisor_dev,
                            hipStream_t s);
hipError_t launch_mul_shift(const void *in, void *out, size_t n, int mul, int shift, hipStream_t s);
hipError_t launch_adjust_basic(const DynSteps &st, const void *b, void *b_adj, int count, int n, hipStream_t s);
hipError_t launch_adjust_output(const DynSteps &st, const DynSteps *st_dev, const void *o, void *o_adj, int n,
                                size_t num, hipStream_t s);
hipError_t launch_adjust_output_static(const void *o, void *o_adj, int mul, int shift, size_t num, hipStream_t s);
hipError_t launch_apply_res(const void *x, const void *res, void *rec, size_t n, hipStream_t s);
hipError_t launch_dyn_book(DynState *st, int stage, hipStream_t s);
void concat_steps(float max1, float max2, int sw1, int sw2, int *stepy1, int *stepy2);
hipError_t launch_quant_blu(const void *u, void *out, size_t hw, int channels, const Requant &rq, hipStream_t s);
hipError_t launch_apply_res_y(const void *x, void *rec, const void *res, int mul, int shift, size_t n,
                              hipStream_t s);
hipError_t launch_sse(const void *a, const void *b, size_t n, unsigned long long *dev_sum, hipStream_t s);
hipError_t launch_sse_acc(const void *a, const void *b, size_t n, unsigned long long *dev_sum, hipStream_t s);
hipError_t launch_sse_clip(const void *in, const void *out, const void *ori, size_t n, int nframes,
                           unsigned long long *sum_before, unsigned long long *sum_after, hipStream_t s);
hipError_t launch_conc_to_ref(const void *conc_internal, void *conc_ref, int H, int W, hipStream_t s);
hipError_t launch_conc_zero_border(void *conc_internal, int nframes, int H, int W, hipStream_t s);
hipError_t launch_fused_c1c2(const FusedArgs &a, int mode, hipStream_t s);  // 0 integer, 1 float pipeline, 2 clamped integer
hipError_t launch_fused_c3c4(const FusedArgs &a, bool exact, hipStream_t s);         // tiled form (kernels_fused_b.hip)
hipError_t launch_fused_c3c4_stream(const FusedArgs &a, bool exact, hipStream_t s);  // streaming form (kernels_fused_s.hip)
hipError_t launch_fused_c3c4_stream2(const FusedArgs &a, bool exact, hipStream_t s);  // the same, two column tiles per wave (kernels_fused_s2.hip)

// float32 configuration (kernels_float.hip)
bool conv_float_supported(int cin, int cout, int ks);
hipError_t launch_conv_float(const void *x, const void *w, const void *b, void *u, int H, int W, int cin, int cout,
                             int ks, int relu, hipStream_t s);
hipError_t launch_ppro_float(const void *x, void *xf, size_t n, hipStream_t s);
hipError_t launch_apply_res_float(const void *x, const void *res, void *rec, size_t n, hipStream_t s);

Requant make_requant(int blu, int mul, int shift);

}  // namespace qcnn
