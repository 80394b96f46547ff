/*
 * qcnn.h -- C ABI of the MI355X (gfx950) static-quantised VRCNN inference engine.
 *
 * The reference has no FFI layer: its public interface is the C++ class API of
 * inference/qvrcnn.cuh:25-59 and inference/cnn.cuh:17-100, driven by
 * inference/kernel.cu:74-116.  This header is the boundary underneath that API:
 * plain pointers and sizes, no C++ or framework types.  include/qvrcnn.h puts the
 * reference's classes back on top of it.
 *
 * Conventions
 *   - every function returns 0 on success and a negative QCNN_E_* code on failure;
 *     qcnn_last_error() returns a thread-local message for the last failure.
 *   - "device" pointers are HIP device addresses on the GPU the object was created on.
 *   - activations in the reference layout NCHW_VECT_C are int8 [C/4][H][W][4];
 *     conv outputs `u` are float32 NCHW (active config INT8x4_EXT_CONFIG,
 *     inference/mat.cuh:59-73).
 *   - an engine is not thread-safe; use one engine per host thread / per GPU
 *     (the reference's threading model, inference/qvrcnn.cu:6).
 */
#ifndef QCNN_H
#define QCNN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QCNN_OK 0
#define QCNN_E_ARG (-1)      /* bad argument (null pointer, size, layer index ...)      */
#define QCNN_E_IO (-2)       /* file cannot be opened / short read                      */
#define QCNN_E_FORMAT (-3)   /* model image malformed (size, mul <= 0, shift range)     */
#define QCNN_E_HIP (-4)      /* a HIP runtime call failed                               */
#define QCNN_E_STATE (-5)    /* call order (forward before load ...)                    */
#define QCNN_E_UNSUPPORTED (-6)

#define QCNN_NUM_LAYERS 6 /* C1, C2_1, C2_2, C3_1, C3_2, C4 (inference/qvrcnn.cu:11-18) */

typedef struct qcnn_engine qcnn_t;

const char *qcnn_last_error(void);
const char *qcnn_version(void);
int qcnn_device_count(int *count);
/* PCI bus id of device `gpu` as text ("0000:c1:00.0", NUL-terminated, at most len bytes): tells two processes
 * whether their device numbers mean the same card (the reference reads the device count and ignores it,
 * inference/kernel.cu:42-43,85-86). */
int qcnn_device_pci_bus_id(int gpu, char *buf, int len);

/* ---- network level ------------------------------------------------------------------ */

/* qvrcnn::qvrcnn(gpu_num, batch, channel, height, width), inference/qvrcnn.cu:4-29.
 * Selects the device, creates one stream and allocates every buffer; forward passes
 * allocate nothing.  batch and channel must be 1 (FRAME/CHANNEL, inference/mat.cuh:23-24). */
int qcnn_create(int gpu, int batch, int channel, int height, int width, qcnn_t **out);

/* qvrcnn::~qvrcnn, inference/qvrcnn.cu:331-335. */
int qcnn_destroy(qcnn_t *e);

/* qvrcnn::load_static_para(filename), inference/qvrcnn.cu:47-63: reads the NCHW_VECT_C
 * static model file (six layers of int8 weights, int32 biases, blu/mul/shift), packs the
 * weights into MFMA operand order and uploads them.  May be called again to switch model. */
int qcnn_load_static(qcnn_t *e, const char *path);

/* Same, from a memory image of the file (60 028 bytes). */
int qcnn_load_static_mem(qcnn_t *e, const void *image, size_t bytes);

/* qvrcnn::load_data(input) -> InputLayer::load, inference/cnn.cu:439-443: blocking copy of
 * one Y frame (height*width bytes) from host memory into I1.x. */
int qcnn_upload(qcnn_t *e, const uint8_t *host_y);

/* Same as qcnn_upload but asynchronous on the engine stream (host_y should be pinned). */
int qcnn_upload_async(qcnn_t *e, const uint8_t *host_y);

/* qvrcnn::forward_blu(), inference/qvrcnn.cu:168-242, as two fused gfx950 kernels enqueued
 * on the engine stream (asynchronous; follow with qcnn_sync or qcnn_download). */
int qcnn_forward_blu(qcnn_t *e);

/* The frame loop of inference/kernel.cu:91-97 for frames already resident in HBM: runs
 * qcnn_forward_blu on `nframes` frames read from dev_y + i*stride and written to dev_rec +
 * i*stride (stride in bytes, >= height*width).  Asynchronous on the engine stream. */
int qcnn_forward_blu_batch(qcnn_t *e, const void *dev_y, void *dev_rec, int nframes, size_t stride);

/* qcnn_forward_blu_batch for callers that cannot allocate device memory while they run (code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference,
 * whose forward pass allocates nothing because every buffer is sized in the constructors
 * (inference/qvrcnn.cu:4-29).  Same arguments, same results, same kernels as qcnn_forward_blu_batch; the
 * conc1 hand-off image between the two fused kernels lives in a static device array of
 * QCNN_NOALLOC_SCRATCH_BYTES that the library's code object carries (each GPU the library runs on holds
 * one copy until the process ends; its address is looked up per device in qcnn_create).  The call makes
 * no hipMalloc / hipFree and records no timing events.  A frame needs 48*(height+4)*(width+4) + 1024
 * bytes of it (the image with its zero border): larger frames are rejected with QCNN_E_UNSUPPORTED
 * (1920x1080 fits, 1936x1088 does not); as many frames per launch as fit are grouped.  All engines on one GPU
 * share that GPU's array: calls that overlap in time on one GPU (two engines, two streams) race on it
 * and must be serialised by the caller. */
#define QCNN_NOALLOC_SCRATCH_BYTES (96u << 20)
int qcnn_forward_blu_batch_noalloc(qcnn_t *e, const void *dev_y, void *dev_rec, int nframes, size_t stride);

/* The whole timed region of inference/kernel.cu:89-101 for a clip held in HOST memory: for each of
 * `nframes` frames load_data (H2D), forward_blu, D2H of I1.x_rec.  Frame i is read from
 * host_y + i*stride and written to host_rec + i*stride (stride in bytes, >= height*width).  Runs as
 * a three-stream pipeline over groups of frames (see qcnn_batch_group): the H2D copy of group k+1 and
 * the D2H copy of group k-1 overlap the kernels of group k when the caller's buffers are page-locked
 * (qcnn_host_alloc_pinned, or qcnn_host_register on memory the caller owns); pageable buffers work too, through
 * the runtime's staging copies.  The call itself registers nothing: sharded callers hand overlapping spans of
 * one clip to several engines.  Blocking: returns when host_rec is complete.
 * If host_ori is not NULL (same stride), sse_before[i] / sse_after[i] (either may be NULL) receive the
 * integer sums of squared differences of frame i against the original before and after the net,
 * computed on the device: the numerators of vrcnn_data::psnr / psnr_pf (inference/yuv_data.cpp:87-112). */
int qcnn_forward_blu_clip(qcnn_t *e, const uint8_t *host_y, uint8_t *host_rec, int nframes, size_t stride,
                          const uint8_t *host_ori, uint64_t *sse_before, uint64_t *sse_after);

/* Allocates what qcnn_forward_blu_clip needs for a clip of `nframes` frames (copy streams, device rings, the
 * per-launch scratch; with_sse != 0: the rings of the original too), so that a caller can keep set-up out of
 * its timed region as inference/kernel.cu:86-90 does (every cudaMalloc of the reference happens in the
 * constructors, before the timer starts).  Optional: qcnn_forward_blu_clip allocates on demand. */
int qcnn_clip_reserve(qcnn_t *e, int nframes, int with_sse);

/* Frames that one kernel launch of qcnn_forward_blu_batch covers (the loop of
 * inference/kernel.cu:91-97 has no such knob: it is the price of the intermediate conc1 image,
 * 48*height*width bytes of device scratch per frame of a launch).  set > 0 changes it, current
 * (may be NULL) receives the value in force.  Default: as many frames as fit 8 GiB, at most 16. */
int qcnn_batch_group(qcnn_t *e, int set, int *current);

/* The same pass run layer by layer through the qcnn_op_* kernels below, materialising every
 * buffer of the reference (x_ppro, u, v, conc) in the reference's layouts: the tap/debug mode
 * (the role of MEM_DBG / viewmem, inference/cnn.cu:203-248). */
int qcnn_forward_blu_layered(qcnn_t *e);

/* ---- dynamic-quantisation path ---------------------------------------------------------- */
/* qvrcnn::load_para(filename), inference/qvrcnn.cu:30-46 -> CovLayer::load_para, inference/cnn.cu:69-89:
 * per layer an int32 step_w, the int8 weights (NCHW_VECT_C) and the int32 biases; 59 980 bytes.  The
 * engine holds one model at a time: this replaces a static model and qcnn_load_static replaces this. */
int qcnn_load_para(qcnn_t *e, const char *path);
int qcnn_load_para_mem(qcnn_t *e, const void *image, size_t bytes);

/* qvrcnn::forward(), inference/qvrcnn.cu:82-167: the pass whose activation steps are derived per frame
 * from the largest |u| of every layer.  The shipped text of forward() calls quantize_out_static (:100),
 * concat_blu (:122, :141) and adjustOutput_static(.., 141, 16) (:157), which read mul / shift / blu that
 * load_para never sets; this entry point runs the alternatives the source keeps beside them --
 * quantize_out (:98, inference/cnn.cu:169-178), concat (:121, :140, inference/cnn.cu:285-320) and
 * adjustOutput(steps, ..) (:156) -- and everything else as written: adjustBasic before every
 * convolution, no activation after C1, ReLU after C2_x / C3_x, insert_w / insert_y, applyRes.
 * (The literal text, static helpers included, runs through the C++ class API: qvrcnn::forward_shipped in
 * include/qvrcnn.h, with the fields load_para does not set supplied by the caller.)
 * save_steps (:163) is a debug file append and is left to the caller.  Asynchronous on the engine
 * stream: the step bookkeeping runs on the device, no maximum travels to the host.  No float32 image
 * is written: each layer's convolution runs once for its largest |u| and once more with the
 * divide-quantiser in its epilogue (C1.v, Conc1.conc, Conc2.conc, C4.u, C4.v and I1.x_rec are
 * materialised; C1.u .. C3_2.u are not). */
int qcnn_forward(qcnn_t *e);

/* The same pass with every u written as float32 NCHW, as the reference does (tap / debug mode: all the
 * QCNN_TAP_* buffers are valid afterwards).  Results are identical to qcnn_forward. */
int qcnn_forward_layered(qcnn_t *e);

/* Host copy of what the reference keeps in qvrcnn::steps and CovLayer::step_w / step_y / max_u / b_adj
 * after forward() (synchronises the engine stream). */
typedef struct {
    int32_t stepw[4], stepy[4];                    /* step_parameters, inference/qvrcnn.cuh:11-15 */
    int32_t step_w[QCNN_NUM_LAYERS], step_y[QCNN_NUM_LAYERS], max_u[QCNN_NUM_LAYERS];
    float b_adj[161];                              /* 64, 32, 16, 16, 32, 1 concatenated */
} qcnn_dynamic_state_t;
int qcnn_dynamic_state(qcnn_t *e, qcnn_dynamic_state_t *out);

/* ---- float32 configuration ---------------------------------------------------------------- */
/* The reference's compile-time alternative FLOAT_CONFIG (inference/mat.cuh:42,89-101): activations and
 * weights float32 NCHW.  Here it is a third model an engine can hold beside the int8 one.
 * qvrcnn::load_para -> CovLayer::load_para, inference/cnn.cu:114-127: per layer ks*ks*cin*cout float32
 * weights [K][C][R][S] and outChannel float32 biases (the NCHW_MODEL file of inference/kernel.cu:9;
 * 218 692 bytes). */
int qcnn_load_float(qcnn_t *e, const char *path);
int qcnn_load_float_mem(qcnn_t *e, const void *image, size_t bytes);

/* qvrcnn::forward of that configuration, inference/qvrcnn.cu:244-285: FLOAT_PPRO, six float32 convolutions
 * on the f32 matrix instructions with bias and ReLU (all layers but C4) in their epilogues, the branch
 * pairs writing straight into Conc1.conc / Conc2.conc, then applyRes.  Asynchronous on the engine stream.
 * Float results depend on the summation order, which the reference leaves to cuDNN: parity is by
 * tolerance (see tests/test_gpu_float.py), not bit-exact. */
int qcnn_forward_float(qcnn_t *e);

/* Device addresses of CovLayer::w / b / u of layer l in the float configuration (C2_x.u and C3_x.u are
 * the channel ranges of Conc1.conc / Conc2.conc). */
void *qcnn_float_w_device_ptr(qcnn_t *e, int layer);
void *qcnn_float_b_device_ptr(qcnn_t *e, int layer);
void *qcnn_float_u_device_ptr(qcnn_t *e, int layer);

/* cudaDeviceSynchronize() of inference/kernel.cu:95, restricted to the engine stream. */
int qcnn_sync(qcnn_t *e);

/* Device address of I1.x_rec (uint8 [H][W]); inference/kernel.cu:96 reads it. */
void *qcnn_recon_device_ptr(qcnn_t *e);
/* Device address of I1.x (uint8 [H][W]); lets a caller keep frames resident in HBM. */
void *qcnn_input_device_ptr(qcnn_t *e);

/* Points the engine at caller-owned device buffers for the next forward passes (frames
 * already resident in HBM); NULL restores the engine's own I1.x / I1.x_rec. */
int qcnn_bind_io(qcnn_t *e, const void *dev_y, void *dev_rec);

/* The D2H copy of inference/kernel.cu:96 (synchronises the stream first). */
int qcnn_download(qcnn_t *e, uint8_t *host_rec);
int qcnn_download_async(qcnn_t *e, uint8_t *host_rec);

/* Sum of squared differences between a device frame and a device original (uint8, n bytes),
 * the integer that vrcnn_data::psnr (inference/yuv_data.cpp:87-97) divides by n. */
int qcnn_sse_device(qcnn_t *e, const void *dev_a, const void *dev_b, size_t n, uint64_t *sse);

/* Taps after qcnn_forward_blu_layered, copied to host in the reference's layouts. */
enum {
    QCNN_TAP_X_PPRO = 0, /* int8  [1][H][W][4]           I1.x_ppro (lanes 1..3 zero)      */
    QCNN_TAP_U_C1 = 1,   /* float [64][H][W]             C1.u                             */
    QCNN_TAP_U_C2_1 = 2, /* float [32][H][W]                                              */
    QCNN_TAP_U_C2_2 = 3, /* float [16][H][W]                                              */
    QCNN_TAP_U_C3_1 = 4, /* float [16][H][W]                                              */
    QCNN_TAP_U_C3_2 = 5, /* float [32][H][W]                                              */
    QCNN_TAP_U_C4 = 6,   /* float [1][H][W]                                               */
    QCNN_TAP_V_C1 = 7,   /* int8  [16][H][W][4]          C1.v                             */
    QCNN_TAP_CONC1 = 8,  /* int8  [12][H][W][4]          Conc1.conc                       */
    QCNN_TAP_CONC2 = 9,  /* int8  [12][H][W][4]          Conc2.conc                       */
    QCNN_TAP_X_REC = 10, /* uint8 [H][W]                 I1.x_rec                         */
    /* after qcnn_forward_blu (fused path): the one intermediate it keeps in HBM, converted
     * back to the reference layout of Conc1.conc */
    QCNN_TAP_FUSED_CONC1 = 11,
    QCNN_TAP_V_C4 = 12,  /* int8  [H][W]                 C4.v after qcnn_forward (adjustOutput)      */
    /* after qcnn_forward_float */
    QCNN_TAP_F_X_PPRO = 13, /* float [H][W]              I1.x_ppro                        */
    QCNN_TAP_F_U_C1 = 14,   /* float [64][H][W]          C1.u (after ReLU)                */
    QCNN_TAP_F_CONC1 = 15,  /* float [48][H][W]          Conc1.conc                       */
    QCNN_TAP_F_CONC2 = 16,  /* float [48][H][W]          Conc2.conc                       */
    QCNN_TAP_F_U_C4 = 17    /* float [1][H][W]           C4.u                             */
};
int qcnn_tap_bytes(qcnn_t *e, int tap, size_t *bytes);
int qcnn_tap(qcnn_t *e, int tap, void *host_out, size_t bytes);
/* Device address of a layered-mode buffer (the public members C*.u / C*.v / Conc*.conc).  NULL for
 * QCNN_TAP_FUSED_CONC1: the fused path keeps conc1 as a bordered image in an internal channel order, and the tap
 * is a converted copy that qcnn_tap makes (of the image the last fused pass wrote, whichever entry point ran it). */
void *qcnn_tap_device_ptr(qcnn_t *e, int tap);

/* blu, mul, shift of layer l as loaded (CovLayer::blu/mul/shift, inference/cnn.cuh:49-50). */
int qcnn_layer_params(qcnn_t *e, int layer, int *blu, int *mul, int *shift);
/* Device addresses of layer l's weights (file order, int8) and float biases (CovLayer::w/b) of the
 * model loaded last (static or dynamic), and of the adjusted biases qcnn_forward computes
 * (CovLayer::b_adj, inference/cnn.cu:62). */
void *qcnn_layer_w_device_ptr(qcnn_t *e, int layer);
void *qcnn_layer_b_device_ptr(qcnn_t *e, int layer);
void *qcnn_layer_b_adj_device_ptr(qcnn_t *e, int layer);

/* Arithmetic of the fused kernels.  At every load the engine proves from the weights, per layer, whether
 * the reference's float32 accumulator pipeline (inference/cnn.cu:148-155, inference/mat.cu:275-291)
 * is identical to integer arithmetic (all |acc|, |bias| < 2^24, requantiser operands inside 24 bits) and
 * then uses integer epilogues.  The two fused kernels choose separately: kernel A (C1, C2_1, C2_2) runs
 * 0 = integer epilogues, 2 = integer epilogues with the C2 accumulators clamped first (a model whose C2
 * accumulator bounds alone fail the proof; one more instruction per value), or 1 = the float32 pipeline
 * replayed literally; kernel B (C3_1, C3_2, C4) runs 0 or 1.  qcnn_is_exact returns
 * path of kernel A | path of kernel B << 4 (0 = plain integers everywhere).  qcnn_set_exact(e, 1) forces the
 * literal pipeline in both kernels from the next load on, qcnn_set_exact(e, 2) the clamped form of kernel A
 * for a model that qualifies for plain integers, qcnn_set_exact(e, 0) returns to the proof's choice. */
int qcnn_set_exact(qcnn_t *e, int on);
int qcnn_is_exact(qcnn_t *e);
/* The engine's hipStream_t (as void*), for callers that enqueue their own work behind it. */
void *qcnn_stream(qcnn_t *e);

/* Average device time, in microseconds, of the two fused kernels over the forward passes
 * enqueued since the last call with reset != 0 (HIP events on the engine stream; enabled by
 * qcnn_set_timing(e, 1), off by default because the events cost launch latency). */
int qcnn_set_timing(qcnn_t *e, int on);
int qcnn_kernel_times_us(qcnn_t *e, double *k_c1c2_us, double *k_c3c4_us, int *passes, int reset);

/* ---- layer level: stateless kernels on raw device pointers ---------------------------- */
/* Each enqueues on `stream` (a hipStream_t passed as void*, NULL = default stream) and
 * returns without synchronising. */

/* InputLayer::ppro / HW2HW_VECT_C_PPRO, inference/cnn.cu:445-468: x_ppro[4i] = x[i]-128,
 * lanes 1..3 written as 0.  x: uint8[n]; x_ppro: int8[4n]. */
int qcnn_op_ppro(const void *x, void *x_ppro, size_t n, void *stream);

/* CovLayer::ConvForward_static, inference/cnn.cu:145-160: u = float(conv_int32(x, w)) + b.
 * x: int8 NCHW_VECT_C [pad4(cin)/4][H][W][4]; w: int8 [cout][pad4(cin)/4][ks][ks][4];
 * b: float[cout]; u: float [cout][H][W].  ks in {1,3,5}. */
int qcnn_op_conv_static(const void *x, const void *w, const void *b, void *u, int height, int width, int cin,
                        int cout, int ks, void *stream);

/* CHW2CHW_VECT_C_QUANT_BLU, inference/mat.cu:262-303 (used by quantize_out_blu,
 * inference/cnn.cu:195-200, and concat_blu, inference/cnn.cu:375-394): ReLU + clip at blu +
 * requantise + relayout.  u: float [channels][hw]; out: int8 [channels/4][hw][4]. */
int qcnn_op_quant_blu(const void *u, void *out, size_t hw, int channels, int blu, int mul, int shift,
                      void *stream);

/* InputLayer::applyRes_y / applyRes_GPU_y, inference/cnn.cu:507-535. */
int qcnn_op_apply_res_y(const void *x, void *rec, const void *res, int mul, int shift, size_t n, void *stream);

/* CovLayer::activate, inference/cnn.cu:161-166 (CUDNN_ACTIVATION_RELU, :57), in place on float u[n]. */
int qcnn_op_activate(void *u, size_t n, void *stream);

/* findMax, inference/mat.cu:28-96: *host_max = largest |u[i]|.  `workspace` is device memory (>= 4 bytes),
 * the reference's `buffer`.  Blocking, like the reference's copy of the result to the host. */
int qcnn_op_find_max(const void *u, void *workspace, size_t n, float *host_max, void *stream);

/* CHW2CHW_VECT_C, inference/mat.cu:197-236 (NCHW2NCHW_VECT_C :237-247 per frame): float32
 * (u -+ divisor/2) / divisor, clipped to [-128, 127], relayout.  u: float [channels][hw];
 * out: int8 [channels/4][hw][4]. */
int qcnn_op_quant_div(const void *u, void *out, size_t hw, int channels, int divisor, void *stream);

/* mul_shift, inference/mat.cu:248-261 (CovLayer::quantize_out_static, inference/cnn.cu:189-194):
 * out[i] = char((((int)u[i] + (1 << shift-1)/mul) * mul) >> shift), flat, no relayout. */
int qcnn_op_mul_shift(const void *u, void *out, size_t n, int mul, int shift, void *stream);

/* adjustBasic, inference/qvrcnn.cu:336-349: b_adj[i] = b[i] * stepw[0..n-1] / stepy[0..n-1], each
 * division rounding half away from zero.  stepw / stepy: host int32[4]; b, b_adj: device float[count]. */
int qcnn_op_adjust_basic(const int32_t *stepw, const int32_t *stepy, const void *b, void *b_adj, int count, int n,
                         void *stream);

/* adjustOutput, inference/qvrcnn.cu:368-385: o_adj[i] = char(o[i] * stepy[0..n-2] / stepw[n-1..0]).
 * o: device float[num]; o_adj: device int8[num]. */
int qcnn_op_adjust_output(const int32_t *stepw, const int32_t *stepy, const void *o, void *o_adj, int n, size_t num,
                          void *stream);

/* adjustOutput_static, inference/qvrcnn.cu:386-397: o_adj[i] = char(((int)o[i] * mul + (1 << shift-1)) >> shift). */
int qcnn_op_adjust_output_static(const void *o, void *o_adj, int mul, int shift, size_t num, void *stream);

/* InputLayer::applyRes / applyRes_GPU, inference/cnn.cu:487-502,524-529: rec = clamp((short)x + res),
 * res int8[n]. */
int qcnn_op_apply_res(const void *x, const void *res, void *rec, size_t n, void *stream);

/* FLOAT_PPRO, inference/cnn.cu:470-484: x_ppro[i] = ((float)x[i] - 128) / 255.  x: uint8[n]; x_ppro: float[n]. */
int qcnn_op_ppro_float(const void *x, void *x_ppro, size_t n, void *stream);

/* CovLayer::ConvForward of the float configuration, inference/cnn.cu:129-144 (+ CovLayer::activate,
 * inference/cnn.cu:161-166, when relu != 0): u = conv(x, w) + b.  x: float NCHW [cin][H][W];
 * w: float [cout][cin][ks][ks]; b: float[cout]; u: float NCHW [cout][H][W].  ks in {1,3,5}, cin in
 * {1 (ks 5 only), 48, 64}, cout <= 64. */
int qcnn_op_conv_float(const void *x, const void *w, const void *b, void *u, int height, int width, int cin, int cout,
                       int ks, int relu, void *stream);

/* applyRes_GPU, float arm, inference/cnn.cu:487-502: rec = clamp((short)(x + res*255 + 0.5), 0, 255). */
int qcnn_op_apply_res_float(const void *x, const void *res, void *rec, size_t n, void *stream);

/* The step selection of ConcatLayer::concat, inference/cnn.cu:295-309 (host only): one step per branch
 * from its own maximum (1 if the maximum fits in 127), then the finer branch coarsened so that
 * step_w * step_y of both branches agree. */
int qcnn_concat_steps(float max1, float max2, int step_w1, int step_w2, int *stepy1, int *stepy2);

/* ---- device memory helpers for hosts that do not link HIP ------------------------------ */
int qcnn_set_device(int gpu);
int qcnn_dev_alloc(void **dev, size_t bytes);
int qcnn_dev_free(void *dev);
int qcnn_dev_h2d(void *dev, const void *host, size_t bytes);
int qcnn_dev_d2h(void *host, const void *dev, size_t bytes);
int qcnn_dev_memset(void *dev, int value, size_t bytes);
int qcnn_dev_sync(void);
/* Free and total device memory of the current device (hipMemGetInfo); either pointer may be NULL. */
int qcnn_dev_mem_info(size_t *free_bytes, size_t *total_bytes);
/* Page-locked host memory, portable: every device of the process may copy from / to it by DMA (the workers of
 * a multi-GPU run share one clip). */
int qcnn_host_alloc_pinned(void **host, size_t bytes);
int qcnn_host_free_pinned(void *host);
/* Page-locks (for every device of the process) / releases `bytes` of memory the caller owns (hipHostRegister /
 * hipHostUnregister), e.g. the three
 * clips of vrcnn_data (inference/yuv_data.cpp:8-10) when they were allocated with new[]. */
int qcnn_host_register(void *host, size_t bytes);
int qcnn_host_unregister(void *host);

/* ---- offline model conversion (host only) ----------------------------------------------- */
/* HWCN2NCHW_VECT_C_CPU, inference/mat.cu:97-119: out[n][c/4][h][w][c%4] = in[h][w][c][n]. */
long qcnn_hwcn2nchw_vect_c(const int8_t *hwcn, int h, int w, int c, int n, int8_t *out);
/* model_qfp_HWCN2NCHW_VECT_C, inference/qvrcnn.cu:558-585, file to file.  `qp` is substituted
 * into both printf-style patterns as the reference does. */
int qcnn_model_qfp_hwcn2nchw_vect_c(const char *filein, const char *fileout, int qp);

/* HWCN2NCHW_CPU, inference/mat.cu:160-176 (float weights): out[n][c][h][w] = in[h][w][c][n]. */
long qcnn_hwcn2nchw(const float *hwcn, int h, int w, int c, int n, float *out);
/* model_HWCN2NCHW, inference/qvrcnn.cu:464-491 (float configuration), file to file. */
int qcnn_model_hwcn2nchw(const char *filein, const char *fileout, int qp);

#ifdef __cplusplus
}
#endif
#endif /* QCNN_H */
