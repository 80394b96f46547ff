/*
 * qvrcnn.h -- the reference's network class (inference/qvrcnn.cuh:25-59) on top of the C ABI.
 * The driver code of inference/kernel.cu:86-96 compiles against it unchanged apart from the
 * CUDA runtime calls, which become qcnn_dev_sync() / qcnn_dev_d2h():
 *
 *     qvrcnn qvrcnn1(0, 1, channel, height, width);
 *     qvrcnn1.load_static_para(model_fn);
 *     qvrcnn1.load_data(frame);
 *     qvrcnn1.forward_blu();
 *     qcnn_dev_sync();
 *     qcnn_dev_d2h(recon, qvrcnn1.I1.x_rec, channel*height*width);
 */
#ifndef QCNN_QVRCNN_H
#define QCNN_QVRCNN_H

#include "cnn.h"

#define LAYER 4
typedef struct {
    int stepw[LAYER];
    int stepy[LAYER];
} step_parameters; /* inference/qvrcnn.cuh:10-15: stepw descending, stepy ascending */

/* Host-callable forms of the reference's bookkeeping kernels (inference/qvrcnn.cuh:21-23), which it
 * launches as adjustBasic<<<1, outChannel>>>(..) etc.; here the element count is an argument.
 * b / b_adj / o / o_adj are device pointers. */
int adjustBasic(step_parameters steps, btype *b, btype *b_adj, int count, int n);
int adjustOutput(step_parameters steps, convtype *o, xwtype *o_adj, int n, int num);
int adjustOutput_static(convtype *o, xwtype *o_adj, int multiplier, int shifts, int num);

/* model_qfp_HWCN2NCHW_VECT_C, inference/qvrcnn.cu:558-585 */
int model_qfp_HWCN2NCHW_VECT_C(const char *filein, const char *fileout, int qp);

class qvrcnn {
public:
    qvrcnn(int gpu_num, int batch, int channel, int height, int width); /* inference/qvrcnn.cu:4-29 */
    int load_para(char *filename);                                      /* inference/qvrcnn.cu:30-46 */
    int load_static_para(char *filename);                               /* inference/qvrcnn.cu:47-63 */
    int load_data(datatype *input);                                     /* inference/qvrcnn.cu:64-68 */
    /* inference/qvrcnn.cu:82-167 with quantize_out / concat / adjustOutput (see qcnn_forward in qcnn.h);
     * one asynchronous sequence of kernels, then steps / step_y / max_u are copied back into the members */
    int forward(void);
    /* the same pass issued call by call through the layer objects, with the host round trips of the
     * reference (findMax copies every maximum to the host) */
    int forward_layers(void);
    /* inference/qvrcnn.cu:82-167 as the shipped text reads, call by call through the layer objects: quantize_out_static
     * (:100), concat_blu (:122, :141) and adjustOutput_static(.., 141, 16) (:157) inside the dynamic pass.  load_para sets
     * none of the fields these read: the caller puts blu / mul / shift into the layer members and a step (>= 1) into
     * C1.step_y first, as the reference's author must have done by hand.  (quantize_out_static does not relayout, so
     * C1.v holds NCHW bytes that C2_x read as NCHW_VECT_C: the text is reproduced, not repaired.) */
    int forward_shipped(void);
    int forward_blu(void);       /* inference/qvrcnn.cu:168-242, as the two fused gfx950 kernels */
    /* the same pass issued call by call through the layer objects, in the order and with the
     * buffers of inference/qvrcnn.cu:168-242 (every u / v / conc is materialised) */
    int forward_blu_layers(void);
    /* inference/qvrcnn.cuh:33: declared by the reference, defined nowhere in it (inference only); prints
     * and returns QCNN_E_UNSUPPORTED */
    int backward(void);
    int save_b_adj(const char *filename);                               /* inference/qvrcnn.cu:288-304 */
    /* The reference's compile-time float32 configuration (FLOAT_CONFIG) as run-time members: load_para of
     * inference/cnn.cu:114-127 (NCHW float model) and forward of inference/qvrcnn.cu:244-285. */
    int load_float_para(char *filename);
    int forward_float(void);
    ~qvrcnn();

    int layers;
    qcnnHandle_t cudnnHandle; /* name kept from the reference */
    qcnn_t *engine;

    InputLayer I1;
    CovLayer C1;
    CovLayer C2_1;
    CovLayer C2_2;
    ConcatLayer Conc1;
    CovLayer C3_1;
    CovLayer C3_2;
    ConcatLayer Conc2;
    CovLayer C4;
    int batch, channel, height, width;
    step_parameters steps;
    void insert_w(int stepw, int layer);
    void insert_y(int stepy, int layer);
    void *workspace = nullptr;
    size_t workspaceSize;

private:
    struct qcnnHandle_s handle_storage_;
};

#endif
