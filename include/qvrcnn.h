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
} step_parameters; /* inference/qvrcnn.cuh:10-15; unused on the static path */

/* model_qfp_HWCN2NCHW_VECT_C, inference/qvrcnn.cu:558-585 */
int model_qfp_HWCN2NCHW_VECT_C(const char *filein, const char *fileout, int qp);

class qvrcnn {
public:
    qvrcnn(int gpu_num, int batch, int channel, int height, int width); /* inference/qvrcnn.cu:4-29 */
    int load_para(char *filename);                                      /* dynamic path: unsupported */
    int load_static_para(char *filename);                               /* inference/qvrcnn.cu:47-63 */
    int load_data(datatype *input);                                     /* inference/qvrcnn.cu:64-68 */
    int forward(void);                                                  /* dynamic path: unsupported */
    int forward_blu(void);       /* inference/qvrcnn.cu:168-242, as the two fused gfx950 kernels */
    /* the same pass issued call by call through the layer objects, in the order and with the
     * buffers of inference/qvrcnn.cu:168-242 (every u / v / conc is materialised) */
    int forward_blu_layers(void);
    int save_b_adj(const char *filename);                               /* unsupported */
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
