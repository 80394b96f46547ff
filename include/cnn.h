/*
 * cnn.h -- the reference's layer API (inference/cnn.cuh:17-100) on top of the C ABI of qcnn.h.
 * Same class names, same public members, same method names and argument meaning; the
 * cudnnHandle_t / cudnnTensorDescriptor_t parameters are the stand-ins of mat.h.  Both the
 * static-quantised members (load_static_para, ConvForward_static, quantize_out_blu, concat_blu,
 * ppro, applyRes_y) and the dynamic-quantisation members (load_para, ConvForward, activate,
 * quantize_out, quantize_out_fix, quantize_out_static, concat, concat_fix, applyRes) are functional.
 */
#ifndef QCNN_CNN_H
#define QCNN_CNN_H

#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "mat.h"

class CovLayer {
public:
    CovLayer();
    int build(qcnnHandle_t handle, qcnnTensorDescriptor_t xDesc, int batch, int height, int width, int inChannel,
              int outChannel, int ksize);                                  /* inference/cnn.cu:13-66 */
    int load_para(FILE *fp);                                               /* inference/cnn.cu:69-89 */
    int load_static_para(FILE *fp);                                        /* inference/cnn.cu:90-112 */
    int ConvForward(qcnnHandle_t handle, qcnnTensorDescriptor_t xDesc, void *x, void *workspace,
                    size_t workspaceSize);                                 /* inference/cnn.cu:129-144 */
    int ConvForward_static(qcnnHandle_t handle, qcnnTensorDescriptor_t xDesc, void *x, void *workspace,
                           size_t workspaceSize);                          /* inference/cnn.cu:145-160 */
    int activate(qcnnHandle_t handle);                                     /* inference/cnn.cu:161-166 */
    int quantize_out(void *workspace);                                     /* inference/cnn.cu:169-178 */
    int quantize_out_fix(int max_u);                                       /* inference/cnn.cu:179-188 */
    int quantize_out_static(void);                                         /* inference/cnn.cu:189-194 */
    int quantize_out_blu(void);                                            /* inference/cnn.cu:195-200 */
    int viewmem(xwtype *x);                                                /* inference/cnn.cu:203-248 */
    /* declared by the reference (inference/cnn.cuh:33-34, "free data u and y" / "set u and y") and defined
     * nowhere in it; here they release / re-allocate u and v of a layer that owns its buffers (build()) and
     * return QCNN_E_UNSUPPORTED on a layer whose buffers belong to an engine (adopt()) */
    int freeMem(void);
    int setMem(void);
    /* adopt buffers owned by a qcnn engine instead of allocating (used by class qvrcnn) */
    int adopt(int batch, int height, int width, int inChannel, int outChannel, int ksize, void *w, void *b, void *u,
              void *v);
    ~CovLayer();

    int batch, height, width, inChannel, outChannel, ksize;
    int wSize, uSize, vSize;
    size_t workspaceSize;
    int max_u;
    int step_w, step_y;
    int blu;
    int mul, shift;
    float alpha = 1;
    float beta = 0;
    qcnnTensorDescriptor_t uDesc, vDesc;
    void *w, *b, *b_adj;
    void *u, *v;

private:
    bool owns_;
};

class ConcatLayer {
public:
    ConcatLayer(void);
    int build(int batch, int height, int width, int inChannel1, int inChannel2);   /* inference/cnn.cu:268-283 */
    int concat(CovLayer *C1, CovLayer *C2, void *workspace);                        /* inference/cnn.cu:285-320 */
    int concat_fix(CovLayer *C1, CovLayer *C2, convtype max1, convtype max2);       /* inference/cnn.cu:321-353 */
    /* inference/cnn.cuh:69; its only body (inference/cnn.cu:354-374) is commented out and launches a kernel
     * the reference does not contain: does the host bookkeeping of that body (step_y, max_u), prints and
     * returns QCNN_E_UNSUPPORTED */
    int concat_static(CovLayer *C1, CovLayer *C2);
    int concat_blu(CovLayer *C1, CovLayer *C2);                                     /* inference/cnn.cu:375-394 */
    int adopt(int batch, int height, int width, int inChannel1, int inChannel2, void *conc);
    ~ConcatLayer();

    int batch, height, width, inChannel1, inChannel2, outChannel;
    qcnnTensorDescriptor_t concDesc;
    void *conc;

private:
    bool owns_;
};

class InputLayer {
public:
    InputLayer(void);
    int build(int batch, int channel, int height, int width);      /* inference/cnn.cu:414-438 */
    int load(datatype *input);                                      /* inference/cnn.cu:439-443 */
    int ppro(void);                                                 /* inference/cnn.cu:454-468 */
    int applyRes(xwtype *res);                                      /* inference/cnn.cu:524-529 */
    int applyRes_y(convtype *res, int mul, int shift);              /* inference/cnn.cu:530-535 */
    int viewmem(xwtype *res);                                       /* inference/cnn.cu:536-549 */
    int adopt(int batch, int channel, int height, int width, void *x, void *x_ppro, void *x_rec);
    ~InputLayer(void);

    int batch, height, width, inChannel, outChannel;
    int inSize, outSize;
    qcnnTensorDescriptor_t xDesc;
    void *x, *x_ppro, *x_rec;

private:
    bool owns_;
};

#endif
