/*
 * mat.h -- types and constants of the reference's active configuration INT8x4_EXT_CONFIG
 * (inference/mat.cuh:59-73) plus the host-side weight relayout it declares (inference/mat.cuh:109).
 * The cuDNN handle / descriptor types of the reference's signatures become the small opaque
 * types below, so that calls keep their arity.
 */
#ifndef QCNN_MAT_H
#define QCNN_MAT_H

#include "qcnn.h"

#define FRAME 1   /* inference/mat.cuh:23 */
#define CHANNEL 1 /* inference/mat.cuh:24 */
#define THRESHOLD 127

typedef char xwtype;            /* int8 activations and weights, NCHW_VECT_C */
typedef float convtype;         /* conv output u: float32 NCHW */
typedef float btype;            /* bias, converted from the file's int32 */
typedef int midtype;
typedef unsigned char datatype; /* 8-bit luma (inference/yuv_data.h:9) */
typedef char restype;

/* stands in for cudnnHandle_t: the layer kernels need no library handle, only a stream */
typedef struct qcnnHandle_s {
    void *stream;
} *qcnnHandle_t;

/* stands in for cudnnTensorDescriptor_t */
typedef struct {
    int n, c, h, w;
} qcnnTensorDescriptor_t;

/* check(status), inference/cnn.cuh:8-15: print, reset nothing we do not own, exit(1) */
#define check(status)                                 \
    do {                                              \
        if ((status) != 0) {                          \
            printf("cudnn returned none 0.\n");       \
            printf("%s\n", qcnn_last_error());        \
            exit(1);                                  \
        }                                             \
    } while (0)

/* HWCN2NCHW_VECT_C_CPU, inference/mat.cu:97-119; caller frees the result with delete[]. */
char *HWCN2NCHW_VECT_C_CPU(char *HWCN, int H, int W, int C, int N, int *outSize);
/* NCHW2NCHW_VECT_C_QUANT_BLU, inference/mat.cu:304-314, on device pointers. */
int NCHW2NCHW_VECT_C_QUANT_BLU(convtype *before, xwtype *after, int N, int C, int H, int W, int blu, int multiplier,
                               int shifts);

/* findMax, inference/mat.cu:28-35: *max = largest |data[i]|; `buffer` is device scratch (>= 4 bytes). */
void findMax(convtype *data, convtype *buffer, int n, convtype *max);
/* NCHW2NCHW_VECT_C, inference/mat.cu:237-247, on device pointers. */
int NCHW2NCHW_VECT_C(convtype *dividend, xwtype *quotient, int divisor, int N, int C, int H, int W);

#endif
