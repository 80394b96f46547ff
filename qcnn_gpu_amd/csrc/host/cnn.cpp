// Layer classes of the reference (inference/cnn.cuh) over the C ABI: host C++ only, no HIP headers.
#include <vector>

#include "cnn.h"

#include <string.h>

static int unsupported(const char *what)
{
    printf("%s: not available in this configuration\n", what);
    return QCNN_E_UNSUPPORTED;
}
static inline int pad4(int c) { return (c + 3) / 4 * 4; }

// ---------------------------------------------------------------------------------------------
CovLayer::CovLayer()
    : batch(0), height(0), width(0), inChannel(0), outChannel(0), ksize(0), wSize(0), uSize(0), vSize(0),
      workspaceSize(0), max_u(0), step_w(0), step_y(0), blu(0), mul(0), shift(0), w(nullptr), b(nullptr),
      b_adj(nullptr), u(nullptr), v(nullptr), owns_(false)
{
}

static void set_shape(CovLayer *l, int batch, int height, int width, int inChannel, int outChannel, int ksize)
{
    l->batch = batch;
    l->height = height;
    l->width = width;
    l->inChannel = inChannel;
    l->outChannel = outChannel;
    l->ksize = ksize;
    l->uSize = batch * outChannel * height * width;
    l->wSize = ksize * ksize * pad4(inChannel) * outChannel;   // inference/cnn.cu:24
    l->vSize = batch * pad4(outChannel) * height * width;      // inference/cnn.cu:25
    l->uDesc = {batch, outChannel, height, width};
    l->vDesc = {batch, pad4(outChannel), height, width};
    l->workspaceSize = 0;                                      // the direct convolution needs no workspace
}

int CovLayer::build(qcnnHandle_t, qcnnTensorDescriptor_t, int batch, int height, int width, int inChannel,
                    int outChannel, int ksize)
{
    set_shape(this, batch, height, width, inChannel, outChannel, ksize);
    check(qcnn_dev_alloc(&w, sizeof(xwtype) * (size_t)wSize));
    check(qcnn_dev_alloc(&b, sizeof(btype) * (size_t)outChannel));
    check(qcnn_dev_alloc(&b_adj, sizeof(btype) * (size_t)outChannel));
    check(qcnn_dev_alloc(&u, sizeof(convtype) * (size_t)uSize));
    check(qcnn_dev_alloc(&v, sizeof(xwtype) * (size_t)vSize));
    owns_ = true;
    return 0;
}

int CovLayer::adopt(int batch, int height, int width, int inChannel, int outChannel, int ksize, void *w_, void *b_,
                    void *u_, void *v_)
{
    set_shape(this, batch, height, width, inChannel, outChannel, ksize);
    w = w_;
    b = b_;
    u = u_;
    v = v_;
    owns_ = false;
    return 0;
}

// The parameter block of one layer as both model formats hold it (inference/cnn.cu:69-112): an optional leading
// int32 (the weight step of the dynamic format), wSize int8 weights in NCHW_VECT_C order, outChannel int32 biases,
// then `tail` more int32 (blu, mul, shift of the static format).  The biases go to the device as floats, as the
// reference converts them (inference/cnn.cu:81,104).  Layers that adopted the engine's buffers only consume their
// part of the file: the engine's own loader has filled those buffers.
static void read_layer_block(CovLayer *L, FILE *fp, int *lead, int *tail, int ntail, bool upload)
{
    std::vector<xwtype> weights((size_t)L->wSize);
    std::vector<int> bias_i((size_t)L->outChannel);
    bool ok = !lead || fread(lead, sizeof(int), 1, fp) == 1;
    ok = ok && fread(weights.data(), sizeof(xwtype), weights.size(), fp) == weights.size();
    ok = ok && fread(bias_i.data(), sizeof(int), bias_i.size(), fp) == bias_i.size();
    ok = ok && (ntail == 0 || fread(tail, sizeof(int), (size_t)ntail, fp) == (size_t)ntail);
    if (!ok) {
        printf("model file is too short.\n");
        exit(1);
    }
    if (!upload) return;
    const std::vector<btype> bias_f(bias_i.begin(), bias_i.end());
    check(qcnn_dev_h2d(L->w, weights.data(), sizeof(xwtype) * weights.size()));
    check(qcnn_dev_h2d(L->b, bias_f.data(), sizeof(btype) * bias_f.size()));
}

int CovLayer::load_static_para(FILE *fp)
{
    int tail[3];
    read_layer_block(this, fp, nullptr, tail, 3, owns_);
    blu = tail[0], mul = tail[1], shift = tail[2];
    return 0;
}

int CovLayer::ConvForward_static(qcnnHandle_t handle, qcnnTensorDescriptor_t, void *x, void *, size_t)
{
    check(qcnn_op_conv_static(x, w, b, u, height, width, inChannel, outChannel, ksize, handle ? handle->stream : nullptr));
    return 0;
}

int CovLayer::quantize_out_blu(void)
{
    return NCHW2NCHW_VECT_C_QUANT_BLU((convtype *)u, (xwtype *)v, batch, outChannel, height, width, blu, mul, shift);
}

int CovLayer::viewmem(xwtype *x)
{
    // the 5x5 corner dump of inference/cnn.cu:203-248
    const size_t hw = (size_t)height * width;
    xwtype *x_h = new xwtype[hw * 4];
    convtype *u_h = new convtype[hw];
    xwtype *v_h = new xwtype[hw * 4];
    xwtype w_h[20];
    btype b_h[1];
    check(qcnn_dev_sync());
    check(qcnn_dev_d2h(x_h, x, hw * 4));
    check(qcnn_dev_d2h(w_h, w, sizeof w_h < (size_t)wSize ? sizeof w_h : (size_t)wSize));
    check(qcnn_dev_d2h(u_h, u, sizeof(convtype) * hw));
    check(qcnn_dev_d2h(v_h, v, hw * 4));
    check(qcnn_dev_d2h(b_h, b, sizeof b_h));
    const int n = height < 5 ? height : 5, m = width < 5 ? width : 5;
    printf("x:\n");
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < m; j++) printf("%d\t", x_h[(i * width + j) * 4]);
        printf("\n");
    }
    printf("weights:\n");
    for (int i = 0; i < 5 && i * 4 < wSize; i++) printf("%d\t", w_h[i * 4]);
    printf("\nbiases:%f\n", b_h[0]);
    printf("u:\n");
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < m; j++) printf("%f\t", u_h[i * width + j]);
        printf("\n");
    }
    printf("mul:%d,shift:%d\n", mul, shift);
    printf("v:\n");
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < m; j++) printf("%d\t", v_h[(i * width + j) * 4]);
        printf("\n");
    }
    delete[] x_h;
    delete[] u_h;
    delete[] v_h;
    return 0;
}

// "free data u and y" / "set u and y" (inference/cnn.cuh:33-34; no body in the reference)
int CovLayer::freeMem(void)
{
    if (!owns_) return unsupported("CovLayer::freeMem on buffers owned by an engine");
    check(qcnn_dev_sync());
    if (u) check(qcnn_dev_free(u));
    if (v) check(qcnn_dev_free(v));
    u = v = nullptr;
    return 0;
}

int CovLayer::setMem(void)
{
    if (!owns_) return unsupported("CovLayer::setMem on buffers owned by an engine");
    if (!u) check(qcnn_dev_alloc(&u, sizeof(convtype) * (size_t)uSize));
    if (!v) check(qcnn_dev_alloc(&v, sizeof(xwtype) * (size_t)vSize));
    return 0;
}

int CovLayer::load_para(FILE *fp)
{
    read_layer_block(this, fp, &step_w, nullptr, 0, owns_);
    return 0;
}

int CovLayer::ConvForward(qcnnHandle_t handle, qcnnTensorDescriptor_t, void *x, void *, size_t)
{
    // convolution + cudnnAddTensor of b_adj (inference/cnn.cu:132-139)
    check(qcnn_op_conv_static(x, w, b_adj, u, height, width, inChannel, outChannel, ksize,
                              handle ? handle->stream : nullptr));
    return 0;
}

int CovLayer::activate(qcnnHandle_t handle)
{
    check(qcnn_op_activate(u, (size_t)uSize, handle ? handle->stream : nullptr));
    return 0;
}

int CovLayer::quantize_out(void *workspace)
{
    convtype max;
    findMax((convtype *)u, (convtype *)workspace, uSize, &max);
    max_u = max;
    step_y = max / (THRESHOLD + 1) + 1;
    NCHW2NCHW_VECT_C((convtype *)u, (xwtype *)v, step_y, batch, outChannel, height, width);
    return 0;
}

int CovLayer::quantize_out_fix(int max_u_)
{
    convtype max = max_u_;
    max_u = max_u_;
    step_y = max / (THRESHOLD + 1) + 1;
    NCHW2NCHW_VECT_C((convtype *)u, (xwtype *)v, step_y, batch, outChannel, height, width);
    return 0;
}

int CovLayer::quantize_out_static(void)
{
    check(qcnn_op_mul_shift(u, v, (size_t)uSize, mul, shift, nullptr));
    return 0;
}

CovLayer::~CovLayer()
{
    if (owns_) {
        qcnn_dev_free(w);
        qcnn_dev_free(b);
        qcnn_dev_free(b_adj);
        qcnn_dev_free(u);
        qcnn_dev_free(v);
    }
}

// ---------------------------------------------------------------------------------------------
ConcatLayer::ConcatLayer(void)
    : batch(0), height(0), width(0), inChannel1(0), inChannel2(0), outChannel(0), conc(nullptr), owns_(false)
{
}

int ConcatLayer::build(int batch_, int height_, int width_, int inChannel1_, int inChannel2_)
{
    adopt(batch_, height_, width_, inChannel1_, inChannel2_, nullptr);
    check(qcnn_dev_alloc(&conc, sizeof(xwtype) * (size_t)batch * outChannel * height * width));
    owns_ = true;
    return 0;
}

int ConcatLayer::adopt(int batch_, int height_, int width_, int inChannel1_, int inChannel2_, void *conc_)
{
    batch = batch_;
    height = height_;
    width = width_;
    inChannel1 = inChannel1_;
    inChannel2 = inChannel2_;
    outChannel = inChannel1_ + inChannel2_;
    concDesc = {batch, outChannel, height, width};
    conc = conc_;
    owns_ = false;
    return 0;
}

int ConcatLayer::concat_blu(CovLayer *C1, CovLayer *C2)
{
    // Each branch is requantised with its own (blu, mul, shift) into its channel range of conc; the second range
    // starts where the first ends, which is legal because the first branch has a multiple of 4 channels.
    const size_t plane = (size_t)C1->height * C1->width;
    const size_t out_frame = plane * (size_t)(C1->outChannel + C2->outChannel);
    size_t first_channel = 0;
    for (CovLayer *L : {C1, C2}) {
        // host bookkeeping of inference/cnn.cu:378-381 (unused by the static path, kept observable)
        L->step_y = ((1 << L->shift) + L->mul / 2) / L->mul;
        L->max_u = 127 * L->step_y;
        const size_t in_frame = plane * (size_t)L->outChannel;
        for (int n = 0; n < C1->batch; n++)
            check(qcnn_op_quant_blu((convtype *)L->u + n * in_frame, (xwtype *)conc + n * out_frame + first_channel * plane,
                                    plane, L->outChannel, L->blu, L->mul, L->shift, nullptr));
        first_channel += (size_t)L->outChannel;
    }
    return 0;
}

int ConcatLayer::concat_static(CovLayer *C1, CovLayer *C2)
{
    // the host half of the commented-out body (inference/cnn.cu:357-361); its device half launches
    // CHW2CHW_VECT_C_QUANT, which exists nowhere in the reference
    for (CovLayer *L : {C1, C2}) {
        if (L->mul <= 0) return unsupported("ConcatLayer::concat_static without mul / shift");
        L->step_y = ((1 << L->shift) + L->mul / 2) / L->mul;
        L->max_u = 127 * L->step_y;
    }
    return unsupported("ConcatLayer::concat_static (no body in the reference; concat_blu is the static concat)");
}

int ConcatLayer::concat(CovLayer *C1, CovLayer *C2, void *workspace)
{
    convtype max1, max2;
    findMax((convtype *)C1->u, (convtype *)workspace, C1->uSize, &max1);
    findMax((convtype *)C2->u, (convtype *)workspace, C2->uSize, &max2);
    return concat_fix(C1, C2, max1, max2);
}

int ConcatLayer::concat_fix(CovLayer *C1, CovLayer *C2, convtype max1, convtype max2)
{
    int stepy1, stepy2;
    C1->max_u = max1;
    C2->max_u = max2;
    check(qcnn_concat_steps(max1, max2, C1->step_w, C2->step_w, &stepy1, &stepy2));  // inference/cnn.cu:295-307
    C1->step_y = stepy1;
    C2->step_y = stepy2;
    const size_t hw = (size_t)height * width;
    for (int i = 0; i < batch; i++) {
        check(qcnn_op_quant_div((convtype *)C1->u + i * inChannel1 * hw, (xwtype *)conc + i * outChannel * hw, hw,
                                inChannel1, stepy1, nullptr));
        check(qcnn_op_quant_div((convtype *)C2->u + i * inChannel2 * hw,
                                (xwtype *)conc + i * outChannel * hw + inChannel1 * hw, hw, inChannel2, stepy2, nullptr));
    }
    return 0;
}

ConcatLayer::~ConcatLayer()
{
    if (owns_) qcnn_dev_free(conc);
}

// ---------------------------------------------------------------------------------------------
InputLayer::InputLayer(void)
    : batch(0), height(0), width(0), inChannel(0), outChannel(0), inSize(0), outSize(0), x(nullptr), x_ppro(nullptr),
      x_rec(nullptr), owns_(false)
{
}

int InputLayer::adopt(int batch_, int channel, int height_, int width_, void *x_, void *x_ppro_, void *x_rec_)
{
    batch = batch_;
    height = height_;
    width = width_;
    inChannel = channel;
    inSize = batch * height * width * channel;
    outChannel = pad4(channel);                  // XWTYPE == INT8x4, inference/cnn.cu:421-425
    outSize = batch * height * width * outChannel;
    xDesc = {batch, outChannel, height, width};
    x = x_;
    x_ppro = x_ppro_;
    x_rec = x_rec_;
    owns_ = false;
    return 0;
}

int InputLayer::build(int batch_, int channel, int height_, int width_)
{
    adopt(batch_, channel, height_, width_, nullptr, nullptr, nullptr);
    check(qcnn_dev_alloc(&x, sizeof(datatype) * (size_t)inSize));
    check(qcnn_dev_alloc(&x_rec, sizeof(datatype) * (size_t)inSize));
    check(qcnn_dev_alloc(&x_ppro, sizeof(xwtype) * (size_t)outSize));
    check(qcnn_dev_memset(x_ppro, 0, sizeof(xwtype) * (size_t)outSize));
    owns_ = true;
    return 0;
}

int InputLayer::load(datatype *input)
{
    check(qcnn_dev_h2d(x, input, sizeof(datatype) * (size_t)batch * inChannel * height * width));
    return 0;
}

int InputLayer::ppro(void)
{
    // one launch per (batch, channel) plane as inference/cnn.cu:460-466; channel c lands in lane c&3
    // of vec4 group c>>2.  With CHANNEL == 1 this is a single launch that also zeroes lanes 1..3.
    if (inChannel != 1) check(unsupported("InputLayer::ppro with more than one channel"));
    for (int i = 0; i < batch; i++)
        check(qcnn_op_ppro((datatype *)x + (size_t)i * height * width, (xwtype *)x_ppro + (size_t)i * height * width * 4,
                           (size_t)height * width, nullptr));
    return 0;
}

int InputLayer::applyRes_y(convtype *res, int mul, int shift)
{
    check(qcnn_op_apply_res_y(x, x_rec, res, mul, shift, (size_t)inSize, nullptr));
    return 0;
}

int InputLayer::applyRes(xwtype *res)
{
    check(qcnn_op_apply_res(x, res, x_rec, (size_t)inSize, nullptr));
    return 0;
}

int InputLayer::viewmem(xwtype *res)
{
    // inference/cnn.cu:536-549 copies res, x and x_rec to the host and drops them again (a debugger hook:
    // nothing is printed); the copies are kept, so a bad pointer fails here as it does there
    const size_t n = (size_t)height * width * inChannel * batch;
    std::vector<xwtype> res_h(n);
    std::vector<datatype> x_h(n), rec_h(n);
    check(qcnn_dev_sync());
    check(qcnn_dev_d2h(res_h.data(), res, sizeof(xwtype) * n));
    check(qcnn_dev_d2h(x_h.data(), x, sizeof(datatype) * n));
    check(qcnn_dev_d2h(rec_h.data(), x_rec, sizeof(datatype) * n));
    return 0;
}

InputLayer::~InputLayer(void)
{
    if (owns_) {
        qcnn_dev_free(x);
        qcnn_dev_free(x_ppro);
        qcnn_dev_free(x_rec);
    }
}

// ---------------------------------------------------------------------------------------------
char *HWCN2NCHW_VECT_C_CPU(char *HWCN, int H, int W, int C, int N, int *outSize)
{
    *outSize = N * ((C + 3) / 4) * H * W * 4;
    char *out = new char[*outSize];
    qcnn_hwcn2nchw_vect_c((const int8_t *)HWCN, H, W, C, N, (int8_t *)out);
    return out;
}

int NCHW2NCHW_VECT_C_QUANT_BLU(convtype *before, xwtype *after, int N, int C, int H, int W, int blu, int multiplier,
                               int shifts)
{
    const size_t frameSize = (size_t)C * H * W;
    for (int i = 0; i < N; i++)
        check(qcnn_op_quant_blu(before + i * frameSize, after + i * frameSize, (size_t)H * W, C, blu, multiplier, shifts,
                                nullptr));
    return 0;
}

void findMax(convtype *data, convtype *buffer, int n, convtype *max)
{
    check(qcnn_op_find_max(data, buffer, (size_t)n, max, nullptr));
}

int NCHW2NCHW_VECT_C(convtype *dividend, xwtype *quotient, int divisor, int N, int C, int H, int W)
{
    const size_t frameSize = (size_t)C * H * W;
    for (int i = 0; i < N; i++)
        check(qcnn_op_quant_div(dividend + i * frameSize, quotient + i * frameSize, (size_t)H * W, C, divisor, nullptr));
    return 0;
}
