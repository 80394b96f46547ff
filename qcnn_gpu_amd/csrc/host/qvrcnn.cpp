// class qvrcnn of the reference (inference/qvrcnn.cuh:25-59) over the C ABI: host C++ only.
#include "qvrcnn.h"

static int unsupported(const char *what)
{
    printf("%s: only the static-quantised (blu) path is implemented\n", what);
    return QCNN_E_UNSUPPORTED;
}

qvrcnn::qvrcnn(int gpu_num, int batch, int channel, int height, int width)
{
    // inference/qvrcnn.cu:4-29: select the device, build the graph, allocate every buffer once
    check(qcnn_set_device(gpu_num));
    engine = nullptr;
    check(qcnn_create(gpu_num, batch, channel, height, width, &engine));
    handle_storage_.stream = nullptr;  // layer-level calls run on the default stream, as the reference's do
    cudnnHandle = &handle_storage_;
    this->batch = batch, this->channel = channel, this->height = height, this->width = width;
    layers = LAYER;
    qcnn_t *e = engine;
    I1.adopt(batch, channel, height, width, qcnn_input_device_ptr(e), qcnn_tap_device_ptr(e, QCNN_TAP_X_PPRO),
             qcnn_recon_device_ptr(e));
    auto conv = [&](CovLayer &L, int l, int inC, int outC, int ks, int utap, void *v) {
        L.adopt(batch, height, width, inC, outC, ks, qcnn_layer_w_device_ptr(e, l), qcnn_layer_b_device_ptr(e, l),
                qcnn_tap_device_ptr(e, utap), v);
    };
    // layer table of inference/qvrcnn.cu:11-18
    conv(C1, 0, I1.outChannel, 64, 5, QCNN_TAP_U_C1, qcnn_tap_device_ptr(e, QCNN_TAP_V_C1));
    C1.inChannel = 1;  // lanes 1..3 of x_ppro carry no data; the file still stores C1 weights padded to 4
    conv(C2_1, 1, C1.outChannel, 32, 3, QCNN_TAP_U_C2_1, nullptr);
    conv(C2_2, 2, C1.outChannel, 16, 5, QCNN_TAP_U_C2_2, nullptr);
    Conc1.adopt(batch, height, width, C2_1.outChannel, C2_2.outChannel, qcnn_tap_device_ptr(e, QCNN_TAP_CONC1));
    conv(C3_1, 3, Conc1.outChannel, 16, 3, QCNN_TAP_U_C3_1, nullptr);
    conv(C3_2, 4, Conc1.outChannel, 32, 1, QCNN_TAP_U_C3_2, nullptr);
    Conc2.adopt(batch, height, width, C3_1.outChannel, C3_2.outChannel, qcnn_tap_device_ptr(e, QCNN_TAP_CONC2));
    conv(C4, 5, Conc2.outChannel, 1, 3, QCNN_TAP_U_C4, nullptr);
    if (!I1.x || !I1.x_ppro || !C1.u || !C4.u || !Conc1.conc || !Conc2.conc) check(QCNN_E_HIP);
    workspaceSize = 0;  // the convolutions are direct / fused: no workspace
    workspace = nullptr;
}

int qvrcnn::load_static_para(char *filename)
{
    // inference/qvrcnn.cu:47-63; the engine parses the same file once for both paths
    if (qcnn_load_static(engine, filename) != QCNN_OK) {
        printf("cannot open model file.\n");
        printf("%s\n", qcnn_last_error());
        exit(1);
    }
    CovLayer *L[6] = {&C1, &C2_1, &C2_2, &C3_1, &C3_2, &C4};
    for (int l = 0; l < 6; ++l) check(qcnn_layer_params(engine, l, &L[l]->blu, &L[l]->mul, &L[l]->shift));
    return 0;
}

int qvrcnn::load_data(datatype *input)
{
    check(qcnn_upload(engine, input));  // blocking, as InputLayer::load (inference/cnn.cu:441)
    return 0;
}

int qvrcnn::forward_blu(void)
{
    check(qcnn_forward_blu(engine));
    return 0;
}

int qvrcnn::forward_blu_layers(void)
{
    // the statement sequence of inference/qvrcnn.cu:168-242
    check(qcnn_dev_sync());
    I1.ppro();
    C1.ConvForward_static(cudnnHandle, I1.xDesc, I1.x_ppro, workspace, workspaceSize);
    C1.quantize_out_blu();
    C2_1.ConvForward_static(cudnnHandle, C1.vDesc, C1.v, workspace, workspaceSize);
    C2_2.ConvForward_static(cudnnHandle, C1.vDesc, C1.v, workspace, workspaceSize);
    Conc1.concat_blu(&C2_1, &C2_2);
    C3_1.ConvForward_static(cudnnHandle, Conc1.concDesc, Conc1.conc, workspace, workspaceSize);
    C3_2.ConvForward_static(cudnnHandle, Conc1.concDesc, Conc1.conc, workspace, workspaceSize);
    Conc2.concat_blu(&C3_1, &C3_2);
    C4.ConvForward_static(cudnnHandle, Conc2.concDesc, Conc2.conc, workspace, workspaceSize);
    I1.applyRes_y((convtype *)C4.u, C4.mul, C4.shift);
    check(qcnn_dev_sync());
    return 0;
}

int qvrcnn::load_para(char *) { check(unsupported("qvrcnn::load_para")); return 0; }
int qvrcnn::forward(void) { check(unsupported("qvrcnn::forward")); return 0; }
int qvrcnn::save_b_adj(const char *) { check(unsupported("qvrcnn::save_b_adj")); return 0; }

void qvrcnn::insert_w(int stepw, int layer)
{
    // inference/qvrcnn.cu:305-317 (bookkeeping of the dynamic path; kept because it is host-only)
    int i, j;
    for (i = 0; i < layer - 1; i++)
        if (steps.stepw[i] < stepw) {
            for (j = layer - 1; j > i; j--) steps.stepw[j] = steps.stepw[j - 1];
            steps.stepw[j] = stepw;
            return;
        }
    steps.stepw[i] = stepw;
}

void qvrcnn::insert_y(int stepy, int layer)
{
    int i, j;
    for (i = 0; i < layer - 1; i++)
        if (steps.stepy[i] > stepy) {
            for (j = layer - 1; j > i; j--) steps.stepy[j] = steps.stepy[j - 1];
            steps.stepy[j] = stepy;
            return;
        }
    steps.stepy[i] = stepy;
}

qvrcnn::~qvrcnn() { qcnn_destroy(engine); }

int model_qfp_HWCN2NCHW_VECT_C(const char *filein, const char *fileout, int qp)
{
    if (qcnn_model_qfp_hwcn2nchw_vect_c(filein, fileout, qp) != QCNN_OK) {
        printf("%s", qcnn_last_error());  // "failed to open file %s", inference/qvrcnn.cu:565,571
        exit(1);
    }
    return 0;
}
