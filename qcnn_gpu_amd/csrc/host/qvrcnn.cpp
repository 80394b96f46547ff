// class qvrcnn of the reference (inference/qvrcnn.cuh:25-59) over the C ABI: host C++ only.
#include "qvrcnn.h"

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
    // the convolutions are direct / fused and need no workspace; findMax keeps its result in it
    workspaceSize = 256;
    check(qcnn_dev_alloc(&workspace, workspaceSize));
    C4.v = qcnn_tap_device_ptr(e, QCNN_TAP_V_C4);
    for (int k = 0; k < LAYER; ++k) steps.stepw[k] = steps.stepy[k] = 0;
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
    for (int l = 0; l < 6; ++l) {
        check(qcnn_layer_params(engine, l, &L[l]->blu, &L[l]->mul, &L[l]->shift));
        L[l]->b = qcnn_layer_b_device_ptr(engine, l);
    }
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

int qvrcnn::load_para(char *filename)
{
    // inference/qvrcnn.cu:30-46
    if (qcnn_load_para(engine, filename) != QCNN_OK) {
        printf("cannot open model file.\n");
        printf("%s\n", qcnn_last_error());
        exit(1);
    }
    qcnn_dynamic_state_t st;
    check(qcnn_dynamic_state(engine, &st));
    CovLayer *L[6] = {&C1, &C2_1, &C2_2, &C3_1, &C3_2, &C4};
    for (int l = 0; l < 6; ++l) {
        L[l]->step_w = st.step_w[l];
        L[l]->b = qcnn_layer_b_device_ptr(engine, l);
        L[l]->b_adj = qcnn_layer_b_adj_device_ptr(engine, l);
    }
    return 0;
}

int qvrcnn::load_float_para(char *filename)
{
    if (qcnn_load_float(engine, filename) != QCNN_OK) {
        printf("cannot open model file.\n");
        printf("%s\n", qcnn_last_error());
        exit(1);
    }
    return 0;
}

int qvrcnn::forward_float(void)
{
    check(qcnn_forward_float(engine));
    return 0;
}

int qvrcnn::forward(void)
{
    check(qcnn_forward(engine));
    qcnn_dynamic_state_t st;
    check(qcnn_dynamic_state(engine, &st));
    CovLayer *L[6] = {&C1, &C2_1, &C2_2, &C3_1, &C3_2, &C4};
    for (int l = 0; l < 6; ++l) L[l]->step_y = st.step_y[l], L[l]->max_u = st.max_u[l];
    for (int k = 0; k < LAYER; ++k) steps.stepw[k] = st.stepw[k], steps.stepy[k] = st.stepy[k];
    return 0;
}

int qvrcnn::forward_layers(void)
{
    // the statement sequence of inference/qvrcnn.cu:82-167, with the calls of :98, :121, :140 and :156
    int layer = 0;
    check(qcnn_dev_sync());
    I1.ppro();
    layer = 1;
    adjustBasic(steps, (btype *)C1.b, (btype *)C1.b_adj, C1.outChannel, layer - 1);
    C1.ConvForward(cudnnHandle, I1.xDesc, I1.x_ppro, workspace, workspaceSize);
    C1.quantize_out(workspace);
    insert_w(C1.step_w, layer);
    insert_y(C1.step_y, layer);
    layer = 2;
    adjustBasic(steps, (btype *)C2_1.b, (btype *)C2_1.b_adj, C2_1.outChannel, layer - 1);
    adjustBasic(steps, (btype *)C2_2.b, (btype *)C2_2.b_adj, C2_2.outChannel, layer - 1);
    C2_1.ConvForward(cudnnHandle, C1.vDesc, C1.v, workspace, workspaceSize);
    C2_1.activate(cudnnHandle);
    C2_2.ConvForward(cudnnHandle, C1.vDesc, C1.v, workspace, workspaceSize);
    C2_2.activate(cudnnHandle);
    Conc1.concat(&C2_1, &C2_2, workspace);
    insert_w(C2_1.step_w, layer);
    insert_y(C2_1.step_y, layer);
    layer = 3;
    adjustBasic(steps, (btype *)C3_1.b, (btype *)C3_1.b_adj, C3_1.outChannel, layer - 1);
    adjustBasic(steps, (btype *)C3_2.b, (btype *)C3_2.b_adj, C3_2.outChannel, layer - 1);
    C3_1.ConvForward(cudnnHandle, Conc1.concDesc, Conc1.conc, workspace, workspaceSize);
    C3_1.activate(cudnnHandle);
    C3_2.ConvForward(cudnnHandle, Conc1.concDesc, Conc1.conc, workspace, workspaceSize);
    C3_2.activate(cudnnHandle);
    Conc2.concat(&C3_1, &C3_2, workspace);
    insert_w(C3_1.step_w, layer);
    insert_y(C3_1.step_y, layer);
    layer = 4;
    adjustBasic(steps, (btype *)C4.b, (btype *)C4.b_adj, C4.outChannel, layer - 1);
    C4.ConvForward(cudnnHandle, Conc2.concDesc, Conc2.conc, workspace, workspaceSize);
    insert_w(C4.step_w, layer);
    adjustOutput(steps, (convtype *)C4.u, (xwtype *)C4.v, layer, C4.uSize);
    I1.applyRes((xwtype *)C4.v);
    check(qcnn_dev_sync());
    return 0;
}

int qvrcnn::forward_shipped(void)
{
    // the statement sequence of inference/qvrcnn.cu:82-167 exactly as shipped (the calls of :100, :122, :141, :157)
    int layer = 0;
    check(qcnn_dev_sync());
    I1.ppro();
    layer = 1;
    adjustBasic(steps, (btype *)C1.b, (btype *)C1.b_adj, C1.outChannel, layer - 1);
    C1.ConvForward(cudnnHandle, I1.xDesc, I1.x_ppro, workspace, workspaceSize);
    C1.quantize_out_static();
    insert_w(C1.step_w, layer);
    insert_y(C1.step_y, layer);
    layer = 2;
    adjustBasic(steps, (btype *)C2_1.b, (btype *)C2_1.b_adj, C2_1.outChannel, layer - 1);
    adjustBasic(steps, (btype *)C2_2.b, (btype *)C2_2.b_adj, C2_2.outChannel, layer - 1);
    C2_1.ConvForward(cudnnHandle, C1.vDesc, C1.v, workspace, workspaceSize);
    C2_1.activate(cudnnHandle);
    C2_2.ConvForward(cudnnHandle, C1.vDesc, C1.v, workspace, workspaceSize);
    C2_2.activate(cudnnHandle);
    Conc1.concat_blu(&C2_1, &C2_2);
    insert_w(C2_1.step_w, layer);
    insert_y(C2_1.step_y, layer);
    layer = 3;
    adjustBasic(steps, (btype *)C3_1.b, (btype *)C3_1.b_adj, C3_1.outChannel, layer - 1);
    adjustBasic(steps, (btype *)C3_2.b, (btype *)C3_2.b_adj, C3_2.outChannel, layer - 1);
    C3_1.ConvForward(cudnnHandle, Conc1.concDesc, Conc1.conc, workspace, workspaceSize);
    C3_1.activate(cudnnHandle);
    C3_2.ConvForward(cudnnHandle, Conc1.concDesc, Conc1.conc, workspace, workspaceSize);
    C3_2.activate(cudnnHandle);
    Conc2.concat_blu(&C3_1, &C3_2);
    insert_w(C3_1.step_w, layer);
    insert_y(C3_1.step_y, layer);
    layer = 4;
    adjustBasic(steps, (btype *)C4.b, (btype *)C4.b_adj, C4.outChannel, layer - 1);
    C4.ConvForward(cudnnHandle, Conc2.concDesc, Conc2.conc, workspace, workspaceSize);
    insert_w(C4.step_w, layer);
    adjustOutput_static((convtype *)C4.u, (xwtype *)C4.v, 141, 16, C4.uSize);
    I1.applyRes((xwtype *)C4.v);
    check(qcnn_dev_sync());
    return 0;
}

int qvrcnn::backward(void)
{
    printf("qvrcnn::backward: not available (declared in inference/qvrcnn.cuh:33, defined nowhere)\n");
    return QCNN_E_UNSUPPORTED;
}

int qvrcnn::save_b_adj(const char *filename)
{
    // inference/qvrcnn.cu:288-304 appends the six b_adj arrays; they live in device memory, so they are
    // copied to the host first
    FILE *fp = fopen(filename, "a+");
    if (!fp) {
        printf("failed to open file %s", filename);
        exit(1);
    }
    CovLayer *L[6] = {&C1, &C2_1, &C2_2, &C3_1, &C3_2, &C4};
    check(qcnn_dev_sync());
    for (int l = 0; l < 6; ++l) {
        btype tmp[64];
        check(qcnn_dev_d2h(tmp, L[l]->b_adj, sizeof(btype) * (size_t)L[l]->outChannel));
        fwrite(tmp, sizeof(btype), (size_t)L[l]->outChannel, fp);
    }
    fclose(fp);
    return 0;
}

int adjustBasic(step_parameters steps, btype *b, btype *b_adj, int count, int n)
{
    check(qcnn_op_adjust_basic(steps.stepw, steps.stepy, b, b_adj, count, n, nullptr));
    return 0;
}

int adjustOutput(step_parameters steps, convtype *o, xwtype *o_adj, int n, int num)
{
    check(qcnn_op_adjust_output(steps.stepw, steps.stepy, o, o_adj, n, (size_t)num, nullptr));
    return 0;
}

int adjustOutput_static(convtype *o, xwtype *o_adj, int multiplier, int shifts, int num)
{
    check(qcnn_op_adjust_output_static(o, o_adj, multiplier, shifts, (size_t)num, nullptr));
    return 0;
}

// Sorted insertion into the first `layer` entries of a step list (what insert_w / insert_y of
// inference/qvrcnn.cu:305-330 maintain): entries 0 .. layer-2 are in order already, the new value goes in
// front of the first entry it should precede and the rest move up by one.
template <typename Before>
static void insert_sorted(int *list, int layer, int value, Before before)
{
    int at = 0;
    while (at < layer - 1 && !before(value, list[at])) ++at;
    for (int k = layer - 1; k > at; --k) list[k] = list[k - 1];
    list[at] = value;
}

// stepw is kept in descending order ...
void qvrcnn::insert_w(int stepw, int layer)
{
    insert_sorted(steps.stepw, layer, stepw, [](int v, int e) { return e < v; });
}

// ... and stepy in ascending order
void qvrcnn::insert_y(int stepy, int layer)
{
    insert_sorted(steps.stepy, layer, stepy, [](int v, int e) { return e > v; });
}

qvrcnn::~qvrcnn()
{
    qcnn_dev_free(workspace);
    qcnn_destroy(engine);
}

int model_qfp_HWCN2NCHW_VECT_C(const char *filein, const char *fileout, int qp)
{
    if (qcnn_model_qfp_hwcn2nchw_vect_c(filein, fileout, qp) != QCNN_OK) {
        printf("%s", qcnn_last_error());  // "failed to open file %s", inference/qvrcnn.cu:565,571
        exit(1);
    }
    return 0;
}
