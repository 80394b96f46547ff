// Exercises the reference-shaped C++ API (include/qvrcnn.h, cnn.h) the way the reference's own
// driver and test_layer do (inference/kernel.cu:28-116), and writes raw outputs for the Python
// test to compare with the oracle:
//   <out>.fused   x_rec of qvrcnn::forward_blu           (fused kernels)
//   <out>.layers  x_rec of qvrcnn::forward_blu_layers    (layer objects, reference call order)
//   <out>.c1v     C1.v of a stand-alone InputLayer + CovLayer pair that own their buffers
// and, when a dynamic-quantisation model is given as 7th argument (inference/qvrcnn.cu:82-167):
//   <out>.dyn         x_rec of qvrcnn::forward           (one asynchronous kernel sequence)
//   <out>.dynlayers   x_rec of qvrcnn::forward_layers    (layer objects, host round trips)
//   <out>.b_adj       the file qvrcnn::save_b_adj appends
//   <out>.c1dyn       C1.v of a stand-alone CovLayer through load_para / ConvForward / quantize_out
//   <out>.shipped     x_rec of qvrcnn::forward_shipped   (the literal text of inference/qvrcnn.cu:82-167)
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>

#include "qvrcnn.h"
#include "yuv_data.h"

static void dump(const std::string &fn, const void *dev, size_t n)
{
    std::vector<unsigned char> h(n);
    check(qcnn_dev_sync());
    check(qcnn_dev_d2h(h.data(), dev, n));
    FILE *f = fopen(fn.c_str(), "wb");
    fwrite(h.data(), 1, n, f);
    fclose(f);
}

int main(int argc, char **argv)
{
    if (argc < 7) return 2;
    char *model = argv[1], *ori = argv[2], *anchor = argv[3];
    const int H = atoi(argv[4]), W = atoi(argv[5]);
    const std::string out = argv[6];
    vrcnn_data d(1, H, W);
    d.read_data(ori, anchor);

    qvrcnn net(0, 1, CHANNEL, H, W);
    net.load_static_para(model);
    net.load_data(d.input);
    net.forward_blu();
    check(qcnn_sync(net.engine));
    dump(out + ".fused", net.I1.x_rec, (size_t)H * W);
    check(qcnn_dev_memset(net.I1.x_rec, 0, (size_t)H * W));
    net.forward_blu_layers();
    dump(out + ".layers", net.I1.x_rec, (size_t)H * W);
    printf("C1 blu=%d mul=%d shift=%d  C4 mul=%d shift=%d\n", net.C1.blu, net.C1.mul, net.C1.shift, net.C4.mul, net.C4.shift);

    // stand-alone layers, as test_layer builds them (inference/kernel.cu:44-52)
    struct qcnnHandle_s hs = {nullptr};
    InputLayer I1;
    CovLayer C1;
    I1.build(1, CHANNEL, H, W);
    C1.build(&hs, I1.xDesc, 1, H, W, CHANNEL, 64, 5);
    FILE *fp = fopen(model, "rb");
    C1.load_static_para(fp);
    fclose(fp);
    I1.load(d.input);
    I1.ppro();
    C1.ConvForward_static(&hs, I1.xDesc, I1.x_ppro, nullptr, 0);
    C1.quantize_out_blu();
    dump(out + ".c1v", C1.v, (size_t)64 * H * W);
    if (getenv("FACADE_VIEWMEM")) C1.viewmem((xwtype *)I1.x_ppro);
    {
        // members the reference declares and never (or only in a comment) defines: inference/cnn.cuh:33-34,69,89,
        // inference/qvrcnn.cuh:33
        const int fm = C1.freeMem(), fu = C1.u == nullptr && C1.v == nullptr;
        const int sm = C1.setMem(), su = C1.u != nullptr && C1.v != nullptr;
        C1.quantize_out_blu();  // the buffers are usable again (u holds garbage now, the call must not fault)
        ConcatLayer cc;
        const int cs = cc.concat_static(&net.C2_1, &net.C2_2);
        const int vm = I1.viewmem((xwtype *)I1.x_rec);
        printf("declared-only members: freeMem %d (%d) setMem %d (%d) adopted freeMem %d concat_static %d step_y %d viewmem %d "
               "backward %d\n",
               fm, fu, sm, su, net.C1.freeMem(), cs, net.C2_1.step_y, vm, net.backward());
    }

    if (argc > 7) {
        char *dmodel = argv[7];
        net.load_para(dmodel);
        net.load_data(d.input);
        net.forward();
        dump(out + ".dyn", net.I1.x_rec, (size_t)H * W);
        printf("dyn steps w %d %d %d %d y %d %d %d %d\n", net.steps.stepw[0], net.steps.stepw[1], net.steps.stepw[2],
               net.steps.stepw[3], net.steps.stepy[0], net.steps.stepy[1], net.steps.stepy[2], net.steps.stepy[3]);
        printf("dyn step_y %d %d %d %d %d max_u %d %d\n", net.C1.step_y, net.C2_1.step_y, net.C2_2.step_y, net.C3_1.step_y,
               net.C3_2.step_y, net.C1.max_u, net.C3_2.max_u);
        remove((out + ".b_adj").c_str());
        net.save_b_adj((out + ".b_adj").c_str());
        check(qcnn_dev_memset(net.I1.x_rec, 0, (size_t)H * W));
        for (int k = 0; k < LAYER; ++k) net.steps.stepw[k] = net.steps.stepy[k] = 0;
        net.forward_layers();
        dump(out + ".dynlayers", net.I1.x_rec, (size_t)H * W);
        printf("layers steps w %d %d %d %d y %d %d %d %d\n", net.steps.stepw[0], net.steps.stepw[1], net.steps.stepw[2],
               net.steps.stepw[3], net.steps.stepy[0], net.steps.stepy[1], net.steps.stepy[2], net.steps.stepy[3]);

        // stand-alone dynamic layer: load_para, ConvForward with b_adj, quantize_out
        CovLayer D1;
        D1.build(&hs, I1.xDesc, 1, H, W, CHANNEL, 64, 5);
        fp = fopen(dmodel, "rb");
        D1.load_para(fp);
        fclose(fp);
        step_parameters none = {{0, 0, 0, 0}, {0, 0, 0, 0}};
        adjustBasic(none, (btype *)D1.b, (btype *)D1.b_adj, D1.outChannel, 0);
        void *ws = nullptr;
        check(qcnn_dev_alloc(&ws, 256));
        D1.ConvForward(&hs, I1.xDesc, I1.x_ppro, ws, 256);
        D1.quantize_out(ws);
        dump(out + ".c1dyn", D1.v, (size_t)64 * H * W);
        printf("standalone step_w %d step_y %d max_u %d\n", D1.step_w, D1.step_y, D1.max_u);
        check(qcnn_dev_free(ws));

        // qvrcnn::forward as its shipped text reads (inference/qvrcnn.cu:100,122,141,157): the layer members still hold
        // blu / mul / shift of the static model loaded first; C1.step_y is set by hand, as the text needs
        check(qcnn_dev_memset(net.I1.x_rec, 0, (size_t)H * W));
        for (int k = 0; k < LAYER; ++k) net.steps.stepw[k] = net.steps.stepy[k] = 0;
        net.C1.step_y = 37;
        net.forward_shipped();
        dump(out + ".shipped", net.I1.x_rec, (size_t)H * W);
        printf("shipped steps w %d %d %d %d y %d %d %d %d step_y %d %d %d %d\n", net.steps.stepw[0], net.steps.stepw[1],
               net.steps.stepw[2], net.steps.stepw[3], net.steps.stepy[0], net.steps.stepy[1], net.steps.stepy[2], net.steps.stepy[3],
               net.C2_1.step_y, net.C2_2.step_y, net.C3_1.step_y, net.C3_2.step_y);
    }
    printf("psnr before %.6f after %.6f\n", d.psnr(d.input), 0.0);
    return 0;
}
