// Exercises the reference-shaped C++ API (include/qvrcnn.h, cnn.h) the way the reference's own
// driver and test_layer do (inference/kernel.cu:28-116), and writes raw outputs for the Python
// test to compare with the oracle:
//   <out>.fused   x_rec of qvrcnn::forward_blu           (fused kernels)
//   <out>.layers  x_rec of qvrcnn::forward_blu_layers    (layer objects, reference call order)
//   <out>.c1v     C1.v of a stand-alone InputLayer + CovLayer pair that own their buffers
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
    printf("psnr before %.6f after %.6f\n", d.psnr(d.input), 0.0);
    return 0;
}
