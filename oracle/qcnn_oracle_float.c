/*
 * CPU restatement of the reference's float32 configuration (FLOAT_CONFIG, inference/mat.cuh:89-101).
 *
 * TEST INFRASTRUCTURE ONLY: called by tests/ and __graft_entry__.smoke(), never by the product library.
 * PARITY UNPINNED: the reference ships no float model and no output of this path, and its convolution is
 * cuDNN's (summation order unknown).  Convolutions here accumulate in double and round once, i.e. they
 * return the value every float32 summation order approximates; tests compare within a stated tolerance.
 *
 *   orf_ppro        FLOAT_PPRO                       inference/cnn.cu:470-484
 *   orf_conv        CovLayer::ConvForward + activate inference/cnn.cu:129-144, 161-166 (descriptor :44-49:
 *                   cross-correlation, zero padding (ks-1)/2, stride 1; filter [K][C][R][S] :33)
 *   orf_apply_res   applyRes_GPU, float arm          inference/cnn.cu:487-502
 *   orf_forward     qvrcnn::forward                  inference/qvrcnn.cu:244-285
 *   orf_hwcn2nchw   HWCN2NCHW_CPU                    inference/mat.cu:160-176
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static const int F_KS[6] = {5, 3, 5, 3, 1, 3};
static const int F_CIN[6] = {1, 64, 64, 48, 48, 48};
static const int F_COUT[6] = {64, 32, 16, 16, 32, 1};

long orf_model_bytes(void)
{
    long n = 0;
    for (int l = 0; l < 6; ++l) n += 4L * (F_KS[l] * F_KS[l] * F_CIN[l] * F_COUT[l] + F_COUT[l]);
    return n;
}

void orf_ppro(const uint8_t *x, float *out, long n)
{
    for (long i = 0; i < n; ++i) out[i] = ((float)x[i] - 128) / 255; /* cnn.cu:476 */
}

/* relu: 0 none, 1 max(u, 0).  acc32 != 0: accumulate in float32 in the order c, r, s (one plausible
 * order, for the summation-order sensitivity the tests report); else in double. */
void orf_conv(const float *x, const float *w, const float *b, float *u, int H, int W, int cin, int cout, int ks,
              int relu, int acc32)
{
    const int p = (ks - 1) / 2;
    const long HW = (long)H * W;
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < cout; ++k)
        for (int y = 0; y < H; ++y)
            for (int xx = 0; xx < W; ++xx) {
                double acc = 0.0;
                float accf = 0.f;
                for (int c = 0; c < cin; ++c)
                    for (int r = 0; r < ks; ++r) {
                        const int iy = y + r - p;
                        if (iy < 0 || iy >= H) continue;
                        for (int s = 0; s < ks; ++s) {
                            const int ix = xx + s - p;
                            if (ix < 0 || ix >= W) continue;
                            const float wv = w[(((long)k * cin + c) * ks + r) * ks + s];
                            const float xv = x[c * HW + (long)iy * W + ix];
                            if (acc32) accf += wv * xv;
                            else acc += (double)wv * (double)xv;
                        }
                    }
                float v = (acc32 ? accf : (float)acc) + b[k]; /* cudnnAddTensor, cnn.cu:141 */
                if (relu && v < 0.f) v = 0.f;
                u[k * HW + (long)y * W + xx] = v;
            }
}

uint8_t orf_apply_res_one(uint8_t x, float res)
{
    /* rec_int = x[tid] + res[tid] * 255 + 0.5 (cnn.cu:496): float product and sum -- fused into one
     * multiply-add, the reference's device compiler's default contraction -- then a double 0.5, then
     * truncation to short */
    const double t = (double)fmaf(res, 255.f, (float)x) + 0.5;
    int r;
    if (!(t > -32768.0)) r = -32768;
    else if (t > 32767.0) r = 32767;
    else r = (int)t;
    return (uint8_t)(r > 255 ? 255 : r < 0 ? 0 : r);
}

void orf_apply_res(const uint8_t *x, const float *res, uint8_t *rec, long n)
{
    for (long i = 0; i < n; ++i) rec[i] = orf_apply_res_one(x[i], res[i]);
}

/* taps (any may be null): xp [HW], u1 [64 HW], conc1 [48 HW], conc2 [48 HW], u4 [HW] */
int orf_forward(const void *model, long model_bytes, const uint8_t *y, int H, int W, uint8_t *rec, float *t_xp,
                float *t_u1, float *t_conc1, float *t_conc2, float *t_u4)
{
    if (model_bytes < orf_model_bytes()) return -1;
    const long HW = (long)H * W;
    const float *wp[6], *bp[6];
    const float *p = (const float *)model;
    for (int l = 0; l < 6; ++l) {
        wp[l] = p;
        p += F_KS[l] * F_KS[l] * F_CIN[l] * F_COUT[l];
        bp[l] = p;
        p += F_COUT[l];
    }
    float *xp = malloc(sizeof(float) * HW), *u1 = malloc(sizeof(float) * 64 * HW);
    float *c1 = malloc(sizeof(float) * 48 * HW), *c2 = malloc(sizeof(float) * 48 * HW), *u4 = malloc(sizeof(float) * HW);
    if (!xp || !u1 || !c1 || !c2 || !u4) return -2;
    orf_ppro(y, xp, HW);
    orf_conv(xp, wp[0], bp[0], u1, H, W, 1, 64, 5, 1, 0);
    orf_conv(u1, wp[1], bp[1], c1, H, W, 64, 32, 3, 1, 0);
    orf_conv(u1, wp[2], bp[2], c1 + 32 * HW, H, W, 64, 16, 5, 1, 0);
    orf_conv(c1, wp[3], bp[3], c2, H, W, 48, 16, 3, 1, 0);
    orf_conv(c1, wp[4], bp[4], c2 + 16 * HW, H, W, 48, 32, 1, 1, 0);
    orf_conv(c2, wp[5], bp[5], u4, H, W, 48, 1, 3, 0, 0);
    orf_apply_res(y, u4, rec, HW);
    if (t_xp) memcpy(t_xp, xp, sizeof(float) * HW);
    if (t_u1) memcpy(t_u1, u1, sizeof(float) * 64 * HW);
    if (t_conc1) memcpy(t_conc1, c1, sizeof(float) * 48 * HW);
    if (t_conc2) memcpy(t_conc2, c2, sizeof(float) * 48 * HW);
    if (t_u4) memcpy(t_u4, u4, sizeof(float) * HW);
    free(xp), free(u1), free(c1), free(c2), free(u4);
    return 0;
}

long orf_hwcn2nchw(const float *hwcn, int H, int W, int C, int N, float *out)
{
    for (int i = 0; i < H; ++i)
        for (int j = 0; j < W; ++j)
            for (int k = 0; k < C; ++k)
                for (int l = 0; l < N; ++l)
                    out[(long)l * C * H * W + (long)k * H * W + i * W + j] = hwcn[(long)i * W * C * N + (long)j * C * N + k * N + l];
    return (long)N * C * H * W;
}
