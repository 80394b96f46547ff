/*
 * CPU oracle for the dynamic-quantisation pass qvrcnn::forward (inference/qvrcnn.cu:82-167).
 *
 * TEST INFRASTRUCTURE ONLY (same rules as qcnn_oracle.c: only tests/, smoke() and the bench's CPU
 * leg may call it, and only as the checker).  PARITY UNPINNED BY THE REFERENCE: no vectors exist
 * for this path; it is pinned by hand-derived known answers (tests/test_dynamic_kat.py) and by
 * agreement with the independent numpy restatement oracle/np_dynamic.py.
 *
 * Restated functions:
 *   CovLayer::load_para                 inference/cnn.cu:69-89     (file: step_w, weights, biases)
 *   adjustBasic                         inference/qvrcnn.cu:336-349
 *   CovLayer::ConvForward (+ b_adj)     inference/cnn.cu:129-144
 *   CovLayer::activate (ReLU)           inference/cnn.cu:161-166, :57
 *   findMax                             inference/mat.cu:28-96
 *   CovLayer::quantize_out              inference/cnn.cu:169-178
 *   CHW2CHW_VECT_C                      inference/mat.cu:197-236
 *   ConcatLayer::concat                 inference/cnn.cu:285-320
 *   mul_shift                           inference/mat.cu:248-261
 *   qvrcnn::insert_w / insert_y         inference/qvrcnn.cu:305-330
 *   adjustOutput / adjustOutput_static  inference/qvrcnn.cu:368-397
 *   applyRes_GPU                        inference/cnn.cu:487-502
 *
 * forward() as shipped calls quantize_out_static, concat_blu and adjustOutput_static(141, 16), which
 * read mul / shift / blu that load_para never sets; the pass restated here uses the alternatives the
 * source keeps beside them (quantize_out at :98, concat at :121/:140, adjustOutput at :156).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define THRESHOLD 127 /* inference/mat.cuh:72 */

/* from qcnn_oracle.c */
void orc_ppro(const uint8_t *x, int8_t *x_ppro, long n);
void orc_conv_static(const int8_t *in, int cin, int H, int W, const int8_t *w, const int32_t *b_int, int cout, int ks,
                     float *u);
int orc_layer_wsize(int l);

static const int DKS[6] = {5, 3, 5, 3, 1, 3};
static const int DCIN[6] = {1, 64, 64, 48, 48, 48};
static const int DCOUT[6] = {64, 32, 16, 16, 32, 1};
static const int DBOFF[6] = {0, 64, 96, 112, 128, 160};

long orc_dyn_model_bytes(void)
{
    long n = 0;
    for (int l = 0; l < 6; ++l) n += 4 + orc_layer_wsize(l) + 4L * DCOUT[l];
    return n;
}

float orc_find_max(const float *u, long n)
{
    float m = 0;
    for (long i = 0; i < n; ++i) {
        const float a = fabsf(u[i]);
        if (m < a) m = a;
    }
    return m;
}

int8_t orc_quant_div_one(float temp, int divisor)
{
    if (temp < 0) {
        temp = (temp - (divisor >> 1)) / divisor;
        return temp < -128 ? (int8_t)-128 : (int8_t)temp;
    }
    temp = (temp + (divisor >> 1)) / divisor;
    return temp > 127 ? (int8_t)127 : (int8_t)temp;
}

void orc_quant_div(const float *u, int8_t *out, long HW, int C, int divisor)
{
    for (int c = 0; c < C; ++c)
        for (long p = 0; p < HW; ++p)
            out[(size_t)(c >> 2) * HW * 4 + 4 * p + (c & 3)] = orc_quant_div_one(u[(size_t)c * HW + p], divisor);
}

void orc_mul_shift(const float *in, int8_t *out, long n, int mul, int shift)
{
    const int bias = (1 << (shift - 1)) / mul;
    for (long i = 0; i < n; ++i)
        out[i] = (int8_t)((int32_t)((uint32_t)((int32_t)in[i] + bias) * (uint32_t)mul) >> shift);
}

void orc_adjust_basic(const int *stepw, const int *stepy, const float *b, float *b_adj, int count, int n)
{
    for (int i = 0; i < count; ++i) {
        long long temp = (long long)b[i];
        for (int k = 0; k < n; ++k) temp *= stepw[k];
        for (int k = 0; k < n; ++k) {
            if (temp > 0)
                temp = (temp + (stepy[k] >> 1)) / stepy[k];
            else
                temp = (temp - (stepy[k] >> 1)) / stepy[k];
        }
        b_adj[i] = (float)temp;
    }
}

void orc_adjust_output(const int *stepw, const int *stepy, const float *o, int8_t *o_adj, int n, long num)
{
    for (long i = 0; i < num; ++i) {
        long long temp = (long long)o[i];
        for (int k = 0; k < n - 1; ++k) temp *= stepy[k];
        if (temp > 0)
            for (int k = n - 1; k >= 0; --k) temp = (temp + (stepw[k] >> 1)) / stepw[k];
        else
            for (int k = n - 1; k >= 0; --k) temp = (temp - (stepw[k] >> 1)) / stepw[k];
        o_adj[i] = (int8_t)temp;
    }
}

void orc_adjust_output_static(const float *o, int8_t *o_adj, int mul, int shift, long num)
{
    const uint32_t bias = 1u << (shift - 1);
    for (long i = 0; i < num; ++i)
        o_adj[i] = (int8_t)((int32_t)((uint32_t)(int32_t)o[i] * (uint32_t)mul + bias) >> shift);
}

void orc_apply_res(const uint8_t *x, const int8_t *res, uint8_t *rec, long n)
{
    for (long i = 0; i < n; ++i) {
        const short r = (short)((short)x[i] + res[i]);
        rec[i] = r > 255 ? 255 : r < 0 ? 0 : (uint8_t)r;
    }
}

int orc_quantize_out_step(float max) { return (int)(max / (THRESHOLD + 1) + 1); }

void orc_concat_steps(float max1, float max2, int sw1, int sw2, int *stepy1, int *stepy2)
{
    int s1 = max1 > THRESHOLD ? (int)(max1 / (THRESHOLD + 1) + 1) : 1;
    int s2 = max2 > THRESHOLD ? (int)(max2 / (THRESHOLD + 1) + 1) : 1;
    if (sw1 * s2 > sw2 * s1)
        s1 = (sw1 * s2 + (sw2 >> 1)) / sw2;
    else
        s2 = (sw2 * s1 + (sw1 >> 1)) / sw1;
    *stepy1 = s1;
    *stepy2 = s2;
}

void orc_insert_w(int *stepw, int v, int layer)
{
    int i, j;
    for (i = 0; i < layer - 1; i++)
        if (stepw[i] < v) {
            for (j = layer - 1; j > i; j--) stepw[j] = stepw[j - 1];
            stepw[j] = v;
            return;
        }
    stepw[i] = v;
}

void orc_insert_y(int *stepy, int v, int layer)
{
    int i, j;
    for (i = 0; i < layer - 1; i++)
        if (stepy[i] > v) {
            for (j = layer - 1; j > i; j--) stepy[j] = stepy[j - 1];
            stepy[j] = v;
            return;
        }
    stepy[i] = v;
}

/* u = float(acc) + b_adj, optional ReLU */
static void conv_dyn(const int8_t *in, int l, int H, int W, const int8_t *w, const float *b_adj, int relu, float *u)
{
    static const int32_t zero[64] = {0};
    const long HW = (long)H * W;
    orc_conv_static(in, DCIN[l], H, W, w, zero, DCOUT[l], DKS[l], u); /* (float)acc + 0.0f == (float)acc */
    for (int k = 0; k < DCOUT[l]; ++k)
        for (long p = 0; p < HW; ++p) {
            float v = u[(size_t)k * HW + p] + b_adj[k];
            if (relu && v < 0) v = 0;
            u[(size_t)k * HW + p] = v;
        }
}

typedef struct {
    int8_t *x_ppro;   /* [H][W][4]      */
    float *u[6];      /* [cout][H][W]   (after ReLU where the pass applies it) */
    int8_t *v1;       /* [16][H][W][4]  */
    int8_t *conc1;    /* [12][H][W][4]  */
    int8_t *conc2;    /* [12][H][W][4]  */
    int8_t *v4;       /* [H][W]         */
    float *b_adj;     /* 161            */
    int32_t *max_u;   /* 6              */
    int32_t *step_y;  /* 6              */
    int32_t *stepw;   /* 4              */
    int32_t *stepy;   /* 4              */
} orc_dyn_taps;

int orc_forward_dynamic(const uint8_t *model, long model_len, const uint8_t *y, int H, int W, uint8_t *rec,
                        const orc_dyn_taps *taps)
{
    if (model_len < orc_dyn_model_bytes()) return -1;
    const int8_t *w[6];
    float b[161], b_adj[161];
    int step_w[6], step_y[6] = {0}, max_u[6] = {0};
    const uint8_t *p = model;
    for (int l = 0; l < 6; ++l) {
        int32_t sw;
        memcpy(&sw, p, 4);
        p += 4;
        if (sw < 1) return -2;
        step_w[l] = sw;
        w[l] = (const int8_t *)p;
        p += orc_layer_wsize(l);
        for (int k = 0; k < DCOUT[l]; ++k) {
            int32_t bi;
            memcpy(&bi, p + 4 * k, 4);
            b[DBOFF[l] + k] = (float)bi; /* inference/cnn.cu:81 */
        }
        p += 4 * DCOUT[l];
    }
    const long HW = (long)H * W;
    orc_dyn_taps t;
    memset(&t, 0, sizeof t);
    if (taps) t = *taps;
    int8_t *x_ppro = t.x_ppro ? t.x_ppro : (int8_t *)malloc(4 * HW);
    int8_t *v1 = t.v1 ? t.v1 : (int8_t *)malloc(64 * HW);
    int8_t *conc1 = t.conc1 ? t.conc1 : (int8_t *)malloc(48 * HW);
    int8_t *conc2 = t.conc2 ? t.conc2 : (int8_t *)malloc(48 * HW);
    int8_t *v4 = t.v4 ? t.v4 : (int8_t *)malloc(HW);
    float *u[6];
    for (int l = 0; l < 6; ++l) u[l] = t.u[l] ? t.u[l] : (float *)malloc(sizeof(float) * DCOUT[l] * HW);
    int stepw[4] = {0, 0, 0, 0}, stepy[4] = {0, 0, 0, 0};

    orc_ppro(y, x_ppro, HW);
    /* layer 1 (inference/qvrcnn.cu:90-105) */
    orc_adjust_basic(stepw, stepy, b, b_adj, 64, 0);
    conv_dyn(x_ppro, 0, H, W, w[0], b_adj, 0, u[0]);
    {
        const float mx = orc_find_max(u[0], 64 * HW);
        max_u[0] = (int)mx;
        step_y[0] = orc_quantize_out_step(mx);
        orc_quant_div(u[0], v1, HW, 64, step_y[0]);
    }
    orc_insert_w(stepw, step_w[0], 1);
    orc_insert_y(stepy, step_y[0], 1);
    /* layers 2 and 3 (inference/qvrcnn.cu:108-143) */
    const int8_t *in = v1;
    for (int layer = 2; layer <= 3; ++layer) {
        const int a = layer == 2 ? 1 : 3, c = a + 1;
        int8_t *conc = layer == 2 ? conc1 : conc2;
        orc_adjust_basic(stepw, stepy, b + DBOFF[a], b_adj + DBOFF[a], DCOUT[a], layer - 1);
        orc_adjust_basic(stepw, stepy, b + DBOFF[c], b_adj + DBOFF[c], DCOUT[c], layer - 1);
        conv_dyn(in, a, H, W, w[a], b_adj + DBOFF[a], 1, u[a]);
        conv_dyn(in, c, H, W, w[c], b_adj + DBOFF[c], 1, u[c]);
        const float m1 = orc_find_max(u[a], DCOUT[a] * HW), m2 = orc_find_max(u[c], DCOUT[c] * HW);
        max_u[a] = (int)m1;
        max_u[c] = (int)m2;
        orc_concat_steps(m1, m2, step_w[a], step_w[c], &step_y[a], &step_y[c]);
        orc_quant_div(u[a], conc, HW, DCOUT[a], step_y[a]);
        orc_quant_div(u[c], conc + (size_t)DCOUT[a] * HW, HW, DCOUT[c], step_y[c]);
        orc_insert_w(stepw, step_w[a], layer);
        orc_insert_y(stepy, step_y[a], layer);
        in = conc;
    }
    /* layer 4 and restore (inference/qvrcnn.cu:146-162) */
    orc_adjust_basic(stepw, stepy, b + 160, b_adj + 160, 1, 3);
    conv_dyn(conc2, 5, H, W, w[5], b_adj + 160, 0, u[5]);
    orc_insert_w(stepw, step_w[5], 4);
    orc_adjust_output(stepw, stepy, u[5], v4, 4, HW);
    orc_apply_res(y, v4, rec, HW);

    if (t.b_adj) memcpy(t.b_adj, b_adj, sizeof b_adj);
    if (t.max_u) memcpy(t.max_u, max_u, sizeof max_u);
    if (t.step_y) memcpy(t.step_y, step_y, sizeof step_y);
    if (t.stepw) memcpy(t.stepw, stepw, sizeof stepw);
    if (t.stepy) memcpy(t.stepy, stepy, sizeof stepy);
    if (!t.x_ppro) free(x_ppro);
    if (!t.v1) free(v1);
    if (!t.conc1) free(conc1);
    if (!t.conc2) free(conc2);
    if (!t.v4) free(v4);
    for (int l = 0; l < 6; ++l)
        if (!t.u[l]) free(u[l]);
    return 0;
}

/* qvrcnn::forward exactly as the shipped text reads (inference/qvrcnn.cu:82-167), i.e. with the three static helpers it
 * calls inside the dynamic pass: CovLayer::quantize_out_static (:100 -> mul_shift, inference/mat.cu:248-261: flat, NO
 * relayout, so C1.v holds NCHW bytes that the next convolutions read as NCHW_VECT_C), ConcatLayer::concat_blu (:122, :141,
 * which also sets step_y = ((1 << shift) + mul / 2) / mul, inference/cnn.cu:378-381) and adjustOutput_static(.., 141, 16)
 * (:157).  load_para sets none of blu / mul / shift / C1.step_y: the caller supplies them (q = blu, mul, shift of the six
 * layers; c1_step_y = the value of the member C1.step_y, which insert_y files at :105). */
void orc_quant_blu(const float *u, int8_t *out, long HW, int C, int blu, int mul, int shift);
int orc_forward_shipped(const uint8_t *model, long model_len, const int32_t *q, int c1_step_y, const uint8_t *y, int H, int W,
                        uint8_t *rec, int32_t *stepw_out, int32_t *stepy_out, int32_t *step_y_out)
{
    if (model_len < orc_dyn_model_bytes()) return -1;
    const int8_t *w[6];
    float b[161], b_adj[161];
    int step_w[6], step_y[6] = {0};
    const uint8_t *p = model;
    for (int l = 0; l < 6; ++l) {
        int32_t sw;
        memcpy(&sw, p, 4);
        p += 4;
        if (sw < 1) return -2;
        step_w[l] = sw;
        w[l] = (const int8_t *)p;
        p += orc_layer_wsize(l);
        for (int k = 0; k < DCOUT[l]; ++k) {
            int32_t bi;
            memcpy(&bi, p + 4 * k, 4);
            b[DBOFF[l] + k] = (float)bi;
        }
        p += 4 * DCOUT[l];
    }
    if (c1_step_y < 1) return -3; /* adjustBasic would divide by it */
    const long HW = (long)H * W;
    int8_t *x_ppro = (int8_t *)malloc(4 * HW), *v1 = (int8_t *)malloc(64 * HW), *conc1 = (int8_t *)malloc(48 * HW);
    int8_t *conc2 = (int8_t *)malloc(48 * HW), *v4 = (int8_t *)malloc(HW);
    float *ua = (float *)malloc(sizeof(float) * 64 * HW), *uc = (float *)malloc(sizeof(float) * 32 * HW);
    int stepw[4] = {0, 0, 0, 0}, stepy[4] = {0, 0, 0, 0};
    step_y[0] = c1_step_y;

    orc_ppro(y, x_ppro, HW);
    /* layer 1 (:90-105): no activation, mul_shift into C1.v */
    orc_adjust_basic(stepw, stepy, b, b_adj, 64, 0);
    conv_dyn(x_ppro, 0, H, W, w[0], b_adj, 0, ua);
    orc_mul_shift(ua, v1, 64 * HW, q[1], q[2]);
    orc_insert_w(stepw, step_w[0], 1);
    orc_insert_y(stepy, step_y[0], 1);
    /* layers 2 and 3 (:108-143): ReLU, concat_blu */
    const int8_t *in = v1;
    for (int layer = 2; layer <= 3; ++layer) {
        const int a = layer == 2 ? 1 : 3, c = a + 1;
        int8_t *conc = layer == 2 ? conc1 : conc2;
        orc_adjust_basic(stepw, stepy, b + DBOFF[a], b_adj + DBOFF[a], DCOUT[a], layer - 1);
        orc_adjust_basic(stepw, stepy, b + DBOFF[c], b_adj + DBOFF[c], DCOUT[c], layer - 1);
        conv_dyn(in, a, H, W, w[a], b_adj + DBOFF[a], 1, ua);
        conv_dyn(in, c, H, W, w[c], b_adj + DBOFF[c], 1, uc);
        step_y[a] = ((1 << q[3 * a + 2]) + q[3 * a + 1] / 2) / q[3 * a + 1];
        step_y[c] = ((1 << q[3 * c + 2]) + q[3 * c + 1] / 2) / q[3 * c + 1];
        orc_quant_blu(ua, conc, HW, DCOUT[a], q[3 * a], q[3 * a + 1], q[3 * a + 2]);
        orc_quant_blu(uc, conc + (size_t)DCOUT[a] * HW, HW, DCOUT[c], q[3 * c], q[3 * c + 1], q[3 * c + 2]);
        orc_insert_w(stepw, step_w[a], layer);
        orc_insert_y(stepy, step_y[a], layer);
        in = conc;
    }
    /* layer 4 and restore (:146-162) */
    orc_adjust_basic(stepw, stepy, b + 160, b_adj + 160, 1, 3);
    conv_dyn(conc2, 5, H, W, w[5], b_adj + 160, 0, ua);
    orc_insert_w(stepw, step_w[5], 4);
    orc_adjust_output_static(ua, v4, 141, 16, HW);
    orc_apply_res(y, v4, rec, HW);
    if (stepw_out) memcpy(stepw_out, stepw, sizeof stepw);
    if (stepy_out) memcpy(stepy_out, stepy, sizeof stepy);
    if (step_y_out) memcpy(step_y_out, step_y, sizeof step_y);
    free(x_ppro), free(v1), free(conc1), free(conc2), free(v4), free(ua), free(uc);
    return 0;
}
