// Host side of the C ABI (include/qcnn.h): device memory, model loading and packing, launches.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "qcnn_internal.h"

namespace qcnn {

static thread_local std::string g_err;

static int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

// (a failed call is reported once: the runtime's own record of it is cleared, or the next kernel launcher, which asks
// hipGetLastError() whether its launch went through, would report this failure as its own)
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            (void)hipGetLastError();                                                               \
            return fail(QCNN_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));                \
        }                                                                                          \
    } while (0)

// Layer table of qvrcnn::qvrcnn (inference/qvrcnn.cu:11-18).
static const int L_KS[6] = {5, 3, 5, 3, 1, 3};
static const int L_CIN[6] = {1, 64, 64, 48, 48, 48};
static const int L_COUT[6] = {64, 32, 16, 16, 32, 1};
static inline int pad4(int c) { return (c + 3) / 4 * 4; }
static inline int wsize(int l) { return L_KS[l] * L_KS[l] * pad4(L_CIN[l]) * L_COUT[l]; }  // inference/cnn.cu:24

static size_t model_bytes()
{
    size_t n = 0;
    for (int l = 0; l < 6; ++l) n += wsize(l) + 4 * L_COUT[l] + 12;
    return n;
}

// ---- scratch of qcnn_forward_blu_batch_noalloc ------------------------------------------------------
// The conc1 hand-off image of the two fused kernels for callers that may not allocate.  The array
// lives in the code object, so every GPU the library runs on holds its own copy at its own address
// until the process ends.  In host code the NAME of the array is only a handle for
// hipGetSymbolAddress: the address kernels and copies use is the per-device pointer looked up once
// per device in scratch_of_device().
static __device__ __align__(256) __attribute__((used)) unsigned char scratch[QCNN_NOALLOC_SCRATCH_BYTES];
constexpr int SCRATCH_MAX_DEVICES = 64;
static int8_t *g_scratch_ptr[SCRATCH_MAX_DEVICES];
static hipError_t g_scratch_err[SCRATCH_MAX_DEVICES];
static bool g_scratch_done[SCRATCH_MAX_DEVICES];
static std::mutex g_scratch_mu;

// Device address of `scratch` on GPU `gpu`, which must be the current device.  The look-up runs
// once per device (it may load the code object onto that device, which allocates; so it belongs to
// set-up: qcnn_create calls it).  Returns null with *err set when the look-up failed.
static int8_t *scratch_of_device(int gpu, hipError_t *err)
{
    if (gpu < 0 || gpu >= SCRATCH_MAX_DEVICES) {
        if (err) *err = hipErrorInvalidDevice;
        return nullptr;
    }
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_done[gpu]) {
        void *p = nullptr;
        g_scratch_err[gpu] = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        g_scratch_ptr[gpu] = g_scratch_err[gpu] == hipSuccess ? static_cast<int8_t *>(p) : nullptr;
        g_scratch_done[gpu] = true;
    }
    if (err) *err = g_scratch_err[gpu];
    return g_scratch_ptr[gpu];
}

}  // namespace qcnn

using namespace qcnn;

struct qcnn_engine {
    int gpu = 0, H = 0, W = 0, ncu = 256;
    size_t HW = 0;
    hipStream_t stream = nullptr;
    // I1
    uint8_t *x = nullptr, *x_rec = nullptr;
    const uint8_t *bound_y = nullptr;
    uint8_t *bound_rec = nullptr;
    // fused path
    int8_t *conc1_int = nullptr;  // [conc_frames] bordered images of conc_bytes(H, W)
    const int8_t *last_conc = nullptr;  // the image the last fused pass wrote frame 0 of its last launch into (QCNN_TAP_FUSED_CONC1)
    int8_t *scratch_ptr = nullptr;  // this GPU's copy of the static scratch (qcnn_forward_blu_batch_noalloc), or null
    hipError_t scratch_err = hipSuccess;
    int conc_frames = 0;          // capacity of conc1_int, in frames
    int group = 1;                // frames per kernel launch of forward_blu_batch
    bool tiled_b = false;         // QCNN_B_TILED=1: the tiled form of kernel B instead of the streaming one
    bool wide_b = true;           // the streaming form with two column tiles per wave (QCNN_B_WIDE=0: one tile per wave)
    int8_t *wpack = nullptr;
    int *bias_i = nullptr;
    float *bias_f = nullptr;
    unsigned long long *sse_dev = nullptr;
    unsigned long long *dbg = nullptr;  // diagnostic builds only
    // layered path (allocated on first use)
    bool layered_ready = false;
    int8_t *x_ppro = nullptr;
    float *u[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int8_t *v1 = nullptr, *conc1 = nullptr, *conc2 = nullptr;
    int8_t *w_file[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // CovLayer::w, file order
    // dynamic-quantisation pass (qcnn_load_para / qcnn_forward)
    bool dyn_loaded = false;
    DynState *dyn = nullptr;  // device
    int8_t *v4 = nullptr;     // C4.v: int8 residual [H][W]
    // float32 configuration (qcnn_load_float / qcnn_forward_float), allocated on first use
    bool float_ready = false, float_loaded = false;
    float *f_xp = nullptr, *f_u1 = nullptr, *f_conc1 = nullptr, *f_conc2 = nullptr, *f_u4 = nullptr;
    float *f_w[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    float *f_b[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    // model
    bool loaded = false;
    // arithmetic of the two fused kernels for the loaded model (PATH_INT / PATH_FLOAT / PATH_WIDE, see prove_paths)
    int path_a = 0, path_b = 0;
    int force_exact = 0;  // 1: float pipeline in both kernels, 2: clamped integers in kernel A (test / bench hooks)
    Requant rq[6];
    // clip pipeline (qcnn_forward_blu_clip): copy streams, two-deep device rings, hand-off events
    hipStream_t s_in = nullptr, s_out = nullptr, s_ori = nullptr;
    hipEvent_t ev_ori[2] = {nullptr, nullptr};
    uint8_t *ring_in[2] = {nullptr, nullptr}, *ring_out[2] = {nullptr, nullptr}, *ring_ori[2] = {nullptr, nullptr};
    int ring_frames = 0;  // capacity of each ring slot, in frames
    unsigned long long *clip_sse = nullptr;  // [2][clip_sse_frames]: per-frame sums before / after the net
    int clip_sse_frames = 0;
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_comp[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
    // timing
    bool timing = false;
    std::vector<hipEvent_t> ev;  // 3 per pass
    size_t ev_used = 0;
};

namespace qcnn {

static int ensure_layered(qcnn_engine *e)
{
    if (e->layered_ready) return QCNN_OK;
    HIP_TRY(hipSetDevice(e->gpu));
    // a buffer that exists is kept, so a call repeated after a failed allocation leaks nothing
    auto need = [&](auto **p, size_t bytes) -> hipError_t { return *p ? hipSuccess : hipMalloc(p, bytes); };
    HIP_TRY(need(&e->x_ppro, 4 * e->HW));
    for (int l = 0; l < 6; ++l) HIP_TRY(need(&e->u[l], sizeof(float) * L_COUT[l] * e->HW));
    HIP_TRY(need(&e->v1, 64 * e->HW));
    HIP_TRY(need(&e->conc1, 48 * e->HW));
    HIP_TRY(need(&e->conc2, 48 * e->HW));
    e->layered_ready = true;
    return QCNN_OK;
}

// CovLayer::build in the float configuration (inference/cnn.cu:29-35, 60-63): weights ks*ks*cin*cout floats,
// u float NCHW; x_ppro one float per sample (inference/cnn.cu:426-430).  The two branch outputs of a
// layer are allocated as one 48-channel image, which is the concatenation (ConcatLayer::concat,
// inference/cnn.cu:396-402, is then no copy).
static inline int fwsize(int l) { return L_KS[l] * L_KS[l] * L_CIN[l] * L_COUT[l]; }
static size_t float_model_bytes()
{
    size_t n = 0;
    for (int l = 0; l < 6; ++l) n += sizeof(float) * ((size_t)fwsize(l) + L_COUT[l]);
    return n;
}

static int ensure_float(qcnn_engine *e)
{
    if (e->float_ready) return QCNN_OK;
    HIP_TRY(hipSetDevice(e->gpu));
    // a buffer that exists is kept, so a call repeated after a failed allocation leaks nothing
    auto need = [&](float **p, size_t floats) -> hipError_t { return *p ? hipSuccess : hipMalloc(p, sizeof(float) * floats); };
    HIP_TRY(need(&e->f_xp, e->HW));
    HIP_TRY(need(&e->f_u1, 64 * e->HW));
    HIP_TRY(need(&e->f_conc1, 48 * e->HW));
    HIP_TRY(need(&e->f_conc2, 48 * e->HW));
    HIP_TRY(need(&e->f_u4, e->HW));
    for (int l = 0; l < 6; ++l) {
        HIP_TRY(need(&e->f_w[l], fwsize(l)));
        HIP_TRY(need(&e->f_b[l], L_COUT[l]));
    }
    e->float_ready = true;
    return QCNN_OK;
}

// CovLayer::u of layer l in the float configuration
static float *float_u(qcnn_engine *e, int l)
{
    switch (l) {
    case 0: return e->f_u1;
    case 1: return e->f_conc1;
    case 2: return e->f_conc1 + 32 * e->HW;
    case 3: return e->f_conc2;
    case 4: return e->f_conc2 + 16 * e->HW;
    default: return e->f_u4;
    }
}

// qvrcnn::load_para -> CovLayer::load_para of the float configuration (inference/cnn.cu:114-127): per layer
// wSize float32 weights [K][C][R][S], then outChannel float32 biases.
static int load_float_image(qcnn_engine *e, const uint8_t *img, size_t bytes)
{
    if (bytes < float_model_bytes())
        return fail(QCNN_E_FORMAT, "float model image is %zu bytes, expected %zu", bytes, float_model_bytes());
    int rc = ensure_float(e);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(e->gpu));
    HIP_TRY(hipStreamSynchronize(e->stream));
    const uint8_t *p = img;
    for (int l = 0; l < 6; ++l) {
        HIP_TRY(hipMemcpy(e->f_w[l], p, sizeof(float) * fwsize(l), hipMemcpyHostToDevice));
        p += sizeof(float) * fwsize(l);
        HIP_TRY(hipMemcpy(e->f_b[l], p, sizeof(float) * L_COUT[l], hipMemcpyHostToDevice));
        p += sizeof(float) * L_COUT[l];
    }
    e->float_loaded = true;
    return QCNN_OK;
}

// Canonical weight accessor on the file layout w[k][c/4][r][s][c%4] (inference/mat.cu:116).
struct LayerView {
    const int8_t *w;
    const int32_t *b;
    int ks, cin, cout;
    int at(int k, int c, int r, int s) const
    {
        return w[((((size_t)k * (pad4(cin) / 4) + (c >> 2)) * ks + r) * ks + s) * 4 + (c & 3)];
    }
};

// Builds the MFMA A-operand fragments (lane order, 16 bytes per lane) the fused kernels expect.
static void pack_fragments(const LayerView L[6], std::vector<int8_t> &out)
{
    out.assign(PackedOffsets::TOTAL, 0);
    // C1: 32x32x32, one fragment per 32-channel tile t.  Row rho of tile t is channel
    // 32t + chperm(rho), chosen so that lane (pixel, h) of the accumulator holds channels 16h..16h+15
    // in registers 0..15.  K = the 25 taps of the window, three kernel rows per k-group: byte j < 15 of
    // k-group h is kernel row 3h + j / 5, column j % 5; byte 15 and kernel rows >= 5 are zero.
    for (int t = 0; t < 2; ++t)
        for (int lane = 0; lane < 64; ++lane) {
            const int rho = lane & 31, h = lane >> 5;
            const int ch = 32 * t + 16 * ((rho >> 2) & 1) + 4 * (rho >> 3) + (rho & 3);
            int8_t *dst = &out[PackedOffsets::C1W + (t * 64 + lane) * 16];
            for (int j = 0; j < 16; ++j) {
                const int row = 3 * h + j / 5, col = j % 5;
                dst[j] = (j < 15 && row < 5) ? (int8_t)L[0].at(ch, 0, row, col) : 0;
            }
        }
    // C2_2 (5x5, 64 -> 16): fragment (s, r); row = channel; k-group g byte j = input channel 16g+j.
    for (int s = 0; s < 5; ++s)
        for (int r = 0; r < 5; ++r)
            for (int lane = 0; lane < 64; ++lane) {
                const int rho = lane & 15, g = lane >> 4;
                int8_t *dst = &out[PackedOffsets::C2W22 + ((s * 5 + r) * 64 + lane) * 16];
                for (int j = 0; j < 16; ++j) dst[j] = (int8_t)L[2].at(rho, 16 * g + j, r, s);
            }
    // C2_1 (3x3, 64 -> 32): fragment (s', r', t); row = channel 16t + rho.
    for (int s = 0; s < 3; ++s)
        for (int r = 0; r < 3; ++r)
            for (int t = 0; t < 2; ++t)
                for (int lane = 0; lane < 64; ++lane) {
                    const int rho = lane & 15, g = lane >> 4;
                    int8_t *dst = &out[PackedOffsets::C2W21 + ((((s * 3 + r) * 2) + t) * 64 + lane) * 16];
                    for (int j = 0; j < 16; ++j) dst[j] = (int8_t)L[1].at(16 * t + rho, 16 * g + j, r, s);
                }
    // C3_1 (3x3, 48 -> 16): the B operand is a conc1 pixel in the internal channel order; k-group
    // g byte j = internal position 16 * kb_chunk_of_kgroup(g) + j; the k-group of chunk 3 is zero.
    for (int s = 0; s < 3; ++s)
        for (int r = 0; r < 3; ++r)
            for (int lane = 0; lane < 64; ++lane) {
                const int rho = lane & 15, g = lane >> 4;
                int8_t *dst = &out[PackedOffsets::C3W31 + ((s * 3 + r) * 64 + lane) * 16];
                if (kb_chunk_of_kgroup(g) < 3)
                    for (int j = 0; j < 16; ++j)
                        dst[j] = (int8_t)L[3].at(rho, refch_of_pos(16 * kb_chunk_of_kgroup(g) + j), r, s);
            }
    // C3_2 (1x1, 48 -> 32): two 16-channel tiles.
    for (int t = 0; t < 2; ++t)
        for (int lane = 0; lane < 64; ++lane) {
            const int rho = lane & 15, g = lane >> 4;
            int8_t *dst = &out[PackedOffsets::C3W32 + (t * 64 + lane) * 16];
            if (kb_chunk_of_kgroup(g) < 3)
                for (int j = 0; j < 16; ++j)
                    dst[j] = (int8_t)L[4].at(16 * t + rho, refch_of_pos(16 * kb_chunk_of_kgroup(g) + j), 0, 0);
        }
    // C4 (3x3, 48 -> 1): fragment s; MFMA row rho < 3 is kernel row r = rho.
    for (int s = 0; s < 3; ++s)
        for (int lane = 0; lane < 64; ++lane) {
            const int rho = lane & 15, g = lane >> 4;
            int8_t *dst = &out[PackedOffsets::C4W + (s * 64 + lane) * 16];
            if (kb_chunk_of_kgroup(g) < 3 && rho < 3)
                for (int j = 0; j < 16; ++j) dst[j] = (int8_t)L[5].at(0, refch_of_pos(16 * kb_chunk_of_kgroup(g) + j), rho, s);
        }
    // C4 for the streaming kernel: one fragment, MFMA row 4s + r = tap (kernel row r, kernel column s); the B operand
    // of lane (pixel, g) is the three dwords a lane of the C3 epilogue owns: internal positions 12g .. 12g+11, the
    // fourth dword zero.
    for (int lane = 0; lane < 64; ++lane) {
        const int rho = lane & 15, g = lane >> 4;
        const int s = rho >> 2, r = rho & 3;
        int8_t *dst = &out[PackedOffsets::C4W9 + lane * 16];
        if (s < 3 && r < 3)
            for (int j = 0; j < 12; ++j) dst[j] = (int8_t)L[5].at(0, refch_of_pos(12 * g + j), r, s);
    }
}

// Range of the accumulator of output channel k of layer l over every possible input: the inputs of C1 are
// x - 128 in [-128, 127], those of the other layers are requantised activations in [0, 127], so the positive
// and the negative weights bound the sum separately (about half of what sum|w| * 127 would allow).
static void acc_range(const LayerView &L, int l, int k, long long *lo, long long *hi)
{
    long long pos = 0, neg = 0;  // sum of the positive weights, sum of |negative weights|
    for (int c = 0; c < L.cin; ++c)
        for (int r = 0; r < L.ks; ++r)
            for (int s = 0; s < L.ks; ++s) {
                const int w = L.at(k, c, r, s);
                if (w > 0) pos += w;
                else neg -= w;
            }
    const long long in_lo = l == 0 ? -128 : 0, in_hi = 127;
    *hi = pos * in_hi - neg * in_lo;
    *lo = pos * in_lo - neg * in_hi;
}

// Decides, per layer, whether integer epilogues reproduce the reference's float32 pipeline for THIS model.
// With every |acc| and |bias| at most 2^24 the conversions to float are exact, and with
// s = acc + bias + rounding offset of the requantiser (which the integer path adds to the bias) inside
// [-2^23, 2^23) the float sum is exact too and s is a signed 24-bit operand of the requantiser's
// multiply; with requant(blu) == 127 the clip-to-127 arm equals requantising min(u, blu), and the
// saturating 32-bit product (see requant4_int) keeps byte 3 at 127 beyond it: PATH_INT.
// If only the accumulator bounds fail, s can still be clamped to [-1, blu + rounding offset] first (one more
// instruction per value, fused_common.h requant12_wide): every s above that limit requantises to 127 like blu
// itself, every negative one to 0, and an accumulator beyond 2^24 (where the conversion to float rounds to an even
// integer) is, with |bias| <= 2^23, at least 2^23 away from both thresholds on the same side before and after
// rounding: PATH_WIDE.  Anything else gets the literal float pipeline: PATH_FLOAT.
enum { PATH_INT = 0, PATH_FLOAT = 1, PATH_WIDE = 2 };
static int prove_layer_path(const LayerView &L, int l, const Requant &q)
{
    const long long LIM24 = 1ll << 24, LIM23 = 1ll << 23;
    if (l < 5) {
        if (q.blu < 0 || q.mul <= 0 || q.shift > 24 || q.bias < 0) return PATH_FLOAT;
        if ((long long)q.blu + q.bias >= LIM23) return PATH_FLOAT;
        const long long top = ((long long)(q.blu + q.bias) * q.mul) >> q.shift;
        if (top != 127) return PATH_FLOAT;  // also bounds (blu + bias) * mulk below 128 << 24 = 2^31
        // pre-shifted multiplier: a positive signed 24-bit operand; bias * mulk <= 2^23
        if (((long long)q.mul << (24 - q.shift)) >= LIM23) return PATH_FLOAT;
        int path = PATH_INT;
        for (int k = 0; k < L.cout; ++k) {
            long long lo, hi;
            acc_range(L, l, k, &lo, &hi);
            const long long b = L.b[k];
            if (std::llabs(b) > LIM24) return PATH_FLOAT;
            if (-lo > LIM24 || hi > LIM24 || lo + b + q.bias < -LIM23 || hi + b + q.bias >= LIM23) {
                // the accumulator itself always fits an int (at most 1600 taps of 127 x 128)
                if (std::llabs(b) > LIM23 || -lo >= (1ll << 30) || hi >= (1ll << 30)) return PATH_FLOAT;
                path = PATH_WIDE;
            }
        }
        return path;
    }
    // C4 + applyRes_y in integers: acc + bias and the multiplier are 24-bit operands, the product plus
    // the rounding term stays below 2^31 (so the reference's unsigned arithmetic does not wrap), and
    // pixel + residual fits the reference's short
    long long lo, hi;
    acc_range(L, 5, 0, &lo, &hi);
    const long long b = L.b[0];
    if (-lo > LIM24 || hi > LIM24 || std::llabs(b) > LIM24) return PATH_FLOAT;
    const long long top = std::max(std::llabs(lo + b), std::llabs(hi + b));
    if (top >= LIM23 || q.mul <= 0 || q.mul >= (1 << 23)) return PATH_FLOAT;
    // the streaming kernel multiplies the accumulator without its bias (the bias rides in the rounding constant)
    if (-lo >= LIM23 || hi >= LIM23 || std::llabs(b) * q.mul + (1ll << (q.shift - 1)) >= (1ll << 31)) return PATH_FLOAT;
    const long long prod = top * q.mul + (1ll << (q.shift - 1));
    if (prod >= (1ll << 31)) return PATH_FLOAT;
    if ((prod >> q.shift) + 255 >= 32767) return PATH_FLOAT;
    return PATH_INT;
}

// Kernel A (C1, C2_1, C2_2) and kernel B (C3_1, C3_2, C4) choose their arithmetic separately.  Kernel A has the
// clamped form for its C2 stage only (C1's 25 taps cannot leave the 24-bit range); kernel B's accumulators
// (at most 432 taps of 127 x 127) cannot either, so it knows the integer and the float form only.
static void prove_paths(const LayerView L[6], const Requant rq[6], int *path_a, int *path_b)
{
    int p[6];
    for (int l = 0; l < 6; ++l) p[l] = prove_layer_path(L[l], l, rq[l]);
    if (p[0] != PATH_INT || p[1] == PATH_FLOAT || p[2] == PATH_FLOAT) *path_a = PATH_FLOAT;
    else *path_a = (p[1] == PATH_WIDE || p[2] == PATH_WIDE) ? PATH_WIDE : PATH_INT;
    *path_b = (p[3] == PATH_INT && p[4] == PATH_INT && p[5] == PATH_INT) ? PATH_INT : PATH_FLOAT;
}

static int load_image(qcnn_engine *e, const uint8_t *img, size_t bytes)
{
    if (bytes < model_bytes())
        return fail(QCNN_E_FORMAT, "model image is %zu bytes, expected %zu", bytes, model_bytes());
    HIP_TRY(hipSetDevice(e->gpu));
    LayerView L[6];
    std::vector<int32_t> bias_i(161);
    std::vector<float> bias_f(161);
    const uint8_t *p = img;
    for (int l = 0; l < 6; ++l) {
        L[l].w = reinterpret_cast<const int8_t *>(p);
        L[l].ks = L_KS[l];
        L[l].cin = L_CIN[l];
        L[l].cout = L_COUT[l];
        p += wsize(l);
        memcpy(&bias_i[BIAS_OFF[l]], p, 4 * L_COUT[l]);
        L[l].b = &bias_i[BIAS_OFF[l]];
        p += 4 * L_COUT[l];
        int32_t q[3];
        memcpy(q, p, 12);
        p += 12;
        if (q[2] < 1 || q[2] > 31) return fail(QCNN_E_FORMAT, "layer %d: shift %d out of range", l, q[2]);
        if (l < 5 && q[1] <= 0) return fail(QCNN_E_FORMAT, "layer %d: mul %d must be positive", l, q[1]);
        e->rq[l] = make_requant(q[0], q[1], q[2]);
        for (int k = 0; k < L_COUT[l]; ++k) bias_f[BIAS_OFF[l] + k] = (float)bias_i[BIAS_OFF[l] + k];  // cnn.cu:104
    }
    std::vector<int8_t> packed;
    pack_fragments(L, packed);
    prove_paths(L, e->rq, &e->path_a, &e->path_b);
    if (e->force_exact == 1) e->path_a = e->path_b = PATH_FLOAT;
    else if (e->force_exact == 2 && e->path_a == PATH_INT) e->path_a = PATH_WIDE;
    // make sure no kernel of a previous forward pass still reads the old parameters
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(e->wpack, packed.data(), packed.size(), hipMemcpyHostToDevice));
    {
        // the integer paths start their accumulators from bias + rounding offset of the requantiser
        std::vector<int32_t> start(bias_i);
        for (int l = 0; l < 5; ++l)
            if ((l < 3 ? e->path_a : e->path_b) != PATH_FLOAT)
                for (int k = 0; k < L_COUT[l]; ++k) start[BIAS_OFF[l] + k] += e->rq[l].bias;
        HIP_TRY(hipMemcpy(e->bias_i, start.data(), 161 * sizeof(int), hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMemcpy(e->bias_f, bias_f.data(), 161 * sizeof(float), hipMemcpyHostToDevice));
    for (int l = 0; l < 6; ++l) HIP_TRY(hipMemcpy(e->w_file[l], L[l].w, wsize(l), hipMemcpyHostToDevice));
    e->loaded = true;
    e->dyn_loaded = false;  // CovLayer::w now holds the static model
    return QCNN_OK;
}

static size_t dyn_model_bytes()
{
    size_t n = 0;
    for (int l = 0; l < 6; ++l) n += 4 + wsize(l) + 4 * L_COUT[l];
    return n;
}

// qvrcnn::load_para (inference/qvrcnn.cu:30-46) -> CovLayer::load_para (inference/cnn.cu:69-89) per layer:
// int32 step_w, wSize int8 weights (NCHW_VECT_C), outChannel int32 biases converted to float.
static int load_dyn_image(qcnn_engine *e, const uint8_t *img, size_t bytes)
{
    if (bytes < dyn_model_bytes())
        return fail(QCNN_E_FORMAT, "dynamic model image is %zu bytes, expected %zu", bytes, dyn_model_bytes());
    HIP_TRY(hipSetDevice(e->gpu));
    std::vector<uint8_t> hostv(sizeof(DynState), 0);
    DynState &st = *reinterpret_cast<DynState *>(hostv.data());
    const uint8_t *p = img;
    const uint8_t *wptr[6];
    for (int l = 0; l < 6; ++l) {
        int32_t sw;
        memcpy(&sw, p, 4);
        p += 4;
        // a step divides in ConcatLayer::concat (inference/cnn.cu:305-307) and in adjustOutput
        if (sw < 1) return fail(QCNN_E_FORMAT, "layer %d: step_w %d must be positive", l, sw);
        st.step_w[l] = sw;
        wptr[l] = p;
        p += wsize(l);
        for (int k = 0; k < L_COUT[l]; ++k) {
            int32_t bi;
            memcpy(&bi, p + 4 * k, 4);
            st.b[BIAS_OFF[l] + k] = (float)bi;  // inference/cnn.cu:81
        }
        p += 4 * L_COUT[l];
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(e->dyn, &st, sizeof(DynState), hipMemcpyHostToDevice));
    for (int l = 0; l < 6; ++l) HIP_TRY(hipMemcpy(e->w_file[l], wptr[l], wsize(l), hipMemcpyHostToDevice));
    e->dyn_loaded = true;
    e->loaded = false;  // CovLayer::w now holds the dynamic model
    return QCNN_OK;
}

// conc1 scratch for `frames` frames per launch (grown on demand, never shrunk).  The larger image is allocated
// before the old one is released, so a failed allocation leaves the engine as it was (still good for launches of
// conc_frames frames).
static int ensure_conc(qcnn_engine *e, int frames)
{
    if (frames <= e->conc_frames && e->conc1_int) return QCNN_OK;
    HIP_TRY(hipSetDevice(e->gpu));
    const size_t bytes = (size_t)frames * conc_bytes(e->H, e->W);
    int8_t *grown = nullptr;
    HIP_TRY(hipMalloc(&grown, bytes));
    // the border stays zero: kernel A writes the frame only (cleared on the engine stream, which does not wait
    // for the null stream)
    hipError_t he = hipMemsetAsync(grown, 0, bytes, e->stream);
    if (he == hipSuccess && e->stream) he = hipStreamSynchronize(e->stream);  // also: nothing reads the old image any more
    if (he != hipSuccess) {
        (void)hipFree(grown);
        return fail(QCNN_E_HIP, "clearing the conc1 image failed: %s", hipGetErrorString(he));
    }
    if (e->conc1_int) (void)hipFree(e->conc1_int);
    if (e->last_conc == e->conc1_int) e->last_conc = nullptr;  // what the tap would convert is gone
    e->conc1_int = grown;
    e->conc_frames = frames;
    return QCNN_OK;
}

static FusedArgs fused_args(qcnn_engine *e, const uint8_t *y, uint8_t *rec, int nframes, size_t stride, int8_t *conc1)
{
    FusedArgs a;
    a.y = y;
    a.rec = rec;
    a.nframes = nframes;
    a.frame_stride = stride;
    a.conc_stride = conc_bytes(e->H, e->W);
    a.conc1 = conc1;
    a.wpack = e->wpack;
    a.bias_i = e->bias_i;
    a.bias_f = e->bias_f;
    a.H = e->H;
    a.W = e->W;
    a.ncu = e->ncu;
    a.dbg = e->dbg;
    {
        const char *sb = getenv("QCNN_STAMP_BLOCK");
        a.dbg_block = sb ? atoi(sb) : 0;
    }
    for (int l = 0; l < 6; ++l) a.rq[l] = e->rq[l];
    return a;
}

}  // namespace qcnn

// ============================================================================================
// C ABI
// ============================================================================================
extern "C" {

const char *qcnn_last_error(void) { return g_err.c_str(); }
const char *qcnn_version(void) { return "qcnn-gfx950 0.1"; }

int qcnn_device_count(int *count)
{
    if (!count) return fail(QCNN_E_ARG, "count is null");
    HIP_TRY(hipGetDeviceCount(count));
    return QCNN_OK;
}

int qcnn_device_pci_bus_id(int gpu, char *buf, int len)
{
    if (!buf || len < 2) return fail(QCNN_E_ARG, "bad buffer");
    HIP_TRY(hipDeviceGetPCIBusId(buf, len, gpu));
    return QCNN_OK;
}

int qcnn_create(int gpu, int batch, int channel, int height, int width, qcnn_t **out)
{
    if (!out) return fail(QCNN_E_ARG, "out is null");
    *out = nullptr;
    if (batch != 1 || channel != 1)
        return fail(QCNN_E_UNSUPPORTED, "batch=%d channel=%d: only single Y frames (1,1) are supported", batch, channel);
    // the conc1 image of one frame is addressed with 32-bit offsets: at most 2 GiB (about 44 million pixels)
    if (height < 1 || width < 1 || (long long)height * width > (1ll << 30) || conc_bytes(height, width) >= (1ull << 31))
        return fail(QCNN_E_ARG, "bad frame size %dx%d", width, height);
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (gpu < 0 || gpu >= ndev) return fail(QCNN_E_ARG, "gpu %d out of range (%d devices)", gpu, ndev);
    HIP_TRY(hipSetDevice(gpu));  // qvrcnn.cu:6
    int ncu = 0;
    HIP_TRY(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, gpu));
    qcnn_engine *e = new qcnn_engine;
    e->ncu = ncu > 0 ? ncu : 256;
    e->gpu = gpu;
    e->H = height;
    e->W = width;
    e->HW = (size_t)height * width;
    auto bail = [&](int rc) {
        qcnn_destroy(e);
        return rc;
    };
#define HIP_TRY_E(expr)                                                                                  \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) return bail(fail(QCNN_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_))); \
    } while (0)
    HIP_TRY_E(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    // this GPU's address of the static scratch; a failure is reported by the entry point that needs it
    e->scratch_ptr = scratch_of_device(gpu, &e->scratch_err);
    HIP_TRY_E(hipMalloc(&e->x, e->HW));
    HIP_TRY_E(hipMalloc(&e->x_rec, e->HW));
    HIP_TRY_E(hipMalloc(&e->conc1_int, conc_bytes(height, width)));
    HIP_TRY_E(hipMemsetAsync(e->conc1_int, 0, conc_bytes(height, width), e->stream));
    e->conc_frames = 1;
    {
        // frames per launch of the batch entry point: up to 16, as many as fit a scratch budget of 8 GiB (of the
        // device's 288: sixteen 3840 x 2160 frames take 6.4 GB of conc1 and run 1.8 % faster than five).  Longer
        // launches were measured on 1080p frames (QCNN_BATCH_GROUP=32 / 64, kernel B still in launches of 16, see
        // launch_frames): kernel A gains 0.9 % at 64 frames, kernel B loses 6 % when its launches no longer follow
        // the kernel A launch of their own frames, the whole pass loses 0.6 %.
        const char *gb = getenv("QCNN_BATCH_GROUP");
        const long long fit = (8ll << 30) / (long long)conc_bytes(height, width);
        e->group = gb ? atoi(gb) : (int)(fit < 1 ? 1 : fit > 16 ? 16 : fit);
        if (e->group < 1) e->group = 1;
    }
    HIP_TRY_E(hipMalloc(&e->wpack, PackedOffsets::TOTAL));
    HIP_TRY_E(hipMalloc(&e->bias_i, 161 * sizeof(int)));
    HIP_TRY_E(hipMalloc(&e->bias_f, 161 * sizeof(float)));
    HIP_TRY_E(hipMalloc(&e->sse_dev, sizeof(unsigned long long)));
    HIP_TRY_E(hipMalloc(&e->dyn, sizeof(DynState)));
    HIP_TRY_E(hipMemset(e->dyn, 0, sizeof(DynState)));
    HIP_TRY_E(hipMalloc(&e->v4, e->HW));
#ifdef QCNN_DEBUG_CONC2
    HIP_TRY_E(hipMalloc(&e->dbg, 48 * e->HW));
    HIP_TRY_E(hipMemset(e->dbg, 0, 48 * e->HW));
#endif
#ifdef QCNN_STAMPS
    HIP_TRY_E(hipMalloc(&e->dbg, (24 * 128 + 3 * 16384) * sizeof(unsigned long long)));
    HIP_TRY_E(hipMemset(e->dbg, 0, (24 * 128 + 3 * 16384) * sizeof(unsigned long long)));
#endif
    for (int l = 0; l < 6; ++l) HIP_TRY_E(hipMalloc(&e->w_file[l], wsize(l)));
#undef HIP_TRY_E
    {
        const char *tb = getenv("QCNN_B_TILED");
        e->tiled_b = tb && atoi(tb) != 0;
        const char *wb = getenv("QCNN_B_WIDE");
        e->wide_b = !wb || atoi(wb) != 0;
    }
    const char *fx = getenv("QCNN_FORCE_EXACT");
    e->force_exact = fx ? atoi(fx) : 0;
    *out = e;
    return QCNN_OK;
}

int qcnn_destroy(qcnn_t *e)
{
    if (!e) return QCNN_OK;
    (void)hipSetDevice(e->gpu);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    for (hipEvent_t ev : e->ev) (void)hipEventDestroy(ev);
    (void)hipFree(e->x);
    (void)hipFree(e->x_rec);
    (void)hipFree(e->conc1_int);
    (void)hipFree(e->wpack);
    (void)hipFree(e->bias_i);
    (void)hipFree(e->bias_f);
    (void)hipFree(e->sse_dev);
    (void)hipFree(e->clip_sse);
    for (int b = 0; b < 2; ++b) {
        (void)hipFree(e->ring_in[b]);
        (void)hipFree(e->ring_out[b]);
        (void)hipFree(e->ring_ori[b]);
        if (e->ev_in[b]) (void)hipEventDestroy(e->ev_in[b]);
        if (e->ev_comp[b]) (void)hipEventDestroy(e->ev_comp[b]);
        if (e->ev_out[b]) (void)hipEventDestroy(e->ev_out[b]);
    }
    if (e->s_in) (void)hipStreamDestroy(e->s_in);
    if (e->s_ori) (void)hipStreamDestroy(e->s_ori);
    for (int b = 0; b < 2; ++b)
        if (e->ev_ori[b]) (void)hipEventDestroy(e->ev_ori[b]);
    if (e->s_out) (void)hipStreamDestroy(e->s_out);
    (void)hipFree(e->dbg);
    (void)hipFree(e->x_ppro);
    for (int l = 0; l < 6; ++l) {
        (void)hipFree(e->u[l]);
        (void)hipFree(e->w_file[l]);
    }
    (void)hipFree(e->v1);
    (void)hipFree(e->dyn);
    (void)hipFree(e->v4);
    (void)hipFree(e->conc1);
    (void)hipFree(e->conc2);
    (void)hipFree(e->f_xp);
    (void)hipFree(e->f_u1);
    (void)hipFree(e->f_conc1);
    (void)hipFree(e->f_conc2);
    (void)hipFree(e->f_u4);
    for (int l = 0; l < 6; ++l) {
        (void)hipFree(e->f_w[l]);
        (void)hipFree(e->f_b[l]);
    }
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
    return QCNN_OK;
}

int qcnn_load_static_mem(qcnn_t *e, const void *image, size_t bytes)
{
    if (!e || !image) return fail(QCNN_E_ARG, "null argument");
    return load_image(e, static_cast<const uint8_t *>(image), bytes);
}

int qcnn_load_static(qcnn_t *e, const char *path)
{
    if (!e || !path) return fail(QCNN_E_ARG, "null argument");
    FILE *fp = fopen(path, "rb");
    if (!fp) return fail(QCNN_E_IO, "cannot open model file %s", path);
    std::vector<uint8_t> buf(model_bytes());
    const size_t got = fread(buf.data(), 1, buf.size(), fp);
    fclose(fp);
    if (got != buf.size()) return fail(QCNN_E_FORMAT, "%s: short model file (%zu of %zu bytes)", path, got, buf.size());
    return load_image(e, buf.data(), buf.size());
}

int qcnn_set_exact(qcnn_t *e, int on)
{
    // test hook (takes effect at the next load): 1 forces the float-pipeline kernels, 2 the clamped integer form
    // of kernel A for a model that would run the plain integer form
    if (!e) return fail(QCNN_E_ARG, "null engine");
    if (on < 0 || on > 2) return fail(QCNN_E_ARG, "qcnn_set_exact: mode %d", on);
    e->force_exact = on;
    return QCNN_OK;
}

int qcnn_is_exact(qcnn_t *e)
{
    if (!e) return fail(QCNN_E_ARG, "null engine");
    return e->path_a | (e->path_b << 4);
}

int qcnn_upload(qcnn_t *e, const uint8_t *host_y)
{
    if (!e || !host_y) return fail(QCNN_E_ARG, "null argument");
    HIP_TRY(hipSetDevice(e->gpu));
    HIP_TRY(hipMemcpyAsync(e->x, host_y, e->HW, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));  // InputLayer::load is a blocking cudaMemcpy (cnn.cu:441)
    return QCNN_OK;
}

int qcnn_upload_async(qcnn_t *e, const uint8_t *host_y)
{
    if (!e || !host_y) return fail(QCNN_E_ARG, "null argument");
    HIP_TRY(hipSetDevice(e->gpu));
    HIP_TRY(hipMemcpyAsync(e->x, host_y, e->HW, hipMemcpyHostToDevice, e->stream));
    return QCNN_OK;
}

// one launch of each fused kernel over `nframes` frames, with `conc1` (room for nframes frames) as the
// hand-off image between them; timed = false never touches the timing events (creating one allocates)
static int launch_frames(qcnn_engine *e, const uint8_t *y, uint8_t *rec, int nframes, size_t stride, int8_t *conc1 = nullptr,
                         bool timed = true)
{
    HIP_TRY(hipSetDevice(e->gpu));
    if (!conc1) conc1 = e->conc1_int;
    if (!conc1 || (conc1 == e->conc1_int && nframes > e->conc_frames))
        return fail(QCNN_E_STATE, "no conc1 image for a launch of %d frame(s) (an earlier allocation failed)", nframes);
    const FusedArgs a = fused_args(e, y, rec, nframes, stride, conc1);
    e->last_conc = conc1;
    hipEvent_t *ev = nullptr;
    if (e->timing && timed) {
        if (e->ev_used + 3 > e->ev.size()) {
            const size_t old = e->ev.size();
            if (old < 3 * 65536) {
                e->ev.resize(old + 3 * 256);
                for (size_t i = old; i < e->ev.size(); ++i) HIP_TRY(hipEventCreate(&e->ev[i]));
            }
        }
        if (e->ev_used + 3 <= e->ev.size()) {
            ev = &e->ev[e->ev_used];
            e->ev_used += 3;
        }
    }
    if (ev) HIP_TRY(hipEventRecord(ev[0], e->stream));
    HIP_TRY(launch_fused_c1c2(a, e->path_a, e->stream));
    if (ev) HIP_TRY(hipEventRecord(ev[1], e->stream));
    // Kernel B in launches of at most 16 frames (only a group above the default of 16 is split): sixteen 1080p frames
    // are one round of its waves, which is the shape it balances (two waves per SIMD that finish together); launches
    // of several rounds measured 3.5 % (32 frames) and 10 % (64 frames) slower per frame.
    constexpr int B_FRAMES = 16;
    for (int i = 0; i < nframes; i += B_FRAMES) {
        FusedArgs b = a;
        b.nframes = nframes - i < B_FRAMES ? nframes - i : B_FRAMES;
        b.y = a.y + (size_t)i * a.frame_stride;
        b.rec = a.rec + (size_t)i * a.frame_stride;
        b.conc1 = a.conc1 + (size_t)i * a.conc_stride;
        if (e->tiled_b) HIP_TRY(launch_fused_c3c4(b, e->path_b == PATH_FLOAT, e->stream));
        else if (e->wide_b) HIP_TRY(launch_fused_c3c4_stream2(b, e->path_b == PATH_FLOAT, e->stream));
        else HIP_TRY(launch_fused_c3c4_stream(b, e->path_b == PATH_FLOAT, e->stream));
    }
    if (ev) HIP_TRY(hipEventRecord(ev[2], e->stream));
    return QCNN_OK;
}

int qcnn_forward_blu(qcnn_t *e)
{
    if (!e) return fail(QCNN_E_ARG, "null engine");
    if (!e->loaded) return fail(QCNN_E_STATE, "forward_blu before load_static_para");
    return launch_frames(e, e->bound_y ? e->bound_y : e->x, e->bound_rec ? e->bound_rec : e->x_rec, 1, e->HW);
}

int qcnn_forward_blu_batch(qcnn_t *e, const void *dev_y, void *dev_rec, int nframes, size_t stride)
{
    if (!e || !dev_y || !dev_rec) return fail(QCNN_E_ARG, "null argument");
    if (!e->loaded) return fail(QCNN_E_STATE, "forward_blu before load_static_para");
    if (nframes < 0 || stride < e->HW) return fail(QCNN_E_ARG, "bad batch: nframes=%d stride=%zu", nframes, stride);
    const int group = nframes < e->group ? nframes : e->group;
    if (group > 0) {
        const int rc = ensure_conc(e, group);
        if (rc) return rc;
    }
    // the persistent kernels walk the tiles of up to `group` frames per launch, so their pipelines
    // fill and drain once per group instead of once per frame
    for (int i = 0; i < nframes; i += group) {
        const int nf = nframes - i < group ? nframes - i : group;
        const int rc = launch_frames(e, static_cast<const uint8_t *>(dev_y) + (size_t)i * stride,
                                     static_cast<uint8_t *>(dev_rec) + (size_t)i * stride, nf, stride);
        if (rc) return rc;
    }
    return QCNN_OK;
}

int qcnn_forward_blu_batch_noalloc(qcnn_t *e, const void *dev_y, void *dev_rec, int nframes, size_t stride)
{
    if (!e || !dev_y || !dev_rec) return fail(QCNN_E_ARG, "null argument");
    if (!e->loaded) return fail(QCNN_E_STATE, "forward_blu before load_static_para");
    if (nframes < 0 || stride < e->HW) return fail(QCNN_E_ARG, "bad batch: nframes=%d stride=%zu", nframes, stride);
    const size_t per_frame = conc_bytes(e->H, e->W);  // 48 bytes per pixel of the frame and of its zero border
    if (per_frame > (size_t)QCNN_NOALLOC_SCRATCH_BYTES)
        return fail(QCNN_E_UNSUPPORTED, "a %dx%d frame needs %zu bytes of conc1 scratch, the static scratch holds %zu", e->W, e->H,
                    per_frame, (size_t)QCNN_NOALLOC_SCRATCH_BYTES);
    if (!e->scratch_ptr)
        return fail(QCNN_E_HIP, "hipGetSymbolAddress(scratch) failed on GPU %d at qcnn_create: %s", e->gpu,
                    hipGetErrorString(e->scratch_err));
    // frames per launch: what the static scratch holds, at most the engine's group
    const size_t fit = (size_t)QCNN_NOALLOC_SCRATCH_BYTES / per_frame;
    const int group = (int)std::min<size_t>(fit, (size_t)std::max(e->group, 1));
    // The array keeps what the last call on this GPU left in it, possibly an image of another geometry: the
    // border of every image this call uses must read as zero, so it is cleared first (a kernel on the engine
    // stream, ahead of kernel A; everything inside the border is written by kernel A before kernel B reads it).
    HIP_TRY(hipSetDevice(e->gpu));
    HIP_TRY(launch_conc_zero_border(e->scratch_ptr, std::min(group, std::max(nframes, 1)), e->H, e->W, e->stream));
    for (int i = 0; i < nframes; i += group) {
        const int nf = nframes - i < group ? nframes - i : group;
        const int rc = launch_frames(e, static_cast<const uint8_t *>(dev_y) + (size_t)i * stride,
                                     static_cast<uint8_t *>(dev_rec) + (size_t)i * stride, nf, stride, e->scratch_ptr, false);
        if (rc) return rc;
    }
    return QCNN_OK;
}

int qcnn_batch_group(qcnn_t *e, int set, int *current)
{
    if (!e) return fail(QCNN_E_ARG, "null engine");
    if (set > 0) e->group = set;
    if (current) *current = e->group;
    return QCNN_OK;
}

// Streams, events and ring slots of the clip pipeline, created on first use / grown on demand.
// sse_frames > 0: also the per-frame SSE buffer for a clip of that many frames.
static int ensure_clip(qcnn_engine *e, int frames, bool with_ori, int sse_frames = 0)
{
    HIP_TRY(hipSetDevice(e->gpu));
    if (!e->s_in) HIP_TRY(hipStreamCreateWithFlags(&e->s_in, hipStreamNonBlocking));
    if (!e->s_ori) HIP_TRY(hipStreamCreateWithFlags(&e->s_ori, hipStreamNonBlocking));
    if (!e->s_out) HIP_TRY(hipStreamCreateWithFlags(&e->s_out, hipStreamNonBlocking));
    for (int b = 0; b < 2; ++b) {
        if (!e->ev_in[b]) HIP_TRY(hipEventCreateWithFlags(&e->ev_in[b], hipEventDisableTiming));
        if (!e->ev_comp[b]) HIP_TRY(hipEventCreateWithFlags(&e->ev_comp[b], hipEventDisableTiming));
        if (!e->ev_out[b]) HIP_TRY(hipEventCreateWithFlags(&e->ev_out[b], hipEventDisableTiming));
        if (!e->ev_ori[b]) HIP_TRY(hipEventCreateWithFlags(&e->ev_ori[b], hipEventDisableTiming));
    }
    if (frames > e->ring_frames) {
        HIP_TRY(hipDeviceSynchronize());
        for (int b = 0; b < 2; ++b) {
            (void)hipFree(e->ring_in[b]);
            (void)hipFree(e->ring_out[b]);
            (void)hipFree(e->ring_ori[b]);
            e->ring_in[b] = e->ring_out[b] = e->ring_ori[b] = nullptr;
        }
        e->ring_frames = 0;
        for (int b = 0; b < 2; ++b) {
            HIP_TRY(hipMalloc(&e->ring_in[b], (size_t)frames * e->HW));
            HIP_TRY(hipMalloc(&e->ring_out[b], (size_t)frames * e->HW));
        }
        e->ring_frames = frames;
    }
    if (with_ori)
        for (int b = 0; b < 2; ++b)
            if (!e->ring_ori[b]) HIP_TRY(hipMalloc(&e->ring_ori[b], (size_t)e->ring_frames * e->HW));
    if (sse_frames > e->clip_sse_frames) {
        HIP_TRY(hipDeviceSynchronize());
        (void)hipFree(e->clip_sse);
        e->clip_sse = nullptr;
        e->clip_sse_frames = 0;
        HIP_TRY(hipMalloc(&e->clip_sse, 2 * (size_t)sse_frames * sizeof(unsigned long long)));
        e->clip_sse_frames = sse_frames;
    }
    return QCNN_OK;
}

int qcnn_clip_reserve(qcnn_t *e, int nframes, int with_sse)
{
    if (!e) return fail(QCNN_E_ARG, "null engine");
    if (nframes < 1) return QCNN_OK;
    const int cg = e->group < 16 ? e->group : 16;  // the clip pipeline's chunk: see qcnn_forward_blu_clip
    const int chunk = nframes < cg ? nframes : cg;
    int rc = ensure_clip(e, chunk, with_sse != 0, with_sse ? nframes : 0);
    return rc ? rc : ensure_conc(e, chunk);
}

int qcnn_forward_blu_clip(qcnn_t *e, const uint8_t *host_y, uint8_t *host_rec, int nframes, size_t stride,
                          const uint8_t *host_ori, uint64_t *sse_before, uint64_t *sse_after)
{
    if (!e || !host_y || !host_rec) return fail(QCNN_E_ARG, "null argument");
    if (!e->loaded) return fail(QCNN_E_STATE, "forward_blu before load_static_para");
    if (nframes < 0 || stride < e->HW) return fail(QCNN_E_ARG, "bad clip: nframes=%d stride=%zu", nframes, stride);
    if ((sse_before || sse_after) && !host_ori) return fail(QCNN_E_ARG, "SSE output without the original frames");
    if (nframes == 0) return QCNN_OK;
    const size_t HW = e->HW;
    // chunks of at most 16 frames whatever the batch entry point's group: the pipeline's fill, drain and ring
    // buffers grow with the chunk
    const int cg = e->group < 16 ? e->group : 16;
    const int chunk = nframes < cg ? nframes : cg;
    const bool sse = host_ori && (sse_before || sse_after);
    int rc = ensure_clip(e, chunk, sse, sse ? nframes : 0);
    if (rc) return rc;
    rc = ensure_conc(e, chunk);
    if (rc) return rc;
    // The caller's buffers are used as they are: page-locked memory (qcnn_host_alloc_pinned,
    // qcnn_host_register) is copied by DMA and overlaps the kernels, pageable memory goes through the
    // runtime's staging path.  Nothing is registered here: spans of sharded callers overlap, and a
    // registration made by one call could be withdrawn under another call's copies.
    unsigned long long *d_sse = e->clip_sse;  // [2][nframes]: before, after
#define HIP_TRY_C(expr)                                                                        \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            (void)hipDeviceSynchronize();                                                      \
            return fail(QCNN_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));            \
        }                                                                                      \
    } while (0)
    if (sse) HIP_TRY_C(hipMemsetAsync(d_sse, 0, 2 * (size_t)nframes * sizeof(unsigned long long), e->stream));
    // Chunk sizes: a long clip starts and ends with chunks of an eighth, a quarter and half a launch group, so that the
    // kernels wait for the copy-in of two frames instead of sixteen at the start and the caller for the copy-out of
    // two at the end.
    std::vector<int> sizes;
    {
        int head[3] = {chunk / 8, chunk / 4, chunk / 2}, left = nframes;
        const bool ramp = chunk >= 8 && nframes >= 4 * chunk && !getenv("QCNN_CLIP_NO_RAMP");
        if (ramp)
            for (int i = 0; i < 3; ++i) sizes.push_back(head[i]), left -= 2 * head[i];
        for (; left > 0; left -= chunk) sizes.push_back(left < chunk ? left : chunk);
        if (ramp)
            for (int i = 2; i >= 0; --i) sizes.push_back(head[i]);
    }
    for (int k = 0, i0 = 0; k < (int)sizes.size(); i0 += sizes[k], ++k) {
        const int b = k & 1;
        const int nf = sizes[k];
        // copy-in stream: slot b is free once the kernels of chunk k-2 have read it
        HIP_TRY_C(hipStreamWaitEvent(e->s_in, e->ev_comp[b], 0));
        HIP_TRY_C(hipMemcpy2DAsync(e->ring_in[b], HW, host_y + (size_t)i0 * stride, stride, HW, nf, hipMemcpyHostToDevice,
                                   e->s_in));
        HIP_TRY_C(hipEventRecord(e->ev_in[b], e->s_in));
        if (sse) {
            // the originals travel on a stream of their own, so that two copy engines feed the device
            HIP_TRY_C(hipStreamWaitEvent(e->s_ori, e->ev_comp[b], 0));
            HIP_TRY_C(hipMemcpy2DAsync(e->ring_ori[b], HW, host_ori + (size_t)i0 * stride, stride, HW, nf,
                                       hipMemcpyHostToDevice, e->s_ori));
            HIP_TRY_C(hipEventRecord(e->ev_ori[b], e->s_ori));
        }
        // compute stream: needs the frames, and the output slot drained by the copy-out of chunk k-2
        HIP_TRY_C(hipStreamWaitEvent(e->stream, e->ev_in[b], 0));
        HIP_TRY_C(hipStreamWaitEvent(e->stream, e->ev_out[b], 0));
        rc = launch_frames(e, e->ring_in[b], e->ring_out[b], nf, HW);
        if (rc) {
            (void)hipDeviceSynchronize();
            return rc;
        }
        if (sse) {
            HIP_TRY_C(hipStreamWaitEvent(e->stream, e->ev_ori[b], 0));
            HIP_TRY_C(launch_sse_clip(e->ring_in[b], e->ring_out[b], e->ring_ori[b], HW, nf, sse_before ? d_sse + i0 : nullptr,
                                      sse_after ? d_sse + nframes + i0 : nullptr, e->stream));
        }
        HIP_TRY_C(hipEventRecord(e->ev_comp[b], e->stream));
        // copy-out stream
        HIP_TRY_C(hipStreamWaitEvent(e->s_out, e->ev_comp[b], 0));
        // (one linear copy for densely packed frames was measured: no faster on 256 frames, slower on 500)
        HIP_TRY_C(hipMemcpy2DAsync(host_rec + (size_t)i0 * stride, stride, e->ring_out[b], HW, HW, nf, hipMemcpyDeviceToHost,
                                   e->s_out));
        HIP_TRY_C(hipEventRecord(e->ev_out[b], e->s_out));
    }
    HIP_TRY_C(hipStreamSynchronize(e->s_out));
    HIP_TRY_C(hipStreamSynchronize(e->stream));
    HIP_TRY_C(hipStreamSynchronize(e->s_in));
    HIP_TRY_C(hipStreamSynchronize(e->s_ori));
    if (sse) {
        std::vector<unsigned long long> h(2 * (size_t)nframes);
        HIP_TRY_C(hipMemcpy(h.data(), d_sse, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        for (int i = 0; i < nframes; ++i) {
            if (sse_before) sse_before[i] = h[i];
            if (sse_after) sse_after[i] = h[nframes + i];
        }
    }
#undef HIP_TRY_C
    return QCNN_OK;
}

int qcnn_forward_blu_layered(qcnn_t *e)
{
    if (!e) return fail(QCNN_E_ARG, "null engine");
    if (!e->loaded) return fail(QCNN_E_STATE, "forward_blu before load_static_para");
    int rc = ensure_layered(e);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(e->gpu));
    hipStream_t s = e->stream;
    const int H = e->H, W = e->W;
    const size_t HW = e->HW;
    const uint8_t *y = e->bound_y ? e->bound_y : e->x;
    uint8_t *rec = e->bound_rec ? e->bound_rec : e->x_rec;
    // the call sequence of qvrcnn::forward_blu, inference/qvrcnn.cu:168-242
    HIP_TRY(launch_ppro(y, e->x_ppro, HW, s));
    HIP_TRY(launch_conv_static(e->x_ppro, e->w_file[0], e->bias_f + BIAS_OFF[0], e->u[0], H, W, 1, 64, 5, 0, s));
    HIP_TRY(launch_quant_blu(e->u[0], e->v1, HW, 64, e->rq[0], s));
    HIP_TRY(launch_conv_static(e->v1, e->w_file[1], e->bias_f + BIAS_OFF[1], e->u[1], H, W, 64, 32, 3, 0, s));
    HIP_TRY(launch_conv_static(e->v1, e->w_file[2], e->bias_f + BIAS_OFF[2], e->u[2], H, W, 64, 16, 5, 0, s));
    HIP_TRY(launch_quant_blu(e->u[1], e->conc1, HW, 32, e->rq[1], s));
    HIP_TRY(launch_quant_blu(e->u[2], e->conc1 + 32 * HW, HW, 16, e->rq[2], s));
    HIP_TRY(launch_conv_static(e->conc1, e->w_file[3], e->bias_f + BIAS_OFF[3], e->u[3], H, W, 48, 16, 3, 0, s));
    HIP_TRY(launch_conv_static(e->conc1, e->w_file[4], e->bias_f + BIAS_OFF[4], e->u[4], H, W, 48, 32, 1, 0, s));
    HIP_TRY(launch_quant_blu(e->u[3], e->conc2, HW, 16, e->rq[3], s));
    HIP_TRY(launch_quant_blu(e->u[4], e->conc2 + 16 * HW, HW, 32, e->rq[4], s));
    HIP_TRY(launch_conv_static(e->conc2, e->w_file[5], e->bias_f + BIAS_OFF[5], e->u[5], H, W, 48, 1, 3, 0, s));
    HIP_TRY(launch_apply_res_y(y, rec, e->u[5], e->rq[5].mul, e->rq[5].shift, HW, s));
    return QCNN_OK;
}

int qcnn_load_para_mem(qcnn_t *e, const void *image, size_t bytes)
{
    if (!e || !image) return fail(QCNN_E_ARG, "null argument");
    return load_dyn_image(e, static_cast<const uint8_t *>(image), bytes);
}

int qcnn_load_para(qcnn_t *e, const char *path)
{
    if (!e || !path) return fail(QCNN_E_ARG, "null argument");
    FILE *fp = fopen(path, "rb");
    if (!fp) return fail(QCNN_E_IO, "cannot open model file %s", path);
    std::vector<uint8_t> buf(dyn_model_bytes());
    const size_t got = fread(buf.data(), 1, buf.size(), fp);
    fclose(fp);
    if (got != buf.size()) return fail(QCNN_E_FORMAT, "%s: short model file (%zu of %zu bytes)", path, got, buf.size());
    return load_dyn_image(e, buf.data(), buf.size());
}

// qvrcnn::forward (inference/qvrcnn.cu:82-167) with quantize_out / concat / adjustOutput in the places
// where the shipped text calls their static counterparts (see include/qcnn.h).  The host bookkeeping
// between the layers runs on the device (k_dyn_book), so nothing here waits for the GPU.
//
// taps = true : every u is written as float32 NCHW, as the reference does; the largest |u| is collected
//               in the convolution's epilogue and a separate kernel divides and relayouts (tap mode).
// taps = false: no float image exists.  Every layer's convolution runs twice on the matrix cores: once
//               keeping only the largest |u|, and, after the bookkeeping kernel has turned it into a step,
//               once more with the divide-quantiser in the epilogue, writing int8 directly.  The pass then
//               moves 1 + 64 + 48 + 48 + 4 + 1 + 1 bytes per pixel instead of the 644 B/px of float images
//               written and read twice.
static int forward_dynamic(qcnn_engine *e, bool taps)
{
    if (!e) return fail(QCNN_E_ARG, "null engine");
    if (!e->dyn_loaded) return fail(QCNN_E_STATE, "forward before load_para");
    int rc = ensure_layered(e);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(e->gpu));
    hipStream_t s = e->stream;
    const int H = e->H, W = e->W;
    const size_t HW = e->HW;
    const uint8_t *y = e->bound_y ? e->bound_y : e->x;
    uint8_t *rec = e->bound_rec ? e->bound_rec : e->x_rec;
    DynState *st = e->dyn;
    const float *badj = st->b_adj;  // device addresses; never dereferenced on the host
    const bool mfma = !conv_direct_only();
    if (!taps && !mfma) taps = true;  // the recomputing pass exists on the matrix-core kernel only
    // convolution + bias (+ ReLU) of layer l; q != null: quantised output only, else float u (taps) or
    // nothing but the maximum
    auto conv = [&](const void *x, int l, int relu, bool want_u, void *q) -> hipError_t {
        const float *b = badj + BIAS_OFF[l];
        unsigned *mx = (l < 5 && !q) ? &st->maxbits[l] : nullptr;
        if (mfma && conv_mfma_supported(L_CIN[l], L_COUT[l], L_KS[l]))
            return launch_conv_mfma(x, e->w_file[l], b, want_u ? e->u[l] : nullptr, q, mx, q ? &st->step_y[l] : nullptr,
                                    1, relu, H, W, L_CIN[l], L_COUT[l], L_KS[l], s);
        // the direct convolution writes the float image only: the divide-quantiser follows as a pass of its own
        hipError_t he = launch_conv_static(x, e->w_file[l], b, e->u[l], H, W, L_CIN[l], L_COUT[l], L_KS[l], relu, s);
        if (he == hipSuccess && mx) he = launch_absmax(e->u[l], (size_t)L_COUT[l] * HW, mx, s);
        if (he == hipSuccess && q) he = launch_quant_div(e->u[l], q, HW, L_COUT[l], 0, &st->step_y[l], s);
        return he;
    };
    // one layer of the trunk: branches a (and c, or -1) read `in`, their quantised outputs are concatenated in `out`
    // both branches of a layer in one launch (the input tile is staged once); q_a / q_c as in conv()
    auto conv_pair = [&](const void *x, int la, int lc, int relu, bool want_u, void *q_a, void *q_c) -> hipError_t {
        auto branch = [&](int l, void *q) {
            ConvPairBranch b;
            b.w = e->w_file[l];
            b.b = badj + BIAS_OFF[l];
            b.u = want_u ? e->u[l] : nullptr;
            b.q = q;
            b.maxbits = q ? nullptr : &st->maxbits[l];
            b.divisor_dev = q ? &st->step_y[l] : nullptr;
            b.divisor = 1;
            b.cout = L_COUT[l];
            b.ks = L_KS[l];
            return b;
        };
        return launch_conv_mfma_pair(x, L_CIN[la], H, W, relu, branch(la, q_a), branch(lc, q_c), s);
    };
    const bool pairs = mfma && getenv("QCNN_NO_PAIR") == nullptr;
    auto layer = [&](const void *in, int a, int c, int relu, int stage, int8_t *out) -> int {
        const bool pair = pairs && c >= 0 && L_CIN[a] == L_CIN[c] &&
                          conv_pair_supported(L_CIN[a], L_COUT[a], L_KS[a], L_COUT[c], L_KS[c]);
        if (pair) {
            HIP_TRY(conv_pair(in, a, c, relu, taps, nullptr, nullptr));
        } else {
            HIP_TRY(conv(in, a, relu, taps, nullptr));
            if (c >= 0) HIP_TRY(conv(in, c, relu, taps, nullptr));
        }
        HIP_TRY(launch_dyn_book(st, stage, s));
        int8_t *out_c = out + (size_t)L_COUT[a] * HW;  // legal because outChannel of branch a is a multiple of 4
        if (taps) {
            HIP_TRY(launch_quant_div(e->u[a], out, HW, L_COUT[a], 0, &st->step_y[a], s));
            if (c >= 0) HIP_TRY(launch_quant_div(e->u[c], out_c, HW, L_COUT[c], 0, &st->step_y[c], s));
        } else if (pair) {
            HIP_TRY(conv_pair(in, a, c, relu, false, out, out_c));
        } else {
            HIP_TRY(conv(in, a, relu, false, out));
            if (c >= 0) HIP_TRY(conv(in, c, relu, false, out_c));
        }
        return QCNN_OK;
    };
    HIP_TRY(launch_dyn_book(st, 0, s));
    HIP_TRY(launch_ppro(y, e->x_ppro, HW, s));
    rc = layer(e->x_ppro, 0, -1, 0, 1, e->v1);  // layer 1: no activation (inference/qvrcnn.cu:96 is commented out)
    if (rc) return rc;
    rc = layer(e->v1, 1, 2, 1, 2, e->conc1);
    if (rc) return rc;
    rc = layer(e->conc1, 3, 4, 1, 3, e->conc2);
    if (rc) return rc;
    // layer 4 and restore
    HIP_TRY(conv(e->conc2, 5, 0, true, nullptr));
    HIP_TRY(launch_dyn_book(st, 4, s));
    HIP_TRY(launch_adjust_output(DynSteps(), &st->steps, e->u[5], e->v4, 4, HW, s));
    HIP_TRY(launch_apply_res(y, e->v4, rec, HW, s));
    return QCNN_OK;
}

int qcnn_forward(qcnn_t *e) { return forward_dynamic(e, false); }
int qcnn_forward_layered(qcnn_t *e) { return forward_dynamic(e, true); }

int qcnn_load_float_mem(qcnn_t *e, const void *image, size_t bytes)
{
    if (!e || !image) return fail(QCNN_E_ARG, "null argument");
    return load_float_image(e, static_cast<const uint8_t *>(image), bytes);
}

int qcnn_load_float(qcnn_t *e, const char *path)
{
    if (!e || !path) return fail(QCNN_E_ARG, "null argument");
    FILE *fp = fopen(path, "rb");
    if (!fp) return fail(QCNN_E_IO, "cannot open model file %s", path);
    std::vector<uint8_t> buf(float_model_bytes());
    const size_t got = fread(buf.data(), 1, buf.size(), fp);
    fclose(fp);
    if (got != buf.size()) return fail(QCNN_E_FORMAT, "%s: short model file (%zu of %zu bytes)", path, got, buf.size());
    return load_float_image(e, buf.data(), buf.size());
}

// qvrcnn::forward of the float configuration, inference/qvrcnn.cu:244-285: ppro, C1 + ReLU, (C2_1 || C2_2) + ReLU,
// concat, (C3_1 || C3_2) + ReLU, concat, C4, applyRes.  The ReLU runs in each convolution's epilogue and the
// branches write straight into the concatenated image.
int qcnn_forward_float(qcnn_t *e)
{
    if (!e) return fail(QCNN_E_ARG, "null engine");
    if (!e->float_loaded) return fail(QCNN_E_STATE, "forward before the float model is loaded");
    HIP_TRY(hipSetDevice(e->gpu));
    hipStream_t s = e->stream;
    const int H = e->H, W = e->W;
    const size_t HW = e->HW;
    const uint8_t *y = e->bound_y ? e->bound_y : e->x;
    uint8_t *rec = e->bound_rec ? e->bound_rec : e->x_rec;
    auto conv = [&](const float *in, int l, int relu) {
        return launch_conv_float(in, e->f_w[l], e->f_b[l], float_u(e, l), H, W, L_CIN[l], L_COUT[l], L_KS[l], relu, s);
    };
    HIP_TRY(launch_ppro_float(y, e->f_xp, HW, s));
    HIP_TRY(conv(e->f_xp, 0, 1));
    HIP_TRY(conv(e->f_u1, 1, 1));
    HIP_TRY(conv(e->f_u1, 2, 1));
    HIP_TRY(conv(e->f_conc1, 3, 1));
    HIP_TRY(conv(e->f_conc1, 4, 1));
    HIP_TRY(conv(e->f_conc2, 5, 0));
    HIP_TRY(launch_apply_res_float(y, e->f_u4, rec, HW, s));
    return QCNN_OK;
}

void *qcnn_float_w_device_ptr(qcnn_t *e, int layer)
{
    return (e && layer >= 0 && layer < 6 && ensure_float(e) == QCNN_OK) ? e->f_w[layer] : nullptr;
}
void *qcnn_float_b_device_ptr(qcnn_t *e, int layer)
{
    return (e && layer >= 0 && layer < 6 && ensure_float(e) == QCNN_OK) ? e->f_b[layer] : nullptr;
}
void *qcnn_float_u_device_ptr(qcnn_t *e, int layer)
{
    return (e && layer >= 0 && layer < 6 && ensure_float(e) == QCNN_OK) ? float_u(e, layer) : nullptr;
}

int qcnn_dynamic_state(qcnn_t *e, qcnn_dynamic_state_t *out)
{
    if (!e || !out) return fail(QCNN_E_ARG, "null argument");
    if (!e->dyn_loaded) return fail(QCNN_E_STATE, "no dynamic model loaded");
    HIP_TRY(hipSetDevice(e->gpu));
    std::vector<uint8_t> hostv(sizeof(DynState));
    HIP_TRY(hipMemcpyAsync(hostv.data(), e->dyn, sizeof(DynState), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    const DynState &st = *reinterpret_cast<const DynState *>(hostv.data());
    for (int k = 0; k < 4; ++k) out->stepw[k] = st.steps.stepw[k], out->stepy[k] = st.steps.stepy[k];
    for (int l = 0; l < 6; ++l) out->step_w[l] = st.step_w[l], out->step_y[l] = st.step_y[l], out->max_u[l] = st.max_u[l];
    memcpy(out->b_adj, st.b_adj, sizeof st.b_adj);
    return QCNN_OK;
}

int qcnn_sync(qcnn_t *e)
{
    if (!e) return fail(QCNN_E_ARG, "null engine");
    HIP_TRY(hipStreamSynchronize(e->stream));
    return QCNN_OK;
}

void *qcnn_recon_device_ptr(qcnn_t *e) { return e ? e->x_rec : nullptr; }
void *qcnn_input_device_ptr(qcnn_t *e) { return e ? e->x : nullptr; }
void *qcnn_stream(qcnn_t *e) { return e ? (void *)e->stream : nullptr; }

int qcnn_bind_io(qcnn_t *e, const void *dev_y, void *dev_rec)
{
    if (!e) return fail(QCNN_E_ARG, "null engine");
    e->bound_y = static_cast<const uint8_t *>(dev_y);
    e->bound_rec = static_cast<uint8_t *>(dev_rec);
    return QCNN_OK;
}

int qcnn_download_async(qcnn_t *e, uint8_t *host_rec)
{
    if (!e || !host_rec) return fail(QCNN_E_ARG, "null argument");
    HIP_TRY(hipSetDevice(e->gpu));
    HIP_TRY(hipMemcpyAsync(host_rec, e->bound_rec ? e->bound_rec : e->x_rec, e->HW, hipMemcpyDeviceToHost, e->stream));
    return QCNN_OK;
}

int qcnn_download(qcnn_t *e, uint8_t *host_rec)
{
    int rc = qcnn_download_async(e, host_rec);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    return QCNN_OK;
}

int qcnn_sse_device(qcnn_t *e, const void *dev_a, const void *dev_b, size_t n, uint64_t *sse)
{
    if (!e || !dev_a || !dev_b || !sse) return fail(QCNN_E_ARG, "null argument");
    HIP_TRY(hipSetDevice(e->gpu));
    HIP_TRY(launch_sse(dev_a, dev_b, n, e->sse_dev, e->stream));
    unsigned long long h = 0;
    HIP_TRY(hipMemcpyAsync(&h, e->sse_dev, sizeof h, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    *sse = h;
    return QCNN_OK;
}

static int tap_info(qcnn_t *e, int tap, void **dev, size_t *bytes)
{
    const size_t HW = e->HW;
    switch (tap) {
    case QCNN_TAP_X_PPRO: *dev = e->x_ppro; *bytes = 4 * HW; return 0;
    case QCNN_TAP_U_C1: case QCNN_TAP_U_C2_1: case QCNN_TAP_U_C2_2: case QCNN_TAP_U_C3_1: case QCNN_TAP_U_C3_2:
    case QCNN_TAP_U_C4: {
        const int l = tap - QCNN_TAP_U_C1;
        *dev = e->u[l];
        *bytes = sizeof(float) * L_COUT[l] * HW;
        return 0;
    }
    case QCNN_TAP_V_C1: *dev = e->v1; *bytes = 64 * HW; return 0;
    case QCNN_TAP_CONC1: *dev = e->conc1; *bytes = 48 * HW; return 0;
    case QCNN_TAP_CONC2: *dev = e->conc2; *bytes = 48 * HW; return 0;
    case QCNN_TAP_X_REC: *dev = e->bound_rec ? e->bound_rec : e->x_rec; *bytes = HW; return 0;
    case QCNN_TAP_FUSED_CONC1: *dev = nullptr; *bytes = 48 * HW; return 0;  // a converted copy: no device address of its own
    case QCNN_TAP_V_C4: *dev = e->v4; *bytes = HW; return 0;
    case QCNN_TAP_F_X_PPRO: *dev = e->f_xp; *bytes = sizeof(float) * HW; return 0;
    case QCNN_TAP_F_U_C1: *dev = e->f_u1; *bytes = sizeof(float) * 64 * HW; return 0;
    case QCNN_TAP_F_CONC1: *dev = e->f_conc1; *bytes = sizeof(float) * 48 * HW; return 0;
    case QCNN_TAP_F_CONC2: *dev = e->f_conc2; *bytes = sizeof(float) * 48 * HW; return 0;
    case QCNN_TAP_F_U_C4: *dev = e->f_u4; *bytes = sizeof(float) * HW; return 0;
    default: return fail(QCNN_E_ARG, "unknown tap %d", tap);
    }
}

int qcnn_tap_bytes(qcnn_t *e, int tap, size_t *bytes)
{
    if (!e || !bytes) return fail(QCNN_E_ARG, "null argument");
    void *dev;
    return tap_info(e, tap, &dev, bytes);
}

void *qcnn_tap_device_ptr(qcnn_t *e, int tap)
{
    if (!e) return nullptr;
    if (tap >= QCNN_TAP_F_X_PPRO) {
        if (ensure_float(e)) return nullptr;
    } else if (tap != QCNN_TAP_X_REC && tap != QCNN_TAP_FUSED_CONC1 && tap != QCNN_TAP_V_C4 && ensure_layered(e))
        return nullptr;
    void *dev = nullptr;
    size_t bytes;
    if (tap_info(e, tap, &dev, &bytes)) return nullptr;
    return dev;
}

int qcnn_tap(qcnn_t *e, int tap, void *host_out, size_t bytes)
{
    if (!e || !host_out) return fail(QCNN_E_ARG, "null argument");
    HIP_TRY(hipSetDevice(e->gpu));
    if (tap != QCNN_TAP_X_REC && tap != QCNN_TAP_FUSED_CONC1) {
        int rc = tap >= QCNN_TAP_F_X_PPRO ? ensure_float(e) : ensure_layered(e);
        if (rc) return rc;
    }
    void *dev = nullptr;
    size_t need = 0;
    int rc = tap_info(e, tap, &dev, &need);
    if (rc) return rc;
    if (bytes < need) return fail(QCNN_E_ARG, "tap %d needs %zu bytes, buffer has %zu", tap, need, bytes);
    if (tap == QCNN_TAP_FUSED_CONC1) {
        // convert the fused path's [H][W][48] image to the layout of Conc1.conc in a scratch buffer
        // (of whichever image the last fused pass wrote: the engine's own, or the static scratch of the entry
        // point that allocates nothing)
        if (!e->last_conc) return fail(QCNN_E_STATE, "no fused forward pass has run on this engine");
        int8_t *tmp = nullptr;
        HIP_TRY(hipMalloc(&tmp, need));
        hipError_t he = launch_conc_to_ref(e->last_conc, tmp, e->H, e->W, e->stream);
        if (he == hipSuccess) he = hipMemcpyAsync(host_out, tmp, need, hipMemcpyDeviceToHost, e->stream);
        if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
        (void)hipFree(tmp);
        if (he != hipSuccess) return fail(QCNN_E_HIP, "fused conc1 tap failed: %s", hipGetErrorString(he));
        return QCNN_OK;
    }
    HIP_TRY(hipMemcpyAsync(host_out, dev, need, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return QCNN_OK;
}

int qcnn_layer_params(qcnn_t *e, int layer, int *blu, int *mul, int *shift)
{
    if (!e || layer < 0 || layer >= 6) return fail(QCNN_E_ARG, "bad layer %d", layer);
    if (!e->loaded) return fail(QCNN_E_STATE, "no model loaded");
    if (blu) *blu = e->rq[layer].blu;
    if (mul) *mul = e->rq[layer].mul;
    if (shift) *shift = e->rq[layer].shift;
    return QCNN_OK;
}

void *qcnn_layer_w_device_ptr(qcnn_t *e, int layer) { return (e && layer >= 0 && layer < 6) ? e->w_file[layer] : nullptr; }
void *qcnn_layer_b_device_ptr(qcnn_t *e, int layer)
{
    if (!e || layer < 0 || layer >= 6) return nullptr;
    return e->dyn_loaded ? (void *)(e->dyn->b + BIAS_OFF[layer]) : (void *)(e->bias_f + BIAS_OFF[layer]);
}
void *qcnn_layer_b_adj_device_ptr(qcnn_t *e, int layer)
{
    return (e && layer >= 0 && layer < 6) ? (void *)(e->dyn->b_adj + BIAS_OFF[layer]) : nullptr;
}

int qcnn_set_timing(qcnn_t *e, int on)
{
    if (!e) return fail(QCNN_E_ARG, "null engine");
    e->timing = on != 0;
    return QCNN_OK;
}

int qcnn_kernel_times_us(qcnn_t *e, double *k_c1c2_us, double *k_c3c4_us, int *passes, int reset)
{
    if (!e) return fail(QCNN_E_ARG, "null engine");
    HIP_TRY(hipStreamSynchronize(e->stream));
    double ta = 0, tb = 0;
    const size_t n = e->ev_used / 3;
    for (size_t i = 0; i < n; ++i) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, e->ev[3 * i], e->ev[3 * i + 1]));
        ta += ms;
        HIP_TRY(hipEventElapsedTime(&ms, e->ev[3 * i + 1], e->ev[3 * i + 2]));
        tb += ms;
    }
    if (k_c1c2_us) *k_c1c2_us = n ? ta * 1000.0 / n : 0;
    if (k_c3c4_us) *k_c3c4_us = n ? tb * 1000.0 / n : 0;
    if (passes) *passes = (int)n;
    if (reset) e->ev_used = 0;
    return QCNN_OK;
}

#ifdef QCNN_DEBUG_CONC2
// diagnostic builds only: conc2 as the streaming kernel computed it, [H][W][48], internal channel order
int qcnn_debug_conc2(qcnn_t *e, void *host_out)
{
    if (!e || !host_out) return fail(QCNN_E_ARG, "null argument");
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(host_out, e->dbg, 48 * e->HW, hipMemcpyDeviceToHost));
    return QCNN_OK;
}
#endif
#ifdef QCNN_STAMPS
// diagnostic builds only: copies the cycle stamps (24 rows x 128 slots) of the last kernels to host
int qcnn_debug_stamps(qcnn_t *e, unsigned long long *host_out)
{
    if (!e || !host_out) return fail(QCNN_E_ARG, "null argument");
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(host_out, e->dbg, 24 * 128 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(e->dbg, 0, 24 * 128 * sizeof(unsigned long long)));
    return QCNN_OK;
}
// diagnostic builds only: start tick, end tick and hardware id of the first 16384 workgroups of the last kernel B launch
int qcnn_debug_blocks(qcnn_t *e, unsigned long long *host_out)
{
    if (!e || !host_out) return fail(QCNN_E_ARG, "null argument");
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(host_out, e->dbg + 24 * 128, 3 * 16384 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return QCNN_OK;
}
#endif

// ---- layer level ---------------------------------------------------------------------------
int qcnn_op_ppro(const void *x, void *x_ppro, size_t n, void *stream)
{
    if (!x || !x_ppro) return fail(QCNN_E_ARG, "null argument");
    HIP_TRY(launch_ppro(x, x_ppro, n, (hipStream_t)stream));
    return QCNN_OK;
}

int qcnn_op_conv_static(const void *x, const void *w, const void *b, void *u, int height, int width, int cin,
                        int cout, int ks, void *stream)
{
    if (!x || !w || !b || !u) return fail(QCNN_E_ARG, "null argument");
    if (height < 1 || width < 1 || cin < 1 || cout < 1) return fail(QCNN_E_ARG, "bad conv shape");
    if (ks != 1 && ks != 3 && ks != 5) return fail(QCNN_E_UNSUPPORTED, "ksize %d (1, 3 or 5)", ks);
    HIP_TRY(launch_conv_static(x, w, b, u, height, width, cin, cout, ks, 0, (hipStream_t)stream));
    return QCNN_OK;
}

int qcnn_op_quant_blu(const void *u, void *out, size_t hw, int channels, int blu, int mul, int shift, void *stream)
{
    if (!u || !out) return fail(QCNN_E_ARG, "null argument");
    if (mul <= 0 || shift < 1 || shift > 31 || channels < 1) return fail(QCNN_E_ARG, "bad requant parameters");
    HIP_TRY(launch_quant_blu(u, out, hw, channels, make_requant(blu, mul, shift), (hipStream_t)stream));
    return QCNN_OK;
}

int qcnn_op_apply_res_y(const void *x, void *rec, const void *res, int mul, int shift, size_t n, void *stream)
{
    if (!x || !rec || !res) return fail(QCNN_E_ARG, "null argument");
    if (shift < 1 || shift > 31) return fail(QCNN_E_ARG, "bad shift %d", shift);
    HIP_TRY(launch_apply_res_y(x, rec, res, mul, shift, n, (hipStream_t)stream));
    return QCNN_OK;
}

int qcnn_op_activate(void *u, size_t n, void *stream)
{
    if (!u) return fail(QCNN_E_ARG, "null argument");
    HIP_TRY(launch_relu(u, n, (hipStream_t)stream));
    return QCNN_OK;
}

int qcnn_op_find_max(const void *u, void *workspace, size_t n, float *host_max, void *stream)
{
    if (!u || !workspace || !host_max) return fail(QCNN_E_ARG, "null argument");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(workspace, 0, sizeof(unsigned), s));
    HIP_TRY(launch_absmax(u, n, (unsigned *)workspace, s));
    HIP_TRY(hipMemcpyAsync(host_max, workspace, sizeof(float), hipMemcpyDeviceToHost, s));  // mat.cu:34
    HIP_TRY(hipStreamSynchronize(s));
    return QCNN_OK;
}

int qcnn_op_quant_div(const void *u, void *out, size_t hw, int channels, int divisor, void *stream)
{
    if (!u || !out) return fail(QCNN_E_ARG, "null argument");
    if (divisor == 0 || channels < 1) return fail(QCNN_E_ARG, "bad divisor %d / channels %d", divisor, channels);
    HIP_TRY(launch_quant_div(u, out, hw, channels, divisor, nullptr, (hipStream_t)stream));
    return QCNN_OK;
}

int qcnn_op_mul_shift(const void *u, void *out, size_t n, int mul, int shift, void *stream)
{
    if (!u || !out) return fail(QCNN_E_ARG, "null argument");
    if (mul <= 0 || shift < 1 || shift > 31) return fail(QCNN_E_ARG, "bad requant parameters");
    HIP_TRY(launch_mul_shift(u, out, n, mul, shift, (hipStream_t)stream));
    return QCNN_OK;
}

static int copy_steps(const int32_t *stepw, const int32_t *stepy, int n_w, int n_y, DynSteps *st)
{
    if (!stepw || !stepy) return fail(QCNN_E_ARG, "null argument");
    for (int k = 0; k < 4; ++k) st->stepw[k] = stepw[k], st->stepy[k] = stepy[k];
    // the steps of a pass are positive (CovLayer::load_para rejects step_w < 1, a step_y is derived from a maximum);
    // the device-side divisions rely on it
    for (int k = 0; k < n_w; ++k)
        if (stepw[k] < 1) return fail(QCNN_E_ARG, "stepw[%d] = %d must be positive", k, stepw[k]);
    for (int k = 0; k < n_y; ++k)
        if (stepy[k] < 1) return fail(QCNN_E_ARG, "stepy[%d] = %d must be positive", k, stepy[k]);
    return QCNN_OK;
}

int qcnn_op_adjust_basic(const int32_t *stepw, const int32_t *stepy, const void *b, void *b_adj, int count, int n,
                         void *stream)
{
    if (!b || !b_adj) return fail(QCNN_E_ARG, "null argument");
    if (n < 0 || n > 4 || count < 0) return fail(QCNN_E_ARG, "bad n %d / count %d", n, count);
    DynSteps st;
    int rc = copy_steps(stepw, stepy, 0, n, &st);  // only stepy divides here
    if (rc) return rc;
    HIP_TRY(launch_adjust_basic(st, b, b_adj, count, n, (hipStream_t)stream));
    return QCNN_OK;
}

int qcnn_op_adjust_output(const int32_t *stepw, const int32_t *stepy, const void *o, void *o_adj, int n, size_t num,
                          void *stream)
{
    if (!o || !o_adj) return fail(QCNN_E_ARG, "null argument");
    if (n < 1 || n > 4) return fail(QCNN_E_ARG, "bad n %d", n);
    DynSteps st;
    int rc = copy_steps(stepw, stepy, n, 0, &st);  // only stepw divides here
    if (rc) return rc;
    HIP_TRY(launch_adjust_output(st, nullptr, o, o_adj, n, num, (hipStream_t)stream));
    return QCNN_OK;
}

int qcnn_op_adjust_output_static(const void *o, void *o_adj, int mul, int shift, size_t num, void *stream)
{
    if (!o || !o_adj) return fail(QCNN_E_ARG, "null argument");
    if (shift < 1 || shift > 31) return fail(QCNN_E_ARG, "bad shift %d", shift);
    HIP_TRY(launch_adjust_output_static(o, o_adj, mul, shift, num, (hipStream_t)stream));
    return QCNN_OK;
}

int qcnn_op_apply_res(const void *x, const void *res, void *rec, size_t n, void *stream)
{
    if (!x || !res || !rec) return fail(QCNN_E_ARG, "null argument");
    HIP_TRY(launch_apply_res(x, res, rec, n, (hipStream_t)stream));
    return QCNN_OK;
}

int qcnn_op_ppro_float(const void *x, void *x_ppro, size_t n, void *stream)
{
    if (!x || !x_ppro) return fail(QCNN_E_ARG, "null pointer");
    HIP_TRY(launch_ppro_float(x, x_ppro, n, (hipStream_t)stream));
    return QCNN_OK;
}

int qcnn_op_conv_float(const void *x, const void *w, const void *b, void *u, int height, int width, int cin, int cout,
                       int ks, int relu, void *stream)
{
    if (!x || !w || !b || !u) return fail(QCNN_E_ARG, "null pointer");
    if (height < 1 || width < 1) return fail(QCNN_E_ARG, "bad frame size %dx%d", width, height);
    if (!conv_float_supported(cin, cout, ks))
        return fail(QCNN_E_UNSUPPORTED, "float convolution %dx%d, %d -> %d channels is not a layer shape of the network",
                    ks, ks, cin, cout);
    HIP_TRY(launch_conv_float(x, w, b, u, height, width, cin, cout, ks, relu, (hipStream_t)stream));
    return QCNN_OK;
}

int qcnn_op_apply_res_float(const void *x, const void *res, void *rec, size_t n, void *stream)
{
    if (!x || !res || !rec) return fail(QCNN_E_ARG, "null pointer");
    HIP_TRY(launch_apply_res_float(x, res, rec, n, (hipStream_t)stream));
    return QCNN_OK;
}

int qcnn_concat_steps(float max1, float max2, int step_w1, int step_w2, int *stepy1, int *stepy2)
{
    if (!stepy1 || !stepy2) return fail(QCNN_E_ARG, "null argument");
    if (step_w1 < 1 || step_w2 < 1) return fail(QCNN_E_ARG, "weight steps must be positive");
    concat_steps(max1, max2, step_w1, step_w2, stepy1, stepy2);
    return QCNN_OK;
}

// ---- device memory helpers ---------------------------------------------------------------------
int qcnn_set_device(int gpu)
{
    HIP_TRY(hipSetDevice(gpu));
    return QCNN_OK;
}
int qcnn_dev_alloc(void **dev, size_t bytes)
{
    if (!dev) return fail(QCNN_E_ARG, "null argument");
    HIP_TRY(hipMalloc(dev, bytes ? bytes : 1));
    return QCNN_OK;
}
int qcnn_dev_free(void *dev)
{
    HIP_TRY(hipFree(dev));
    return QCNN_OK;
}
int qcnn_dev_h2d(void *dev, const void *host, size_t bytes)
{
    HIP_TRY(hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice));
    return QCNN_OK;
}
int qcnn_dev_d2h(void *host, const void *dev, size_t bytes)
{
    HIP_TRY(hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost));
    return QCNN_OK;
}
int qcnn_dev_memset(void *dev, int value, size_t bytes)
{
    HIP_TRY(hipMemset(dev, value, bytes));
    return QCNN_OK;
}
int qcnn_dev_sync(void)
{
    HIP_TRY(hipDeviceSynchronize());
    return QCNN_OK;
}
int qcnn_dev_mem_info(size_t *free_bytes, size_t *total_bytes)
{
    size_t f = 0, t = 0;
    HIP_TRY(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return QCNN_OK;
}
int qcnn_host_alloc_pinned(void **host, size_t bytes)
{
    if (!host) return fail(QCNN_E_ARG, "null argument");
    // portable: page-locked for every device of the process, not only the current one (the workers of a
    // multi-GPU run all copy out of one clip)
    HIP_TRY(hipHostMalloc(host, bytes ? bytes : 1, hipHostMallocPortable));
    return QCNN_OK;
}
int qcnn_host_free_pinned(void *host)
{
    HIP_TRY(hipHostFree(host));
    return QCNN_OK;
}
int qcnn_host_register(void *host, size_t bytes)
{
    if (!host) return fail(QCNN_E_ARG, "null argument");
    // portable (every device of the process may copy from it by DMA) where the runtime takes the flag
    if (hipHostRegister(host, bytes, hipHostRegisterPortable) == hipSuccess) return QCNN_OK;
    (void)hipGetLastError();
    HIP_TRY(hipHostRegister(host, bytes, hipHostRegisterDefault));
    return QCNN_OK;
}
int qcnn_host_unregister(void *host)
{
    if (!host) return fail(QCNN_E_ARG, "null argument");
    HIP_TRY(hipHostUnregister(host));
    return QCNN_OK;
}

// ---- offline model conversion --------------------------------------------------------------------
long qcnn_hwcn2nchw_vect_c(const int8_t *hwcn, int h, int w, int c, int n, int8_t *out)
{
    if (!hwcn || !out || h < 1 || w < 1 || c < 1 || n < 1) return fail(QCNN_E_ARG, "bad argument");
    const int c4 = (c + 3) / 4;
    const long total = (long)n * c4 * h * w * 4;
    memset(out, 0, (size_t)total);
    for (int in = 0; in < n; ++in)
        for (int ic = 0; ic < c; ++ic)
            for (int ih = 0; ih < h; ++ih)
                for (int iw = 0; iw < w; ++iw)
                    out[((((long)in * c4 + (ic >> 2)) * h + ih) * w + iw) * 4 + (ic & 3)] =
                        hwcn[(((long)ih * w + iw) * c + ic) * n + in];
    return total;
}

int qcnn_model_qfp_hwcn2nchw_vect_c(const char *filein, const char *fileout, int qp)
{
    if (!filein || !fileout) return fail(QCNN_E_ARG, "null argument");
    char fn_in[512], fn_out[512];
    snprintf(fn_in, sizeof fn_in, filein, qp);
    snprintf(fn_out, sizeof fn_out, fileout, qp);
    FILE *fi = fopen(fn_in, "rb");
    if (!fi) return fail(QCNN_E_IO, "failed to open file %s", fn_in);
    FILE *fo = fopen(fn_out, "wb");
    if (!fo) {
        fclose(fi);
        return fail(QCNN_E_IO, "failed to open file %s", fn_out);
    }
    int rc = QCNN_OK;
    for (int l = 0; l < 6 && rc == QCNN_OK; ++l) {
        const int ks = L_KS[l], c = L_CIN[l], n = L_COUT[l];
        std::vector<int8_t> win((size_t)ks * ks * c * n), wout((size_t)wsize(l));
        std::vector<int32_t> tail(n + 3);
        if (fread(win.data(), 1, win.size(), fi) != win.size() ||
            fread(tail.data(), 4, tail.size(), fi) != tail.size()) {
            rc = fail(QCNN_E_FORMAT, "%s: short HWCN model file at layer %d", fn_in, l);
            break;
        }
        qcnn_hwcn2nchw_vect_c(win.data(), ks, ks, c, n, wout.data());
        if (fwrite(wout.data(), 1, wout.size(), fo) != wout.size() ||
            fwrite(tail.data(), 4, tail.size(), fo) != tail.size())
            rc = fail(QCNN_E_IO, "%s: write failed", fn_out);
    }
    fclose(fi);
    fclose(fo);
    return rc;
}

long qcnn_hwcn2nchw(const float *hwcn, int h, int w, int c, int n, float *out)
{
    if (!hwcn || !out || h < 1 || w < 1 || c < 1 || n < 1) return fail(QCNN_E_ARG, "bad argument");
    for (int ih = 0; ih < h; ++ih)
        for (int iw = 0; iw < w; ++iw)
            for (int ic = 0; ic < c; ++ic)
                for (int in = 0; in < n; ++in)
                    out[(((long)in * c + ic) * h + ih) * w + iw] = hwcn[(((long)ih * w + iw) * c + ic) * n + in];
    return (long)n * c * h * w;
}

int qcnn_model_hwcn2nchw(const char *filein, const char *fileout, int qp)
{
    if (!filein || !fileout) return fail(QCNN_E_ARG, "null argument");
    char fn_in[512], fn_out[512];
    snprintf(fn_in, sizeof fn_in, filein, qp);
    snprintf(fn_out, sizeof fn_out, fileout, qp);
    FILE *fi = fopen(fn_in, "rb");
    if (!fi) return fail(QCNN_E_IO, "failed to open file %s", fn_in);
    FILE *fo = fopen(fn_out, "wb");
    if (!fo) {
        fclose(fi);
        return fail(QCNN_E_IO, "failed to open file %s", fn_out);
    }
    int rc = QCNN_OK;
    for (int l = 0; l < 6 && rc == QCNN_OK; ++l) {
        const int ks = L_KS[l], c = L_CIN[l], n = L_COUT[l];
        std::vector<float> win((size_t)fwsize(l)), wout((size_t)fwsize(l)), b(n);
        if (fread(win.data(), 4, win.size(), fi) != win.size() || fread(b.data(), 4, b.size(), fi) != b.size()) {
            rc = fail(QCNN_E_FORMAT, "%s: short HWCN model file at layer %d", fn_in, l);
            break;
        }
        qcnn_hwcn2nchw(win.data(), ks, ks, c, n, wout.data());
        if (fwrite(wout.data(), 4, wout.size(), fo) != wout.size() || fwrite(b.data(), 4, b.size(), fo) != b.size())
            rc = fail(QCNN_E_IO, "%s: write failed", fn_out);
    }
    fclose(fi);
    fclose(fo);
    return rc;
}

}  // extern "C"
