// gfx950 kernels of the dynamic-quantisation pass qvrcnn::forward (inference/qvrcnn.cu:82-167): the
// per-frame step sizes come from the largest |u| of each layer.  The reference copies every maximum to
// the host and derives the steps there (findMax, inference/mat.cu:28-35; ConcatLayer::concat,
// inference/cnn.cu:291-309), which costs a device round trip per layer.  Here the maxima stay in HBM:
// a one-workgroup bookkeeping kernel turns them into steps and adjusted biases, and the quantisers read
// their divisor from device memory, so a whole pass is one uninterrupted sequence of launches.
#include "qcnn_internal.h"

namespace qcnn {

// findMax (inference/mat.cu:28-96): largest |u|.  |u| >= 0, so the float order is the order of the bit
// patterns and one unsigned atomic max per wave collects it.  *maxbits must be 0 before the launch.
__global__ __launch_bounds__(256) void k_absmax(const float *__restrict__ u, size_t n, unsigned *maxbits)
{
    float m = 0.f;
    const size_t stride = (size_t)gridDim.x * 256;
    const size_t n4 = n / 4;
    const float4 *u4 = reinterpret_cast<const float4 *>(u);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const float4 v = u4[i];
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) m = fmaxf(m, fabsf(u[i]));
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off, 64));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(maxbits, __float_as_uint(m));
}

hipError_t launch_absmax(const void *u, size_t n, unsigned *maxbits, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    const size_t want = (n / 4 + 255) / 256;
    const int grid = (int)(want < 1 ? 1 : want > 2048 ? 2048 : want);
    hipLaunchKernelGGL(k_absmax, dim3(grid), dim3(256), 0, s, (const float *)u, n, maxbits);
    return hipGetLastError();
}

// CovLayer::activate (inference/cnn.cu:161-166): ReLU in place.
__global__ __launch_bounds__(256) void k_relu(float *u, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float v = u[i];
        if (v < 0.f) u[i] = 0.f;
    }
}

hipError_t launch_relu(void *u, size_t n, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_relu, dim3(grid), dim3(256), 0, s, (float *)u, n);
    return hipGetLastError();
}

// The arithmetic of CHW2CHW_VECT_C (inference/mat.cu:209-225) on one value: float32 add, IEEE float32
// division, clip, truncation by the store to char.
__device__ __forceinline__ int quant_div_one(float t, float half, float d)
{
    if (t < 0.f) {
        t = __fdiv_rn(t - half, d);
        return t < -128.f ? -128 : (int)t;
    }
    t = __fdiv_rn(t + half, d);
    return t > 127.f ? 127 : (int)t;
}

// CHW2CHW_VECT_C (inference/mat.cu:197-236): one thread per (pixel, group of 4 channels), one coalesced
// dword store per full group.  The divisor comes from device memory when divisor_dev is not null.
__global__ __launch_bounds__(256) void k_quant_div(const float *__restrict__ u, int8_t *__restrict__ out, size_t hw,
                                                   int C, int divisor, const int *divisor_dev)
{
    if (divisor_dev) divisor = *divisor_dev;
    const float d = (float)divisor, half = (float)(divisor >> 1);
    const size_t groups = (size_t)(C + 3) / 4;
    const size_t total = hw * groups;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const size_t grp = idx / hw, p = idx - grp * hw;
        const int c0 = (int)grp * 4;
        if (c0 + 4 <= C) {
            uint32_t pk = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                pk |= (uint32_t)(uint8_t)quant_div_one(u[(size_t)(c0 + i) * hw + p], half, d) << (8 * i);
            reinterpret_cast<uint32_t *>(out)[grp * hw + p] = pk;
        } else {
            for (int i = 0; c0 + i < C; ++i)
                out[(grp * hw + p) * 4 + i] = (int8_t)quant_div_one(u[(size_t)(c0 + i) * hw + p], half, d);
        }
    }
}

hipError_t launch_quant_div(const void *u, void *out, size_t hw, int channels, int divisor, const int *divisor_dev,
                            hipStream_t s)
{
    const size_t total = hw * (size_t)((channels + 3) / 4);
    if (total == 0) return hipSuccess;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(k_quant_div, dim3(grid), dim3(256), 0, s, (const float *)u, (int8_t *)out, hw, channels, divisor,
                       divisor_dev);
    return hipGetLastError();
}

// mul_shift (inference/mat.cu:248-261): flat, no relayout, no clip.
__global__ __launch_bounds__(256) void k_mul_shift(const float *__restrict__ in, int8_t *__restrict__ out, size_t n,
                                                   int mul, int shift)
{
    const int bias = (int)((1u << (shift - 1)) / (unsigned)mul);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = (int8_t)((int)((unsigned)((int)in[i] + bias) * (unsigned)mul) >> shift);
}

hipError_t launch_mul_shift(const void *in, void *out, size_t n, int mul, int shift, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_mul_shift, dim3(grid), dim3(256), 0, s, (const float *)in, (int8_t *)out, n, mul, shift);
    return hipGetLastError();
}

// adjustBasic (inference/qvrcnn.cu:336-349) on one bias.
__host__ __device__ inline float adjust_basic_one(const int *stepw, const int *stepy, float b, int n)
{
    long long temp = (long long)b;
    for (int k = 0; k < n; ++k) temp *= stepw[k];
    for (int k = 0; k < n; ++k) {
        const int sy = stepy[k];
        temp = temp > 0 ? (temp + (sy >> 1)) / sy : (temp - (sy >> 1)) / sy;
    }
    return (float)temp;
}

__global__ void k_adjust_basic(DynSteps st, const float *b, float *b_adj, int count, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) b_adj[i] = adjust_basic_one(st.stepw, st.stepy, b[i], n);
}

hipError_t launch_adjust_basic(const DynSteps &st, const void *b, void *b_adj, int count, int n, hipStream_t s)
{
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_adjust_basic, dim3((count + 63) / 64), dim3(64), 0, s, st, (const float *)b, (float *)b_adj,
                       count, n);
    return hipGetLastError();
}

// num / d truncated toward zero, as the reference's long long division does.  The four divisors are uniform
// over the kernel, so their reciprocals are computed once, in double: below 2^51 the product num * (1/d) is
// within one of the quotient, and the remainder (exact in 64-bit integers) says in which direction to
// correct.  Larger dividends take the 64-bit division itself.
__device__ __forceinline__ long long div_trunc_by(long long num, int d, double inv)
{
    const long long mag = num < 0 ? -num : num;
    if (mag >= (1ll << 51)) return num / d;
    long long q = (long long)((double)num * inv);
    long long r = num - q * d;
    if (r >= d) {
        q += 1;
        r -= d;
    } else if (r <= -(long long)d) {
        q -= 1;
        r += d;
    }
    if (num >= 0 && r < 0) q -= 1;
    else if (num < 0 && r > 0) q += 1;
    return q;
}

// adjustOutput (inference/qvrcnn.cu:368-385).  The steps come from device memory when st_dev is not null.
__global__ __launch_bounds__(256) void k_adjust_output(DynSteps st, const DynSteps *st_dev, const float *__restrict__ o,
                                                       int8_t *__restrict__ o_adj, int n, size_t num)
{
    if (st_dev) st = *st_dev;
    double inv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) inv[k] = k < n && st.stepw[k] != 0 ? 1.0 / (double)st.stepw[k] : 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < num; i += (size_t)gridDim.x * 256) {
        long long temp = (long long)o[i];
        for (int k = 0; k < n - 1; ++k) temp *= st.stepy[k];
        if (temp > 0) {
#pragma unroll
            for (int k = 3; k >= 0; --k)
                if (k < n) temp = div_trunc_by(temp + (st.stepw[k] >> 1), st.stepw[k], inv[k]);
        } else {
#pragma unroll
            for (int k = 3; k >= 0; --k)
                if (k < n) temp = div_trunc_by(temp - (st.stepw[k] >> 1), st.stepw[k], inv[k]);
        }
        o_adj[i] = (int8_t)temp;
    }
}

hipError_t launch_adjust_output(const DynSteps &st, const DynSteps *st_dev, const void *o, void *o_adj, int n,
                                size_t num, hipStream_t s)
{
    if (num == 0) return hipSuccess;
    const int grid = (int)((num + 255) / 256 < 4096 ? (num + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_adjust_output, dim3(grid), dim3(256), 0, s, st, st_dev, (const float *)o, (int8_t *)o_adj, n,
                       num);
    return hipGetLastError();
}

// adjustOutput_static (inference/qvrcnn.cu:386-397).
__global__ __launch_bounds__(256) void k_adjust_output_static(const float *__restrict__ o, int8_t *__restrict__ o_adj,
                                                              int mul, int shift, size_t num)
{
    const unsigned bias = 1u << (shift - 1);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < num; i += (size_t)gridDim.x * 256)
        o_adj[i] = (int8_t)((int)((unsigned)(int)o[i] * (unsigned)mul + bias) >> shift);
}

hipError_t launch_adjust_output_static(const void *o, void *o_adj, int mul, int shift, size_t num, hipStream_t s)
{
    if (num == 0) return hipSuccess;
    const int grid = (int)((num + 255) / 256 < 4096 ? (num + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_adjust_output_static, dim3(grid), dim3(256), 0, s, (const float *)o, (int8_t *)o_adj, mul,
                       shift, num);
    return hipGetLastError();
}

// applyRes_GPU (inference/cnn.cu:487-502), int8 residual.
__global__ __launch_bounds__(256) void k_apply_res(const uint8_t *__restrict__ x, const int8_t *__restrict__ res,
                                                   uint8_t *__restrict__ rec, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const short r = (short)((short)x[i] + res[i]);
        rec[i] = r > 255 ? (uint8_t)255 : r < 0 ? (uint8_t)0 : (uint8_t)r;
    }
}

hipError_t launch_apply_res(const void *x, const void *res, void *rec, size_t n, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_apply_res, dim3(grid), dim3(256), 0, s, (const uint8_t *)x, (const int8_t *)res, (uint8_t *)rec,
                       n);
    return hipGetLastError();
}

// qvrcnn::insert_w / insert_y (inference/qvrcnn.cu:305-330).
__host__ __device__ inline void insert_w(int *stepw, int v, int layer)
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
__host__ __device__ inline void insert_y(int *stepy, int v, int layer)
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

// Step selection of ConcatLayer::concat (inference/cnn.cu:295-309).
__host__ __device__ inline void concat_steps_impl(float max1, float max2, int sw1, int sw2, int *stepy1, int *stepy2)
{
    int s1 = max1 > 127 ? (int)(max1 / 128 + 1) : 1;
    int s2 = max2 > 127 ? (int)(max2 / 128 + 1) : 1;
    if (sw1 * s2 > sw2 * s1)
        s1 = (sw1 * s2 + (sw2 >> 1)) / sw2;
    else
        s2 = (sw2 * s1 + (sw1 >> 1)) / sw1;
    *stepy1 = s1;
    *stepy2 = s2;
}

void concat_steps(float max1, float max2, int sw1, int sw2, int *stepy1, int *stepy2)
{
    concat_steps_impl(max1, max2, sw1, sw2, stepy1, stepy2);
}

// The host bookkeeping of qvrcnn::forward between two layers, on the device.  One 64-lane workgroup:
// lane 0 turns the maxima of the layer just computed into step sizes and updates the sorted step
// lists, then all lanes rescale the biases of the next layer(s) (adjustBasic).
//   stage 0: start of a pass           -> b_adj of C1 (n = 0)
//   stage 1: after C1                  -> step_y[C1], lists, b_adj of C2_1 / C2_2 (n = 1)
//   stage 2: after C2_1 / C2_2         -> concat steps, lists, b_adj of C3_1 / C3_2 (n = 2)
//   stage 3: after C3_1 / C3_2         -> concat steps, lists, b_adj of C4 (n = 3)
//   stage 4: after C4                  -> stepw list complete (adjustOutput reads it)
__global__ __launch_bounds__(64) void k_dyn_book(DynState *st, int stage)
{
    const int lane = threadIdx.x;
    if (lane == 0) {
        if (stage == 0) {
            for (int k = 0; k < 4; ++k) st->steps.stepw[k] = st->steps.stepy[k] = 0;
            for (int l = 0; l < 6; ++l) st->maxbits[l] = 0, st->step_y[l] = 0, st->max_u[l] = 0;
        } else if (stage == 1) {
            const float mx = __uint_as_float(st->maxbits[0]);
            st->max_u[0] = (int)mx;                          // inference/cnn.cu:174
            st->step_y[0] = (int)(mx / 128 + 1);             // inference/cnn.cu:175
            insert_w(st->steps.stepw, st->step_w[0], 1);
            insert_y(st->steps.stepy, st->step_y[0], 1);
        } else if (stage == 2 || stage == 3) {
            const int a = stage == 2 ? 1 : 3, c = a + 1;
            const float m1 = __uint_as_float(st->maxbits[a]), m2 = __uint_as_float(st->maxbits[c]);
            st->max_u[a] = (int)m1;
            st->max_u[c] = (int)m2;
            concat_steps_impl(m1, m2, st->step_w[a], st->step_w[c], &st->step_y[a], &st->step_y[c]);
            insert_w(st->steps.stepw, st->step_w[a], stage);
            insert_y(st->steps.stepy, st->step_y[a], stage);
        } else if (stage == 4) {
            insert_w(st->steps.stepw, st->step_w[5], 4);
        }
    }
    __syncthreads();
    int first = -1, count = 0;  // biases to rescale: a contiguous range of the 161
    if (stage == 0) first = 0, count = 64;
    else if (stage == 1) first = 64, count = 48;
    else if (stage == 2) first = 112, count = 48;
    else if (stage == 3) first = 160, count = 1;
    if (lane < count)
        st->b_adj[first + lane] = adjust_basic_one(st->steps.stepw, st->steps.stepy, st->b[first + lane], stage);
}

hipError_t launch_dyn_book(DynState *st, int stage, hipStream_t s)
{
    hipLaunchKernelGGL(k_dyn_book, dim3(1), dim3(64), 0, s, st, stage);
    return hipGetLastError();
}

}  // namespace qcnn
