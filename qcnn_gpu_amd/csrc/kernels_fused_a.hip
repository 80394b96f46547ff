// Kernel A of the fused forward pass: Y -> I1.ppro -> C1 -> requant -> (C2_1 || C2_2) -> requant ->
// concat, written to HBM as [H][W][48] int8 (see fused_common.h for the overall design).
//
// Persistent, wave-specialised: one 512-thread workgroup per CU walks a list of 16 x 32 output
// tiles.  Waves 0-3 are PRODUCERS (Y tile, im2col, C1 MFMAs, C1 requant epilogue: VALU-heavy),
// waves 4-7 are CONSUMERS (C2 MFMAs, C2 requant epilogue: MFMA-heavy).  Wave w and wave w+4 share
// a SIMD, so each SIMD always has one VALU-bound and one MFMA-bound instruction stream to
// interleave.  The C1 tile is double-buffered in LDS; in step n the producers fill buffer n&1
// with tile n while the consumers read buffer (n-1)&1; one workgroup barrier ends each step.
#include "fused_common.h"

namespace qcnn {

// A fragments of one kernel column s of the C2 stage: C2_2 rows r = 0..4 and, for the three inner
// columns, C2_1 rows r' = 0..2 x two 16-channel tiles.
struct C2Weights {
    v4i f22[5];
    v4i f21[3][2];
};
__device__ __forceinline__ void load_c2_weights(const v4i *w22, const v4i *w21, int s, C2Weights &o)
{
#pragma unroll
    for (int r = 0; r < 5; ++r) o.f22[r] = w22[(s * 5 + r) * 64];
    if (s >= 1 && s <= 3) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            o.f21[r][0] = w21[(((s - 1) * 3 + r) * 2 + 0) * 64];
            o.f21[r][1] = w21[(((s - 1) * 3 + r) * 2 + 1) * 64];
        }
    }
}

// End of a pipeline step: every LDS write of this wave has landed, then the workgroup barrier.
// (A raw s_barrier: global loads and stores in flight are deliberately not waited for.)
__device__ __forceinline__ void step_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <bool EXACT>
__global__ __launch_bounds__(512, 2) void k_c1c2(const FusedArgs a)
{
    using namespace ka;
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
    // [2][C1R][C1C][64] int8 C1 output tiles (chunk-swizzled), one private Y tile per producer,
    // then all 43 C2 A fragments (loaded once per workgroup, read by every consumer tile)
    unsigned char *sYall = smem + 2 * SC1_BYTES;
    unsigned char *sW = smem + 2 * SC1_BYTES + 4 * SYW_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = a.H, W = a.W;
    const int tiles_x = (W + TW - 1) / TW;
    const int tpf = tiles_x * ((H + TH - 1) / TH);  // tiles per frame
    const int ntiles = tpf * a.nframes;
    const int first = blockIdx.x, stride = gridDim.x;
    const int N = first < ntiles ? (ntiles - first + stride - 1) / stride : 0;  // tiles of this workgroup
    STAMP_INIT();

    // C2 weights -> LDS, once.  They are first read after the barrier that ends step 0.
    {
        const v4i *src = reinterpret_cast<const v4i *>(a.wpack + PackedOffsets::C2W22);
        v4i *dst = reinterpret_cast<v4i *>(sW);
        for (int i = tid; i < SW_BYTES / 16; i += 512) dst[i] = src[i];
    }

    if (wv < 4) {
        // =============================== producer ===============================
        // Producer p owns C1 region rows 5p .. 5p+4 (all C1C columns): 180 pixels = 6 MFMA pixel
        // tiles of 32.  It needs Y tile rows 5p .. 5p+8, which it loads itself into a private LDS
        // image, so producers never wait for each other.
        const int p = wv;
        unsigned char *sYw = sYall + p * SYW_BYTES;  // [YWR][YWC] int8 (x - 128), 0 outside the frame
        const v4i *wp = reinterpret_cast<const v4i *>(a.wpack + PackedOffsets::C1W);
        const v4i A0 = wp[lane], A1 = wp[64 + lane];
        const int h = lane >> 5;
        const Requant q = a.rq[0];
        v16i bias0 = {}, bias1 = {};
        float bf0[16], bf1[16];
        if constexpr (!EXACT) {
            bias0 = *reinterpret_cast<const v16i *>(a.bias_i + 16 * h);
            bias1 = *reinterpret_cast<const v16i *>(a.bias_i + 32 + 16 * h);
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                bf0[i] = a.bias_f[16 * h + i];
                bf1[i] = a.bias_f[32 + 16 * h + i];
            }
        }
        constexpr int YLOADS = (9 * YWC + 63) / 64;  // bytes of the private Y tile per lane
        unsigned char ypre[YLOADS];
        auto fetch_y = [&](int t) {
            const int f = t / tpf, tf = t - f * tpf;
            const int ty = tf / tiles_x, tx = tf - ty * tiles_x;
            const int x0 = tx * TW, y0 = ty * TH;
            const uint8_t *yf = a.y + (size_t)f * a.frame_stride;
#pragma unroll
            for (int it = 0; it < YLOADS; ++it) {
                const int idx = it * 64 + lane;
                const int row = idx / YWC, col = idx - row * YWC;
                const int gy = y0 - 4 + PROD_ROWS * p + row, gx = x0 - 4 + col;
                unsigned char v = 0x80;  // I1.ppro of the zero padding: (x - 128) == 0
                if (idx < 9 * YWC && gy >= 0 && gy < H && gx >= 0 && gx < W) v = yf[(size_t)gy * W + gx];
                ypre[it] = v ^ 0x80;
            }
        };
        if (N > 0) fetch_y(first);
        for (int n = 0; n <= N; ++n) {
            STAMP();
            if (n < N) {
                const int t = first + n * stride;
                const int tf = t % tpf;
                const int ty = tf / tiles_x, tx = tf - ty * tiles_x;
                const int x0 = tx * TW, y0 = ty * TH;
                unsigned char *buf = smem + (n & 1) * SC1_BYTES;
#pragma unroll
                for (int it = 0; it < YLOADS; ++it) {
                    const int idx = it * 64 + lane;
                    if (idx < 9 * YWC) sYw[idx] = ypre[it];
                }
                if (n + 1 < N) fetch_y(t + stride);  // next tile's bytes travel while this one is computed
                const unsigned *sYd = reinterpret_cast<const unsigned *>(sYw);
                constexpr int NPIX = PROD_ROWS * C1C;  // 180
#pragma unroll 1
                for (int k = 0; k < (NPIX + 31) / 32; ++k) {
                    const int qi = k * 32 + (lane & 31);
                    const int qc = qi < NPIX ? qi : NPIX - 1;
                    const int il = qc / C1C, j = qc - il * C1C;
                    // B fragment = 16 bytes of im2col: k-group h covers kernel rows 3h..3h+2; bytes
                    // 0..11 = the three rows at columns j..j+3, bytes 12..14 = column j+4, byte 15 = 0
                    const int sh = j & 3;
                    unsigned lo4[3], b5[3];
#pragma unroll
                    for (int rr = 0; rr < 3; ++rr) {
                        const int d = (il + 3 * h + rr) * (YWC / 4) + (j >> 2);
                        const unsigned lo = sYd[d], hi = sYd[d + 1];
                        lo4[rr] = __builtin_amdgcn_alignbyte(hi, lo, sh);
                        b5[rr] = (hi >> (8 * sh)) & 0xffu;
                    }
                    v4i B;
                    B.x = (int)lo4[0];
                    B.y = (int)lo4[1];
                    B.z = (int)lo4[2];
                    B.w = (int)(b5[0] | (b5[1] << 8) | (b5[2] << 16));
                    v16i d0 = mfma32(A0, B, bias0);
                    v16i d1 = mfma32(A1, B, bias1);
                    mfma_drain();
                    const int i = PROD_ROWS * p + il;
                    const int gy = y0 - 2 + i, gx = x0 - 2 + j;
                    const bool inside = gy >= 0 && gy < H && gx >= 0 && gx < W;
                    v4i o0, o1;
                    if constexpr (!EXACT) {
                        o0.x = quant4<false>(d0[0], d0[1], d0[2], d0[3], 0.f, 0.f, 0.f, 0.f, q);
                        o0.y = quant4<false>(d0[4], d0[5], d0[6], d0[7], 0.f, 0.f, 0.f, 0.f, q);
                        o0.z = quant4<false>(d0[8], d0[9], d0[10], d0[11], 0.f, 0.f, 0.f, 0.f, q);
                        o0.w = quant4<false>(d0[12], d0[13], d0[14], d0[15], 0.f, 0.f, 0.f, 0.f, q);
                        o1.x = quant4<false>(d1[0], d1[1], d1[2], d1[3], 0.f, 0.f, 0.f, 0.f, q);
                        o1.y = quant4<false>(d1[4], d1[5], d1[6], d1[7], 0.f, 0.f, 0.f, 0.f, q);
                        o1.z = quant4<false>(d1[8], d1[9], d1[10], d1[11], 0.f, 0.f, 0.f, 0.f, q);
                        o1.w = quant4<false>(d1[12], d1[13], d1[14], d1[15], 0.f, 0.f, 0.f, 0.f, q);
                    } else {
                        o0.x = quant4<true>(d0[0], d0[1], d0[2], d0[3], bf0[0], bf0[1], bf0[2], bf0[3], q);
                        o0.y = quant4<true>(d0[4], d0[5], d0[6], d0[7], bf0[4], bf0[5], bf0[6], bf0[7], q);
                        o0.z = quant4<true>(d0[8], d0[9], d0[10], d0[11], bf0[8], bf0[9], bf0[10], bf0[11], q);
                        o0.w = quant4<true>(d0[12], d0[13], d0[14], d0[15], bf0[12], bf0[13], bf0[14], bf0[15], q);
                        o1.x = quant4<true>(d1[0], d1[1], d1[2], d1[3], bf1[0], bf1[1], bf1[2], bf1[3], q);
                        o1.y = quant4<true>(d1[4], d1[5], d1[6], d1[7], bf1[4], bf1[5], bf1[6], bf1[7], q);
                        o1.z = quant4<true>(d1[8], d1[9], d1[10], d1[11], bf1[8], bf1[9], bf1[10], bf1[11], q);
                        o1.w = quant4<true>(d1[12], d1[13], d1[14], d1[15], bf1[12], bf1[13], bf1[14], bf1[15], q);
                    }
                    if (!inside) {  // zero padding of the C2 input: C1 output does not exist outside the frame
                        o0 = v4i{0, 0, 0, 0};
                        o1 = v4i{0, 0, 0, 0};
                    }
                    if (qi < NPIX) {
                        unsigned char *px = buf + (i * C1C + j) * 64;
                        *reinterpret_cast<v4i *>(px + swz(h, j)) = o0;      // channels 16h .. 16h+15
                        *reinterpret_cast<v4i *>(px + swz(2 + h, j)) = o1;  // channels 32+16h .. 32+16h+15
                    }
                }
            }
            STAMP();
            step_barrier();
        }
    } else {
        // =============================== consumer ===============================
        // C2_1 (3x3, 64 -> 32) and C2_2 (5x5, 64 -> 16) on the TH x TW tile, 16x16x64 MFMA.
        // Consumer = 8 output rows x 16 columns x 48 channels.  LDS row = output row + r, LDS column
        // = output column + s (the 3x3 taps are r-1, s-1 of the 5x5 window).
        const int cw = wv - 4;
        const int c = lane & 15, g = lane >> 4;
        const int cb = (cw & 1) * 16, R0 = (cw >> 1) * 8;
        const v4i *w22 = reinterpret_cast<const v4i *>(sW) + lane;
        const v4i *w21 = reinterpret_cast<const v4i *>(sW + (PackedOffsets::C2W21 - PackedOffsets::C2W22)) + lane;
        const Requant q21 = a.rq[1], q22 = a.rq[2];
        v4i b0 = {0, 0, 0, 0}, b1 = {0, 0, 0, 0}, b2 = {0, 0, 0, 0};
        float bf[12];
        if constexpr (!EXACT) {
            b0 = *reinterpret_cast<const v4i *>(a.bias_i + BIAS_OFF[1] + 4 * g);
            b1 = *reinterpret_cast<const v4i *>(a.bias_i + BIAS_OFF[1] + 16 + 4 * g);
            b2 = *reinterpret_cast<const v4i *>(a.bias_i + BIAS_OFF[2] + 4 * g);
#pragma unroll
            for (int i = 0; i < 12; ++i) bf[i] = 0.f;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                bf[i] = a.bias_f[BIAS_OFF[1] + 4 * g + i];
                bf[4 + i] = a.bias_f[BIAS_OFF[1] + 16 + 4 * g + i];
                bf[8 + i] = a.bias_f[BIAS_OFF[2] + 4 * g + i];
            }
        }
        for (int n = 0; n <= N; ++n) {
            STAMP();
            if (n >= 1) {
                const int t = first + (n - 1) * stride;
                const int f = t / tpf, tf = t - f * tpf;
                const int ty = tf / tiles_x, tx = tf - ty * tiles_x;
                const int x0 = tx * TW, y0 = ty * TH;
                int8_t *concf = a.conc1 + (size_t)f * a.conc_stride;
                const unsigned char *buf = smem + ((n - 1) & 1) * SC1_BYTES;
                v4i acc21[2][8], acc22[8];
#pragma unroll
                for (int y = 0; y < 8; ++y) {
                    acc21[0][y] = b0;
                    acc21[1][y] = b1;
                    acc22[y] = b2;
                }
                const unsigned char *rowbase = buf + (R0 * C1C + cb + c) * 64;
#pragma unroll
                for (int s = 0; s < 5; ++s) {
                    const bool inner_s = s >= 1 && s <= 3;
                    C2Weights cur;
                    load_c2_weights(w22, w21, s, cur);  // A fragments of kernel column s, from LDS
                    const unsigned char *colbase = rowbase + s * 64 + swz(g, cb + c + s);
#pragma unroll
                    for (int yy = 0; yy < 12; ++yy) {
                        const v4i B = *reinterpret_cast<const v4i *>(colbase + yy * (C1C * 64));
#pragma unroll
                        for (int r = 0; r < 5; ++r) {
                            const int y = yy - r;
                            if (y >= 0 && y < 8) {
                                acc22[y] = mfma16(cur.f22[r], B, acc22[y]);
                                if (inner_s && r >= 1 && r <= 3) {
                                    acc21[0][y] = mfma16(cur.f21[r - 1][0], B, acc21[0][y]);
                                    acc21[1][y] = mfma16(cur.f21[r - 1][1], B, acc21[1][y]);
                                }
                            }
                        }
                    }
                }
                // epilogue: lane (pixel c, g) owns channels 4g..4g+3 of each 16-channel tile = 12
                // consecutive bytes at position 12g of the pixel in the internal channel order.
                STAMP();
                mfma_drain();
                const int gx = x0 + cb + c;
#pragma unroll
                for (int y = 0; y < 8; ++y) {
                    const int gy = y0 + R0 + y;
                    const unsigned d0 = quant4<EXACT>(acc21[0][y][0], acc21[0][y][1], acc21[0][y][2], acc21[0][y][3],
                                                      bf[0], bf[1], bf[2], bf[3], q21);
                    const unsigned d1 = quant4<EXACT>(acc21[1][y][0], acc21[1][y][1], acc21[1][y][2], acc21[1][y][3],
                                                      bf[4], bf[5], bf[6], bf[7], q21);
                    const unsigned d2 = quant4<EXACT>(acc22[y][0], acc22[y][1], acc22[y][2], acc22[y][3], bf[8], bf[9],
                                                      bf[10], bf[11], q22);
                    if (gy < H && gx < W) {
                        unsigned *dst = reinterpret_cast<unsigned *>(concf + ((size_t)gy * W + gx) * 48 + 12 * g);
                        dst[0] = d0;
                        dst[1] = d1;
                        dst[2] = d2;
                    }
                }
            }
            STAMP();
            step_barrier();
        }
    }
}

hipError_t launch_fused_c1c2(const FusedArgs &a, bool exact, hipStream_t s)
{
    const int ntiles = ((a.W + ka::TW - 1) / ka::TW) * ((a.H + ka::TH - 1) / ka::TH) * a.nframes;
    if (ntiles < 1) return hipSuccess;
    const int ncu = a.ncu > 0 ? a.ncu : 256;
    dim3 grid(ntiles < ncu ? ntiles : ncu);
    if (exact) hipLaunchKernelGGL(k_c1c2<true>, grid, dim3(512), 0, s, a);
    else hipLaunchKernelGGL(k_c1c2<false>, grid, dim3(512), 0, s, a);
    return hipGetLastError();
}

}  // namespace qcnn
