// Kernel B of the fused forward pass, streaming form: conc1 -> (C3_1 || C3_2) -> requant -> concat ->
// C4 -> applyRes_y residual add + clamp -> I1.x_rec (see fused_common.h for the overall design).
//
// No workgroup barrier, no shared tile: every WAVE is on its own.  A wave owns a strip of 14 output
// columns and walks it down a band of rows, one conc1 row at a time:
//   * conc1 row y1 of the strip (18 pixels = 864 contiguous bytes of the bordered HBM image) arrives in
//     LDS in groups of ks::R rows, staged through registers one group ahead of its use; LDS is only the
//     place where the row turns into MFMA B fragments (16 region pixels x 3 column shifts), each read once;
//   * C3_1 (3x3) as nine 16x16x64 MFMAs per row into three rolling accumulators (conc1 row y1 feeds the
//     conc2 rows y1+1, y1, y1-1 with kernel rows 0, 1, 2), C3_2 (1x1) as two more on the centre fragment;
//   * when conc1 row y1 is done, conc2 row y1-1 is complete: its requantised bytes never leave the lane's
//     registers.  In the internal channel order a lane of the C3 epilogue owns 12 consecutive positions of
//     its pixel, which is exactly one k-group of a B operand, so C4 is ONE more MFMA per row whose A operand
//     holds all nine taps as rows (row 4s + r = kernel row r, kernel column s) and whose B operand is the
//     unshifted pixel: lane (pixel, s) register r = w4[r][s] . conc2[row][pixel];
//   * the three kernel rows are summed in the lane over three consecutive conc2 rows (rolling partial sums),
//     the three kernel columns with two cross-lane gathers (ds_bpermute: pixel + s of lane row s), then
//     applyRes_y and a byte store for the 14 output pixels of the row.
// The frame is cut into strips x bands (x frames): thousands of independent waves, twelve per CU, whose
// MFMA and vector phases interleave as the hardware schedules them.  Zero padding: conc1 comes with a
// border of zeros in HBM (qcnn_internal.h, conc_bytes), conc2 pixels outside the frame are zeroed in
// registers.
#include "fused_common.h"

namespace qcnn {

namespace {
using namespace ks;

// lane (of lane row g) that holds region pixel `pix` (0..15): the inverse of kb_pixel_of_lane
__device__ __forceinline__ int lane_of_pixel(int pix, int g)
{
    const int c = (pix & 1) == 0 ? 4 + (pix >> 1) : pix < 8 ? (pix >> 1) : 12 + ((pix - 9) >> 1);
    return 16 * g + c;
}
}  // namespace

template <bool EXACT>
__global__ __launch_bounds__(ks::WAVES * 64) __attribute__((amdgpu_waves_per_eu(3, 3)))
void k_c3c4s(const FusedArgs a, int nstrips, int nseg, int seg_rows, int total_units)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[WAVES * RING_BYTES];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int unit = blockIdx.x * WAVES + wv;
    if (unit >= total_units) return;  // the waves of a workgroup never meet again: no barrier anywhere
    const int strip = unit % nstrips;
    unit /= nstrips;
    const int seg = unit % nseg, f = unit / nseg;
    const int H = a.H, W = a.W;
    const int x0 = strip * SW, ya = seg * seg_rows;
    const int yb = ya + seg_rows < H ? ya + seg_rows : H;  // output rows ya .. yb-1
    if (ya >= yb) return;
    unsigned char *ring = smem + wv * RING_BYTES;
    const int g = lane >> 4;
    const int pix = kb_pixel_of_lane(lane & 15);  // region pixel of this lane: image column x0 - 1 + pix
    const int gc = kb_chunk_of_kgroup(g);

    // A fragments: resident for the whole strip
    v4i f31[3][3], f32[2], f4;
    {
        const v4i *w31 = reinterpret_cast<const v4i *>(a.wpack + PackedOffsets::C3W31) + lane;
        const v4i *w32 = reinterpret_cast<const v4i *>(a.wpack + PackedOffsets::C3W32) + lane;
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int r = 0; r < 3; ++r) f31[s][r] = w31[(s * 3 + r) * 64];
        f32[0] = w32[0];
        f32[1] = w32[64];
        f4 = reinterpret_cast<const v4i *>(a.wpack + PackedOffsets::C4W9)[lane];
    }
    const Requant q31 = a.rq[3], q32 = a.rq[4];
    float bf[12];
    v4i b0 = {0, 0, 0, 0}, b1 = {0, 0, 0, 0}, b2 = {0, 0, 0, 0};  // integer path: bias starts the accumulator
    if constexpr (EXACT) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bf[i] = a.bias_f[BIAS_OFF[3] + 4 * g + i];
            bf[4 + i] = a.bias_f[BIAS_OFF[4] + 4 * g + i];
            bf[8 + i] = a.bias_f[BIAS_OFF[4] + 16 + 4 * g + i];
        }
    } else {
#pragma unroll
        for (int i = 0; i < 12; ++i) bf[i] = 0.f;
        b0 = *reinterpret_cast<const v4i *>(a.bias_i + BIAS_OFF[3] + 4 * g);
        b1 = *reinterpret_cast<const v4i *>(a.bias_i + BIAS_OFF[4] + 4 * g);
        b2 = *reinterpret_cast<const v4i *>(a.bias_i + BIAS_OFF[4] + 16 + 4 * g);
    }
    const float b4 = a.bias_f[BIAS_OFF[5]];
    const int b4i = EXACT ? 0 : a.bias_i[BIAS_OFF[5]];
    const int mul = a.rq[5].mul, shift = a.rq[5].shift;

    // staging: one wave-wide 16-byte load per conc1 row.  Lane l < 54 fetches piece l of the row's 864 bytes; the
    // other ten lanes fetch the 160 bytes behind it (inside the image or its slack) and drop them into LDS behind
    // the row, where the next row's pieces overwrite them (a wave's LDS writes land in order) or, after the last
    // row, the buffer's slack takes them.
    // conc1 row m of this band (m = 0, 1, ...) is image row ya - 2 + m = row ya + m of the bordered image; the
    // strip's first column x0 - 2 is column x0 of it.  The loads are buffer loads: a uniform descriptor of this
    // frame's image, a scalar offset per row, and lane * 16, so that no lane carries an address.
    const size_t row_bytes = (size_t)conc_wp(W) * 48;
    const __amdgpu_buffer_rsrc_t conc_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<int8_t *>(a.conc1 + (size_t)f * a.conc_stride), 0, (int)a.conc_stride, 0x00020000);
    int gstep = ya * (int)row_bytes + x0 * 48;  // byte offset of the current group inside the image
    auto load_row = [&](int i) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(conc_rsrc, lane * 16, gstep + i * (int)row_bytes, 0);
        return v4i{(int)v[0], (int)v[1], (int)v[2], (int)v[3]};
    };
    unsigned char *wr = ring + lane * 16;
    const unsigned char *rd = ring + pix * 48 + gc * 16;  // + row * 864 + s * 48

    // residual input and output: lane (pixel o, lane row 0) owns output column x0 + o, o < 14
    // (addresses as a uniform frame base + a 32-bit lane offset: a frame holds at most 2^30 pixels)
    const bool outlane = g == 0 && pix < SW && x0 + pix < W;
    const unsigned xcol = x0 + pix < W ? x0 + pix : W - 1;
    const uint8_t *yf = a.y + (size_t)f * a.frame_stride;
    uint8_t *recf = a.rec + (size_t)f * a.frame_stride;
    const int addr1 = 4 * lane_of_pixel(pix + 1 < 16 ? pix + 1 : 15, 1);
    const int addr2 = 4 * lane_of_pixel(pix + 2 < 16 ? pix + 2 : 15, 2);
    // conc2 pixels outside the frame are the zero padding of C4.  Everything a lane derives from C4 belongs to one
    // region column, so a column outside the frame is masked once, in the sum the gathers read (colmask); a row
    // outside the frame skips its C4 product.
    const int cx = x0 - 1 + pix;
    const int colmask = (cx >= 0 && cx < W) ? -1 : 0;

    // conc1 rows to walk: output row yb-1 is complete after m = yb - ya + 3
    const int T = (yb - ya + 4 + R - 1) / R;
    v4i S[R];
#pragma unroll
    for (int i = 0; i < R; ++i) S[i] = load_row(i);
    // residual input of output row y3 = ya - 4 + m (m = R t + j): fetched one step ahead, right after the pixel of
    // row m - R has been used
    auto yoff = [&](int y3) { return (unsigned)(y3 < 0 ? 0 : y3 >= H ? H - 1 : y3) * (unsigned)W + xcol; };
    unsigned char ypix[R];
#pragma unroll
    for (int j = 0; j < R; ++j) ypix[j] = yf[yoff(ya - 4 + j)];

    v4i acc[3] = {v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}};
    v4i a32[2][2] = {{v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}}, {v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}}};
    int Pa = 0, Pb = 0;  // partial sums of output rows y2 - 1 and y2 (see below)

    for (int t = 0; t < T; ++t) {
        const int m0 = R * t;
        // the staged group becomes this step's LDS image; the next group and the next step's Y pixels start
        // their way from HBM
#pragma unroll
        for (int i = 0; i < R; ++i) *reinterpret_cast<v4i *>(wr + i * ROW_BYTES) = S[i];  // ascending: see above
        if (t + 1 < T) {
            gstep += R * (int)row_bytes;
#pragma unroll
            for (int i = 0; i < R; ++i) S[i] = load_row(i);
        }
        v4i B[R][3];
#pragma unroll
        for (int s = 0; s < 3; ++s) B[0][s] = *reinterpret_cast<const v4i *>(rd + s * 48);
        static_for<R>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (j + 1 < R) {
#pragma unroll
                for (int s = 0; s < 3; ++s) B[j + 1][s] = *reinterpret_cast<const v4i *>(rd + (j + 1) * ROW_BYTES + s * 48);
            }
            __builtin_amdgcn_sched_barrier(0);
            // conc1 row m = m0 + j (image row y1 = ya - 2 + m) feeds conc2 rows y1 + 1 (kernel row 0: a new
            // accumulator), y1 (kernel row 1) and y1 - 1 (kernel row 2).  Kernel row 2 goes first: it completes
            // conc2 row y2 = y1 - 1 (C3_1 in acc[s2]; its C3_2 sits in a32[(j + 1) % 2] since the row before), whose
            // requantisation is then threaded between the remaining matrix instructions, two of them ahead of the
            // first piece (the accumulator must have left the matrix pipe: inline assembly gets no wait states).
            constexpr int s0 = j % 3, s1 = (j + 2) % 3, s2 = (j + 1) % 3;
#pragma unroll
            for (int s = 0; s < 3; ++s) acc[s2] = mfma16(f31[s][2], B[j][s], acc[s2]);
            __builtin_amdgcn_sched_barrier(0);
            const int y2 = ya - 3 + m0 + j;
            const v4i c31 = acc[s2];
            const v4i c320 = a32[(j + 1) % 2][0], c321 = a32[(j + 1) % 2][1];
            unsigned d0, d1, d2;
            auto other_rows = [&](auto sc) {
                constexpr int s = decltype(sc)::value;
                acc[s1] = mfma16(f31[s][1], B[j][s], acc[s1]);
                acc[s0] = mfma16(f31[s][0], B[j][s], s == 0 ? b0 : acc[s0]);
                __builtin_amdgcn_sched_barrier(0);
            };
            other_rows(std::integral_constant<int, 0>{});
            if constexpr (!EXACT) requant12_piece<0, false>(c320[0], c321[0], c31[0], q32, q31, d1, d2, d0);
            __builtin_amdgcn_sched_barrier(0);
            other_rows(std::integral_constant<int, 1>{});
            if constexpr (!EXACT) requant12_piece<1, false>(c320[1], c321[1], c31[1], q32, q31, d1, d2, d0);
            __builtin_amdgcn_sched_barrier(0);
            other_rows(std::integral_constant<int, 2>{});
            if constexpr (!EXACT) requant12_piece<2, false>(c320[2], c321[2], c31[2], q32, q31, d1, d2, d0);
            __builtin_amdgcn_sched_barrier(0);
            // C3_2 of conc2 row y1 (its centre pixel): used one row later
            a32[j % 2][0] = mfma16(f32[0], B[j][1], b1);
            a32[j % 2][1] = mfma16(f32[1], B[j][1], b2);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!EXACT) {
                requant12_piece<3, true>(c320[3], c321[3], c31[3], q32, q31, d1, d2, d0);
            } else {
                d0 = quant4<EXACT>(c31[0], c31[1], c31[2], c31[3], bf[0], bf[1], bf[2], bf[3], q31);
                d1 = quant4<EXACT>(c320[0], c320[1], c320[2], c320[3], bf[4], bf[5], bf[6], bf[7], q32);
                d2 = quant4<EXACT>(c321[0], c321[1], c321[2], c321[3], bf[8], bf[9], bf[10], bf[11], q32);
            }
            __builtin_amdgcn_sched_barrier(0);
#ifdef QCNN_DEBUG_CONC2
            // diagnostic build only: conc2 as this kernel sees it, [H][W][48] in the internal channel order
            if ((unsigned)y2 < (unsigned)H && colmask && a.dbg) {
                unsigned *dst = reinterpret_cast<unsigned *>(reinterpret_cast<unsigned char *>(a.dbg) + ((size_t)y2 * W + cx) * 48 + 12 * g);
                dst[0] = d0, dst[1] = d1, dst[2] = d2;
            }
#endif
            // C4: lane (pixel, s) register r = w4[r][s] . conc2[y2][pixel]; conc2 row y2 is kernel row r of output
            // row y2 - r + 1
            v4i D4 = {0, 0, 0, 0};
            if ((unsigned)y2 < (unsigned)H) D4 = mfma16(f4, v4i{(int)d0, (int)d1, (int)d2, 0}, v4i{0, 0, 0, 0});
            const int P = (Pa + D4[2]) & colmask;  // output row y3 = y2 - 1: kernel rows 0, 1, 2 summed
            Pa = Pb + D4[1];
            Pb = D4[0];
            const int y3 = y2 - 1;
            if (y3 >= ya && y3 < yb) {
                // kernel columns: out[o] = P_0[o] + P_1[o + 1] + P_2[o + 2] (lane rows 0, 1, 2)
                const int v1 = __builtin_amdgcn_ds_bpermute(addr1, P);
                const int v2 = __builtin_amdgcn_ds_bpermute(addr2, P);
                const int sum = P + v1 + v2 + b4i;
                uint8_t out;
                if constexpr (EXACT) {
                    const float u = (float)sum + b4;  // C4.u, inference/cnn.cu:148-155
                    int temp = (int)u;                // applyRes_GPU_y, inference/cnn.cu:515-520
                    temp = (int)((unsigned)temp * (unsigned)mul + (1u << (shift - 1))) >> shift;
                    const short rec = (short)((int)ypix[j] + temp);
                    out = rec > 255 ? (uint8_t)255 : rec < 0 ? (uint8_t)0 : (uint8_t)rec;
                } else {
                    // the same in integers (the host has proved the operand ranges, see prove_integer_path)
                    const int temp = (__mul24(sum, mul) + (1 << (shift - 1))) >> shift;
                    const int rec = (int)ypix[j] + temp;
                    out = (uint8_t)min(max(rec, 0), 255);
                }
                if (outlane) recf[(unsigned)y3 * (unsigned)W + (unsigned)(x0 + pix)] = out;
            }
            ypix[j] = yf[yoff(y3 + R)];
        });
    }
}

hipError_t launch_fused_c3c4_stream(const FusedArgs &a, bool exact, hipStream_t s)
{
    if (a.nframes < 1 || a.H < 1 || a.W < 1) return hipSuccess;
    const int nstrips = (a.W + ks::SW - 1) / ks::SW;
    const int ncu = a.ncu > 0 ? a.ncu : 256;
    const double slots = 12.0 * ncu;  // waves the device keeps resident (three per SIMD)
    // Bands: every band of rows costs 4 extra conc1 rows (and up to R - 1 wasted ones), every round of waves the
    // device cannot fill completely costs its idle share.  Pick the count with the best product of the two.
    int seg_rows = 0;
    if (const char *sr = getenv("QCNN_SEG_ROWS")) seg_rows = atoi(sr);
    if (seg_rows < 1) {
        double best = -1;
        const int maxseg = a.H / 24 > 1 ? a.H / 24 : 1;
        for (int n = 1; n <= maxseg && n <= 256; ++n) {
            const int rows = (a.H + n - 1) / n;
            const int nn = (a.H + rows - 1) / rows;
            const double rounds = (double)a.nframes * nstrips * nn / slots;
            const double full = rounds / std::ceil(rounds);
            const double eff = full * rows / (rows + 4.0 + (ks::R - 1) / 2.0);
            if (eff > best + 1e-9) best = eff, seg_rows = rows;
        }
    }
    if (seg_rows > a.H) seg_rows = a.H;
    const int nseg = (a.H + seg_rows - 1) / seg_rows;
    const long long units = (long long)a.nframes * nstrips * nseg;
    if (units > (1ll << 30)) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((units + ks::WAVES - 1) / ks::WAVES));
    if (exact) hipLaunchKernelGGL(k_c3c4s<true>, grid, dim3(ks::WAVES * 64), 0, s, a, nstrips, nseg, seg_rows, (int)units);
    else hipLaunchKernelGGL(k_c3c4s<false>, grid, dim3(ks::WAVES * 64), 0, s, a, nstrips, nseg, seg_rows, (int)units);
    return hipGetLastError();
}

}  // namespace qcnn
