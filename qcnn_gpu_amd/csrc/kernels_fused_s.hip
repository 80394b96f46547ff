// Kernel B of the fused forward pass, streaming form: conc1 -> (C3_1 || C3_2) -> requant -> concat ->
// C4 -> applyRes_y residual add + clamp -> I1.x_rec (see fused_common.h for the overall design).
//
// No workgroup barrier, no shared tile: every WAVE is on its own.  A wave owns a strip of 14 output
// columns and walks it down a band of rows, one conc1 row at a time:
//   * conc1 row y1 of the strip (18 pixels = 864 contiguous bytes of the bordered HBM image) is loaded
//     straight into LDS (buffer load with the LDS bit: no register in between) into a ring of 2 ks::R rows,
//     twelve rows ahead of its use; LDS is only the place where the row turns into MFMA B fragments (16
//     region pixels x 3 column shifts), each read once.  The residual input Y travels the same way;
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
// The frame is cut into strips x bands (x frames): thousands of waves, twelve per CU, whose MFMA and vector
// phases interleave as the hardware schedules them.
// Every wave is on its own (no barrier anywhere).  An experiment build (ks::SHARED, measured slower, see
// qcnn_internal.h) lets the waves of a workgroup walk neighbouring strips of one band and share ONE ring of conc1
// rows in three groups of ks::R rows: each wave fetches its share of every row, so the four columns two
// neighbouring strips have in common come from HBM once, and one workgroup barrier per ks::R rows keeps a wave
// from reading a row its neighbours have not fetched yet or fetching into a slot they still read.  Zero padding: conc1 comes with a
// border of zeros in HBM (qcnn_internal.h, conc_bytes), conc2 pixels outside the frame are zeroed in
// registers.
#define STAMP_ROW (wv + 12)  // diagnostic builds: rows 12..15 of the stamp buffer (kernel A owns 0..11)
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

// Diagnostic builds only (make variant ALT=-DQCNN_ABL_...): parts of the row body switched off to see what
// bounds it; such a build computes garbage and is never shipped.
#if defined(QCNN_ABL_NOMFMA)
#define MFMA3(a, b, c) (c)
#else
#define MFMA3(a, b, c) mfma16(a, b, c)
#endif

// Cache policy bits of the conc1 loads.  conc1 is read once (apart from the four columns two neighbouring strips
// share): the non-temporal hint takes 8 % off the kernel, the scope bits sc0 / sc1 change nothing (measured;
// experiment builds pick another policy with -DQCNN_CONC_LOAD_POLICY=n; neither does keeping the shared columns
// as ordinary loads in an instruction of their own: 28.6 against 26.9 us per 1080p frame).  The same hint on the Y loads and the
// byte stores of the output costs 16 %: their lines are shared by neighbouring strips and rows.
#ifndef QCNN_CONC_LOAD_POLICY
#define QCNN_CONC_LOAD_POLICY 1
#endif
#if QCNN_CONC_LOAD_POLICY == 0
#define QCNN_CONC_LOAD_AUX ""
#elif QCNN_CONC_LOAD_POLICY == 1
#define QCNN_CONC_LOAD_AUX " nt"
#elif QCNN_CONC_LOAD_POLICY == 2
#define QCNN_CONC_LOAD_AUX " sc1"
#elif QCNN_CONC_LOAD_POLICY == 3
#define QCNN_CONC_LOAD_AUX " sc0"
#elif QCNN_CONC_LOAD_POLICY == 4
#define QCNN_CONC_LOAD_AUX " sc0 sc1"
#else
#define QCNN_CONC_LOAD_AUX " sc0 sc1 nt"
#endif

template <bool EXACT>
__global__ __launch_bounds__(ks::WAVES * 64) __attribute__((amdgpu_waves_per_eu(3, 3)))
void k_c3c4s(const FusedArgs a, int nstrips, int nseg, int seg_rows, int total_units)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[WG_LDS];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // Workgroups are dealt to the 8 XCDs round robin; XCD x takes the x-th eighth of the units, so that strips that
    // share conc1 columns run on one XCD at about the same time and meet in its L2.
    // (the grid is a multiple of 8 workgroups, so the map is one to one)
    const int per_xcd = gridDim.x >> 3;
    const int wg = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    int strip, seg, f, xw;  // xw: first column of the ring rows in the bordered image
    if constexpr (SHARED) {
        // a workgroup = WAVES neighbouring strips of one band of one frame; total_units counts workgroups.  A wave
        // whose strip lies past the last one still fetches its share of the rows and meets the barriers; it owns
        // no pixel (every test on x0 below fails for it).
        if (wg >= total_units) return;
        const int nsw = (nstrips + WAVES - 1) / WAVES;
        const int sg = wg % nsw, rest = wg / nsw;
        seg = rest % nseg, f = rest / nseg;
        strip = sg * WAVES + wv;
        xw = sg * WAVES * SW;
    } else {
        int unit = wg * WAVES + wv;
        if (unit >= total_units) return;  // the waves of a workgroup never meet again: no barrier anywhere
        strip = unit % nstrips;
        unit /= nstrips;
        seg = unit % nseg, f = unit / nseg;
        xw = strip * SW;
    }
    const int H = a.H, W = a.W;
    const int x0 = strip * SW, ya = seg * seg_rows;
    const int yb = ya + seg_rows < H ? ya + seg_rows : H;  // output rows ya .. yb-1
    if (ya >= yb) return;  // (the same for every wave of a workgroup)
    unsigned char *ring = SHARED ? smem : smem + wv * RING_BYTES;                             // conc1 rows: GROUPS groups of R
    unsigned char *yring = smem + (SHARED ? 1 : WAVES) * RING_BYTES + wv * YRING_BYTES;        // Y rows: 2 R x 16 dwords
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

    // staging: conc1 row m of this band (m = 0, 1, ...) is image row ya - 2 + m = row ya + m of the bordered
    // image; the strip's first column x0 - 2 is column x0 of it: 864 contiguous bytes.  One buffer load with the
    // LDS bit moves a row: lane l < 54 fetches piece l and the hardware writes it to LDS at M0 + 16 l, no register
    // in between (the other ten lanes are switched off for it).  The row goes into slot m % 2R of the wave's
    // ring and is requested while conc1 row m - 2R is processed, whose fragments left that slot a row earlier:
    // twelve rows of flight time.  The residual input Y of an output row travels the same way (a byte load with
    // the LDS bit: lane o < 16 drops pixel x0 + o as a dword at M0 + 4 o) into a ring of its own.
    //
    // The compiler knows nothing of these loads, so EVERY vector memory instruction of the main loop is inline
    // assembly and the waits are counted by hand.  Each conc1 row issues exactly three of them, in this order:
    // the output byte store, the conc1 load, the Y load (a store or load that is not wanted carries an offset
    // past the end of its buffer, which the hardware drops, resp. answers with zeros).  The vector memory
    // instructions of a wave complete in the order issued, and row m needs the conc1 load that row m - 11
    // issued (the fragments of row m + 1) and the Y load that row m - 12 issued: behind the younger of the two
    // 1 + 3 * 10 = 31 instructions may still be in flight.
    const int row_bytes = conc_wp(W) * 48;
    const __amdgpu_buffer_rsrc_t conc_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<int8_t *>(a.conc1 + QCNN_CONC_OFF(a, f)), 0, (int)a.conc_stride, 0x00020000);
    // residual input and output go through buffer descriptors of the frame (at most 2^30 bytes): a scalar row
    // offset plus the lane's column, or an offset past the end for the lanes that own no pixel
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t *>(a.y + (size_t)f * a.frame_stride), 0, H * W, 0x00020000);
    const __amdgpu_buffer_rsrc_t rec_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(a.rec + (size_t)f * a.frame_stride, 0, H * W, 0x00020000);
    constexpr int OOB = 0x7f000000;
    const int lane16 = lane * 16;
    const int y_voff = (lane < SW && x0 + lane < W) ? lane : OOB;
    const unsigned long long row_lanes = (1ull << ROW_PIECES) - 1;

    const unsigned ring_s = __builtin_amdgcn_readfirstlane((unsigned)(size_t)ring);  // LDS byte addresses
    const unsigned yring_s = __builtin_amdgcn_readfirstlane((unsigned)(size_t)yring);
    // this wave's share of a ring row: all of it, or pieces wv * ROW_PIECES .. of the workgroup's shared row
    const int share = SHARED ? wv * ROW_PIECES * 16 : 0;
    // requests conc1 row `crow` of the bordered image into row j of ring group `grp` and the Y pixels of output row
    // `yrow` into row j of Y group `ygrp`
    auto request = [&](int grp, int ygrp, int j, int crow, int yrow, bool clamp) {
        if (clamp) {  // rows past the border / outside the frame are never used: any row inside will do
            crow = crow <= H + 3 ? crow : H + 3;
            yrow = yrow < 0 ? 0 : yrow >= H ? H - 1 : yrow;
        }
#ifdef QCNN_ABL_L2LOADS  // diagnostic builds only: every load hits a few rows that stay in L2
        crow &= 7;
#endif
        const unsigned lds_c = ring_s + grp * GROUP_BYTES + j * ROW_BYTES + share, lds_y = yring_s + (ygrp * R + j) * YSLOT;
#ifdef QCNN_ABL_STRIPMAJOR  // diagnostic builds only (garbage results): what a strip-major conc1 image would give -- the
        // rows of a strip follow one another in memory (864 contiguous bytes each); strips beyond the frame's image wrap
        const int off_c = ((strip % 106) * (H + 4) + crow) * ROW_BYTES + share, off_y = yrow * W + x0;
#else
        const int off_c = crow * row_bytes + xw * 48 + share, off_y = yrow * W + x0;
#endif
        unsigned long long saved;
        // (m0 is written twice and is listed as clobbered, so the compiler keeps nothing of its own in it across the
        // block; between an s_mov to m0 and the LDS-direct load that reads it sits one other instruction, the wait
        // state this hazard needs.  The compiler calls m0 a reserved register and warns about seeing it in a clobber
        // list; the clobber still becomes an implicit definition of m0 on the statement, which is the point.)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
        asm volatile("s_mov_b64 %0, exec\n\t"
                     "s_mov_b32 m0, %1\n\t"
                     "s_mov_b64 exec, %2\n\t"
                     "buffer_load_dwordx4 %3, %4, %5 offen" QCNN_CONC_LOAD_AUX " lds\n\t"
                     "s_mov_b32 m0, %6\n\t"
                     "s_mov_b64 exec, 0x3fff\n\t"
                     "buffer_load_ubyte %7, %8, %9 offen lds\n\t"
                     "s_mov_b64 exec, %0"
                     : "=&s"(saved)
                     : "s"(lds_c), "s"(row_lanes), "v"(lane16), "s"(conc_rsrc), "s"(off_c), "s"(lds_y), "v"(y_voff), "s"(y_rsrc),
                       "s"(off_y)
                     : "memory", "m0");
#pragma clang diagnostic pop
    };
    // the lane's read addresses: B fragments of a conc1 row (+ group * GROUP_BYTES + row * 864 + s * 48), the Y
    // pixel of its column (+ slot * YSLOT)
    const unsigned char *rd = ring + ((SHARED ? wv * SW : 0) + pix) * 48 + gc * 16;
    const unsigned char *rdy = yring + (pix < SW ? pix : 0) * 4;  // (lanes past the strip read a pixel they never use)

    // output: lane (pixel o, lane row 0) owns output column x0 + o, o < 14
    const bool outlane = g == 0 && pix < SW && x0 + pix < W;
    const int out_off = outlane ? x0 + pix : OOB;
    const int addr1 = 4 * lane_of_pixel(pix + 1 < 16 ? pix + 1 : 15, 1);
    const int addr2 = 4 * lane_of_pixel(pix + 2 < 16 ? pix + 2 : 15, 2);
    // conc2 pixels outside the frame are the zero padding of C4.  Everything a lane derives from C4 belongs to one
    // region column, so a column outside the frame is masked once, in the sum the gathers read (colmask); a row
    // outside the frame skips its C4 product.
    const int cx = x0 - 1 + pix;
    const int colmask = (cx >= 0 && cx < W) ? -1 : 0;
    const bool edge_strip = x0 == 0 || x0 + 14 >= W;  // uniform: some region column lies outside the frame
    // applyRes_y: ((acc + bias) * mul + half) >> shift with the bias and the rounding term in one constant (the
    // host has proved that acc, acc + bias and the whole product fit: prove_integer_path)
    const int round4 = (1 << (shift - 1)) + b4i * mul;
    int round4v;  // in a vector register: the multiply-add takes one scalar operand only
    asm volatile("v_mov_b32 %0, %1" : "=v"(round4v) : "s"(round4));

    // conc1 rows to walk: conc1 row m completes conc2 row ya - 3 + m, whose C4 product completes output row
    // ya - 4 + m, which is written while conc1 row m + 1 is processed (so row m writes output row ya - 5 + m):
    // yb - ya + 5 rows
    const int T = (yb - ya + 5 + R - 1) / R;
    // the ring's first 2R rows, all landed before the first row starts
#pragma unroll
    for (int i = 0; i < 2 * R; ++i) request(i / R, i / R, i % R, ya + i, ya - 5 + i, true);
    // (as a builtin: the compiler then knows that its own loads above, the weights and biases, have landed too,
    // and places no wait for them inside the loops)
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
    asm volatile("" ::: "memory");
    if constexpr (SHARED) asm volatile("s_barrier" ::: "memory");  // ... and every other wave's share of them

    v4i acc[3] = {v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}};
    v4i a32[2][2] = {{v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}}, {v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}}};
    int Pa = 0, Pb = 0;            // partial sums (kernel rows) of the two output rows still open
    v4i D4p = {0, 0, 0, 0};        // C4 product of the conc2 row completed by the previous conc1 row

    v4i Bq[2][3];  // B fragments of the current conc1 row (slot row & 1) and of the next one
#pragma unroll
    for (int s = 0; s < 3; ++s) Bq[0][s] = *reinterpret_cast<const v4i *>(rd + s * 48);

    // ring groups of step t: read (g3), next (g3n), refilled by this step's requests (g3r)
    int g3 = 0, g3n = 1, g3r = SHARED ? 2 : 0;
    auto next_groups = [&] {
        if constexpr (SHARED) {
            const int old = g3;
            g3 = g3n, g3n = g3r, g3r = old;
        } else {
            g3 ^= 1, g3n ^= 1, g3r = g3;
        }
    };
    STAMP_INIT();
    // One conc1 row.  CHECK = false: a step whose six conc2 rows lie inside the frame and whose six output rows lie
    // inside the band, in a strip with no column outside the frame; every test below is then known to hold.
    auto row = [&](auto jc, auto checkc, int t) {
        constexpr int j = decltype(jc)::value;
        constexpr bool CHECK = decltype(checkc)::value;
        const int m0 = R * t;
        // ring group of this step, of the next one, and the one this step refills (per-wave rings: two groups, the
        // slot a row has just emptied; shared ring: three, the group all waves left at the last barrier)
        const unsigned char *rdc = rd + g3 * GROUP_BYTES, *rdo = rd + g3n * GROUP_BYTES;
        static_assert(R % 2 == 0 && R % 3 == 0, "the fragment and accumulator slots repeat with the step");
        v4i(&B)[3] = Bq[j & 1];
        v4i(&Bn)[3] = Bq[(j + 1) & 1];
        // conc1 row m + 1 and the Y pixels of this row's output row have landed (see the count above; the last two
        // steps request nothing, so only their 10 stores are certain to be younger)
        if (CHECK && t >= T - 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(31)" ::: "memory");
        if constexpr (SHARED && j == R - 1) {
            // Once per step, ahead of the first read of the next group: this wave's share of that group has landed
            // (its last row was requested R rows ago; behind that load sit its Y load and three instructions of each
            // of the R - 1 rows since: 16; the last steps issue fewer), and after the barrier so has everybody
            // else's.  The barrier also tells that every wave has read the last row of the previous group (its
            // fragments are in registers: lgkmcnt(0)), which the next step refills.
#ifdef QCNN_ABL_NOBAR  // diagnostic builds only (garbage results): what the barrier costs
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
#else
            if (CHECK && t >= T - 3) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
        }
        const int yv = *reinterpret_cast<const int *>(rdy + ((t & 1) * R + j) * YSLOT);
        // the fragments of the next row (the first row of the other group after the last row of this one)
        if constexpr (j + 1 < R) {
#pragma unroll
            for (int s = 0; s < 3; ++s) Bn[s] = *reinterpret_cast<const v4i *>(rdc + (j + 1) * ROW_BYTES + s * 48);
        } else {
#pragma unroll
            for (int s = 0; s < 3; ++s) Bn[s] = *reinterpret_cast<const v4i *>(rdo + s * 48);
        }
        __builtin_amdgcn_sched_barrier(0);
        STAMP();
        // conc1 row m = m0 + j (image row y1 = ya - 2 + m) feeds conc2 rows y1 + 1 (kernel row 0: a new
        // accumulator), y1 (kernel row 1) and y1 - 1 (kernel row 2, which completes it: C3_1 of conc2 row
        // y2 = y1 - 1 is then acc[s2]; its C3_2 sits in a32[(j + 1) % 2] since the row before).
        //
        // The row is a software pipeline over three conc2 rows, so that no result is waited for where it is
        // produced: the matrix instructions of conc1 row m run next to the requantisation of conc2 row y2
        // (four pieces, each at least two matrix instructions behind the accumulators' last write: inline
        // assembly gets no hazard wait states from the compiler) and to the C4 epilogue of conc2 row y2 - 1,
        // whose C4 product was issued at the end of the previous row (its gathers are issued early in this
        // row and read late).  Successive matrix instructions never use the same accumulator.
        constexpr int s0 = j % 3, s1 = (j + 2) % 3, s2 = (j + 1) % 3;
        constexpr int jp = (j + R - 1) % R;  // the row before this one (its Y pixel)
        const int y2 = ya - 3 + m0 + j;
        acc[s2] = MFMA3(f31[0][2], B[0], acc[s2]);
        acc[s1] = MFMA3(f31[0][1], B[0], acc[s1]);
        __builtin_amdgcn_sched_barrier(0);
        // C4 of conc2 row y2 - 1 (D4p): it is kernel row r of output row y2 - 1 - r + 1; output row y3 = y2 - 2 is
        // complete with it.  Kernel columns: out[o] = P_0[o] + P_1[o + 1] + P_2[o + 2] (lane rows 0, 1, 2).
        int P = Pa + D4p[2];
        if constexpr (CHECK) P &= colmask;
        Pa = Pb + D4p[1];
        Pb = D4p[0];
#ifdef QCNN_ABL_NOGATHER
        const int v1 = P, v2 = P;
#else
        const int v1 = __builtin_amdgcn_ds_bpermute(addr1, P);
        const int v2 = __builtin_amdgcn_ds_bpermute(addr2, P);
#endif
        __builtin_amdgcn_sched_barrier(0);
        acc[s2] = MFMA3(f31[1][2], B[1], acc[s2]);
        acc[s1] = MFMA3(f31[1][1], B[1], acc[s1]);
        acc[s2] = MFMA3(f31[2][2], B[2], acc[s2]);
        acc[s1] = MFMA3(f31[2][1], B[2], acc[s1]);
        acc[s0] = MFMA3(f31[0][0], B[0], b0);
        a32[j % 2][0] = MFMA3(f32[0], B[1], b1);  // C3_2 of conc2 row y1 (its centre pixel): used one row later
        __builtin_amdgcn_sched_barrier(0);
        const v4i c31 = acc[s2];
        const v4i c320 = a32[(j + 1) % 2][0], c321 = a32[(j + 1) % 2][1];
        unsigned d0, d1, d2;
#ifdef QCNN_ABL_NOREQ
        d0 = c31[0] ^ c31[1] ^ c31[2] ^ c31[3], d1 = c320[0] ^ c320[1] ^ c320[2] ^ c320[3], d2 = c321[0] ^ c321[1] ^ c321[2] ^ c321[3];
#define REQ_PIECE(I, LAST) do { } while (0)
#else
#define REQ_PIECE(I, LAST) requant12_piece<I, LAST>(c320[I], c321[I], c31[I], q32, q31, d1, d2, d0)
#endif
        if constexpr (!EXACT) REQ_PIECE(0, false);
        __builtin_amdgcn_sched_barrier(0);
        acc[s0] = MFMA3(f31[1][0], B[1], acc[s0]);
        a32[j % 2][1] = MFMA3(f32[1], B[1], b2);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!EXACT) REQ_PIECE(1, false);
        __builtin_amdgcn_sched_barrier(0);
        acc[s0] = MFMA3(f31[2][0], B[2], acc[s0]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!EXACT) REQ_PIECE(2, false);
        __builtin_amdgcn_sched_barrier(0);
        STAMP();
        // ---- applyRes_y of output row y3 = y2 - 2 (stored only if it lies in the band), then the requests for the
        // slots this row has emptied
        const int y3 = y2 - 2;
        {
            const int sum = P + v1 + v2;  // C4 accumulator (its bias sits in round4)
            int out;
            if constexpr (EXACT) {
                const float u = (float)sum + b4;  // C4.u, inference/cnn.cu:148-155 (b4i is 0 on this path)
                int temp = (int)u;                // applyRes_GPU_y, inference/cnn.cu:515-520
                temp = (int)((unsigned)temp * (unsigned)mul + (1u << (shift - 1))) >> shift;
                const short rec = (short)(yv + temp);
                out = rec > 255 ? 255 : rec < 0 ? 0 : rec;
            } else {
                // the same in integers (the host has proved the operand ranges, see prove_integer_path)
                const int temp = (__mul24(sum, mul) + round4v) >> shift;
                out = min(max(yv + temp, 0), 255);
            }
            const int voff = (!CHECK || (y3 >= ya && y3 < yb)) ? out_off : OOB;
            asm volatile("buffer_store_byte %0, %1, %2, %3 offen" ::"v"(out), "v"(voff), "s"(rec_rsrc), "s"(y3 * W) : "memory");
        }
        // (rows past the last one the band walks are not requested: when the wave ends and its LDS is handed to
        // the next workgroup, no load may still be on its way into it)
        if (!CHECK || t <= T - 3) request(g3r, t & 1, j, ya + m0 + j + 2 * R, y3 + 2 * R, CHECK);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!EXACT) {
            REQ_PIECE(3, true);
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
        // C4 of conc2 row y2: lane (pixel, s) register r = w4[r][s] . conc2[y2][pixel]; read in the next row.
        // A row outside the frame is the zero padding of C4.
        if (!CHECK || (unsigned)y2 < (unsigned)H) D4p = mfma16(f4, v4i{(int)d0, (int)d1, (int)d2, 0}, v4i{0, 0, 0, 0});
        else D4p = v4i{0, 0, 0, 0};
        STAMP();
    };
    // Steps [t_lo, t_hi) need no test: their conc2 rows ya - 3 + R t .. + R - 1 lie inside the frame, their output
    // rows ya - 5 + R t .. + R - 1 inside the band, and the strip has no column outside the frame.  Three loops
    // rather than a test per step: two copies of the unrolled step inside one loop cost the register allocator
    // dozens of spills.
    // ... and the residual rows (output row + 2 R) and the conc1 rows (row + 2 R) they request lie inside the
    // frame / the bordered image, so that the fast loop clamps nothing.
    int t_lo = T, t_hi = T;
    if (!edge_strip) {
        int last = (H + 3 - R - ya < yb + 5 - R - ya ? H + 3 - R - ya : yb + 5 - R - ya);  // R t may be at most this: ya - 3 + R t + R - 1 <= H - 1 and ya - 5 + R t + R - 1 <= yb - 1
        last = last < H + 5 - 3 * R - ya ? last : H + 5 - 3 * R - ya;      // ya - 5 + R t + 3 R - 1 <= H - 1
        last = last < H + 4 - 3 * R - ya ? last : H + 4 - 3 * R - ya;      // ya + (t + 3) R - 1 <= H + 3
        t_lo = ya >= 3 ? 1 : (3 - ya + R - 1) / R > 1 ? (3 - ya + R - 1) / R : 1;
        t_hi = last >= 0 ? last / R + 1 : 0;
        if (t_hi > T - 2) t_hi = T - 2 > 0 ? T - 2 : 0;  // the last two steps request nothing
        if (t_lo > t_hi) t_lo = t_hi;
    }
    int t = 0;
    for (; t < t_lo; ++t) {
        static_for<R>([&](auto jc) { row(jc, std::true_type{}, t); });
        next_groups();
    }
    for (; t < t_hi; ++t) {
        static_for<R>([&](auto jc) { row(jc, std::false_type{}, t); });
        next_groups();
    }
    for (; t < T; ++t) {
        static_for<R>([&](auto jc) { row(jc, std::true_type{}, t); });
        next_groups();
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
    // units the kernel counts: workgroups of WAVES neighbouring strips (shared ring) or single strips
    const long long per_band = ks::SHARED ? (nstrips + ks::WAVES - 1) / ks::WAVES : nstrips;
    const long long units = (long long)a.nframes * per_band * nseg;
    if (units > (1ll << 30)) return hipErrorInvalidValue;
    const long long wgs = ks::SHARED ? units : (units + ks::WAVES - 1) / ks::WAVES;
    const dim3 grid((unsigned)((wgs + 7) / 8 * 8));  // a multiple of 8: see the unit map
    if (exact) hipLaunchKernelGGL(k_c3c4s<true>, grid, dim3(ks::WAVES * 64), 0, s, a, nstrips, nseg, seg_rows, (int)units);
    else hipLaunchKernelGGL(k_c3c4s<false>, grid, dim3(ks::WAVES * 64), 0, s, a, nstrips, nseg, seg_rows, (int)units);
    return hipGetLastError();
}

}  // namespace qcnn
