// Winograd F(2x4, 3x3) convolution for gfx950 (MI355X): forward and backward-data of the 3x3 stride-1 layers of the RAM-Net path at the
// training batch — ConvGRU gates / candidate (submodules.py:447-452), residual blocks (:200-215) and every backward-data launch of those
// layers (launches of >= 150 64-channel x 256-pixel output blocks: wino6_eligible) — exact-fp32 arithmetic on v_mfma_f32_32x32x2_f32.
//
//   Y = A2^T [ sum_ci (G2 g G4^T) .* (B2^T d B4) ] A4        F(2,3) down the rows, F(4,3) along the columns (Lavin & Gray 2016)
//
// A 2 x 4 output tile costs a 4 x 6 grid of products: 3 multiplies per output and channel pair against 4 for F(2x2,3x3) and 9 direct.
// Why 2 x 4 and not 4 x 4 (2.25 per output): the register-transform formulation of conv_wino.hip gives every wave ONE ROW of the
// transform grid (the lane builds its row of B^T d B in registers and that row IS the MFMA's A operand).  Four rows = four waves = one
// per SIMD, all equally loaded; six rows put two waves on two of the four SIMDs (75 % of the pipe at best = 3.0 multiplies per output,
// the same as here) or need 288 accumulator registers per wave.  The columns carry the larger transform instead: a wave owns 6 positions
// for 32 tiles x 32 output channels — 96 accumulator registers, two workgroups per CU.  (A 64-channel form — 192 accumulators in AGPRs, 48
// MFMAs per chunk for the same transform work, one wave per SIMD — was built and measured: its main loop is as fast, but with ONE
// workgroup per CU nothing hides a workgroup's prologue and epilogue: fixed cost per launch 74 against 39 us, slower at every depth the
// network has — profiles/r04_h_tuning_notes.md.  Removed.)
// Everything else follows conv_wino.hip: only the raw patch is shared (fused loaders, double-buffered, one barrier per chunk, three
// (pixel, quad) slots per thread for the 18 x 18 / 34 x 10 / 10 x 34 patch of a 256-pixel workgroup tile), weights stream from L2 in
// B-operand lane order one chunk ahead, the waves exchange their column-transformed rows (4 of 6 columns) through LDS once per
// workgroup in front of the two-phase channel-quad epilogue.
#include <stdlib.h>
#include <type_traits>
#include "common.hpp"
#include "conv_epilogue.hpp"
#include "conv_wino_common.hpp"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace ramnet {

// Tuning builds only (tools/probe_wino6.sh, -DRAMNET_PROBE): thread 0 of every workgroup stamps the shader clock at the phase boundaries of
// its life into g_probe6[workgroup][16] (+ the 100 MHz wall counter at entry: slot 7, HW_ID / XCC_ID registers: slots 8 / 9); ramnet_probe6_read copies the table out.
#ifdef RAMNET_PROBE
__device__ unsigned long long g_probe6[16384 * 16];
#define RAMNET_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 16384) g_probe6[blockIdx.x * 16 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define RAMNET_STAMP(k) do { } while (0)
#endif

constexpr int W6_BN = 64;                        // output channels per 64-channel block of the packed weights
constexpr int W6_U_FLOATS = 24 * W6_BN * WK;     // weights of one (chunk, 64-channel block): 24 positions x 64 x 8 = 48 KB

// TXG = tiles per workgroup row: 4 (16 x 16 output pixels), 2 (32 rows x 8 columns) or 8 (8 x 32)
// LDS layout of a patch plane: pixel (py, px) at slot py * PWS + px + ((py >> 1) & 3), PWS = PW + 3.  A lane reads the 16 bytes of pixel
// (2 tty + r, 4 ttx + j) of its tile; ds_read_b128 serves 16 lanes per cycle ({0-3, 12-15, 20-27}, ...) from 64 banks, and with the dense
// layout all 16 fall on the same 16 banks (4-way conflicts: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.65, the LDS pipe ~60 % busy).
// The skew makes the low two bits of the slot index differ between the four tile rows of every lane group and the tile column supplies
// the next two: conflict-free for the three tile shapes (brute-forced over all rows, columns and lane groups).  Planes are padded to
// 4 mod 8 slots so that the two quads of a 16-byte store land on different banks.
template <int TXG> struct R6Geom {
    static constexpr int TYG = 32 / TXG, TH = 2 * TYG, TW = 4 * TXG, PH = TH + 2, PW = TW + 2, PWS = PW + 3;
    static constexpr int PSLOTS = PH * PWS + ((4 - (PH * PWS) % 8) + 8) % 8;
    static constexpr int PLANE = PSLOTS * 4;     // floats of one channel-quad plane of the patch
    static constexpr int PFLOATS = 2 * PLANE;
    static_assert(PH * PW * 2 <= 768, "three patch slots per thread");
};

template <int TXG, int MODE>
__global__ void __launch_bounds__(256, 2) conv_wino_r6_kernel(const ramnet_conv_desc p, const WinoParams q) {
    constexpr int NF = 1;                        // 32-channel output blocks per workgroup (the index algebra below keeps the general form)
    constexpr int RO_LD = NF * 32 + 4;           // row of the exchange buffer [wave 4][column 4][tile 32][channels + pad]
    using G = R6Geom<TXG>;
    constexpr int RP_PLANE = G::PLANE, RP_FLOATS = G::PFLOATS, RPW = G::PWS, RTW = G::TW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *patch = smem;                         // [2 buffers][2 quads][PH x PW pixels][4] + 256 scratch cells; the epilogue reuses the space

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hq = lane >> 5;
    RAMNET_STAMP(0);
#ifdef RAMNET_PROBE
    if (threadIdx.x == 0 && blockIdx.x < 16384) {
        g_probe6[blockIdx.x * 16 + 7] = __builtin_amdgcn_s_memrealtime();
        g_probe6[blockIdx.x * 16 + 8] = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_REG_HW_ID
        g_probe6[blockIdx.x * 16 + 9] = __builtin_amdgcn_s_getreg((31 << 11) | 20);     // HW_REG_XCC_ID
    }
#endif

    // XCD-aware order as in conv_wino.hip: the channel blocks of ONE spatial tile are consecutive on one XCD
    // (round 6: quotients through the float reciprocals the launcher passes — operands far below 2^23, one correction step makes them exact —
    // instead of three integer divisions in the workgroup's set-up)
    auto fdiv = [](int n, int d, float inv) {
        int qv = (int)((float)n * inv);
        const int r = n - qv * d;
        qv += (r >= d ? 1 : 0) - (r < 0 ? 1 : 0);
        return qv;
    };
    const int xslot = blockIdx.x >> 3, xcd = blockIdx.x & 7;
    const int nbl = (q.nblk * (2 / NF)) >> q.xg;
    const int xq = fdiv(xslot, nbl, q.inv_nbl);
    const int nblk_v = ((xslot - xq * nbl) << q.xg) + (xcd & ((1 << q.xg) - 1));
    const int nblk_i = NF == 1 ? nblk_v >> 1 : nblk_v, fh = NF == 1 ? nblk_v & 1 : 0;
    int bid = xq * (8 >> q.xg) + (xcd >> q.xg);
    if (bid >= q.tiles_x * q.tiles_y * p.B) return;
    const int bq = fdiv(bid, q.tiles_x, q.inv_tx);
    const int tx_i = bid - bq * q.tiles_x;
    const int b = fdiv(bq, q.tiles_y, q.inv_ty);
    const int ty_i = bq - b * q.tiles_y;
    const int n0 = nblk_i * W6_BN + fh * 32;
    const int oy0 = ty_i * G::TH, ox0 = tx_i * G::TW;
    const int iy0 = oy0 + q.dy0, ix0 = ox0 + q.dx0;

    // row `wave` of B2^T d: rows (ra, rb) of the tile's 4 x 6 window, t[j] = d[ra][j] + sb * d[rb][j]
    const int tty = l31 / TXG, ttx = l31 % TXG;
    const int ra = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int rb = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
    const float sb = wave == 1 ? 1.f : -1.f;
    const int pra = hq * RP_PLANE + ((2 * tty + ra) * RPW + 4 * ttx + (((2 * tty + ra) >> 1) & 3)) * 4;
    const int prb = hq * RP_PLANE + ((2 * tty + rb) * RPW + 4 * ttx + (((2 * tty + rb) >> 1) & 3)) * 4;
    // weights: [chunk][block64][wave 4][position-in-row 6][n-block 2][lane 64][channel j 4]

    f32x16 acc[6][NF];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][f][r] = 0.f;

    const int nch = q.nchunks;
    const int clast = (nch - 1) * WK;
    WinoPatch<MODE, 3> pr;
    pr.template init<G::PH, G::PW, G::PWS, G::PLANE, true>(q.src, b, iy0, ix0, tid, clast, 2 * RP_FLOATS);
    // weights: buffer loads with the chunk / position offset in a scalar register (no per-load address arithmetic on the vector side)
    const auto wrs = wino_rsrc(p.w, (unsigned)((size_t)nch * q.nblk * W6_U_FLOATS * sizeof(float)));
    const unsigned wvo = (unsigned)((wave * 3072 + fh * 256 + lane * 4) * 4);
    const int wblk = nblk_i * W6_U_FLOATS * 4, wchunk = q.nblk * W6_U_FLOATS * 4;      // bytes
    auto wload = [&](int chunk, int i) {             // B operands (position i >> 1, n-block i & 1) of `chunk`
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wrs, (int)wvo, chunk * wchunk + wblk + i * 1024, 0));
    };
    float4 breg[6][NF];
    float tA[6][4], tB[6][4];                        // row `wave` of B2^T d for the chunk in flight / the next one: [column][channel]
    // A operands (4 channels = 4 K steps) of the positions in flight and the next ones: a ring of 4.  LD = how many positions ahead
    // the column transform runs: consecutive MFMAs of a wave must go to DIFFERENT accumulators (an instruction between two MFMAs on the
    // same accumulator costs ~43 cycles, MI355X_MICROARCH.md): the positions are taken in PAIRS (p, p + 1), alternating, so two
    // positions' operands are live at a time and the transform runs two positions ahead.
    constexpr int LD = 2;
    float vb[4][4];
    float4 qa[6], qb[6];
    float sa[4], sd[4];                              // shared sub-expressions of positions (1, 2) and (3, 4)
    // one slice (i = 0..7) of the column transform B4 of position q from the row t: <= 2 channels per slice
    //   B4^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
    // (vi = ring slot of the position: (q + 2 * chunk parity) & 3 — six positions do not divide the ring of four, so the slots of
    // consecutive chunks are shifted by two: the next chunk's first positions never land on the ones the last MFMAs still read)
    auto colop = [&](int qp, int vi, int i, const float (&t)[6][4]) {
        float(&v)[4] = vb[vi];
        if (qp == 0) {
            if (i < 4) v[i] = fmaf(-5.f, t[2][i], t[4][i]);
            else v[i - 4] = fmaf(4.f, t[0][i - 4], v[i - 4]);
        } else if (qp == 1) {
            if (i < 2) sa[2 * i] = fmaf(-4.f, t[2][2 * i], t[4][2 * i]), sa[2 * i + 1] = fmaf(-4.f, t[2][2 * i + 1], t[4][2 * i + 1]);
            else if (i < 4) sd[2 * i - 4] = fmaf(-4.f, t[1][2 * i - 4], t[3][2 * i - 4]), sd[2 * i - 3] = fmaf(-4.f, t[1][2 * i - 3], t[3][2 * i - 3]);
            else v[i - 4] = sa[i - 4] + sd[i - 4];
        } else if (qp == 2) {
            if (i >= 4) v[i - 4] = sa[i - 4] - sd[i - 4];
        } else if (qp == 3) {
            if (i < 2) sa[2 * i] = t[4][2 * i] - t[2][2 * i], sa[2 * i + 1] = t[4][2 * i + 1] - t[2][2 * i + 1];
            else if (i < 4) sd[2 * i - 4] = t[3][2 * i - 4] - t[1][2 * i - 4], sd[2 * i - 3] = t[3][2 * i - 3] - t[1][2 * i - 3];
            else v[i - 4] = fmaf(2.f, sd[i - 4], sa[i - 4]);
        } else if (qp == 4) {
            if (i >= 4) v[i - 4] = fmaf(-2.f, sd[i - 4], sa[i - 4]);
        } else {
            if (i < 4) v[i] = fmaf(-5.f, t[3][i], t[5][i]);
            else v[i - 4] = fmaf(4.f, t[1][i - 4], v[i - 4]);
        }
    };

    // (requesting the patches of the first TWO chunks together — one memory round trip per workgroup instead of two — measured no
    // difference, round 5: 281.0-281.5 against 280.7-280.8 ms per step; the other workgroup of the CU covers the prologue)
    RAMNET_STAMP(1);
    pr.load(q.src, 0, clast);
#pragma unroll
    for (int i = 0; i < 12; ++i)
        if (!(i & 1)) breg[i >> 1][0] = wload(0, i);
    pr.store(patch, q.src, 0);
    pr.load(q.src, min(WK, clast), clast);
    __syncthreads();
    RAMNET_STAMP(2);
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        const float4 x = ld4(patch + pra + c * 4), y = ld4(patch + prb + c * 4);
        tA[c][0] = x.x + sb * y.x, tA[c][1] = x.y + sb * y.y, tA[c][2] = x.z + sb * y.z, tA[c][3] = x.w + sb * y.w;
    }
#pragma unroll
    for (int qp = 0; qp < LD; ++qp)
#pragma unroll
        for (int i = 0; i < 8; ++i) colop(qp, qp, i, tA);
    pr.store(patch + RP_FLOATS, q.src, min(WK, clast));
    pr.load(q.src, min(2 * WK, clast), clast);
    __syncthreads();
    RAMNET_STAMP(3);
    // One chunk = 6 positions x 4 K steps of MFMAs.  Everything else of the chunk is cut into 48 SLOTS, two behind every MFMA, each a
    // handful of instructions: a block of 12 VALU between two MFMAs is a bubble in the pipe, so no gap carries more than ~6:
    //   slots  0..11   LDS reads of the NEXT chunk's two window rows (column j = slot >> 1)
    //   slots  4..27   their combination t[j][c] = d[ra][j][c] + sb * d[rb][j][c], one channel per slot
    //   slots 8g..8g+7 the column transform of position g + LD (positions 0 .. LD-1 of the NEXT chunk for g + LD >= 6)
    //   slots 28..30   patch of chunk + 2 (registers -> LDS), slots 33 / 35 / 37: request the patch of chunk + 3
    //   behind the last MFMA that reads them: the B operands of the same positions for the next chunk
    auto body = [&](auto par, int chunk, const float (&tc)[6][4], float (&tn)[6][4]) {
        constexpr int PAR = decltype(par)::value;    // chunk & 1 as a compile-time constant: the two instantiations alternate
        const float *pnext = patch + ((chunk + 1) & 1) * RP_FLOATS;     // patch(i+1)
        float *pfree = patch + (chunk & 1) * RP_FLOATS;                 // patch(i), consumed during chunk i-1 -> patch(i+2)
        const int cw = min(chunk + 1, nch - 1);
        const int c2 = min((chunk + 2) * WK, clast), c3 = min((chunk + 3) * WK, clast);
        auto slot = [&](int s) {                     // compile-time constant after unrolling
            const int g = s >> 3, i = s & 7;
            if (s < 12) {
                if (!(s & 1)) qa[s >> 1] = ld4(pnext + pra + (s >> 1) * 4);
                else qb[s >> 1] = ld4(pnext + prb + (s >> 1) * 4);
            }
            if (s >= 4 && s < 28) {
                const int j = (s - 4) >> 2, c = (s - 4) & 3;
                const float x = c == 0 ? qa[j].x : c == 1 ? qa[j].y : c == 2 ? qa[j].z : qa[j].w;
                const float y = c == 0 ? qb[j].x : c == 1 ? qb[j].y : c == 2 ? qb[j].z : qb[j].w;
                tn[j][c] = fmaf(sb, y, x);
                asm volatile("" : "+v"(tn[j][c]));  // (a use at this point: the compiler otherwise sinks the row into the next chunk's code)
            }
            if (g + LD < 6) colop(g + LD, (g + LD + 2 * PAR) & 3, i, tc);
            else colop(g + LD - 6, (g + LD - 6 + 2 * (PAR ^ 1)) & 3, i, tn);
            if (s >= 28 && s < 31) pr.store_slot(pfree, q.src, c2, s - 28);
            if (s == 33 || s == 35 || s == 37) pr.load_slot(q.src, c3, (s - 33) >> 1, clast);
            if ((s & 15) == 15) breg[g - 1][0] = wload(cw, (g - 1) * 2), breg[g][0] = wload(cw, g * 2);
        };
#pragma unroll
        for (int gp = 0; gp < 3; ++gp) {
            const int p0 = 2 * gp, p1 = 2 * gp + 1;
            const float b0[4] = {breg[p0][0].x, breg[p0][0].y, breg[p0][0].z, breg[p0][0].w};
            const float b1[4] = {breg[p1][0].x, breg[p1][0].y, breg[p1][0].z, breg[p1][0].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                __builtin_amdgcn_sched_barrier(0);
                // (weights first: D = [channel][tile] — a lane's register quad is four consecutive channels of its tile, see the exchange)
                acc[p0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[j], vb[(p0 + 2 * PAR) & 3][j], acc[p0][0], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                slot(gp * 16 + j * 4), slot(gp * 16 + j * 4 + 1);
                __builtin_amdgcn_sched_barrier(0);
                acc[p1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b1[j], vb[(p1 + 2 * PAR) & 3][j], acc[p1][0], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                slot(gp * 16 + j * 4 + 2), slot(gp * 16 + j * 4 + 3);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                           // patch(i+2) visible; patch(i+1) free
    };
    // (do-while: nch >= 1.  A `for` loop leaves a path around the loop on which the exchange below would read the zero-initialised
    // accumulators — the compiler then carries 192 zeros in VGPRs across the loop beside the AGPR accumulators and spills the loop's
    // own addresses: 256 VGPRs + 29-85 scratch reloads with `s_waitcnt vmcnt(0)` per chunk, against 197 VGPRs and none this way)
    int chunk = 0;
    do {
        body(std::integral_constant<int, 0>{}, chunk, tA, tB);
        if (chunk + 1 < nch) body(std::integral_constant<int, 1>{}, chunk + 1, tB, tA);      // (uniform over the workgroup)
        chunk += 2;
    } while (chunk < nch);
    RAMNET_STAMP(4);

    // (the epilogue's bias quad is requested here: its round trip passes under the exchange instead of in front of the first output)
    const int qd = tid & 7, nq = n0 + qd * 4;
    const bool nok = nq < p.Cout;
    const float4 bias4 = p.bias ? ld4(p.bias + (nok ? nq : 0)) : f4zero();
    // ---- exchange: column transform of the wave's row (M A4: 4 of 6 columns), all waves -> LDS.
    // A4^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
    // D of the 32x32 MFMA (weights first, round 6): col = lane & 31 (TILE), row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) (channel): registers
    // 4 g .. 4 g + 3 are channels 8 g + 4 hq .. + 3 of the lane's tile — 16 ds_write_b128 instead of 64 ds_write_b32 (the same products in the
    // same order: bit-identical results)
    float *P = smem;
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float c0[4], c1[4], c2[4], c3[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * g + e;
                const float m0 = acc[0][f][r], m1 = acc[1][f][r], m2 = acc[2][f][r], m3 = acc[3][f][r], m4 = acc[4][f][r], m5 = acc[5][f][r];
                const float a12 = m1 + m2, d12 = m1 - m2, a34 = m3 + m4, d34 = m3 - m4;
                c0[e] = m0 + a12 + a34, c1[e] = d12 + 2.f * d34, c2[e] = a12 + 4.f * a34, c3[e] = d12 + 8.f * d34 + m5;
            }
            float *dst = P + ((wave * 4) * 32 + l31) * RO_LD + f * 32 + 8 * g + 4 * hq;
            st4(dst + 0 * 32 * RO_LD, make_float4(c0[0], c0[1], c0[2], c0[3]));
            st4(dst + 1 * 32 * RO_LD, make_float4(c1[0], c1[1], c1[2], c1[3]));
            st4(dst + 2 * 32 * RO_LD, make_float4(c2[0], c2[1], c2[2], c2[3]));
            st4(dst + 3 * 32 * RO_LD, make_float4(c3[0], c3[1], c3[2], c3[3]));
        }
    __syncthreads();
    RAMNET_STAMP(5);
    // ---- channel-quad epilogue.  Thread (tid >> 3, qd = tid & 7) owns channel quad qd of the 8 output pixels pxl = (tid >> 3) + 32 j of the
    // TH x TW tile: one column, rows py0 + j * (32 / TW).  Straight-line code (round 5: a workgroup spent 4.3 us alone / 7.2 us beside its CU
    // partner here, a fifth of its life, in 8 x 4 serial chains of LDS read -> activation -> 64-bit address -> predicated store;
    // tools/probe_wino6.py): every tensor is addressed through a buffer resource over image b with 32-bit byte offsets — a pixel outside the
    // map or a quad beyond Cout gets the offset WOOB, its loads return zero and its stores are dropped, no branch —, and per half (4 pixels)
    // all global operands, then all 12 LDS reads are requested before the first value is used.  The row transform A2^T over the waves:
    // even rows t0 + t1 + t2, odd rows t1 - t2 - t3 = r0 + s r1 + s r2 with the first row and the sign selected by the parity (same sums,
    // same order).  The launcher only selects this kernel for 16-byte-accessible operands (q.vec4), unit output strides and images < 2 GB.
    const int epi = p.epi;
    constexpr int NI = 4, RSTEP = 32 / RTW;                   // pixels per half; rows between consecutive pixels of a thread
    const int px0 = (tid >> 3) % RTW, py0 = (tid >> 3) / RTW;
    const bool colok = nok && ox0 + px0 < p.Wo;
    const unsigned pix0 = (unsigned)((oy0 + py0) * p.WoF + ox0 + px0);      // inside image b (osy = osx = 1, no offsets: launcher)
    const size_t img = (size_t)b * p.HoF * p.WoF;
    const float *lbase = P + ((px0 & 3) * 32 + (px0 >> 2)) * RO_LD + qd * 4;
    auto rsrc_of = [&](const float *ptr, int ld) { return wino_rsrc(ptr ? ptr + img * ld : nullptr, WOOB); };
    auto bld = [](decltype(wino_rsrc(nullptr, 0u)) r, unsigned off) { return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0)); };
    auto bst = [](decltype(wino_rsrc(nullptr, 0u)) r, unsigned off, float4 v) { __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)off, 0, 0); };
    // byte offset of the thread's quad at pixel j of a tensor with row stride ld: the offset at j = 0 (WOOB: no such tensor; uniform) + j
    // uniform steps, OR-ed with the pixel's own WOOB when it has nothing to do (selects of constants: no branch, no per-pixel multiply)
    unsigned bad[2 * NI];
#pragma unroll
    for (int j = 0; j < 2 * NI; ++j) bad[j] = (colok && oy0 + py0 + j * RSTEP < p.Ho) ? 0u : WOOB;
    auto off0_of = [&](int ld, bool have, int dn = 0) { return have ? (pix0 * (unsigned)ld + (unsigned)(nq + dn)) * 4u : WOOB; };
    auto step_of = [&](int ld) { return (unsigned)(RSTEP * p.WoF * ld * 4); };
    const auto r_out = rsrc_of(p.out, p.ldo);
    auto run = [&](auto kind) {
        // 0: linear / ReLU (+ beta * old), 4: sigmoid, 5: sigmoid + h.r (gates), 1: residual + ReLU, 2: GRU blend, 3: GRU backward stage B
        constexpr int K = decltype(kind)::value;
        const bool addold = K == 0 && p.beta != 0.f && (epi == RAMNET_EPI_RELU || epi == RAMNET_EPI_LINEAR);
        const bool relu = epi == RAMNET_EPI_RELU;
        const auto r_e0 = (K >= 1 && K <= 3) ? rsrc_of(p.e0, p.lde0) : r_out;
        const auto r_e1 = (K == 2 || K == 3 || K == 5) ? rsrc_of(p.e1, p.lde1) : r_out;
        const auto r_o1 = (K == 2 || K == 3 || K == 5) ? rsrc_of(p.o1, p.ldo1) : r_out;
        const unsigned o_out = off0_of(p.ldo, true), s_out = step_of(p.ldo);
        const unsigned o_e0 = off0_of(p.lde0, K >= 1 && K <= 3), s_e0 = step_of(p.lde0);
        // (K = 5: the quads of the reset gate only — a per-thread condition, folded into the offset like every other predicate)
        const unsigned o_e1 = off0_of(p.lde1, ((K == 2 || K == 3) && p.e1 != nullptr) || (K == 5 && nq >= p.Cout / 2), (K == 3 || K == 5) ? -(p.Cout / 2) : 0), s_e1 = step_of(p.lde1);
        const unsigned o_o1 = off0_of(p.ldo1, ((K == 2 || K == 3) && p.o1 != nullptr) || (K == 5 && nq >= p.Cout / 2), K == 5 ? -(p.Cout / 2) : 0), s_o1 = step_of(p.ldo1);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            unsigned oo[NI];
            float4 ea[NI], eb[NI], ec[NI], t0[NI], t1[NI], t2[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int j = half * NI + i;
                oo[i] = (o_out + j * s_out) | bad[j];
                if (K == 0) ea[i] = addold ? bld(r_out, oo[i]) : f4zero();      // (uniform)
                if (K >= 1 && K <= 3) ea[i] = bld(r_e0, (o_e0 + j * s_e0) | bad[j]);
                if (K == 2 || K == 3 || K == 5) eb[i] = bld(r_e1, (o_e1 + j * s_e1) | bad[j]);
                if (K == 3) ec[i] = bld(r_out, oo[i]);
            }
            RAMNET_STAMP(11 + 2 * half);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int py = py0 + (half * NI + i) * RSTEP;
                const float *bb = lbase + ((py >> 1) * TXG + (py & 1) * 128) * RO_LD;
                t0[i] = ld4(bb), t1[i] = ld4(bb + 128 * RO_LD), t2[i] = ld4(bb + 256 * RO_LD);
            }
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const float sg = ((py0 + (half * NI + i) * RSTEP) & 1) ? -1.f : 1.f;
                float4 v = make_float4(fmaf(sg, t2[i].x, fmaf(sg, t1[i].x, t0[i].x)), fmaf(sg, t2[i].y, fmaf(sg, t1[i].y, t0[i].y)),
                                       fmaf(sg, t2[i].z, fmaf(sg, t1[i].z, t0[i].z)), fmaf(sg, t2[i].w, fmaf(sg, t1[i].w, t0[i].w)));
                v = f4add(v, bias4);
                if (K == 0) {
                    if (addold) v = make_float4(v.x + p.beta * ea[i].x, v.y + p.beta * ea[i].y, v.z + p.beta * ea[i].z, v.w + p.beta * ea[i].w);
                    if (relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
                } else if (K == 4) {
                    v = make_float4(sigmoidf_(v.x), sigmoidf_(v.y), sigmoidf_(v.z), sigmoidf_(v.w));
                } else if (K == 5) {      // gates: the reset gate's quads also leave h.r (RAMNET_EPI_SIGMOID_HR; other quads: offset WOOB, h = 0)
                    v = make_float4(sigmoidf_(v.x), sigmoidf_(v.y), sigmoidf_(v.z), sigmoidf_(v.w));
                    const float4 h = eb[i];
                    bst(r_o1, (o_o1 + (half * NI + i) * s_o1) | bad[half * NI + i], make_float4(h.x * v.x, h.y * v.y, h.z * v.z, h.w * v.w));
                } else if (K == 1) {
                    v = make_float4(fmaxf(v.x + ea[i].x, 0.f), fmaxf(v.y + ea[i].y, 0.f), fmaxf(v.z + ea[i].z, 0.f), fmaxf(v.w + ea[i].w, 0.f));
                } else if (K == 3) {      // stage B of the ConvGRU backward on the d(h.r) half (RAMNET_EPI_GRU_BWD; conv_epilogue.hpp: gru_bwd_quad)
                    const float4 g = v, r = ea[i], h = eb[i], old = ec[i];
                    bst(r_o1, (o_o1 + (half * NI + i) * s_o1) | bad[half * NI + i], make_float4(g.x * h.x * r.x * (1.0f - r.x), g.y * h.y * r.y * (1.0f - r.y),
                                                                                g.z * h.z * r.z * (1.0f - r.z), g.w * h.w * r.w * (1.0f - r.w)));
                    v = make_float4(old.x + g.x * r.x, old.y + g.y * r.y, old.z + g.z * r.z, old.w + g.w * r.w);
                } else {
                    const float4 o = make_float4(tanhf_(v.x), tanhf_(v.y), tanhf_(v.z), tanhf_(v.w)), u = ea[i], h = eb[i];
                    bst(r_o1, (o_o1 + (half * NI + i) * s_o1) | bad[half * NI + i], o);
                    v = make_float4(h.x * (1.0f - u.x) + o.x * u.x, h.y * (1.0f - u.y) + o.y * u.y, h.z * (1.0f - u.z) + o.z * u.z,
                                    h.w * (1.0f - u.w) + o.w * u.w);
                }
                bst(r_out, oo[i], v);
            }
            RAMNET_STAMP(12 + 2 * half);
        }
    };
    if (q.s2d_shift) {
        // out_s2d (backward-data of a stride-2 5x5 encoder over its space-to-depth view; LINEAR, no bias: checked on the host): output channel
        // quad nq of logical pixel (oy, ox) is channel quad nq - g * C of full-resolution pixel (2 oy + (g >> 1), 2 ox + (g & 1)), g = nq / C
        const int g = nq >> q.s2d_shift;
        const unsigned fp0 = (unsigned)((2 * (oy0 + py0) + (g >> 1)) * p.WoF + 2 * (ox0 + px0) + (g & 1));
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float4 t0[NI], t1[NI], t2[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int py = py0 + (half * NI + i) * RSTEP;
                const float *bb = lbase + ((py >> 1) * TXG + (py & 1) * 128) * RO_LD;
                t0[i] = ld4(bb), t1[i] = ld4(bb + 128 * RO_LD), t2[i] = ld4(bb + 256 * RO_LD);
            }
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int j = half * NI + i;
                const float sg = ((py0 + j * RSTEP) & 1) ? -1.f : 1.f;
                const unsigned off = ((fp0 * (unsigned)p.ldo + (unsigned)(nq - (g << q.s2d_shift))) * 4u + j * 2 * step_of(p.ldo)) | bad[j];
                bst(r_out, off, make_float4(fmaf(sg, t2[i].x, fmaf(sg, t1[i].x, t0[i].x)), fmaf(sg, t2[i].y, fmaf(sg, t1[i].y, t0[i].y)),
                                            fmaf(sg, t2[i].z, fmaf(sg, t1[i].z, t0[i].z)), fmaf(sg, t2[i].w, fmaf(sg, t1[i].w, t0[i].w))));
            }
        }
        return;
    }
    if (epi == RAMNET_EPI_GRU_BLEND) run(std::integral_constant<int, 2>{});
    else if (epi == RAMNET_EPI_RES_RELU) run(std::integral_constant<int, 1>{});
    else if (epi == RAMNET_EPI_GRU_BWD && n0 >= p.Cout / 2) run(std::integral_constant<int, 3>{});      // (a block lies in one half: launcher)
    else if (epi == RAMNET_EPI_SIGMOID) run(std::integral_constant<int, 4>{});
    else if (epi == RAMNET_EPI_SIGMOID_HR) run(std::integral_constant<int, 5>{});
    else run(std::integral_constant<int, 0>{});
#ifdef RAMNET_PROBE
    __builtin_amdgcn_s_waitcnt(0);          // (the stores have left the wave)
    RAMNET_STAMP(6);
    if (threadIdx.x == 0 && blockIdx.x < 16384) g_probe6[blockIdx.x * 16 + 10] = __builtin_amdgcn_s_memrealtime();
#endif
}

// OIHW 3x3 -> U = G2 g G4^T (evaluated in double) in the lane order of the kernel's B operand:
// index = (((((chunk * nblk + nb) * 4 + w) * 6 + pl) * 2 + f) * 64 + lane) * 4 + j  ->
// U[row w][column pl][input channel chunk*8 + 4*(lane >> 5) + j][output channel nb*64 + f*32 + (lane & 31)]
__global__ void pack_weight_wino_r6_kernel(const float *__restrict__ w, float *__restrict__ wp, int Cout, int Cin, int transposed,
                                           int R, int N, int nchunks, int nblk, size_t total) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i & 3), lane = (int)((i >> 2) & 63), f = (int)((i >> 8) & 1);
        size_t jj = i >> 9;
        const int pl = (int)(jj % 6);
        jj /= 6;
        const int wv = (int)(jj & 3);
        jj >>= 2;
        const int nb = (int)(jj % nblk), chunk = (int)(jj / nblk);
        const int n = f * 32 + (lane & 31);
        const int r = chunk * WK + 4 * (lane >> 5) + j;
        const int no = nb * W6_BN + n;
        float v = 0.f;
        if (r < R && no < N) {
            double g[3][3];
            for (int a = 0; a < 3; ++a)
                for (int bb = 0; bb < 3; ++bb)
                    g[a][bb] = transposed ? (double)w[((size_t)r * Cin + no) * 9 + (2 - a) * 3 + (2 - bb)]
                                          : (double)w[((size_t)no * Cin + r) * 9 + a * 3 + bb];
            const double G2[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
            const double G4[6][3] = {{1.0 / 4, 0, 0},           {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                     {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
            double s = 0;
            for (int a = 0; a < 3; ++a)
                for (int bb = 0; bb < 3; ++bb) s += G2[wv][a] * g[a][bb] * G4[pl][bb];
            v = (float)s;
        }
        wp[i] = v;
    }
}

static void wino6_geometry(int Cout, int Cin, int transposed, int &R, int &N, int &nchunks, int &nblk) {
    R = transposed ? Cout : Cin;
    N = transposed ? Cin : Cout;
    nchunks = cdiv(R, WK);
    nblk = cdiv(N, W6_BN);
}

// Workgroup tile of F(2x4,3x3) for an Ho x Wo map: 32 tiles of 2 x 4 pixels as 16 x 16 (TXG 4), 32 x 8 (TXG 2) or 8 x 32 (TXG 8),
// whichever pads the map least; returns the padded area.
static long wino6_tile(int Ho, int Wo, int &txg) {
    const int shapes[3] = {4, 2, 8};
    long best = -1;
    for (int s = 0; s < 3; ++s) {
        const int t = shapes[s], th = 2 * (32 / t), tw = 4 * t;
        const long a = (long)cdiv(Ho, th) * th * cdiv(Wo, tw) * tw;
        if (best < 0 || a < best) best = a, txg = t;
    }
    return best;
}

static bool wino6_vec4(const ramnet_conv_desc &d) {
    auto al16 = [](const void *ptr) { return ptr == nullptr || ((uintptr_t)ptr & 15) == 0; };
    return d.Cout % 4 == 0 && d.ldo % 4 == 0 && al16(d.out) && al16(d.bias) && (!d.o1 || (d.ldo1 % 4 == 0 && al16(d.o1))) &&
           (!d.e0 || (d.lde0 % 4 == 0 && al16(d.e0))) && (!d.e1 || (d.lde1 % 4 == 0 && al16(d.e1)));
}

// Does this WINOGRAD-eligible launch run F(2x4,3x3)?  Dense 3x3 layers with plain / concatenated / masked inputs and the channel-quad
// epilogues (no ConvLSTM cell, no space-to-depth view), 64-channel output blocks, on maps where (a) the 2 x 4 tiling wastes less than
// a quarter of what it saves and (b) the launch still fills the chip with 64-channel workgroups at ONE per CU.
static int g_w6_min_wgs = 150;                  // ramnet_wino2x4_config(): launch-size threshold (64-channel workgroups of 256 pixels)

int wino6_eligible(const ramnet_conv_desc &d, int force) {
    if (d.ntaps != 9 || d.stride != 1 || d.frame) return 0;
    // (space-to-depth views of the stride-2 5x5 encoders, round 5: dense — the zero slices of the view are not skipped here; 3.0 products per
    // output against the 3.5 the F(2x2) kernel realises with its column masks)
    if (d.in_mode != RAMNET_IN_PLAIN && d.in_mode != RAMNET_IN_CAT && d.in_mode != RAMNET_IN_CAT_MUL && d.in_mode != RAMNET_IN_RELUMASK &&
        d.in_mode != RAMNET_IN_S2D) return 0;
    if (d.in_mode == RAMNET_IN_S2D && (d.C0 < WK || (d.C0 & (d.C0 - 1)) != 0)) return 0;
    if (d.out_s2d && (d.out_s2d < 8 || (d.out_s2d & (d.out_s2d - 1)) != 0 || d.Cout != 4 * d.out_s2d || d.epi != RAMNET_EPI_LINEAR || d.bias || d.beta != 0.f ||
                      d.HoF != 2 * d.Ho || d.WoF != 2 * d.Wo || (d.in_mode != RAMNET_IN_PLAIN && d.in_mode != RAMNET_IN_RELUMASK))) return 0;
    if (d.epi == RAMNET_EPI_LSTM || d.Cout % 64 != 0 || !wino6_vec4(d)) return 0;
    if (d.epi == RAMNET_EPI_GRU_BWD && d.Cout % 128 != 0) return 0;          // a 64-channel block lies in one half of [dx | d(h.r)]
    int txg;
    const long a6 = wino6_tile(d.Ho, d.Wo, txg);
    const long a4t = (long)cdiv(d.Wo, 4) * 4 * cdiv(d.Ho, 32) * 32, a4w = (long)cdiv(d.Wo, 16) * 16 * cdiv(d.Ho, 8) * 8;
    const long a4 = a4t < a4w ? a4t : a4w;
    if (force) return 1;                                            // (tests: every structurally eligible launch)
    if (3 * a6 > 4 * a4 * 0.9) return 0;                            // less than 10 % fewer MFMAs: not worth the larger tiles
    const long wgs = a6 / 256 * d.B * cdiv(d.Cout, 64);
    return wgs >= g_w6_min_wgs ? 1 : 0;
}

int launch_wino6(const ramnet_conv_desc &d, hipStream_t st) {
    RAMNET_CHECK_ARG(d.ntaps == 9 && d.stride == 1 && !d.frame && d.epi != RAMNET_EPI_LSTM);
    RAMNET_CHECK_ARG(d.in_mode == RAMNET_IN_PLAIN || d.in_mode == RAMNET_IN_CAT || d.in_mode == RAMNET_IN_CAT_MUL || d.in_mode == RAMNET_IN_RELUMASK ||
                     d.in_mode == RAMNET_IN_S2D);
    auto log2_exact = [](int v) { int sh = 0; while ((1 << sh) < v) ++sh; return (1 << sh) == v ? sh : -1; };
    if (d.in_mode == RAMNET_IN_S2D) RAMNET_CHECK_ARG(d.C0 >= WK && log2_exact(d.C0) > 0);                 // a chunk lies in one parity group
    if (d.in_mode == RAMNET_IN_CAT || d.in_mode == RAMNET_IN_CAT_MUL) RAMNET_CHECK_ARG(d.C0 % WK == 0);   // chunks do not straddle the concatenation
    RAMNET_CHECK_ARG(wino6_vec4(d) && d.osy == 1 && d.osx == 1 && d.ooy == 0 && d.oox == 0);
    if (d.out_s2d) RAMNET_CHECK_ARG(d.out_s2d >= 8 && log2_exact(d.out_s2d) > 0 && d.Cout == 4 * d.out_s2d && d.epi == RAMNET_EPI_LINEAR && !d.bias &&
                                    d.beta == 0.f && d.HoF == 2 * d.Ho && d.WoF == 2 * d.Wo);
    if (d.epi == RAMNET_EPI_GRU_BWD) RAMNET_CHECK_ARG(d.Cout % 128 == 0);
    int dymin = 127, dxmin = 127;
    unsigned seen = 0;
    for (int t = 0; t < 9; ++t) {
        dymin = d.dy[t] < dymin ? d.dy[t] : dymin;
        dxmin = d.dx[t] < dxmin ? d.dx[t] : dxmin;
    }
    for (int t = 0; t < 9; ++t) {
        const int a = d.dy[t] - dymin, c = d.dx[t] - dxmin;
        RAMNET_CHECK_ARG(a >= 0 && a < 3 && c >= 0 && c < 3);
        seen |= 1u << (a * 3 + c);
    }
    RAMNET_CHECK_ARG(seen == 0x1ffu);
    const bool cat = d.in_mode == RAMNET_IN_CAT || d.in_mode == RAMNET_IN_CAT_MUL;
    WinoParams q;
    q.src.x0 = d.x0, q.src.x1 = d.x1, q.src.xm = d.xm;
    q.src.ld0 = d.ld0, q.src.ld1 = d.ld1, q.src.ldm = d.ldm;
    q.src.C0 = d.C0, q.src.Cin = d.C0 + (cat ? d.C1 : 0);
    q.src.mode = d.in_mode, q.src.Hin = d.Hin, q.src.Win = d.Win;
    if (d.in_mode == RAMNET_IN_S2D) q.src.Cin = 4 * d.C0, q.src.ld1 = log2_exact(d.C0);
    q.nchunks = cdiv(q.src.Cin, WK), q.nblk = cdiv(d.Cout, W6_BN);
    int txg;
    wino6_tile(d.Ho, d.Wo, txg);
    q.tiles_x = cdiv(d.Wo, 4 * txg), q.tiles_y = cdiv(d.Ho, 2 * (32 / txg));
    q.dy0 = dymin, q.dx0 = dxmin;
    q.vec4 = 1, q.s2d_shift = d.out_s2d ? log2_exact(d.out_s2d) : 0, q.sparse = 0;
    // XCD-pinned channel groups for weights that do not fit an L2 (conv_wino.hip)
    const size_t wbytes = (size_t)q.nchunks * q.nblk * W6_U_FLOATS * sizeof(float);
    q.xg = wbytes > (12u << 20) ? 2 : wbytes > (3u << 20) ? 1 : 0;
    while (q.xg > 0 && (q.nblk % (1 << q.xg)) != 0) --q.xg;
    const int lanes = 8 >> q.xg;
    const int nf = 1;                                               // 32-channel workgroups
    q.inv_nbl = 1.0f / (float)((q.nblk * (2 / nf)) >> q.xg), q.inv_tx = 1.0f / (float)q.tiles_x, q.inv_ty = 1.0f / (float)q.tiles_y;
    dim3 grid(cdiv(q.tiles_x * q.tiles_y * d.B, lanes) * 8 * ((q.nblk * (2 / nf)) >> q.xg));
    const size_t ex = (size_t)4 * 4 * 32 * (nf * 32 + 4) * sizeof(float);
    {
        const unsigned long long px = (unsigned long long)d.Hin * d.Win * (d.in_mode == RAMNET_IN_S2D ? 4 : 1);
        int ldmax = d.ld0 > d.ld1 ? d.ld0 : d.ld1;
        ldmax = ldmax > d.ldm ? ldmax : d.ldm;
        RAMNET_CHECK_ARG(px * ldmax * 4ull < (unsigned long long)WOOB);        // per-image 32-bit byte offsets
        int lo = d.ldo > d.ldo1 ? d.ldo : d.ldo1;                              // ... of the epilogue's tensors too
        lo = lo > d.lde0 ? lo : d.lde0;
        lo = lo > d.lde1 ? lo : d.lde1;
        RAMNET_CHECK_ARG((unsigned long long)d.HoF * d.WoF * lo * 4ull < (unsigned long long)WOOB);
    }
    note_kernel("conv_wino_r6_kernel<%d,%d>", txg, d.in_mode);
    size_t probe_pad = 0;                       // (probe builds: RAMNET_PROBE_LDS_KB pads the allocation — 90: ONE workgroup per CU)
#ifdef RAMNET_PROBE
    if (const char *e = getenv("RAMNET_PROBE_LDS_KB")) probe_pad = (size_t)atoi(e) * 1024;
#endif
#define RAMNET_GO6(TXv, MDv)                                                                                        \
    case (TXv) * 100 + (MDv): {                                                                                     \
        const size_t pf = (size_t)(2 * R6Geom<TXv>::PFLOATS + 256 * 4) * sizeof(float);                             \
        RAMNET_FULL_LDS((conv_wino_r6_kernel<TXv, MDv>));                                                           \
        hipLaunchKernelGGL((conv_wino_r6_kernel<TXv, MDv>), grid, dim3(256), (ex > pf ? ex : pf) + probe_pad, st, d, q); \
    } break;
#define RAMNET_GO6_TX(TXv)                                                                                          \
    RAMNET_GO6(TXv, RAMNET_IN_PLAIN) RAMNET_GO6(TXv, RAMNET_IN_CAT) RAMNET_GO6(TXv, RAMNET_IN_CAT_MUL) RAMNET_GO6(TXv, RAMNET_IN_RELUMASK) \
    RAMNET_GO6(TXv, RAMNET_IN_S2D)
    switch (txg * 100 + d.in_mode) {
        RAMNET_GO6_TX(4)
        RAMNET_GO6_TX(2)
        RAMNET_GO6_TX(8)
    default:
        RAMNET_CHECK_ARG(!"conv_wino_r6: unsupported (tile, input mode) combination");
    }
#undef RAMNET_GO6_TX
#undef RAMNET_GO6
    RAMNET_LAUNCH_CHECK();
    return 0;
}

}  // namespace ramnet

using namespace ramnet;

#ifdef RAMNET_PROBE
extern "C" int ramnet_probe6_read(unsigned long long *dst, size_t n) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(ramnet::g_probe6), n * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int ramnet_wino2x4_config(int min_wgs) {
    if (min_wgs >= 0) g_w6_min_wgs = min_wgs;
    return 0;
}

extern "C" int ramnet_conv_wino_variant(const ramnet_conv_desc *d, int force) { return d ? wino6_eligible(*d, force) : 0; }

extern "C" size_t ramnet_packed_weight_elems_wino2x4(int Cout, int Cin, int transposed) {
    int R, N, nchunks, nblk;
    wino6_geometry(Cout, Cin, transposed, R, N, nchunks, nblk);
    return (size_t)nchunks * nblk * W6_U_FLOATS;
}

extern "C" int ramnet_pack_weight_wino2x4(const float *w, float *wp, int Cout, int Cin, int transposed, void *stream) {
    RAMNET_CHECK_ARG(w && wp && Cout > 0 && Cin > 0);
    int R, N, nchunks, nblk;
    wino6_geometry(Cout, Cin, transposed, R, N, nchunks, nblk);
    const size_t total = (size_t)nchunks * nblk * W6_U_FLOATS;
    size_t blocks = (total + 255) / 256;
    if (blocks > 65535) blocks = 65535;
    hipLaunchKernelGGL(pack_weight_wino_r6_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, wp, Cout, Cin,
                       transposed, R, N, nchunks, nblk, total);
    RAMNET_LAUNCH_CHECK();
    return 0;
}
