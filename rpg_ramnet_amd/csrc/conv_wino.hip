// Winograd F(2x2, 3x3) convolution for gfx950 (MI355X): forward and backward-data of the 3x3 stride-1 layers of the
// RAM-Net path — ConvGRU gates / candidate (submodules.py:447-452), ConvLSTM gates + cell (submodules.py:346-358), residual
// blocks (submodules.py:200-215), and the stride-2 5x5 encoders as 3x3 convolutions of the space-to-depth input — exact-fp32
// arithmetic on v_mfma_f32_32x32x2_f32.
//
//   Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A         (Lavin & Gray 2016; 2.25x fewer multiplies than direct)
//
// A workgroup (4 waves) owns 32 Winograd tiles (8 x 16 or 32 x 4 output pixels) and 64 output channels.  The ROWS of the 4 x 4
// transform grid are split over the waves: wave w owns positions 4w .. 4w+3 for all 32 tiles x 64 channels (M = 32 tiles,
// N = 2 x 32 channels, K = 2 input channels per MFMA; 4 x 2 accumulators of 16 = 128 VGPRs).  Lane (tile = lane & 31,
// half = lane >> 5) transforms row w of B^T d B for its tile and the channel quad `half` of the 8-channel chunk — and that IS
// the A operand of the MFMA (lanes 0-31 supply K index 0, lanes 32-63 K index 1: channel j of quad 0 pairs with channel j of
// quad 1), so the transformed input never goes through LDS.  Only the raw input patch is shared: staged once per chunk with the
// fused loaders of the path (concatenation, the GRU's h*r product, the ReLU mask of the backward pass, the space-to-depth view),
// double-buffered, ONE barrier per chunk.  Weights come from global memory (L2-resident) in MFMA B-operand lane order, 16-byte
// loads, one chunk ahead in a register ring.  The output transform needs all four rows of a tile, so once per workgroup the
// waves exchange their column-transformed partial sums (A^T M A keeps 2 of 4 columns) through LDS (70 KB) before the fused
// epilogue (bias / ReLU / sigmoid / residual / GRU blend / ConvLSTM cell, 16-byte channel quads).  XCD-aware workgroup order.
// Measured against the previous formulation (transformed input V[16][32][8] in LDS, 16x16x4 MFMAs, every wave all 16 positions;
// same-box A/B, round 2): 7.38 -> 6.74 us per 8-channel chunk, fixed cost per launch 44 -> 38 us, training step 174 -> 186
// samples/s; profiles/r01_g_winograd_notes.md has the design log of the LDS version.
// Nothing between two MFMAs branches or selects: the input mode and the s2d_5x5 masks are template parameters (one instantiation
// per combination that occurs), the patch prefetch is one buffer load per slot (zero padding = out-of-range offset), the
// concatenation a descriptor / offset select — 235-313 instructions per two chunks instead of 560, training step 190 -> 197.
#include <stdlib.h>
#include <type_traits>
#include "common.hpp"
#include "conv_epilogue.hpp"
#include "conv_wino_common.hpp"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace ramnet {

constexpr int WBN = 64;                       // output channels per workgroup
constexpr int WTH = 8, WTW = 16;              // output pixels per workgroup (4 x 8 tiles of 2 x 2)
constexpr int WPH = WTH + 2, WPW = WTW + 2;   // input patch
constexpr int WU_FLOATS = 16 * WBN * WK;      // weights of one (chunk, 64-channel block): 32 KB

// TX = Winograd tiles per workgroup row: 8 (8 x 16 output pixels) or 2 (32 x 4 pixels: the 43- and 86-pixel-wide maps of the two
// coarse scales lose 2 % instead of 10 % of the MFMA work to the partial last tile column).
constexpr int RO_LD = WBN + 4;                   // row of the exchange buffer [wave 4][column 2][tile 32][64 channels + pad]
constexpr int RO_FLOATS = 4 * 2 * 32 * RO_LD;
template <int TX> struct RGeom {
    static constexpr int TY = 32 / TX, TH = 2 * TY, TW = 2 * TX, PH = TH + 2, PW = TW + 2;
    static constexpr int PLANE = PH * PW * 4;    // floats of one channel-quad plane of the patch
    static constexpr int PFLOATS = 2 * PLANE;
    static_assert(PH * PW * 2 <= 512, "two patch slots per thread");
};

// SP = WinoParams.sparse as a compile-time constant: the dense kernel (SP = 0) carries no trace of the position masks
// NF = 32-channel output blocks per workgroup: 2 (64 channels), or 1 for launches that would otherwise put fewer than one
// workgroup on a CU (batch-1 streaming on the coarse maps: 44-88 workgroups of 32-64 chunks each): twice the workgroups, half the
// MFMAs per chunk and wave — the same transform work per workgroup, but the launch is latency-bound there, not pipe-bound.
template <int TX, int SP, int MODE, int NF = 2>
__global__ void __launch_bounds__(256, 2) conv_wino_r_kernel(const ramnet_conv_desc p, const WinoParams q) {
    // NF = 1: the exchange buffer holds 32 columns (37 KB instead of 70 KB of LDS), so that these workgroups — of DIFFERENT launches:
    // the per-scale update chains of the streaming runtime run side by side — share a CU with the decoders' (building it for three
    // waves per SIMD as well measured no gain on the stream and cost the ConvGRU candidate launch 61 -> 69 us: spills)
    constexpr int RO_LD = NF == 1 ? 32 + 4 : ramnet::RO_LD;
    using G = RGeom<TX>;
    constexpr int RP_PLANE = G::PLANE, RP_FLOATS = G::PFLOATS, RPW = G::PW, RTW = G::TW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *patch = smem;                  // [2 buffers][2 quads][10 x 18 pixels][4]; the epilogue reuses the space

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hq = lane >> 5;

    // XCD-aware order (1-D grid): consecutive workgroup ids are dealt round-robin to the 8 XCDs, each with a private L2.  Within
    // an XCD the 64-channel blocks of ONE spatial tile are consecutive, so that the input patch they all read is fetched
    // from HBM / Infinity Cache once and served to the others by that XCD's L2.
    // Layers whose packed weights exceed an L2 (4 MB) additionally split the channel blocks into 2^xg groups, each pinned to a set
    // of XCDs: an XCD then streams only its group's slice of the weights (which stays resident) at the price of the patch being
    // fetched once per group.  xg = 0: every XCD runs all channel blocks of its tiles.
    const int xslot = blockIdx.x >> 3, xcd = blockIdx.x & 7;
    const int nbl = (q.nblk * (2 / NF)) >> q.xg;                     // (virtual, NF = 1: 32-channel) blocks per group
    const int nblk_v = ((xslot % nbl) << q.xg) + (xcd & ((1 << q.xg) - 1));
    const int nblk_i = NF == 1 ? nblk_v >> 1 : nblk_v, fh = NF == 1 ? nblk_v & 1 : 0;     // 64-channel block, half of it
    int bid = (xslot / nbl) * (8 >> q.xg) + (xcd >> q.xg);
    if (bid >= q.tiles_x * q.tiles_y * p.B) return;
    const int tx_i = bid % q.tiles_x;
    bid /= q.tiles_x;
    const int ty_i = bid % q.tiles_y;
    const int b = bid / q.tiles_y;
    const int n0 = nblk_i * WBN + fh * 32;
    const int oy0 = ty_i * G::TH, ox0 = tx_i * G::TW;
    const int iy0 = oy0 + q.dy0, ix0 = ox0 + q.dx0;

    // row `wave` of B^T d B: rows (ra, rb) of the tile's 4 x 4 window, te = d[ra] + sb * d[rb]
    const int tty = l31 / TX, ttx = l31 % TX;
    const int ra = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int rb = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
    const float sb = wave == 1 ? 1.f : -1.f;
    const int pra = hq * RP_PLANE + ((2 * tty + ra) * RPW + 2 * ttx) * 4;
    const int prb = hq * RP_PLANE + ((2 * tty + rb) * RPW + 2 * ttx) * 4;
    // weights: [chunk][block64][wave 4][position-in-row 4][n-block 2][lane 64][channel j 4]
    const float *wsrc = p.w + (size_t)nblk_i * WU_FLOATS + wave * 2048 + fh * 256 + lane * 4;     // (+ the split's first chunk, below)
    const size_t wchunk = (size_t)q.nblk * WU_FLOATS;

    f32x16 acc[4][NF];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][f][r] = 0.f;

    // Split reduction (q.ksplit > 1; NF = 1 launches far below one workgroup per CU — batch-1 streaming on the coarse maps): workgroup
    // blockIdx.y reduces chunks [ch0, ch0 + nch) only; the partial OUTPUT tiles (the inverse transform is linear) meet in a workspace
    // and the last workgroup to arrive sums them in split order and runs the epilogue (below).  cb = first channel of the range.
    const int ksp = NF == 1 ? q.ksplit : 1;
    const int cps = (q.nchunks + ksp - 1) / ksp, ch0 = NF == 1 ? (int)blockIdx.y * cps : 0;
    const int nch = NF == 1 ? min(cps, q.nchunks - ch0) : q.nchunks;
    const int cb = ch0 * WK;
    const int clast = cb + (nch - 1) * WK;
    if (NF == 1) wsrc += (size_t)ch0 * wchunk;
    WinoPatch<MODE> pr;
    pr.template init<G::PH, G::PW>(q.src, b, iy0, ix0, tid, clast, 2 * RP_FLOATS);
    float4 breg[4][NF];
    float4 tcur[4], tnext[4], ta, tb;
    auto te = [&](float4 x, float4 y) { return make_float4(x.x + sb * y.x, x.y + sb * y.y, x.z + sb * y.z, x.w + sb * y.w); };
    auto f4sub = [](float4 x, float4 y) { return make_float4(x.x - y.x, x.y - y.y, x.z - y.z, x.w - y.w); };

    pr.load(q.src, cb, clast);
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (NF == 2 || !(i & 1)) breg[i >> 1][NF == 2 ? i & 1 : 0] = ld4(wsrc + i * 256);
    pr.store(patch, q.src, cb);
    pr.load(q.src, min(cb + WK, clast), clast);
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c) tcur[c] = te(ld4(patch + pra + c * 4), ld4(patch + prb + c * 4));
    pr.store(patch + RP_FLOATS, q.src, min(cb + WK, clast));
    pr.load(q.src, min(cb + 2 * WK, clast), clast);
    __syncthreads();
    // one chunk: MFMAs on the transformed rows in `tc`, while the rows of the next chunk are built in `tn` (the loop below
    // alternates the two register sets instead of copying them)
    // s2d_5x5 layers (3x3 view of a 5x5 stride-2 filter): a channel group of column parity 1 has no tap at dx = +1, so its column
    // position 3 of G g G^T is zero, and one of row parity 1 none at dy = +1 (row 3 = this kernel's wave 3); for backward-data the
    // filter is flipped (position 0 / wave 0) and the group is that of the OUTPUT channels.  Those MFMAs multiply by exact zeros and
    // are not issued.  runm: bit (pl * 2 + f) = issue MFMA (pl, f); forward: per chunk, backward-data: fixed per workgroup.
    unsigned runm = 0xffu;
    if (SP == 2) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const int g = (n0 + f * 32) >> q.s2d_shift;
            if ((g & 2) && wave == 0) runm &= ~(0x55u << f);
            if (g & 1) runm &= ~(1u << f);
        }
    }
    auto body = [&](int chunk, const float4 (&tc)[4], float4 (&tn)[4]) {
        if (SP == 1) {
            const int g = (cb + chunk * WK) >> q.src.ld1;
            runm = ((g & 2) && wave == 3) ? 0u : (g & 1) ? 0x3fu : 0xffu;
        }
        const float *pnext = patch + ((chunk + 1) & 1) * RP_FLOATS;     // patch(i+1)
        float *pfree = patch + (chunk & 1) * RP_FLOATS;                 // patch(i), consumed during chunk i-1 -> patch(i+2)
        const float *wnext = wsrc + (size_t)min(chunk + 1, nch - 1) * wchunk;
        const int c2 = min(cb + (chunk + 2) * WK, clast), c3 = min(cb + (chunk + 3) * WK, clast);
        auto side = [&](int k) {                    // compile-time constant after unrolling: one slice behind every MFMA
            if (k < 8) {
                if (!(k & 1)) ta = ld4(pnext + pra + (k >> 1) * 4), tb = ld4(pnext + prb + (k >> 1) * 4);
                else tn[k >> 1] = te(ta, tb);
            } else if (k < 10) pr.store_slot(pfree, q.src, c2, k - 8);
            else if (k < 12) pr.load_slot(q.src, c3, k - 10, clast);
        };
#pragma unroll
        for (int pl = 0; pl < 4; ++pl) {
            const float4 v = pl == 0 ? f4sub(tc[0], tc[2]) : pl == 1 ? f4add(tc[1], tc[2]) : pl == 2 ? f4sub(tc[2], tc[1]) : f4sub(tc[1], tc[3]);
            const float va[4] = {v.x, v.y, v.z, v.w};
            const float b0[4] = {breg[pl][0].x, breg[pl][0].y, breg[pl][0].z, breg[pl][0].w};
            const float b1[4] = {breg[pl][NF - 1].x, breg[pl][NF - 1].y, breg[pl][NF - 1].z, breg[pl][NF - 1].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                __builtin_amdgcn_sched_barrier(0);
                if (SP == 0 || ((runm >> (pl * 2)) & 1u)) acc[pl][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[j], b0[j], acc[pl][0], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (NF == 2) {
                    side(pl * 8 + j * 2);
                    __builtin_amdgcn_sched_barrier(0);
                    if (SP == 0 || ((runm >> (pl * 2 + 1)) & 1u)) acc[pl][NF - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[j], b1[j], acc[pl][NF - 1], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    side(pl * 8 + j * 2 + 1);
                } else {
                    side(pl * 4 + j);             // 16 gaps per chunk for the same 12 slices
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            breg[pl][0] = ld4(wnext + (pl * 2) * 256);
            if (NF == 2) breg[pl][NF - 1] = ld4(wnext + (pl * 2 + 1) * 256);
        }
        __syncthreads();                           // patch(i+2) visible; patch(i+1) free
    };
    for (int chunk = 0; chunk < nch; chunk += 2) {
        body(chunk, tcur, tnext);
        if (chunk + 1 < nch) body(chunk + 1, tnext, tcur);      // (uniform over the workgroup)
    }

    // ---- exchange: column transform of the wave's row (M A: 2 of 4 columns), all waves -> LDS.
    // D of the 32x32 MFMA: col = lane & 31 (channel), row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) (tile)
    float *P = smem;
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float m0 = acc[0][f][r], m1 = acc[1][f][r], m2 = acc[2][f][r], m3 = acc[3][f][r];
            const int m = (r & 3) + 8 * (r >> 2) + 4 * hq;
            P[((wave * 2 + 0) * 32 + m) * RO_LD + f * 32 + l31] = m0 + m1 + m2;
            P[((wave * 2 + 1) * 32 + m) * RO_LD + f * 32 + l31] = m1 - m2 - m3;
        }
    __syncthreads();
    // row transform A^T (.) across the waves for output pixel pxl (0..127 of the TH x TW tile) and channels col .. col+3
    auto out4 = [&](int pxl, int col) {
        const int py = pxl / RTW, px = pxl % RTW;
        const float *base = P + ((px & 1) * 32 + (py >> 1) * TX + (px >> 1)) * RO_LD + col;
        const float4 t1 = ld4(base + 1 * 64 * RO_LD), t2 = ld4(base + 2 * 64 * RO_LD);
        if (py & 1) {
            const float4 t3 = ld4(base + 3 * 64 * RO_LD);
            return make_float4(t1.x - t2.x - t3.x, t1.y - t2.y - t3.y, t1.z - t2.z - t3.z, t1.w - t2.w - t3.w);
        }
        const float4 t0 = ld4(base);
        return make_float4(t0.x + t1.x + t2.x, t0.y + t1.y + t2.y, t0.z + t1.z + t2.z, t0.w + t1.w + t2.w);
    };
    const int epi = p.epi;
    if (NF == 2 && q.vec4 && epi == RAMNET_EPI_LSTM) {
        // ConvLSTM cell (submodules.py:346-358): the block's 64 columns are 16 hidden channels x gates (i, f, o, g)
        const int C = p.Cout;
        // (two phases like the quad epilogue below: the four gate biases — ONE quad per thread — and the previous cell state of both
        // pixels are requested before the gates are transformed out of LDS)
        const int qd = tid & 3, chn = nblk_i * 16 + qd * 4;
        const bool cok = chn < C;
        const int chs = cok ? chn : 0;
        const float4 bi = p.bias ? ld4(p.bias + chs) : f4zero(), bf = p.bias ? ld4(p.bias + C + chs) : f4zero();
        const float4 bo = p.bias ? ld4(p.bias + 2 * C + chs) : f4zero(), bg = p.bias ? ld4(p.bias + 3 * C + chs) : f4zero();
        bool ok[2];
        size_t pixv[2];
        int pxlv[2];
        float4 cpv[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            pxlv[i] = (tid + i * 256) >> 2;
            const int oy = oy0 + pxlv[i] / RTW, ox = ox0 + pxlv[i] % RTW;
            ok[i] = cok && oy < p.Ho && ox < p.Wo;
            pixv[i] = ok[i] ? ((size_t)b * p.HoF + (oy * p.osy + p.ooy)) * p.WoF + (ox * p.osx + p.oox) : 0;
            cpv[i] = p.e1 ? ld4(p.e1 + pixv[i] * p.lde1 + chs) : f4zero();
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (!ok[i]) continue;
            const int pxl = pxlv[i];
            const size_t pix = pixv[i];
            const float4 ai = f4add(out4(pxl, qd * 4), bi), af = f4add(out4(pxl, 16 + qd * 4), bf);
            const float4 ao = f4add(out4(pxl, 32 + qd * 4), bo), ag = f4add(out4(pxl, 48 + qd * 4), bg);
            const float4 gi = make_float4(sigmoidf_(ai.x), sigmoidf_(ai.y), sigmoidf_(ai.z), sigmoidf_(ai.w));
            const float4 gf = make_float4(sigmoidf_(af.x), sigmoidf_(af.y), sigmoidf_(af.z), sigmoidf_(af.w));
            const float4 go = make_float4(sigmoidf_(ao.x), sigmoidf_(ao.y), sigmoidf_(ao.z), sigmoidf_(ao.w));
            const float4 gc = make_float4(tanhf_(ag.x), tanhf_(ag.y), tanhf_(ag.z), tanhf_(ag.w));
            const float4 cp = cpv[i];
            const float4 cn = make_float4(gf.x * cp.x + gi.x * gc.x, gf.y * cp.y + gi.y * gc.y, gf.z * cp.z + gi.z * gc.z, gf.w * cp.w + gi.w * gc.w);
            st4(p.out + pix * p.ldo + chn, make_float4(go.x * tanhf_(cn.x), go.y * tanhf_(cn.y), go.z * tanhf_(cn.z), go.w * tanhf_(cn.w)));
            st4(p.o1 + pix * p.ldo1 + chn, cn);
            if (p.o2) {
                float *g = p.o2 + pix * p.ldo2 + chn;
                st4(g, gi), st4(g + C, gf), st4(g + 2 * C, go), st4(g + 3 * C, gc);
            }
        }
        return;
    }
    if (q.vec4 && !q.s2d_shift) {
        // Channel-quad epilogue, straight-line (as in conv_wino6.hip, round 5): the thread owns channel quad qd of the NI output pixels
        // pxl = p0 + PSTEP i of the TH x TW tile — one column, rows RS apart.  Every tensor is addressed through a buffer resource over image
        // b with 32-bit byte offsets (a pixel outside the map or a quad beyond Cout: WOOB — its loads return zero, its stores are dropped; no
        // branch, no 64-bit address arithmetic); all global operands, then all LDS reads of the quads are requested before the first value is
        // used.  Row transform A^T over the waves: even rows t0 + t1 + t2, odd rows t1 - t2 - t3 = r0 + s r1 + s r2 (same sums, same order).
        // One epilogue kind per instantiation.  The launcher checks 16-byte-accessible operands (q.vec4) and images < 2 GB.
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        constexpr int NI = 4 * NF, PSTEP = 256 / (8 * NF), RS = PSTEP / RTW;
        static_assert(PSTEP % RTW == 0, "a thread's pixels lie in one column of the tile");
        const int qd = NF == 2 ? tid & 15 : tid & 7, nq = n0 + qd * 4;       // (the same quad for all i: 256 threads = 16 x 16 quads)
        const bool nok = nq < p.Cout;
        const float4 bias4 = p.bias ? ld4(p.bias + (nok ? nq : 0)) : f4zero();
        const int p0 = NF == 2 ? tid >> 4 : tid >> 3, px0 = p0 % RTW, py0 = p0 / RTW;
        // Split reduction: this workgroup's partial quads -> its slab of the tile's workspace [split][128 pixels x 32 channels]; the LAST
        // arrival (a counter per tile, left at zero for the next launch) adds the slabs in split order — every sum has a fixed order
        // whoever arrives last: bit-reproducible — and goes on to the epilogue, the others are done.  The partials cross XCDs (private
        // L2s): they are written and read with device-scope accesses (write-through / L2-bypassing), ordered around the counter by the
        // wave's own s_waitcnt — a device-scope FENCE instead writes back and invalidates the whole L2 (measured: +30-50 us per launch).
        float4 joined[NI];
        if (NF == 1 && ksp > 1) {
            // (16-byte buffer accesses with the sc1 cache-policy bit = what the compiler emits for device-scope atomics, four floats at a
            // time: as 4-byte atomic stores / loads at a 16-byte lane stride the join cost 12 us per launch)
            constexpr int AUX_SC1 = 16;
            const auto wr = wino_rsrc(q.ws + (size_t)blockIdx.x * ksp * 4096, (unsigned)(ksp * 4096 * 4));
#pragma unroll
            for (int i = 0; i < NI; ++i)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, out4(p0 + PSTEP * i, qd * 4)), wr, (tid + i * 256) * 16, (int)blockIdx.y * 16384, AUX_SC1);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // the wave's stores have left (s_waitcnt), no cache maintenance
            __syncthreads();                           // (also: every out4 read of P is done, P[0] can carry the verdict)
            if (tid == 0) {
                const int arrived = __hip_atomic_fetch_add(q.cnt + blockIdx.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (arrived == ksp - 1) __hip_atomic_store(q.cnt + blockIdx.x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                reinterpret_cast<int *>(P)[0] = arrived == ksp - 1;
            }
            __syncthreads();
            if (!reinterpret_cast<int *>(P)[0]) return;
#pragma unroll
            for (int i = 0; i < NI; ++i) joined[i] = f4zero();
            for (int s = 0; s < ksp; ++s)
#pragma unroll
                for (int i = 0; i < NI; ++i)
                    joined[i] = f4add(joined[i], __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wr, (tid + i * 256) * 16, s * 16384, AUX_SC1)));
        }
        const bool colok = nok && ox0 + px0 < p.Wo;
        const unsigned pix0 = (unsigned)(((oy0 + py0) * p.osy + p.ooy) * p.WoF + (ox0 + px0) * p.osx + p.oox);      // inside image b
        const size_t img = (size_t)b * p.HoF * p.WoF;
        const float *lbase = P + ((px0 & 1) * 32 + (px0 >> 1)) * RO_LD + qd * 4;
        auto rsrc_of = [&](const float *ptr, int ld) { return wino_rsrc(ptr ? ptr + img * ld : nullptr, WOOB); };
        auto bld = [](decltype(wino_rsrc(nullptr, 0u)) r, unsigned off) { return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0)); };
        auto bst = [](decltype(wino_rsrc(nullptr, 0u)) r, unsigned off, float4 v) { __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)off, 0, 0); };
        unsigned bad[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) bad[i] = (colok && oy0 + py0 + i * RS < p.Ho) ? 0u : WOOB;
        auto off0_of = [&](int ld, bool have, int dn = 0) { return have ? (pix0 * (unsigned)ld + (unsigned)(nq + dn)) * 4u : WOOB; };
        auto step_of = [&](int ld) { return (unsigned)(RS * p.osy * p.WoF * ld * 4); };
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
            constexpr int HN = 4;                        // quads per half (NF = 2: two halves, the GRU blend's operands stay in registers)
#pragma unroll
            for (int half = 0; half < NI / HN; ++half) {
                unsigned oo[HN];
                float4 ea[HN], eb[HN], ec[HN], t0[HN], t1[HN], t2[HN];
#pragma unroll
                for (int i = 0; i < HN; ++i) {
                    const int j = half * HN + i;
                    oo[i] = (o_out + j * s_out) | bad[j];
                    if (K == 0) ea[i] = addold ? bld(r_out, oo[i]) : f4zero();      // (uniform)
                    if (K >= 1 && K <= 3) ea[i] = bld(r_e0, (o_e0 + j * s_e0) | bad[j]);
                    if (K == 2 || K == 3 || K == 5) eb[i] = bld(r_e1, (o_e1 + j * s_e1) | bad[j]);
                    if (K == 3) ec[i] = bld(r_out, oo[i]);
                }
                if (!(NF == 1 && ksp > 1)) {
#pragma unroll
                    for (int i = 0; i < HN; ++i) {
                        const int py = py0 + (half * HN + i) * RS;
                        const float *bb = lbase + ((py >> 1) * TX + (py & 1) * 64) * RO_LD;
                        t0[i] = ld4(bb), t1[i] = ld4(bb + 64 * RO_LD), t2[i] = ld4(bb + 128 * RO_LD);
                    }
                }
#pragma unroll
                for (int i = 0; i < HN; ++i) {
                    const int j = half * HN + i;
                    const float sg = ((py0 + j * RS) & 1) ? -1.f : 1.f;
                    float4 v = (NF == 1 && ksp > 1) ? joined[j]
                                                    : make_float4(fmaf(sg, t2[i].x, fmaf(sg, t1[i].x, t0[i].x)), fmaf(sg, t2[i].y, fmaf(sg, t1[i].y, t0[i].y)),
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
                        bst(r_o1, (o_o1 + j * s_o1) | bad[j], make_float4(h.x * v.x, h.y * v.y, h.z * v.z, h.w * v.w));
                    } else if (K == 1) {
                        v = make_float4(fmaxf(v.x + ea[i].x, 0.f), fmaxf(v.y + ea[i].y, 0.f), fmaxf(v.z + ea[i].z, 0.f), fmaxf(v.w + ea[i].w, 0.f));
                    } else if (K == 3) {      // stage B of the ConvGRU backward on the d(h.r) half (RAMNET_EPI_GRU_BWD; conv_epilogue.hpp: gru_bwd_quad)
                        const float4 g = v, r = ea[i], h = eb[i], old = ec[i];
                        bst(r_o1, (o_o1 + j * s_o1) | bad[j], make_float4(g.x * h.x * r.x * (1.0f - r.x), g.y * h.y * r.y * (1.0f - r.y),
                                                                          g.z * h.z * r.z * (1.0f - r.z), g.w * h.w * r.w * (1.0f - r.w)));
                        v = make_float4(old.x + g.x * r.x, old.y + g.y * r.y, old.z + g.z * r.z, old.w + g.w * r.w);
                    } else {
                        const float4 o = make_float4(tanhf_(v.x), tanhf_(v.y), tanhf_(v.z), tanhf_(v.w)), u = ea[i], h = eb[i];
                        bst(r_o1, (o_o1 + j * s_o1) | bad[j], o);
                        v = make_float4(h.x * (1.0f - u.x) + o.x * u.x, h.y * (1.0f - u.y) + o.y * u.y, h.z * (1.0f - u.z) + o.z * u.z,
                                        h.w * (1.0f - u.w) + o.w * u.w);
                    }
                    bst(r_out, oo[i], v);
                }
            }
        };
        if (epi == RAMNET_EPI_GRU_BLEND) run(std::integral_constant<int, 2>{});
        else if (epi == RAMNET_EPI_RES_RELU) run(std::integral_constant<int, 1>{});
        else if (epi == RAMNET_EPI_GRU_BWD && n0 >= p.Cout / 2) run(std::integral_constant<int, 3>{});      // (a block lies in one half: launcher)
        else if (epi == RAMNET_EPI_SIGMOID) run(std::integral_constant<int, 4>{});
        else if (epi == RAMNET_EPI_SIGMOID_HR) run(std::integral_constant<int, 5>{});
        else run(std::integral_constant<int, 0>{});
        return;
    }
#pragma unroll
    for (int i = 0; i < 4 * NF; ++i) {
        const int sl = tid + i * 256, pxl = NF == 2 ? sl >> 4 : sl >> 3, qd = NF == 2 ? sl & 15 : sl & 7;
        const int oy = oy0 + pxl / RTW, ox = ox0 + pxl % RTW, nq = n0 + qd * 4;
        if (oy >= p.Ho || ox >= p.Wo || nq >= p.Cout) continue;
        const float4 y = out4(pxl, qd * 4);
        if (q.s2d_shift) {      // out_s2d: the quad's parity group picks the full-resolution pixel (LINEAR, no bias: checked on the host)
            const int g = nq >> q.s2d_shift;
            const size_t pix = ((size_t)b * p.HoF + 2 * oy + (g >> 1)) * p.WoF + 2 * ox + (g & 1);
            st4(p.out + pix * p.ldo + (nq - (g << q.s2d_shift)), y);
            continue;
        }
        const size_t pix = ((size_t)b * p.HoF + (oy * p.osy + p.ooy)) * p.WoF + (ox * p.osx + p.oox);
        const bool addold = epilogue_addold(p, oy * p.osy + p.ooy, ox * p.osx + p.oox);
        if (q.vec4) {
            epilogue_store4(p, epi, pix, nq, y, addold);
        } else {                // channel counts / strides that rule out 16-byte accesses
            const float ys[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (nq + e < p.Cout) epilogue_store(p, epi, pix, nq + e, ys[e], addold);
        }
    }
}

// OIHW 3x3 -> U = G g G^T (evaluated in double) in the lane order of the kernel's B operand: index = (((((chunk * nblk + nb) * 4 + w) * 4 + pl) * 2 + f) * 64 + lane) * 4 + j  ->
// U[position 4w + pl][input channel chunk*8 + 4*(lane >> 5) + j][output channel nb*64 + f*32 + (lane & 31)]
__global__ void pack_weight_wino_r_kernel(const float *__restrict__ w, float *__restrict__ wp, int Cout, int Cin, int transposed,
                                          int gates, int R, int N, int nchunks, int nblk, size_t total) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i & 3), lane = (int)((i >> 2) & 63), f = (int)((i >> 8) & 1), pl = (int)((i >> 9) & 3), wv = (int)((i >> 11) & 3);
        const size_t jj = i >> 13;
        const int nb = (int)(jj % nblk), chunk = (int)(jj / nblk);
        const int n = f * 32 + (lane & 31), pos = 4 * wv + pl;
        const int r = chunk * WK + 4 * (lane >> 5) + j;
        int no = nb * WBN + n;
        bool ok = r < R && no < N;
        if (gates > 1) {    // ConvLSTM: a 64-column block = 16 hidden channels x (i, f, o, g)
            const int C = N / gates, chn = nb * 16 + (n & 15);
            ok = r < R && chn < C;
            no = (n >> 4) * C + chn;
        }
        float v = 0.f;
        if (ok) {
            double g[3][3];
            for (int a = 0; a < 3; ++a)
                for (int bb = 0; bb < 3; ++bb)
                    g[a][bb] = transposed ? (double)w[((size_t)r * Cin + no) * 9 + (2 - a) * 3 + (2 - bb)]
                                          : (double)w[((size_t)no * Cin + r) * 9 + a * 3 + bb];
            const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
            const int pi = pos >> 2, pj = pos & 3;
            double s = 0;
            for (int a = 0; a < 3; ++a)
                for (int bb = 0; bb < 3; ++bb) s += G[pi][a] * g[a][bb] * G[pj][bb];
            v = (float)s;
        }
        wp[i] = v;
    }
}

static void wino_geometry(int Cout, int Cin, int transposed, int gates, int &R, int &N, int &nchunks, int &nblk) {
    R = transposed ? Cout : Cin;
    N = transposed ? Cin : Cout;
    nchunks = cdiv(R, WK);
    nblk = gates > 1 ? cdiv(N / gates, 16) : cdiv(N, WBN);
}

// Geometry and variant of a launch: everything launch_wino needs, and what ramnet_conv_splitk_bytes sizes the workspace from.
// ksplit = the split of the reduction the launch WANTS (1 = none); it is used when the descriptor carries a workspace.
static int wino_plan(const ramnet_conv_desc &d, WinoParams &q, bool &tall, int &nf, int &ksplit, unsigned &gridx) {
    RAMNET_CHECK_ARG(d.ntaps == 9 && d.stride == 1);
    RAMNET_CHECK_ARG(d.in_mode != RAMNET_IN_UP2X && d.in_mode != RAMNET_IN_UP2X_SKIP);
    if (d.in_mode == RAMNET_IN_CAT || d.in_mode == RAMNET_IN_CAT_MUL) RAMNET_CHECK_ARG(d.C0 % WK == 0);   // chunks do not straddle the concatenation
    auto log2_exact = [](int v) { int sh = 0; while ((1 << sh) < v) ++sh; return (1 << sh) == v ? sh : -1; };
    if (d.in_mode == RAMNET_IN_S2D) RAMNET_CHECK_ARG(d.C0 >= WK && log2_exact(d.C0) > 0);                 // a chunk lies in one parity group
    // the taps must be the dense 3x3 window; which weight slice each one reads is baked into the Winograd pack
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
    q.src.x0 = d.x0, q.src.x1 = d.x1, q.src.xm = d.xm;
    q.src.ld0 = d.ld0, q.src.ld1 = d.ld1, q.src.ldm = d.ldm;
    q.src.C0 = d.C0, q.src.Cin = d.C0 + (cat ? d.C1 : 0);
    q.src.mode = d.in_mode, q.src.Hin = d.Hin, q.src.Win = d.Win;
    if (d.in_mode == RAMNET_IN_S2D) q.src.Cin = 4 * d.C0, q.src.ld1 = log2_exact(d.C0);
    q.nchunks = cdiv(q.src.Cin, WK), q.nblk = d.epi == RAMNET_EPI_LSTM ? cdiv(d.Cout, 16) : cdiv(d.Cout, WBN);
    q.tiles_x = cdiv(d.Wo, WTW), q.tiles_y = cdiv(d.Ho, WTH);
    // 8 x 16 or 32 x 4 output pixels per workgroup, whichever pads the map less
    tall = (long)cdiv(d.Wo, 4) * 4 * cdiv(d.Ho, 32) * 32 < (long)cdiv(d.Wo, 16) * 16 * cdiv(d.Ho, 8) * 8;
    if (tall) q.tiles_x = cdiv(d.Wo, 4), q.tiles_y = cdiv(d.Ho, 32);
    q.dy0 = dymin, q.dx0 = dxmin;
    auto al16 = [](const void *ptr) { return ptr == nullptr || ((uintptr_t)ptr & 15) == 0; };
    q.vec4 = d.Cout % 4 == 0 && d.ldo % 4 == 0 && al16(d.out) && al16(d.bias) && (!d.o1 || (d.ldo1 % 4 == 0 && al16(d.o1))) &&
             (!d.e0 || (d.lde0 % 4 == 0 && al16(d.e0))) && (!d.e1 || (d.lde1 % 4 == 0 && al16(d.e1))) &&
             (!d.o2 || (d.ldo2 % 4 == 0 && al16(d.o2)));
    if (d.epi == RAMNET_EPI_LSTM) RAMNET_CHECK_ARG(q.vec4);      // the cell epilogue works on channel quads of the staged tile
    if (d.epi == RAMNET_EPI_GRU_BWD) RAMNET_CHECK_ARG(q.vec4 && d.Cout % 128 == 0);      // a 64-channel block lies in one half of [dx | d(h.r)]
    q.s2d_shift = 0;
    q.sparse = 0;
    if (d.s2d_5x5) {
        RAMNET_CHECK_ARG(d.in_mode == RAMNET_IN_S2D || d.out_s2d);
        q.sparse = d.in_mode == RAMNET_IN_S2D ? 1 : (d.out_s2d >= 32 ? 2 : 0);      // (a 32-channel block must lie in one parity group)
    }
    if (d.out_s2d) {
        RAMNET_CHECK_ARG(d.out_s2d >= 8 && log2_exact(d.out_s2d) > 0 && d.Cout == 4 * d.out_s2d && q.vec4 && d.epi == RAMNET_EPI_LINEAR &&
                         !d.bias && d.beta == 0.f && d.HoF == 2 * d.Ho && d.WoF == 2 * d.Wo);
        q.s2d_shift = log2_exact(d.out_s2d);
    }
    // XCD-pinned channel groups for weights that do not fit an L2: 2 groups above 3 MB, 4 above 12 MB (when nblk divides)
    const size_t wbytes = (size_t)q.nchunks * q.nblk * WU_FLOATS * sizeof(float);
    q.xg = wbytes > (12u << 20) ? 2 : wbytes > (3u << 20) ? 1 : 0;
    while (q.xg > 0 && (q.nblk % (1 << q.xg)) != 0) --q.xg;
    const int lanes = 8 >> q.xg;
    // Less than one full round of 64-channel workgroups (2 per CU: 512 slots) -> 32-channel workgroups, twice as many: at batch 1
    // (44-176 workgroups) the launch is bound by the latency of a workgroup's chunk chain, which halves; at the training batch the
    // 32 x 43 maps give 352 workgroups (residual blocks, ConvGRU candidate, encoder 2: 42-46 % MFMA-busy on a chip whose CUs hold one or
    // two of them), and 704 half-size workgroups — three fit a CU: 158 VGPRs, 37 KB of LDS — spread evenly over it.
    const bool nf1_mode = d.in_mode == RAMNET_IN_PLAIN || d.in_mode == RAMNET_IN_CAT || d.in_mode == RAMNET_IN_CAT_MUL ||
                          d.in_mode == RAMNET_IN_S2D || d.in_mode == RAMNET_IN_RELUMASK;
    const int wgs2 = q.tiles_x * q.tiles_y * d.B * q.nblk;
    // (same-box A/B of the threshold 512 against 256 at B = 8: residual conv 0.093 -> 0.081 ms forward, 0.094 -> 0.084 backward-data,
    // ConvGRU candidate at 32 x 43 0.177 -> 0.163 ms, training step 202.2 -> 203.4 / 204.0 samples/s)
    // threshold sweep at B = 8 (same box): 512 -> 204.6 / 204.5 samples/s, 1024 -> 203.5 (ConvGRU gates at 32 x 43: 0.258 -> 0.276 ms),
    // 1536 -> 201.6, everything -> 198.7: beyond the partial first round the doubled transform work costs more than the fill buys
    nf = (wgs2 < 512 && d.epi != RAMNET_EPI_LSTM && q.sparse != 2 && nf1_mode && d.Cout % 64 == 0) ? 1 : 2;
    // a launch of fewer workgroups than CUs is latency-bound: skipping the MFMAs of the zero slices buys nothing there, the
    // wave-uniform tests around them cost (enc2 at batch 1: 1.7 instead of 0.9 us per chunk) -> dense
    if (nf == 1 && q.sparse == 1 && wgs2 < 256) q.sparse = 0;
    gridx = cdiv(q.tiles_x * q.tiles_y * d.B, lanes) * 8 * ((q.nblk * (2 / nf)) >> q.xg);
    // Split reduction: a launch of 32-channel workgroups that does not fill the chip's 512 workgroup slots (batch-1 streaming on the two
    // coarse scales: 88-352 workgroups of 32-64 chunks each) is bound by the length of ONE workgroup's chunk chain (0.65 us per chunk) —
    // split it 2-4 ways towards ~704 workgroups, not below 8 chunks per split.  Only the channel-quad epilogue joins partials.
    // Same-box sweep (tools/bench_split.py, us per launch unsplit / 2 / 4 splits): residual conv 32x43 256->256 29.6 / - / 19.8, ConvGRU
    // candidate 32x43 512->256 48.1 / 29.7 / 28.8, gates 512->512 51.7 / 49.2 / 41.0, gates 64x86 256->256 (352 workgroups) 47.3 / 39.9 /
    // 42.4, encoder-2 view 52.0 / 30.2 / 29.3; 16-chunk launches gain nothing (the join costs about what 5 chunks do).
    ksplit = 1;
    if (nf == 1 && q.vec4 && !q.s2d_shift && g_opt_wino_ksplit) {
        const int wgs1 = 2 * wgs2;
        ksplit = 704 / wgs1 < 4 ? 704 / wgs1 : 4;
        if (q.nchunks / 8 < ksplit) ksplit = q.nchunks / 8;
        if (q.nchunks < 32) ksplit = 1;                 // the join costs about what 5 chunks do
        if (g_opt_wino_ksplit > 1) ksplit = g_opt_wino_ksplit < q.nchunks ? g_opt_wino_ksplit : q.nchunks;      // (forced: tuning runs)
        if (ksplit < 1) ksplit = 1;
        while (ksplit > 1 && (ksplit - 1) * cdiv(q.nchunks, ksplit) >= q.nchunks) --ksplit;      // (every split owns at least one chunk)
    }
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
    return 0;
}

// workspace of a split launch: [gridx counters, padded to 64][gridx][ksplit][128 pixels x 32 channels]
static size_t wino_ksplit_floats(unsigned gridx, int ksplit) { return (size_t)cdiv((int)gridx, 64) * 64 + (size_t)gridx * ksplit * 4096; }

int launch_wino(const ramnet_conv_desc &d, hipStream_t st) {
    WinoParams q;
    bool tall;
    int nf, ksplit;
    unsigned gridx;
    if (int rc = wino_plan(d, q, tall, nf, ksplit, gridx)) return rc;
    q.ksplit = d.splitk_ws ? ksplit : 1;
    q.cnt = reinterpret_cast<int *>(d.splitk_ws);
    q.ws = d.splitk_ws ? d.splitk_ws + cdiv((int)gridx, 64) * 64 : nullptr;
    if (q.ksplit > 1) RAMNET_CHECK_ARG(d.splitk_floats >= wino_ksplit_floats(gridx, ksplit) && ((uintptr_t)d.splitk_ws & 15) == 0);
    dim3 grid(gridx, q.ksplit);
    const size_t lds = (size_t)(nf == 1 ? 4 * 2 * 32 * (32 + 4) : RO_FLOATS) * sizeof(float);       // (the two patch buffers, 2 x 2 planes, and the scratch cells are smaller)
    // one instantiation per (tile shape, sparse mode, input mode) that occurs: the kernel body has no run-time mode branches
    const int key = (tall ? 0 : 1000) + q.sparse * 100 + d.in_mode + (nf == 1 ? 10000 : 0);
    note_kernel("conv_wino_r_kernel<%d,%d,%d,%d>", tall ? 2 : 8, q.sparse, d.in_mode, nf);
#define RAMNET_GO(TXv, SPv, MDv)                                                                        \
    case ((TXv) == 2 ? 0 : 1000) + (SPv) * 100 + (MDv):                                                 \
        RAMNET_FULL_LDS((conv_wino_r_kernel<TXv, SPv, MDv>));                                           \
        hipLaunchKernelGGL((conv_wino_r_kernel<TXv, SPv, MDv>), grid, dim3(256), lds, st, d, q);        \
        break;
#define RAMNET_GO1(TXv, SPv, MDv)                                                                       \
    case 10000 + ((TXv) == 2 ? 0 : 1000) + (SPv) * 100 + (MDv):                                         \
        RAMNET_FULL_LDS((conv_wino_r_kernel<TXv, SPv, MDv, 1>));                                        \
        hipLaunchKernelGGL((conv_wino_r_kernel<TXv, SPv, MDv, 1>), grid, dim3(256), lds, st, d, q);     \
        break;
#define RAMNET_GO_TX(TXv)                                                                               \
    RAMNET_GO(TXv, 0, RAMNET_IN_PLAIN) RAMNET_GO(TXv, 0, RAMNET_IN_CAT) RAMNET_GO(TXv, 0, RAMNET_IN_CAT_MUL)        \
    RAMNET_GO(TXv, 0, RAMNET_IN_RELUMASK) RAMNET_GO(TXv, 0, RAMNET_IN_S2D) RAMNET_GO(TXv, 1, RAMNET_IN_S2D)         \
    RAMNET_GO(TXv, 2, RAMNET_IN_PLAIN) RAMNET_GO(TXv, 2, RAMNET_IN_RELUMASK)                                        \
    RAMNET_GO1(TXv, 0, RAMNET_IN_PLAIN) RAMNET_GO1(TXv, 0, RAMNET_IN_CAT) RAMNET_GO1(TXv, 0, RAMNET_IN_CAT_MUL)     \
    RAMNET_GO1(TXv, 0, RAMNET_IN_S2D) RAMNET_GO1(TXv, 1, RAMNET_IN_S2D) RAMNET_GO1(TXv, 0, RAMNET_IN_RELUMASK)
    switch (key) {
        RAMNET_GO_TX(2)
        RAMNET_GO_TX(8)
    default:
        RAMNET_CHECK_ARG(!"conv_wino_r: unsupported (sparse, input mode) combination");
    }
#undef RAMNET_GO_TX
#undef RAMNET_GO1
#undef RAMNET_GO
    RAMNET_LAUNCH_CHECK();
    return 0;
}

}  // namespace ramnet

using namespace ramnet;

extern "C" size_t ramnet_conv_splitk_floats(const ramnet_conv_desc *d) {
    if (d && d->algo == RAMNET_ALGO_WINOGRAD24) return wino24_splitk_floats(*d);
    if (!d || d->algo != RAMNET_ALGO_WINOGRAD) return 0;
    WinoParams q;
    bool tall;
    int nf, ksplit;
    unsigned gridx;
    if (wino_plan(*d, q, tall, nf, ksplit, gridx) != 0 || ksplit <= 1) return 0;
    return wino_ksplit_floats(gridx, ksplit);
}

extern "C" size_t ramnet_packed_weight_elems_wino(int Cout, int Cin, int transposed, int gates) {
    int R, N, nchunks, nblk;
    wino_geometry(Cout, Cin, transposed, gates, R, N, nchunks, nblk);
    return (size_t)nchunks * nblk * WU_FLOATS;
}

extern "C" int ramnet_pack_weight_wino(const float *w, float *wp, int Cout, int Cin, int transposed, int gates, void *stream) {
    RAMNET_CHECK_ARG(w && wp && Cout > 0 && Cin > 0);
    RAMNET_CHECK_ARG(gates == 1 || (gates == 4 && !transposed && Cout % 4 == 0));
    int R, N, nchunks, nblk;
    wino_geometry(Cout, Cin, transposed, gates, R, N, nchunks, nblk);
    const size_t total = (size_t)nchunks * nblk * WU_FLOATS;
    size_t blocks = (total + 255) / 256;
    if (blocks > 65535) blocks = 65535;
    hipLaunchKernelGGL(pack_weight_wino_r_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, wp, Cout, Cin,
                       transposed, gates, R, N, nchunks, nblk, total);
    RAMNET_LAUNCH_CHECK();
    return 0;
}
