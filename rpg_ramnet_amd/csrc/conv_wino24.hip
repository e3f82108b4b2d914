// Winograd F(2x2, 4x4) for the folded upsample-conv of the decoders (statenet.py:305-308, submodules.py:69-97) on gfx950.
//
// conv5x5(bilinear_x2(x + skip)) is, per output parity (py, px), a 4x4 stride-1 convolution of the replicate-padded low-res
// sum.  Each of those four convolutions is evaluated here as
//     Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A,     5x5 input tile -> 2x2 outputs, 25 instead of 64 multiplies
// (Toom-Cook points 0, 1, -1, 2, inf), i.e. 6.25 multiplies per output and input channel instead of 16 (25 for the plain 5x5).
//
// Workgroup = 8 waves: 32 tiles (8 x 4 tiles = 16 x 8 pixels of one parity grid) x 64 output channels (NCQ = 4; for 32-channel
// layers NCQ = 2: 64 tiles x 32 channels, chunks of 8 input channels, 8-byte operand reads).  Per chunk of 16
// input channels every thread loads the 5x5 window of ONE (tile, channel) straight from global memory (the padded input needs
// no bounds logic), transforms it in registers and writes the 25 values to V[25][32 tiles][16] in LDS (double-buffered, one
// barrier per chunk).  Wave (tile half, channel quarter) accumulates all 25 positions of its 16 tiles x 16 channels on
// v_mfma_f32_16x16x4_f32 (100 accumulator VGPRs): A = one 16-byte LDS read per position (swizzled, conflict-free), B = one
// 16-byte global (L2) read per position from weights packed in exactly that lane order.  Loads and the transform of chunk
// i+1 are issued between the MFMAs of chunk i.  The output transform is register-local; bias / ReLU and the border
// corrections of the folded layer (ramnet_conv_desc.frame) are applied by the shared epilogue.
#include <stdlib.h>
#include "common.hpp"
#include "conv_epilogue.hpp"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace ramnet {

// tiles per workgroup (2x2 outputs each): 8 x 4 or 4 x 8 (rows x columns, TXW = columns) for 32 tiles, 8 x 8 for 64
constexpr int W24_PS = 512;                       // floats per position in V: tiles x chunk channels (32 x 16 or 64 x 8)
constexpr int W24_V = 25 * W24_PS;                // floats per V buffer (51.2 KB)

__device__ __forceinline__ float2 ld2f(const float *p) { return *reinterpret_cast<const float2 *>(p); }

// Buffer resource over [p, p + bytes): loads then take a 32-bit per-lane byte offset plus a scalar one, so the window and
// weight streams of the main loop need no 64-bit address arithmetic.  The pointer goes through readfirstlane so that the
// compiler knows the descriptor is wave-uniform (cdna_hip_programming.md, buffer addressing).
__device__ __forceinline__ auto make_rsrc(const void *p, unsigned bytes) {
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), (short)0, (int)bytes, 0x00020000);
}

#ifdef W24_TRACE   // tools/wino24_trace.hip: per-wave timestamps (s_memtime, 100 MHz): [block][wave][16 chunks][6 slots] + 4 kernel slots
__device__ unsigned long long *g_w24_trace;
#define W24_STAMP(chunk_, slot)                                                                                          \
    do {                                                                                                                 \
        if (lane == 0 && (chunk_) <= 16) g_w24_trace[(((size_t)blockIdx.x * 8 + wave) * 17 + (chunk_)) * 6 + (slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define W24_STAMP(chunk_, slot)
#endif

struct Wino24Params {
    const float *x;          // replicate-padded low-res input [B][Hp][Wp][Cin]
    const float *wp;         // [4 classes][Cin/16][Cout/64][25][4][64 lanes][4]
    int Hp, Wp, ldx, nchunks, nblk, tiles_x, tiles_y, Hc, Wc;
    int cpc;                  // backward-data (DG): chunks per parity class = Cout_fwd / chunk size
    unsigned xbytes, wbytes;  // extents of x and wp (buffer addressing: both below 4 GB)
    int ksplit;               // forward: gridDim.y = splits of the channel reduction (1 = none; an even number of chunks each)
    float *ws;                // ksplit > 1: partial outputs [gridDim.x][ksplit][512 threads x 16]
    int *cnt;                 //             arrival counters [gridDim.x], zero between launches
};

// NCQ = 16-channel groups of output channels per workgroup: 4 (32 tiles, chunks of 16) or 2 (64 tiles, chunks of 8)
// DG = backward-data of the folded layer (RAMNET_IN_PARITY4): the input is the full-resolution gradient [B][2*Hc][2*Wc][C]
// (Hp, Wp = its extent), whose four parity sub-grids are the reduction blocks (class = chunk / cpc: window origin and pixel
// stride 2, zero outside — a buffer load past the tensor returns 0); the output is the dense (Hc+4) x (Wc+4) grid of the padded
// low-resolution tensor.
// PAIR (32-channel layers: the last decoder): the two column parities px = 0 / 1 of a row parity share ONE transformed input — the
// class-(py, 1) grid is tiled with its tile origins shifted by one column, so that both classes read the same 5 x 5 windows — and
// the workgroup's 64 columns are (px, 32 channels): the input transform (25 loads + ~160 VALU per tile and channel, the cost that
// bounds this kernel at 32 output channels: 35 % MFMA-busy) feeds twice the MFMAs.  NCQ = 4 geometry, chunks of 16.
template <int NCQ, bool DG, int TXW, bool PAIR = false>
__global__ void __launch_bounds__(512, 1) conv_wino24_kernel(const ramnet_conv_desc p, const Wino24Params q) {
    static_assert(!PAIR || (NCQ == 4 && !DG), "pair mode: forward, 64-column workgroups");
    constexpr int W24_K = NCQ == 4 ? 16 : 8;          // input channels per chunk
    constexpr int W24_TX = TXW, W24_TY = (NCQ == 4 ? 32 : 64) / TXW;      // tile columns / rows per workgroup
    constexpr int VEC = W24_K / 4;                    // floats per lane and operand read (k = VEC*ks + j)
    constexpr int W24_U = 25 * NCQ * 64 * VEC;        // packed weights of one (class, chunk, channel block)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *V = smem;                               // [2][25][32][16]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, ks = lane >> 4;
    const int th = NCQ == 4 ? wave & 1 : wave & 3, cq = NCQ == 4 ? wave >> 1 : wave >> 2;       // 16-tile group, 16-channel group

    // blockIdx.x = ((tile block * nblk + channel block) * 4 + class): the four parities of a tile block read the same input
    int bid = blockIdx.x;
    const int cls = DG ? 0 : PAIR ? bid & 1 : bid & 3;          // PAIR: the row parity; both column parities in this workgroup
    if (!DG) bid >>= PAIR ? 1 : 2;
    const int nb = bid % q.nblk;
    bid /= q.nblk;
    const int tbx = bid % q.tiles_x;
    bid /= q.tiles_x;
    const int tby = bid % q.tiles_y, b = bid / q.tiles_y;
    const int py = PAIR ? cls : cls >> 1, px = PAIR ? 0 : cls & 1;
    const int n0 = nb * 16 * NCQ;

    // ---- input transform item of this thread: (tile, channel of the chunk)
    const int it = tid / W24_K, ik = tid % W24_K;
    // window rows / columns past the padded input only feed outputs past the grid: clamp them one by one
    const int iy0 = 2 * (tby * W24_TY + it / W24_TX) + py, ix0 = 2 * (tbx * W24_TX + it % W24_TX) + px;
    // byte offsets: lane part (image + row, column + channel: one add per load) + scalar part (chunk)
    unsigned rowo[5], colo[5];
    int cur_cls = -1;
    auto set_class = [&](int c) {                   // DG: window of parity class c = (c >> 1, c & 1); invalid -> offset past the tensor
        cur_cls = c;
        const int cy = c >> 1, cx = c & 1;
        const int u0 = 2 * (tby * W24_TY + it / W24_TX), v0 = 2 * (tbx * W24_TX + it % W24_TX);
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            const int i = u0 - cy - 3 + r, j = v0 - cx - 3 + r;           // class-grid row / column
            rowo[r] = (unsigned)i < (unsigned)q.Hc ? 4u * (unsigned)((b * q.Hp + 2 * i + cy) * q.Wp * q.ldx) : 0x40000000u;
            colo[r] = (unsigned)j < (unsigned)q.Wc ? 4u * (unsigned)((2 * j + cx) * q.ldx + ik) : 0x40000000u;
        }
    };
    if (!DG) {
#pragma unroll
        for (int r = 0; r < 5; ++r)
            rowo[r] = 4u * (unsigned)((b * q.Hp + min(iy0 + r, q.Hp - 1)) * q.Wp * q.ldx), colo[r] = 4u * (unsigned)(min(ix0 + r, q.Wp - 1) * q.ldx + ik);
    }
    // Split reduction (forward launches far below one workgroup per CU: the first decoders at batch 1): workgroup blockIdx.y reduces
    // chunks [cbeg, cbeg + nch) and the partial outputs are joined in front of the epilogue (as in conv_wino.hip)
    const int ksp = DG ? 1 : q.ksplit;
    const int cps = ((q.nchunks / 2 + ksp - 1) / ksp) * 2, cbeg = DG ? 0 : (int)blockIdx.y * cps;
    const auto xrs = make_rsrc(q.x + cbeg * W24_K, q.xbytes - (unsigned)cbeg * W24_K * 4u);
    // operand reads are VEC floats per lane; the XOR swizzle spreads the 16 tiles of a read over all banks
    const int vdst = NCQ == 4 ? it * 16 + (((ik >> 2) ^ ((it >> 2) & 3)) << 2) + (ik & 3) : it * 8 + (((ik >> 1) ^ ((it >> 3) & 1)) << 1) + (ik & 1);
    float raw[25];
    auto load_raw = [&](int chunk) {
        int soff = chunk * (W24_K * 4);                 // uniform
        if (DG) {
            const int c = chunk / q.cpc;
            if (c != cur_cls) set_class(c);
            soff = (chunk - c * q.cpc) * (W24_K * 4);
        }
#pragma unroll
        for (int r = 0; r < 5; ++r)
#pragma unroll
            for (int c = 0; c < 5; ++c)
                raw[r * 5 + c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, (int)(rowo[r] + colo[c]), soff, 0));
    };
    // one window row at a time (the main loop requests row i as soon as tr_row(i) has consumed the previous window's row i: five loads per
    // slot instead of 25 in one)
    int lr_soff = 0;
    auto load_row_begin = [&](int chunk) {
        lr_soff = chunk * (W24_K * 4);
        if (DG) {
            const int c = chunk / q.cpc;
            if (c != cur_cls) set_class(c);
            lr_soff = (chunk - c * q.cpc) * (W24_K * 4);
        }
    };
    auto load_row = [&](int r) {
#pragma unroll
        for (int c = 0; c < 5; ++c)
            raw[r * 5 + c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, (int)(rowo[r] + colo[c]), lr_soff, 0));
    };
    // B^T = [2 -1 -2 1 0; 0 -2 -1 1 0; 0 2 -3 1 0; 0 -1 0 1 0; 0 2 -1 -2 1]
    auto tr_col = [&](int c) {
        const float d0 = raw[c], d1 = raw[5 + c], d2 = raw[10 + c], d3 = raw[15 + c], d4 = raw[20 + c];
        raw[c] = 2.f * (d0 - d2) - d1 + d3;
        raw[5 + c] = d3 - d2 - 2.f * d1;
        raw[10 + c] = 2.f * d1 - 3.f * d2 + d3;
        raw[15 + c] = d3 - d1;
        raw[20 + c] = 2.f * (d1 - d3) - d2 + d4;
    };
    auto tr_row = [&](float *vbuf, int i) {
        const float d0 = raw[i * 5], d1 = raw[i * 5 + 1], d2 = raw[i * 5 + 2], d3 = raw[i * 5 + 3], d4 = raw[i * 5 + 4];
        float *dst = vbuf + (i * 5) * W24_PS + vdst;
        dst[0 * W24_PS] = 2.f * (d0 - d2) - d1 + d3;
        dst[1 * W24_PS] = d3 - d2 - 2.f * d1;
        dst[2 * W24_PS] = 2.f * d1 - 3.f * d2 + d3;
        dst[3 * W24_PS] = d3 - d1;
        dst[4 * W24_PS] = 2.f * (d1 - d3) - d2 + d4;
    };

    // ---- MFMA operands
    const int tile = th * 16 + l15;
    const int aoff = NCQ == 4 ? tile * 16 + ((ks ^ ((tile >> 2) & 3)) << 2) : tile * 8 + ((ks ^ ((tile >> 3) & 1)) << 1);
    constexpr int WPOS = NCQ * 64 * VEC;              // packed floats per position
    const auto wrs = make_rsrc(q.wp, q.wbytes);
    const int wsrc = (((cls * q.nchunks + cbeg) * q.nblk + nb) * W24_U + cq * (64 * VEC)) * 4;     // uniform byte offset; lane part below
    const int wlane = lane * VEC * 4;
    auto ldv = [&](const float *ptr) {                // VEC floats -> float4 (upper half unused for VEC = 2)
        if (VEC == 4) return ld4(ptr);
        const float2 t = ld2f(ptr);
        return make_float4(t.x, t.y, 0.f, 0.f);
    };
    auto ldw = [&](int soff) {                       // weights: scalar byte offset + lane offset
        if constexpr (VEC == 4) {
            return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wrs, wlane, soff, 0));
        } else {
            const float2 t = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(wrs, wlane, soff, 0));
            return make_float4(t.x, t.y, 0.f, 0.f);
        }
    };
    const int wchunk = q.nblk * W24_U * 4;

    f32x4 acc[25];
#pragma unroll
    for (int i = 0; i < 25; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nch = DG ? q.nchunks : min(cps, q.nchunks - cbeg);
    W24_STAMP(16, 0);
    load_raw(0);
    // Weight ring: 10 slots, prefetch distance 9 positions.  Vector loads return in order, so a weight load issued after the
    // 25 window loads of a chunk can only be consumed once those have landed: the ring is deep enough to give them ~9 positions
    // (~1 us) of MFMA work, and the chunk loop is unrolled by two so that the slot of a position is a compile-time constant.
    constexpr int RING = 10, DIST = 8;
    float4 bq[RING];
#pragma unroll
    for (int i = 0; i < DIST; ++i) bq[i] = ldw(wsrc + i * (WPOS * 4));
#pragma unroll
    for (int c = 0; c < 5; ++c) tr_col(c);
#pragma unroll
    for (int i = 0; i < 5; ++i) tr_row(V, i);
    load_raw(min(1, nch - 1));
    __syncthreads();

    W24_STAMP(16, 1);
    for (int chunk0 = 0; chunk0 < nch; chunk0 += 2) {             // nch is even (checked on the host)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int chunk = chunk0 + half;
            const float *vb = V + half * W24_V;
            float *vn = V + (half ^ 1) * W24_V;
            // The window of chunk+1 is already in registers (loaded at the end of the previous phase: a weight load issued
            // after it can only be consumed once it has landed, and there it has a barrier and four position pairs to do so);
            // it is transformed into the other V buffer during this phase and the window of chunk+2 requested right after.
            const int cnext = min(chunk + 1, nch - 1), cnext2 = min(chunk + 2, nch - 1);     // past the end: re-stage the last chunk
            const int wcur = wsrc + chunk * wchunk, wnext = wsrc + cnext * wchunk;
            W24_STAMP(chunk, 0);
            // positions go in pairs (two independent accumulator chains: a 16x16x4 MFMA can be issued every 32 cycles but its
            // result is only available to a dependent one after 40); the last position runs alone
            float4 aq[2][2];                                   // A operands of the current / next pair (ping-pong, no copies)
            aq[0][0] = ldv(vb + aoff), aq[0][1] = ldv(vb + W24_PS + aoff);
#pragma unroll
            for (int pp = 0; pp < 13; ++pp) {
                const int pos = 2 * pp, g = half * 25 + pos;
                const bool two = pos + 1 < 25;
                if (pos + 2 < 25) aq[(pp + 1) & 1][0] = ldv(vb + (pos + 2) * W24_PS + aoff);
                if (pos + 3 < 25) aq[(pp + 1) & 1][1] = ldv(vb + (pos + 3) * W24_PS + aoff);
                const float4 a0 = aq[pp & 1][0], a1 = aq[pp & 1][1];
#if !defined(W24_ABLATE) || !(W24_ABLATE & 2)
                bq[(g + DIST) % RING] = pos + DIST < 25 ? ldw(wcur + (pos + DIST) * (WPOS * 4)) : ldw(wnext + (pos + DIST - 25) * (WPOS * 4));
                if (two) bq[(g + 1 + DIST) % RING] = pos + 1 + DIST < 25 ? ldw(wcur + (pos + 1 + DIST) * (WPOS * 4)) : ldw(wnext + (pos + 1 + DIST - 25) * (WPOS * 4));
#endif
                const float4 b0 = bq[g % RING], b1 = bq[(g + 1) % RING];
                __builtin_amdgcn_sched_barrier(0);
                acc[pos] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0.x, acc[pos], 0, 0, 0);
                if (two) acc[pos + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b1.x, acc[pos + 1], 0, 0, 0);
                acc[pos] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b0.y, acc[pos], 0, 0, 0);
                if (two) acc[pos + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b1.y, acc[pos + 1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                // The side work of the chunk in SLOTS (slot = 2 pp + side), one piece behind every group of four MFMAs instead of blocks of
                // 24-34 VALU in three gaps and 25 loads in one: column transform c in slots 4..8, row transform i (+ its 5 LDS stores) in
                // slots 9, 11, .., 17, and the five window loads of row i of chunk + 2 right behind it (slots 10, 12, .., 18: the row's
                // registers are free, and the loads have six position pairs + the barrier to land before the next chunk's slot 4).
                auto slot = [&](int sl) {
                    if (sl == 4) W24_STAMP(chunk, 1);
                    if (sl >= 4 && sl <= 8) tr_col(sl - 4);
                    if (sl == 9) W24_STAMP(chunk, 2);
                    if (sl >= 9 && sl <= 17 && (sl & 1)) tr_row(vn, (sl - 9) >> 1);
#if !defined(W24_ABLATE) || !(W24_ABLATE & 1)      // timing experiments (tools/wino24_trace.hip): 1 = no window loads, 2 = no weight loads
                    if (sl == 10) load_row_begin(cnext2);
                    if (sl >= 10 && sl <= 18 && !(sl & 1)) load_row((sl - 10) >> 1);
#endif
                    if (sl == 18) W24_STAMP(chunk, 3);
                };
                slot(2 * pp);
                __builtin_amdgcn_sched_barrier(0);
                if (VEC == 4) {
                    acc[pos] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b0.z, acc[pos], 0, 0, 0);
                    if (two) acc[pos + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b1.z, acc[pos + 1], 0, 0, 0);
                    acc[pos] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b0.w, acc[pos], 0, 0, 0);
                    if (two) acc[pos + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b1.w, acc[pos + 1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                slot(2 * pp + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            W24_STAMP(chunk, 4);
            __syncthreads();
            W24_STAMP(chunk, 5);
        }
    }
    W24_STAMP(16, 2);

    // ---- output transform A^T M A (A^T = [1 1 1 1 0; 0 1 -1 2 1]) of the lane's 4 tiles x 1 channel, fused epilogue.
    // All 16 outputs of the lane are finished in registers (bias — ONE load —, border corrections, ReLU) and stored at the end: with the
    // stores interleaved, every border-correction block (a conditional load) was followed by a wait for ALL memory operations, i.e. for
    // the previous output's store (stores count in vmcnt on gfx9): 16 serialized write latencies per workgroup, which holds its CU alone.
    const int ncol = n0 + cq * 16 + l15;
    const int pxc = PAIR ? ncol >> 5 : px, n = PAIR ? ncol & 31 : ncol;      // PAIR: column = (column parity, channel)
    if (n >= p.Cout) return;
    const int epi = p.epi;
    const float bias_n = (!DG && p.bias) ? p.bias[n] : 0.f;
    const bool relu = epi == RAMNET_EPI_RELU;
    float ov[16];
    size_t op[16];
    bool oo[16];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float s[2][5];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const float m0 = acc[j][r], m1 = acc[5 + j][r], m2 = acc[10 + j][r], m3 = acc[15 + j][r], m4 = acc[20 + j][r];
            s[0][j] = m0 + m1 + m2 + m3;
            s[1][j] = m1 - m2 + 2.f * m3 + m4;
        }
#pragma unroll
        for (int a2 = 0; a2 < 2; ++a2) {
            ov[(r * 2 + a2) * 2] = s[a2][0] + s[a2][1] + s[a2][2] + s[a2][3];
            ov[(r * 2 + a2) * 2 + 1] = s[a2][1] - s[a2][2] + 2.f * s[a2][3] + s[a2][4];
        }
    }
    if (!DG && ksp > 1) {
        // join of the split reduction: partial outputs -> slab blockIdx.y of this tile's workspace, the last arrival adds the slabs in
        // split order (device-scope 16-byte accesses, no fences: see conv_wino.hip) and goes on; no lane has left (Cout fills the block)
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        constexpr int AUX_SC1 = 16;
        const auto wr = make_rsrc(q.ws + (size_t)blockIdx.x * ksp * 8192, (unsigned)(ksp * 8192 * 4));
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, make_float4(ov[4 * i], ov[4 * i + 1], ov[4 * i + 2], ov[4 * i + 3])), wr,
                                                   (i * 512 + tid) * 16, (int)blockIdx.y * 32768, AUX_SC1);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        if (tid == 0) {
            const int arrived = __hip_atomic_fetch_add(q.cnt + blockIdx.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (arrived == ksp - 1) __hip_atomic_store(q.cnt + blockIdx.x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            reinterpret_cast<int *>(smem)[0] = arrived == ksp - 1;
        }
        __syncthreads();
        if (!reinterpret_cast<int *>(smem)[0]) return;
#pragma unroll
        for (int i = 0; i < 16; ++i) ov[i] = 0.f;
        for (int sp = 0; sp < ksp; ++sp)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 t = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wr, (i * 512 + tid) * 16, sp * 32768, AUX_SC1));
                ov[4 * i] += t.x, ov[4 * i + 1] += t.y, ov[4 * i + 2] += t.z, ov[4 * i + 3] += t.w;
            }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int t = th * 16 + 4 * ks + r;
        const int oy0 = 2 * (tby * W24_TY + t / W24_TX), ox0 = 2 * (tbx * W24_TX + t % W24_TX) - (PAIR ? pxc : 0);
#pragma unroll
        for (int a2 = 0; a2 < 2; ++a2) {
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) {
                const int oy = oy0 + a2, ox = ox0 + c2, idx = (r * 2 + a2) * 2 + c2;
                if (DG) {                           // dense output grid, plain store
                    oo[idx] = oy < p.Ho && ox < p.Wo;
                    op[idx] = (((size_t)b * p.Ho + oy) * p.Wo + ox) * p.ldo + n;
                    continue;
                }
                oo[idx] = !(oy >= q.Hc || ox >= q.Wc || (PAIR && ox < 0));
                const int oyF = 2 * oy + py, oxF = 2 * ox + pxc;
                op[idx] = (((size_t)b * p.HoF + oyF) * p.WoF + oxF) * p.ldo + n;
                float v = ov[idx] + bias_n;
                if (oo[idx]) v += epilogue_side(p, epi, b, oyF, oxF, n);
                ov[idx] = relu ? fmaxf(v, 0.f) : v;
            }
        }
    }
#pragma unroll
    for (int idx = 0; idx < 16; ++idx)
        if (oo[idx]) p.out[op[idx]] = ov[idx];
    W24_STAMP(16, 3);
}

// OIHW 5x5 weights of an UpsampleConvLayer -> U = G W4 G^T of the four 4x4 parity filters W4 = A_py w A_px^T (the bilinear x2
// upsample folded into the filter) in the lane order of the kernel's B operand; evaluated in double.
__global__ void pack_weight_fold_wino_kernel(const float *__restrict__ w, float *__restrict__ wp, int Cout, int Cin, int kc, int ncq,
                                             int pair, size_t total) {
    const double FA[2][4][5] = {{{.25, 0, 0, 0, 0}, {.75, .75, .25, 0, 0}, {0, .25, .75, .75, .25}, {0, 0, 0, .25, .75}},
                                {{.75, .25, 0, 0, 0}, {.25, .75, .75, .25, 0}, {0, 0, .25, .75, .75}, {0, 0, 0, 0, .25}}};
    const double G[5][4] = {{0.5, 0, 0, 0}, {-0.5, -0.5, -0.5, -0.5}, {-1.0 / 6, 1.0 / 6, -1.0 / 6, 1.0 / 6}, {1.0 / 6, 1.0 / 3, 2.0 / 3, 4.0 / 3}, {0, 0, 0, 1}};
    const int vec = kc / 4, nblk = pair ? 1 : Cout / (16 * ncq), nch = Cin / kc;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        // i = ((((cls*nch + chunk)*nblk + nb)*25 + pos)*ncq + cq)*64*vec + (ks*16 + l15)*vec + j
        size_t r = i;
        const int j = (int)(r % vec);
        r /= vec;
        const int l15 = (int)(r % 16);
        r /= 16;
        const int ks = (int)(r % 4);
        r /= 4;
        const int cq = (int)(r % ncq);
        r /= ncq;
        const int pos = (int)(r % 25);
        r /= 25;
        const int nb = (int)(r % nblk);
        r /= nblk;
        const int chunk = (int)(r % nch), cls = (int)(r / nch);
        const int k = chunk * kc + ks * vec + j, col = (nb * ncq + cq) * 16 + l15;
        // pair layout (32-channel layers): class = row parity, column = (column parity, channel)
        const int n = pair ? col & 31 : col;
        const int py = pair ? cls : cls >> 1, px = pair ? col >> 5 : cls & 1, a = pos / 5, b = pos % 5;
        // U[a][b] = sum_{t,s} G[a][t] G[b][s] sum_{kh,kw} FA[py][t][kh] FA[px][s][kw] w[n][k][kh][kw]
        double ga[5], gb[5];                        // rows of G^T... combined with the fold: ga[kh] = sum_t G[a][t] FA[py][t][kh]
        for (int kh = 0; kh < 5; ++kh) {
            ga[kh] = 0, gb[kh] = 0;
            for (int t = 0; t < 4; ++t) ga[kh] += G[a][t] * FA[py][t][kh], gb[kh] += G[b][t] * FA[px][t][kh];
        }
        double u = 0;
        const float *wk = w + ((size_t)n * Cin + k) * 25;
        for (int kh = 0; kh < 5; ++kh)
            for (int kw = 0; kw < 5; ++kw) u += ga[kh] * gb[kw] * (double)wk[kh * 5 + kw];
        wp[i] = (float)u;
    }
}

static bool fold_wino_pair(int Cout, int Cin) {
    return Cout == 32 && Cin % 32 == 0 && g_opt_fold_pair;      // (ramnet_set_option("fold_pair", 0): the 32-channel form, 64 tiles x 32 channels, chunks of 8)
}

static bool fold_wino_geometry(int Cout, int Cin, int &kc, int &ncq) {
    kc = ((Cout % 64 == 0 && Cin % 16 == 0) || fold_wino_pair(Cout, Cin)) ? 16 : 8, ncq = kc == 16 ? 4 : 2;
    return Cout % 32 == 0 && Cin % (2 * kc) == 0;
}

// Backward-data of the folded layer: x0 = g = dy * mask [B][Hin = 2H][Win = 2W][C0 = Cout_fwd], out = gradient of the padded
// low-res tensor [B][Ho = H+4][Wo = W+4][Cout = Cin_fwd]; w = Winograd weights of the flipped parity filters over 4*C0 reduction
// channels (ops.pack_fold_wino_dgrad).
static int launch_wino24_dgrad(const ramnet_conv_desc &d, hipStream_t st) {
    RAMNET_CHECK_ARG(d.stride == 1 && d.C0 % 16 == 0 && d.Cout % 64 == 0);
    RAMNET_CHECK_ARG(d.Hin % 2 == 0 && d.Win % 2 == 0 && d.Ho == d.Hin / 2 + 4 && d.Wo == d.Win / 2 + 4 && d.HoF == d.Ho && d.WoF == d.Wo);
    RAMNET_CHECK_ARG(d.epi == RAMNET_EPI_LINEAR && !d.bias && d.beta == 0.f && d.frame == 0 && d.out_s2d == 0);
    Wino24Params q;
    q.x = d.x0, q.wp = d.w, q.Hp = d.Hin, q.Wp = d.Win, q.ldx = d.ld0;
    q.cpc = d.C0 / 16, q.nchunks = 4 * q.cpc, q.nblk = d.Cout / 64;
    q.ksplit = 1, q.ws = nullptr, q.cnt = nullptr;
    q.Hc = d.Hin / 2, q.Wc = d.Win / 2;
    const size_t xb = (size_t)d.B * d.Hin * d.Win * d.ld0 * sizeof(float), wb = (size_t)100 * d.C0 * d.Cout * sizeof(float);
    RAMNET_CHECK_ARG(xb < 0x40000000ull && wb < 0x7fffffffull);          // invalid window elements use offsets >= 2^30
    q.xbytes = (unsigned)xb, q.wbytes = (unsigned)wb;
    // 16 x 8 or 8 x 16 output pixels per workgroup: whichever covers the grid with fewer workgroups
    const bool flat = cdiv(d.Wo, 16) * cdiv(d.Ho, 8) < cdiv(d.Wo, 8) * cdiv(d.Ho, 16);
    q.tiles_x = cdiv(d.Wo, flat ? 16 : 8), q.tiles_y = cdiv(d.Ho, flat ? 8 : 16);
    const size_t lds = (size_t)2 * W24_V * sizeof(float);
    const dim3 grid((unsigned)(q.tiles_x * q.tiles_y * d.B * q.nblk));
    if (flat) {
        RAMNET_FULL_LDS((conv_wino24_kernel<4, true, 8>));
        note_kernel("conv_wino24_kernel<4,1,8,0>");
        hipLaunchKernelGGL((conv_wino24_kernel<4, true, 8>), grid, dim3(512), lds, st, d, q);
    } else {
        RAMNET_FULL_LDS((conv_wino24_kernel<4, true, 4>));
        note_kernel("conv_wino24_kernel<4,1,4,0>");
        hipLaunchKernelGGL((conv_wino24_kernel<4, true, 4>), grid, dim3(512), lds, st, d, q);
    }
    RAMNET_LAUNCH_CHECK();
    return 0;
}

// Split of the channel reduction a forward launch wants (1 = none): workgroups of 512 threads, one per CU — a launch that leaves more than
// half of the CUs empty (decoder 0 at batch 1: 96 workgroups of 16 chunks) splits towards one full round of 256 workgroups, at least four
// chunks (an even number) per split; every lane of a workgroup must own an output channel (the join has barriers).  Measured
// (tools/bench_split.py, whole layer): decoder 0 87.0 -> 65.8 us with 2 splits (4: 67.5); decoder 1 (176 workgroups) 63.9 -> 64.8 with 2: a
// second round of workgroups costs what the shorter chains save — not split.
static int wino24_ksplit(const ramnet_conv_desc &d, int gridx, int nchunks) {
    if (!g_opt_wino_ksplit || d.in_mode != RAMNET_IN_PLAIN) return 1;
    const bool pair = fold_wino_pair(d.Cout, d.C0), wide = pair || (d.Cout % 64 == 0 && d.C0 % 16 == 0);
    if (!pair && d.Cout % (wide ? 64 : 32) != 0) return 1;
    int ks = 256 / gridx < 4 ? 256 / gridx : 4;
    if (nchunks / 4 < ks) ks = nchunks / 4;
    if (g_opt_wino_ksplit > 1) ks = g_opt_wino_ksplit < nchunks / 2 ? g_opt_wino_ksplit : nchunks / 2;
    if (ks < 1) ks = 1;
    while (ks > 1 && (ks - 1) * (cdiv(nchunks / 2, ks) * 2) >= nchunks) --ks;        // every split owns chunks
    return ks;
}
static size_t wino24_ksplit_floats(unsigned gridx, int ks) { return (size_t)cdiv((int)gridx, 64) * 64 + (size_t)gridx * ks * 8192; }

// ramnet_conv_splitk_floats() for RAMNET_ALGO_WINOGRAD24 descriptors (same geometry as launch_wino24 below)
size_t wino24_splitk_floats(const ramnet_conv_desc &d) {
    if (d.in_mode != RAMNET_IN_PLAIN || d.C0 % 16 != 0 || d.Cout % 32 != 0) return 0;
    const bool pair = fold_wino_pair(d.Cout, d.C0), wide = pair || (d.Cout % 64 == 0 && d.C0 % 16 == 0);
    if (d.C0 % (wide ? 32 : 16) != 0) return 0;
    const int nchunks = d.C0 / (wide ? 16 : 8), nblk = pair ? 1 : d.Cout / (wide ? 64 : 32);
    const int Wt = pair ? d.Wo + 2 : d.Wo;
    const bool flat = wide && cdiv(Wt, 16) * cdiv(d.Ho, 8) < cdiv(Wt, 8) * cdiv(d.Ho, 16);
    const int tiles_x = cdiv(Wt, wide && !flat ? 8 : 16), tiles_y = cdiv(d.Ho, flat ? 8 : 16);
    const unsigned gridx = (unsigned)(tiles_x * tiles_y * d.B * nblk * (pair ? 2 : 4));
    const int ks = wino24_ksplit(d, (int)gridx, nchunks);
    return ks > 1 ? wino24_ksplit_floats(gridx, ks) : 0;
}

int launch_wino24(const ramnet_conv_desc &d, hipStream_t st) {
    if (d.in_mode == RAMNET_IN_PARITY4) return launch_wino24_dgrad(d, st);
    // d.x0 = replicate-padded low-res input [B][Hin = H+4][Win = W+4][C0]; Ho, Wo = the parity grid (H, W); HoF = 2H, WoF = 2W
    RAMNET_CHECK_ARG(d.in_mode == RAMNET_IN_PLAIN && d.stride == 1);
    const bool pair = fold_wino_pair(d.Cout, d.C0);            // both column parities of a 32-channel layer in one workgroup
    const bool wide = pair || (d.Cout % 64 == 0 && d.C0 % 16 == 0);      // 64-column workgroups, chunks of 16; else 32 channels, chunks of 8
    RAMNET_CHECK_ARG(d.C0 % (wide ? 32 : 16) == 0);            // an even number of chunks (the chunk loop is unrolled by two)
    RAMNET_CHECK_ARG(d.C0 % 8 == 0 && d.Cout % 32 == 0 && d.Hin == d.Ho + 4 && d.Win == d.Wo + 4 && d.HoF == 2 * d.Ho && d.WoF == 2 * d.Wo);
    RAMNET_CHECK_ARG((d.epi == RAMNET_EPI_RELU || d.epi == RAMNET_EPI_LINEAR) && d.beta == 0.f && d.out_s2d == 0 && d.Ho >= 2 && d.Wo >= 2);
    Wino24Params q;
    q.x = d.x0, q.wp = d.w, q.Hp = d.Hin, q.Wp = d.Win, q.ldx = d.ld0;
    q.nchunks = d.C0 / (wide ? 16 : 8), q.nblk = pair ? 1 : d.Cout / (wide ? 64 : 32);
    q.Hc = d.Ho, q.Wc = d.Wo, q.cpc = 1;
    const size_t xb = (size_t)d.B * d.Hin * d.Win * d.ld0 * sizeof(float), wb = (size_t)100 * d.C0 * d.Cout * sizeof(float);
    RAMNET_CHECK_ARG(xb < 0xffffffffull && wb < 0x7fffffffull);
    q.xbytes = (unsigned)xb, q.wbytes = (unsigned)wb;
    const int Wt = pair ? d.Wo + 2 : d.Wo;                     // pair: one more tile column (the shifted tiling of column parity 1)
    const bool flat = wide && cdiv(Wt, 16) * cdiv(d.Ho, 8) < cdiv(Wt, 8) * cdiv(d.Ho, 16);      // 8 x 16 instead of 16 x 8 pixels
    q.tiles_x = cdiv(Wt, wide && !flat ? 8 : 16), q.tiles_y = cdiv(d.Ho, flat ? 8 : 16);
    const size_t lds = (size_t)2 * W24_V * sizeof(float);
    const unsigned gridx = (unsigned)(q.tiles_x * q.tiles_y * d.B * q.nblk * (pair ? 2 : 4));
    const int want = wino24_ksplit(d, (int)gridx, q.nchunks);
    q.ksplit = d.splitk_ws ? want : 1;
    q.cnt = reinterpret_cast<int *>(d.splitk_ws);
    q.ws = d.splitk_ws ? d.splitk_ws + cdiv((int)gridx, 64) * 64 : nullptr;
    if (q.ksplit > 1) RAMNET_CHECK_ARG(d.splitk_floats >= wino24_ksplit_floats(gridx, want) && ((uintptr_t)d.splitk_ws & 15) == 0);
    const dim3 grid(gridx, q.ksplit);
    if (pair && flat) {
        RAMNET_FULL_LDS((conv_wino24_kernel<4, false, 8, true>));
        note_kernel("conv_wino24_kernel<4,0,8,1>");
        hipLaunchKernelGGL((conv_wino24_kernel<4, false, 8, true>), grid, dim3(512), lds, st, d, q);
    } else if (pair) {
        RAMNET_FULL_LDS((conv_wino24_kernel<4, false, 4, true>));
        note_kernel("conv_wino24_kernel<4,0,4,1>");
        hipLaunchKernelGGL((conv_wino24_kernel<4, false, 4, true>), grid, dim3(512), lds, st, d, q);
    } else if (wide && flat) {
        RAMNET_FULL_LDS((conv_wino24_kernel<4, false, 8>));
        note_kernel("conv_wino24_kernel<4,0,8,0>");
        hipLaunchKernelGGL((conv_wino24_kernel<4, false, 8>), grid, dim3(512), lds, st, d, q);
    } else if (wide) {
        RAMNET_FULL_LDS((conv_wino24_kernel<4, false, 4>));
        note_kernel("conv_wino24_kernel<4,0,4,0>");
        hipLaunchKernelGGL((conv_wino24_kernel<4, false, 4>), grid, dim3(512), lds, st, d, q);
    } else {
        RAMNET_FULL_LDS((conv_wino24_kernel<2, false, 8>));
        note_kernel("conv_wino24_kernel<2,0,8,0>");
        hipLaunchKernelGGL((conv_wino24_kernel<2, false, 8>), grid, dim3(512), lds, st, d, q);
    }
    RAMNET_LAUNCH_CHECK();
    return 0;
}

// Backward-data of the folded layer on conv_wino24_kernel (RAMNET_IN_PARITY4): Winograd weights of the FLIPPED parity filters with the roles
// of the channels swapped — reduce over k = (parity class (p, q), output channel n), produce input channels c.  Layout = the forward pack's
// with one class, chunks of 16 and 64-column blocks: i = ((((chunk*NB + nb)*25 + pos)*4 + cq)*4 + ks)*64 + l15*4 + j,
// k = chunk*16 + ks*4 + j = (p*2 + q)*Cout + n, c = nb*64 + cq*16 + l15;
// U[a][b][k][c] = sum_{t,s} G[a][t] G[b][s] W4[n][c][p][q][3-t][3-s],  W4 = FA_p w FA_q^T (double arithmetic, like the forward pack)
__global__ void pack_weight_fold_wino_dgrad_kernel(const float *__restrict__ w, float *__restrict__ wp, int Cout, int Cin, size_t total) {
    const double FA[2][4][5] = {{{.25, 0, 0, 0, 0}, {.75, .75, .25, 0, 0}, {0, .25, .75, .75, .25}, {0, 0, 0, .25, .75}},
                                {{.75, .25, 0, 0, 0}, {.25, .75, .75, .25, 0}, {0, 0, .25, .75, .75}, {0, 0, 0, 0, .25}}};
    const double G[5][4] = {{0.5, 0, 0, 0}, {-0.5, -0.5, -0.5, -0.5}, {-1.0 / 6, 1.0 / 6, -1.0 / 6, 1.0 / 6}, {1.0 / 6, 1.0 / 3, 2.0 / 3, 4.0 / 3}, {0, 0, 0, 1}};
    const int NB = Cin / 64;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t r = i;
        const int j = (int)(r & 3), l15 = (int)((r >> 2) & 15), ks = (int)((r >> 6) & 3), cq = (int)((r >> 8) & 3);
        r >>= 10;
        const int pos = (int)(r % 25);
        r /= 25;
        const int nb = (int)(r % NB), chunk = (int)(r / NB);
        const int k = chunk * 16 + ks * 4 + j, c = nb * 64 + cq * 16 + l15;
        const int pq = k / Cout, n = k - pq * Cout, pp = pq >> 1, qq = pq & 1, a = pos / 5, b = pos % 5;
        double ga[5], gb[5];
        for (int kh = 0; kh < 5; ++kh) {
            ga[kh] = 0, gb[kh] = 0;
            for (int t = 0; t < 4; ++t) ga[kh] += G[a][t] * FA[pp][3 - t][kh], gb[kh] += G[b][t] * FA[qq][3 - t][kh];
        }
        double u = 0;
        const float *wk = w + ((size_t)n * Cin + c) * 25;
        for (int kh = 0; kh < 5; ++kh)
            for (int kw = 0; kw < 5; ++kw) u += ga[kh] * gb[kw] * (double)wk[kh * 5 + kw];
        wp[i] = (float)u;
    }
}

// Border matrices of the folded layer (ops._folded_upsample_conv): the layer zero-pads where the folded form replicate-extends, so the taps
// that fall outside the image are taken out again through two small GEMMs per border.  rows[side][kx*Cin + ci][slot*Cout + co] =
// -sum_{a in LOST[side][slot]} w[co][ci][a][kx]; cols the same with the lost direction along kx; *_t = their transposes (backward-data).
__global__ void pack_border_weights_kernel(const float *__restrict__ w, float *__restrict__ rows, float *__restrict__ cols, float *__restrict__ rows_t,
                                           float *__restrict__ cols_t, int Cout, int Cin, size_t total) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t r = i;
        const int co = (int)(r % Cout);
        r /= Cout;
        const int slot = (int)(r & 1);
        r >>= 1;
        const int ci = (int)(r % Cin);
        r /= Cin;
        const int kb = (int)(r % 5), side = (int)(r / 5);
        // LOST = [[(0, 1), (0,)], [(4,), (3, 4)]]
        const int a0 = side == 0 ? 0 : (slot == 0 ? 4 : 3), na = side == 0 ? (slot == 0 ? 2 : 1) : (slot == 0 ? 1 : 2);
        const float *wk = w + ((size_t)co * Cin + ci) * 25;
        float sr = 0.f, sc = 0.f;
        for (int e = 0; e < na; ++e) sr += wk[(a0 + e) * 5 + kb], sc += wk[kb * 5 + (a0 + e)];
        const size_t K = (size_t)kb * Cin + ci, N = (size_t)slot * Cout + co;
        rows[((size_t)side * 5 * Cin + K) * 2 * Cout + N] = -sr;
        cols[((size_t)side * 5 * Cin + K) * 2 * Cout + N] = -sc;
        rows_t[((size_t)side * 2 * Cout + N) * 5 * Cin + K] = -sr;
        cols_t[((size_t)side * 2 * Cout + N) * 5 * Cin + K] = -sc;
    }
}

// End of a backward pass of a folded decoder: dW5[o][i][k][l] += sum_{p,q,t,s} FA[p][t][k] FA[q][s][l] dW4[p][q][t][s][i][o]
//   - (border-GEMM gradients routed back to the taps they summed),   dW4 = w4 (direct parity launches) + G^T dU G (Winograd-domain launches).
// One thread per (o, i); the workspaces it reads are zeroed for the next pass.
__global__ void fold_unpack_wgrad_kernel(float *__restrict__ w4, float *__restrict__ dU, float *__restrict__ wr, float *__restrict__ wc,
                                         float *__restrict__ grad, int Cout, int Cin, int CinWs) {
    const float FA[2][4][5] = {{{.25f, 0, 0, 0, 0}, {.75f, .75f, .25f, 0, 0}, {0, .25f, .75f, .75f, .25f}, {0, 0, 0, .25f, .75f}},
                               {{.75f, .25f, 0, 0, 0}, {.25f, .75f, .75f, .25f, 0}, {0, 0, .25f, .75f, .75f}, {0, 0, 0, 0, .25f}}};
    const float G[5][4] = {{0.5f, 0, 0, 0}, {-0.5f, -0.5f, -0.5f, -0.5f}, {-1.f / 6, 1.f / 6, -1.f / 6, 1.f / 6}, {1.f / 6, 1.f / 3, 2.f / 3, 4.f / 3}, {0, 0, 0, 1.f}};
    const size_t total = (size_t)Cout * Cin;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int o = (int)(idx % Cout), i = (int)(idx / Cout);        // (consecutive threads: consecutive o = the workspaces' fastest index)
        float g5[25];
        for (int e = 0; e < 25; ++e) g5[e] = 0.f;
        for (int pq = 0; pq < 4; ++pq) {
            const int pp = pq >> 1, qq = pq & 1;
            float d4[16];
            for (int ts = 0; ts < 16; ++ts) {
                float *cell = w4 + ((size_t)(pq * 16 + ts) * CinWs + i) * Cout + o;
                d4[ts] = *cell;
                *cell = 0.f;
            }
            if (dU != nullptr) {
                for (int ab = 0; ab < 25; ++ab) {
                    float *cell = dU + ((size_t)(pq * 25 + ab) * CinWs + i) * Cout + o;
                    const float u = *cell;
                    *cell = 0.f;
                    const int a = ab / 5, b = ab % 5;
                    for (int t = 0; t < 4; ++t)
                        for (int s = 0; s < 4; ++s) d4[t * 4 + s] += G[a][t] * G[b][s] * u;
                }
            }
            for (int t = 0; t < 4; ++t)
                for (int s = 0; s < 4; ++s) {
                    const float v = d4[t * 4 + s];
                    for (int k = 0; k < 5; ++k) {
                        const float fk = FA[pp][t][k] * v;
                        for (int l = 0; l < 5; ++l) g5[k * 5 + l] += fk * FA[qq][s][l];
                    }
                }
        }
        for (int side = 0; side < 2; ++side)
            for (int slot = 0; slot < 2; ++slot) {
                const int a0 = side == 0 ? 0 : (slot == 0 ? 4 : 3), na = side == 0 ? (slot == 0 ? 2 : 1) : (slot == 0 ? 1 : 2);
                for (int kb = 0; kb < 5; ++kb) {
                    float *rc = wr + (((size_t)side * 5 + kb) * Cin + i) * 2 * Cout + (size_t)slot * Cout + o;
                    float *cc = wc + (((size_t)side * 5 + kb) * Cin + i) * 2 * Cout + (size_t)slot * Cout + o;
                    const float r = *rc, c = *cc;
                    *rc = 0.f, *cc = 0.f;
                    for (int e = 0; e < na; ++e) g5[(a0 + e) * 5 + kb] -= r, g5[kb * 5 + (a0 + e)] -= c;
                }
            }
        float *g = grad + ((size_t)o * Cin + i) * 25;
        for (int e = 0; e < 25; ++e) g[e] += g5[e];
    }
}

}  // namespace ramnet

using namespace ramnet;

extern "C" int ramnet_fold_wino_supported(int Cout, int Cin) {
    int kc, ncq;
    return fold_wino_geometry(Cout, Cin, kc, ncq) ? 1 : 0;
}

extern "C" size_t ramnet_packed_weight_elems_fold_wino(int Cout, int Cin) { return (size_t)100 * Cout * Cin; }

extern "C" int ramnet_pack_weight_fold_wino(const float *w, float *wp, int Cout, int Cin, void *stream) {
    int kc, ncq;
    RAMNET_CHECK_ARG(w && wp && Cout > 0 && Cin > 0 && fold_wino_geometry(Cout, Cin, kc, ncq));
    const size_t total = (size_t)100 * Cout * Cin;
    size_t blocks = (total + 255) / 256;
    if (blocks > 65535) blocks = 65535;
    hipLaunchKernelGGL(pack_weight_fold_wino_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, wp, Cout, Cin, kc, ncq,
                       (int)fold_wino_pair(Cout, Cin), total);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int ramnet_pack_weight_fold_wino_dgrad(const float *w, float *wp, int Cout, int Cin, void *stream) {
    RAMNET_CHECK_ARG(w && wp && Cout > 0 && Cin > 0 && Cin % 64 == 0 && (4 * Cout) % 16 == 0);
    const size_t total = (size_t)100 * Cout * Cin;
    size_t blocks = (total + 255) / 256;
    if (blocks > 65535) blocks = 65535;
    hipLaunchKernelGGL(pack_weight_fold_wino_dgrad_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, wp, Cout, Cin, total);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int ramnet_pack_border_weights(const float *w, float *rows, float *cols, float *rows_t, float *cols_t, int Cout, int Cin, void *stream) {
    RAMNET_CHECK_ARG(w && rows && cols && rows_t && cols_t && Cout > 0 && Cin > 0);
    const size_t total = (size_t)2 * 5 * Cin * 2 * Cout;
    size_t blocks = (total + 255) / 256;
    if (blocks > 65535) blocks = 65535;
    hipLaunchKernelGGL(pack_border_weights_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, rows, cols, rows_t, cols_t, Cout, Cin, total);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int ramnet_fold_unpack_wgrad(float *w4, float *dU, float *wr, float *wc, float *grad, int Cout, int Cin, int CinWs, void *stream) {
    RAMNET_CHECK_ARG(w4 && wr && wc && grad && Cout > 0 && Cin > 0 && CinWs >= Cin);
    const size_t total = (size_t)Cout * Cin;
    size_t blocks = (total + 127) / 128;
    if (blocks > 65535) blocks = 65535;
    hipLaunchKernelGGL(fold_unpack_wgrad_kernel, dim3((unsigned)blocks), dim3(128), 0, (hipStream_t)stream, w4, dU, wr, wc, grad, Cout, Cin, CinWs);
    RAMNET_LAUNCH_CHECK();
    return 0;
}
