// Implicit-GEMM convolution for gfx950 (MI355X): forward and backward-data of every conv on the
// RAM-Net path (5x5 s1/s2 encoders, 3x3 ConvGRU/ConvLSTM gates, residual blocks, upsample-conv
// decoders), fp32 in / fp32 accumulate on v_mfma_f32_32x32x2_f32 (exact fp32, 157 TF peak).
//
// Formulation ("halo patch"): a workgroup (4 waves) owns TH x 16 output pixels x BN output
// channels.  For every chunk of 16 input channels it stages the input patch (tile + halo) ONCE in
// LDS — concatenation [x,h], the h*r product of the GRU candidate, the bilinear x2 upsample (+skip
// sum) of the decoders and the ReLU mask of the backward pass are all applied while staging — and
// then walks the filter taps; each tap is a K=16 slab of the GEMM whose A operand is the patch
// shifted by the tap offset (no im2col traffic: the patch is read from HBM/L2 once per 16 channels
// instead of once per tap) and whose B operand is a [BN][16] weight tile streamed through a
// double-buffered LDS ring (global loads of tap t+1 are in flight under the MFMAs of tap t).
// Fused epilogues: bias/ReLU/sigmoid, residual add, GRU blend, full LSTM cell.
#include <stdlib.h>
#include "common.hpp"
#include "conv_epilogue.hpp"

namespace ramnet {

struct ConvCommon {
    InSrc src;
    int nchunks, CoutPad;            // weight geometry
    int patch_floats;                // LDS floats reserved per patch plane (max over the launch's classes, multiple of 4)
    int nclass;
    int xcd_swizzle;
};

// One output class of a launch: a plain convolution has one; the backward-data of a stride-2 layer (and the forward of
// the transposed-conv decoder) has four — one per output parity — folded into ONE launch (blockIdx.z) so that the
// quarter-size sub-problems fill the chip together instead of as four short, badly quantised launches.
struct ConvClass {
    int ntaps, Ho, Wo, ooy, oox;
    int PH, PW, dymin, dxmin;        // LDS patch geometry
    int tiles_x, tiles_y;
    int toff[25];                    // per-tap patch offset (floats)
    unsigned woff[25];               // per-tap weight slice offset in 64-byte rows
};
struct ConvClasses { ConvClass c[4]; };

// Exact fp32 (v_mfma_f32_32x32x2_f32), 16 channels per chunk.
template <int BM, int BN, int WM, int WN>
__global__ void __launch_bounds__(256) conv_igemm_kernel(const ramnet_conv_desc p, const ConvCommon qc, const ConvClasses qk) {
    // XCD-aware tile order: the dispatcher deals consecutive (flattened) workgroup ids round-robin to the 8 XCDs, each with
    // a private L2.  Remap the flattened id so that every XCD walks a CONTIGUOUS range of (class, channel tile, spatial
    // tile): neighbouring spatial tiles then share their halo rows and their weight tile through ONE L2 instead of
    // fetching them on 8.  Bijective for any grid size (q/r split); only speed depends on the placement assumption.
    int flat = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if (qc.xcd_swizzle) {
        const int nwg = gridDim.x * gridDim.y * gridDim.z, xcd = flat & 7, qq = nwg >> 3, rr = nwg & 7;
        flat = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (flat >> 3);
    }
    int bid = flat % gridDim.x;
    const int by = (flat / gridDim.x) % gridDim.y, bz = flat / (gridDim.x * gridDim.y);
    const ConvClass &q = qk.c[bz];
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int TH = BM / TWID;
    constexpr int CKC = CK;                        // input channels per chunk
    static_assert(WM * WN == 4 && TM >= 1 && TN >= 1, "4 waves per workgroup");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *patch = smem;                            // [PH*PW][LDP]
    float *wsm = smem + qc.patch_floats;            // [2][BN][LDP]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, kk = lane >> 5;

    if (bid >= q.tiles_x * q.tiles_y * p.B) return;                   // classes of one launch may differ by a tile
    const int tx_i = bid % q.tiles_x;
    bid /= q.tiles_x;
    const int ty_i = bid % q.tiles_y;
    const int b = bid / q.tiles_y;
    const int n0 = by * BN;
    const int oy0 = ty_i * TH, ox0 = tx_i * TWID;
    const int iy0 = oy0 * p.stride + q.dymin, ix0 = ox0 * p.stride + q.dxmin;

    int aBase[TM], bBase[TN];
#pragma unroll
    for (int ms = 0; ms < TM; ++ms) {
        const int m = (wm * TM + ms) * 32 + l31;
        aBase[ms] = (((m >> 4) * p.stride) * q.PW + (m & 15) * p.stride) * LDP + 4 * kk;
    }
#pragma unroll
    for (int ns = 0; ns < TN; ++ns) bBase[ns] = ((wn * TN + ns) * 32 + l31) * LDP + 4 * kk;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int ms = 0; ms < TM; ++ms)
#pragma unroll
        for (int ns = 0; ns < TN; ++ns)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ms][ns][r] = 0.f;

    // ---- weight tile ring: per (chunk, tap) [BN] rows x 64 bytes, contiguous in global memory
    constexpr int WF4 = BN * 4;                     // 16-byte units per tile
    constexpr int WPT = (WF4 + 255) / 256;
    float4 wreg[WPT];
    const int ntaps = q.ntaps;
    auto load_w = [&](int chunk, int t) {
        const float *src = p.w + ((size_t)q.woff[t] + ((size_t)chunk * qc.CoutPad + n0)) * 16;
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int f = tid + i * 256;
            if (WF4 % 256 == 0 || f < WF4) wreg[i] = ld4(src + (size_t)f * 4);
        }
    };
    auto store_w = [&](int buf) {
        float *dst = wsm + buf * (BN * LDP);
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int f = tid + i * 256;
            if (WF4 % 256 == 0 || f < WF4) st4(dst + (f >> 2) * LDP + (f & 3) * 4, wreg[i]);
        }
    };

    load_w(0, 0);
    int buf = 0;
    for (int chunk = 0; chunk < qc.nchunks; ++chunk) {
        __syncthreads();   // every wave is done reading the previous chunk's patch
        stage_patch<CKC / 4, LDP, 4, 256>(patch, qc.src, b, iy0, ix0, chunk * CKC, q.PH, q.PW, tid);
        for (int t = 0; t < ntaps; ++t) {
            store_w(buf);
            __syncthreads();   // patch + weight tile visible; the other ring slot is free again
            if (t + 1 < ntaps) load_w(chunk, t + 1);
            else if (chunk + 1 < qc.nchunks) load_w(chunk + 1, 0);
            const float *pa = patch + q.toff[t];
            const float *wb = wsm + buf * (BN * LDP);
#pragma unroll
            for (int k8 = 0; k8 < 2; ++k8) {
                float4 a[TM], bb[TN];
#pragma unroll
                for (int ms = 0; ms < TM; ++ms) a[ms] = ld4(pa + aBase[ms] + k8 * 8);
#pragma unroll
                for (int ns = 0; ns < TN; ++ns) bb[ns] = ld4(wb + bBase[ns] + k8 * 8);
                // lanes 0-31 feed channels k8*8+j, lanes 32-63 channels k8*8+4+j: 4 MFMAs cover 8 channels
#pragma unroll
                for (int ms = 0; ms < TM; ++ms)
#pragma unroll
                    for (int ns = 0; ns < TN; ++ns) {
                        acc[ms][ns] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ms].x, bb[ns].x, acc[ms][ns], 0, 0, 0);
                        acc[ms][ns] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ms].y, bb[ns].y, acc[ms][ns], 0, 0, 0);
                        acc[ms][ns] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ms].z, bb[ns].z, acc[ms][ns], 0, 0, 0);
                        acc[ms][ns] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ms].w, bb[ns].w, acc[ms][ns], 0, 0, 0);
                    }
            }
            buf ^= 1;
        }
    }

    // ---- epilogue.  D layout of 32x32 MFMA: col = lane&31 (output channel), row = (r&3)+8*(r>>2)+4*(lane>>5) (pixel)
    const int epi = p.epi;
#pragma unroll
    for (int ms = 0; ms < TM; ++ms) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (wm * TM + ms) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
            const int oy = oy0 + (m >> 4), ox = ox0 + (m & 15);
            if (oy >= q.Ho || ox >= q.Wo) continue;
            const size_t pix = ((size_t)b * p.HoF + (oy * p.osy + q.ooy)) * p.WoF + (ox * p.osx + q.oox);
            const bool addold = epilogue_addold(p, oy * p.osy + q.ooy, ox * p.osx + q.oox);
            if (epi == RAMNET_EPI_LSTM) {
                if constexpr (TN == 4) {
                    // packed N order = (channel block of 32, gate, channel): ns is the gate (i, f, o, g)
                    const int C = p.Cout, ch = by * 32 + l31;
                    if (ch < C) {
                        const float gi = sigmoidf_(acc[ms][0][r] + p.bias[ch]);
                        const float gf = sigmoidf_(acc[ms][1][r] + p.bias[C + ch]);
                        const float go = sigmoidf_(acc[ms][2][r] + p.bias[2 * C + ch]);
                        const float gc = tanhf_(acc[ms][3][r] + p.bias[3 * C + ch]);
                        const float cp = p.e1 ? p.e1[pix * p.lde1 + ch] : 0.f;
                        const float cn = gf * cp + gi * gc;
                        p.out[pix * p.ldo + ch] = go * tanhf_(cn);
                        p.o1[pix * p.ldo1 + ch] = cn;
                        if (p.o2) {
                            float *g = p.o2 + pix * p.ldo2 + ch;
                            g[0] = gi, g[C] = gf, g[2 * C] = go, g[3 * C] = gc;
                        }
                    }
                }
                continue;
            }
#pragma unroll
            for (int ns = 0; ns < TN; ++ns) {
                const int n = n0 + (wn * TN + ns) * 32 + l31;
                if (n >= p.Cout) continue;
                epilogue_store(p, epi, pix, n, acc[ms][ns][r] + epilogue_side(p, epi, b, oy * p.osy + q.ooy, ox * p.osx + q.oox, n), addold);
            }
        }
    }
}

template <int BM, int BN, int WM, int WN>
static int launch_cfg(const ramnet_conv_desc &d, const ConvCommon &qc, const ConvClasses &qk, int max_tiles, hipStream_t st) {
    auto kern = conv_igemm_kernel<BM, BN, WM, WN>;
    const size_t lds = ((size_t)qc.patch_floats + 2 * BN * LDP) * sizeof(float);
    static size_t lds_set = 0;   // raise the dynamic-LDS cap once per instantiation
    if (lds > lds_set) {
        RAMNET_FULL_LDS((kern));
        lds_set = 160 * 1024;
    }
    if (lds > 160 * 1024) {
        set_error("conv patch does not fit LDS (%zu bytes)", lds);
        return RAMNET_E_UNSUPPORTED;
    }
    dim3 grid(max_tiles * d.B, qc.CoutPad / BN, qc.nclass);
    note_kernel("conv_igemm_kernel<%d,%d,%d,%d>", BM, BN, WM, WN);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, d, qc, qk);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

static int check_desc(const ramnet_conv_desc &d) {
    RAMNET_CHECK_ARG(d.x0 && d.w && d.out);
    RAMNET_CHECK_ARG(d.ntaps >= 1 && d.ntaps <= 25 && (d.stride == 1 || d.stride == 2));
    RAMNET_CHECK_ARG(d.B > 0 && d.Ho > 0 && d.Wo > 0 && d.Hin > 0 && d.Win > 0 && d.Cout > 0);
    RAMNET_CHECK_ARG(d.C0 > 0 && d.C0 % 4 == 0 && d.ld0 % 4 == 0 && d.ldo > 0);
    RAMNET_CHECK_ARG(d.osy >= 1 && d.osx >= 1);
    const bool cat = d.in_mode == RAMNET_IN_CAT || d.in_mode == RAMNET_IN_CAT_MUL;
    if (cat) RAMNET_CHECK_ARG(d.x1 && d.C1 > 0 && d.C1 % 4 == 0 && d.ld1 % 4 == 0);
    if (d.in_mode == RAMNET_IN_CAT_MUL || d.in_mode == RAMNET_IN_RELUMASK) RAMNET_CHECK_ARG(d.xm && d.ldm % 4 == 0);
    if (d.in_mode == RAMNET_IN_UP2X_SKIP) RAMNET_CHECK_ARG(d.x1 && d.ld1 % 4 == 0);
    if (d.in_mode == RAMNET_IN_UP2X || d.in_mode == RAMNET_IN_UP2X_SKIP) RAMNET_CHECK_ARG(d.Hin % 2 == 0 && d.Win % 2 == 0);
    if (d.epi == RAMNET_EPI_RES_RELU) RAMNET_CHECK_ARG(d.e0);
    if (d.epi == RAMNET_EPI_GRU_BLEND) RAMNET_CHECK_ARG(d.e0);
    if (d.epi == RAMNET_EPI_LSTM) RAMNET_CHECK_ARG(d.o1 && d.bias);
    if (d.epi == RAMNET_EPI_GRU_BWD) RAMNET_CHECK_ARG(d.e0 && d.o1 && d.Cout % 8 == 0 && !d.bias && d.beta == 0.f && d.frame == 0 && d.out_s2d == 0);
    if (d.epi == RAMNET_EPI_SIGMOID_HR) RAMNET_CHECK_ARG(d.e1 && d.o1 && d.Cout % 8 == 0 && d.beta == 0.f && d.frame == 0 && d.out_s2d == 0 &&
                                                         d.algo != RAMNET_ALGO_WINOGRAD24 && d.algo != RAMNET_ALGO_HEAD);
    // the patch loaders address their source tensors with 32-bit element offsets (16 GB per tensor: batch ~45 at 256x344x32)
    {
        unsigned long long ld = (unsigned long long)d.ld0;
        if (cat || d.in_mode == RAMNET_IN_UP2X_SKIP) ld = ld > (unsigned long long)d.ld1 ? ld : (unsigned long long)d.ld1;
        if (d.xm) ld = ld > (unsigned long long)d.ldm ? ld : (unsigned long long)d.ldm;
        const unsigned long long px = (unsigned long long)d.B * d.Hin * d.Win * (d.in_mode == RAMNET_IN_S2D ? 4 : 1);
        RAMNET_CHECK_ARG(px * ld < (1ull << 32));
    }
    RAMNET_CHECK_ARG(d.frame >= 0);
    if (d.frame > 0) RAMNET_CHECK_ARG(d.e0 && d.e1 && (d.epi == RAMNET_EPI_RELU || d.epi == RAMNET_EPI_LINEAR));
    return 0;
}

// n descriptors that differ ONLY in their tap lists and output sub-grid (Ho, Wo, ooy, oox) -> one launch
static int launch_classes(const ramnet_conv_desc *ds, int n, hipStream_t st) {
    const ramnet_conv_desc &d = ds[0];
    for (int i = 0; i < n; ++i) {
        const int rc = check_desc(ds[i]);
        if (rc) return rc;
        RAMNET_CHECK_ARG(ds[i].x0 == d.x0 && ds[i].w == d.w && ds[i].out == d.out && ds[i].stride == d.stride &&
                         ds[i].Cout == d.Cout && ds[i].epi == d.epi && ds[i].osy == d.osy && ds[i].osx == d.osx &&
                         ds[i].B == d.B && ds[i].in_mode == d.in_mode);
    }
    const bool cat = d.in_mode == RAMNET_IN_CAT || d.in_mode == RAMNET_IN_CAT_MUL;
    ConvCommon qc;
    ConvClasses qk;
    qc.src.x0 = d.x0, qc.src.x1 = d.x1, qc.src.xm = d.xm;
    qc.src.ld0 = d.ld0, qc.src.ld1 = d.ld1, qc.src.ldm = d.ldm;
    qc.src.C0 = d.C0, qc.src.Cin = d.C0 + (cat ? d.C1 : 0);
    qc.src.mode = d.in_mode, qc.src.Hin = d.Hin, qc.src.Win = d.Win;
    qc.nchunks = cdiv(qc.src.Cin, CK);
    qc.CoutPad = d.epi == RAMNET_EPI_LSTM ? 4 * roundup(d.Cout, 32) : roundup(d.Cout, 32);
    qc.nclass = n;
    // measured on MI355X (profiles/r01_b_tuning_notes.md): forward 5.55 -> 5.50 ms, backward-data 5.40 -> 5.56 ms per pass,
    // i.e. null within noise — these kernels are MFMA-bound and their operands sit in L2 / Infinity Cache either way.
    // Off.
    qc.xcd_swizzle = 0;
    // ---- tile configuration.  Low-resolution layers (32x43 .. 64x86 pixels) give few 128-pixel tiles: a grid that does
    // not cover the 256 CUs ~3x over leaves CUs idle in the last round (tile quantisation), so shrink the tile there.
    const int lstm = d.epi == RAMNET_EPI_LSTM;
    int BM = 128, BN = lstm ? 128 : (qc.CoutPad % 128 == 0 ? 128 : qc.CoutPad % 64 == 0 ? 64 : 32);
    auto blocks = [&](int bm, int bn) {
        long t = 0;
        for (int i = 0; i < n; ++i) t += (long)cdiv(ds[i].Wo, TWID) * cdiv(ds[i].Ho, bm / TWID);
        return t * d.B * (qc.CoutPad / bn);
    };
    if (!lstm && BN >= 64) {
        const long want = 768;                                  // minimum workgroups before shrinking tiles
        if (blocks(BM, BN) < want) BM = 64;
        if (blocks(BM, BN) < want && BN == 128) BN = 64;
    }
    if (!lstm && BN == 32 && blocks(256, 32) >= 1024)
        BM = 256;                    // 32-channel outputs: 16x16-pixel tiles, 2 accumulators per wave per tap
    const int TH = BM / TWID;
    int max_tiles = 0;
    qc.patch_floats = 0;
    for (int i = 0; i < n; ++i) {
        const ramnet_conv_desc &e = ds[i];
        ConvClass &q = qk.c[i];
        int dymin = 127, dymax = -127, dxmin = 127, dxmax = -127;
        for (int t = 0; t < e.ntaps; ++t) {
            dymin = e.dy[t] < dymin ? e.dy[t] : dymin, dymax = e.dy[t] > dymax ? e.dy[t] : dymax;
            dxmin = e.dx[t] < dxmin ? e.dx[t] : dxmin, dxmax = e.dx[t] > dxmax ? e.dx[t] : dxmax;
        }
        q.ntaps = e.ntaps, q.Ho = e.Ho, q.Wo = e.Wo, q.ooy = e.ooy, q.oox = e.oox;
        q.dymin = dymin, q.dxmin = dxmin;
        q.PH = (TH - 1) * e.stride + (dymax - dymin) + 1;
        q.PW = (TWID - 1) * e.stride + (dxmax - dxmin) + 1;
        q.tiles_x = cdiv(e.Wo, TWID), q.tiles_y = cdiv(e.Ho, TH);
        for (int t = 0; t < e.ntaps; ++t) {
            q.toff[t] = ((e.dy[t] - dymin) * q.PW + (e.dx[t] - dxmin)) * LDP;
            q.woff[t] = (unsigned)e.wtap[t] * (unsigned)qc.nchunks * (unsigned)qc.CoutPad;
        }
        if (q.PH * q.PW * LDP > qc.patch_floats) qc.patch_floats = q.PH * q.PW * LDP;
        if (q.tiles_x * q.tiles_y > max_tiles) max_tiles = q.tiles_x * q.tiles_y;
    }
    if (lstm) return launch_cfg<128, 128, 4, 1>(d, qc, qk, max_tiles, st);
    if (BM == 128 && BN == 128) return launch_cfg<128, 128, 2, 2>(d, qc, qk, max_tiles, st);
    if (BM == 128 && BN == 64) return launch_cfg<128, 64, 2, 2>(d, qc, qk, max_tiles, st);
    if (BM == 64 && BN == 128) return launch_cfg<64, 128, 2, 2>(d, qc, qk, max_tiles, st);
    if (BM == 64 && BN == 64) return launch_cfg<64, 64, 2, 2>(d, qc, qk, max_tiles, st);
    if (BM == 256) return launch_cfg<256, 32, 4, 1>(d, qc, qk, max_tiles, st);
    return launch_cfg<128, 32, 4, 1>(d, qc, qk, max_tiles, st);
}

}  // namespace ramnet

using namespace ramnet;

extern "C" int ramnet_conv_launch(const ramnet_conv_desc *dp, void *stream) {
    RAMNET_CHECK_ARG(dp != nullptr);
    if (dp->algo == RAMNET_ALGO_WINOGRAD) {
        const int rc = check_desc(*dp);
        return rc ? rc : launch_wino(*dp, (hipStream_t)stream);
    }
    if (dp->algo == RAMNET_ALGO_WINOGRAD_2X4) {
        const int rc = check_desc(*dp);
        return rc ? rc : launch_wino6(*dp, (hipStream_t)stream);
    }
    if (dp->algo == RAMNET_ALGO_WINOGRAD_2X4_SPLIT) {
        const int rc = check_desc(*dp);
        return rc ? rc : launch_wino6s(*dp, (hipStream_t)stream);
    }
    if (dp->algo == RAMNET_ALGO_WINOGRAD24) {
        const int rc = check_desc(*dp);
        return rc ? rc : launch_wino24(*dp, (hipStream_t)stream);
    }
    if (dp->algo == RAMNET_ALGO_HEAD) {
        const int rc = check_desc(*dp);
        return rc ? rc : launch_head(*dp, (hipStream_t)stream);
    }
    RAMNET_CHECK_ARG(dp->algo == RAMNET_ALGO_DIRECT && dp->in_mode != RAMNET_IN_S2D && dp->out_s2d == 0);   // fused space-to-depth: Winograd kernels only
    return launch_classes(dp, 1, (hipStream_t)stream);
}

extern "C" int ramnet_conv_launch_multi(const ramnet_conv_desc *descs, int n, void *stream) {
    RAMNET_CHECK_ARG(descs != nullptr && n >= 1 && n <= 4);
    for (int i = 0; i < n; ++i) RAMNET_CHECK_ARG(descs[i].in_mode != RAMNET_IN_S2D && descs[i].out_s2d == 0);
    return launch_classes(descs, n, (hipStream_t)stream);
}
