/*
 * ramnet_hip.h — C ABI of librpg_ramnet_hip.so (MI355X / gfx950 RAM-Net hot path).
 *
 * The reference (uzh-rpg/rpg_ramnet) is pure Python on torch.nn ops and has no FFI; each entry
 * point below names the reference code it replaces (file:line relative to RAM_Net/).  All
 * pointers are DEVICE pointers (fp32 unless noted), `stream` is a hipStream_t passed as void*,
 * every function returns 0 on success or a non-zero code (hipError_t value, or RAMNET_E_*);
 * ramnet_last_error() returns a static, thread-local message for the last failure.
 *
 * Activation layout: NHWC ("channels_last") — element (b, y, x, c) of a tensor with leading
 * dimension ld lives at ((b*H + y)*W + x)*ld + c.  `ld` lets a tensor be a channel slice of a
 * wider one.  Weights are given in PyTorch OIHW and re-packed by ramnet_pack_weight().
 */
#ifndef RAMNET_HIP_H
#define RAMNET_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RAMNET_ABI_VERSION 24      /* 24: ramnet_cat_batch_add_masked (the gradient of a time-batched ReLU feature leaves its fan-in already masked), ramnet_pred_sigmoid_si_bwd takes the forward's scratch (fixed-order join of the weight / bias partial sums); 23: ramnet_wgrad_desc.algo = RAMNET_ALGO_DIRECT_SPLIT (direct 3x3 backward-weights on the bf16 matrix pipe, split operands: csrc/conv_wgrad_dsplit.hip) + ramnet_wgrad_dsplit_slabs; 22: RAMNET_ALGO_WINOGRAD_2X4_SPLIT + ramnet_conv_wino_split_ok / ramnet_pack_weight_wino2x4_split (split bf16 operands on the F(2x4,3x3) forward / backward-data launches); 21: RAMNET_EPI_SIGMOID_HR (the ConvGRU gates launch also writes h.r: the candidate convolution and its backward-weights read a plain concatenation); 20 (never released on its own: shipped together with 21): ramnet_wgrad_desc.nseg / segs (multi-segment backward-weights launches: deferred ConvGRU cell updates); 19: ramnet_cat_batch_add (gradient of a time-batched feature); 18: RAMNET_EPI_GRU_BWD (stage B of the ConvGRU backward in the epilogue of the candidate convolution's backward-data launch) + ramnet_gru_bwd_a2; 17: ramnet_wgrad_desc.algo = RAMNET_ALGO_WINOGRAD_2X4 (F(2x4,3x3) backward-weights, csrc/conv_wgrad_wino6.hip) + ramnet_wgrad_wino2x4_slabs / ramnet_unpack_wgrad_wino2x4, option "wgrad_wino_nf"; 16: ramnet_conv_desc.splitk_ws / splitk_floats + ramnet_conv_splitk_floats (split channel reduction of latency-bound Winograd launches), option "wino_ksplit"; 15: ramnet_si_loss_from_stats (data-parallel exact loss), ramnet_si_log_loss_* / ramnet_mse_loss_*, ramnet_reflect_pad, ramnet_wgrad_desc.dw_slabs + ramnet_reduce_slabs, ramnet_set_option (environment knobs removed), fold weight-algebra kernels, RAMNET_ALGO_WINOGRAD_2X4 + ramnet_conv_wino_variant / ramnet_pack_weight_wino2x4; 14: ramnet_norm_* (BatchNorm / InstanceNorm); 13: pair layout of ramnet_pack_weight_fold_wino, head kernel for 10 input channels */
#define RAMNET_E_BADARG 10001
#define RAMNET_E_UNSUPPORTED 10002

/* ---- input-tile sources of the implicit-GEMM convolution (what feeds the LDS patch) ---------- */
enum ramnet_in_mode {
    RAMNET_IN_PLAIN = 0,     /* x0[C0]                                                          */
    RAMNET_IN_CAT = 1,       /* cat(x0[C0], x1[C1])            ConvGRU/ConvLSTM  submodules.py:343,447 */
    RAMNET_IN_CAT_MUL = 2,   /* cat(x0[C0], x1[C1] * xm[C1])   ConvGRU candidate submodules.py:450  */
    RAMNET_IN_UP2X = 3,      /* bilinear x2 (align_corners=False) of x0          submodules.py:88   */
    RAMNET_IN_UP2X_SKIP = 4, /* bilinear x2 of (x0 + x1)       decoder skip sum  statenet.py:305-308 */
    RAMNET_IN_RELUMASK = 5,  /* x0 * (xm > 0)                  backward through a ReLU             */
    RAMNET_IN_S2D = 6,       /* space-to-depth view of x0 [B][2*Hin][2*Win][C0]: logical channel (a*2+c)*C0 + ch of pixel
                              * (i, j) = x0(2i+a, 2j+c, ch), Cin = 4*C0 — the stride-2 5x5 encoders (submodules.py:22-48) as
                              * 3x3 stride-1 convolutions without materialising the view.  WINOGRAD launches only; C0 a power
                              * of two >= 8                                                       */
    RAMNET_IN_PARITY4 = 7    /* RAMNET_ALGO_WINOGRAD24 only: backward-data of the folded upsample-conv.  x0 = dy * mask
                              * [B][Hin = 2H][Win = 2W][C0]; its four parity sub-grids are the reduction blocks (4*C0 channels,
                              * zero outside); out = gradient of the replicate-padded low-res tensor [B][Ho = H+4][Wo = W+4][Cout],
                              * w = Winograd weights of the flipped parity filters; C0 % 16 == 0, Cout % 64 == 0                */
};

/* ---- arithmetic of the MFMA contraction: exact fp32 on v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 (bit-identical to
 * an fmaf chain, 157 TFLOP/s peak).  There is no reduced-precision mode. */
/* RAMNET_ALGO_WINOGRAD: F(2x2,3x3), fp32; only the dense 3x3 stride-1 tap list, w from ramnet_pack_weight_wino() */
/* RAMNET_ALGO_HEAD: the 5x5 stride-1 head layers with 1, 3 or 5 real input channels and <= 32 outputs (statenet.py:160-175):
 * dense (tap, channel) reduction, weights from ramnet_pack_weight_head() held in registers; fp32 */
/* RAMNET_ALGO_WINOGRAD24: all four output parities of the folded upsample-conv (decoders, statenet.py:305-308) as Winograd
 * F(2x2,4x4) convolutions of the replicate-padded low-res input: x0 = [B][Hin = H+4][Win = W+4][C0] (ramnet_pad2_sum),
 * Ho, Wo = H, W (one parity grid), out = [B][HoF = 2H][WoF = 2W][Cout]; C0 % 16 == 0 (an even number of chunks), Cout % 32 == 0; bias, LINEAR / RELU and
 * `frame` as for the direct launch.  With (KC, NCQ) = (16, 4) when Cout % 64 == 0 and C0 % 16 == 0, else (8, 2):
 * w = U[class = py*2+px][C0/KC][Cout/(16*NCQ)][25 positions][NCQ][64][KC/4] floats with
 * U[cls][pos = a*5+b][k][n] = sum_{t,s} G[a][t] W4[n][k][py][px][t][s] G[b][s] (W4 = the 4x4 parity filters, G below) stored at
 * ((((cls*(C0/KC) + k/KC)*(Cout/(16*NCQ)) + n/(16*NCQ))*25 + pos)*NCQ + (n%(16*NCQ))/16)*16*KC + (((k%KC)/(KC/4))*16 + n%16)*(KC/4) + k%(KC/4).
 * G = [1/2 0 0 0; -1/2 -1/2 -1/2 -1/2; -1/6 1/6 -1/6 1/6; 1/6 1/3 2/3 4/3; 0 0 0 1] (Toom-Cook points 0, 1, -1, 2, inf).  */
/* RAMNET_ALGO_WINOGRAD_2X4: F(2x4,3x3) — 2 x 4 output tiles, 3 instead of 4 multiplies per output (csrc/conv_wino6.hip); the same launches
 * as RAMNET_ALGO_WINOGRAD restricted to what ramnet_conv_wino_variant() accepts, w from ramnet_pack_weight_wino2x4()                 */
/* RAMNET_ALGO_WINOGRAD_2X4_SPLIT (ABI 22): the launches of RAMNET_ALGO_WINOGRAD_2X4 with every Winograd-domain product evaluated on the bf16
 * matrix pipe from three-term bf16 splits of BOTH operands (six of the nine partial products, fp32 accumulation: fp32-level accuracy;
 * csrc/conv_wino6s.hip), where ramnet_conv_wino_split_ok() accepts the descriptor; w from ramnet_pack_weight_wino2x4_split()                */
/* RAMNET_ALGO_DIRECT_SPLIT (ABI 23; ramnet_wgrad_desc only): backward-weights of the launches RAMNET_ALGO_WINOGRAD_2X4 accepts (dense 3x3 stride-1
 * taps; plain / concatenated / masked / space-to-depth inputs; nseg = 0) in DIRECT form on the bf16 matrix pipe: both operands split into three
 * bf16 terms when a strip is staged, six of the nine partial products, fp32 accumulation (csrc/conv_wgrad_dsplit.hip); dw = the blocked
 * layout of ramnet_wgrad_dsplit_ws_floats() (ramnet_unpack_wgrad_dsplit), dw_slabs as for the Winograd forms with at most
 * ramnet_wgrad_dsplit_slabs() slabs.                                                                                                         */
enum ramnet_algo { RAMNET_ALGO_DIRECT = 0, RAMNET_ALGO_WINOGRAD = 1, RAMNET_ALGO_HEAD = 2, RAMNET_ALGO_WINOGRAD24 = 3, RAMNET_ALGO_WINOGRAD_2X4 = 4,
                   RAMNET_ALGO_WINOGRAD_2X4_SPLIT = 5, RAMNET_ALGO_DIRECT_SPLIT = 6 };

/* ---- fused epilogues ------------------------------------------------------------------------- */
enum ramnet_epilogue {
    RAMNET_EPI_LINEAR = 0,    /* acc + bias                                                      */
    RAMNET_EPI_RELU = 1,      /* relu(acc + bias)                    ConvLayer submodules.py:26-35 */
    RAMNET_EPI_SIGMOID = 2,   /* sigmoid(acc + bias)                 GRU gates submodules.py:448-449 */
    RAMNET_EPI_RES_RELU = 3,  /* relu(acc + bias + e0)               ResidualBlock submodules.py:212-214 */
    RAMNET_EPI_GRU_BLEND = 4, /* o=tanh(acc+bias); out=h*(1-u)+o*u   submodules.py:450-452; e0=u, e1=h, o1<-o */
    RAMNET_EPI_LSTM = 5,      /* i,f,o,g -> c'=f*c+i*g, h'=o*tanh(c') submodules.py:346-358; e1=c, out<-h', o1<-c', o2<-gates */
    RAMNET_EPI_GRU_BWD = 6,   /* backward-data of the candidate convolution W_o*[x, h.r] (ABI 18; Cout = 2C, no bias, beta = 0): channels n < C
                               * (dx) are stored as they are; channels n >= C carry g = d(h.r): with r = e0[pix*lde0 + n] (e0 = [u|r]),
                               * h = e1[pix*lde1 + n - C] (NULL: 0): o1[pix*ldo1 + n] <- g*h*r*(1-r) (the reset gate's pre-activation gradient)
                               * and out <- out_old + g*r, out_old = dh'*(1-u) left there by ramnet_gru_bwd_a2 — ramnet_gru_bwd_b without
                               * its launch.  Winograd kernels: C % 64 == 0                                                        */
    RAMNET_EPI_SIGMOID_HR = 7 /* ConvGRU gates [u | r] = sigmoid(acc + bias) (ABI 21; Cout = 2C, beta = 0): as RAMNET_EPI_SIGMOID, and for the
                               * channels n >= C (the reset gate r) also o1[pix*ldo1 + n - C] <- e1[pix*lde1 + n - C] * r, the h.r operand of
                               * the candidate convolution W_o*[x, h.r] (submodules.py:450) — that launch, and the backward-weights launch of
                               * W_o, then read a plain concatenation [x | h.r] instead of forming the product in their loaders.  e1 = h, o1 = h.r */
};

/* One convolution launch.  out(a,b) = epi( sum_t sum_c in(a*stride+dy[t], b*stride+dx[t], c) * W[wtap[t]][c][n] ).
 * A plain KxK conv is the tap list dy,dx in [-pad, K-1-pad]; a transposed (backward-data) conv of a
 * stride-2 layer is four such launches, one per output parity class (os?/oo? below).            */
typedef struct ramnet_conv_desc {
    const float *x0, *x1, *xm;      /* input sources, see ramnet_in_mode                            */
    int ld0, ld1, ldm;              /* their leading dimensions (floats per pixel)                  */
    int C0, C1;                     /* channel counts (multiples of 4); Cin = C0 (+ C1 for CAT*)    */
    int in_mode;
    int B, Hin, Win;                /* logical input extent (AFTER the x2 upsample for UP2X*)       */
    int ntaps, stride;
    int8_t dy[25], dx[25];          /* tap offsets in input pixels                                  */
    uint8_t wtap[25];               /* which packed weight slice each tap multiplies with           */
    const float *w;                 /* packed weights from ramnet_pack_weight()                     */
    const float *bias;              /* [Cout] or NULL                                               */
    int Cout;                       /* real output channels (LSTM: hidden size C, weights hold 4C)  */
    int Ho, Wo;                     /* extent of the (a,b) grid                                     */
    int HoF, WoF;                   /* full output tensor extent                                    */
    int osy, osx, ooy, oox;         /* output pixel = (a*osy+ooy, b*osx+oox)                        */
    int epi;
    float beta;                     /* pre-activation += beta*out_old (LINEAR and RELU; 0 = overwrite), see `frame` */
    const float *e0, *e1;           /* epilogue operands                                            */
    int lde0, lde1;
    float *out, *o1, *o2;
    int ldo, ldo1, ldo2;
    int algo;                       /* RAMNET_ALGO_DIRECT, RAMNET_ALGO_WINOGRAD or RAMNET_ALGO_HEAD    */
    int frame;                      /* > 0 (folded upsample-conv, LINEAR / RELU epilogues): border corrections are added to the
                                     * pre-activation of the outermost `frame` (= 2) rows / columns of the FULL output:
                                     * rows from e1 [2 sides][B][WoF][lde1 = frame*Cout], columns from e0 [2][B][HoF][lde0]  */
    int out_s2d;                    /* C > 0 (WINOGRAD, LINEAR epilogue, no bias, beta = 0): the Cout = 4*C output channels are the
                                     * space-to-depth view of out [B][HoF = 2*Ho][WoF = 2*Wo][C]: channel (a*2+c)*C + ch of pixel (i, j)
                                     * is stored at out(2i+a, 2j+c, ch) — backward-data of the encoders.  C a power of two >= 8  */
    int head_cin;                   /* RAMNET_ALGO_HEAD: real input channels (1, 3 or 5) among the C0 padded ones of x0       */
    int s2d_5x5;                    /* 1 (WINOGRAD with in_mode RAMNET_IN_S2D, or with out_s2d): the 3x3 filter over the space-to-depth
                                     * view is that of a 5x5 stride-2 layer — its slices (dy = +1, row parity 1) and (dx = +1, column
                                     * parity 1) are zero (11 of 36) — and the kernel skips the Winograd positions those zeros annihilate
                                     * (12.25 instead of 16 multiplies per tile and channel on average).  0: dense filter.              */
    float *splitk_ws;               /* RAMNET_ALGO_WINOGRAD, optional (ABI 16): workspace of splitk_floats >= ramnet_conv_splitk_floats(d) floats,
                                     * 16-byte aligned, ZERO when first handed over and owned by this layer (launches that share it must be
                                     * ordered on one stream).  A launch far below one workgroup per CU (batch-1 streaming on the coarse
                                     * scales) then splits its channel reduction over 2-4 workgroups per output tile; the partial tiles meet
                                     * here, the last arrival sums them in split order (bit-reproducible) and runs the epilogue, and leaves
                                     * the arrival counters at zero.  NULL: never split.                                               */
    size_t splitk_floats;
} ramnet_conv_desc;

/* Weight-gradient launch: dW[t][c][n] += sum_{b,a,b'} in(a*stride+dy[t], b'*stride+dx[t], c) * g(a,b',n)
 * where g = dout (optionally * (gmask > 0)).  Accumulates (atomically) into a [ntaps][Cin][Cout] fp32
 * workspace and, when dbias != NULL, sum_pixels g into dbias[Cout].                               */
/* One segment of a multi-segment backward-weights launch (ABI 20): the tensors of ONE of several launches of the same layer whose
 * weight gradients are accumulated in a single launch — the K deferred ConvGRU cell updates of a data package (submodules.py:436-454:
 * the same three convolutions at every update).  Shapes, leading dimensions, channel counts and modes are the descriptor's. */
#define RAMNET_WGRAD_MAX_SEGMENTS 48
typedef struct ramnet_wgrad_seg {
    const float *x0, *x1, *xm, *dout, *gmask;
} ramnet_wgrad_seg;

typedef struct ramnet_wgrad_desc {
    const float *x0, *x1, *xm;
    int ld0, ld1, ldm, C0, C1, in_mode;
    int B, Hin, Win;
    int ntaps, stride;
    int8_t dy[25], dx[25];
    const float *dout, *gmask;      /* gradient wrt the layer's pre-activation output (+ ReLU mask)  */
    int ldg, ldgm;
    int Cout, Ho, Wo;
    float *dw;                      /* [ntaps][Cin][Cout] accumulation workspace ([16][Cin][Cout] for WINOGRAD) */
    float *dbias;                   /* [Cout] or NULL                                               */
    int algo;                       /* RAMNET_ALGO_DIRECT, or RAMNET_ALGO_WINOGRAD: dense 3x3 stride-1 taps in kh*3+kw order; dw then
                                     * accumulates the transformed-domain gradient dU, folded by ramnet_unpack_wgrad_wino();
                                     * or RAMNET_ALGO_WINOGRAD_2X4 (same launches, plain / concatenated / masked inputs): dw = dU, 24 positions in the blocked
                                     * layout of ramnet_wgrad_wino2x4_ws_floats(), of F(2x4,3x3) (24 multiplies per 8 outputs instead of 32), folded by ramnet_unpack_wgrad_wino2x4();
                                     * or RAMNET_ALGO_WINOGRAD24 (folded upsample-conv): x0 = [B][Hin = Ho+4][Win = Wo+4][C0] as for the
                                     * forward launch, dout / gmask = the full-resolution [B][HoG = 2*Ho][WoG = 2*Wo][Cout] tensors,
                                     * dw = dU [4 classes][25 positions][C0][Cout] (dW4 = G^T dU G per class); C0 % 32 == 0 and Cout % 64 == 0, or C0 % 64 == 0 and Cout % 32 == 0 */
    int gsy, gsx, goy, gox;         /* dout / gmask are read at pixel (oy*gsy + goy, ox*gsx + gox) of a [B, HoG, WoG] tensor      */
    int HoG, WoG;                   /* (all 0 = dense [B, Ho, Wo]; DIRECT only): one output parity of the folded upsample-conv  */
    int head_cin;                   /* RAMNET_ALGO_HEAD (dense 5x5 stride-1 taps, Cout <= 32): real input channels (1, 3 or 5); dw keeps
                                     * the DIRECT layout [25][C0][Cout]                                                          */
    int dw_slabs;                   /* RAMNET_ALGO_WINOGRAD: 0 = dw is ONE [16][Cin][Cout] workspace, the tile splits of a launch meet in it by
                                     * atomic adds; S > 0 = dw is [S][16][Cin][Cout] and dbias [S][Cout] (S >= 1, ramnet_wgrad_wino_slabs()):
                                     * split s owns slab s and joins it by plain read-modify-write — no atomics, and with the launches of
                                     * a layer serialised on one stream the gradient is bit-reproducible; ramnet_reduce_slabs() folds the
                                     * slabs into slab 0 before ramnet_unpack_wgrad_wino().  Other algorithms ignore it.           */
    int nseg;                       /* 0: one launch over x0 / x1 / xm / dout / gmask above.  1..RAMNET_WGRAD_MAX_SEGMENTS (RAMNET_ALGO_WINOGRAD_2X4
                                     * only — ramnet_wgrad_launch rejects nseg > 0 for every other algorithm): `segs` (HOST memory, copied into the kernel arguments by the launch) lists nseg tensor
                                     * sets of B images each; the launch reduces over all of them (one prologue / one join of the tile splits'
                                     * partial sums instead of nseg).  x0 ... gmask above must still be non-NULL where the mode uses them.   */
    const ramnet_wgrad_seg *segs;
} ramnet_wgrad_desc;

/* Process-wide A/B options (they replace the RAMNET_* environment knobs of rounds 1-3): "voxel_sorted" (1; 0 = row-band / atomic
 * voxelizer forms), "fold_pair" (1; 0 = 32-channel folded decoders as 64 tiles x 32 channels — changes the layout ramnet_pack_weight_fold_wino
 * writes: re-pack), "wgrad_blocks" (512: workgroups per launch of the DIRECT backward-weights kernel), "wgrad_wino_blocks" (384 since round 5 — with the splits dealt to the XCDs: 279.7 ms per step against 281.0 at 320 —, <= 384: the same for the
 * Winograd backward-weights kernel — measured with F(2x4) in place: 384 -> 215.9, 320 -> 210.7, 256 -> 209.7, 192 -> 193.6 samples/s).
 * "wgrad_wino_nf" (1; 2: 32-channel output blocks per workgroup of the Winograd backward-weights kernel — 1 = 32 x 32 channels, 168 registers, three
 * workgroups per CU: six ConvGRU launches 1.54 -> 1.31 ms on their own; 2 = 32 x 64 channels, two per CU: what a caller that co-schedules these
 * launches with a backward-data chain on another stream wants (training step 217 vs 211 samples/s) —
 * "wino_ksplit" (1 = the library's heuristic; 0 = ramnet_conv_splitk_floats answers 0: no launch splits its reduction; 2..16 = that many
 * splits for every launch whose epilogue can join partials: tuning runs).
 * "pred_si_cap" (256) / "pred_si_bwd_cap" (1024): workgroups per launch, over all segments, of ramnet_pred_sigmoid_si_fwd / _bwd (every
 * workgroup ends on per-segment atomics: tools/bench_pred_si.py has the sweep); ramnet_pred_si_scratch_doubles follows the option, so set it
 * before sizing the scratch.
 * ramnet_get_option: -1 if unknown.  */
int ramnet_set_option(const char *name, int value);
int ramnet_get_option(const char *name);
const char *ramnet_last_error(void);
int ramnet_abi_version(void);
/* Symbol (template arguments included, as rocprofv3 prints it without spaces) of the MFMA kernel the calling thread's most
 * recent ramnet_conv_launch / ramnet_conv_launch_multi / ramnet_wgrad_launch enqueued; "" before the first launch.  For profilers. */
const char *ramnet_last_kernel(void);
/* Stream `to` waits for the work enqueued on stream `from` so far (one reused event per calling thread and device; legal inside a stream
 * capture, where it joins `to` to the capture) — the fork of the backward-weights stream off the backward-data chain (ABI 20).        */
int ramnet_stream_fork(void *from, void *to);

/* ---- small fp32 GEMMs (border corrections of the folded upsample-conv and their gradients; csrc/gemm_skinny.hip) ----
 * trans_a = 0:  C[M][N] (=, or += when accumulate) A[M][K] * B[K][N]       (K % 4 == 0, rows of A 16-byte aligned)
 * trans_a = 1:  C[M][N] (=, +=)                    A[K][M]^T * B[K][N]    (A is [K][lda >= M]: the reduction runs over its rows)
 * All matrices row-major with leading dimensions lda / ldb / ldc (floats).  accumulate != 0 adds into C with atomics (and
 * splits the reduction over workgroups: backward pass); accumulate == 0 overwrites C, bit-reproducibly (one wave per block).
 * batch >= 1 independent products in one launch: entry i uses A + i*stride_a, B + i*stride_b, C + i*stride_c (floats).      */
int ramnet_gemm(const float *A, const float *B, float *C, int M, int N, int K, int lda, int ldb, int ldc, int trans_a,
                int accumulate, int batch, long stride_a, long stride_b, long stride_c, void *stream);
/* Two such batched products in ONE launch (the row and the column border of a decoder layer): same N, leading dimensions and mode,
 * their own operands, M, K (trans_a: the reduction length) and batch counts.                                              */
int ramnet_gemm2(const float *A, const float *B, float *C, int M, long stride_a, long stride_b, long stride_c, int batch, const float *A2,
                 const float *B2, float *C2, int M2, long stride_a2, long stride_b2, long stride_c2, int batch2, int N, int K, int K2, int lda,
                 int ldb, int ldc, int trans_a, int accumulate, void *stream);

/* ---- layout plumbing -------------------------------------------------------------------------- */
/* NCHW [B,C,H,W] -> NHWC [B,H,W,Cpad] zero-padded (model inputs: model.py:177,200 `.to(self.gpu)`). */
int ramnet_nchw_to_nhwc_pad(const float *src, float *dst, int B, int C, int H, int W, int Cpad, void *stream);
/* Full-frame mode: reflect-pad [B][C][H][W] to [Hc][Wc] with `top` rows above and `left` columns to the left (the rest below / right),
 * torch.nn.ReflectionPad2d semantics — what utils/inference_utils.py:287-314 (CropParameters) does in front of the network for sizes
 * that are not multiples of 2^num_encoders (260 x 346 -> 264 x 352).  nhwc = 1: written as the model's NHWC input with channels
 * zero-padded to Cpad (the repack of ramnet_nchw_to_nhwc_pad fused in); nhwc = 0: NCHW (Cpad ignored).                        */
int ramnet_reflect_pad(const float *src, float *dst, int B, int C, int H, int W, int Cpad, int top, int left, int Hc, int Wc, int nhwc,
                       void *stream);
/* Slabs the Winograd backward-weights launches of a Cin -> Cout layer use at most (ramnet_wgrad_desc.dw_slabs), and the ordered fold
 * slab 0 += slab 1 + ... + slab S-1 (n floats each, n % 4 == 0; slabs 1.. are zeroed) at the end of a backward pass.            */
int ramnet_wgrad_wino_slabs(int Cin, int Cout);
int ramnet_wgrad_wino2x4_slabs(int Cin, int Cout);      /* the same for RAMNET_ALGO_WINOGRAD_2X4 launches */
int ramnet_wgrad_dsplit_slabs(int Cin, int Cout);       /* the same for RAMNET_ALGO_DIRECT_SPLIT launches (ABI 23) */
/* Floats of ONE slab of the RAMNET_ALGO_DIRECT_SPLIT workspace (ABI 23): blocked [9 taps][ceil(Cin/32)][ceil(Cout/32)][64 lanes][16], element
 * (tap, c, n) where ramnet_wgrad_wino2x4_ws_floats() puts (position, c, n); ramnet_unpack_wgrad_dsplit adds it into the OIHW gradient
 * [Cout][Cin][3][3] of output channels n_off .. n_off + Cout - 1 of a [CinWs][CoutWs] workspace.                                    */
size_t ramnet_wgrad_dsplit_ws_floats(int Cin, int Cout);
int ramnet_unpack_wgrad_dsplit(const float *ws, float *grad, int Cout, int Cin, int CinWs, int CoutWs, int n_off, void *stream);
/* Floats of ONE slab of the RAMNET_ALGO_WINOGRAD_2X4 workspace (ABI 20): blocked [24 positions][ceil(Cin/32)][ceil(Cout/32)][64 lanes][16] —
 * element (position, c, n) at ((position * nCiB + c/32) * nCoB + n/32) * 1024 + ((n%32) + 32 * ((c%32 >> 2) & 1)) * 16 + (c%4) + 4 * (c%32 >> 3):
 * the accumulator registers of one lane of the 32 x 32 MFMA are 64 contiguous bytes (16-byte read-modify-write joins).                */
size_t ramnet_wgrad_wino2x4_ws_floats(int Cin, int Cout);
int ramnet_reduce_slabs(float *ws, int slabs, size_t n, void *stream);
/* Number of floats of a packed weight (forward: reduce over Cin; transposed: reduce over Cout).    */
size_t ramnet_packed_weight_elems(int Cout, int Cin, int KH, int KW, int transposed, int gates);
/* Winograd F(2x2,3x3) weights U = G g G^T of a 3x3 conv in the lane order of the kernel's B operand
 * ([Cin/8][Cout/64][8 position pairs][64][4][2][2], zero padded).  transposed=1 packs the backward-data operator
 * (flipped taps, reduce over O); gates=4 (forward only) groups the ConvLSTM gates of 16 hidden channels per block.  */
size_t ramnet_packed_weight_elems_wino(int Cout, int Cin, int transposed, int gates);
int ramnet_pack_weight_wino(const float *w_oihw, float *wp, int Cout, int Cin, int transposed, int gates, void *stream);
/* Winograd F(2x4,3x3) weights U = G2 g G4^T (G2 of F(2,3), G4 of F(4,3)) in the lane order of conv_wino_r6_kernel's B operand:
 * [Cin/8][Cout/64][row 4][column 6][n-block 2][lane 64][4], zero padded; transposed=1: the backward-data operator.            */
size_t ramnet_packed_weight_elems_wino2x4(int Cout, int Cin, int transposed);
int ramnet_pack_weight_wino2x4(const float *w_oihw, float *wp, int Cout, int Cin, int transposed, void *stream);
/* 1 when a launch that qualifies for RAMNET_ALGO_WINOGRAD (d->algo set so, every other field final) runs faster as
 * RAMNET_ALGO_WINOGRAD_2X4 — the caller then sets d->algo and d->w (ramnet_pack_weight_wino2x4) accordingly; else 0.              */
int ramnet_conv_wino_variant(const ramnet_conv_desc *d, int force);   /* force: skip the size heuristics (tests) */
/* 1 when a launch that ramnet_conv_wino_variant() accepts can run as RAMNET_ALGO_WINOGRAD_2X4_SPLIT (16-channel chunks: the boundary of a
 * concatenation at a multiple of 16 channels, space-to-depth views of >= 16 channels); the caller then sets d->algo and d->w
 * (ramnet_pack_weight_wino2x4_split: three bf16 planes, 6 bytes per Winograd-domain weight; the size is returned in 4-byte units).           */
int ramnet_conv_wino_split_ok(const ramnet_conv_desc *d, int force);
size_t ramnet_packed_weight_elems_wino2x4_split(int Cout, int Cin, int transposed);
int ramnet_pack_weight_wino2x4_split(const float *w_oihw, float *wp, int Cout, int Cin, int transposed, void *stream);
/* Floats of split-reduction workspace a RAMNET_ALGO_WINOGRAD launch of this descriptor (every other field final) would use, 0 when it
 * would not split (enough workgroups, epilogue kinds that do not join partials, option "wino_ksplit" = 0).                              */
size_t ramnet_conv_splitk_floats(const ramnet_conv_desc *d);
/* Process-wide tuning of the F(2x4,3x3) selection (tests, A/B runs): min_wgs = 64-channel x 256-pixel blocks of output a launch must
 * have (default 150; < 0 keeps the current value).                                                                              */
int ramnet_wino2x4_config(int min_wgs);
/* Folded upsample-conv (RAMNET_ALGO_WINOGRAD24): OIHW 5x5 weights of an UpsampleConvLayer (submodules.py:69-97) -> Winograd-domain
 * weights of the four 4x4 parity filters in the kernel's layout (see ramnet_algo above); 100*Cout*Cin floats.
 * ramnet_fold_wino_supported: Cout % 32 == 0 and an even number of input-channel chunks (Cin % 32 == 0, or Cin % 16 == 0 with
 * 32-channel workgroups).  ABI 13: a layer with Cout == 32 and Cin % 32 == 0 is packed in the PAIR layout — class = row parity
 * py (2 classes), the workgroup's 64 columns = (column parity px, channel): [py][Cin/16][25 positions][4 column groups of 16]
 * [64 lanes][4] — and launched with both column parities in one workgroup sharing the transformed input (RAMNET_FOLD_PAIR=0 in
 * the environment of BOTH packer and launcher restores the [4 classes] x 32-channel layout).                    */
int ramnet_fold_wino_supported(int Cout, int Cin);
size_t ramnet_packed_weight_elems_fold_wino(int Cout, int Cin);
int ramnet_pack_weight_fold_wino(const float *w_oihw, float *wp, int Cout, int Cin, void *stream);
/* The rest of a folded decoder's weight algebra (once per optimizer step and layer; torch.einsum chains until round 3):
 * _dgrad: Winograd weights of the flipped parity filters for the backward-data launch (RAMNET_IN_PARITY4; Cin % 64 == 0), 100*Cout*Cin floats;
 * border weights: rows / cols [2 sides][5*Cin][2*Cout] = minus the taps the zero padding removes at the image border, and their
 *   transposes [2][2*Cout][5*Cin] (backward-data operands);
 * fold_unpack_wgrad: grad [Cout][Cin][5][5] += the fold of the pass's workspaces — w4 [64 = (py,px,ty,tx)][CinWs][Cout] (direct parity launches),
 *   dU [4][25][CinWs][Cout] (Winograd-domain launches; may be NULL), wr / wc [2][5*Cin][2*Cout] (border-GEMM gradients) — which it zeroes.   */
int ramnet_pack_weight_fold_wino_dgrad(const float *w_oihw, float *wp, int Cout, int Cin, void *stream);
int ramnet_pack_border_weights(const float *w_oihw, float *rows, float *cols, float *rows_t, float *cols_t, int Cout, int Cin, void *stream);
int ramnet_fold_unpack_wgrad(float *w4, float *dU, float *wr, float *wc, float *grad, int Cout, int Cin, int CinWs, void *stream);
/* Head layers (RAMNET_ALGO_HEAD): OIHW [Cout<=32][Cin][5][5] -> [25*Cin rounded up to even][32], row = tap*Cin + channel.
 * ramnet_head_supported: does the head kernel serve this channel pair (Cin in {1,3,5,10}, Cout <= 32)?           */
size_t ramnet_packed_weight_elems_head(int Cin);
int ramnet_head_supported(int Cin, int Cout);
int ramnet_pack_weight_head(const float *w_oihw, float *wp, int Cout, int Cin, void *stream);
/* OIHW -> kernel layout [tap][chunk][n][16].  transposed=1 packs the backward-data operator
 * (reduce over O, produce I).  gates=4 interleaves ConvLSTM gate blocks so that one wave owns
 * i,f,o,g of a channel (forward only).  CinValid rows beyond Cin are zero (padded inputs).        */
int ramnet_pack_weight(const float *w_oihw, float *wp, int Cout, int Cin, int KH, int KW,
                       int transposed, int gates, void *stream);
/* [tap][CinWs][CoutWs] gradient workspace -> OIHW: grad_oihw[n][c][tap] += ws[tap][c][n_off + n].
 * (CoutWs/n_off: fused launches such as the GRU's update|reset gates share one workspace.)        */
int ramnet_unpack_wgrad(const float *ws, float *grad_oihw, int Cout, int Cin, int CinWs, int CoutWs, int n_off,
                        int KH, int KW, void *stream);

/* Winograd backward-weights workspace [16][CinWs][CoutWs] (dU) -> OIHW 3x3: grad += G^T dU G.                 */
int ramnet_unpack_wgrad_wino(const float *ws, float *grad_oihw, int Cout, int Cin, int CinWs, int CoutWs, int n_off,
                             void *stream);
/* F(2x4,3x3) backward-weights workspace (blocked: ramnet_wgrad_wino2x4_ws_floats(CinWs, CoutWs)) -> OIHW 3x3: grad += G_r^T dU G_c.  */
int ramnet_unpack_wgrad_wino2x4(const float *ws, float *grad_oihw, int Cout, int Cin, int CinWs, int CoutWs, int n_off,
                                void *stream);

/* ---- the two MFMA kernels ----------------------------------------------------------------------- */
int ramnet_conv_launch(const ramnet_conv_desc *d, void *stream);    /* forward and backward-data   */
/* n <= 4 descriptors that differ only in tap list and output sub-grid (ntaps, dy, dx, wtap, Ho, Wo, ooy, oox): the four
 * output-parity classes of a stride-2 backward-data / transposed convolution, executed as ONE launch.            */
int ramnet_conv_launch_multi(const ramnet_conv_desc *descs, int n, void *stream);
int ramnet_wgrad_launch(const ramnet_wgrad_desc *d, void *stream);  /* backward-weights (+bias)    */

/* ---- HBM-bound point-wise / reduction kernels ------------------------------------------------- */
/* pred = sigmoid(conv1x1(x) + b): statenet.py:116-117,313.  x NHWC [npix, C], y [npix].            */
int ramnet_pred_sigmoid_fwd(const float *x, int ldx, int C, const float *w, const float *b, float *y,
                            size_t npix, void *stream);
/* backward of the above: dx[npix,C] = dz*w, dw[C] += sum dz*x, db += sum dz, dz = dy*y*(1-y).      */
int ramnet_pred_sigmoid_bwd(const float *x, int ldx, int C, const float *w, const float *y, const float *dy,
                            float *dx, int lddx, float *dw, float *db, size_t npix, void *stream);
/* The prediction layer TOGETHER with the scale-invariant loss of the maps it supervises (ABI 20; model/loss.py:6-9 on the output of
 * statenet.py:313): the batch is nseg (<= RAMNET_PRED_SI_MAX_SEGMENTS) segments of seg_pix pixels, targets[i] (HOST array of device pointers)
 * = the seg_pix target values of segment i (NaN = invalid).  fwd: y as ramnet_pred_sigmoid_fwd, stats[i][0..3] = (sum d, sum d^2, n, 0) in
 * double, loss[i] = weight * (mean d^2 - lambda * mean(d)^2); scratch = ramnet_pred_si_scratch_doubles(seg_pix, nseg) doubles, ZERO before the
 * first use (the kernel leaves the tickets at zero; the partial sums are joined in a fixed order: bit-reproducible).  bwd: the gradient of
 * sum_i gscale[i] * loss[i] (+ dy . y when dy != NULL) w.r.t. x, w and b — ramnet_si_loss_bwd and ramnet_pred_sigmoid_bwd in one pass.  Its `scratch`
 * (ABI 24) = the buffer the forward launch used (same size, the options unchanged in between; its tail — the tickets — still zero): the workgroups' partial sums
 * of dw / db are joined in a fixed order — bit-reproducible — by the last workgroup to arrive; NULL = fp32 atomics (arrival order = the last digits).
 * mask_x != 0 (ABI 24): dx = dz w (x > 0) — x is the output of a ReLU layer whose only consumer this is (statenet.py:305-313: the last decoder), and its
 * backward receives the gradient already masked. */
#define RAMNET_PRED_SI_MAX_SEGMENTS 8
size_t ramnet_pred_si_scratch_doubles(size_t seg_pix, int nseg);
int ramnet_pred_sigmoid_si_fwd(const float *x, int ldx, int C, const float *w, const float *b, float *y, size_t seg_pix, int nseg,
                               const float *const *targets, float weight, float lambda, double *scratch, double *stats, float *loss,
                               void *stream);
int ramnet_pred_sigmoid_si_bwd(const float *x, int ldx, int C, const float *w, const float *y, const float *dy, size_t seg_pix, int nseg,
                               const float *const *targets, const double *stats, const float *gscale, float weight, float lambda,
                               float *dx, int lddx, float *dw, float *db, double *scratch, int mask_x, void *stream);
/* The same layer WITHOUT the sigmoid (a normalisation follows: `norm` BN / IN, submodules.py:29-33): z = conv1x1(x) [+ b] (b may be
 * NULL) and its backward dx = dz*w, dw += sum dz*x, db += sum dz (db may be NULL).                                       */
int ramnet_pred_linear_fwd(const float *x, int ldx, int C, const float *w, const float *b, float *z, size_t npix, void *stream);
int ramnet_pred_linear_bwd(const float *x, int ldx, int C, const float *w, const float *dz, float *dx, int lddx, float *dw, float *db,
                           size_t npix, void *stream);
/* dx = dy * (y > 0) */
int ramnet_relu_bwd(const float *dy, const float *y, float *dx, size_t n, void *stream);
/* Folded upsample-conv (UpsampleConvLayer forward as four 4x4 parity convolutions of the LOW-resolution input):
 * out[b][i+2][j+2][c] = (x + skip)[clamp(i)][clamp(j)] for i in [-2, H+1], j in [-2, W+1]  (skip may be NULL).       */
int ramnet_pad2_sum(const float *x, const float *skip, float *out, int B, int H, int W, int C, void *stream);
/* Border lines of u = up2x(x + skip) unrolled along the 5 filter taps, the A operands of the border-correction GEMMs:
 * rows [2][B*2W][5][C] (top / bottom image row, columns clamped), cols [2][B*2H][5][C] (left / right image column, rows outside
 * the image zero).                                                                                                      */
int ramnet_up2x_border_im2col(const float *x, const float *skip, float *rows, float *cols, int B, int H, int W, int C, void *stream);
/* ramnet_pad2_sum and ramnet_up2x_border_im2col in one launch (the forward of a folded decoder layer needs both). */
int ramnet_pad2_sum_im2col(const float *x, const float *skip, float *out, float *rows, float *cols, int B, int H, int W, int C, void *stream);
/* Backward-data of the folded upsample-conv: adjoint of ramnet_pad2_sum (dx[B][H][W][C] = dxpad summed over the padded pixels that
 * copy each pixel; also the skip gradient) and adjoint of ramnet_up2x_border_im2col (gradients of the unrolled border lines
 * rows [2][B*2W][5][C] / cols [2][B*2H][5][C] gathered by the border pixels of dx through their bilinear weights, +=).                                */
int ramnet_unpad2_fold(const float *dxpad, float *dx, int B, int H, int W, int C, void *stream);
int ramnet_up2x_border_col2im(const float *rows, const float *cols, float *dx, int B, int H, int W, int C, void *stream);
/* Outermost two rows / columns of dy (* (mask > 0) when mask != NULL) of a [B, H2, W2, C] tensor, in the layout of the
 * border-correction GEMMs of the folded upsample-conv: rows [2][B*W2][2][C], cols [2][B*H2][2][C].                       */
int ramnet_frame_gather(const float *dy, const float *mask, float *rows, float *cols, int B, int H2, int W2, int C, void *stream);
/* Space-to-depth by 2: [B,H,W,C] -> [B,H/2,W/2,4C], channel (a*2 + c)*C + ch = pixel parity (a, c); inverse=1: the way back.
 * A stride-2 5x5 convolution of x is a stride-1 3x3 convolution of it (9 taps x 4C channels, 11 of the 36 slices zero).   */
int ramnet_space_to_depth2(const float *x, float *out, int B, int H, int W, int C, int inverse, void *stream);
/* Adjoint of the bilinear x2 upsample: dup [B,2H,2W,C] -> dx [B,H,W,C] (backward of submodules.py:88). */
int ramnet_upsample2x_bwd(const float *dup, float *dx, int B, int H, int W, int C, void *stream);
/* ConvGRU backward, point-wise parts (derivation in DESIGN.md):
 *  stage A: from dh' and saved u,o,h: dpo = dh'*u*(1-o^2) ; dpu = dh'*(o-h)*u*(1-u) ; dh = dh'*(1-u)
 *  stage B: from d(h*r) (dgrad of the candidate conv) : dpr = dhr*h*r*(1-r) ; dh += dhr*r            */
int ramnet_gru_bwd_a(const float *dhn, const float *ur, const float *o, const float *h, float *dpo,
                     float *dpur, float *dh, size_t npix, int C, int ld_dhn /* floats per pixel of dhn (>= C: channel slices) */,
                     void *stream);
int ramnet_gru_bwd_a2(const float *dhn, const float *ur, const float *o, const float *h, float *dpo,
                      float *dpur, float *dh, size_t npix, int C, int ld_dhn, int ld_dh /* floats per pixel of dh (ABI 18) */,
                      void *stream);
int ramnet_gru_bwd_b(const float *dxhr, const float *ur, const float *h, float *dpur, float *dh,
                     size_t npix, int C, void *stream);
/* ConvLSTM backward point-wise: gates [npix,4C] (activated i,f,o,g), c_prev, c_new, dh', dc' ->
 * dgates_pre [npix,4C], dc_prev.                                                                   */
int ramnet_lstm_bwd(const float *gates, const float *cprev, const float *cnew, const float *dhn,
                    const float *dcn, float *dpre, float *dcprev, size_t npix, int C, void *stream);
/* db[C] += sum_pixels dy * (mask > 0): bias gradient of the transposed-conv decoder (submodules.py:38-66). */
int ramnet_bias_grad(const float *dy, const float *mask, float *db, size_t npix, int C, void *stream);
/* ---- BatchNorm / InstanceNorm of `norm: "BN" | "IN"` layers: submodules.py:13-24, 29-30, 52-62, 82-94, 188-193, 203-210 ------
 * NHWC tensors [groups][npix][C] (row stride ld* >= C); groups = 1 for BatchNorm (npix = B*H*W), B for InstanceNorm (npix = H*W).
 * ramnet_norm_partial: part[((g * nslab + s) * C + c) * 2 + {0, 1}] = (sum a', sum a'*b) over the pixels p = s (mod nslab) ... of
 *   group g in fp64, a' = a when y == NULL (forward: a = b = x gives mean and variance) or a * act'(y) (backward: a = dy; act:
 *   0 none, 1 ReLU, 2 sigmoid, y = the layer's activated output); nslab from ramnet_norm_slabs; the caller sums over s.
 * ramnet_norm_finalize: mean, rstd [groups][C] (fp64), scale = gamma * rstd, shift = beta - mean * scale (fp32; gamma / beta may be
 *   NULL = 1 / 0) from the partial sums — or, use_running != 0 (eval mode, groups = 1), from the running buffers.  update_running:
 *   running = (1 - momentum) * running + momentum * mean over the groups of (mean, UNBIASED variance), as torch's BatchNorm2d /
 *   InstanceNorm2d(track_running_stats=True) in training mode; num_batches_tracked (int64, may be NULL) += 1.
 * ramnet_norm_finalize_bwd: from the backward's partial sums: c1, c2, c3 [groups][C] of ramnet_norm_bwd (batch_stats = the forward
 *   normalised with the statistics of x itself; 0 = running statistics: c2 = c3 = 0), dgamma / dbeta [C] (=, may be NULL).
 * ramnet_norm_apply:   out = act(x * scale[g][c] + shift[g][c] [+ res]).
 * ramnet_norm_bwd:     dx = c1[g][c] * dy' + c2[g][c] * x + c3[g][c], dy' = dy * act'(y); dres (optional) = dy'.            */
int ramnet_norm_slabs(int groups, long npix, int C);
int ramnet_norm_partial(const float *a, int lda, const float *y, int ldy, int act, const float *b, int ldb, int groups, long npix,
                        int C, int nslab, double *part, void *stream);
int ramnet_norm_finalize(const double *part, int groups, int nslab, int C, long npix, double eps, const float *gamma, const float *beta,
                         float *running_mean, float *running_var, double momentum, int update_running, int use_running,
                         long long *num_batches_tracked, double *mean, double *rstd, float *scale, float *shift, void *stream);
int ramnet_norm_finalize_bwd(const double *part, int groups, int nslab, int C, long npix, const double *mean, const double *rstd,
                             const float *gamma, int batch_stats, float *c1, float *c2, float *c3, float *dgamma, float *dbeta, void *stream);
int ramnet_norm_apply(const float *x, int ldx, const float *scale, const float *shift, const float *res, int ldr, int act, float *out,
                      int ldo, int groups, long npix, int C, void *stream);
int ramnet_norm_bwd(const float *dy, int lddy, const float *y, int ldy, int act, const float *x, int ldx, const float *c1, const float *c2,
                    const float *c3, float *dx, int lddx, float *dres, int lddres, int groups, long npix, int C, void *stream);
/* y = a + b (gradient fan-in) */
int ramnet_add(const float *a, const float *b, float *y, size_t n, void *stream);
/* y [npix][Ca + Cb] = channel concatenation of a [npix][lda >= Ca] and b [npix][ldb >= Cb] (UNet skip_type 'concat',
 * unet.py:11-13; channel counts multiples of 4).  ramnet_split2 is its gradient: the two channel slices of y [npix][ldy], dense. */
int ramnet_concat2(const float *a, int lda, int Ca, const float *b, int ldb, int Cb, float *y, size_t npix, void *stream);
/* out [n * npix][C] (dense) = the n tensors parts[k] [npix][ld >= C] one behind the other along the BATCH axis, + base [n * npix][C]
 * when base != NULL: the gradient of a feature that went through a layer chain at batch n x B and was consumed slice by slice (the n
 * state updates of a package) and, as a whole, by the next layer of the chain (ABI 19; `parts` = host array of n <= 8 device pointers). */
int ramnet_cat_batch_add(const float *const *parts, int n, size_t npix, int C, int ld, const float *base, float *out, void *stream);
/* The same with the ReLU mask of the feature itself applied to the sum (ABI 24): out = (cat(parts) [+ base]) * (mask > 0), mask [n * npix][C] dense = the
 * feature (the output of a ConvLayer with ReLU, submodules.py:26-35).  The layer's backward-data / backward-weights launches then read the gradient as
 * ONE plain operand instead of (gradient, output) pairs through RAMNET_IN_RELUMASK / gmask: the same values, so bit-identical gradients. */
int ramnet_cat_batch_add_masked(const float *const *parts, int n, size_t npix, int C, int ld, const float *base, const float *mask, float *out,
                                void *stream);
int ramnet_split2(const float *y, int ldy, int Ca, int Cb, float *a, float *b, size_t npix, void *stream);

/* ---- scale-invariant loss: model/loss.py:6-9 -------------------------------------------------- */
/* stats[0..2] = (sum d, sum d^2, count) over non-NaN d = pred - target; loss = w*(S2/n - lambda*(S1/n)^2).  `stats` holds FOUR
 * doubles ([3] is scratch of the reduction; the backward reads [0..2]).  One launch: every workgroup stores its partial sums to a
 * scratch the library owns per (device, stream), the last one to arrive adds them in a fixed order (bit-reproducible loss); inside a
 * stream capture that finds no scratch yet: zero-fill + atomics on `stats`.                                              */
int ramnet_si_loss_fwd(const float *pred, const float *target, size_t n, float weight, float lambda,
                       double *stats, float *loss, void *stream);
/* loss = w*(S2/n - lambda*(S1/n)^2) from GIVEN statistics stats[0..2] — the exact data-parallel form: every rank computes the
 * sums of its maps (ramnet_si_loss_fwd), the sums are all-reduced, and loss / gradient follow from the global ones, so that
 * mean(d)^2 is taken over the whole batch as model/loss.py:9 does (ramnet_si_loss_bwd reads the same stats).                    */
int ramnet_si_loss_from_stats(const double *stats, float weight, float lambda, float *loss, void *stream);
/* dpred = gscale * w * (2 d/n - 2 lambda mean/n) on valid pixels, 0 elsewhere (gscale: device scalar). */
int ramnet_si_loss_bwd(const float *pred, const float *target, size_t n, float weight, float lambda,
                       const double *stats, const float *gscale, float *dpred, void *stream);

/* scale_invariant_log_loss (model/loss.py:12-15): the same statistic on d = log(pred) - log(target) (no weight argument in the reference).
 * stats: 4 doubles as above (zeroed here).                                                                             */
int ramnet_si_log_loss_fwd(const float *pred, const float *target, size_t n, float lambda, double *stats, float *loss, void *stream);
int ramnet_si_log_loss_bwd(const float *pred, const float *target, size_t n, float lambda, const double *stats, const float *gscale,
                           float *dpred, void *stream);
/* mse_loss (model/loss.py:18-19) as the trainer's extra term uses it (lstm_trainer.py:169-185): mean squared error over the non-NaN
 * TARGET entries of [B][H][W] maps; half = 1: both maps first go through F.interpolate(scale_factor=0.5, 'bilinear', align_corners=False)
 * (2 x 2 block means; a NaN in a target block masks the cell; odd trailing rows / columns are dropped).  stats: 4 doubles
 * ([0] = sum d^2, [1] = count, zeroed here); dpred [B][H][W] = gscale * d loss / d pred (every element written).            */
int ramnet_mse_loss_fwd(const float *pred, const float *target, int B, int H, int W, int half, double *stats, float *loss, void *stream);
int ramnet_mse_loss_bwd(const float *pred, const float *target, int B, int H, int W, int half, const double *stats, const float *gscale,
                        float *dpred, void *stream);

/* ---- depth post-processing + error sums: evaluation.py:74-96, :201-241; model/metric.py:8-33 --------------
 * pred/target: normalised log depth; mask = nan_to_num(metric target) < cutoff (evaluation.py:367).  out11 (zeroed here):
 * n (non-NaN targets in the mask), n_mask, and over those n pixels: sum |d|/(t+1e-6), sum d^2/(t^2+1e-6), sum d^2,
 * sum ld^2, sum |ld| (ld = log(t+1e-5) - log(p+1e-5)), sum |d|, count(ratio <= 1.25), (<= 1.25^2), (<= 1.25^3) with
 * ratio = max(t/(p+1e-5), p/(t+1e-5)).                                                                          */
int ramnet_depth_metrics(const float *pred, const float *target, size_t n, float clip_distance, float reg_factor,
                         float cutoff, double *out11, void *stream);

/* ---- multi-scale gradient loss: model/loss.py:22-70 (kornia Sobel restated; PARITY UNPINNED) -------
 * ws / dws: float workspaces of ramnet_msg_workspace_elems() elements; stats: 2*num_scales doubles.          */
size_t ramnet_msg_workspace_elems(int B, int H, int W, int num_scales);
int ramnet_msg_loss_fwd(const float *pred, const float *target, int B, int H, int W, int num_scales, float *ws,
                        double *stats, float *loss, void *stream);
int ramnet_msg_loss_bwd(const float *ws, const double *stats, const float *gscale, int B, int H, int W, int num_scales,
                        float *dws, float *dpred, void *stream);

/* ---- event -> voxel grid: utils/event_tensor_utils.py:120-187, :52-66 ----------------------- */
/* events: [N,4] float64 rows (t,x,y,p) sorted by t, on device (16-byte aligned).  grid [bins,H,W] fp32: every cell is written. */
int ramnet_voxelize(const double *events, size_t n_events, int bins, int W, int H, float *grid, void *stream);
/* same index arithmetic, but emits the int64 flat indices (or -1) for bit-exactness tests.        */
int ramnet_voxel_indices(const double *events, size_t n_events, int bins, int W, int H,
                         long long *idx_left, long long *idx_right, void *stream);
/* zero-mean/unit-std over non-zero entries, in place; scratch: 3 doubles.                          */
int ramnet_normalize_nonzero(float *grid, size_t n, double *scratch, void *stream);
/* Batched forms (one launch for the B x K grids of a batch of packages): event lists concatenated in `events`, list g =
 * rows offsets[g] .. offsets[g+1] (device int64 [n_grids+1]; each list sorted by t, normalised by ITS first / last stamp),
 * max_events = longest list; grids [n_grids][bins][H][W] (zeroed here).  normalize: n = bins*H*W (multiple of 4) per grid,
 * scratch = 3*n_grids doubles.  Same arithmetic per grid as the single-grid entry points.  Launches of >= 16 grids resolve the
 * votes in LDS row bands (no global atomics, no zero-fill pass); smaller ones scatter with global fp32 atomics.             */
int ramnet_voxelize_batch(const double *events, const long long *offsets, int n_grids, size_t max_events, int bins, int W, int H,
                          float *grids, void *stream);
int ramnet_normalize_nonzero_batch(float *grids, int n_grids, size_t n, double *scratch, void *stream);

#ifdef __cplusplus
}
#endif
#endif
