// Small fp32 GEMMs of the path — the border corrections of the folded upsample-conv ([border pixels x 5*Cin] x
// [5*Cin x 2*Cout], their backward-data and backward-weights forms) — on v_mfma_f32_32x32x2_f32, without LDS and without
// barriers: every WAVE owns one 32 x 32 block of C and reads its operands from global memory directly in MFMA lane layout
// (the matrices are a few MB and were written by the previous kernel: L2 / Infinity-Cache hits).
//
//   NN:  C[M][N]  = A[M][K] * B[K][N]          lane (row m = lane & 31, half = lane >> 5) loads A[m][k0 + 16*half .. +15] as four
//                                              16-byte loads (the MFMA's K index 0 / 1 = the two halves: a row's 128-byte line
//                                              is consumed whole), B[k][n = lane & 31] as 128-byte-coalesced 4-byte loads
//   TN:  C[K][N] += A[R][K]^T * B[R][N]        (weight gradient: reduction over the R pixel rows, split over gridDim.z and
//                                              joined by atomics) — both operands coalesced along their channel index
// Loads run one K step (32 values, 1024 MFMA cycles) ahead of the MFMAs.  The accumulating forms (backward pass) split the
// reduction over workgroups until ~8 waves per SIMD are in flight (partial sums meet by atomics): the operand loads are
// latency-bound, occupancy is what hides them.  The plain product (forward pass) splits the reduction over the four waves of the
// block's workgroup, which join through LDS in a fixed order: bit-reproducible.
#include <stdlib.h>
#include "common.hpp"

namespace ramnet {

// A launch may carry TWO problems of the same N, leading dimensions and mode (the row and the column border of a decoder layer differ in
// M — forward / backward-data — or in the reduction length K — weight gradient): batch entries [0, nb1) belong to (A, B, C, M), the rest to the second set.
struct Gemm2nd {
    const float *A, *B;
    float *C;
    int M, K, nb1;
    long sa, sb, sc;
};

template <bool TA>
__global__ void __launch_bounds__(256) gemm32_kernel(const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C,
                                                     int M, int N, int K, int lda, int ldb, int ldc, int ksplit, int atomic,
                                                     int intra, long sa, long sb, long sc, const Gemm2nd two) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    // accumulating forms: the 4 waves of a workgroup own 4 neighbouring blocks (they share the A rows through the L1) and the
    // reduction is split over gridDim.z; plain product (intra): the 4 waves split the reduction of ONE block and join through LDS
    // in a fixed order — four times the waves in flight, still bit-reproducible
    const int nb = intra ? blockIdx.x : blockIdx.x * 4 + wave, mb = blockIdx.y;
    if (nb * 32 >= N) return;                                           // (uniform over the workgroup when intra)
    int bi = blockIdx.z / ksplit;                                        // batch entry (independent products)
    const int ks = intra ? wave : blockIdx.z - bi * ksplit;             // reduction slice
    if (bi >= two.nb1) {
        bi -= two.nb1;
        A = two.A, B = two.B, C = two.C, M = two.M, K = two.K, sa = two.sa, sb = two.sb, sc = two.sc;
    }
    if (mb * 32 >= M) return;                                           // (uniform over the workgroup: the other problem is taller)
    A += bi * sa, B += bi * sb, C += bi * sc;
    const int m = mb * 32 + l31, n = nb * 32 + l31;
    const int nsl = intra ? 4 : ksplit;
    const int kper = ((K + nsl - 1) / nsl + 31) / 32 * 32;
    const int kbeg = ks * kper, kend = min(K, kbeg + kper);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const bool mok = m < M, nok = n < N;
    // Operands through buffer loads: the per-lane byte offsets of a K step are computed once (rows / columns past M, N get an
    // out-of-range offset and read as 0), the step moves the descriptor, whose extent ends with this slice of the reduction —
    // so the loads of the main loop carry no mask, no branch and no per-lane address arithmetic.
    // one K step = 32 values: half 0 of the wave takes k0 .. k0+15, half 1 k0+16 .. k0+31 (MFMA i pairs k0+i with k0+16+i), so a
    // row of A is read as one full 128-byte line by its two lanes (4 x 16 bytes each) — no reliance on the L1 keeping it
    unsigned ao[TA ? 16 : 4], bo[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        bo[j] = nok ? (unsigned)(((16 * kh + j) * ldb + n) * 4) : WOOB;
        if (TA) ao[j] = mok ? (unsigned)(((16 * kh + j) * lda + m) * 4) : WOOB;
    }
    if (!TA) {
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) ao[qd] = mok ? (unsigned)((m * lda + 16 * kh + 4 * qd) * 4) : WOOB;
    }
    auto load = [&](int k0, float (&a)[16], float (&b)[16]) {
        // descriptors from row k0 to the end of the slice (the range check looks at the per-lane offset only): rows past kend read 0
        const auto rb = wino_rsrc(B + (size_t)k0 * ldb, (unsigned)min((size_t)(kend - k0) * ldb * 4, (size_t)WOOB));
        const auto ra = TA ? wino_rsrc(A + (size_t)k0 * lda, (unsigned)min((size_t)(kend - k0) * lda * 4, (size_t)WOOB)) : wino_rsrc(A + k0, WOOB);
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            if (!TA) {      // K % 4 == 0 and 16-byte aligned rows (host): a quad never straddles kend
                const unsigned o = k0 + 16 * kh + 4 * qd < kend ? ao[qd] : WOOB;
                const float4 v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ra, (int)o, 0, 0));
                a[4 * qd] = v.x, a[4 * qd + 1] = v.y, a[4 * qd + 2] = v.z, a[4 * qd + 3] = v.w;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (TA) a[4 * qd + j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, (int)ao[4 * qd + j], 0, 0));
                b[4 * qd + j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, (int)bo[4 * qd + j], 0, 0));
            }
        }
    };
    auto mm = [&](const float (&a)[16], const float (&b)[16]) {
#pragma unroll
        for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc, 0, 0, 0);
    };
    float a0[16], b0[16], a1[16], b1[16];       // loads run one step (1024 MFMA cycles) ahead
    if (kbeg < kend) load(kbeg, a0, b0);
    for (int k0 = kbeg; k0 < kend; k0 += 64) {
        if (k0 + 32 < kend) load(k0 + 32, a1, b1);
        mm(a0, b0);
        if (k0 + 32 >= kend) break;
        if (k0 + 64 < kend) load(k0 + 64, a0, b0);
        mm(a1, b1);
    }
    if (intra) {
        __shared__ float red[3][16][64];
        if (wave > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[wave - 1][r][lane] = acc[r];
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = ((acc[r] + red[0][r][lane]) + red[1][r][lane]) + red[2][r][lane];
    }
    if (!nok) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (row >= M) continue;
        if (atomic) atomicAdd(C + (size_t)row * ldc + n, acc[r]);
        else C[(size_t)row * ldc + n] = acc[r];
    }
}

}  // namespace ramnet

using namespace ramnet;

static int launch_gemm(const float *A, const float *B, float *C, int M, int N, int K, int lda, int ldb, int ldc, int trans_a, int accumulate,
                       int batch, long stride_a, long stride_b, long stride_c, const float *A2, const float *B2, float *C2, int M2, int K2, int batch2,
                       long stride_a2, long stride_b2, long stride_c2, void *stream) {
    const int Mmax = M > M2 ? M : M2, Kmax = K > K2 ? K : K2;
    RAMNET_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0 && ldb >= N && ldc >= N && batch >= 1 && batch2 >= 0);
    if (batch2 > 0) RAMNET_CHECK_ARG(A2 && B2 && C2 && M2 > 0 && K2 > 0 && ((uintptr_t)A2 & 15) == 0 && stride_a2 % 4 == 0 && (trans_a || K2 % 4 == 0));
    if (trans_a) RAMNET_CHECK_ARG(lda >= Mmax);
    else RAMNET_CHECK_ARG(lda >= Kmax && K % 4 == 0 && lda % 4 == 0 && ((uintptr_t)A & 15) == 0 && stride_a % 4 == 0);
    // operands are addressed through 32-bit per-lane byte offsets and buffer extents clamped to WOOB (like the conv launchers)
    RAMNET_CHECK_ARG((unsigned long long)(trans_a ? Kmax : Mmax) * lda * 4ull < WOOB && (unsigned long long)Kmax * ldb * 4ull < WOOB &&
                     (unsigned long long)Mmax * ldc * 4ull < WOOB);
    // accumulate (backward pass): the operand loads are latency-bound and occupancy is what hides them, so the reduction is split
    // over gridDim.z until a few thousand waves are in flight; partial sums meet by atomics in C.  A plain product (forward pass)
    // splits it over the waves of one workgroup per block instead (fixed-order LDS join): bit-reproducible, as every forward kernel.
    const int nb = batch + batch2;
    const int blocks = nb * cdiv(Mmax, 32) * cdiv(N, 32);
    int ksplit = 1;
    if (accumulate)      // (splitting less — 512 ... 2048 blocks — measured the same training step: 198.2 - 199.2 samples/s)
        while (blocks * ksplit < 8192 && Kmax / (ksplit * 2) >= 128) ksplit *= 2;
    const int intra = !accumulate && Kmax >= 128;
    const dim3 grid(intra ? cdiv(N, 32) : cdiv(cdiv(N, 32), 4), cdiv(Mmax, 32), ksplit * nb);
    const Gemm2nd two = {A2, B2, C2, M2, K2, batch, stride_a2, stride_b2, stride_c2};
    if (trans_a)
        hipLaunchKernelGGL(gemm32_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, A, B, C, M, N, K, lda, ldb, ldc, ksplit, accumulate,
                           intra, stride_a, stride_b, stride_c, two);
    else
        hipLaunchKernelGGL(gemm32_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, A, B, C, M, N, K, lda, ldb, ldc, ksplit, accumulate,
                           intra, stride_a, stride_b, stride_c, two);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int ramnet_gemm(const float *A, const float *B, float *C, int M, int N, int K, int lda, int ldb, int ldc, int trans_a,
                           int accumulate, int batch, long stride_a, long stride_b, long stride_c, void *stream) {
    return launch_gemm(A, B, C, M, N, K, lda, ldb, ldc, trans_a, accumulate, batch, stride_a, stride_b, stride_c, nullptr, nullptr, nullptr, 0, 0, 0, 0,
                       0, 0, stream);
}

extern "C" int ramnet_gemm2(const float *A, const float *B, float *C, int M, long stride_a, long stride_b, long stride_c, int batch, const float *A2,
                            const float *B2, float *C2, int M2, long stride_a2, long stride_b2, long stride_c2, int batch2, int N, int K, int K2, int lda,
                            int ldb, int ldc, int trans_a, int accumulate, void *stream) {
    RAMNET_CHECK_ARG(batch2 >= 1);
    return launch_gemm(A, B, C, M, N, K, lda, ldb, ldc, trans_a, accumulate, batch, stride_a, stride_b, stride_c, A2, B2, C2, M2, K2, batch2, stride_a2,
                       stride_b2, stride_c2, stream);
}
