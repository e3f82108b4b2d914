// BatchNorm / InstanceNorm of the RAM-Net layers (`norm: "BN" | "IN"`: ConvLayer submodules.py:13-24, 29-30; TransposedConvLayer
// :52-62; UpsampleConvLayer :82-94; ResidualBlock :188-193, 203-210) on NHWC fp32 tensors for gfx950.  No shipped config enables
// them (all use "none"), so these are plain HBM-bound kernels, not fused into the convolution epilogues:
//
//   ramnet_norm_partial   per (group, channel) partial sums  sum a, sum a*b  in fp64  (forward: a = b = x -> mean / variance;
//                         backward: a = dy * act'(y), b = x -> the two reductions of the normalisation's gradient)
//   ramnet_norm_finalize  partial sums (or the running buffers) -> mean, rstd (fp64), scale = gamma * rstd, shift = beta - mean * scale,
//                         and torch's running-buffer update, one thread per channel
//   ramnet_norm_apply     y = act(x * scale[g][c] + shift[g][c] [+ residual])
//   ramnet_norm_finalize_bwd  partial sums of the backward -> c1, c2, c3 [groups][C], dgamma, dbeta
//   ramnet_norm_bwd       dx = c1[g][c] * (dy * act'(y)) + c2[g][c] * x + c3[g][c]   [, dres = dy * act'(y)]
//
// group = the statistics' extent: 1 for BatchNorm (all B*H*W pixels), B for InstanceNorm (H*W pixels of one image).
// Lanes run along channels (consecutive addresses: 16-byte loads when C % 4 == 0), a workgroup's thread rows along pixels.
#include "common.hpp"

namespace ramnet {

__device__ __forceinline__ float norm_act_grad(float dy, float y, int act) {
    return act == 1 ? (y > 0.f ? dy : 0.f) : act == 2 ? dy * y * (1.f - y) : dy;
}

// grid (nslab, groups, channel blocks of TC lanes); part[((g * nslab + slab) * C + c) * 2 + {0, 1}]
template <int V>
__global__ void __launch_bounds__(256) norm_partial_kernel(const float *__restrict__ a, int lda, const float *__restrict__ y, int ldy, int act,
                                                          const float *__restrict__ b, int ldb, long npix, int C, int TC,
                                                          double *__restrict__ part) {
    const int tid = threadIdx.x, rows = 256 / TC;
    const int cl = blockIdx.z * TC + tid % TC, prow = tid / TC;
    const int c = cl * V, g = blockIdx.y, nslab = gridDim.x;
    double s0[V], s1[V];
#pragma unroll
    for (int j = 0; j < V; ++j) s0[j] = 0.0, s1[j] = 0.0;
    if (prow < rows && c < C) {
        const size_t base = (size_t)g * npix;
        for (long p = (long)blockIdx.x * rows + prow; p < npix; p += (long)nslab * rows) {
            float av[V], bv[V], yv[V];
            if constexpr (V == 4) {
                const float4 t = ld4(a + (base + p) * lda + c), u = ld4(b + (base + p) * ldb + c);
                const float4 w = y != nullptr ? ld4(y + (base + p) * ldy + c) : f4zero();
                av[0] = t.x, av[1] = t.y, av[2] = t.z, av[3] = t.w;
                bv[0] = u.x, bv[1] = u.y, bv[2] = u.z, bv[3] = u.w;
                yv[0] = w.x, yv[1] = w.y, yv[2] = w.z, yv[3] = w.w;
            } else {
                av[0] = a[(base + p) * lda + c], bv[0] = b[(base + p) * ldb + c];
                yv[0] = y != nullptr ? y[(base + p) * ldy + c] : 0.f;
            }
            if (y != nullptr) {
#pragma unroll
                for (int j = 0; j < V; ++j) av[j] = norm_act_grad(av[j], yv[j], act);
            }
#pragma unroll
            for (int j = 0; j < V; ++j) s0[j] += (double)av[j], s1[j] += (double)av[j] * (double)bv[j];
        }
    }
    __shared__ double red[256][2 * V];
#pragma unroll
    for (int j = 0; j < V; ++j) red[tid][2 * j] = s0[j], red[tid][2 * j + 1] = s1[j];
    __syncthreads();
    if (prow == 0 && c < C) {               // fixed order over the thread rows: bit-reproducible
        for (int r = 1; r < rows; ++r)
#pragma unroll
            for (int j = 0; j < V; ++j) s0[j] += red[r * TC + tid][2 * j], s1[j] += red[r * TC + tid][2 * j + 1];
        double *o = part + (((size_t)g * nslab + blockIdx.x) * C + c) * 2;
#pragma unroll
        for (int j = 0; j < V; ++j) o[2 * j] = s0[j], o[2 * j + 1] = s1[j];
    }
}

// MODE 0: out = act(x * k0 + k1 [+ res]);  MODE 1: dx = k0 * g + k1 * x + k2, dres = g with g = dy * act'(y)  (x0 = dy)
template <int V, int MODE>
__global__ void __launch_bounds__(256) norm_pointwise_kernel(const float *__restrict__ x0, int ld0, const float *__restrict__ y, int ldy, int act,
                                                            const float *__restrict__ x, int ldx, const float *__restrict__ k0,
                                                            const float *__restrict__ k1, const float *__restrict__ k2,
                                                            const float *__restrict__ res, int ldr, float *__restrict__ out, int ldo,
                                                            float *__restrict__ out2, int ldo2, long npix, int groups, int C) {
    const int CQ = C / V;
    const size_t total = (size_t)groups * npix * CQ;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % CQ) * V;
        const size_t pix = i / CQ;
        const int g = (int)(pix / npix);
        float v[V], w[V], r[V], o[V], o2[V];
        const float *rp = MODE == 0 ? (res ? res + pix * ldr + c : nullptr) : (y ? y + pix * ldy + c : nullptr);
        if constexpr (V == 4) {
            const float4 t = ld4(x0 + pix * ld0 + c), u = MODE == 1 ? ld4(x + pix * ldx + c) : f4zero(), q = rp ? ld4(rp) : f4zero();
            v[0] = t.x, v[1] = t.y, v[2] = t.z, v[3] = t.w;
            w[0] = u.x, w[1] = u.y, w[2] = u.z, w[3] = u.w;
            r[0] = q.x, r[1] = q.y, r[2] = q.z, r[3] = q.w;
        } else {
            v[0] = x0[pix * ld0 + c], w[0] = MODE == 1 ? x[pix * ldx + c] : 0.f, r[0] = rp ? rp[0] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const int gc = g * C + c + j;
            if (MODE == 0) {
                const float t = v[j] * k0[gc] + k1[gc] + r[j];
                o[j] = act == 1 ? fmaxf(t, 0.f) : act == 2 ? sigmoidf_(t) : t;
                o2[j] = 0.f;
            } else {
                const float gg = y ? norm_act_grad(v[j], r[j], act) : v[j];
                o[j] = k0[gc] * gg + k1[gc] * w[j] + k2[gc];
                o2[j] = gg;
            }
        }
        if constexpr (V == 4) {
            st4(out + pix * ldo + c, make_float4(o[0], o[1], o[2], o[3]));
            if (MODE == 1 && out2) st4(out2 + pix * ldo2 + c, make_float4(o2[0], o2[1], o2[2], o2[3]));
        } else {
            out[pix * ldo + c] = o[0];
            if (MODE == 1 && out2) out2[pix * ldo2 + c] = o2[0];
        }
    }
}

// Sum of the per-slab partials of (group g, channel c) by the 16 slab lanes of a channel (fixed order: bit-reproducible); the
// result is valid in lane 0.  Block = 256 threads = 16 channels x 16 slab lanes.
__device__ __forceinline__ void norm_slab_sum(const double *__restrict__ part, int g, int nslab, int C, int c, int lane, double (*red)[16][2],
                                              double &s0, double &s1) {
    const int cl = threadIdx.x & 15;
    s0 = 0.0, s1 = 0.0;
    if (c < C)
        for (int s = lane; s < nslab; s += 16) {
            const double *q = part + (((size_t)g * nslab + s) * C + c) * 2;
            s0 += q[0], s1 += q[1];
        }
    __syncthreads();                    // (the previous group's reads of `red` are done)
    red[lane][cl][0] = s0, red[lane][cl][1] = s1;
    __syncthreads();
    if (lane == 0)
        for (int l = 1; l < 16; ++l) s0 += red[l][cl][0], s1 += red[l][cl][1];
}

// 16 channels per block: fixed-order sum over the slabs, statistics of every group, running-buffer update (torch semantics:
// running = (1 - m) * running + m * mean over the groups of (mean, UNBIASED variance); num_batches_tracked += 1 when given).
// use_running: mean / variance ARE the running buffers (eval mode; groups = 1, `part` unused).
__global__ void norm_finalize_kernel(const double *__restrict__ part, int groups, int nslab, int C, double npix, double eps,
                                     const float *__restrict__ gamma, const float *__restrict__ beta, float *__restrict__ rmean,
                                     float *__restrict__ rvar, double momentum, int update, int use_running, long long *__restrict__ tracked,
                                     double *__restrict__ mean, double *__restrict__ rstd, float *__restrict__ scale, float *__restrict__ shift) {
    __shared__ double red[16][16][2];
    const int c = blockIdx.x * 16 + (threadIdx.x & 15), lane = threadIdx.x >> 4;
    const bool own = lane == 0 && c < C;
    if (c == 0 && lane == 0 && tracked != nullptr && update) tracked[0] += 1;
    const double ga = own && gamma ? (double)gamma[c] : 1.0, be = own && beta ? (double)beta[c] : 0.0;
    double am = 0.0, av = 0.0;
    for (int g = 0; g < groups; ++g) {
        double m = 0.0, v = 1.0;
        if (use_running) {
            if (own) m = (double)rmean[c], v = (double)rvar[c];
        } else {
            double s0, s1;
            norm_slab_sum(part, g, nslab, C, c, lane, red, s0, s1);
            m = s0 / npix;
            v = s1 / npix - m * m;
            v = v > 0.0 ? v : 0.0;
            am += m, av += v * (npix / (npix - 1.0));
        }
        if (!own) continue;
        const double r = 1.0 / sqrt(v + eps), sc = ga * r;
        mean[g * C + c] = m, rstd[g * C + c] = r;
        scale[g * C + c] = (float)sc, shift[g * C + c] = (float)(be - m * sc);
    }
    if (own && update && !use_running && rmean != nullptr) {
        rmean[c] = (float)((1.0 - momentum) * (double)rmean[c] + momentum * am / groups);
        rvar[c] = (float)((1.0 - momentum) * (double)rvar[c] + momentum * av / groups);
    }
}

// part = (sum g, sum g * x) per (group, slab, channel), g = dy * act'(y)  ->  dx = c1 * g + c2 * x + c3 with
// c1 = gamma * rstd, and under batch statistics c2 = -gamma * rstd^2 * S2 / N, c3 = -gamma * rstd * S1 / N - c2 * mean
// (S1 = sum g, S2 = sum g * xhat = rstd * (sum g x - mean * S1)); dgamma = sum_groups S2, dbeta = sum_groups S1.
__global__ void norm_finalize_bwd_kernel(const double *__restrict__ part, int groups, int nslab, int C, double npix,
                                         const double *__restrict__ mean, const double *__restrict__ rstd, const float *__restrict__ gamma,
                                         int batch_stats, float *__restrict__ c1, float *__restrict__ c2, float *__restrict__ c3,
                                         float *__restrict__ dgamma, float *__restrict__ dbeta) {
    __shared__ double red[16][16][2];
    const int c = blockIdx.x * 16 + (threadIdx.x & 15), lane = threadIdx.x >> 4;
    const bool own = lane == 0 && c < C;
    const double ga = own && gamma ? (double)gamma[c] : 1.0;
    double dg = 0.0, db = 0.0;
    for (int g = 0; g < groups; ++g) {
        double s1, sx;
        norm_slab_sum(part, g, nslab, C, c, lane, red, s1, sx);
        if (!own) continue;
        const double m = mean[g * C + c], r = rstd[g * C + c], s2 = r * (sx - m * s1);
        const double k2 = batch_stats ? -ga * r * r * s2 / npix : 0.0;
        c1[g * C + c] = (float)(ga * r);
        c2[g * C + c] = (float)k2;
        c3[g * C + c] = (float)(batch_stats ? -ga * r * s1 / npix - k2 * m : 0.0);
        dg += s2, db += s1;
    }
    if (own && dgamma) dgamma[c] = (float)dg;
    if (own && dbeta) dbeta[c] = (float)db;
}

static int pointwise_blocks(size_t total) {
    size_t b = (total + 255) / 256;
    return (int)(b > 16384 ? 16384 : b < 1 ? 1 : b);
}

}  // namespace ramnet

using namespace ramnet;

extern "C" int ramnet_norm_slabs(int groups, long npix, int C) {
    const int V = C % 4 == 0 ? 4 : 1, CQ = C / V, TC = CQ < 256 ? CQ : 256, rows = 256 / TC;
    long n = (npix + (long)rows * 8 - 1) / ((long)rows * 8);
    const long cap = 1024 / (groups < 1024 ? groups : 1024);        // ~1024 workgroups stream at full rate; the joins stay short
    if (n > cap) n = cap;
    return (int)(n < 1 ? 1 : n);
}

extern "C" int ramnet_norm_partial(const float *a, int lda, const float *y, int ldy, int act, const float *b, int ldb, int groups, long npix,
                                   int C, int nslab, double *part, void *stream) {
    RAMNET_CHECK_ARG(a && b && part && groups > 0 && npix > 0 && C > 0 && nslab > 0 && lda >= C && ldb >= C && (!y || ldy >= C));
    RAMNET_CHECK_ARG(act >= 0 && act <= 2);
    const bool v4 = C % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && (!y || ldy % 4 == 0) && (((uintptr_t)a | (uintptr_t)b | (uintptr_t)y) & 15) == 0;
    const int V = v4 ? 4 : 1, CQ = C / V, TC = CQ < 256 ? CQ : 256;
    const dim3 grid(nslab, groups, cdiv(CQ, TC));
    if (v4) hipLaunchKernelGGL(norm_partial_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, a, lda, y, ldy, act, b, ldb, npix, C, TC, part);
    else hipLaunchKernelGGL(norm_partial_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, a, lda, y, ldy, act, b, ldb, npix, C, TC, part);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int ramnet_norm_apply(const float *x, int ldx, const float *scale, const float *shift, const float *res, int ldr, int act, float *out,
                                 int ldo, int groups, long npix, int C, void *stream) {
    RAMNET_CHECK_ARG(x && scale && shift && out && groups > 0 && npix > 0 && C > 0 && ldx >= C && ldo >= C && (!res || ldr >= C));
    RAMNET_CHECK_ARG(act >= 0 && act <= 2);
    const bool v4 = C % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && (!res || ldr % 4 == 0) && (((uintptr_t)x | (uintptr_t)out | (uintptr_t)res) & 15) == 0;
    const int V = v4 ? 4 : 1;
    const int blocks = pointwise_blocks((size_t)groups * npix * (C / V));
    if (V == 4)
        hipLaunchKernelGGL((norm_pointwise_kernel<4, 0>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, nullptr, 0, act, nullptr, 0, scale,
                           shift, nullptr, res, ldr, out, ldo, nullptr, 0, npix, groups, C);
    else
        hipLaunchKernelGGL((norm_pointwise_kernel<1, 0>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, nullptr, 0, act, nullptr, 0, scale,
                           shift, nullptr, res, ldr, out, ldo, nullptr, 0, npix, groups, C);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int ramnet_norm_bwd(const float *dy, int lddy, const float *y, int ldy, int act, const float *x, int ldx, const float *c1, const float *c2,
                               const float *c3, float *dx, int lddx, float *dres, int lddres, int groups, long npix, int C, void *stream) {
    RAMNET_CHECK_ARG(dy && x && c1 && c2 && c3 && dx && groups > 0 && npix > 0 && C > 0 && lddy >= C && ldx >= C && lddx >= C);
    RAMNET_CHECK_ARG((!y || ldy >= C) && (!dres || lddres >= C) && act >= 0 && act <= 2 && (act == 0 || y));
    const bool v4 = C % 4 == 0 && lddy % 4 == 0 && ldx % 4 == 0 && lddx % 4 == 0 && (!y || ldy % 4 == 0) && (!dres || lddres % 4 == 0) &&
                    (((uintptr_t)dy | (uintptr_t)y | (uintptr_t)x | (uintptr_t)dx | (uintptr_t)dres) & 15) == 0;
    const int V = v4 ? 4 : 1;
    const int blocks = pointwise_blocks((size_t)groups * npix * (C / V));
    if (V == 4)
        hipLaunchKernelGGL((norm_pointwise_kernel<4, 1>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, lddy, y, ldy, act, x, ldx, c1, c2, c3,
                           nullptr, 0, dx, lddx, dres, lddres, npix, groups, C);
    else
        hipLaunchKernelGGL((norm_pointwise_kernel<1, 1>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, lddy, y, ldy, act, x, ldx, c1, c2, c3,
                           nullptr, 0, dx, lddx, dres, lddres, npix, groups, C);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int ramnet_norm_finalize(const double *part, int groups, int nslab, int C, long npix, double eps, const float *gamma, const float *beta,
                                    float *running_mean, float *running_var, double momentum, int update_running, int use_running,
                                    long long *num_batches_tracked, double *mean, double *rstd, float *scale, float *shift, void *stream) {
    RAMNET_CHECK_ARG(groups > 0 && C > 0 && npix > 0 && mean && rstd && scale && shift && (use_running || (part && nslab > 0)));
    RAMNET_CHECK_ARG(!use_running || (running_mean && running_var && groups == 1));
    RAMNET_CHECK_ARG(!update_running || use_running || !running_mean || npix > 1);
    RAMNET_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr));
    hipLaunchKernelGGL(norm_finalize_kernel, dim3(cdiv(C, 16)), dim3(256), 0, (hipStream_t)stream, part, groups, nslab, C, (double)npix, eps, gamma,
                       beta, running_mean, running_var, momentum, update_running, use_running, num_batches_tracked, mean, rstd, scale, shift);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int ramnet_norm_finalize_bwd(const double *part, int groups, int nslab, int C, long npix, const double *mean, const double *rstd,
                                        const float *gamma, int batch_stats, float *c1, float *c2, float *c3, float *dgamma, float *dbeta,
                                        void *stream) {
    RAMNET_CHECK_ARG(part && groups > 0 && nslab > 0 && C > 0 && npix > 0 && mean && rstd && c1 && c2 && c3);
    hipLaunchKernelGGL(norm_finalize_bwd_kernel, dim3(cdiv(C, 16)), dim3(256), 0, (hipStream_t)stream, part, groups, nslab, C, (double)npix, mean,
                       rstd, gamma, batch_stats, c1, c2, c3, dgamma, dbeta);
    RAMNET_LAUNCH_CHECK();
    return 0;
}
