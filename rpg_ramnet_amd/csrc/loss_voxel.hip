// Scale-invariant loss reductions and event -> voxel-grid binning (HBM / atomic bound).
#include <map>
#include <mutex>
#include <utility>
#include "common.hpp"

namespace ramnet {

static inline int grid_for(size_t n_items, int block = 256) {
    size_t g = (n_items + block - 1) / block;
    if (g > 256 * 8) g = 256 * 8;
    if (g < 1) g = 1;
    return (int)g;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    return v;
}

template <bool LOG> __device__ __forceinline__ float si_diff(float p, float t) { return LOG ? logf(p) - logf(t) : p - t; }

// stats[0] += sum d, stats[1] += sum d^2, stats[2] += count over non-NaN d = pred - target; stats[3] = arrival ticket: the LAST
// workgroup to arrive reads the three sums back and writes the loss (model/loss.py:6-9) — one launch instead of statistics +
// finalize.  At most 256 workgroups of 16-byte loads: the three double atomics per workgroup all hit the same three addresses
// (~12 ns each, serialised), which is what the 2048-workgroup version spent its 27 us on.
// LOG: scale_invariant_log_loss (model/loss.py:12-15): d = log(pred) - log(target) instead of pred - target
template <bool LOG>
__global__ void __launch_bounds__(1024) si_stats_kernel(const float *__restrict__ pred, const float *__restrict__ target, size_t n, float weight, float lambda,
                                double *stats, float *loss) {
    double s1 = 0.0, s2 = 0.0, cnt = 0.0;
    auto acc = [&](float d) {
        if (d == d) {
            s1 += (double)d;
            s2 += (double)d * (double)d;
            cnt += 1.0;
        }
    };
    const bool vec = ((reinterpret_cast<uintptr_t>(pred) | reinterpret_cast<uintptr_t>(target)) & 15) == 0;
    const size_t n4 = vec ? n / 4 : 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 p = ld4(pred + 4 * i), t = ld4(target + 4 * i);
        acc(si_diff<LOG>(p.x, t.x)), acc(si_diff<LOG>(p.y, t.y)), acc(si_diff<LOG>(p.z, t.z)), acc(si_diff<LOG>(p.w, t.w));
    }
    for (size_t i = 4 * n4 + blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc(si_diff<LOG>(pred[i], target[i]));
    __shared__ double red[3][16];
    s1 = wave_sum(s1), s2 = wave_sum(s2), cnt = wave_sum(cnt);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (lane == 0) red[0][wave] = s1, red[1][wave] = s2, red[2][wave] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 0; k < 3; ++k) {
            double t = 0.0;
            for (int w = 0; w < nw; ++w) t += red[k][w];
            atomicAdd(stats + k, t);
        }
        __threadfence();                                        // the sums are performed before the ticket
        const unsigned long long ticket = atomicAdd(reinterpret_cast<unsigned long long *>(stats + 3), 1ull);
        if (ticket == (unsigned long long)gridDim.x - 1) {      // every other workgroup's sums are in: read them where they live
            __threadfence();
            const double S1 = atomicAdd(stats + 0, 0.0), S2 = atomicAdd(stats + 1, 0.0), N = atomicAdd(stats + 2, 0.0);
            const double m = S1 / N;
            *loss = (float)((double)weight * (S2 / N - (double)lambda * m * m));
        }
    }
}

// The same statistics without the zero-fill launch and without contended atomics: every workgroup stores its three partial sums to
// part[workgroup][3] (a scratch owned by the library, one per device and stream), the LAST one to arrive — a self-resetting ticket behind
// the partials — adds them in a fixed order with all its threads and writes stats[0..2] and the loss.  One launch, bit-reproducible.
template <bool LOG>
__global__ void __launch_bounds__(1024) si_stats_fold_kernel(const float *__restrict__ pred, const float *__restrict__ target, size_t n, float weight,
                                                            float lambda, double *stats, float *loss, double *part, unsigned long long *ticket) {
    double s1 = 0.0, s2 = 0.0, cnt = 0.0;
    auto acc = [&](float d) {
        if (d == d) {
            s1 += (double)d;
            s2 += (double)d * (double)d;
            cnt += 1.0;
        }
    };
    const bool vec = ((reinterpret_cast<uintptr_t>(pred) | reinterpret_cast<uintptr_t>(target)) & 15) == 0;
    const size_t n4 = vec ? n / 4 : 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 p = ld4(pred + 4 * i), t = ld4(target + 4 * i);
        acc(si_diff<LOG>(p.x, t.x)), acc(si_diff<LOG>(p.y, t.y)), acc(si_diff<LOG>(p.z, t.z)), acc(si_diff<LOG>(p.w, t.w));
    }
    for (size_t i = 4 * n4 + blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc(si_diff<LOG>(pred[i], target[i]));
    __shared__ double red[3][16];
    __shared__ int is_last;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    s1 = wave_sum(s1), s2 = wave_sum(s2), cnt = wave_sum(cnt);
    if (lane == 0) red[0][wave] = s1, red[1][wave] = s2, red[2][wave] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 0; k < 3; ++k) {
            double t = 0.0;
            for (int w = 0; w < nw; ++w) t += red[k][w];
            part[blockIdx.x * 3 + k] = t;
        }
        __threadfence();                                        // the partials are visible before the ticket
        is_last = atomicAdd(ticket, 1ull) == (unsigned long long)gridDim.x - 1;
    }
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    double v[3] = {0.0, 0.0, 0.0};
    if (threadIdx.x < gridDim.x)
        for (int k = 0; k < 3; ++k) v[k] = __hip_atomic_load(part + threadIdx.x * 3 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int k = 0; k < 3; ++k) v[k] = wave_sum(v[k]);
    __syncthreads();                                            // (red is reused)
    if (lane == 0) red[0][wave] = v[0], red[1][wave] = v[1], red[2][wave] = v[2];
    __syncthreads();
    if (threadIdx.x == 0) {
        double S[3];
        for (int k = 0; k < 3; ++k) {
            S[k] = 0.0;
            for (int w = 0; w < nw; ++w) S[k] += red[k][w];
            stats[k] = S[k];
        }
        stats[3] = 0.0;
        const double m = S[0] / S[2];
        *loss = (float)((double)weight * (S[1] / S[2] - (double)lambda * m * m));
        atomicExch(ticket, 0ull);                               // ready for the next launch on this stream
    }
}

// loss from given statistics (the data-parallel exact form: stats = the all-reduced sums of every rank's maps)
__global__ void si_from_stats_kernel(const double *__restrict__ stats, float weight, float lambda, float *loss) {
    const double m = stats[0] / stats[2];
    *loss = (float)((double)weight * (stats[1] / stats[2] - (double)lambda * m * m));
}

template <bool LOG>
__global__ void si_bwd_kernel(const float *__restrict__ pred, const float *__restrict__ target, size_t n, float weight, float lambda,
                              const double *__restrict__ stats, const float *__restrict__ gscale, float *__restrict__ dpred) {
    // d - lambda*mean is formed in double: rounding the mean to fp32 would add the SAME offset to every pixel, and
    // gradients that are sums over pixels of this map (e.g. the last bias) cancel to ~0 and would keep only that offset.
    const double cnt = stats[2], mean = stats[0] / cnt;
    const double s2 = 2.0 * (double)((gscale ? *gscale : 1.0f) * weight) / cnt;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float d = si_diff<LOG>(pred[i], target[i]);
        const double g = s2 * ((double)d - (double)lambda * mean);
        dpred[i] = (d == d) ? (float)(LOG ? g / (double)pred[i] : g) : 0.f;      // (LOG: d log(pred) / d pred)
    }
}

// mse_loss (model/loss.py:18-19) of the trainer's extra term (lstm_trainer.py:169-185): F.mse_loss over the non-NaN TARGET entries, at full
// resolution (HALF = false) or after F.interpolate(scale_factor=0.5, mode='bilinear', align_corners=False) of both maps (HALF: output cell
// (i, j) = the mean of the 2 x 2 block at (2i, 2j), evaluated in torch's order of operations; a NaN anywhere in the target block makes the
// cell NaN, i.e. masked; an odd trailing row / column is dropped).  stats[0] = sum d^2, stats[1] = count, stats[3] = arrival ticket.
template <bool HALF> __device__ __forceinline__ float mse_cell(const float *__restrict__ m, int W, int y, int x) {
    if (!HALF) return m[(size_t)y * W + x];
    const float *r0 = m + (size_t)(2 * y) * W + 2 * x, *r1 = r0 + W;
    return 0.5f * (0.5f * r0[0] + 0.5f * r0[1]) + 0.5f * (0.5f * r1[0] + 0.5f * r1[1]);
}

template <bool HALF>
__global__ void __launch_bounds__(1024) mse_stats_kernel(const float *__restrict__ pred, const float *__restrict__ target, int B, int H, int W,
                                                        double *stats, float *loss) {
    const int Ho = HALF ? H / 2 : H, Wo = HALF ? W / 2 : W;
    const size_t n = (size_t)B * Ho * Wo;
    double s2 = 0.0, cnt = 0.0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % Wo), y = (int)((i / Wo) % Ho);
        const size_t b = i / ((size_t)Wo * Ho);
        const float t = mse_cell<HALF>(target + b * H * W, W, y, x);
        if (t == t) {
            const float d = mse_cell<HALF>(pred + b * H * W, W, y, x) - t;
            s2 += (double)d * (double)d, cnt += 1.0;
        }
    }
    __shared__ double red[2][16];
    s2 = wave_sum(s2), cnt = wave_sum(cnt);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (lane == 0) red[0][wave] = s2, red[1][wave] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, c = 0.0;
        for (int w = 0; w < nw; ++w) a += red[0][w], c += red[1][w];
        atomicAdd(stats + 0, a), atomicAdd(stats + 1, c);
        __threadfence();
        const unsigned long long ticket = atomicAdd(reinterpret_cast<unsigned long long *>(stats + 3), 1ull);
        if (ticket == (unsigned long long)gridDim.x - 1) {
            __threadfence();
            const double S = atomicAdd(stats + 0, 0.0), N = atomicAdd(stats + 1, 0.0);
            *loss = (float)(S / N);
        }
    }
}

template <bool HALF>
__global__ void mse_bwd_kernel(const float *__restrict__ pred, const float *__restrict__ target, int B, int H, int W,
                               const double *__restrict__ stats, const float *__restrict__ gscale, float *__restrict__ dpred) {
    const int Ho = HALF ? H / 2 : H, Wo = HALF ? W / 2 : W;
    const size_t n = (size_t)B * H * W;
    const double s = 2.0 * (double)(gscale ? *gscale : 1.0f) / stats[1] * (HALF ? 0.25 : 1.0);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), y = (int)((i / W) % H);
        const size_t b = i / ((size_t)W * H);
        const int cy = HALF ? y >> 1 : y, cx = HALF ? x >> 1 : x;
        float g = 0.f;
        if (cy < Ho && cx < Wo) {
            const float t = mse_cell<HALF>(target + b * H * W, W, cy, cx);
            if (t == t) g = (float)(s * (double)(mse_cell<HALF>(pred + b * H * W, W, cy, cx) - t));
        }
        dpred[i] = g;
    }
}

// ---------------------------------------------------------------------------------------- depth metrics
// Normalised log depth -> metric depth (evaluation.py:74-96: d = exp(reg*(y-1))*clip, prediction clipped to
// [exp(-reg)*clip, clip]) and the error sums of model/metric.py:8-33 / evaluation.py:201-241 over non-NaN targets whose
// metric depth is <= cutoff:  out[0..6] = n, sum|t-p|/(t+eps) (abs-rel), sum (t-p)^2/(t^2+eps) (sq-rel), sum (t-p)^2,
// sum (log t - log p)^2, sum (log t - log p), sum |t-p|;  out[7..9] = count(max(t/p,p/t) < 1.25^k), k=1..3.
__global__ void depth_metrics_kernel(const float *__restrict__ pred, const float *__restrict__ target, size_t n, float clip,
                                     float reg, float cutoff, double *out) {
    // out: n (non-NaN pixels of the mask), n_mask, sum |d|/(t+1e-6), sum d^2/(t^2+1e-6), sum d^2, sum ld^2, sum |ld|, sum |d|,
    // counts of ratio <= 1.25^k — operators and epsilons of evaluation.py:201-241 / metric.py:8-33
    double acc[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const float lo = expf(-reg) * clip;
    const double eps = 1e-5;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float tn = target[i];
        const bool nan = !(tn == tn);
        const float t = expf(reg * (tn - 1.0f)) * clip;
        if (!nan && !(t < cutoff)) continue;          // mask: nan_to_num(target) < cutoff  (NaN -> 0: inside)
        acc[1] += 1.0;
        if (nan) continue;
        const float p = fminf(fmaxf(expf(reg * (pred[i] - 1.0f)) * clip, lo), clip);
        const double d = (double)t - (double)p, ld = log((double)t + eps) - log((double)p + eps);
        acc[0] += 1.0, acc[2] += fabs(d) / ((double)t + 1e-6), acc[3] += d * d / ((double)t * t + 1e-6), acc[4] += d * d;
        acc[5] += ld * ld, acc[6] += fabs(ld), acc[7] += fabs(d);
        const double r = fmax((double)t / ((double)p + eps), (double)p / ((double)t + eps));
        acc[8] += r <= 1.25, acc[9] += r <= 1.5625, acc[10] += r <= 1.953125;
    }
    __shared__ double red[11][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 11; ++k) {
        const double v = wave_sum(acc[k]);
        if (lane == 0) red[k][wave] = v;
    }
    __syncthreads();
    if (threadIdx.x < 11) atomicAdd(out + threadIdx.x, red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// ---------------------------------------------------------------------------------------- multi-scale gradient loss
// MultiScaleGradient (model/loss.py:22-70): diff = pred - target; for k = 1,2,4,8: P = AvgPool2d(k,k)(diff);
// g = kornia spatial_gradient(P) (Sobel/8, replicate padding) ; loss_k = sum|g| over non-NaN / count * B * 2; mean over k.
// PARITY UNPINNED (kornia 0.4.0 absent): follows the restatement in oracle/loss_ref.py.
struct MsgScale { size_t off; int h, w, k; };
struct MsgArgs { MsgScale sc[4]; int ns, B, H, W; };

__global__ void msg_pool_kernel(const float *__restrict__ pred, const float *__restrict__ target, float *__restrict__ ws, MsgArgs a) {
    for (int s = 0; s < a.ns; ++s) {
        const MsgScale sc = a.sc[s];
        const size_t n = (size_t)a.B * sc.h * sc.w;
        const float inv = 1.0f / (float)(sc.k * sc.k);
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
            const int x = (int)(i % sc.w);
            const size_t j = i / sc.w;
            const int y = (int)(j % sc.h), b = (int)(j / sc.h);
            float acc = 0.f;
            for (int dy = 0; dy < sc.k; ++dy)
                for (int dx = 0; dx < sc.k; ++dx) {
                    const size_t p = ((size_t)b * a.H + (y * sc.k + dy)) * a.W + (x * sc.k + dx);
                    acc += pred[p] - target[p];          // NaN targets propagate, exactly like avg_pool2d
                }
            ws[sc.off + i] = acc * inv;
        }
    }
}

__device__ __forceinline__ void msg_sobel(const float *__restrict__ P, int h, int w, int y, int x, float &gx, float &gy) {
    const int ym = y > 0 ? y - 1 : 0, yp = y < h - 1 ? y + 1 : h - 1, xm = x > 0 ? x - 1 : 0, xp = x < w - 1 ? x + 1 : w - 1;
    const float a = P[ym * w + xm], b = P[ym * w + x], c = P[ym * w + xp];
    const float d = P[y * w + xm], e = P[y * w + x], f = P[y * w + xp];
    const float g = P[yp * w + xm], hh = P[yp * w + x], i = P[yp * w + xp];
    // the reference convolves with the full 3x3 kernel: a NaN anywhere in the window (even under a zero tap: 0*NaN)
    // makes BOTH components NaN
    const float nanprop = 0.f * (a + b + c + d + e + f + g + hh + i);
    gx = (-a + c - 2.f * d + 2.f * f - g + i) * 0.125f + nanprop;
    gy = (-a - 2.f * b - c + g + 2.f * hh + i) * 0.125f + nanprop;
}

__global__ void msg_stats_kernel(const float *__restrict__ ws, MsgArgs a, double *stats) {
    for (int s = 0; s < a.ns; ++s) {
        const MsgScale sc = a.sc[s];
        const size_t n = (size_t)a.B * sc.h * sc.w;
        double sum = 0.0, cnt = 0.0;
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
            const int x = (int)(i % sc.w);
            const size_t j = i / sc.w;
            const int y = (int)(j % sc.h), b = (int)(j / sc.h);
            float gx, gy;
            msg_sobel(ws + sc.off + (size_t)b * sc.h * sc.w, sc.h, sc.w, y, x, gx, gy);
            if (gx == gx) sum += fabsf(gx), cnt += 1.0;
            if (gy == gy) sum += fabsf(gy), cnt += 1.0;
        }
        __shared__ double red[2][4];
        sum = wave_sum(sum), cnt = wave_sum(cnt);
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        __syncthreads();
        if (lane == 0) red[0][wave] = sum, red[1][wave] = cnt;
        __syncthreads();
        if (threadIdx.x < 2) atomicAdd(stats + 2 * s + threadIdx.x, red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]);
    }
}

__global__ void msg_finalize_kernel(const double *stats, int ns, int B, float *loss) {
    double t = 0.0;
    for (int s = 0; s < ns; ++s) t += stats[2 * s] / stats[2 * s + 1] * (double)B * 2.0;
    *loss = (float)(t / ns);
}

// d loss / d P_s scattered through the transposed Sobel stencil (replicate padding = clamped scatter targets)
__global__ void msg_bwd_scatter_kernel(const float *__restrict__ ws, float *__restrict__ dws, MsgArgs a, const double *__restrict__ stats,
                                       const float *__restrict__ gscale) {
    const float up = gscale ? *gscale : 1.0f;
    for (int s = 0; s < a.ns; ++s) {
        const MsgScale sc = a.sc[s];
        const size_t n = (size_t)a.B * sc.h * sc.w;
        const float coef = up * (float)((double)a.B * 2.0 / stats[2 * s + 1] / a.ns) * 0.125f;
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
            const int x = (int)(i % sc.w);
            const size_t j = i / sc.w;
            const int y = (int)(j % sc.h), b = (int)(j / sc.h);
            const size_t base = sc.off + (size_t)b * sc.h * sc.w;
            float gx, gy;
            msg_sobel(ws + base, sc.h, sc.w, y, x, gx, gy);
            const float sx = gx == gx ? (gx > 0.f ? coef : gx < 0.f ? -coef : 0.f) : 0.f;
            const float sy = gy == gy ? (gy > 0.f ? coef : gy < 0.f ? -coef : 0.f) : 0.f;
            if (sx == 0.f && sy == 0.f) continue;
            const int h = sc.h, w = sc.w;
            const int ym = y > 0 ? y - 1 : 0, yp = y < h - 1 ? y + 1 : h - 1, xm = x > 0 ? x - 1 : 0, xp = x < w - 1 ? x + 1 : w - 1;
            float *D = dws + base;
            atomicAdd(D + ym * w + xm, -sx - sy);
            atomicAdd(D + ym * w + x, -2.f * sy);
            atomicAdd(D + ym * w + xp, sx - sy);
            atomicAdd(D + y * w + xm, -2.f * sx);
            atomicAdd(D + y * w + xp, 2.f * sx);
            atomicAdd(D + yp * w + xm, -sx + sy);
            atomicAdd(D + yp * w + x, 2.f * sy);
            atomicAdd(D + yp * w + xp, sx + sy);
        }
    }
}

// dpred[b,Y,X] = sum_s dP_s[b, Y/k, X/k] / k^2 (transposed average pooling, gathered)
__global__ void msg_bwd_gather_kernel(const float *__restrict__ dws, MsgArgs a, float *__restrict__ dpred) {
    const size_t n = (size_t)a.B * a.H * a.W;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int X = (int)(i % a.W);
        const size_t j = i / a.W;
        const int Y = (int)(j % a.H), b = (int)(j / a.H);
        float g = 0.f;
        for (int s = 0; s < a.ns; ++s) {
            const MsgScale sc = a.sc[s];
            const int y = Y / sc.k, x = X / sc.k;
            if (y < sc.h && x < sc.w) g += dws[sc.off + ((size_t)b * sc.h + y) * sc.w + x] / (float)(sc.k * sc.k);
        }
        dpred[i] = g;
    }
}

static int msg_args(int B, int H, int W, int ns, MsgArgs &a, size_t &total) {
    if (ns < 1 || ns > 4) return 1;
    a.ns = ns, a.B = B, a.H = H, a.W = W;
    total = 0;
    for (int s = 0; s < ns; ++s) {
        const int k = 1 << s;
        a.sc[s].k = k, a.sc[s].h = H / k, a.sc[s].w = W / k, a.sc[s].off = total;
        if (a.sc[s].h < 1 || a.sc[s].w < 1) return 1;
        total += (size_t)B * a.sc[s].h * a.sc[s].w;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------- voxel grid
// Index arithmetic restated from events_to_voxel_grid_pytorch (utils/event_tensor_utils.py:152-180):
// float64 normalised time, floor, float32 votes pol*(1-dt) / pol*dt, `0 <= ti < bins` guards.
#pragma clang fp contract(off)
// (xs, ys, pol) of an event and whether it lies inside the image; then the two votes of (t, xs, ys, pol) — split so that the sorted form
// can carry the integer coordinates between its two passes; voxel_vote() is the composition, operation for operation as before.
__device__ __forceinline__ bool voxel_coord(double x, double y, double p, int W, int H, long long &xs, long long &ys, float &pol) {
    xs = (long long)x, ys = (long long)y;
    pol = (float)p;
    if (pol == 0.f) pol = -1.f;
    return xs >= 0 && xs < W && ys >= 0 && ys < H;        // the reference would raise on an OOB index
}

__device__ __forceinline__ void voxel_vote_t(double t, long long xs, long long ys, float pol, bool inside, double t0, double t1, int bins, int W,
                                             int H, long long &il, float &vl, long long &ir, float &vr, long long *til_out = nullptr,
                                             long long *base_out = nullptr) {
    double dT = t1 - t0;
    if (dT == 0.0) dT = 1.0;
    const double ts = ((double)(bins - 1) * (t - t0)) / dT;
    const double tis = floor(ts);
    const long long til = (long long)tis;
    const float dts = (float)(ts - tis);
    vl = pol * (1.0f - dts);
    vr = pol * dts;
    const long long base = xs + ys * (long long)W;
    il = (inside && tis < (double)bins && tis >= 0.0) ? base + til * (long long)W * H : -1;
    ir = (inside && (tis + 1.0) < (double)bins && tis >= 0.0) ? base + (til + 1) * (long long)W * H : -1;
    if (til_out) *til_out = til;
    if (base_out) *base_out = base;
}

__device__ __forceinline__ void voxel_vote(double t, double x, double y, double p, double t0, double t1, int bins, int W, int H,
                                           long long &il, float &vl, long long &ir, float &vr, long long *til_out = nullptr,
                                           long long *base_out = nullptr) {
    long long xs, ys;
    float pol;
    const bool inside = voxel_coord(x, y, p, W, H, xs, ys, pol);
    voxel_vote_t(t, xs, ys, pol, inside, t0, t1, bins, W, H, il, vl, ir, vr, til_out, base_out);
}

__device__ __forceinline__ void voxel_event(const double *__restrict__ ev, size_t i, size_t n, int bins, int W, int H,
                                            long long &il, float &vl, long long &ir, float &vr) {
    voxel_vote(ev[i * 4], ev[i * 4 + 1], ev[i * 4 + 2], ev[i * 4 + 3], ev[0], ev[(n - 1) * 4], bins, W, H, il, vl, ir, vr);
}

__global__ void voxelize_kernel(const double *__restrict__ ev, size_t n, int bins, int W, int H, float *__restrict__ grid) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        long long il, ir;
        float vl, vr;
        voxel_event(ev, i, n, bins, W, H, il, vl, ir, vr);
        if (il >= 0) atomicAdd(grid + il, vl);
        if (ir >= 0) atomicAdd(grid + ir, vr);
    }
}

// G event lists in ONE launch (a batch of B x K grids): blockIdx.y = grid; list g = events[off[g] .. off[g+1]) with its own
// first / last timestamp, scattered into grids[g].
__global__ void voxelize_batch_kernel(const double *__restrict__ ev, const long long *__restrict__ off, int bins, int W, int H,
                                      float *__restrict__ grids) {
    const int g = blockIdx.y;
    const long long e0 = off[g], n = off[g + 1] - e0;
    const double *e = ev + (size_t)e0 * 4;
    float *grid = grids + (size_t)g * bins * W * H;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (size_t)n; i += (size_t)gridDim.x * blockDim.x) {
        long long il, ir;
        float vl, vr;
        voxel_event(e, i, (size_t)n, bins, W, H, il, vl, ir, vr);
        if (il >= 0) atomicAdd(grid + il, vl);
        if (ir >= 0) atomicAdd(grid + ir, vr);
    }
}

// Row-band form of the scatter-add (the default since round 3): workgroup (band, grid) owns rows [y0, y0 + rows) of ALL bins of one
// grid in LDS (bins x rows x W floats <= 160 KB), walks the whole event list of its grid, keeps the events whose row falls
// into its band (same index / vote arithmetic, op for op: voxel_event) and resolves the votes with LDS atomics; then every cell of the
// band is written once with plain coalesced stores — no zero-fill pass, no global atomics.  The global-atomic form above is bound
// by ~19 G random fp32 atomics/s of the memory side (858 us for the 8 M events of a package batch, 0.066 of the HBM roofline);
// here a list is re-read once per band from L2 / Infinity Cache (6.4 MB per grid) and the cost is the event walk itself.
// Band id of every event of every grid (0 .. nbands-1, 255 = outside the image rows): one byte per event, so that a band workgroup
// walks 200 KB of ids instead of the 6.4 MB list of its grid and fetches whole events for its hits only.  ids[g * stride + j].
__global__ void voxel_band_ids_kernel(const double *__restrict__ ev, const long long *__restrict__ off, long long n_single, int H, int rows,
                                      unsigned char *__restrict__ ids, long long stride) {
    const int g = blockIdx.y;
    const long long e0 = off ? off[g] : 0, n = off ? off[g + 1] - e0 : n_single;
    const double *e = ev + (size_t)e0 * 4;
    unsigned char *idg = ids + (size_t)g * stride;
    for (long long j = blockIdx.x * (long long)blockDim.x + threadIdx.x; j < n; j += (long long)gridDim.x * blockDim.x) {
        const double y = e[j * 4 + 2];
        // (long long)y truncates towards zero: rows -0.x are row 0; the int conversion is exact for the in-range values
        const bool in = y > -1.0 && y < (double)H;
        idg[j] = in ? (unsigned char)((y < 0.0 ? 0 : (int)y) / rows) : (unsigned char)255;
    }
}

template <bool IDS>
__global__ void __launch_bounds__(1024) voxelize_bands_kernel(const double *__restrict__ ev, const long long *__restrict__ off,
                                                              long long n_single, int n_grids, int bins, int W, int H, int rows,
                                                              float *__restrict__ grids, const unsigned char *__restrict__ ids,
                                                              long long ids_stride) {
    extern __shared__ __attribute__((aligned(16))) float band[];          // [bins][rows][W]
    // All bands of a grid walk the same event list: they are dealt to ONE XCD (workgroup b runs on XCD b % 8) as consecutive
    // slots, so that the list streams through that XCD's L2 once instead of once per band from the Infinity Cache (12 bands x 40
    // grids x 6.4 MB = 3 GB per launch, 5.6 TB/s: the walk was bound by exactly that)
    const int nbands = (H + rows - 1) / rows;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int g = (slot / nbands) * 8 + xcd, y0 = (slot % nbands) * rows;
    if (g >= n_grids) return;
    const int nr = min(rows, H - y0);
    const long long e0 = off ? off[g] : 0, n = off ? off[g + 1] - e0 : n_single;
    const double *e = ev + (size_t)e0 * 4;
    const int cells = bins * rows * W;
    for (int i = threadIdx.x; i < cells; i += blockDim.x) band[i] = 0.f;
    if (threadIdx.x < 2) reinterpret_cast<int *>(band + cells)[(IDS ? 8192 : 4096) + threadIdx.x] = 0;
    __syncthreads();
    const long long plane = (long long)W * H;
    const double t0 = n > 0 ? e[0] : 0.0, t1 = n > 0 ? e[(n - 1) * 4] : 0.0;
    const double2 *e2 = reinterpret_cast<const double2 *>(e);
    auto vote = [&](long long i) {                  // (the event is re-read: it was loaded a moment ago, L1 / L2 resident)
        const double2 tx = e2[i * 2], yp = e2[i * 2 + 1];
        long long il, ir, til, base;
        float vl, vr;
        voxel_vote(tx.x, tx.y, yp.x, yp.y, t0, t1, bins, W, H, il, vl, ir, vr, &til, &base);
        // flat index = x + y*W + bin*W*H  ->  band cell (bin*rows + y - y0)*W + x
        const int cell = (int)(til * rows * W + base - (long long)y0 * W);
        if (il >= 0) atomicAdd(band + cell, vl);
        if (ir >= 0) atomicAdd(band + cell + rows * W, vr);
    };
    // A workgroup walks the whole list of its grid but only ~rows/H of the events fall into its band.  Phase 1 of a batch tests
    // the row of every event on the double itself ((long long)y truncates towards zero: trunc(y) in [y0, y0 + nr) <=> y in [y0, y0 +
    // nr) for y0 > 0, y in (-1, nr) for y0 = 0) — whole events, two 16-byte loads per lane, a wave covers 2 KB contiguous — and
    // queues the hits in LDS; phase 2 votes on the queue with all lanes busy.  Voting inside the walk ran the ~300-instruction vote
    // path (f64 division, floor, conversions) in EVERY wave iteration for the 5 lanes of 64 that had a hit: 250 us per workgroup.
    const double band_lo = y0 == 0 ? -1.0 : (double)y0, band_hi = (double)(y0 + nr);
    auto in_band = [&](double y) { return (y0 == 0 ? y > band_lo : y >= band_lo) && y < band_hi; };
    constexpr int U = IDS ? 8 : 4, QCAP = U * 1024;  // events per thread and batch; a batch has U * 1024 = QCAP events
    int *queue = reinterpret_cast<int *>(band + cells);            // [QCAP] event offsets inside the batch, behind the band
    int *qn = queue + QCAP;                                      // [2] fill counters, alternating by batch
    const long long stride = blockDim.x;
    // The rows of batch k+1 are requested before batch k is tested and voted on: the two barriers of a batch wait for the LDS
    // traffic only (s_waitcnt lgkmcnt(0); a __syncthreads() would also drain the loads in flight), so the walk is not a chain of
    // exposed load latencies (78 % of the wave cycles were waits: 49 batches x ~5 us per workgroup).
    auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    int par = 0;
    // queue append, aggregated per wave: ONE LDS atomic per wave (lane 0, by the number of hits of the wave) instead of one per hit —
    // ~550 returning atomics per batch on the same LDS word serialise
    auto push = [&](bool hit, int value) {
        const unsigned long long mask = __ballot(hit);
        const int lane = threadIdx.x & 63;
        int base = 0;
        if (lane == 0 && mask) base = atomicAdd(qn + par, __popcll(mask));
        base = __shfl(base, 0);
        if (hit) queue[base + __popcll(mask & ((1ull << lane) - 1ull))] = value;
    };
    auto drain = [&](long long b0) {
        lds_barrier();
        const int cnt = qn[par];
        if (threadIdx.x == 0) qn[par ^ 1] = 0;      // (read again only behind the next barrier)
        for (int k = threadIdx.x; k < cnt; k += (int)stride) vote(b0 + queue[k]);
        lds_barrier();                              // the queue is drained before the next batch refills it
        par ^= 1;
    };
    const long long bstep = (long long)U * stride;
    if constexpr (IDS) {
        // eight band ids per thread and batch (one 8-byte load); a batch of 8192 events leaves ~550 hits in the queue
        const unsigned char *idg = ids + (size_t)g * ids_stride;
        const unsigned char mine = (unsigned char)(y0 / rows);
        auto load_ids = [&](long long b0) {
            const long long j = b0 + (long long)threadIdx.x * U;
            return j < n ? *reinterpret_cast<const uint2 *>(idg + j) : make_uint2(0xffffffffu, 0xffffffffu);     // (stride % 16 == 0)
        };
        uint2 cur = load_ids(0);
        for (long long b0 = 0; b0 < n; b0 += bstep) {
            const uint2 nxt = load_ids(b0 + bstep);                  // in flight across the barriers of this batch
            const long long j = b0 + (long long)threadIdx.x * U;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const unsigned b = ((u < 4 ? cur.x : cur.y) >> (8 * (u & 3))) & 0xffu;
                push((b == mine) & (j + u < n), (int)threadIdx.x * U + u);
            }
            drain(b0);
            cur = nxt;
        }
    } else {
        auto load_rows = [&](long long b0, double2 (&yp)[U]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long long i = b0 + u * stride + threadIdx.x;       // (clamped, not predicated: no exec-mask regions around the loads)
                yp[u] = e2[(i < n ? i : n - 1) * 2 + 1];
            }
        };
        auto batch = [&](long long b0, const double2 (&cur)[U], double2 (&nxt)[U]) {
            load_rows(min(b0 + bstep, n - 1), nxt);                      // in flight until the NEXT batch tests them
#pragma unroll
            for (int u = 0; u < U; ++u) push(in_band(cur[u].x) & (b0 + u * stride + threadIdx.x < n), u * (int)stride + (int)threadIdx.x);
            drain(b0);
        };
        double2 ra[U], rb[U];                           // two register sets in ping-pong (a copy would wait for the loads in flight)
        if (n > 0) load_rows(0, ra);
        for (long long b0 = 0; b0 < n; b0 += 2 * bstep) {
            batch(b0, ra, rb);
            if (b0 + bstep < n) batch(b0 + bstep, rb, ra);     // (uniform over the workgroup)
        }
    }
    __syncthreads();
    float *grid = grids + (size_t)g * bins * plane;
    const int per = nr * W;                                                  // cells of one bin's band: contiguous in the grid
    for (int bin = 0; bin < bins; ++bin)
        for (int i = threadIdx.x; i < per; i += blockDim.x) grid[(size_t)bin * plane + (size_t)y0 * W + i] = band[bin * rows * W + i];
}

// ---- sorted form (round 3, batches of >= 16 grids): every band reads ITS events as one contiguous run per chunk instead of walking its
// grid's whole list.  Pass 1 (voxel_sort_chunks_kernel): a workgroup takes a chunk of VS_CH consecutive events of one grid (whole events,
// coalesced), converts the coordinates exactly like voxel_vote (voxel_coord), drops the events outside the image (they have no votes), counts
// the rest per band with LDS atomics (the returned value is the event's rank inside its band) and writes a 16-byte record
// {t (f64), x | y << 16, polarity (f32)} per event, grouped by band, to rec[g][chunk * VS_CH ...] plus the chunk's band offsets to
// tab[g][chunk][band].  Pass 2 (voxelize_sorted_kernel): workgroup (band, grid) holds the band in LDS as before and, chunk by chunk, reads its
// run of records (contiguous 16-byte loads, ~VS_CH / bands entries) and votes (voxel_vote_t: the same double arithmetic on t) — no scan of
// foreign events, no gather, no barrier inside the walk, four chunks of loads in flight.  HBM: the events once (256 MB per 8 M), the records
// written and read once (2 x 128 MB), the grids once.  The order of the LDS vote atomics is as undefined as in the other forms.
constexpr int VS_CH = 8192;          // events per sort chunk: 1024 threads x 8
constexpr double VS_ONE = 1099511627776.0;       // 2^40: fixed-point unit of the vote sums

__global__ void __launch_bounds__(1024) voxel_sort_chunks_kernel(const double *__restrict__ ev, const long long *__restrict__ off, int bins, int W, int H,
                                                                 int rows, int nbands, int4 *__restrict__ rec, long long rec_stride,
                                                                 int *__restrict__ tab, int nchunks) {
    __shared__ int cnt[256];
    const int g = blockIdx.y, w = blockIdx.x, tid = threadIdx.x;
    const long long e0 = off[g], n = off[g + 1] - e0;
    const double2 *e2 = reinterpret_cast<const double2 *>(ev + (size_t)e0 * 4);
    // voxel_vote_t()'s double arithmetic on t runs HERE (this pass waits for HBM; the second pass was bound by it: ~100 instructions per event,
    // the division alone ~15): the record carries the bin code and the fp32 time fraction — the same operations in the same order, so the
    // same cells get the same votes.  Bin code: -1 = no vote (t before the first timestamp, NaN), bins = past the last bin, else floor(ts).
    const double t0 = n > 0 ? ev[(size_t)e0 * 4] : 0.0, t1 = n > 0 ? ev[((size_t)e0 + n - 1) * 4] : 0.0;
    double dT = t1 - t0;
    if (dT == 0.0) dT = 1.0;
    const double scale = (double)(bins - 1), dbins = (double)bins;
    if (tid < 256) cnt[tid] = 0;
    __syncthreads();
    const long long j0 = (long long)w * VS_CH;
    int id[8], rank[8];
    int4 r[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const long long j = j0 + u * 1024 + tid;
        double2 tx = make_double2(0.0, -1.0), yp = make_double2(-1.0, 0.0);
        if (j < n) tx = e2[j * 2], yp = e2[j * 2 + 1];
        // voxel_coord() without its 64-bit conversions: (long long)v truncates towards zero, so 0 <= (long long)v < N  <=>  -1 < v < N
        // (a NaN fails both tests, as it fails `inside`), and inside that range the 32-bit conversion is the same integer
        const bool inside = j < n && tx.y > -1.0 && tx.y < (double)W && yp.x > -1.0 && yp.x < (double)H;
        const int xs = inside ? (int)tx.y : 0, ys = inside ? (int)yp.x : 0;
        float pol = (float)yp.y;
        if (pol == 0.f) pol = -1.f;
        id[u] = inside ? ys / rows : 255;
        const double ts = (scale * (tx.x - t0)) / dT;
        const double tis = floor(ts);
        const float dts = (float)(ts - tis);
        const int tcode = !(tis >= 0.0) ? -1 : (tis < dbins ? (int)tis : bins);
        r[u] = make_int4(tcode, __float_as_int(dts), xs | (ys << 16), __float_as_int(pol));
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) rank[u] = id[u] < 255 ? atomicAdd(&cnt[id[u]], 1) : 0;
    __syncthreads();
    int *t = tab + ((size_t)g * nchunks + w) * (nbands + 1);
    if (tid == 0) {
        int acc = 0;
        for (int b = 0; b < nbands; ++b) {
            const int c = cnt[b];
            cnt[b] = acc, t[b] = acc;
            acc += c;
        }
        t[nbands] = acc;
    }
    __syncthreads();
    int4 *rg = rec + (size_t)g * rec_stride + j0;
#pragma unroll
    for (int u = 0; u < 8; ++u)
        if (id[u] < 255) rg[cnt[id[u]] + rank[u]] = r[u];
}

__global__ void __launch_bounds__(1024) voxelize_sorted_kernel(const double *__restrict__ ev, const long long *__restrict__ off, int n_grids, int bins,
                                                               int W, int H, int rows, float *__restrict__ grids, const int4 *__restrict__ rec,
                                                               long long rec_stride, const int *__restrict__ tab, int nchunks) {
    // Votes are summed as 64-bit FIXED-POINT integers (2^-40 units): an fp32 vote of magnitude >= 2^-16 is represented exactly, so the sum of
    // a cell is exact and independent of the order of the atomics — bit-reproducible, and at least as accurate as the reference's sequential
    // fp32 sum (one rounding, at the final conversion).  And it is what makes the pass fast: measured on gfx950, LDS integer atomics (32- and
    // 64-bit) cost ~3 us of this kernel, ds_add_f32 73 us (16 M votes; ~0.4 float atomics per clock and CU).
    extern __shared__ __attribute__((aligned(16))) long long band64[];    // [bins][rows][W]
    long long *band = band64;
    const int nbands = (H + rows - 1) / rows;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;              // the bands of a grid on one XCD, as voxelize_bands_kernel
    const int g = (slot / nbands) * 8 + xcd, bi = slot % nbands, y0 = bi * rows;
    if (g >= n_grids) return;
    const int nr = min(rows, H - y0), tid = threadIdx.x;
    const long long e0 = off[g], n = off[g + 1] - e0;
    const double *e = ev + (size_t)e0 * 4;
    const int cells = bins * rows * W;
    for (int i = tid; i < cells; i += blockDim.x) band[i] = 0;
    const long long plane = (long long)W * H;
    // the votes of a record {bin code, fp32 time fraction, x | y << 16, polarity} (the first pass ran voxel_vote_t's double arithmetic on t):
    // vl = pol (1 - dts) into bin code, vr = pol dts into the next one, under voxel_vote_t's own guards (0 <= tis, tis [+ 1] < bins)
    const int rw = rows * W;
    auto vote = [&](int4 r) {
        const int tc = r.x;
        const float dts = __int_as_float(r.y), pol = __int_as_float(r.w);
        const float vl = pol * (1.0f - dts), vr = pol * dts;
        const bool okl = tc >= 0 && tc < bins, okr = tc >= 0 && tc + 1 < bins;
        const int cell = (okl ? tc : 0) * rw + ((int)((unsigned)r.z >> 16) - y0) * W + (r.z & 0xffff);       // (bin * rows + y - y0) * W + x
        unsigned long long *acc = reinterpret_cast<unsigned long long *>(band);
        if (okl) atomicAdd(acc + cell, (unsigned long long)__double2ll_rn((double)vl * VS_ONE));
        if (okr) atomicAdd(acc + cell + rw, (unsigned long long)__double2ll_rn((double)vr * VS_ONE));
    };
    const int4 *rg = rec + (size_t)g * rec_stride;
    // This band's run in every chunk -> LDS behind the band: tl[2 w] = first record of the run in chunk w, tl[2 w + 1] = events of the band in the
    // chunks before w (exclusive prefix; tl[2 nchunks + 1] = all of them).  The walk then goes over the band's events as ONE flat list, eight per
    // thread and trip with their loads issued together (item j lies in the chunk a binary search of the prefixes names): two dependent memory round
    // trips per workgroup — table, records — instead of one per four chunks with a third of the threads busy (round 6: 77 -> 66 us per 40 grids).
    int *tl = reinterpret_cast<int *>(band + cells);
    {
        const int *tg = tab + (size_t)g * nchunks * (nbands + 1) + bi;
        for (int w = tid; w < nchunks; w += blockDim.x) {
            const int a = tg[(size_t)w * (nbands + 1)], b = tg[(size_t)w * (nbands + 1) + 1];
            tl[2 * w] = w * VS_CH + a, tl[2 * w + 1] = b - a;
        }
    }
    __syncthreads();
    if (tid < 64) {                                   // exclusive scan of the run lengths, 64 chunks per trip of wave 0
        int carry = 0;
        for (int w0 = 0; w0 < nchunks; w0 += 64) {
            const int w = w0 + tid;
            const int len = w < nchunks ? tl[2 * w + 1] : 0;
            int inc = len;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int up = __shfl_up(inc, o);
                if (tid >= o) inc += up;
            }
            if (w < nchunks) tl[2 * w + 1] = carry + inc - len;
            carry += __shfl(inc, 63);
        }
        if (tid == 0) tl[2 * nchunks + 1] = carry;
    }
    __syncthreads();
    const int total = tl[2 * nchunks + 1];
    auto locate = [&](int j) {                        // record index of the band's j-th event
        int lo = 0, hi = nchunks - 1;                 // the LAST chunk whose prefix is <= j: its run is not empty (a later prefix would be <= j too)
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (tl[2 * mid + 1] <= j) lo = mid; else hi = mid - 1;
        }
        return tl[2 * lo] + (j - tl[2 * lo + 1]);
    };
    for (int j0 = tid; j0 < total; j0 += 8 * 1024) {
        int4 r[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = j0 + u * 1024;
            r[u] = j < total ? rg[locate(j)] : make_int4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (j0 + u * 1024 < total) vote(r[u]);
    }
    __syncthreads();
    float *grid = grids + (size_t)g * bins * plane;
    const int per = nr * W;                                                  // cells of one bin's band: contiguous in the grid
    for (int bin = 0; bin < bins; ++bin)
        for (int i = tid; i < per; i += blockDim.x)
            grid[(size_t)bin * plane + (size_t)y0 * W + i] = (float)((double)band[bin * rows * W + i] * (1.0 / VS_ONE));
}

__global__ void voxel_indices_kernel(const double *__restrict__ ev, size_t n, int bins, int W, int H, long long *il_out, long long *ir_out) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        long long il, ir;
        float vl, vr;
        voxel_event(ev, i, n, bins, W, H, il, vl, ir, vr);
        il_out[i] = il, ir_out[i] = ir;
    }
}

__global__ void nonzero_stats_kernel(const float *__restrict__ g, size_t n, double *stats) {
    double s1 = 0.0, s2 = 0.0, cnt = 0.0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = g[i];
        s1 += (double)v, s2 += (double)v * (double)v;
        cnt += v != 0.f ? 1.0 : 0.0;
    }
    __shared__ double red[3][4];
    s1 = wave_sum(s1), s2 = wave_sum(s2), cnt = wave_sum(cnt);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[0][wave] = s1, red[1][wave] = s2, red[2][wave] = cnt;
    __syncthreads();
    if (threadIdx.x < 3) atomicAdd(stats + threadIdx.x, red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// batched forms: blockIdx.y = grid, stats[3*g ..]
__global__ void nonzero_stats_batch_kernel(const float *__restrict__ grids, size_t n, double *stats) {
    const float *g = grids + (size_t)blockIdx.y * n;
    double s1 = 0.0, s2 = 0.0, cnt = 0.0;
    for (size_t i = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * blockDim.x * 4) {
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (i + 3 < n) {
            const float4 q = ld4(g + i);
            v[0] = q.x, v[1] = q.y, v[2] = q.z, v[3] = q.w;
        } else {
            for (size_t j = i; j < n; ++j) v[j - i] = g[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) s1 += (double)v[j], s2 += (double)v[j] * (double)v[j], cnt += v[j] != 0.f ? 1.0 : 0.0;
    }
    __shared__ double red[3][4];
    s1 = wave_sum(s1), s2 = wave_sum(s2), cnt = wave_sum(cnt);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[0][wave] = s1, red[1][wave] = s2, red[2][wave] = cnt;
    __syncthreads();
    if (threadIdx.x < 3)
        atomicAdd(stats + 3 * blockIdx.y + threadIdx.x, red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]);
}

__global__ void normalize_nonzero_batch_kernel(float *__restrict__ grids, size_t n, const double *__restrict__ stats) {
    float *g = grids + (size_t)blockIdx.y * n;
    const double cnt = stats[3 * blockIdx.y + 2];
    if (cnt == 0.0) return;
    const float mean = (float)(stats[3 * blockIdx.y] / cnt);
    const float sd = sqrtf((float)(stats[3 * blockIdx.y + 1] / cnt) - mean * mean);
    if (!(sd > 0.f)) return;      // a single distinct nonzero value: left unchanged (event_dataset.py:150 `if stddev > 0`)
    for (size_t i = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * blockDim.x * 4) {
        if (i + 3 < n) {
            float4 q = ld4(g + i);
            q.x = q.x != 0.f ? (q.x - mean) / sd : 0.f, q.y = q.y != 0.f ? (q.y - mean) / sd : 0.f;
            q.z = q.z != 0.f ? (q.z - mean) / sd : 0.f, q.w = q.w != 0.f ? (q.w - mean) / sd : 0.f;
            st4(g + i, q);
        } else {
            for (size_t j = i; j < n; ++j) g[j] = g[j] != 0.f ? (g[j] - mean) / sd : 0.f;
        }
    }
}

__global__ void normalize_nonzero_kernel(float *__restrict__ g, size_t n, const double *__restrict__ stats) {
    const double cnt = stats[2];
    if (cnt == 0.0) return;
    const float mean = (float)(stats[0] / cnt);
    const float sd = sqrtf((float)(stats[1] / cnt) - mean * mean);
    if (!(sd > 0.f)) return;      // a single distinct nonzero value: left unchanged (event_dataset.py:150 `if stddev > 0`;
                                  // EventPreprocessor, event_tensor_utils.py:64-66, would turn the whole grid into NaN)
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = g[i];
        g[i] = v != 0.f ? (v - mean) / sd : 0.f;
    }
}

}  // namespace ramnet

using namespace ramnet;

static double *si_scratch(hipStream_t st);      // (defined with the other library-owned scratch buffers below)

extern "C" int ramnet_si_loss_fwd(const float *pred, const float *target, size_t n, float weight, float lambda, double *stats, float *loss, void *stream) {
    RAMNET_CHECK_ARG(pred && target && stats && loss && n > 0);
    hipStream_t st = (hipStream_t)stream;
    int g = grid_for(n / 4 + 1, 1024);               // 1024-thread workgroups: a few loads per thread, few arrivals on the three sums
    if (g > 256) g = 256;
    if (double *part = si_scratch(st)) {             // one launch, no zero-fill (NULL inside a stream capture before the scratch exists)
        if (g > 64) g = 64;                          // (64 workgroups: 11.2 us in rocprofv3 against 13.2 with 256, 12.4 with 128, 11.7 with 32)
        hipLaunchKernelGGL(si_stats_fold_kernel<false>, dim3(g), dim3(1024), 0, st, pred, target, n, weight, lambda, stats, loss, part,
                           reinterpret_cast<unsigned long long *>(part + 3 * 256));
        RAMNET_LAUNCH_CHECK();
        return 0;
    }
    RAMNET_HIP(hipMemsetAsync(stats, 0, 4 * sizeof(double), st));
    hipLaunchKernelGGL(si_stats_kernel<false>, dim3(g), dim3(1024), 0, st, pred, target, n, weight, lambda, stats, loss);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int ramnet_si_log_loss_fwd(const float *pred, const float *target, size_t n, float lambda, double *stats, float *loss, void *stream) {
    RAMNET_CHECK_ARG(pred && target && stats && loss && n > 0);
    hipStream_t st = (hipStream_t)stream;
    int g = grid_for(n / 4 + 1, 1024);
    if (g > 256) g = 256;
    RAMNET_HIP(hipMemsetAsync(stats, 0, 4 * sizeof(double), st));
    hipLaunchKernelGGL(si_stats_kernel<true>, dim3(g), dim3(1024), 0, st, pred, target, n, 1.0f, lambda, stats, loss);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int ramnet_si_log_loss_bwd(const float *pred, const float *target, size_t n, float lambda, const double *stats,
                                      const float *gscale, float *dpred, void *stream) {
    RAMNET_CHECK_ARG(pred && target && stats && dpred && n > 0);
    hipLaunchKernelGGL(si_bwd_kernel<true>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, pred, target, n, 1.0f, lambda, stats, gscale, dpred);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int ramnet_mse_loss_fwd(const float *pred, const float *target, int B, int H, int W, int half, double *stats, float *loss, void *stream) {
    RAMNET_CHECK_ARG(pred && target && stats && loss && B > 0 && H > 0 && W > 0 && (!half || (H >= 2 && W >= 2)));
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)B * (half ? H / 2 : H) * (half ? W / 2 : W);
    int g = grid_for(n, 1024);
    if (g > 256) g = 256;
    RAMNET_HIP(hipMemsetAsync(stats, 0, 4 * sizeof(double), st));
    if (half) hipLaunchKernelGGL(mse_stats_kernel<true>, dim3(g), dim3(1024), 0, st, pred, target, B, H, W, stats, loss);
    else hipLaunchKernelGGL(mse_stats_kernel<false>, dim3(g), dim3(1024), 0, st, pred, target, B, H, W, stats, loss);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int ramnet_mse_loss_bwd(const float *pred, const float *target, int B, int H, int W, int half, const double *stats,
                                   const float *gscale, float *dpred, void *stream) {
    RAMNET_CHECK_ARG(pred && target && stats && dpred && B > 0 && H > 0 && W > 0);
    const size_t n = (size_t)B * H * W;
    if (half) hipLaunchKernelGGL(mse_bwd_kernel<true>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, pred, target, B, H, W, stats, gscale, dpred);
    else hipLaunchKernelGGL(mse_bwd_kernel<false>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, pred, target, B, H, W, stats, gscale, dpred);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int ramnet_si_loss_from_stats(const double *stats, float weight, float lambda, float *loss, void *stream) {
    RAMNET_CHECK_ARG(stats && loss);
    hipLaunchKernelGGL(si_from_stats_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, stats, weight, lambda, loss);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int ramnet_si_loss_bwd(const float *pred, const float *target, size_t n, float weight, float lambda, const double *stats,
                                  const float *gscale, float *dpred, void *stream) {
    RAMNET_CHECK_ARG(pred && target && stats && dpred && n > 0);
    hipLaunchKernelGGL(si_bwd_kernel<false>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, pred, target, n, weight, lambda, stats, gscale, dpred);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int ramnet_depth_metrics(const float *pred, const float *target, size_t n, float clip_distance, float reg_factor,
                                    float cutoff, double *out11, void *stream) {
    RAMNET_CHECK_ARG(pred && target && out11 && n > 0 && clip_distance > 0.f);
    hipStream_t st = (hipStream_t)stream;
    RAMNET_HIP(hipMemsetAsync(out11, 0, 11 * sizeof(double), st));
    hipLaunchKernelGGL(depth_metrics_kernel, dim3(grid_for(n)), dim3(256), 0, st, pred, target, n, clip_distance, reg_factor, cutoff, out11);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t ramnet_msg_workspace_elems(int B, int H, int W, int num_scales) {
    MsgArgs a;
    size_t total = 0;
    return msg_args(B, H, W, num_scales, a, total) ? 0 : total;
}

extern "C" int ramnet_msg_loss_fwd(const float *pred, const float *target, int B, int H, int W, int num_scales, float *ws,
                                   double *stats, float *loss, void *stream) {
    RAMNET_CHECK_ARG(pred && target && ws && stats && loss && B > 0);
    MsgArgs a;
    size_t total;
    RAMNET_CHECK_ARG(msg_args(B, H, W, num_scales, a, total) == 0);
    hipStream_t st = (hipStream_t)stream;
    RAMNET_HIP(hipMemsetAsync(stats, 0, 2 * num_scales * sizeof(double), st));
    hipLaunchKernelGGL(msg_pool_kernel, dim3(grid_for((size_t)B * H * W)), dim3(256), 0, st, pred, target, ws, a);
    hipLaunchKernelGGL(msg_stats_kernel, dim3(grid_for((size_t)B * H * W)), dim3(256), 0, st, ws, a, stats);
    hipLaunchKernelGGL(msg_finalize_kernel, dim3(1), dim3(1), 0, st, stats, num_scales, B, loss);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int ramnet_msg_loss_bwd(const float *ws, const double *stats, const float *gscale, int B, int H, int W, int num_scales,
                                   float *dws, float *dpred, void *stream) {
    RAMNET_CHECK_ARG(ws && stats && dws && dpred && B > 0);
    MsgArgs a;
    size_t total;
    RAMNET_CHECK_ARG(msg_args(B, H, W, num_scales, a, total) == 0);
    hipStream_t st = (hipStream_t)stream;
    RAMNET_HIP(hipMemsetAsync(dws, 0, total * sizeof(float), st));
    hipLaunchKernelGGL(msg_bwd_scatter_kernel, dim3(grid_for((size_t)B * H * W)), dim3(256), 0, st, ws, dws, a, stats, gscale);
    hipLaunchKernelGGL(msg_bwd_gather_kernel, dim3(grid_for((size_t)B * H * W)), dim3(256), 0, st, dws, a, dpred);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

// rows of a band: as many as 160 KB of LDS hold for all bins beside the hit queue (0: not even one row fits, or a launch too small
// to fill the chip -> global-atomic form)
static int voxel_band_rows(int bins, int W, int H, int n_grids, int queue_ints, int min_grids = 16) {
    // Row bands resolve the votes in LDS and write every cell once (no global atomics, no zero-fill), but every band of a grid
    // has to find ITS events in that grid's list.  Walking the list itself (6.4 MB per workgroup, 12 bands: 3 GB of L2 fills per
    // 8 M events) measured 464-514 us against 858 us for the atomic form; with a one-byte band id per event written by a first
    // pass, a band walks 200 KB and fetches whole events for its hits only.  A launch of a few grids (batch-1 streams: 5 grids =
    // 60-75 workgroups) cannot fill the chip either way (283 vs 99 us) -> atomic form.
    if (n_grids < min_grids) return 0;
    const long long cap = (160 * 1024 - 512) / 4 - (queue_ints + 2);
    long long rows = cap / ((long long)bins * W);
    if (rows > H) rows = H;
    if (rows > 0 && (H + rows - 1) / rows > 254) return 0;          // band ids are bytes
    return (int)rows;
}

// Partial sums + ticket of si_stats_fold_kernel: [256][3] doubles and one zeroed counter per (device, stream); never allocated while
// the stream is capturing (the two-launch form is used then).
static double *si_scratch(hipStream_t st) {
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, double *> bufs;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    // never handed to a CAPTURING stream, existing buffer or not: a hipGraph would bake the pointer and the self-resetting ticket in, and
    // graphs replayed concurrently (torch's captures share one stream handle) would share them — captures use memset + atomics
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    auto it = bufs.find({dev, st});
    if (it != bufs.end()) return it->second;
    void *p = nullptr;
    const size_t bytes = (3 * 256 + 1) * sizeof(double);
    if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, bytes) != hipSuccess) { (void)hipFree(p); return nullptr; }
    bufs[{dev, st}] = static_cast<double *>(p);
    return static_cast<double *>(p);
}

// Scratch for the band ids (one byte per event): owned by the library, one buffer per (device, stream), grown on demand —
// hipMalloc synchronises, so never while the stream is capturing (the id-less walk is used then).
static unsigned char *voxel_id_scratch(hipStream_t st, size_t bytes) {
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, std::pair<unsigned char *, size_t>> bufs;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;      // (a capture never sees library scratch: it could be freed / regrown under the graph)
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    auto &b = bufs[{dev, st}];
    if (b.second >= bytes) return b.first;
    if (b.first) {
        if (hipStreamSynchronize(st) != hipSuccess) return nullptr;      // launches that still read the old buffer
        (void)hipFree(b.first);
        b = {nullptr, 0};
    }
    void *p = nullptr;
    const size_t want = bytes + bytes / 4;
    if (hipMalloc(&p, want) != hipSuccess) return nullptr;
    b = {static_cast<unsigned char *>(p), want};
    return b.first;
}

static int launch_voxel_bands(const double *events, const long long *offsets, long long n_single, int n_grids, size_t max_events,
                              int bins, int W, int H, float *grids, hipStream_t st) {
    if (offsets != nullptr && max_events > 0 && max_events < (1u << 30) && W <= 32767 && H <= 32767 && g_opt_voxel_sorted) {      // sorted form
        const int nchunks = (int)((max_events + VS_CH - 1) / VS_CH);
        const int rows = voxel_band_rows(2 * bins, W, H, n_grids, 2 * nchunks + 16, 2);    // (8-byte cells)
        const int nbands = rows > 0 ? cdiv(H, rows) : 0;
        const long long rstride = (long long)nchunks * VS_CH;
        const size_t rec_bytes = (size_t)n_grids * rstride * sizeof(int4), tab_bytes = (size_t)n_grids * nchunks * (nbands + 1) * sizeof(int);
        unsigned char *scratch = rows > 0 ? voxel_id_scratch(st, rec_bytes + tab_bytes) : nullptr;
        if (scratch) {
            int4 *rec = reinterpret_cast<int4 *>(scratch);
            int *tab = reinterpret_cast<int *>(scratch + rec_bytes);
            hipLaunchKernelGGL(voxel_sort_chunks_kernel, dim3(nchunks, n_grids), dim3(1024), 0, st, events, offsets, bins, W, H, rows, nbands, rec, rstride,
                               tab, nchunks);
            RAMNET_FULL_LDS(voxelize_sorted_kernel);
            const size_t lds = (size_t)bins * rows * W * sizeof(long long) + (2 * nchunks + 16) * sizeof(int);
            hipLaunchKernelGGL(voxelize_sorted_kernel, dim3(8 * cdiv(n_grids, 8) * nbands), dim3(1024), lds, st, events, offsets, n_grids, bins, W, H,
                               rows, grids, rec, rstride, tab, nchunks);
            RAMNET_LAUNCH_CHECK();
            return 0;
        }
    }
    const long long stride = (long long)((max_events + 15) / 16 * 16);
    const int rows_ids = voxel_band_rows(bins, W, H, n_grids, 8192);
    unsigned char *ids = rows_ids > 0 && max_events > 0 ? voxel_id_scratch(st, (size_t)stride * n_grids) : nullptr;
    if (ids) {
        int gx = (int)((max_events + 255) / 256);
        const int cap = (2048 * 4 + n_grids - 1) / n_grids;
        if (gx > cap) gx = cap;
        hipLaunchKernelGGL(voxel_band_ids_kernel, dim3(gx, n_grids), dim3(256), 0, st, events, offsets, n_single, H, rows_ids, ids, stride);
        RAMNET_FULL_LDS(voxelize_bands_kernel<true>);
        const size_t lds = ((size_t)bins * rows_ids * W + 8192 + 2) * sizeof(float);
        hipLaunchKernelGGL(voxelize_bands_kernel<true>, dim3(8 * cdiv(n_grids, 8) * cdiv(H, rows_ids)), dim3(1024), lds, st, events, offsets,
                           n_single, n_grids, bins, W, H, rows_ids, grids, ids, stride);
        RAMNET_LAUNCH_CHECK();
        return 0;
    }
    const int rows = voxel_band_rows(bins, W, H, n_grids, 4096);
    if (rows <= 0) return -1;                                          // caller falls back to the atomic form
    RAMNET_FULL_LDS(voxelize_bands_kernel<false>);
    const size_t lds = ((size_t)bins * rows * W + 4096 + 2) * sizeof(float);
    hipLaunchKernelGGL(voxelize_bands_kernel<false>, dim3(8 * cdiv(n_grids, 8) * cdiv(H, rows)), dim3(1024), lds, st, events, offsets, n_single,
                       n_grids, bins, W, H, rows, grids, nullptr, 0);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int ramnet_voxelize(const double *events, size_t n_events, int bins, int W, int H, float *grid, void *stream) {
    RAMNET_CHECK_ARG(grid && bins > 0 && W > 0 && H > 0);
    hipStream_t st = (hipStream_t)stream;
    RAMNET_HIP(hipMemsetAsync(grid, 0, (size_t)bins * W * H * sizeof(float), st));
    if (n_events == 0) return 0;
    RAMNET_CHECK_ARG(events != nullptr);
    hipLaunchKernelGGL(voxelize_kernel, dim3(grid_for(n_events)), dim3(256), 0, st, events, n_events, bins, W, H, grid);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int ramnet_voxelize_batch(const double *events, const long long *offsets, int n_grids, size_t max_events, int bins, int W,
                                     int H, float *grids, void *stream) {
    RAMNET_CHECK_ARG(grids && offsets && n_grids > 0 && n_grids <= 65535 && bins > 0 && W > 0 && H > 0);
    hipStream_t st = (hipStream_t)stream;
    // (sorted form from 2 lists up — 5 lists of 200 k events, the batch-1 stream: 33 us against 104 us for the global-atomic form; one list:
    // 33 against 27 us —, the older row-band forms behind it from 16)
    if (n_grids >= 2 && (events != nullptr || max_events == 0)) {
        const int rc = launch_voxel_bands(events, offsets, 0, n_grids, max_events, bins, W, H, grids, st);
        if (rc >= 0) return rc;
    }
    RAMNET_HIP(hipMemsetAsync(grids, 0, (size_t)n_grids * bins * W * H * sizeof(float), st));
    if (max_events == 0) return 0;
    RAMNET_CHECK_ARG(events != nullptr);
    int gx = (int)((max_events + 255) / 256);
    const int cap = (2048 * 4 + n_grids - 1) / n_grids;       // ~32 workgroups per CU over the whole launch
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(voxelize_batch_kernel, dim3(gx, n_grids), dim3(256), 0, st, events, offsets, bins, W, H, grids);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int ramnet_normalize_nonzero_batch(float *grids, int n_grids, size_t n, double *scratch, void *stream) {
    RAMNET_CHECK_ARG(grids && scratch && n > 0 && n % 4 == 0 && n_grids > 0 && n_grids <= 65535);
    hipStream_t st = (hipStream_t)stream;
    RAMNET_HIP(hipMemsetAsync(scratch, 0, (size_t)3 * n_grids * sizeof(double), st));
    int gx = (int)((n / 4 + 255) / 256);
    const int cap = (2048 * 4 + n_grids - 1) / n_grids;
    if (gx > cap) gx = cap;
    hipLaunchKernelGGL(nonzero_stats_batch_kernel, dim3(gx, n_grids), dim3(256), 0, st, grids, n, scratch);
    hipLaunchKernelGGL(normalize_nonzero_batch_kernel, dim3(gx, n_grids), dim3(256), 0, st, grids, n, scratch);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int ramnet_voxel_indices(const double *events, size_t n_events, int bins, int W, int H, long long *idx_left, long long *idx_right, void *stream) {
    RAMNET_CHECK_ARG(events && idx_left && idx_right && n_events > 0 && bins > 0 && W > 0 && H > 0);
    hipLaunchKernelGGL(voxel_indices_kernel, dim3(grid_for(n_events)), dim3(256), 0, (hipStream_t)stream, events, n_events, bins, W, H, idx_left, idx_right);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int ramnet_normalize_nonzero(float *grid, size_t n, double *scratch, void *stream) {
    RAMNET_CHECK_ARG(grid && scratch && n > 0);
    hipStream_t st = (hipStream_t)stream;
    RAMNET_HIP(hipMemsetAsync(scratch, 0, 3 * sizeof(double), st));
    hipLaunchKernelGGL(nonzero_stats_kernel, dim3(grid_for(n)), dim3(256), 0, st, grid, n, scratch);
    hipLaunchKernelGGL(normalize_nonzero_kernel, dim3(grid_for(n)), dim3(256), 0, st, grid, n, scratch);
    RAMNET_LAUNCH_CHECK();
    return 0;
}
