// HBM-bound kernels of the RAM-Net path: layout packing, the 1x1 prediction head, gate
// backward formulas, the adjoint of the bilinear upsample.  All are float4-vectorised grid-stride
// loops (coalesced 16 B per lane, 1 KiB per wavefront instruction).
#include <stdarg.h>
#include <mutex>
#include <unordered_map>
#include <unordered_set>
#include <stdlib.h>
#include <string.h>
#include "common.hpp"

namespace ramnet {

static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static thread_local char g_kernel[96] = "";
void note_kernel(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
    va_end(ap);
}

hipError_t allow_full_lds(const void *kernel) {
    // the attribute is scoped to the CURRENT device: key the cache on (device, kernel) so that a process driving several GPUs
    // opts every one of them in
    static std::mutex mu;
    static std::unordered_set<unsigned long long> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long key = ((unsigned long long)(uintptr_t)kernel << 8) ^ (unsigned long long)(dev & 0xff);
    std::lock_guard<std::mutex> lock(mu);
    if (done.count(key)) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) done.insert(key);
    return e;
}

}  // namespace ramnet

// `to` waits for everything enqueued on `from` so far — hipEventRecord + hipStreamWaitEvent on ONE reused event per (thread, device): a wait
// refers to the record that precedes it, so the event can be re-recorded at once.  The host cost of forking the backward-weights stream off
// the backward-data chain ~400 times per training step (torch: a Stream object, an Event object and a context manager per fork: ~30 us).
extern "C" int ramnet_stream_fork(void *from, void *to) {
    // one event per (thread, device that OWNS `from`): the current device need not be the streams' (wgrad_side passes the side stream of its
    // operands' device), and an event recorded on a stream of another device is an invalid-handle error
    static thread_local std::unordered_map<int, hipEvent_t> ev;
    int dev = 0, cur = 0;
    RAMNET_HIP(hipGetDevice(&cur));
    dev = cur;
    if (from && hipStreamGetDevice((hipStream_t)from, &dev) != hipSuccess) (void)hipGetLastError(), dev = cur;
    hipEvent_t &e = ev[dev];
    if (!e) {
        if (dev != cur) RAMNET_HIP(hipSetDevice(dev));
        const hipError_t ce = hipEventCreateWithFlags(&e, hipEventDisableTiming);
        if (dev != cur) RAMNET_HIP(hipSetDevice(cur));
        RAMNET_HIP(ce);
    }
    RAMNET_HIP(hipEventRecord(e, (hipStream_t)from));
    RAMNET_HIP(hipStreamWaitEvent((hipStream_t)to, e, 0));
    return 0;
}

namespace ramnet {

static inline int grid_for(size_t n_items, int block = 256) {
    size_t g = (n_items + block - 1) / block;
    if (g > 256 * 8) g = 256 * 8;   // 256 CUs x 8 workgroups, grid-stride the rest
    if (g < 1) g = 1;
    return (int)g;
}

// ------------------------------------------------------------------------------------------ layout
__global__ void nchw_to_nhwc_pad_kernel(const float *__restrict__ src, float *__restrict__ dst, int B, int C, int HW, int Cpad) {
    const size_t npix = (size_t)B * HW;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / HW, s = i - b * HW;
        const float *in = src + b * C * HW + s;
        float *out = dst + i * Cpad;
        for (int c = 0; c < Cpad; c += 4) {
            float4 v;
            v.x = c + 0 < C ? in[(size_t)(c + 0) * HW] : 0.f;
            v.y = c + 1 < C ? in[(size_t)(c + 1) * HW] : 0.f;
            v.z = c + 2 < C ? in[(size_t)(c + 2) * HW] : 0.f;
            v.w = c + 3 < C ? in[(size_t)(c + 3) * HW] : 0.f;
            st4(out + c, v);
        }
    }
}

// Full-frame mode (utils/inference_utils.py:287-314 CropParameters: ReflectionPad2d to the next multiple of 2^num_encoders): the model
// input [B][C][H][W] is reflect-padded to [Hc][Wc] (top / left = ceil of half the excess) WHILE it is repacked — NHWC with channels
// zero-padded to Cpad (nhwc = 1, the model's input repack) or NCHW (CropParameters.pad).  Reflection without the border pixel:
// k < 0 -> -k, k >= n -> 2 (n - 1) - k (torch.nn.ReflectionPad2d).
__global__ void reflect_pad_kernel(const float *__restrict__ src, float *__restrict__ dst, int B, int C, int H, int W, int Cpad, int top, int left,
                                   int Hc, int Wc, int nhwc) {
    const size_t npix = (size_t)B * Hc * Wc;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % Wc), y = (int)((i / Wc) % Hc);
        const size_t b = i / ((size_t)Wc * Hc);
        int sy = y - top, sx = x - left;
        sy = sy < 0 ? -sy : (sy >= H ? 2 * (H - 1) - sy : sy);
        sx = sx < 0 ? -sx : (sx >= W ? 2 * (W - 1) - sx : sx);
        const float *in = src + (b * C * H + sy) * (size_t)W + sx;
        if (nhwc) {
            float *out = dst + i * Cpad;
            for (int c = 0; c < Cpad; c += 4) {
                float4 v;
                v.x = c + 0 < C ? in[(size_t)(c + 0) * H * W] : 0.f;
                v.y = c + 1 < C ? in[(size_t)(c + 1) * H * W] : 0.f;
                v.z = c + 2 < C ? in[(size_t)(c + 2) * H * W] : 0.f;
                v.w = c + 3 < C ? in[(size_t)(c + 3) * H * W] : 0.f;
                st4(out + c, v);
            }
        } else {
            for (int c = 0; c < C; ++c) dst[((b * C + c) * Hc + y) * (size_t)Wc + x] = in[(size_t)c * H * W];
        }
    }
}

// OIHW -> [tap][chunk][n][16].  transposed: reduce over O (backward-data), outputs = I.
__global__ void pack_weight_kernel(const float *__restrict__ w, float *__restrict__ wp, int Cout, int Cin, int T,
                                   int transposed, int gates, int R, int N, int nchunks, int NPad, size_t total) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ck = (int)(i % CK);
        size_t j = i / CK;
        const int n = (int)(j % NPad);
        j /= NPad;
        const int chunk = (int)(j % nchunks);
        const int t = (int)(j / nchunks);
        const int r = chunk * CK + ck;
        int no = n;
        bool ok = r < R;
        if (gates > 1) {   // packed n = (channel block of 32, gate, channel in block) -> original gate*C + channel
            const int C = N / gates, blk = n / (32 * gates), g = (n / 32) % gates, ch = blk * 32 + (n % 32);
            ok = ok && ch < C;
            no = g * C + ch;
        } else {
            ok = ok && n < N;
        }
        float v = 0.f;
        if (ok) v = transposed ? w[((size_t)r * Cin + no) * T + t] : w[((size_t)no * Cin + r) * T + t];
        wp[i] = v;
    }
}

// ws [T][CinWs][CoutWs] -> grad OIHW [Cout][Cin][T] (+=)
__global__ void unpack_wgrad_kernel(const float *__restrict__ ws, float *__restrict__ g, int Cout, int Cin, int CinWs, int CoutWs,
                                    int n_off, int T, size_t total) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int t = (int)(i % T);
        const size_t j = i / T;
        const int c = (int)(j % Cin), n = (int)(j / Cin);
        g[i] += ws[((size_t)t * CinWs + c) * CoutWs + n_off + n];
    }
}

// ------------------------------------------------------------------------------------------ simple maps
__global__ void relu_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ y, float *__restrict__ dx, size_t n) {
    const size_t n4 = n / 4, stride = (size_t)gridDim.x * blockDim.x, i0 = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (size_t i = i0; i < n4; i += stride) {
        const float4 g = ld4(dy + 4 * i), v = ld4(y + 4 * i);
        st4(dx + 4 * i, make_float4(v.x > 0.f ? g.x : 0.f, v.y > 0.f ? g.y : 0.f, v.z > 0.f ? g.z : 0.f, v.w > 0.f ? g.w : 0.f));
    }
    for (size_t i = n4 * 4 + i0; i < n; i += stride) dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}

__global__ void add_kernel(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ y, size_t n) {
    const size_t n4 = n / 4, stride = (size_t)gridDim.x * blockDim.x, i0 = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (size_t i = i0; i < n4; i += stride) st4(y + 4 * i, f4add(ld4(a + 4 * i), ld4(b + 4 * i)));
    for (size_t i = n4 * 4 + i0; i < n; i += stride) y[i] = a[i] + b[i];
}

// y[pix][0:Ca] = a[pix], y[pix][Ca:Ca+Cb] = b[pix]: channel concatenation of two NHWC tensors (UNet skip_type 'concat', unet.py:11-13)
__global__ void concat2_kernel(const float *__restrict__ a, int lda, int Ca, const float *__restrict__ b, int ldb, int Cb,
                               float *__restrict__ y, size_t npix) {
    const int q = (Ca + Cb) / 4;
    const size_t total = npix * q;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t pix = i / q;
        const int c = (int)(i - pix * q) * 4;
        st4(y + pix * (Ca + Cb) + c, c < Ca ? ld4(a + pix * lda + c) : ld4(b + pix * ldb + (c - Ca)));
    }
}

// inverse of concat2 (its gradient): a[pix] = y[pix][0:Ca], b[pix] = y[pix][Ca:Ca+Cb], both dense
__global__ void split2_kernel(const float *__restrict__ y, int ldy, int Ca, int Cb, float *__restrict__ a, float *__restrict__ b, size_t npix) {
    const int q = (Ca + Cb) / 4;
    const size_t total = npix * q;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t pix = i / q;
        const int c = (int)(i - pix * q) * 4;
        const float4 v = ld4(y + pix * ldy + c);
        if (c < Ca) st4(a + pix * Ca + c, v);
        else st4(b + pix * Cb + (c - Ca), v);
    }
}

// db[c] += sum_pixels dy[pix][c] * (mask[pix][c] > 0)   (bias gradient of a layer whose wgrad launch cannot carry it)
__global__ void bias_grad_kernel(const float *__restrict__ dy, const float *__restrict__ mask, float *__restrict__ db,
                                 size_t npix, int C) {
    const int C4 = C / 4, q = threadIdx.x % C4, rows = blockDim.x / C4, r0 = threadIdx.x / C4;
    float4 s = f4zero();
    if (r0 < rows)
        for (size_t pix = blockIdx.x * (size_t)rows + r0; pix < npix; pix += (size_t)gridDim.x * rows) {
            float4 g = ld4(dy + pix * C + q * 4);
            if (mask) {
                const float4 m = ld4(mask + pix * C + q * 4);
                g = make_float4(m.x > 0.f ? g.x : 0.f, m.y > 0.f ? g.y : 0.f, m.z > 0.f ? g.z : 0.f, m.w > 0.f ? g.w : 0.f);
            }
            s = f4add(s, g);
        }
    if (r0 < rows) {
        atomicAdd(db + q * 4 + 0, s.x);
        atomicAdd(db + q * 4 + 1, s.y);
        atomicAdd(db + q * 4 + 2, s.z);
        atomicAdd(db + q * 4 + 3, s.w);
    }
}

// ------------------------------------------------------------------------------------------ prediction head
// 8 lanes per pixel, each owns channel quads q, q+8, ...; butterfly over the 8 lanes.
template <bool SIG>
__global__ void pred_sigmoid_fwd_kernel(const float *__restrict__ x, int ldx, int C, const float *__restrict__ w,
                                        const float *__restrict__ bias, float *__restrict__ y, size_t npix) {
    const int sub = threadIdx.x & 7;
    const size_t stride = (size_t)gridDim.x * blockDim.x / 8;
    const float b0 = bias ? bias[0] : 0.f;
    for (size_t pix = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) / 8; pix < npix; pix += stride) {
        float s = 0.f;
        for (int c = sub * 4; c < C; c += 32) {
                const float4 v = ld4(x + pix * ldx + c), ww = ld4(w + c);
                s += v.x * ww.x + v.y * ww.y + v.z * ww.z + v.w * ww.w;
            }
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        if (sub == 0) y[pix] = SIG ? sigmoidf_(s + b0) : s + b0;
    }
}

__global__ void pred_sigmoid_bwd_kernel(const float *__restrict__ x, int ldx, int C, const float *__restrict__ w,
                                        const float *__restrict__ y, const float *__restrict__ dy, float *__restrict__ dx,
                                        int lddx, float *__restrict__ dw, float *__restrict__ db, size_t npix) {
    // thread (pixel slot, channel quad q = sub): fixed channel quads per thread -> private dw partials
    __shared__ float red[32 * 4 * 8 + 32];
    const int sub = threadIdx.x & 7, slot = threadIdx.x >> 3;
    const size_t stride = (size_t)gridDim.x * (blockDim.x / 8);
    float4 dwp[4] = {f4zero(), f4zero(), f4zero(), f4zero()};   // C <= 128
    float4 ww[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) ww[k] = sub * 4 + 32 * k < C ? ld4(w + sub * 4 + 32 * k) : f4zero();
    float dbp = 0.f;
    // four pixels per trip: their loads are issued back to back (the kernel is a pure stream: what limits it is bytes in flight)
    for (size_t pix0 = blockIdx.x * (size_t)(blockDim.x / 8) + slot; pix0 < npix; pix0 += 4 * stride) {
        float dz[4];
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t pix = pix0 + u * stride;
            const bool ok = pix < npix;
            const float yy = ok && y ? y[pix] : 0.f;         // y == NULL: the linear layer (no sigmoid)
            dz[u] = ok ? (y ? dy[pix] * yy * (1.0f - yy) : dy[pix]) : 0.f;
            v[u] = ok && sub * 4 < C ? ld4(x + pix * ldx + sub * 4) : f4zero();
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t pix = pix0 + u * stride;
            if (pix >= npix) break;
            if (sub == 0) dbp += dz[u];
#pragma unroll
            for (int k = 0; k < 4; ++k) {            // static register indices (C <= 128)
                const int c = sub * 4 + 32 * k;
                if (c < C) {
                    const float4 xv = k == 0 ? v[u] : ld4(x + pix * ldx + c);
                    if (dx) st4(dx + pix * lddx + c, f4scale(ww[k], dz[u]));
                    dwp[k] = f4add(dwp[k], f4scale(xv, dz[u]));
                }
            }
        }
    }
    // reduce over the 32 pixel slots of the block, then one atomic per channel
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k * 32 >= C) break;
        __syncthreads();
        st4(red + (slot * 8 + sub) * 4, dwp[k]);
        __syncthreads();
        if (threadIdx.x < 32) {
            const int c = k * 32 + threadIdx.x;
            float s = 0.f;
            for (int g = 0; g < 32; ++g) s += red[g * 32 + threadIdx.x];
            if (c < C) atomicAdd(dw + c, s);
        }
    }
    __syncthreads();
    if (sub == 0) red[slot] = dbp;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int g = 0; g < 32; ++g) s += red[g];
        if (db) atomicAdd(db, s);
    }
}

// ------------------------------------------------------------------------------------------ prediction head + scale-invariant loss
// pred = sigmoid(conv1x1(x) + b) (statenet.py:116-117, 313) AND the statistics of scale_invariant_loss (model/loss.py:6-9) of the maps it
// supervises, in the pass that produces them: the prediction is read where it is made instead of once more by a loss launch.  The batch is
// `nseg` segments of seg_pix pixels (one supervised measurement each: events4 and image of a package decoded as one chain), blockIdx.y =
// segment.  Sums in double; every workgroup leaves its three partial sums in part[segment][workgroup][3], the LAST one of a segment to
// arrive (ticket) adds them in workgroup order — bit-reproducible — and writes stats[segment] = (sum d, sum d^2, n, 0) and the loss.
struct PredSiTargets {
    const float *t[RAMNET_PRED_SI_MAX_SEGMENTS];
};

__device__ __forceinline__ double pw_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    return v;
}

constexpr int PRED_SI_THREADS = 256;       // (1024-thread workgroups, 128-256 of them: 36.7-37.6 us against 37.5 — the same floor; tools/bench_pred_si.py)
__global__ void __launch_bounds__(PRED_SI_THREADS) pred_sigmoid_si_fwd_kernel(const float *__restrict__ x, int ldx, int C, const float *__restrict__ w,
                                                                  const float *__restrict__ bias, float *__restrict__ y, size_t seg_pix,
                                                                  PredSiTargets tg, float weight, float lambda, double *__restrict__ part,
                                                                  unsigned long long *__restrict__ ticket, double *__restrict__ stats,
                                                                  float *__restrict__ loss) {
    // Eight lanes share a pixel's 128-byte channel row (one 16-byte load each per 32 channels) and take EIGHT pixels per trip: eight loads in
    // flight per lane, then a transpose-reduction (7 shuffles: lane i of the group ends up with the dot product of pixel i) so that EVERY lane
    // finishes one pixel — sigmoid, store, target, the three double-precision sums.  (Round 5 form: one pixel per trip, lane 0 of the group
    // alone behind a 3-shuffle butterfly: 7 of 8 lanes idle through the double arithmetic, one load in flight: 0.22 of the HBM peak.)
    const int sub = threadIdx.x & 7, seg = blockIdx.y;
    const size_t stride = (size_t)gridDim.x * blockDim.x, base = (size_t)seg * seg_pix;        // pixels per trip of the whole grid (8 per group)
    const float b0 = bias ? bias[0] : 0.f;
    const float *__restrict__ tgt = tg.t[seg];
    double s1 = 0.0, s2 = 0.0, cnt = 0.0;
    // (C = 32, the prediction layer of the network: one 16-byte load per lane and pixel, requested ONE TRIP AHEAD together with the lane's target
    // value, so that a trip's arithmetic runs under the next trip's memory round trip; other channel counts take the plain loop)
    const size_t first = ((blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 3) << 3;
    const bool c32 = C == 32;
    float4 xn[8];
    float tn = 0.f;
    auto request = [&](size_t p0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const size_t pj = p0 + j < seg_pix ? p0 + j : seg_pix - 1;          // (clamped: a valid address; its result is dropped below)
            xn[j] = ld4(x + (base + pj) * ldx + sub * 4);
        }
        tn = tgt[p0 + sub < seg_pix ? p0 + sub : seg_pix - 1];
    };
    const float4 w32 = c32 ? ld4(w + sub * 4) : f4zero();
    if (c32 && first < seg_pix) request(first);
    for (size_t p0 = first; p0 < seg_pix; p0 += stride) {
        float s[8];
        float tcur;
        if (c32) {
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] = xn[j].x * w32.x + xn[j].y * w32.y + xn[j].z * w32.z + xn[j].w * w32.w;
            tcur = tn;
            if (p0 + stride < seg_pix) request(p0 + stride);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const size_t pj = p0 + j < seg_pix ? p0 + j : seg_pix - 1;
                float a = 0.f;
                for (int c = sub * 4; c < C; c += 32) {
                    const float4 v = ld4(x + (base + pj) * ldx + c), ww = ld4(w + c);
                    a += v.x * ww.x + v.y * ww.y + v.z * ww.z + v.w * ww.w;
                }
                s[j] = a;
            }
            tcur = tgt[p0 + sub < seg_pix ? p0 + sub : seg_pix - 1];
        }
        float t4[4], t2[2];
#pragma unroll
        for (int j = 0; j < 4; ++j) {               // lanes with sub & 4 keep pixels 4..7, the others 0..3
            const float keep = (sub & 4) ? s[j + 4] : s[j], give = (sub & 4) ? s[j] : s[j + 4];
            t4[j] = keep + __shfl_xor(give, 4);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float keep = (sub & 2) ? t4[j + 2] : t4[j], give = (sub & 2) ? t4[j] : t4[j + 2];
            t2[j] = keep + __shfl_xor(give, 2);
        }
        const float keep = (sub & 1) ? t2[1] : t2[0], give = (sub & 1) ? t2[0] : t2[1];
        const float dot = keep + __shfl_xor(give, 1);                           // pixel p0 + sub
        const size_t pm = p0 + sub;
        if (pm < seg_pix) {
            const float yy = sigmoidf_(dot + b0);
            y[base + pm] = yy;
            const float d = yy - tcur;
            if (d == d) {
                s1 += (double)d;
                s2 += (double)d * (double)d;
                cnt += 1.0;
            }
        }
    }
    constexpr int NW = PRED_SI_THREADS / 64;
    __shared__ double red[3][NW];
    __shared__ int is_last;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    s1 = pw_wave_sum(s1), s2 = pw_wave_sum(s2), cnt = pw_wave_sum(cnt);
    if (lane == 0) red[0][wave] = s1, red[1][wave] = s2, red[2][wave] = cnt;
    __syncthreads();
    auto fold = [&](int k) {                     // the waves' sums in a fixed order (pairs of neighbours, then the pairs in sequence)
        double a = 0.0;
        for (int i = 0; i < NW; i += 2) a += red[k][i] + red[k][i + 1];
        return a;
    };
    // The partial sums travel as device-scope (sc1) buffer accesses behind a WORKGROUP-scope release and a relaxed device-scope ticket: no cache
    // maintenance (round 6: the agent-scope __threadfence() in front of the ticket wrote back the L2 of the workgroup's XCD — what the "27 ns per
    // workgroup" of the workgroup sweep was; the backward launch's join measured the same thing at 135 ns with its dx rows dirty in the cache)
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    constexpr int AUX_SC1 = 16;
    const auto prs = wino_rsrc(part + ((size_t)seg * gridDim.x) * 3, (unsigned)(gridDim.x * 3 * sizeof(double)));
    if (threadIdx.x == 0) {
        for (int k = 0; k < 3; ++k)
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, fold(k)), prs, (int)((blockIdx.x * 3 + k) * 8), 0, AUX_SC1);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        const unsigned long long arrived = __hip_atomic_fetch_add(ticket + seg, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = arrived == (unsigned long long)gridDim.x - 1;
        if (is_last) __hip_atomic_store(ticket + seg, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!is_last) return;
    double v[3] = {0.0, 0.0, 0.0};
    for (int i = threadIdx.x; i < (int)gridDim.x; i += blockDim.x) {       // (fixed assignment of partials to threads: a fixed order)
        double t[3];
        for (int k = 0; k < 3; ++k) t[k] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(prs, (i * 3 + k) * 8, 0, AUX_SC1));
        for (int k = 0; k < 3; ++k) v[k] += t[k];
    }
    for (int k = 0; k < 3; ++k) v[k] = pw_wave_sum(v[k]);
    __syncthreads();
    if (lane == 0) red[0][wave] = v[0], red[1][wave] = v[1], red[2][wave] = v[2];
    __syncthreads();
    if (threadIdx.x == 0) {
        double S[3];
        for (int k = 0; k < 3; ++k) S[k] = fold(k), stats[seg * 4 + k] = S[k];
        stats[seg * 4 + 3] = 0.0;
        const double m = S[0] / S[2];
        loss[seg] = (float)((double)weight * (S[1] / S[2] - (double)lambda * m * m));
    }
}

// Backward of the same: the gradient w.r.t. the prediction is what arrives densely (dy, may be NULL) PLUS, per segment, the scale-invariant
// term gscale[seg] * weight * 2 (d - lambda * mean) / n over the valid pixels (si_bwd_kernel's arithmetic, formed in double) — no dense
// gradient map of the loss exists; then the sigmoid's derivative, dx = dz * w and the [C] + 1 weight / bias partials as in pred_sigmoid_bwd_kernel.
__global__ void pred_sigmoid_si_bwd_kernel(const float *__restrict__ x, int ldx, int C, const float *__restrict__ w, const float *__restrict__ y,
                                           const float *__restrict__ dy, PredSiTargets tg, const double *__restrict__ stats,
                                           const float *__restrict__ gscale, float weight, float lambda, float *__restrict__ dx, int lddx,
                                           float *__restrict__ dw, float *__restrict__ db, size_t seg_pix, float *__restrict__ part, int ldp,
                                           unsigned long long *__restrict__ ticket, int mask_x) {
    // mask_x: x is the output of a ReLU layer with no other consumer (the last decoder): dx leaves with that layer's mask applied, dx = dz w (x > 0) —
    // the layer's backward then skips its own mask pass over the full-resolution gradient and reads dx as one plain operand
    // part != NULL (round 6): every workgroup leaves its C + 1 partial sums in part[workgroup][ldp] and the LAST one to arrive (ticket) adds them
    // in a fixed order into dw / db — bit-reproducible weight and bias gradients (the bias gradient is a sum of ~1e6 terms that cancels to ~1e-6
    // of their size: with fp32 atomics its last digits depended on the arrival order) and no 33 contended atomics per workgroup, so the launch
    // can have the 1024 workgroups its loads want.  part == NULL: fp32 atomics as before.
    __shared__ float red[32 * 4 * 8 + 32];
    __shared__ int is_last;
    const int sub = threadIdx.x & 7, slot = threadIdx.x >> 3, seg = blockIdx.y;
    const size_t stride = (size_t)gridDim.x * (blockDim.x / 8), base = (size_t)seg * seg_pix;
    const float *__restrict__ tgt = tg.t[seg];
    constexpr int AUX_SC1 = 16;                                 // device-scope cache policy of the buffer accesses to `part`
    const bool row = part != nullptr;
    const int nwg = (int)(gridDim.x * gridDim.y), rowi = (int)(blockIdx.y * gridDim.x + blockIdx.x);
    const auto prs = wino_rsrc(part, part ? (unsigned)((size_t)nwg * ldp * 4) : 0u);
    const double cnt = stats[seg * 4 + 2], mean = stats[seg * 4] / cnt;
    const double s2 = 2.0 * (double)(gscale[seg] * weight) / cnt, lm = (double)lambda * mean;
    float4 dwp[4] = {f4zero(), f4zero(), f4zero(), f4zero()};   // C <= 128
    float4 ww[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) ww[k] = sub * 4 + 32 * k < C ? ld4(w + sub * 4 + 32 * k) : f4zero();
    float dbp = 0.f;
    for (size_t p0 = blockIdx.x * (size_t)(blockDim.x / 8) + slot; p0 < seg_pix; p0 += 4 * stride) {
        float dz[4];
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t p = p0 + u * stride, pix = base + p;
            const bool ok = p < seg_pix;
            const float yy = ok ? y[pix] : 0.f;
            const float d = ok ? yy - tgt[p] : 0.f;
            float g = (ok && dy) ? dy[pix] : 0.f;
            if (ok && d == d) g += (float)(s2 * ((double)d - lm));
            dz[u] = g * yy * (1.0f - yy);
            v[u] = ok && sub * 4 < C ? ld4(x + pix * ldx + sub * 4) : f4zero();
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t p = p0 + u * stride, pix = base + p;
            if (p >= seg_pix) break;
            if (sub == 0) dbp += dz[u];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = sub * 4 + 32 * k;
                if (c < C) {
                    const float4 xv = k == 0 ? v[u] : ld4(x + pix * ldx + c);
                    if (dx) {
                        float4 g = f4scale(ww[k], dz[u]);
                        if (mask_x) g.x = xv.x > 0.f ? g.x : 0.f, g.y = xv.y > 0.f ? g.y : 0.f, g.z = xv.z > 0.f ? g.z : 0.f, g.w = xv.w > 0.f ? g.w : 0.f;
                        st4(dx + pix * lddx + c, g);
                    }
                    dwp[k] = f4add(dwp[k], f4scale(xv, dz[u]));
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k * 32 >= C) break;
        __syncthreads();
        st4(red + (slot * 8 + sub) * 4, dwp[k]);
        __syncthreads();
        if (threadIdx.x < 32) {
            const int c = k * 32 + threadIdx.x;
            float s = 0.f;
            for (int g = 0; g < 32; ++g) s += red[g * 32 + threadIdx.x];
            if (c < C) {
                if (row) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, s), prs, (int)((rowi * ldp + c) * 4), 0, AUX_SC1);
                else atomicAdd(dw + c, s);
            }
        }
    }
    __syncthreads();
    if (sub == 0) red[slot] = dbp;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int g = 0; g < 32; ++g) s += red[g];
        if (row) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, s), prs, (int)((rowi * ldp + C) * 4), 0, AUX_SC1);
        else if (db) atomicAdd(db, s);
    }
    if (!part) return;
    // (device-scope stores above, a WORKGROUP-scope release = the waves wait for their stores, a relaxed device-scope ticket: no cache maintenance —
    // an agent-scope fence here writes back the L2 of the workgroup's XCD, dirty with its dx rows: 135 ns per workgroup, serialised; csrc/conv_wino.hip
    // joins its split reductions the same way)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) {
        // two levels of tickets — groups of 32 workgroups (ticket[1 + group]), then the groups' last arrivals (ticket[0]): same-address atomics with
        // a return value are served one at a time (~27 ns each: 28 us of a 75 us launch with ONE ticket for 1024 workgroups)
        const int grp = rowi >> 5, ngrp = (nwg + 31) >> 5, gsize = grp == ngrp - 1 ? nwg - 32 * (ngrp - 1) : 32;
        int last = 0;
        if (__hip_atomic_fetch_add(ticket + 1 + grp, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned long long)gsize - 1) {
            __hip_atomic_store(ticket + 1 + grp, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__hip_atomic_fetch_add(ticket, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned long long)ngrp - 1) {
                __hip_atomic_store(ticket, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = 1;
            }
        }
        is_last = last;
    }
    __syncthreads();
    if (!is_last) return;
    // the last workgroup: thread (row lane rl = tid / 16, group = tid % 16) sums rows rl, rl + 16, ... of one 16-byte column group — device-scope
    // buffer loads, 64 independent ones per thread at 1024 workgroups (relaxed atomic loads were issued one at a time: 282 us instead of 75) —,
    // then the 16 row lanes of a group are added in order 0..15: a fixed assignment and a fixed order, whatever order the workgroups arrived in
    const int ngroups = (C + 4) / 4, rl = threadIdx.x >> 4, gl = threadIdx.x & 15;
    for (int g0 = 0; g0 < ngroups; g0 += 16) {
        const int gi = g0 + gl;
        float4 acc = f4zero();
        if (gi < ngroups)
            for (int i = rl; i < nwg; i += 16 * 8) {            // eight rows requested together (a row past the buffer reads as zero), added in row order
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    v[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(prs, ((i + 16 * u) * ldp + gi * 4) * 4, 0, AUX_SC1));
#pragma unroll
                for (int u = 0; u < 8; ++u) acc = f4add(acc, v[u]);
            }
        __syncthreads();
        st4(red + (rl * 16 + gl) * 4, acc);
        __syncthreads();
        if (rl == 0 && gi < ngroups) {
            float4 t = ld4(red + gl * 4);
            for (int r = 1; r < 16; ++r) t = f4add(t, ld4(red + (r * 16 + gl) * 4));
            const float tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = gi * 4 + e;
                if (c < C) dw[c] += tv[e];
                else if (c == C && db) db[0] += tv[e];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ upsample adjoint
__device__ __forceinline__ float up2x_weight(int dst, int src, int n_src) {
    int i0, i1;
    float l1;
    up2x_coord(dst, n_src, i0, i1, l1);
    return (i0 == src ? 1.0f - l1 : 0.f) + (i1 == src ? l1 : 0.f);
}

__global__ void upsample2x_bwd_kernel(const float *__restrict__ dup, float *__restrict__ dx, int B, int H, int W, int C) {
    const int C4 = C / 4;
    const size_t total = (size_t)B * H * W * C4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        size_t j = i / C4;
        const int x = (int)(j % W);
        j /= W;
        const int y = (int)(j % H), b = (int)(j / H);
        float4 s = f4zero();
        for (int Y = 2 * y - 1; Y <= 2 * y + 2; ++Y) {
            if (Y < 0 || Y >= 2 * H) continue;
            const float wy = up2x_weight(Y, y, H);
            if (wy == 0.f) continue;
            for (int X = 2 * x - 1; X <= 2 * x + 2; ++X) {
                if (X < 0 || X >= 2 * W) continue;
                const float wgt = wy * up2x_weight(X, x, W);
                if (wgt == 0.f) continue;
                s = f4add(s, f4scale(ld4(dup + (((size_t)b * 2 * H + Y) * 2 * W + X) * C + c), wgt));
            }
        }
        st4(dx + i * 4, s);
    }
}

// ---- folded upsample-conv helpers ---------------------------------------------------------------------------------
// replicate-padded sum: out[b][i+2][j+2] = (x + skip)[clamp i][clamp j]
__device__ __forceinline__ void pad2_sum_item(const float *__restrict__ x, const float *__restrict__ skip, float *__restrict__ out, size_t i, int H,
                                              int W, int C) {
    const int C4 = C / 4, Hp = H + 4, Wp = W + 4;
    const int c = (int)(i % C4) * 4;
    size_t j = i / C4;
    const int xx = (int)(j % Wp);
    j /= Wp;
    const int yy = (int)(j % Hp), b = (int)(j / Hp);
    const int sy = min(max(yy - 2, 0), H - 1), sx = min(max(xx - 2, 0), W - 1);
    const size_t src = (((size_t)b * H + sy) * W + sx) * C + c;
    float4 v = ld4(x + src);
    if (skip) v = f4add(v, ld4(skip + src));
    st4(out + i * 4, v);
}

__global__ void pad2_sum_kernel(const float *__restrict__ x, const float *__restrict__ skip, float *__restrict__ out, int B, int H, int W, int C) {
    const size_t total = (size_t)B * (H + 4) * (W + 4) * (C / 4);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) pad2_sum_item(x, skip, out, i, H, W, C);
}

// u(r, c) = bilinear x2 upsample of (x + skip) at high-res pixel (r, c) INSIDE the image (F.interpolate, align_corners=False)
__device__ __forceinline__ float4 up2x_at(const float *__restrict__ x, const float *__restrict__ skip, int b, int H, int W, int C,
                                          int r, int c, int ch) {
    int y0, y1, x0, x1;
    float ly, lx;
    up2x_coord(r, H, y0, y1, ly);
    up2x_coord(c, W, x0, x1, lx);
    const float hy = 1.0f - ly, hx = 1.0f - lx;
    const size_t r0 = ((size_t)b * H + y0) * W, r1 = ((size_t)b * H + y1) * W;
    float4 v00 = ld4(x + (r0 + x0) * C + ch), v01 = ld4(x + (r0 + x1) * C + ch);
    float4 v10 = ld4(x + (r1 + x0) * C + ch), v11 = ld4(x + (r1 + x1) * C + ch);
    if (skip) {
        v00 = f4add(v00, ld4(skip + (r0 + x0) * C + ch)), v01 = f4add(v01, ld4(skip + (r0 + x1) * C + ch));
        v10 = f4add(v10, ld4(skip + (r1 + x0) * C + ch)), v11 = f4add(v11, ld4(skip + (r1 + x1) * C + ch));
    }
    const float4 top = f4add(f4scale(v00, hx), f4scale(v01, lx)), bot = f4add(f4scale(v10, hx), f4scale(v11, lx));
    return f4add(f4scale(top, hy), f4scale(bot, ly));
}

// Border lines of u = up2x(x + skip), unrolled along the 5 taps (im2col) for the border-correction GEMMs of the folded
// upsample-conv:  rows [2 sides][B*2W][5][C]: side 0/1 = top/bottom image row, entry (o_x, kx) = u[row][clamp(o_x + kx - 2)];
//                 cols [2 sides][B*2H][5][C]: side 0/1 = left/right image column, entry (o_y, ky) = u[o_y + ky - 2][col], 0 outside.
__device__ __forceinline__ void up2x_border_im2col_item(const float *__restrict__ x, const float *__restrict__ skip, float *__restrict__ rows,
                                                        float *__restrict__ cols, size_t i, int B, int H, int W, int C) {
    const int C4 = C / 4, H2 = 2 * H, W2 = 2 * W;
    const size_t nrows = (size_t)2 * B * W2 * 5 * C4;
    const bool isrow = i < nrows;
    size_t j = isrow ? i : i - nrows;
    const int ch = (int)(j % C4) * 4;
    j /= C4;
    const int k = (int)(j % 5);
    j /= 5;
    const int L = isrow ? W2 : H2;
    const int o = (int)(j % L);
    j /= L;
    const int b = (int)(j % B), side = (int)(j / B);
    float4 v = f4zero();
    if (isrow) {
        v = up2x_at(x, skip, b, H, W, C, side ? H2 - 1 : 0, min(max(o + k - 2, 0), W2 - 1), ch);
    } else {
        const int r = o + k - 2;
        if (r >= 0 && r < H2) v = up2x_at(x, skip, b, H, W, C, r, side ? W2 - 1 : 0, ch);
    }
    st4((isrow ? rows : cols) + (isrow ? i : i - nrows) * 4, v);
}

__global__ void up2x_border_im2col_kernel(const float *__restrict__ x, const float *__restrict__ skip, float *__restrict__ rows,
                                          float *__restrict__ cols, int B, int H, int W, int C) {
    const size_t n = (size_t)2 * B * (2 * W + 2 * H) * 5 * (C / 4);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        up2x_border_im2col_item(x, skip, rows, cols, i, B, H, W, C);
}

// Both of the above in ONE launch (they read the same two tensors and are needed by the same decoder launch): items [0, npad) are cells
// of the padded sum, the rest entries of the unrolled border lines.
__global__ void pad2_sum_im2col_kernel(const float *__restrict__ x, const float *__restrict__ skip, float *__restrict__ out, float *__restrict__ rows,
                                       float *__restrict__ cols, int B, int H, int W, int C) {
    const size_t npad = (size_t)B * (H + 4) * (W + 4) * (C / 4), n = npad + (size_t)2 * B * (2 * W + 2 * H) * 5 * (C / 4);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (i < npad) pad2_sum_item(x, skip, out, i, H, W, C);
        else up2x_border_im2col_item(x, skip, rows, cols, i - npad, B, H, W, C);
    }
}

// Adjoint of the replicate padding by 2 (ramnet_pad2_sum): dx[b][i][j] = sum of dxpad over the padded pixels that copy (i, j) —
// the pixel itself and, on the image border, the ring pixels clamped onto it.
__global__ void unpad2_fold_kernel(const float *__restrict__ dxpad, float *__restrict__ dx, int B, int H, int W, int C) {
    const int C4 = C / 4, Wp = W + 4;
    const size_t total = (size_t)B * H * W * C4;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(idx % C4) * 4;
        size_t j = idx / C4;
        const int x = (int)(j % W);
        j /= W;
        const int y = (int)(j % H), b = (int)(j / H);
        const int y0 = y == 0 ? 0 : y + 2, y1 = y == H - 1 ? H + 3 : y + 2;
        const int x0 = x == 0 ? 0 : x + 2, x1 = x == W - 1 ? W + 3 : x + 2;
        float4 acc = f4zero();
        for (int r = y0; r <= y1; ++r)
            for (int c = x0; c <= x1; ++c) acc = f4add(acc, ld4(dxpad + (((size_t)b * (H + 4) + r) * Wp + c) * C + ch));
        st4(dx + idx * 4, acc);
    }
}

// Adjoint of up2x_border_im2col as a gather: every border pixel of dx [B][H][W][C] collects (+=) the gradients of the unrolled
// border lines whose bilinear source it is.  Line position cu of a side sums S[cu] = the entries (o, k) with clamp(o + k - 2)
// = cu (rows: clamped like the forward) or o + k - 2 = cu (columns: zero outside); pixel j takes S[cu] * its bilinear weight for
// the (at most four) positions cu = 2j-1 .. 2j+2.  `part` 0 = the two border rows, 1 = the two border columns (two launches:
// the corner pixels belong to both).
__global__ void up2x_border_col2im_kernel(const float *__restrict__ rows, const float *__restrict__ cols, float *__restrict__ dx, int B,
                                          int H, int W, int C, int part) {
    const int C4 = C / 4, H2 = 2 * H, W2 = 2 * W;
    const int L = part ? H : W, L2 = 2 * L;                  // pixels / line positions along the border line
    const float *src = part ? cols : rows;
    const size_t total = (size_t)2 * B * L * C4;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(idx % C4) * 4;
        size_t t = idx / C4;
        const int j = (int)(t % L);
        t /= L;
        const int b = (int)(t % B), side = (int)(t / B);
        const float *line = src + ((size_t)(side * B + b) * L2) * 5 * C + ch;        // [L2][5][C]
        float4 acc = f4zero();
        for (int cu = max(2 * j - 1, 0); cu <= min(2 * j + 2, L2 - 1); ++cu) {
            int p0, p1;
            float l1;
            up2x_coord(cu, L, p0, p1, l1);
            const float wgt = (p0 == j ? 1.0f - l1 : 0.0f) + (p1 == j ? l1 : 0.0f);
            if (wgt == 0.0f) continue;
            float4 sum = f4zero();
            for (int k = 0; k < 5; ++k) {
                int olo = cu + 2 - k, ohi = olo;
                if (!part) {                                  // rows: positions left of 0 / right of L2-1 are clamped onto them
                    if (cu == 0) olo = 0;
                    if (cu == L2 - 1) ohi = L2 - 1;
                }
                olo = max(olo, 0), ohi = min(ohi, L2 - 1);
                for (int o = olo; o <= ohi; ++o) sum = f4add(sum, ld4(line + ((size_t)o * 5 + k) * C));
            }
            acc = f4add(acc, f4scale(sum, wgt));
        }
        const int y = part ? j : (side ? H - 1 : 0), x = part ? (side ? W - 1 : 0) : j;
        float *dst = dx + (((size_t)b * H + y) * W + x) * C + ch;
        st4(dst, f4add(ld4(dst), acc));
    }
}

// Outermost two rows / columns of the (ReLU-masked) output gradient in the layout of the border-correction GEMMs:
// rows [2 sides][B*W2][2 slots][C], cols [2 sides][B*H2][2 slots][C]  (side 0 = top / left, slot = distance into the band).
__global__ void frame_gather_kernel(const float *__restrict__ dy, const float *__restrict__ mask, float *__restrict__ rows,
                                    float *__restrict__ cols, int B, int H2, int W2, int C) {
    const int C4 = C / 4;
    const size_t nrows = (size_t)2 * B * W2 * 2 * C4, ncols = (size_t)2 * B * H2 * 2 * C4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nrows + ncols; i += (size_t)gridDim.x * blockDim.x) {
        const bool isrow = i < nrows;
        size_t j = isrow ? i : i - nrows;
        const int ch = (int)(j % C4) * 4;
        j /= C4;
        const int slot = (int)(j % 2);
        j /= 2;
        const int L = isrow ? W2 : H2;
        const int o = (int)(j % L);
        j /= L;
        const int b = (int)(j % B), side = (int)(j / B);
        const int edge = side ? (isrow ? H2 : W2) - 2 + slot : slot;
        const int oy = isrow ? edge : o, ox = isrow ? o : edge;
        const size_t src = (((size_t)b * H2 + oy) * W2 + ox) * C + ch;
        float4 v = ld4(dy + src);
        if (mask) {
            const float4 m = ld4(mask + src);
            v = make_float4(m.x > 0.f ? v.x : 0.f, m.y > 0.f ? v.y : 0.f, m.z > 0.f ? v.z : 0.f, m.w > 0.f ? v.w : 0.f);
        }
        st4((isrow ? rows : cols) + (isrow ? i : i - nrows) * 4, v);
    }
}

// ---- space-to-depth (stride-2 5x5 convolution as a 3x3 stride-1 convolution over the four input parities) ----------
// out[b][i][j][(a*2 + c)*C + ch] = x[b][2i + a][2j + c][ch];  inverse = the same map read backwards.
__global__ void space_to_depth2_kernel(const float *__restrict__ x, float *__restrict__ out, int B, int Ho, int Wo, int C, int inverse) {
    const int C4 = C / 4;
    const size_t total = (size_t)B * Ho * Wo * 4 * C4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % C4) * 4;
        size_t j = i / C4;
        const int par = (int)(j % 4);
        j /= 4;
        const int xx = (int)(j % Wo);
        j /= Wo;
        const int yy = (int)(j % Ho), b = (int)(j / Ho);
        const size_t deep = i * 4;                                                        // [b][yy][xx][par][ch]
        const size_t flat = ((((size_t)b * 2 * Ho + 2 * yy + (par >> 1)) * 2 * Wo) + 2 * xx + (par & 1)) * C + ch;
        if (inverse) st4(out + flat, ld4(x + deep));
        else st4(out + deep, ld4(x + flat));
    }
}

// ------------------------------------------------------------------------------------------ GRU / LSTM backward maps
// ur = [u | r] (2C per pixel).  dpur = [dpu | dpr].
__global__ void gru_bwd_a_kernel(const float *__restrict__ dhn, const float *__restrict__ ur, const float *__restrict__ o,
                                 const float *__restrict__ h, float *__restrict__ dpo, float *__restrict__ dpur,
                                 float *__restrict__ dh, size_t npix, int C, int ldg, int lddh) {
    const int C4 = C / 4;
    const size_t total = npix * C4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t pix = i / C4;
        const int c = (int)(i - pix * C4) * 4;
        const float4 g = ld4(dhn + pix * ldg + c), u = ld4(ur + pix * 2 * C + c), oo = ld4(o + pix * C + c);
        const float4 hh = h ? ld4(h + pix * C + c) : f4zero();
        float4 a, bq, d;
#define RN_ONE(f)                                   \
    a.f = g.f * u.f * (1.0f - oo.f * oo.f);         \
    bq.f = g.f * (oo.f - hh.f) * u.f * (1.0f - u.f); \
    d.f = g.f * (1.0f - u.f);
        RN_ONE(x) RN_ONE(y) RN_ONE(z) RN_ONE(w)
#undef RN_ONE
        st4(dpo + pix * C + c, a);
        st4(dpur + pix * 2 * C + c, bq);
        st4(dh + pix * lddh + c, d);
    }
}

// dxhr = [dx1 | d(h*r)] from the candidate conv's backward-data.  In place: second half <- dh_direct + dhr*r.
__global__ void gru_bwd_b_kernel(float *__restrict__ dxhr, const float *__restrict__ ur, const float *__restrict__ h,
                                 float *__restrict__ dpur, const float *__restrict__ dh, size_t npix, int C) {
    const int C4 = C / 4;
    const size_t total = npix * C4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t pix = i / C4;
        const int c = (int)(i - pix * C4) * 4;
        const float4 dhr = ld4(dxhr + pix * 2 * C + C + c), r = ld4(ur + pix * 2 * C + C + c), d0 = ld4(dh + pix * C + c);
        const float4 hh = h ? ld4(h + pix * C + c) : f4zero();
        float4 pr, d;
#define RN_ONE(f)                                   \
    pr.f = dhr.f * hh.f * r.f * (1.0f - r.f);       \
    d.f = d0.f + dhr.f * r.f;
        RN_ONE(x) RN_ONE(y) RN_ONE(z) RN_ONE(w)
#undef RN_ONE
        st4(dpur + pix * 2 * C + C + c, pr);
        st4(dxhr + pix * 2 * C + C + c, d);
    }
}

// gates = activated [i|f|o|g] (4C per pixel), natural order.
__global__ void lstm_bwd_kernel(const float *__restrict__ gates, const float *__restrict__ cprev, const float *__restrict__ cnew,
                                const float *__restrict__ dhn, const float *__restrict__ dcn, float *__restrict__ dpre,
                                float *__restrict__ dcprev, size_t npix, int C) {
    const int C4 = C / 4;
    const size_t total = npix * C4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t pix = i / C4;
        const int c = (int)(i - pix * C4) * 4;
        const float *gp = gates + pix * 4 * C + c;
        const float4 gi = ld4(gp), gf = ld4(gp + C), go = ld4(gp + 2 * C), gc = ld4(gp + 3 * C);
        const float4 cp = cprev ? ld4(cprev + pix * C + c) : f4zero(), cn = ld4(cnew + pix * C + c);
        const float4 dh = dhn ? ld4(dhn + pix * C + c) : f4zero(), dc = dcn ? ld4(dcn + pix * C + c) : f4zero();
        float4 pi, pf, po, pg, dp;
#define RN_ONE(f)                                                   \
    {                                                               \
        const float tc = tanhf_(cn.f);                               \
        const float dct = dc.f + dh.f * go.f * (1.0f - tc * tc);    \
        po.f = dh.f * tc * go.f * (1.0f - go.f);                    \
        pf.f = dct * cp.f * gf.f * (1.0f - gf.f);                   \
        pi.f = dct * gc.f * gi.f * (1.0f - gi.f);                   \
        pg.f = dct * gi.f * (1.0f - gc.f * gc.f);                   \
        dp.f = dct * gf.f;                                          \
    }
        RN_ONE(x) RN_ONE(y) RN_ONE(z) RN_ONE(w)
#undef RN_ONE
        float *op = dpre + pix * 4 * C + c;
        st4(op, pi), st4(op + C, pf), st4(op + 2 * C, po), st4(op + 3 * C, pg);
        st4(dcprev + pix * C + c, dp);
    }
}

}  // namespace ramnet

using namespace ramnet;

extern "C" const char *ramnet_last_error(void) { return g_err; }
extern "C" const char *ramnet_last_kernel(void) { return g_kernel; }
extern "C" int ramnet_abi_version(void) { return RAMNET_ABI_VERSION; }

// Process-wide A/B options (tests and tuning runs; the environment variables they replace are gone since round 4)
namespace ramnet {
int g_opt_voxel_sorted = 1, g_opt_fold_pair = 1, g_opt_wgrad_blocks = 512, g_opt_wgrad_wino_blocks = 384, g_opt_wino_ksplit = 1, g_opt_wgrad_wino_nf = 1, g_opt_pred_si_cap = 256, g_opt_pred_si_bwd_cap = 1024;
}
extern "C" int ramnet_set_option(const char *name, int value) {
    RAMNET_CHECK_ARG(name != nullptr);
    if (!strcmp(name, "voxel_sorted")) ramnet::g_opt_voxel_sorted = value != 0;
    else if (!strcmp(name, "fold_pair")) ramnet::g_opt_fold_pair = value != 0;
    else if (!strcmp(name, "wgrad_blocks")) { RAMNET_CHECK_ARG(value >= 1); ramnet::g_opt_wgrad_blocks = value; }
    else if (!strcmp(name, "wgrad_wino_nf")) { RAMNET_CHECK_ARG(value == 1 || value == 2); ramnet::g_opt_wgrad_wino_nf = value; }
    else if (!strcmp(name, "wino_ksplit")) { RAMNET_CHECK_ARG(value >= 0 && value <= 16); ramnet::g_opt_wino_ksplit = value; }
    else if (!strcmp(name, "wgrad_wino_blocks")) { RAMNET_CHECK_ARG(value >= 1 && value <= 384); ramnet::g_opt_wgrad_wino_blocks = value; }
    else if (!strcmp(name, "pred_si_bwd_cap")) { RAMNET_CHECK_ARG(value >= 8 && value <= 65536); ramnet::g_opt_pred_si_bwd_cap = value; }
    else if (!strcmp(name, "pred_si_cap")) { RAMNET_CHECK_ARG(value >= 8 && value <= 65536); ramnet::g_opt_pred_si_cap = value; }
    else RAMNET_CHECK_ARG(!"ramnet_set_option: unknown option");
    return 0;
}
extern "C" int ramnet_get_option(const char *name) {
    if (name == nullptr) return -1;
    if (!strcmp(name, "voxel_sorted")) return ramnet::g_opt_voxel_sorted;
    if (!strcmp(name, "fold_pair")) return ramnet::g_opt_fold_pair;
    if (!strcmp(name, "wgrad_blocks")) return ramnet::g_opt_wgrad_blocks;
    if (!strcmp(name, "wgrad_wino_blocks")) return ramnet::g_opt_wgrad_wino_blocks;
    if (!strcmp(name, "wino_ksplit")) return ramnet::g_opt_wino_ksplit;
    if (!strcmp(name, "wgrad_wino_nf")) return ramnet::g_opt_wgrad_wino_nf;
    if (!strcmp(name, "pred_si_bwd_cap")) return ramnet::g_opt_pred_si_bwd_cap;
    if (!strcmp(name, "pred_si_cap")) return ramnet::g_opt_pred_si_cap;
    return -1;
}

extern "C" int ramnet_nchw_to_nhwc_pad(const float *src, float *dst, int B, int C, int H, int W, int Cpad, void *stream) {
    RAMNET_CHECK_ARG(src && dst && B > 0 && C > 0 && H > 0 && W > 0 && Cpad >= C && Cpad % 4 == 0);
    const size_t npix = (size_t)B * H * W;
    hipLaunchKernelGGL(nchw_to_nhwc_pad_kernel, dim3(grid_for(npix)), dim3(256), 0, (hipStream_t)stream, src, dst, B, C, H * W, Cpad);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int ramnet_reflect_pad(const float *src, float *dst, int B, int C, int H, int W, int Cpad, int top, int left, int Hc, int Wc, int nhwc,
                                  void *stream) {
    RAMNET_CHECK_ARG(src && dst && B > 0 && C > 0 && H > 0 && W > 0 && top >= 0 && left >= 0 && Hc >= H + top && Wc >= W + left);
    RAMNET_CHECK_ARG(top < H && Hc - H - top < H && left < W && Wc - W - left < W);          // reflection needs pad < extent
    RAMNET_CHECK_ARG(!nhwc || (Cpad >= C && Cpad % 4 == 0));
    const size_t npix = (size_t)B * Hc * Wc;
    hipLaunchKernelGGL(reflect_pad_kernel, dim3(grid_for(npix)), dim3(256), 0, (hipStream_t)stream, src, dst, B, C, H, W, Cpad, top, left, Hc, Wc, nhwc);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

static void pack_geometry(int Cout, int Cin, int transposed, int gates, int &R, int &N, int &nchunks, int &NPad) {
    R = transposed ? Cout : Cin;    // reduction extent
    N = transposed ? Cin : Cout;    // produced channels
    nchunks = cdiv(R, CK);
    NPad = gates > 1 ? gates * roundup(N / gates, 32) : roundup(N, 32);
}

extern "C" size_t ramnet_packed_weight_elems(int Cout, int Cin, int KH, int KW, int transposed, int gates) {
    int R, N, nchunks, NPad;
    pack_geometry(Cout, Cin, transposed, gates, R, N, nchunks, NPad);
    return (size_t)KH * KW * nchunks * NPad * CK;
}

extern "C" int ramnet_pack_weight(const float *w, float *wp, int Cout, int Cin, int KH, int KW, int transposed, int gates, void *stream) {
    RAMNET_CHECK_ARG(w && wp && Cout > 0 && Cin > 0 && KH > 0 && KW > 0 && KH * KW <= 64);      // (64 = the folded upsample-conv: 4 parities x 16 taps)
    RAMNET_CHECK_ARG(gates == 1 || (gates == 4 && !transposed && Cout % 4 == 0));
    int R, N, nchunks, NPad;
    pack_geometry(Cout, Cin, transposed, gates, R, N, nchunks, NPad);
    const size_t total = (size_t)KH * KW * nchunks * NPad * CK;
    hipLaunchKernelGGL(pack_weight_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, w, wp, Cout, Cin, KH * KW,
                       transposed, gates, R, N, nchunks, NPad, total);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int ramnet_unpack_wgrad(const float *ws, float *grad, int Cout, int Cin, int CinWs, int CoutWs, int n_off, int KH, int KW,
                                   void *stream) {
    RAMNET_CHECK_ARG(ws && grad && Cout > 0 && Cin > 0 && CinWs >= Cin && n_off >= 0 && CoutWs >= n_off + Cout);
    const size_t total = (size_t)Cout * Cin * KH * KW;
    hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, ws, grad, Cout, Cin, CinWs, CoutWs, n_off, KH * KW, total);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

// out [n * npix][C] = parts[0..n-1] [npix][ld] one behind the other along the batch axis (+ base): gradient of a time-batched feature
struct CatParts {
    const float *p[8];
};
__global__ void cat_batch_add_kernel(const CatParts parts, int n, size_t npix, int C, int ld, const float *__restrict__ base,
                                     const float *__restrict__ mask, float *__restrict__ out) {
    const int C4 = C / 4;
    const size_t per = npix * C4, total = per * n;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i / per);
        const size_t r = i - (size_t)k * per, pix = r / C4;
        const int c = (int)(r - pix * C4) * 4;
        const float *src = k == 0 ? parts.p[0] : k == 1 ? parts.p[1] : k == 2 ? parts.p[2] : k == 3 ? parts.p[3] : k == 4 ? parts.p[4]
                           : k == 5 ? parts.p[5] : k == 6 ? parts.p[6] : parts.p[7];
        float4 v = ld4(src + pix * ld + c);
        if (base) v = f4add(v, ld4(base + i * 4));
        if (mask) {             // the feature is a ReLU output: its gradient leaves here already masked (the consumer's loaders then read ONE operand)
            const float4 m = ld4(mask + i * 4);
            v.x = m.x > 0.f ? v.x : 0.f, v.y = m.y > 0.f ? v.y : 0.f, v.z = m.z > 0.f ? v.z : 0.f, v.w = m.w > 0.f ? v.w : 0.f;
        }
        st4(out + i * 4, v);
    }
}

static int cat_batch_add(const float *const *parts, int n, size_t npix, int C, int ld, const float *base, const float *mask, float *out, void *stream) {
    RAMNET_CHECK_ARG(parts && out && n >= 1 && n <= 8 && npix > 0 && C > 0 && C % 4 == 0 && ld >= C && ld % 4 == 0);
    CatParts q;
    for (int k = 0; k < 8; ++k) {
        q.p[k] = k < n ? parts[k] : nullptr;
        if (k < n) RAMNET_CHECK_ARG(parts[k] && ((uintptr_t)parts[k] & 15) == 0);
    }
    hipLaunchKernelGGL(cat_batch_add_kernel, dim3(grid_for((size_t)n * npix * (C / 4))), dim3(256), 0, (hipStream_t)stream, q, n, npix, C, ld, base, mask, out);
    RAMNET_LAUNCH_CHECK();
    return 0;
}
extern "C" int ramnet_cat_batch_add(const float *const *parts, int n, size_t npix, int C, int ld, const float *base, float *out, void *stream) {
    return cat_batch_add(parts, n, npix, C, ld, base, nullptr, out, stream);
}
extern "C" int ramnet_cat_batch_add_masked(const float *const *parts, int n, size_t npix, int C, int ld, const float *base, const float *mask, float *out,
                                           void *stream) {
    RAMNET_CHECK_ARG(mask != nullptr && ((uintptr_t)mask & 15) == 0);
    return cat_batch_add(parts, n, npix, C, ld, base, mask, out, stream);
}

extern "C" int ramnet_relu_bwd(const float *dy, const float *y, float *dx, size_t n, void *stream) {
    RAMNET_CHECK_ARG(dy && y && dx);
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, dy, y, dx, n);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int ramnet_bias_grad(const float *dy, const float *mask, float *db, size_t npix, int C, void *stream) {
    RAMNET_CHECK_ARG(dy && db && C > 0 && C % 4 == 0 && C <= 1024);
    hipLaunchKernelGGL(bias_grad_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, dy, mask, db, npix, C);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int ramnet_concat2(const float *a, int lda, int Ca, const float *b, int ldb, int Cb, float *y, size_t npix, void *stream) {
    RAMNET_CHECK_ARG(a && b && y && Ca > 0 && Cb > 0 && Ca % 4 == 0 && Cb % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && lda >= Ca && ldb >= Cb);
    hipLaunchKernelGGL(concat2_kernel, dim3(grid_for(npix * ((Ca + Cb) / 4))), dim3(256), 0, (hipStream_t)stream, a, lda, Ca, b, ldb, Cb, y, npix);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int ramnet_split2(const float *y, int ldy, int Ca, int Cb, float *a, float *b, size_t npix, void *stream) {
    RAMNET_CHECK_ARG(y && a && b && Ca > 0 && Cb > 0 && Ca % 4 == 0 && Cb % 4 == 0 && ldy % 4 == 0 && ldy >= Ca + Cb);
    hipLaunchKernelGGL(split2_kernel, dim3(grid_for(npix * ((Ca + Cb) / 4))), dim3(256), 0, (hipStream_t)stream, y, ldy, Ca, Cb, a, b, npix);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int ramnet_add(const float *a, const float *b, float *y, size_t n, void *stream) {
    RAMNET_CHECK_ARG(a && b && y);
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, a, b, y, n);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int ramnet_pred_sigmoid_fwd(const float *x, int ldx, int C, const float *w, const float *b, float *y, size_t npix, void *stream) {
    RAMNET_CHECK_ARG(x && w && y && C > 0 && C % 4 == 0 && ldx % 4 == 0);
    hipLaunchKernelGGL(pred_sigmoid_fwd_kernel<true>, dim3(grid_for(npix * 8)), dim3(256), 0, (hipStream_t)stream, x, ldx, C, w, b, y, npix);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

static int pred_bwd(const float *x, int ldx, int C, const float *w, const float *y, const float *dy, float *dx, int lddx, float *dw, float *db,
                    size_t npix, void *stream) {
    if (dx) RAMNET_CHECK_ARG(lddx % 4 == 0);
    int g = grid_for(npix * 8);      // every workgroup ends with 33 atomics on the SAME 33 addresses: few, fat workgroups
    if (g > 512) g = 512;
    hipLaunchKernelGGL(pred_sigmoid_bwd_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, x, ldx, C, w, y, dy, dx, lddx, dw, db, npix);
    RAMNET_LAUNCH_CHECK();
    return 0;
}
extern "C" int ramnet_pred_sigmoid_bwd(const float *x, int ldx, int C, const float *w, const float *y, const float *dy, float *dx,
                                       int lddx, float *dw, float *db, size_t npix, void *stream) {
    RAMNET_CHECK_ARG(x && w && y && dy && dw && db && C > 0 && C % 4 == 0 && C <= 128 && ldx % 4 == 0);
    return pred_bwd(x, ldx, C, w, y, dy, dx, lddx, dw, db, npix, stream);
}
static int pred_si_grid(size_t seg_pix, int nseg) {
    // Workgroups per segment: at most "pred_si_cap" (256 = one per CU) over all segments: 1.41 M pixels x 32 channels take 32.2 us with 256 or 512
    // workgroups (0.745 of the HBM peak), 35.7 with 1024, 50.3 with 128 (tools/bench_pred_si.py; with an agent-scope fence in front of every
    // workgroup's ticket it was 37.5 / 43.9 / 58.3: profiles/r06_h_tuning_notes.md sections 5 and 10) — and a count that gives every 8-lane group
    // the SAME number of 8-pixel trips (a ragged last trip left a third of the chip idle)
    const size_t groups = (seg_pix + 7) / 8;
    const size_t cap = (size_t)ramnet::g_opt_pred_si_cap / (size_t)(nseg > 0 ? nseg : 1);
    constexpr size_t GPW = PRED_SI_THREADS / 8;                 // 8-lane groups per workgroup
    const size_t trips = (groups + cap * GPW - 1) / (cap * GPW);
    size_t g = (groups + GPW * trips - 1) / (GPW * trips);
    if (g > cap) g = cap;
    return g < 1 ? 1 : (int)g;
}

constexpr int PRED_SI_BWD_LDP = 132;       // floats per workgroup row of the backward launch's partial sums (C <= 128 weights + the bias, 16-byte rows)
static size_t pred_si_bwd_tickets() { return 1 + ((size_t)ramnet::g_opt_pred_si_bwd_cap + 31) / 32; }      // [0]: the groups' last arrivals, [1 + g]: group g of 32 workgroups
static size_t pred_si_bwd_doubles() { return ((size_t)ramnet::g_opt_pred_si_bwd_cap * PRED_SI_BWD_LDP + 1) / 2; }
extern "C" size_t ramnet_pred_si_scratch_doubles(size_t seg_pix, int nseg) {
    // forward: [nseg][workgroups][3] partial sums + [nseg] tickets (zero before the first use; the kernel leaves them at zero); the backward launch
    // reuses the buffer: [workgroups][132] fp32 partial sums of the weight / bias gradients, its tickets in the LAST doubles (never touched by the forward)
    if (nseg <= 0) return 0;
    const size_t fwd = (size_t)nseg * pred_si_grid(seg_pix, nseg) * 3 + nseg, bwd = pred_si_bwd_doubles();
    return (fwd > bwd ? fwd : bwd) + pred_si_bwd_tickets();
}

extern "C" int ramnet_pred_sigmoid_si_fwd(const float *x, int ldx, int C, const float *w, const float *b, float *y, size_t seg_pix, int nseg,
                                          const float *const *targets, float weight, float lambda, double *scratch, double *stats, float *loss,
                                          void *stream) {
    RAMNET_CHECK_ARG(x && w && y && C > 0 && C % 4 == 0 && ldx % 4 == 0 && seg_pix > 0);
    RAMNET_CHECK_ARG(nseg >= 1 && nseg <= RAMNET_PRED_SI_MAX_SEGMENTS && targets && scratch && stats && loss);
    PredSiTargets tg;
    for (int i = 0; i < RAMNET_PRED_SI_MAX_SEGMENTS; ++i) tg.t[i] = i < nseg ? targets[i] : nullptr;
    for (int i = 0; i < nseg; ++i) RAMNET_CHECK_ARG(tg.t[i] != nullptr);
    const int g = pred_si_grid(seg_pix, nseg);
    hipLaunchKernelGGL(pred_sigmoid_si_fwd_kernel, dim3(g, nseg), dim3(PRED_SI_THREADS), 0, (hipStream_t)stream, x, ldx, C, w, b, y, seg_pix, tg, weight, lambda,
                       scratch, reinterpret_cast<unsigned long long *>(scratch + (size_t)nseg * g * 3), stats, loss);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int ramnet_pred_sigmoid_si_bwd(const float *x, int ldx, int C, const float *w, const float *y, const float *dy, size_t seg_pix, int nseg,
                                          const float *const *targets, const double *stats, const float *gscale, float weight, float lambda,
                                          float *dx, int lddx, float *dw, float *db, double *scratch, int mask_x, void *stream) {
    RAMNET_CHECK_ARG(x && w && y && dw && db && C > 0 && C % 4 == 0 && C <= 128 && ldx % 4 == 0 && seg_pix > 0);
    RAMNET_CHECK_ARG(nseg >= 1 && nseg <= RAMNET_PRED_SI_MAX_SEGMENTS && targets && stats && gscale);
    if (dx) RAMNET_CHECK_ARG(lddx % 4 == 0);
    PredSiTargets tg;
    for (int i = 0; i < RAMNET_PRED_SI_MAX_SEGMENTS; ++i) tg.t[i] = i < nseg ? targets[i] : nullptr;
    for (int i = 0; i < nseg; ++i) RAMNET_CHECK_ARG(tg.t[i] != nullptr);
    int g = grid_for(seg_pix * 8);
    // workgroups over all segments: "pred_si_bwd_cap" (1024: 74.5 us for 1.41 M pixels x 32 channels, 0.62 of the HBM peak; 512 87.5, 2048 79.8, 256
    // 146.8: tools/bench_pred_si.py) when the partial sums are joined through `scratch`; without it every workgroup ends with 33 fp32 atomics on the
    // SAME 33 addresses and their arrival order is the noise of the bias gradient: at most 512 then
    const int cap = scratch ? ramnet::g_opt_pred_si_bwd_cap : (ramnet::g_opt_pred_si_bwd_cap < 512 ? ramnet::g_opt_pred_si_bwd_cap : 512);
    if (g > cap / nseg) g = cap / nseg;
    if (g < 1) g = 1;
    // scratch = the forward launch's buffer (ramnet_pred_si_scratch_doubles doubles, its last one zero): fixed-order join of the partial sums
    float *part = reinterpret_cast<float *>(scratch);
    unsigned long long *ticket = scratch ? reinterpret_cast<unsigned long long *>(scratch + ramnet_pred_si_scratch_doubles(seg_pix, nseg) - pred_si_bwd_tickets()) : nullptr;
    hipLaunchKernelGGL(pred_sigmoid_si_bwd_kernel, dim3(g, nseg), dim3(256), 0, (hipStream_t)stream, x, ldx, C, w, y, dy, tg, stats, gscale, weight,
                       lambda, dx, lddx, dw, db, seg_pix, part, PRED_SI_BWD_LDP, ticket, mask_x);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int ramnet_pred_linear_fwd(const float *x, int ldx, int C, const float *w, const float *b, float *z, size_t npix, void *stream) {
    RAMNET_CHECK_ARG(x && w && z && C > 0 && C % 4 == 0 && ldx % 4 == 0);
    hipLaunchKernelGGL(pred_sigmoid_fwd_kernel<false>, dim3(grid_for(npix * 8)), dim3(256), 0, (hipStream_t)stream, x, ldx, C, w, b, z, npix);
    RAMNET_LAUNCH_CHECK();
    return 0;
}
extern "C" int ramnet_pred_linear_bwd(const float *x, int ldx, int C, const float *w, const float *dz, float *dx, int lddx, float *dw, float *db,
                                      size_t npix, void *stream) {
    RAMNET_CHECK_ARG(x && w && dz && dw && C > 0 && C % 4 == 0 && C <= 128 && ldx % 4 == 0);
    return pred_bwd(x, ldx, C, w, nullptr, dz, dx, lddx, dw, db, npix, stream);
}

extern "C" int ramnet_upsample2x_bwd(const float *dup, float *dx, int B, int H, int W, int C, void *stream) {
    RAMNET_CHECK_ARG(dup && dx && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0);
    hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(grid_for((size_t)B * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream, dup, dx, B, H, W, C);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int ramnet_pad2_sum(const float *x, const float *skip, float *out, int B, int H, int W, int C, void *stream) {
    RAMNET_CHECK_ARG(x && out && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0);
    hipLaunchKernelGGL(pad2_sum_kernel, dim3(grid_for((size_t)B * (H + 4) * (W + 4) * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, skip, out, B, H, W, C);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int ramnet_up2x_border_im2col(const float *x, const float *skip, float *rows, float *cols, int B, int H, int W, int C, void *stream) {
    RAMNET_CHECK_ARG(x && rows && cols && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0);
    const size_t n = (size_t)2 * B * (2 * W + 2 * H) * 5 * (C / 4);
    hipLaunchKernelGGL(up2x_border_im2col_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, skip, rows, cols, B, H, W, C);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int ramnet_pad2_sum_im2col(const float *x, const float *skip, float *out, float *rows, float *cols, int B, int H, int W, int C,
                                      void *stream) {
    RAMNET_CHECK_ARG(x && out && rows && cols && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0);
    const size_t n = (size_t)B * (H + 4) * (W + 4) * (C / 4) + (size_t)2 * B * (2 * W + 2 * H) * 5 * (C / 4);
    hipLaunchKernelGGL(pad2_sum_im2col_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, skip, out, rows, cols, B, H, W, C);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int ramnet_unpad2_fold(const float *dxpad, float *dx, int B, int H, int W, int C, void *stream) {
    RAMNET_CHECK_ARG(dxpad && dx && B > 0 && H >= 2 && W >= 2 && C > 0 && C % 4 == 0);
    hipLaunchKernelGGL(unpad2_fold_kernel, dim3(grid_for((size_t)B * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream, dxpad, dx, B, H, W, C);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int ramnet_up2x_border_col2im(const float *rows, const float *cols, float *dx, int B, int H, int W, int C, void *stream) {
    RAMNET_CHECK_ARG(rows && cols && dx && B > 0 && H >= 2 && W >= 2 && C > 0 && C % 4 == 0);
    for (int part = 0; part < 2; ++part)
        hipLaunchKernelGGL(up2x_border_col2im_kernel, dim3(grid_for((size_t)2 * B * (part ? H : W) * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                           rows, cols, dx, B, H, W, C, part);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int ramnet_frame_gather(const float *dy, const float *mask, float *rows, float *cols, int B, int H2, int W2, int C, void *stream) {
    RAMNET_CHECK_ARG(dy && rows && cols && B > 0 && H2 >= 4 && W2 >= 4 && C > 0 && C % 4 == 0);
    const size_t n = (size_t)2 * B * (W2 + H2) * 2 * (C / 4);
    hipLaunchKernelGGL(frame_gather_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dy, mask, rows, cols, B, H2, W2, C);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int ramnet_space_to_depth2(const float *x, float *out, int B, int H, int W, int C, int inverse, void *stream) {
    RAMNET_CHECK_ARG(x && out && B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C > 0 && C % 4 == 0);
    hipLaunchKernelGGL(space_to_depth2_kernel, dim3(grid_for((size_t)B * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, out, B, H / 2,
                       W / 2, C, inverse);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int ramnet_gru_bwd_a(const float *dhn, const float *ur, const float *o, const float *h, float *dpo, float *dpur,
                                float *dh, size_t npix, int C, int ld_dhn, void *stream) {
    RAMNET_CHECK_ARG(dhn && ur && o && dpo && dpur && dh && C % 4 == 0 && ld_dhn >= C && ld_dhn % 4 == 0);
    hipLaunchKernelGGL(gru_bwd_a_kernel, dim3(grid_for(npix * (C / 4))), dim3(256), 0, (hipStream_t)stream, dhn, ur, o, h, dpo, dpur, dh, npix, C,
                       ld_dhn, C);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

// the same with dh written through its own leading dimension: into the [.., C:] half of the [dx | dh] tensor that the candidate
// convolution's backward-data launch then completes in its epilogue (RAMNET_EPI_GRU_BWD) — stage B without a launch of its own
extern "C" int ramnet_gru_bwd_a2(const float *dhn, const float *ur, const float *o, const float *h, float *dpo, float *dpur,
                                 float *dh, size_t npix, int C, int ld_dhn, int ld_dh, void *stream) {
    RAMNET_CHECK_ARG(dhn && ur && o && dpo && dpur && dh && C % 4 == 0 && ld_dhn >= C && ld_dhn % 4 == 0 && ld_dh >= C && ld_dh % 4 == 0);
    hipLaunchKernelGGL(gru_bwd_a_kernel, dim3(grid_for(npix * (C / 4))), dim3(256), 0, (hipStream_t)stream, dhn, ur, o, h, dpo, dpur, dh, npix, C,
                       ld_dhn, ld_dh);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int ramnet_gru_bwd_b(const float *dxhr, const float *ur, const float *h, float *dpur, float *dh, size_t npix, int C, void *stream) {
    RAMNET_CHECK_ARG(dxhr && ur && dpur && dh && C % 4 == 0);
    hipLaunchKernelGGL(gru_bwd_b_kernel, dim3(grid_for(npix * (C / 4))), dim3(256), 0, (hipStream_t)stream, const_cast<float *>(dxhr), ur, h, dpur, dh, npix, C);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int ramnet_lstm_bwd(const float *gates, const float *cprev, const float *cnew, const float *dhn, const float *dcn,
                               float *dpre, float *dcprev, size_t npix, int C, void *stream) {
    RAMNET_CHECK_ARG(gates && cnew && dpre && dcprev && C % 4 == 0);
    hipLaunchKernelGGL(lstm_bwd_kernel, dim3(grid_for(npix * (C / 4))), dim3(256), 0, (hipStream_t)stream, gates, cprev, cnew, dhn, dcn, dpre, dcprev, npix, C);
    RAMNET_LAUNCH_CHECK();
    return 0;
}
