// Scale-invariant loss reductions and event -> voxel-grid binning (HBM / atomic bound).
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

// stats[0] += sum d, stats[1] += sum d^2, stats[2] += count over non-NaN d = pred - target
__global__ void si_stats_kernel(const float *__restrict__ pred, const float *__restrict__ target, size_t n, double *stats) {
    double s1 = 0.0, s2 = 0.0, cnt = 0.0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float d = pred[i] - target[i];
        if (d == d) {
            s1 += (double)d;
            s2 += (double)d * (double)d;
            cnt += 1.0;
        }
    }
    __shared__ double red[3][4];
    s1 = wave_sum(s1), s2 = wave_sum(s2), cnt = wave_sum(cnt);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[0][wave] = s1, red[1][wave] = s2, red[2][wave] = cnt;
    __syncthreads();
    if (threadIdx.x < 3) atomicAdd(stats + threadIdx.x, red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]);
}

__global__ void si_finalize_kernel(const double *stats, float weight, float lambda, float *loss) {
    const double n = stats[2], m = stats[0] / n;
    *loss = (float)((double)weight * (stats[1] / n - (double)lambda * m * m));
}

__global__ void si_bwd_kernel(const float *__restrict__ pred, const float *__restrict__ target, size_t n, float weight, float lambda,
                              const double *__restrict__ stats, const float *__restrict__ gscale, float *__restrict__ dpred) {
    // d - lambda*mean is formed in double: rounding the mean to fp32 would add the SAME offset to every pixel, and
    // gradients that are sums over pixels of this map (e.g. the last bias) cancel to ~0 and would keep only that offset.
    const double cnt = stats[2], mean = stats[0] / cnt;
    const double s2 = 2.0 * (double)((gscale ? *gscale : 1.0f) * weight) / cnt;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float d = pred[i] - target[i];
        dpred[i] = (d == d) ? (float)(s2 * ((double)d - (double)lambda * mean)) : 0.f;
    }
}

// ---------------------------------------------------------------------------------------- voxel grid
// Index arithmetic restated from events_to_voxel_grid_pytorch (utils/event_tensor_utils.py:152-180):
// float64 normalised time, floor, float32 votes pol*(1-dt) / pol*dt, `0 <= ti < bins` guards.
#pragma clang fp contract(off)
__device__ __forceinline__ void voxel_event(const double *__restrict__ ev, size_t i, size_t n, int bins, int W, int H,
                                            long long &il, float &vl, long long &ir, float &vr) {
    const double t0 = ev[0], t1 = ev[(n - 1) * 4];
    double dT = t1 - t0;
    if (dT == 0.0) dT = 1.0;
    const double ts = ((double)(bins - 1) * (ev[i * 4] - t0)) / dT;
    const long long xs = (long long)ev[i * 4 + 1], ys = (long long)ev[i * 4 + 2];
    float pol = (float)ev[i * 4 + 3];
    if (pol == 0.f) pol = -1.f;
    const double tis = floor(ts);
    const long long til = (long long)tis;
    const float dts = (float)(ts - tis);
    vl = pol * (1.0f - dts);
    vr = pol * dts;
    const bool inside = xs >= 0 && xs < W && ys >= 0 && ys < H;   // the reference would raise on an OOB index
    const long long base = xs + ys * (long long)W;
    il = (inside && tis < (double)bins && tis >= 0.0) ? base + til * (long long)W * H : -1;
    ir = (inside && (tis + 1.0) < (double)bins && tis >= 0.0) ? base + (til + 1) * (long long)W * H : -1;
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

__global__ void normalize_nonzero_kernel(float *__restrict__ g, size_t n, const double *__restrict__ stats) {
    const double cnt = stats[2];
    if (cnt == 0.0) return;
    const float mean = (float)(stats[0] / cnt);
    const float sd = sqrtf((float)(stats[1] / cnt) - mean * mean);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = g[i];
        g[i] = v != 0.f ? (v - mean) / sd : 0.f;
    }
}

}  // namespace ramnet

using namespace ramnet;

extern "C" int ramnet_si_loss_fwd(const float *pred, const float *target, size_t n, float weight, float lambda, double *stats, float *loss, void *stream) {
    RAMNET_CHECK_ARG(pred && target && stats && loss && n > 0);
    hipStream_t st = (hipStream_t)stream;
    RAMNET_HIP(hipMemsetAsync(stats, 0, 3 * sizeof(double), st));
    hipLaunchKernelGGL(si_stats_kernel, dim3(grid_for(n)), dim3(256), 0, st, pred, target, n, stats);
    hipLaunchKernelGGL(si_finalize_kernel, dim3(1), dim3(1), 0, st, stats, weight, lambda, loss);
    RAMNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int ramnet_si_loss_bwd(const float *pred, const float *target, size_t n, float weight, float lambda, const double *stats,
                                  const float *gscale, float *dpred, void *stream) {
    RAMNET_CHECK_ARG(pred && target && stats && dpred && n > 0);
    hipLaunchKernelGGL(si_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, pred, target, n, weight, lambda, stats, gscale, dpred);
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
