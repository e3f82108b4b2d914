// Issue-rate probe for v_mfma_f32_16x16x4_f32 on gfx950: how many cycles per MFMA does ONE wave (and two waves per SIMD)
// sustain when the MFMA stream is mixed with LDS reads / global loads / VALU work, the way the Winograd kernel mixes them.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int DEP>
__global__ void __launch_bounds__(256, 2) probe(const float *g, float *out, unsigned long long *cyc, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 8192; i += 256) lds[i] = g[i];
    __syncthreads();
    f32x4 acc[16][2];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i][0] = acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    float2 breg[16], areg[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) breg[i] = make_float2(g[lane + i], g[lane + 64 + i]);
    areg[0] = make_float2(g[lane], g[lane + 1]), areg[1] = make_float2(g[lane + 2], g[lane + 3]);
    float4 vv = make_float4(g[lane], g[lane + 1], g[lane + 2], g[lane + 3]);
    const float *gp = g + (size_t)lane * 2;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int pos = 0; pos < 16; ++pos) {
            float2 a0 = areg[0], a1 = areg[1];
            const float2 bb = breg[pos];
            if (MODE & 1) {   // A operands from LDS (addresses vary with pos / it so that nothing is hoisted)
                a0 = *reinterpret_cast<const float2 *>(lds + ((lane * 8 + pos * 256 + (it & 1) * 4096) & 8190));
                a1 = *reinterpret_cast<const float2 *>(lds + ((lane * 8 + pos * 256 + 128 + (it & 1) * 4096) & 8190));
            }
            __builtin_amdgcn_sched_barrier(0);
            if (DEP == 2) {   // acc0, acc1, acc0, acc1: dependent distance 2
                acc[pos][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, bb.x, acc[pos][0], 0, 0, 0);
                acc[pos][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, bb.x, acc[pos][1], 0, 0, 0);
                acc[pos][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, bb.y, acc[pos][0], 0, 0, 0);
                acc[pos][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, bb.y, acc[pos][1], 0, 0, 0);
            } else {          // distance 4: two positions interleaved
                const int p2 = pos ^ 8;
                acc[pos][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, bb.x, acc[pos][0], 0, 0, 0);
                acc[pos][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, bb.x, acc[pos][1], 0, 0, 0);
                acc[p2][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, bb.y, acc[p2][0], 0, 0, 0);
                acc[p2][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, bb.y, acc[p2][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (MODE & 2) breg[pos] = *reinterpret_cast<const float2 *>(gp + ((pos * 512 + it * 8192) & 65535));
            if (MODE & 4) {   // ~8 VALU
                vv.x = vv.x * 1.0001f + vv.y, vv.y = vv.y * 0.9999f + vv.z, vv.z = vv.z * 1.0002f + vv.w, vv.w = vv.w * 0.9998f + vv.x;
                vv.x = vv.x * 1.0001f + vv.y, vv.y = vv.y * 0.9999f + vv.z, vv.z = vv.z * 1.0002f + vv.w, vv.w = vv.w * 0.9998f + vv.x;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = vv.x + vv.y + vv.z + vv.w;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0][0] + acc[i][1][1] + acc[i][0][2] + acc[i][1][3];
    out[blockIdx.x * 256 + tid] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
}

template <int MODE, int DEP>
void run(const char *name, const float *g, float *out, unsigned long long *cyc, int blocks_per_cu) {
    const int iters = 200, nb = 256 * blocks_per_cu;
    hipLaunchKernelGGL((probe<MODE, DEP>), dim3(nb), dim3(256), 0, 0, g, out, cyc, iters);
    hipLaunchKernelGGL((probe<MODE, DEP>), dim3(nb), dim3(256), 0, 0, g, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(nb * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    printf("%-44s dep %d  %d wave/SIMD: %6.1f cycles per MFMA per wave  (pipe busy %.0f %%)\n", name, DEP, blocks_per_cu,
           s / h.size() / (iters * 64.0), 100.0 * 32.0 * blocks_per_cu / (s / h.size() / (iters * 64.0)));
}

int main() {
    float *g, *out;
    unsigned long long *cyc;
    hipMalloc(&g, 1 << 20), hipMalloc(&out, 512 * 256 * 4), hipMalloc(&cyc, 512 * 4 * 8);
    hipMemset(g, 0, 1 << 20);
    for (int w = 1; w <= 2; ++w) {
        run<0, 2>("MFMA only", g, out, cyc, w);
        run<0, 4>("MFMA only", g, out, cyc, w);
        run<1, 2>("+ A operands from LDS (2 x ds_read_b64)", g, out, cyc, w);
        run<2, 2>("+ B reload from global (dwordx2)", g, out, cyc, w);
        run<3, 2>("+ LDS + global", g, out, cyc, w);
        run<4, 2>("+ 8 VALU per 4 MFMA", g, out, cyc, w);
        run<7, 2>("+ LDS + global + VALU", g, out, cyc, w);
        run<7, 4>("+ LDS + global + VALU", g, out, cyc, w);
    }
    return 0;
}
