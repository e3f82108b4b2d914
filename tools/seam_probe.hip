// What does the seam between the two launches of a ConvGRU step cost on MI355X — as a launch boundary, and as a grid barrier inside one
// launch?  (the fused single-launch ConvGRU step of the north star needs `h*r` of ALL channel blocks of a tile before the
// candidate convolution can start, i.e. a grid-wide dependency between its two phases.)
// Phase A: every workgroup writes a 32 x 64 fp32 tile (8 KB) of a buffer U; phase B: every workgroup reads the tile that ANOTHER
// workgroup wrote (a different XCD's: workgroup b reads tile (b + 1) % grid) and writes a checksum.  Three forms:
//   two launches (stream order), one launch + XCD-hierarchical grid barrier with agent-scope release / acquire, one launch without the
//   barrier (wrong results: the floor).  G workgroups of 256 threads, G <= resident capacity.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/seam_probe.hip -o /tmp/seam_probe && /tmp/seam_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int TILE = 32 * 64;

__device__ __forceinline__ void phase_a(float *U, int b, int it) {
    float4 *dst = reinterpret_cast<float4 *>(U + (size_t)b * TILE);
    const float v = (float)(b + it);
    for (int i = threadIdx.x; i < TILE / 4; i += 256) dst[i] = make_float4(v, v + 1.f, v + 2.f, v + 3.f);
}

__device__ __forceinline__ void phase_b(const float *U, float *out, int b, int grid) {
    const float4 *src = reinterpret_cast<const float4 *>(U + (size_t)((b + 1) % grid) * TILE);
    float s = 0.f;
    for (int i = threadIdx.x; i < TILE / 4; i += 256) {
        const float4 v = src[i];
        s += v.x + v.y + v.z + v.w;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(out + b, s);
}

__global__ void __launch_bounds__(256) kern_a(float *U, int it) { phase_a(U, blockIdx.x, it); }
__global__ void __launch_bounds__(256) kern_b(const float *U, float *out) { phase_b(U, out, blockIdx.x, gridDim.x); }

// monotonic-counter grid barrier: per-XCD arrival counters, the last arriver of an XCD bumps the top counter, everybody polls the top
// generation (relaxed agent-scope loads + s_sleep); release fence before arriving, acquire fence after leaving.
__device__ __forceinline__ void grid_barrier(unsigned *ctr, unsigned gen, int grid) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int spin = 0; spin < (1 << 22) && __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen * (unsigned)grid; ++spin)
            __builtin_amdgcn_s_sleep(2);                    // (bounded: a probe must not hang the box)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

template <bool BARRIER>
__global__ void __launch_bounds__(256) kern_fused(float *U, float *out, unsigned *ctr, unsigned gen, int it) {
    phase_a(U, blockIdx.x, it);
    if (BARRIER) grid_barrier(ctr, gen, gridDim.x);
    phase_b(U, out, blockIdx.x, gridDim.x);
}

int main() {
    for (int grid : {128, 256, 512}) {
        float *U, *out;
        unsigned *ctr;
        CHECK(hipMalloc(&U, (size_t)grid * TILE * 4));
        CHECK(hipMalloc(&out, grid * 4));
        CHECK(hipMalloc(&ctr, 4));
        CHECK(hipMemset(ctr, 0, 4));
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        const int reps = 200;
        float ms[3] = {0, 0, 0};
        unsigned gen = 0;
        for (int form = 0; form < 3; ++form) {
            for (int warm = 0; warm < 2; ++warm) {
                CHECK(hipEventRecord(e0, 0));
                for (int it = 0; it < reps; ++it) {
                    if (form == 0) {
                        hipLaunchKernelGGL(kern_a, dim3(grid), dim3(256), 0, 0, U, it);
                        hipLaunchKernelGGL(kern_b, dim3(grid), dim3(256), 0, 0, U, out);
                    } else if (form == 1) {
                        hipLaunchKernelGGL(kern_fused<true>, dim3(grid), dim3(256), 0, 0, U, out, ctr, ++gen, it);
                    } else {
                        hipLaunchKernelGGL(kern_fused<false>, dim3(grid), dim3(256), 0, 0, U, out, ctr, 0u, it);
                    }
                }
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                CHECK(hipEventElapsedTime(&ms[form], e0, e1));
            }
        }
        // correctness of the barrier form: one more run, checksum of every tile
        CHECK(hipMemset(out, 0, grid * 4));
        hipLaunchKernelGGL(kern_fused<true>, dim3(grid), dim3(256), 0, 0, U, out, ctr, ++gen, 1000);
        std::vector<float> h(grid);
        CHECK(hipMemcpy(h.data(), out, grid * 4, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int b = 0; b < grid; ++b) {
            const float v = (float)((b + 1) % grid + 1000);
            const float want = (TILE / 4) * (4.f * v + 6.f);
            if (h[b] != want) ++bad;
        }
        printf("grid %4d workgroups: two launches %.2f us per step | one launch + grid barrier %.2f us | one launch, no barrier (floor, wrong) %.2f us | "
               "seam: boundary %.2f us, barrier %.2f us | barrier form stale tiles: %d\n",
               grid, 1e3 * ms[0] / reps, 1e3 * ms[1] / reps, 1e3 * ms[2] / reps, 1e3 * (ms[0] - ms[2]) / reps, 1e3 * (ms[1] - ms[2]) / reps, bad);
        (void)hipFree(U), (void)hipFree(out), (void)hipFree(ctr);
    }
    return 0;
}
