// Phase-timeline probe of the Winograd F(2x2,4x4) decoder kernel: per-wave s_memtime stamps (100 MHz) inside the chunk loop.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DW24_TRACE -Iinclude tools/wino24_trace.hip -o /tmp/w24_trace && /tmp/w24_trace
#include <cstdarg>
#include <cstdio>
#include <vector>
#include <algorithm>
#include "../rpg_ramnet_amd/csrc/conv_wino24.hip"

namespace ramnet {      // what the kernel file expects from the rest of the library
int g_opt_voxel_sorted = 1, g_opt_fold_pair = 1, g_opt_wgrad_blocks = 512, g_opt_wgrad_wino_blocks = 384, g_opt_wino_ksplit = 1, g_opt_wgrad_wino_nf = 1;
void set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
void note_kernel(const char *, ...) {}
hipError_t allow_full_lds(const void *kernel) { return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }
}

int main(int argc, char **argv) {
    const int which = argc > 1 ? atoi(argv[1]) : 0;      // 0: dec0 shape, 1: dec1, 2: dec2
    const int B = 8, H = 32 << which, W = which == 0 ? 43 : which == 1 ? 86 : 172, Cin = 256 >> which, Cout = 128 >> which;
    float *x, *y, *w;
    const size_t nx = (size_t)B * (H + 4) * (W + 4) * Cin, ny = (size_t)B * 4 * H * W * Cout, nw = (size_t)4 * 25 * Cin * Cout;
    hipMalloc(&x, nx * 4), hipMalloc(&y, ny * 4), hipMalloc(&w, nw * 4);
    hipMemset(x, 0, nx * 4), hipMemset(w, 0, nw * 4);
    const bool wide = Cout % 64 == 0;
    const int nblocks = B * ((H + 15) / 16) * ((W + (wide ? 7 : 15)) / (wide ? 8 : 16)) * (Cout / (wide ? 64 : 32)) * 4;
    const size_t nt = (size_t)nblocks * 8 * 17 * 6;
    unsigned long long *tr;
    hipMalloc(&tr, nt * 8);
    hipMemset(tr, 0, nt * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(ramnet::g_w24_trace), &tr, sizeof(tr));
    ramnet_conv_desc d = {};
    d.x0 = x, d.ld0 = Cin, d.C0 = Cin, d.in_mode = RAMNET_IN_PLAIN, d.B = B, d.Hin = H + 4, d.Win = W + 4;
    d.ntaps = 16, d.stride = 1;
    d.w = w, d.Cout = Cout, d.Ho = H, d.Wo = W, d.HoF = 2 * H, d.WoF = 2 * W, d.osy = d.osx = 1;
    d.epi = RAMNET_EPI_RELU, d.out = y, d.ldo = Cout, d.algo = RAMNET_ALGO_WINOGRAD24;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(e0, 0);
        if (ramnet::launch_wino24(d, 0)) return 1;
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("launch %d: %.3f ms, %d blocks\n", it, ms, nblocks);
    }
    std::vector<unsigned long long> h(nt);
    hipMemcpy(h.data(), tr, nt * 8, hipMemcpyDeviceToHost);
    const int nch = Cin / (wide ? 16 : 8);
    std::vector<double> ph[6], tot, pro, loop, epi;
    for (int bl = 0; bl < nblocks; ++bl)
        for (int wv = 0; wv < 8; ++wv) {
            const unsigned long long *base = &h[((size_t)bl * 8 + wv) * 17 * 6];
            for (int c = 1; c < std::min(nch, 16) - 1; ++c) {
                const unsigned long long *t = base + c * 6, *tn = t + 6;
                for (int k = 0; k < 5; ++k) ph[k].push_back((double)(t[k + 1] - t[k]));
                ph[5].push_back((double)(tn[0] - t[0]));
            }
            const unsigned long long *k = base + 16 * 6;
            pro.push_back((double)(k[1] - k[0])), loop.push_back((double)(k[2] - k[1])), epi.push_back((double)(k[3] - k[2])), tot.push_back((double)(k[3] - k[0]));
        }
    auto stat = [](const char *n, std::vector<double> &v) {
        std::sort(v.begin(), v.end());
        double s = 0;
        for (double e : v) s += e;
        printf("%-34s mean %8.1f  p10 %8.0f  p50 %8.0f  p90 %8.0f  (cycles)\n", n, s / v.size(), v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10]);
    };
    stat("chunk start -> pos 13", ph[0]), stat("column pass (pos 13-17)", ph[1]), stat("pos 17 -> row pass end (22)", ph[2]);
    stat("pos 22 -> barrier", ph[3]), stat("barrier wait", ph[4]), stat("whole chunk", ph[5]);
    stat("prologue", pro), stat("main loop", loop), stat("epilogue", epi), stat("kernel (per wave)", tot);
    return 0;
}
