// Phase-timeline probe of the Winograd kernel (LDS-transform variant): per-wave s_memtime stamps at the two barriers of
// every chunk.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DWINO_TRACE tools/wino_trace.hip -o /tmp/wino_trace && RAMNET_WINO_LDS_TRANSFORM=1 /tmp/wino_trace
#include <cstdarg>
#include <cstdio>
#include <vector>
#include <algorithm>
#include "../rpg_ramnet_amd/csrc/conv_wino.hip"

namespace ramnet { void set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); } }

int main() {
    const int B = 8, H = 64, W = 86, Cin = 256, Cout = 256;
    float *x, *y, *w;
    const size_t nx = (size_t)B * H * W * Cin, ny = (size_t)B * H * W * Cout;
    const size_t nw = ramnet_packed_weight_elems_wino(Cout, Cin, 0, 1);
    hipMalloc(&x, nx * 4), hipMalloc(&y, ny * 4), hipMalloc(&w, nw * 4);
    hipMemset(x, 0, nx * 4), hipMemset(w, 0, nw * 4);
    const int nblocks = B * ((H + 7) / 8) * ((W + 15) / 16) * (Cout / 64);
    const size_t nt = (size_t)nblocks * 4 * 32 * 4;
    unsigned long long *tr;
    hipMalloc(&tr, nt * 8);
    hipMemset(tr, 0, nt * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(ramnet::g_wino_trace), &tr, sizeof(tr));
    ramnet_conv_desc d = {};
    d.x0 = x, d.ld0 = Cin, d.C0 = Cin, d.in_mode = RAMNET_IN_PLAIN, d.B = B, d.Hin = H, d.Win = W;
    d.ntaps = 9, d.stride = 1;
    for (int t = 0; t < 9; ++t) d.dy[t] = t / 3 - 1, d.dx[t] = t % 3 - 1, d.wtap[t] = t;
    d.w = w, d.Cout = Cout, d.Ho = H, d.Wo = W, d.HoF = H, d.WoF = W, d.osy = d.osx = 1;
    d.epi = RAMNET_EPI_RELU, d.out = y, d.ldo = Cout, d.algo = RAMNET_ALGO_WINOGRAD;
    for (int it = 0; it < 3; ++it) {
        if (ramnet::launch_wino(d, 0)) return 1;
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(nt);
    hipMemcpy(h.data(), tr, nt * 8, hipMemcpyDeviceToHost);
    // per (block, wave, chunk): p1 = t1-t0 (phase-1 issue), wY = t2-t1, p2 = t3-t2, wX = t0(next)-t3
    std::vector<double> p1, wy, p2, wx;
    for (int bl = 0; bl < nblocks; ++bl)
        for (int wv = 0; wv < 4; ++wv)
            for (int c = 2; c < 30; ++c) {
                const unsigned long long *t = &h[(((size_t)bl * 4 + wv) * 32 + c) * 4], *tn = t + 4;
                p1.push_back((double)(t[1] - t[0])), wy.push_back((double)(t[2] - t[1]));
                p2.push_back((double)(t[3] - t[2])), wx.push_back((double)(tn[0] - t[3]));
            }
    auto stat = [](const char *n, std::vector<double> &v) {
        std::sort(v.begin(), v.end());
        double s = 0;
        for (double e : v) s += e;
        printf("%-22s mean %8.0f  p10 %8.0f  p50 %8.0f  p90 %8.0f  (memtime ticks)\n", n, s / v.size(), v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10]);
    };
    stat("phase 1 (mma 0-7)", p1), stat("wait barrier Y", wy), stat("phase 2 (mma 8-15)", p2), stat("wait barrier X", wx);
    {   // weights-from-global variant: chunk 0 slot 2 = kernel start, chunk 31 slot 2 = loop end, slot 3 = kernel end
        std::vector<double> pro, loop, epi;
        for (int bl = 0; bl < nblocks; ++bl)
            for (int wv = 0; wv < 4; ++wv) {
                const unsigned long long *t0 = &h[(((size_t)bl * 4 + wv) * 32 + 0) * 4], *t31 = &h[(((size_t)bl * 4 + wv) * 32 + 31) * 4];
                pro.push_back((double)(t0[0] - t0[2])), loop.push_back((double)(t31[2] - t0[0])), epi.push_back((double)(t31[3] - t31[2]));
            }
        stat("prologue", pro), stat("main loop (32 chunks)", loop), stat("epilogue", epi);
    }
    // whole-iteration time and the offset between the two blocks' first stamps is not recoverable without CU ids; print block 0
    for (int wv = 0; wv < 2; ++wv) {
        printf("block 0 wave %d:", wv);
        for (int c = 4; c < 8; ++c) {
            const unsigned long long *t = &h[(((size_t)0 * 4 + wv) * 32 + c) * 4];
            printf("  [%llu %llu %llu %llu]", t[0] - h[(size_t)wv * 32 * 4 + 16], t[1] - t[0], t[2] - t[1], t[3] - t[2]);
        }
        printf("\n");
    }
    return 0;
}
