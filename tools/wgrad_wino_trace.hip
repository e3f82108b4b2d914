// Phase-timeline probe of the Winograd backward-weights kernel: per-wave s_memtime stamps around the MFMA phase, the two
// barriers and the transform phase of every batch.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DWW_TRACE tools/wgrad_wino_trace.hip -o /tmp/ww_trace && /tmp/ww_trace [blocks]
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "../rpg_ramnet_amd/csrc/conv_wgrad_wino.hip"

namespace ramnet { void set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); } }

int main(int argc, char **argv) {
    const int B = 8, H = 64, W = 86, Cin = 256, Cout = 256;
    if (argc > 1) setenv("RAMNET_WGRAD_BLOCKS", argv[1], 1);
    const int blocks = argc > 1 ? atoi(argv[1]) : 512;
    float *x, *dy, *ws;
    const size_t nx = (size_t)B * H * W * Cin, ny = (size_t)B * H * W * Cout, nw = (size_t)16 * Cin * Cout;
    hipMalloc(&x, nx * 4), hipMalloc(&dy, ny * 4), hipMalloc(&ws, nw * 4);
    hipMemset(x, 0, nx * 4), hipMemset(dy, 0, ny * 4), hipMemset(ws, 0, nw * 4);
    const size_t nt = (size_t)blocks * 4 * 32 * 4;
    unsigned long long *tr;
    hipMalloc(&tr, nt * 8);
    hipMemset(tr, 0, nt * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(ramnet::g_ww_trace), &tr, sizeof(tr));
    ramnet_wgrad_desc d = {};
    d.x0 = x, d.ld0 = Cin, d.C0 = Cin, d.in_mode = RAMNET_IN_PLAIN, d.B = B, d.Hin = H, d.Win = W;
    d.ntaps = 9, d.stride = 1;
    for (int t = 0; t < 9; ++t) d.dy[t] = t / 3 - 1, d.dx[t] = t % 3 - 1;
    d.dout = dy, d.ldg = Cout, d.Cout = Cout, d.Ho = H, d.Wo = W, d.dw = ws, d.algo = RAMNET_ALGO_WINOGRAD;
    for (int it = 0; it < 3; ++it) {
        if (ramnet::launch_wgrad_wino(d, 0)) return 1;
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(nt);
    hipMemcpy(h.data(), tr, nt * 8, hipMemcpyDeviceToHost);
    std::vector<double> mma, wc, tf, wb;
    for (int bl = 0; bl < blocks; ++bl)
        for (int wv = 0; wv < 4; ++wv)
            for (int c = 2; c < 30; ++c) {
                const unsigned long long *t = &h[(((size_t)bl * 4 + wv) * 32 + c) * 4], *tn = t + 4;
                if (!t[0] || !tn[0]) continue;
                mma.push_back((double)(t[1] - t[0])), wc.push_back((double)(t[2] - t[1]));
                tf.push_back((double)(t[3] - t[2])), wb.push_back((double)(tn[0] - t[3]));
            }
    auto stat = [](const char *n, std::vector<double> &v) {
        if (v.empty()) return;
        std::sort(v.begin(), v.end());
        double s = 0;
        for (double e : v) s += e;
        printf("%-28s mean %7.0f  p10 %7.0f  p50 %7.0f  p90 %7.0f  cycles\n", n, s / v.size(), v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10]);
    };
    printf("%d workgroups\n", blocks);
    stat("MFMA phase (32 MFMA + fills)", mma), stat("wait barrier 1", wc), stat("transform phase", tf), stat("wait barrier 2", wb);
    return 0;
}
