// Fused per-element epilogues shared by the direct (conv_igemm.hip) and Winograd (conv_wino.hip) convolution kernels.
#pragma once
#include "common.hpp"

namespace ramnet {

// value `acc` of output channel n at flattened output pixel `pix` -> bias, activation / residual / GRU blend, store.
__device__ __forceinline__ void epilogue_store(const ramnet_conv_desc &p, int epi, size_t pix, int n, float acc) {
    float v = acc + (p.bias ? p.bias[n] : 0.f);
    if (epi == RAMNET_EPI_RELU) {
        v = fmaxf(v, 0.f);
    } else if (epi == RAMNET_EPI_SIGMOID) {
        v = sigmoidf_(v);
    } else if (epi == RAMNET_EPI_RES_RELU) {
        v = fmaxf(v + p.e0[pix * p.lde0 + n], 0.f);
    } else if (epi == RAMNET_EPI_GRU_BLEND) {
        const float o = tanhf(v);
        const float u = p.e0[pix * p.lde0 + n];
        const float h = p.e1 ? p.e1[pix * p.lde1 + n] : 0.f;
        if (p.o1) p.o1[pix * p.ldo1 + n] = o;
        v = h * (1.0f - u) + o * u;
    } else if (p.beta != 0.f) {
        v += p.beta * p.out[pix * p.ldo + n];
    }
    p.out[pix * p.ldo + n] = v;
}

}  // namespace ramnet
