// Fused per-element epilogues shared by the direct (conv_igemm.hip) and Winograd (conv_wino.hip) convolution kernels.
#pragma once
#include "common.hpp"

namespace ramnet {

// value `acc` of output channel n at flattened output pixel `pix` -> bias, activation / residual / GRU blend, store.
// addold: this pixel accumulates beta*out_old into the pre-activation (LINEAR / RELU)
__device__ __forceinline__ bool epilogue_addold(const ramnet_conv_desc &p, int oyF, int oxF) { return p.beta != 0.f; }

// Border corrections of the folded upsample-conv (ramnet_conv_desc.frame > 0): the outermost `frame` rows add
// e1[side][b][oxF][slot*Cout + n] (side 0 top / 1 bottom, slot = distance into the band), the outermost columns
// e0[side][b][oyF][slot*Cout + n] (left / right); corner pixels get both.
__device__ __forceinline__ float epilogue_side(const ramnet_conv_desc &p, int epi, int b, int oyF, int oxF, int n) {
    if (p.frame == 0 || (epi != RAMNET_EPI_RELU && epi != RAMNET_EPI_LINEAR)) return 0.f;
    float v = 0.f;
    if (oyF < p.frame || oyF >= p.HoF - p.frame) {
        const int side = oyF < p.frame ? 0 : 1, slot = side ? oyF - (p.HoF - p.frame) : oyF;
        v += p.e1[(((size_t)side * p.B + b) * p.WoF + oxF) * p.lde1 + slot * p.Cout + n];
    }
    if (oxF < p.frame || oxF >= p.WoF - p.frame) {
        const int side = oxF < p.frame ? 0 : 1, slot = side ? oxF - (p.WoF - p.frame) : oxF;
        v += p.e0[(((size_t)side * p.B + b) * p.HoF + oyF) * p.lde0 + slot * p.Cout + n];
    }
    return v;
}

__device__ __forceinline__ void epilogue_store(const ramnet_conv_desc &p, int epi, size_t pix, int n, float acc, bool addold) {
    float v = acc + (p.bias ? p.bias[n] : 0.f);
    if (addold && (epi == RAMNET_EPI_RELU || epi == RAMNET_EPI_LINEAR)) v += p.beta * p.out[pix * p.ldo + n];
    if (epi == RAMNET_EPI_RELU) {
        v = fmaxf(v, 0.f);
    } else if (epi == RAMNET_EPI_SIGMOID) {
        v = sigmoidf_(v);
    } else if (epi == RAMNET_EPI_SIGMOID_HR) {      // gates [u | r]; the r half also writes h.r (ramnet_hip.h)
        v = sigmoidf_(v);
        if (n >= p.Cout / 2) p.o1[pix * p.ldo1 + n - p.Cout / 2] = p.e1[pix * p.lde1 + n - p.Cout / 2] * v;
    } else if (epi == RAMNET_EPI_RES_RELU) {
        v = fmaxf(v + p.e0[pix * p.lde0 + n], 0.f);
    } else if (epi == RAMNET_EPI_GRU_BLEND) {
        const float o = tanhf_(v);
        const float u = p.e0[pix * p.lde0 + n];
        const float h = p.e1 ? p.e1[pix * p.lde1 + n] : 0.f;
        if (p.o1) p.o1[pix * p.ldo1 + n] = o;
        v = h * (1.0f - u) + o * u;
    } else if (epi == RAMNET_EPI_GRU_BWD && n >= p.Cout / 2) {     // stage B of the ConvGRU backward (ramnet_hip.h)
        const float r = p.e0[pix * p.lde0 + n];
        const float h = p.e1 ? p.e1[pix * p.lde1 + n - p.Cout / 2] : 0.f;
        p.o1[pix * p.ldo1 + n] = v * h * r * (1.0f - r);
        v = p.out[pix * p.ldo + n] + v * r;
    }
    p.out[pix * p.ldo + n] = v;
}

// RAMNET_EPI_GRU_BWD, channels of the d(h.r) half: g = the convolution's result, r, h, old = the direct path already in `out`;
// writes the reset gate's pre-activation gradient and returns the completed dh.
__device__ __forceinline__ float4 gru_bwd_quad(float4 g, float4 r, float4 h, float4 old, float *dpr) {
    st4(dpr, make_float4(g.x * h.x * r.x * (1.0f - r.x), g.y * h.y * r.y * (1.0f - r.y), g.z * h.z * r.z * (1.0f - r.z),
                         g.w * h.w * r.w * (1.0f - r.w)));
    return make_float4(old.x + g.x * r.x, old.y + g.y * r.y, old.z + g.z * r.z, old.w + g.w * r.w);
}

// Four consecutive output channels n..n+3 of one pixel (all pointers 16-byte aligned, Cout % 4 == 0: checked on the host).
__device__ __forceinline__ void epilogue_store4(const ramnet_conv_desc &p, int epi, size_t pix, int n, float4 acc, bool addold) {
    float4 v = p.bias ? f4add(acc, ld4(p.bias + n)) : acc;
    if (addold && (epi == RAMNET_EPI_RELU || epi == RAMNET_EPI_LINEAR)) {
        const float4 old = ld4(p.out + pix * p.ldo + n);
        v = make_float4(v.x + p.beta * old.x, v.y + p.beta * old.y, v.z + p.beta * old.z, v.w + p.beta * old.w);
    }
    if (epi == RAMNET_EPI_RELU) {
        v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
    } else if (epi == RAMNET_EPI_SIGMOID) {
        v = make_float4(sigmoidf_(v.x), sigmoidf_(v.y), sigmoidf_(v.z), sigmoidf_(v.w));
    } else if (epi == RAMNET_EPI_SIGMOID_HR) {      // (Cout / 2 % 4 == 0: a quad lies in one half)
        v = make_float4(sigmoidf_(v.x), sigmoidf_(v.y), sigmoidf_(v.z), sigmoidf_(v.w));
        if (n >= p.Cout / 2) {
            const float4 h = ld4(p.e1 + pix * p.lde1 + n - p.Cout / 2);
            st4(p.o1 + pix * p.ldo1 + n - p.Cout / 2, make_float4(h.x * v.x, h.y * v.y, h.z * v.z, h.w * v.w));
        }
    } else if (epi == RAMNET_EPI_RES_RELU) {
        const float4 e = ld4(p.e0 + pix * p.lde0 + n);
        v = make_float4(fmaxf(v.x + e.x, 0.f), fmaxf(v.y + e.y, 0.f), fmaxf(v.z + e.z, 0.f), fmaxf(v.w + e.w, 0.f));
    } else if (epi == RAMNET_EPI_GRU_BLEND) {
        const float4 o = make_float4(tanhf_(v.x), tanhf_(v.y), tanhf_(v.z), tanhf_(v.w));
        const float4 u = ld4(p.e0 + pix * p.lde0 + n);
        const float4 h = p.e1 ? ld4(p.e1 + pix * p.lde1 + n) : f4zero();
        if (p.o1) st4(p.o1 + pix * p.ldo1 + n, o);
        v = make_float4(h.x * (1.0f - u.x) + o.x * u.x, h.y * (1.0f - u.y) + o.y * u.y, h.z * (1.0f - u.z) + o.z * u.z,
                        h.w * (1.0f - u.w) + o.w * u.w);
    } else if (epi == RAMNET_EPI_GRU_BWD && n >= p.Cout / 2) {     // (Cout / 2 % 4 == 0: a quad lies in one half)
        v = gru_bwd_quad(v, ld4(p.e0 + pix * p.lde0 + n), p.e1 ? ld4(p.e1 + pix * p.lde1 + n - p.Cout / 2) : f4zero(),
                         ld4(p.out + pix * p.ldo + n), p.o1 + pix * p.ldo1 + n);
    }
    st4(p.out + pix * p.ldo + n, v);
}

}  // namespace ramnet
