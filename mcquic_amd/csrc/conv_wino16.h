// Launch interface between conv_mfma.hip (mcq_conv2d_f32 / mcq_conv2d_multi_f32 dispatch) and conv_wino16.hip
// (F(2x2, 3x3) on v_mfma_f32_16x16x4_f32, two waves per SIMD).  Internal to the library.
#pragma once
#include <stdint.h>

constexpr int W16_MAX_MULTI = 4;        // problems of one geometry per launch (= MCQ_CONV_MAX_MULTI)

struct W16Ptrs { const float* x; const float* wp; const float* bias; float* y; float* y2; const float* res; };

struct W16K {
    const float* x; const float* wp; const float* bias; float* y; float* y2; const float* res;
    W16Ptrs alt[W16_MAX_MULTI - 1];     // problems 1 .. nprob - 1
    int nprob;
    int N, Cin, H, W, Cout, Ho, Wo;
    int G;                              // groups of four input channels
    int bw_log2, nbx, nby;              // a tile block is (16 >> bw_log2) rows x (1 << bw_log2) tiles of 2 x 2 pixels
    unsigned flags; float res_scale;
};

// fills G / bw_log2 / nbx / nby and launches; returns MCQ_OK / MCQ_EINVAL / MCQ_ETOOLARGE / MCQ_ELAUNCH
int mcq_wino16_launch(W16K& k, void* stream);
