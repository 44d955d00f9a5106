// MFMA forms of the soft assignment's two backward contractions (vq_bwd_mfma.hip), dispatched by mcq_vq_soft_bwd_f32 (vq_train.hip).
#pragma once
#include <stdint.h>

struct VqBwdK {
    const float* ddist;      // [rows, k], rows = N m hw ordered (n, g, pixel)
    const float* rowsum;     // [rows]
    const float* x;          // [N, m d, hw]
    const float* xt;         // [N hw, m d]   channel-major copy of x
    const float* dqt;        // [N hw, m d]   channel-major copy of the incoming gradient
    const int64_t* index;    // [rows] sampled codeword
    const float* hot;        // [rows] its straight-through value
    const float* cb;         // [m, k, d]
    float* dx;               // [N, m d, hw]
    float* dcb;              // [m, k, d]
    int N, m, d, hw, k, rows;
};

bool mcq_vq_dc_mfma_ok(const VqBwdK& p);
bool mcq_vq_dx_mfma_ok(const VqBwdK& p);
void mcq_vq_dc_mfma_launch(const VqBwdK& p, void* stream);
void mcq_vq_dx_mfma_launch(const VqBwdK& p, void* stream);
