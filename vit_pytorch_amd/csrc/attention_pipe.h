// attention_pipe.h -- host-side entry of the persistent, LDS-DMA-pipelined attention kernels (attention_pipe.hip), called by the
// C-ABI functions in attention.hip.
#pragma once
#include <stdint.h>
#include "../../include/vitk.h"

// shapes the pipelined kernels serve: dim_head 64, 32 < N <= 224 (two to seven 32-row steps; ViT-B/L at 224^2: N = 197 / 196)
bool attn_pipe_supported(int64_t N, int64_t d);

// NS = 1: 16-bit operands, 16-bit outputs.  NS = 2: every operand as hi + lo (vitk_split2), f32 outputs (o, dq, dk, dv are float
// tensors with element strides; `o` of the backward too).  Arrays hold NS entries.
struct AttnPipeFwd {
    int ns;
    vitk_bhnd q[2], k[2], v[2], o;
    float* lse;
    int64_t B, H, N;
    float scale, drop_p;
    uint32_t drop_seed;
};
struct AttnPipeBwd {
    int ns;
    vitk_bhnd q[2], k[2], v[2], dout[2], o, dq, dk, dv;
    const float* lse;
    float* delta;
    int64_t B, H, N;
    float scale, drop_p;
    uint32_t drop_seed;
};
int attn_pipe_fwd(const AttnPipeFwd& a, void* stream);
// which: bit 0 = dQ (+ delta), bit 1 = dK / dV (needs the delta a dQ kernel wrote before)
int attn_pipe_bwd(const AttnPipeBwd& a, void* stream, int which);
// the single-kernel backward (dQ, dK, dV in one pass over the scores): 16-bit, 192 < N <= 208, no dropout; mask bit 3
bool attn_fused_bwd_supported(int64_t N, int64_t d, float drop_p);
int attn_pipe_bwd_fused(const AttnPipeBwd& a, void* stream);
// selection of the 16-bit kernels: bit 0 forward, bit 1 dQ, bit 2 dK/dV, bit 3 the single-kernel backward where it applies
// (default 2 | 8; 0 while vitk_set_cu_reserve > 0; VITK_ATTN_PIPE overrides)
int attn_pipe_mask();
