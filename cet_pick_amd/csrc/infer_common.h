// Types shared by the inference kernels (infer_nms.hip, infer_gauss.hip, infer_greedy.hip).
#pragma once
#include "common.h"
#include <algorithm>
#include <cstdlib>

enum { MI_LOAD_PLAIN = 0, MI_LOAD_SIGMOID = 1, MI_LOAD_DOG = 2 };

constexpr int MI_HIST_SHIFT = 20;                 // score bits >> 20: sign + exponent + 3 mantissa
constexpr int MI_HIST_BINS = 1 << (32 - MI_HIST_SHIFT - 1);   // 2048 (positive floats)
constexpr int MI_SEL_CAP = 16384;                 // keys sortable in one workgroup's LDS

struct MarchParams {
    const float* in;        // value source (PLAIN / SIGMOID) or smaller-sigma Gaussian (DOG)
    const float* in2;       // DOG: larger-sigma Gaussian (value = in2 - in)
    float* val_out;         // pre-NMS value (e.g. sigmoid heat-map) or null
    float* nms_out;         // dense NMS'd volume or null
    int accumulate;         // nms_out <- max(nms_out, this level)
    int mode, fiber;
    int D, H, W, zchunk, vec_ok;
    int bz, by, bx;         // DOG border zeroing
    uint2* cands;           // (score bits, flat index) of positive NMS survivors, or null
    unsigned* cand_count;
    unsigned cand_cap;
    unsigned* hist;         // MI_HIST_BINS global bins or null
    double* stats;          // per-workgroup {count, sum, sumsq} of positive survivors, or null
};

struct DecodeHeader {
    unsigned cand_count, sel_count, pad0, pad1;
    unsigned hist[MI_HIST_BINS];
};

dim3 mi_march_grid(int D, int H, int W, int* zchunk_out);
int mi_launch_march(MarchParams p, int kz, int kxy, hipStream_t s);
