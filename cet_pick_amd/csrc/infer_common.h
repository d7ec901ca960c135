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

// register march of the (3,3,3) window (infer_peak3.hip): every wave owns one candidate segment
struct Peak3Params {
    const float* in;
    float* val_out;         // pre-NMS value (sigmoid heat-map) or null
    float* nms_out;         // dense NMS'd volume or null
    int D, H, W, zchunk;
    uint2* cands;           // segment s = cands + s * seg_cap, or null
    unsigned* seg_count;    // entries of segment s (written by its wave, zero included)
    unsigned seg_cap;
    unsigned* hist;         // MI_HIST_BINS global bins or null
};
struct Peak3Grid {
    int gx, gy, gz, zchunk;
    unsigned n_seg, seg_cap;
};
Peak3Grid mi_peak3_grid(int D, int H, int W);
bool mi_peak3_usable(const float* in, const float* val_out, const float* nms_out, int D, int H, int W);
int mi_launch_peak3(Peak3Params p, const Peak3Grid& g, bool sigmoid, hipStream_t s);

// fused x pass + DoG + 3x3 xy-NMS + statistics + candidate compaction of the picker (infer_dogx.hip)
struct DogxParams {
    const float* y1;        // smaller sigma, after the z and y passes
    const float* y2;        // larger sigma, after the z and y passes
    float* nms_out;         // dense NMS'd DoG volume or null
    int D, H, W;
    int bz, by, bx;         // zeroed border
    int ychunk, n_ychunks;
    uint2* cands;           // segment of wave g = cands + g * seg_cap
    unsigned* seg_count;
    unsigned seg_cap;
    unsigned* overflow;     // bit 0 set when a segment overflowed
    double* stats;          // {count, sum, sumsq} of the positive survivors, per wave
};
struct DogxGrid {
    int ychunk, n_ychunks;
    unsigned n_seg, seg_cap;
};
DogxGrid mi_dogx_grid(int D, int H, int W);
bool mi_dogx_usable(const float* y1, const float* y2, const float* nms_out, int D, int H, int W, float s1, float s2, int k);
int mi_launch_dogx(DogxParams p, const DogxGrid& g, float s1, float s2, hipStream_t st);

dim3 mi_march_grid(int D, int H, int W, int* zchunk_out);
int mi_launch_march(MarchParams p, int kz, int kxy, hipStream_t s);
