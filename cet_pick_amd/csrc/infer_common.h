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

// the decode in one launch (infer_decode1.hip): march + per-workgroup best lists + last-arriver selection
size_t mi_decode1_extra_bytes(int D, int H, int W);
bool mi_decode1_usable(const float* in, const float* val_out, int D, int H, int W, int K);
int mi_launch_decode1(const float* in, float* val_out, int D, int H, int W, bool sigmoid, int K, float* dets, int* n_valid_out,
                      DecodeHeader* hdr, uint2* cands, unsigned* seg_count, void* extra, hipStream_t s);

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

// z + x passes | y pass + DoG + 3x3 xy-NMS + statistics + candidates: the picker's filter stage in two launches (infer_dogf.hip)
struct DogfParams {
    const float* rec;       // tomogram (D, H, W)
    float* g1;              // smaller sigma after the z and x passes (planes [bz, D - bz), rows [ylo, yhi) are written)
    float* g2;              // larger sigma, likewise
    float* nms_out;         // dense NMS'd DoG volume (the live box is written; the caller zeroes the rest) or null
    int D, H, W;
    int bz, by, bx;         // zeroed border
    int ylo, yhi;           // rows the y pass reads
    unsigned vol_bytes;     // D * H * W * 4 (< 2 GiB)
    int ychunk, n_ychunks, n_strips;
    unsigned n_seg, n_wg;   // waves / workgroups of the y march
    uint2* cands;           // segment of wave g = cands + g * seg_cap
    unsigned* seg_count;
    unsigned seg_cap;
    unsigned* overflow;     // bit 0 set when a segment overflowed
    double* stats;          // {count, sum, sumsq} of the positive survivors, per workgroup of the y march
    unsigned* clr[2];       // word ranges the first launch zeroes on the side (header, candidate bitmap) or null
    unsigned clr_n[2];
};
struct DogfGrid {
    int ychunk, n_ychunks, n_strips;
    unsigned n_seg, n_wg, seg_cap;
};
DogfGrid mi_dogf_grid(int D, int H, int W, int bz, int bxy);
bool mi_dogf_usable(const float* rec, const float* g1, const float* g2, const float* nms_out, int D, int H, int W,
                    float s1, float s2, int k, int bz, int bxy);
int mi_launch_dogf(DogfParams p, const DogfGrid& g, float s1, float s2, hipStream_t st);
int mi_dogf_own();         // columns a strip of the y march owns


dim3 mi_march_grid(int D, int H, int W, int* zchunk_out);
int mi_launch_march(MarchParams p, int kz, int kxy, hipStream_t s);
