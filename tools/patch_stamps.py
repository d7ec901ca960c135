"""Writes tools/var/src/infer_decode1.hip = the product kernel + s_memrealtime stamps (tools/ab/decode1_stamps.py reads them;
tools/build_var.sh builds the variant library from it).   python tools/patch_stamps.py"""
import os, shutil
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(R, "cet_pick_amd/csrc/infer_decode1.hip")
dst = os.path.join(R, "tools/var/src/infer_decode1.hip")
s = open(src).read()
def ins(anchor, k):
    global s
    assert anchor in s, anchor
    s = s.replace(anchor, '    if (tid == 0) dbg[%d] = __builtin_amdgcn_s_memrealtime();\n' % k + anchor, 1)
s = s.replace("    if (tid == 0) { L.keep_n = 0; L.spilled = 0; L.wg_total = 0; }",
  "    unsigned long long* dbg = reinterpret_cast<unsigned long long*>(p.cands + (size_t)p.n_seg * p.seg_cap) - (size_t)(p.n_wg - wg) * 16;\n    if (tid == 0) dbg[0] = __builtin_amdgcn_s_memrealtime();\n    if (tid == 0) { L.keep_n = 0; L.spilled = 0; L.wg_total = 0; }", 1)
ins("    uint2 le[LCAP]; ", 1)
ins("    unsigned above = 0;\n    const int T = lds_threshold_bin(L.hist, a_M, lane, &above);", 2)
ins("    if (wv == 0) {\n        // all TSLOT slots are written", 3)
ins("    if (tid == 0) L.last = (__hip_atomic_fetch_add(&a_hdr->pad0", 4)
ins("    if (!L.last) return;", 5)
ins("    unsigned mx = 0, bad = 0;\n    for (unsigned w = tid; w < a_nwg; w += NT) {", 9)
ins("    for (int i = tid; i < MI_HIST_BINS; i += NT) L.hist[i] = 0u;\n    if (tid < 256) L.sub[tid] = 0u;\n    if (tid == 0) { L.fallback = 0u;", 10)
ins("    n_t_mine = wave_sum_u32(n_t_mine);", 11)
ins("    const unsigned n_t = L.keep_n;                          // entries in the tables", 6)
ins("    __syncthreads();\n    if (!fallback) {\n        for (unsigned c = 0; c < n_chunks; ++c) {\n            if (n_chunks > 1) load_chunk(c);\n            // this thread's survivors", 12)
ins("    if (!fallback) {\n        const unsigned n_s = L.n_s;", 7)
ins("        {   // start[b] = keys in buckets above b", 13)
ins("        // place: a bucket's keys end up in P", 14)
ins("    for (int r = n_valid + tid; r < K; r += NT) d_emit_det(a_dets, r, 0ull, a_H, a_W, false);", 8)
os.makedirs(os.path.dirname(dst), exist_ok=True)
open(dst, "w").write(s)
