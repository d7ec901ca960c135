// Library identity for the C-ABI (include/cetpick_hip.h).
#include "common.h"
#include <cstdlib>

extern "C" int mi_abi_version(void) { return 3; }
extern "C" const char* mi_build_arch(void) { return "gfx950"; }

// Nodes of a captured hipGraph by type (measurement aid: launches per replayed step).  counts[0..3] = kernel, memcpy,
// memset, other.
extern "C" int mi_graph_node_counts(void* graph, int* counts) {
    if (!graph || !counts) return MI_E_ARG;
    size_t n = 0;
    MI_HIP(hipGraphGetNodes((hipGraph_t)graph, nullptr, &n));
    counts[0] = counts[1] = counts[2] = counts[3] = 0;
    if (n == 0) return MI_OK;
    hipGraphNode_t* nodes = (hipGraphNode_t*)malloc(sizeof(hipGraphNode_t) * n);
    if (!nodes) return MI_E_ARG;
    hipError_t e = hipGraphGetNodes((hipGraph_t)graph, nodes, &n);
    for (size_t i = 0; e == hipSuccess && i < n; ++i) {
        hipGraphNodeType t;
        e = hipGraphNodeGetType(nodes[i], &t);
        if (e != hipSuccess) break;
        if (t == hipGraphNodeTypeKernel) ++counts[0];
        else if (t == hipGraphNodeTypeMemcpy) ++counts[1];
        else if (t == hipGraphNodeTypeMemset) ++counts[2];
        else ++counts[3];
    }
    free(nodes);
    return e == hipSuccess ? MI_OK : (int)e;
}

// Measurement aid: one thread writes the constant-rate (100 MHz) wall clock into *slot.  A launch of its own, so that it
// can be recorded into a captured step at the points whose time is asked for (tools/stamp_step.py).
__global__ void debug_stamp_kernel(unsigned long long* slot) { *slot = wall_clock64(); }

extern "C" int mi_debug_stamp(void* slot, mi_stream_t stream) {
    if (!slot) return MI_E_ARG;
    hipLaunchKernelGGL(debug_stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)slot);
    return (int)hipGetLastError();
}
