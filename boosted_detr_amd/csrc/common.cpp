// Error reporting shared by every translation unit of libbdetr.
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void bdetr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* bdetr_last_error(void) { return g_err; }
extern "C" int bdetr_abi_version(void) { return BDETR_ABI_VERSION; }

// A non-blocking HIP stream of the LOWEST priority the device offers: the host runs the weight-gradient
// GEMMs on it so that the workgroup dispatcher serves the critical path (the caller's stream) first and
// lets the side work fill what is left.  torch.cuda.Stream only exposes the normal and high classes.
extern "C" int bdetr_low_priority_stream_create(void** out) {
    BDETR_CHECK_ARG(out != nullptr, "bdetr_low_priority_stream_create: null out");
    int least = 0, greatest = 0;
    hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (e == hipSuccess) {
        hipStream_t s = nullptr;
        e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, least);
        if (e == hipSuccess) { *out = (void*)s; return 0; }
    }
    bdetr_set_error("bdetr_low_priority_stream_create: %s", hipGetErrorString(e));
    return (int)e;
}
extern "C" int bdetr_stream_priority_range(int* least, int* greatest) {
    hipError_t e = hipDeviceGetStreamPriorityRange(least, greatest);
    if (e != hipSuccess) { bdetr_set_error("bdetr_stream_priority_range: %s", hipGetErrorString(e)); return (int)e; }
    return 0;
}

// Census of a captured hipGraph's nodes by hipGraphNodeType (counts[t] for t < ncounts; types beyond go to counts[ncounts - 1]).
// The replay path's soundness rests on "kernel nodes only" (a hipMemset node replayed wrongly in round 4): engine.SegmentedCapture
// checks it after capturing when BDETR_GRAPH_CENSUS=1, tests/test_training_gpu.py always.
extern "C" int bdetr_graph_node_census(void* graph, int64_t* counts, int ncounts) {
    BDETR_CHECK_ARG(graph != nullptr && counts != nullptr && ncounts > 0, "bdetr_graph_node_census: bad arguments");
    for (int i = 0; i < ncounts; ++i) counts[i] = 0;
    size_t n = 0;
    hipError_t e = hipGraphGetNodes((hipGraph_t)graph, nullptr, &n);
    if (e != hipSuccess) { bdetr_set_error("bdetr_graph_node_census: hipGraphGetNodes: %s", hipGetErrorString(e)); return (int)e; }
    if (n == 0) return 0;
    hipGraphNode_t* nodes = (hipGraphNode_t*)malloc(n * sizeof(hipGraphNode_t));
    if (!nodes) { bdetr_set_error("bdetr_graph_node_census: out of host memory"); return (int)hipErrorOutOfMemory; }
    e = hipGraphGetNodes((hipGraph_t)graph, nodes, &n);
    for (size_t i = 0; e == hipSuccess && i < n; ++i) {
        hipGraphNodeType t;
        e = hipGraphNodeGetType(nodes[i], &t);
        if (e == hipSuccess) counts[(int)t < ncounts ? (int)t : ncounts - 1] += 1;
    }
    free(nodes);
    if (e != hipSuccess) { bdetr_set_error("bdetr_graph_node_census: %s", hipGetErrorString(e)); return (int)e; }
    return 0;
}
