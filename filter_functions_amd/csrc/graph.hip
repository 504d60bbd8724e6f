// graph.hip -- hipGraph capture and replay of device-pointer entry points.
//
// Every `_dev` entry point of include/ffk.h enqueues kernels on the caller's stream and nothing
// else (no allocation, no synchronisation, no host transfer): a sequence of such calls between
// ffk_graph_capture_begin and ffk_graph_capture_end becomes ONE graph, replayed with one
// hipGraphLaunch.  What it is for: a pass of the hot path (reference call:
// PulseSequence.get_filter_function, pulse_sequence.py:691-805 -> numeric.py:707-881) is 6 launches
// of 5-85 us kernels; enqueued one by one through a foreign-function binding the host side costs
// 27-70 us per step -- as much as the device side (profiles/r02_p_*).  Arguments are baked in at
// capture time: the buffers a captured call names must stay where they are for as long as the
// graph lives.
#include <cstdio>

#include "ffk.h"
#include "ffk_internal.h"

struct ffk_graph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    size_t nodes = 0;
};

extern "C" {

namespace {
int graph_fail(const char* what, hipError_t e) {
    char message[256];
    snprintf(message, sizeof message, "%s failed: %s", what, hipGetErrorString(e));
    ffk::set_last_error(message);
    return FFK_EHIP;
}
}  // namespace

int ffk_graph_capture_begin(void* stream) {
    if (!stream) {
        ffk::set_last_error("capture needs a created stream (the null stream cannot be captured)");
        return FFK_EINVAL;
    }
    // relaxed: calls that other threads (or this one) make outside the captured stream -- PyTorch's
    // allocator, another rank's set-up -- are none of the capture's business
    (void)ffk::kernel_fault_word();   // the kernels' fault word exists before anything is captured
    hipError_t e = hipStreamBeginCapture(static_cast<hipStream_t>(stream), hipStreamCaptureModeRelaxed);
    return e == hipSuccess ? FFK_OK : graph_fail("hipStreamBeginCapture", e);
}

int ffk_graph_capture_end(void* stream, ffk_graph** out) {
    if (!stream || !out) return FFK_EINVAL;
    *out = nullptr;
    hipGraph_t graph = nullptr;
    hipError_t e = hipStreamEndCapture(static_cast<hipStream_t>(stream), &graph);
    if (e != hipSuccess || !graph) {
        (void)hipGetLastError();
        return graph_fail("hipStreamEndCapture", e == hipSuccess ? hipErrorUnknown : e);
    }
    ffk_graph* g = new ffk_graph;
    g->graph = graph;
    (void)hipGraphGetNodes(graph, nullptr, &g->nodes);
    e = hipGraphInstantiate(&g->exec, graph, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        (void)hipGraphDestroy(graph);
        delete g;
        return graph_fail("hipGraphInstantiate", e);
    }
    *out = g;
    return FFK_OK;
}

// abandon a capture after a failed call inside it (the stream leaves capture mode)
int ffk_graph_capture_abort(void* stream) {
    if (!stream) return FFK_EINVAL;
    hipGraph_t graph = nullptr;
    (void)hipStreamEndCapture(static_cast<hipStream_t>(stream), &graph);
    if (graph) (void)hipGraphDestroy(graph);
    (void)hipGetLastError();
    return FFK_OK;
}

int ffk_graph_launch(ffk_graph* g, void* stream) {
    if (!g || !g->exec) return FFK_EINVAL;
    hipError_t e = hipGraphLaunch(g->exec, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? FFK_OK : graph_fail("hipGraphLaunch", e);
}

int ffk_graph_node_count(const ffk_graph* g, int* nodes) {
    if (!g || !nodes) return FFK_EINVAL;
    *nodes = static_cast<int>(g->nodes);
    return FFK_OK;
}

int ffk_graph_destroy(ffk_graph* g) {
    if (!g) return FFK_OK;
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
    return FFK_OK;
}

}  // extern "C"
