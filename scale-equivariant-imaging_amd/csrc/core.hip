// Library identity entry points, and HIP events that stay usable across a hipGraph boundary.
#include "sei_common.h"
#include <string.h>
#include <vector>

extern "C" int sei_abi_version(void) { return SEI_ABI_VERSION; }

extern "C" int sei_build_target(char *name, int n) {
    SEI_REQUIRE(name && n > 0);
    strncpy(name, "gfx950", (size_t)n);
    name[n - 1] = 0;
    return SEI_OK;
}

// ---- events recorded INSIDE a captured graph and waited for OUTSIDE it -----------------------------------------
// torch.cuda.Event(external=True) is refused on ROCm builds of PyTorch, but HIP has the mechanism:
// an event-record NODE in the captured graph (hipGraphAddEventRecordNode) is recorded anew whenever a replay
// reaches that point, and an ordinary hipStreamWaitEvent issued on
// another stream AFTER the replay was enqueued waits for exactly that point. Used to release the bottleneck
// block's gradients to the RCCL stream before the captured backward has finished (graphs.py, parallel.py).
extern "C" int sei_event_create(void **event) {
    SEI_REQUIRE(event);
    hipEvent_t e;
    const hipError_t rc = hipEventCreateWithFlags(&e, hipEventDisableTiming);
    if (rc != hipSuccess) return (int)rc;
    *event = (void *)e;
    return SEI_OK;
}

extern "C" int sei_event_destroy(void *event) {
    SEI_REQUIRE(event);
    return (int)hipEventDestroy((hipEvent_t)event);
}

extern "C" int sei_event_record_external(void *event, void *stream) {
    SEI_REQUIRE(event);
    hipStream_t s = (hipStream_t)stream;
    hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
    unsigned long long id = 0;
    hipGraph_t graph = nullptr;
    const hipGraphNode_t *deps = nullptr;
    size_t ndeps = 0;
    hipError_t rc = hipStreamGetCaptureInfo_v2(s, &status, &id, &graph, &deps, &ndeps);
    if (rc != hipSuccess) return (int)rc;
    if (status != hipStreamCaptureStatusActive)          // not capturing: an ordinary record
        return (int)hipEventRecord((hipEvent_t)event, s);
    // hipEventRecordWithFlags(hipEventRecordExternal) returns hipErrorInvalidValue under capture on ROCm 7.2
    // (tools/probe_external_event.py); inserting the node by hand and making it the stream's new capture
    // dependency is equivalent and works.
    hipGraphNode_t node = nullptr;
    rc = hipGraphAddEventRecordNode(&node, graph, deps, ndeps, (hipEvent_t)event);
    if (rc != hipSuccess) return (int)rc;
    return (int)hipStreamUpdateCaptureDependencies(s, &node, 1, hipStreamSetCaptureDependencies);
}

extern "C" int sei_stream_wait_event(void *stream, void *event) {
    SEI_REQUIRE(event);
    return (int)hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0);
}

// ---- measurement aid: how many nodes a captured graph has (each kernel node costs ~1.5 us of launch structure on replay)
extern "C" int sei_graph_node_counts(void *graph, int *kernel_nodes, int *all_nodes) {
    SEI_REQUIRE(graph && kernel_nodes && all_nodes);
    size_t n = 0;
    hipError_t rc = hipGraphGetNodes((hipGraph_t)graph, nullptr, &n);
    if (rc != hipSuccess) return (int)rc;
    std::vector<hipGraphNode_t> nodes(n);
    if (n) {
        rc = hipGraphGetNodes((hipGraph_t)graph, nodes.data(), &n);
        if (rc != hipSuccess) return (int)rc;
    }
    int kernels = 0;
    for (size_t i = 0; i < n; ++i) {
        hipGraphNodeType type;
        rc = hipGraphNodeGetType(nodes[i], &type);
        if (rc != hipSuccess) return (int)rc;
        kernels += type == hipGraphNodeTypeKernel;
    }
    *kernel_nodes = kernels;
    *all_nodes = (int)n;
    return SEI_OK;
}
