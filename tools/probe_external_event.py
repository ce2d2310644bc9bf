"""Probe: which way of putting an event-record node into a torch-captured hipGraph does HIP accept?"""
import ctypes, torch
hip = ctypes.CDLL("libamdhip64.so")
P = ctypes.c_void_p
def mk(flags):
    e = P(); rc = hip.hipEventCreateWithFlags(ctypes.byref(e), ctypes.c_uint(flags)); assert rc == 0, rc; return e
x = torch.zeros(1 << 20, device="cuda"); y = torch.zeros_like(x)
for name, flags in (("default-flags event", 0), ("disable-timing event", 2)):
    ev = mk(flags)
    g = torch.cuda.CUDAGraph()
    rc_rec = None
    try:
        with torch.cuda.graph(g):
            x.add_(1.0)
            s = torch.cuda.current_stream().cuda_stream
            rc_rec = hip.hipEventRecordWithFlags(ev, P(s), ctypes.c_uint(1))
            y.add_(1.0)
        print(name, ": hipEventRecordWithFlags(External) in capture ->", rc_rec)
    except Exception as ex:
        print(name, ": capture raised", type(ex).__name__, str(ex)[:120], "rc", rc_rec)
# explicit node insertion
ev = mk(2)
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        x.add_(1.0)
        s = P(torch.cuda.current_stream().cuda_stream)
        status = ctypes.c_int(); cid = ctypes.c_ulonglong(); graph = P(); deps = ctypes.POINTER(P)(); ndeps = ctypes.c_size_t()
        rc1 = hip.hipStreamGetCaptureInfo_v2(s, ctypes.byref(status), ctypes.byref(cid), ctypes.byref(graph), ctypes.byref(deps), ctypes.byref(ndeps))
        node = P()
        rc2 = hip.hipGraphAddEventRecordNode(ctypes.byref(node), graph, deps, ndeps, ev)
        rc3 = hip.hipStreamUpdateCaptureDependencies(s, ctypes.byref(node), ctypes.c_size_t(1), ctypes.c_uint(1))
        y.add_(1.0)
    print("explicit node: getinfo", rc1, "status", status.value, "ndeps", ndeps.value, "addnode", rc2, "updatedeps", rc3)
    side = torch.cuda.Stream()
    for it in range(2):
        x.zero_(); y.zero_(); torch.cuda.synchronize()
        g.replay()
        rcw = hip.hipStreamWaitEvent(P(side.cuda_stream), ev, ctypes.c_uint(0))
        with torch.cuda.stream(side):
            z = x.clone()
        torch.cuda.synchronize()
        print("replay", it, "wait rc", rcw, "side stream saw x =", float(z[0]), "(1.0 = waited for the node)")
except Exception as ex:
    print("explicit node: raised", type(ex).__name__, str(ex)[:200])
