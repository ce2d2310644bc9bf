// How fast can 256 workgroups (one per CU) stream the 8192 x 32768 bf16 weight of the bottleneck level the way the GEMM
// reads it? Pattern 0: row-major (N, K) storage, a k-tile of a workgroup = 256 rows x 128 B at a 16-KB pitch (what
// gemm_bf16pq_kernel's B pieces are); pattern 1: the same bytes stored tile by tile (32 KB contiguous per k-tile).
// LOADS = 16-byte loads in flight per thread (4 = one k-tile). Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/micro/weight_stream.hip -o /tmp/ws && /tmp/ws
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int PATTERN, int LOADS>
__global__ __launch_bounds__(512) void stream_kernel(const char *W, unsigned *sink, int ktiles_per_wg) {
    __shared__ char big[140 * 1024];                  // one workgroup per CU, as the GEMM
    const int w = blockIdx.x, t = threadIdx.x;
    const int ntile = w % 128, khalf = w / 128;
    u32x4 acc = {0, 0, 0, 0};
    constexpr int TILES = LOADS / 4;                  // k-tiles per iteration
    for (int kt = 0; kt < ktiles_per_wg; kt += TILES) {
        u32x4 v[LOADS];
#pragma unroll
        for (int j = 0; j < LOADS; ++j) {
            const int k = khalf * ktiles_per_wg + kt + j / 4;
            const int idx = (j % 4) * 512 + t;        // 2048 x 16 B = 256 rows x 128 B
            size_t off;
            if (PATTERN == 0) off = (size_t)(ntile * 256 + idx / 8) * 16384 + (size_t)k * 128 + (idx % 8) * 16;
            else off = ((size_t)ntile * 128 + k) * 32768 + (size_t)idx * 16;
            v[j] = *reinterpret_cast<const u32x4 *>(W + off);
        }
#pragma unroll
        for (int j = 0; j < LOADS; ++j) acc ^= v[j];
    }
    if (acc.x == 0x12345678u && acc.y == 1u) { big[t] = 1; sink[w] = acc.z + big[(t + 1) % 512]; }
}
template <int PATTERN, int LOADS>
void run(const char *W, unsigned *sink) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((stream_kernel<PATTERN, LOADS>), dim3(256), dim3(512), 0, 0, W, sink, 64);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (r > 0 && ms < best) best = ms;
    }
    printf("pattern %d (%s), %2d loads in flight per thread: %7.1f us  %5.2f TB/s\n", PATTERN, PATTERN ? "tile-major" : "row-major ", LOADS,
           best * 1e3, 536.870912e6 / (best * 1e-3) / 1e12);
}
int main() {
    char *W; unsigned *sink;
    hipMalloc(&W, (size_t)8192 * 32768 * 2); hipMalloc(&sink, 4096);
    hipMemset(W, 1, (size_t)8192 * 32768 * 2);
    run<0, 4>(W, sink); run<1, 4>(W, sink);
    run<0, 8>(W, sink); run<1, 8>(W, sink);
    run<0, 16>(W, sink); run<1, 16>(W, sink);
    run<0, 32>(W, sink); run<1, 32>(W, sink);
    return 0;
}
