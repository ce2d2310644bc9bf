// Tools-only probe (tools/probe_torch_philox.py): rocRAND's Philox4x32-10 normal4 / uniform4 per subsequence, to find out
// how torch's HIP generator maps (seed, offset, element) to engine calls. Built twice: contraction fast / off.
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_philox4x32_10.h>
#include <rocrand/rocrand_normal.h>
#include <rocrand/rocrand_uniform.h>
__global__ void probe_kernel(unsigned long long seed, unsigned long long offset, int n, float *normal4, float *uniform4,
                             unsigned *raw4) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    rocrand_state_philox4x32_10 st;
    rocrand_init(seed, i, offset, &st);
    const float4 v = rocrand_normal4(&st);
    normal4[4 * i] = v.x; normal4[4 * i + 1] = v.y; normal4[4 * i + 2] = v.z; normal4[4 * i + 3] = v.w;
    rocrand_init(seed, i, offset, &st);
    const float4 u = rocrand_uniform4(&st);
    uniform4[4 * i] = u.x; uniform4[4 * i + 1] = u.y; uniform4[4 * i + 2] = u.z; uniform4[4 * i + 3] = u.w;
    rocrand_init(seed, i, offset, &st);
    const uint4 r = rocrand4(&st);
    raw4[4 * i] = r.x; raw4[4 * i + 1] = r.y; raw4[4 * i + 2] = r.z; raw4[4 * i + 3] = r.w;
}
extern "C" int philox_probe(unsigned long long seed, unsigned long long offset, int n, float *normal4, float *uniform4,
                            unsigned *raw4, void *stream) {
    hipLaunchKernelGGL(probe_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, seed, offset, n, normal4,
                       uniform4, raw4);
    return (int)hipGetLastError();
}
