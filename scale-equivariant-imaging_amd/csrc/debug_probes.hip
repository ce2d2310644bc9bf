// Hardware probes used while developing kernels (tools/probe_tr_read.py). Not on any product path.
#include "sei_common.h"

namespace {
typedef short v4s __attribute__((ext_vector_type(4)));

// LDS image: 64 rows x 128 columns of 16-bit values, plain row-major (256-byte rows, no swizzle).
// Each 16-lane group g reads the 4x16 block with top-left (r0 + 4*g, c0): lane 4q+p supplies the
// address of row q, columns 4p..4p+3 (8 bytes).
__global__ void tr_probe_kernel(const unsigned short *in, unsigned short *out, int r0, int c0) {
    __shared__ __attribute__((aligned(16))) unsigned short sm[64 * 128];
    for (int i = threadIdx.x; i < 64 * 128; i += 64) sm[i] = in[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
    auto *ptr = (__attribute__((address_space(3))) v4s *)(sm + (r0 + 4 * g + q) * 128 + c0 + 4 * p);
    const v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16(ptr);
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = (unsigned short)r[e];
}
}  // namespace

extern "C" int sei_debug_tr_probe(const uint16_t *in, uint16_t *out, int r0, int c0, void *stream) {
    SEI_REQUIRE(in && out);
    hipLaunchKernelGGL(tr_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, in, out, r0, c0);
    return sei_launch_status();
}
