// ia_zero.h -- zero a small device buffer with a KERNEL instead of hipMemsetAsync: on the launch-rate-bound 4096-ray training step the
// device idled ~30 us in front of every memset of the step (17 of them, profiles/r06_launch_audit_after_stepops.json) against ~5 us in
// front of a kernel.  Included only by the translation units that issue such per-call clears.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ia {

static __global__ void zero_words_kernel(uint32_t* __restrict__ p, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0u;
}

// bytes must be a multiple of 4 and p 4-byte aligned
static inline void zero_bytes(void* p, size_t bytes, hipStream_t s)
{
    const int64_t n = (int64_t)(bytes / 4);
    if (n <= 0) return;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    zero_words_kernel<<<(int)blocks, 256, 0, s>>>(reinterpret_cast<uint32_t*>(p), n);
}

}  // namespace ia
