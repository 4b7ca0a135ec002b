// LDS atomic throughput probe: ds_add_f32 vs ds_add_u32 vs ds_add_u64 on pseudo-random addresses of a 64 KB table.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int MODE>
__global__ __launch_bounds__(256) void k(int iters, float* out)
{
    __shared__ float tab[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) tab[i] = 0.f;
    __syncthreads();
    uint32_t s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int it = 0; it < iters; it++) {
        s = s * 1664525u + 1013904223u;
        const uint32_t idx = (s >> 10) & 8191u;
        if (MODE == 0) { unsafeAtomicAdd(tab + 2 * idx, 1.0f); unsafeAtomicAdd(tab + 2 * idx + 1, 0.5f); }
        if (MODE == 1) { atomicAdd((unsigned*)tab + 2 * idx, 3u); atomicAdd((unsigned*)tab + 2 * idx + 1, 5u); }
        if (MODE == 2) { atomicAdd((unsigned long long*)tab + idx, 0x0000000300000005ull); }
        if (MODE == 3) { tab[2 * idx] += 1.0f; tab[2 * idx + 1] += 0.5f; }     // non-atomic RMW (wrong, rate reference)
        if (MODE == 4) { unsafeAtomicAdd(tab + 2 * (idx & ~63u) + 2 * (threadIdx.x & 63), 1.0f); unsafeAtomicAdd(tab + 2 * (idx & ~63u) + 2 * (threadIdx.x & 63) + 1, 0.5f); }   // conflict-free
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = tab[5] + tab[1000];
}
int main()
{
    float* out; hipMalloc(&out, 4096 * 4);
    const int iters = 4096, grid = 512;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[5] = {"ds_add_f32 x2", "ds_add_u32 x2", "ds_add_u64 x1", "plain rmw x2", "ds_add_f32 x2 conflict-free"};
    for (int m = 0; m < 5; m++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (m == 0) k<0><<<grid, 256>>>(iters, out); if (m == 1) k<1><<<grid, 256>>>(iters, out);
            if (m == 2) k<2><<<grid, 256>>>(iters, out); if (m == 3) k<3><<<grid, 256>>>(iters, out);
            if (m == 4) k<4><<<grid, 256>>>(iters, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep == 1) printf("%-30s %.3f ms  -> %.1f G record-updates/s (%.2f per clk per CU @2.4GHz, 256 CUs)\n", names[m], ms,
                                 (double)grid * 256 * iters / ms / 1e6, (double)grid * 256 * iters / (ms * 1e-3) / 256 / 2.4e9);
        }
    }
    return 0;
}
