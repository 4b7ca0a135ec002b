// does a partially masked 16-byte gather cost less on the CU's vector-memory path?  (decides whether caching the 8
// corners of the previous Broyden fetch in registers can pay: 43 % of the fetches stay in the same voxel cell)
// build: hipcc --offload-arch=gfx950 -O2 -o tools/probes/tcp_mask_probe tools/probes/tcp_mask_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void gather(const float4* __restrict__ tab, uint32_t n_vox, int iters, int mode, float* out)
{
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    uint32_t s = tid * 2654435761u + 12345u;
    float acc = 0.f;
    for (int it = 0; it < iters; it++) {
        s = s * 1664525u + 1013904223u;
        bool active = true;
        if (mode == 1) active = ((s >> 9) % 100u) < 57u;          // random 57 % of the lanes
        if (mode == 2) active = ((lane >> 2) & 1) == 0;           // every other quad (50 %)
        if (mode == 3) active = lane < 32;                        // first half of the wave (50 %)
        if (mode == 4) active = (lane & 1) == 0;                  // every other lane (50 %)
        const uint32_t v = (s >> 4) % (n_vox - 200u);
        if (active) {
#pragma unroll
            for (int c = 0; c < 4; c++) {                         // 4 x 96 contiguous bytes, like the trilinear fetch
                const float4* p = tab + (size_t)(v + c * 37u) * 3u;
#pragma unroll
                for (int k = 0; k < 6; k++) { const float4 q = p[k]; acc += q.x + q.y + q.z + q.w; }
            }
        }
    }
    out[tid] = acc;
}

// cooperative variant: 8 lanes share one 96-byte run (lanes 0..5 of the group read its six 16-byte pieces), so a wave
// instruction touches 8 runs (<= 16 lines) instead of 64 lines.  Same bytes per "fetch" (4 runs) as gather(): one wave
// serves 64 fetches with 4 passes x 8 instructions.
__global__ __launch_bounds__(256) void gather_coop(const float4* __restrict__ tab, uint32_t n_vox, int iters, float* out)
{
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int grp = lane >> 3, piece = lane & 7;
    uint32_t s = tid * 2654435761u + 12345u;
    float acc = 0.f;
    for (int it = 0; it < iters; it++) {
        s = s * 1664525u + 1013904223u;
        const uint32_t v_own = (s >> 4) % (n_vox - 200u);          // this lane's fetch (voxel index)
#pragma unroll
        for (int c = 0; c < 4; c++) {                              // pass = run c of every fetch
#pragma unroll
            for (int k = 0; k < 8; k++) {                          // instruction k serves the fetches of lanes 8k .. 8k+7
                const uint32_t v = __shfl(v_own, 8 * k + grp, 64);
                if (piece < 6) {
                    const float4 q = tab[(size_t)(v + c * 37u) * 3u + piece];
                    acc += q.x + q.y + q.z + q.w;
                }
            }
        }
    }
    out[tid] = acc;
}

int main()
{
    const uint32_t n_vox = 32 * 128 * 128;
    float4* tab; float* out;
    hipMalloc(&tab, (size_t)n_vox * 48 + 4096);
    hipMemset(tab, 0, (size_t)n_vox * 48 + 4096);
    const int grid = 256 * 20, iters = 200;
    hipMalloc(&out, grid * 256 * sizeof(float));
    const char* names[] = {"all lanes", "random 57%", "alternate quads", "half wave", "alternate lanes"};
    for (int mode = 0; mode < 5; mode++) {
        gather<<<grid, 256>>>(tab, n_vox, 10, mode, out);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        gather<<<grid, 256>>>(tab, n_vox, iters, mode, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fetches = (double)grid * 256 * iters;
        printf("%-16s %7.3f ms   %.2f G lane-fetches/s issued (384 B each when active)\n", names[mode], ms, fetches / ms / 1e6);
    }
    {
        gather_coop<<<grid, 256>>>(tab, n_vox, 10, out);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        gather_coop<<<grid, 256>>>(tab, n_vox, iters, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fetches = (double)grid * 256 * iters;
        printf("%-16s %7.3f ms   %.2f G lane-fetches/s (384 B each, 8 lanes per 96-byte run)\n", "cooperative", ms, fetches / ms / 1e6);
    }
    return 0;
}
