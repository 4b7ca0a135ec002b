// which XCD does workgroup b run on?  (the XCD-partitioned hash forward assumes b % 8; used for speed only)
// build: hipcc --offload-arch=gfx950 -O2 -o tools/probes/xcc_probe tools/probes/xcc_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ void probe(int* xcc)
{
    if (threadIdx.x == 0) {
        // HW_REG_XCC_ID = 20, bits [3:0]
        const unsigned v = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
        xcc[blockIdx.x] = (int)(v & 15);
    }
}

int main()
{
    const int n = 8192;
    int* d;
    hipMalloc(&d, n * sizeof(int));
    for (int rep = 0; rep < 3; rep++) {
        probe<<<n, 256>>>(d);
        std::vector<int> h(n);
        hipMemcpy(h.data(), d, n * sizeof(int), hipMemcpyDeviceToHost);
        int match = 0, hist[16] = {0};
        for (int b = 0; b < n; b++) { match += (h[b] == (b & 7)); hist[h[b] & 15]++; }
        printf("rep %d: %d of %d workgroups on XCD (b %% 8); per-XCD counts:", rep, match, n);
        for (int k = 0; k < 8; k++) printf(" %d", hist[k]);
        printf("   first 16:");
        for (int b = 0; b < 16; b++) printf(" %d", h[b]);
        printf("\n");
    }
    return 0;
}
