// Micro-probe: what HW_REG_LDS_ALLOC and HW_REG_HW_ID read for the workgroups that share a CU (256 threads, 64 KiB of LDS).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <map>
__global__ __launch_bounds__(256) void k(unsigned *out) {
    __shared__ float big[16384];
    big[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) {
        out[blockIdx.x * 4 + 0] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 6);   // LDS_ALLOC
        out[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);   // HW_ID
        out[blockIdx.x * 4 + 2] = (unsigned)big[5];
        out[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 23);  // XCC_ID (gfx940+)
    }
    for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(100);   // keep the workgroups resident together
}
int main() {
    unsigned *d; hipMalloc(&d, 512 * 16);
    hipLaunchKernelGGL(k, dim3(512), dim3(256), 0, 0, d);
    unsigned h[2048]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    std::map<unsigned, int> la;
    for (int b = 0; b < 512; ++b) la[h[b * 4]]++;
    for (auto &kv : la) printf("LDS_ALLOC 0x%08x: %d workgroups\n", kv.first, kv.second);
    for (int b = 0; b < 24; ++b) printf("block %3d: LDS_ALLOC 0x%08x HW_ID 0x%08x XCC_ID 0x%08x\n", b, h[b * 4], h[b * 4 + 1], h[b * 4 + 3]);
    return 0;
}
