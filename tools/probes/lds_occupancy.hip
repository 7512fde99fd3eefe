// tools/probes/lds_occupancy.hip -- round 6: how many workgroups of 320 / 576 threads with S bytes of LDS does a CU of this device hold?
// (hipOccupancyMaxActiveBlocksPerMultiprocessor; and, measured: a kernel whose workgroups spin until `expect` of them have arrived on
// every CU would hang if the answer were optimistic, so the probe instead counts the distinct workgroups alive at the same time.)
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/lds_occupancy tools/probes/lds_occupancy.hip && /tmp/lds_occupancy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
extern __shared__ unsigned char dyn[];
__global__ void k_hold(unsigned* alive, unsigned* peak, int spin)
{
    if (threadIdx.x == 0) {
        dyn[0] = 1;
        const unsigned a = atomicAdd(alive, 1u) + 1u;
        atomicMax(peak, a);
        for (int i = 0; i < spin; i++) __builtin_amdgcn_s_sleep(100);
        atomicSub(alive, 1u);
    }
    __syncthreads();
}
int main()
{
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    printf("%s: %d CUs, %zu bytes of LDS per workgroup (maxSharedMemoryPerMultiProcessor %zu)\n", pr.gcnArchName, pr.multiProcessorCount, pr.sharedMemPerBlock, pr.maxSharedMemoryPerMultiProcessor);
    unsigned* d; hipMalloc(&d, 8);
    printf("threads   LDS KB   occupancy API   measured workgroups alive at once / CUs\n");
    for (int threads : {320, 576}) for (int kb : {16, 32, 36, 40, 48, 52, 56, 60, 64, 68, 72, 80, 100, 136, 150}) {
        const size_t lds = (size_t)kb * 1024;
        hipFuncSetAttribute((const void*)k_hold, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        int nb = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_hold, threads, lds);
        hipMemset(d, 0, 8);
        hipLaunchKernelGGL(k_hold, dim3(pr.multiProcessorCount * 8), dim3(threads), lds, 0, d, d + 1, 2000);
        hipDeviceSynchronize();
        unsigned h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("%7d %8d %15d %12u = %.2f per CU\n", threads, kb, nb, h[1], (double)h[1] / pr.multiProcessorCount);
    }
    return 0;
}
