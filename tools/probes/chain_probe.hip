// tools/probes/chain_probe.hip -- cycles per sample of the LDS-resident prefilter chain (dev probe, not shipped).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -o /tmp/chain_probe tools/probes/chain_probe.hip
#include "../../s2p_amd/csrc/warp_kernels.hip"
#include <cstdio>
using namespace s2p;
__global__ __launch_bounds__(256) void k_probe(int len, int nl, unsigned long long* out)
{
    extern __shared__ float4 lds4[];
    float* lds = (float*)lds4;
    const int S = lds_line_stride(len);
    for (int i = threadIdx.x; i < S * nl; i += 256) lds[i] = (float)(i % 97);
    __syncthreads();
    unsigned long long t0 = clock64(), w0 = wall_clock64();
    if ((int)threadIdx.x < nl) {
        prefilter_pole_lds(lds + threadIdx.x * S, len, BS_Z1);
        prefilter_pole_lds(lds + threadIdx.x * S, len, BS_Z2);
    }
    unsigned long long t1 = clock64(), w1 = wall_clock64();
    __syncthreads();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; out[2] = (unsigned long long)lds[5]; }
}
__global__ void k_fma(float z, float x, unsigned long long* out, float* sink)
{
    float p = x;
    unsigned long long t0 = clock64();
    #pragma unroll 64
    for (int i = 0; i < 4096; i++) p = __builtin_fmaf(z, p, x);
    unsigned long long t1 = clock64();
    float q = x;
    #pragma unroll 64
    for (int i = 0; i < 4096; i++) q = __builtin_fmaf(BS_Z1, q, x);
    unsigned long long t2 = clock64();
    float r = x;
    #pragma unroll 64
    for (int i = 0; i < 4096; i++) { r = z * r; r = r + x; }
    unsigned long long t3 = clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = t2 - t1; out[2] = t3 - t2; sink[0] = p + q + r; }
}
int main()
{
    {
        unsigned long long* d; float* f; hipMalloc(&d, 64); hipMalloc(&f, 64);
        unsigned long long h[3];
        for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k_fma, dim3(1), dim3(64), 0, 0, -0.43f, 1.5f, d, f); hipMemcpy(h, d, 24, hipMemcpyDeviceToHost); }
        printf("dependent fma (reg z) %.2f cyc, fmamk literal %.2f cyc, mul+add %.2f cyc per step\n", h[0] / 4096.0, h[1] / 4096.0, h[2] / 4096.0);
    }
    unsigned long long* d; hipMalloc(&d, 64);
    hipFuncSetAttribute((const void*)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int nl : {1, 8, 32}) for (int len : {1200, 600}) {
        if ((size_t)lds_line_stride(len) * nl * 4 > 160 * 1024) continue;
        unsigned long long h[3];
        for (int rep = 0; rep < 3; rep++) {
            hipLaunchKernelGGL(k_probe, dim3(1), dim3(256), lds_line_stride(len) * nl * 4, 0, len, nl, d);
            hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        }
        printf("nl %2d len %4d: %llu cycles (%.2f per sample-sweep), wall %llu ticks (100 MHz) => %.2f us, %.0f MHz\n", nl, len, h[0],
               (double)h[0] / (4.0 * len), h[1], h[1] / 100.0, h[0] / (h[1] / 100.0));
    }
    return 0;
}
// link stubs for the host symbols warp_kernels.hip references (never called here)
namespace s2p { StageScope::StageScope(s2p_hip_ctx*, const char*) {} StageScope::~StageScope() {} void set_last_error(const char*, ...) {} }
