// tools/probes/hbm_bw.hip -- streaming HBM rates on this device (dev probe, not shipped): pure read, pure write, copy.
// hipcc --offload-arch=gfx950 -O3 -o tools/probes/hbm_bw tools/probes/hbm_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k_read(const uint4* __restrict__ p, size_t n, uint4* out)
{
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        uint4 v = p[i];
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    if (acc.x == 0x12345678u) out[0] = acc;
}
// nine streams side by side, 8 or 16 bytes per lane (the WTA pattern)
template <typename T>
__global__ __launch_bounds__(256) void k_read9(const T* __restrict__ p, size_t nper, T* out)
{
    T acc = {};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nper; i += (size_t)gridDim.x * 256) {
        #pragma unroll
        for (int r = 0; r < 9; r++) { T v = p[r * nper + i]; acc.x ^= v.x; acc.y ^= v.y; }
    }
    if (acc.x == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_write(uint4* __restrict__ p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = make_uint4(1, 2, 3, 4);
}
__global__ __launch_bounds__(256) void k_copy(const uint4* __restrict__ s, uint4* __restrict__ d, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d[i] = s[i];
}
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_read_nt(const u32x4* __restrict__ p, size_t n, u32x4* out)
{
    u32x4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc ^= __builtin_nontemporal_load(p + i);
    if (acc.x == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_write_nt(u32x4* __restrict__ p, size_t n)
{
    const u32x4 v = {1, 2, 3, 4};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) __builtin_nontemporal_store(v, p + i);
}
int main()
{
    const size_t bytes = 1207959552;   // 9 x 128 MiB, the census WTA's read set
    uint4 *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, double gb, auto&& f) {
        for (int i = 0; i < 2; i++) f();
        hipEventRecord(e0); for (int i = 0; i < 10; i++) f(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        printf("%-34s %.3f ms  %.2f TB/s\n", name, ms, gb / ms);
    };
    const size_t n = bytes / 16;
    for (int blocks : {1024, 2048, 4096, 16384}) {
        char nm[64];
        snprintf(nm, 64, "read b128 %d blocks", blocks); timeit(nm, bytes / 1e9, [&] { hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, a, n, b); });
        snprintf(nm, 64, "read 9 streams b64 %d blocks", blocks); timeit(nm, bytes / 1e9, [&] { hipLaunchKernelGGL(k_read9<uint2>, dim3(blocks), dim3(256), 0, 0, (const uint2*)a, bytes / 9 / 8, (uint2*)b); });
        snprintf(nm, 64, "read 9 streams b128 %d blocks", blocks); timeit(nm, bytes / 1e9, [&] { hipLaunchKernelGGL(k_read9<uint4>, dim3(blocks), dim3(256), 0, 0, a, bytes / 9 / 16, b); });
        snprintf(nm, 64, "write b128 %d blocks", blocks); timeit(nm, bytes / 1e9, [&] { hipLaunchKernelGGL(k_write, dim3(blocks), dim3(256), 0, 0, b, n); });
        snprintf(nm, 64, "copy b128 %d blocks (r+w bytes)", blocks); timeit(nm, 2 * bytes / 1e9, [&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, a, b, n); });
    }
    for (int blocks : {2048, 16384}) {
        char nm[64];
        snprintf(nm, 64, "read nt b128 %d blocks", blocks); timeit(nm, bytes / 1e9, [&] { hipLaunchKernelGGL(k_read_nt, dim3(blocks), dim3(256), 0, 0, (const u32x4*)a, n, (u32x4*)b); });
        snprintf(nm, 64, "write nt b128 %d blocks", blocks); timeit(nm, bytes / 1e9, [&] { hipLaunchKernelGGL(k_write_nt, dim3(blocks), dim3(256), 0, 0, (u32x4*)b, n); });
    }
    {
        float ms_w, ms_r;
        for (int rep = 0; rep < 3; rep++) {
            hipEvent_t a0, a1, a2; hipEventCreate(&a0); hipEventCreate(&a1); hipEventCreate(&a2);
            hipEventRecord(a0);
            hipLaunchKernelGGL(k_write_nt, dim3(2048), dim3(256), 0, 0, (u32x4*)b, n);
            hipEventRecord(a1);
            hipLaunchKernelGGL(k_read_nt, dim3(2048), dim3(256), 0, 0, (const u32x4*)b, n, (u32x4*)a);
            hipEventRecord(a2); hipEventSynchronize(a2);
            hipEventElapsedTime(&ms_w, a0, a1); hipEventElapsedTime(&ms_r, a1, a2);
        }
        printf("nt write 1.2 GB then nt read it back: write %.3f ms, read %.3f ms (%.2f TB/s)\n", ms_w, ms_r, bytes / 1e9 / ms_r);
    }
    // a read that follows a large write pays for the predecessor's dirty lines (L2 + Infinity Cache write-back)
    {
        float ms_w, ms_r;
        for (int rep = 0; rep < 3; rep++) {
            hipEvent_t a0, a1, a2; hipEventCreate(&a0); hipEventCreate(&a1); hipEventCreate(&a2);
            hipEventRecord(a0);
            hipLaunchKernelGGL(k_write, dim3(2048), dim3(256), 0, 0, b, n);
            hipEventRecord(a1);
            hipLaunchKernelGGL(k_read9<uint2>, dim3(2048), dim3(256), 0, 0, (const uint2*)b, bytes / 9 / 8, (uint2*)a);
            hipEventRecord(a2); hipEventSynchronize(a2);
            hipEventElapsedTime(&ms_w, a0, a1); hipEventElapsedTime(&ms_r, a1, a2);
        }
        printf("write 1.2 GB then read it back: write %.3f ms, read %.3f ms (%.2f TB/s)\n", ms_w, ms_r, bytes / 1e9 / ms_r);
    }
    return 0;
}
