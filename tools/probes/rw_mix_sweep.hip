// tools/probes/rw_mix_sweep.hip -- does the memory system give the band kernel's read / write mix MORE when more waves issue it?  (dev probe)
// The mix of tools/probes/pmc_calib.hip (calib_rw_b64_band: per wave-step one 8 B-per-lane load of a 128-byte pixel and one non-temporal
// 8 B-per-lane store of a 128-byte pixel, 4 skewed rows per wave, 1/2 GiB each way) with the same bytes cut into more and shorter rows, i.e.
// more waves: 1 024 ... 32 768 (the band launch has 2 048 compute waves, 8 per CU; a CU holds 32), with the load queue of each wave kept 1, 4 or
// 16 steps deep as the band kernel's prefetch does, and the pure-write and pure-read legs beside it.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/rw_mix_sweep tools/probes/rw_mix_sweep.hip && tools/probes/rw_mix_sweep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
#define BUF_FLAGS 0x00020000
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, size_t n) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)n, BUF_FLAGS); }

// MODE 0: load + store, 1: store only, 2: load only.  rowlen pixels per row; rows = half / (rowlen * 128).  PF loads in flight per wave.
template <int MODE, int PF>
__global__ __launch_bounds__(256) void k_mix(uint8_t* buf, uint32_t* sink, size_t bytes, int rowlen, int rows)
{
    const __amdgpu_buffer_rsrc_t rs = rsrc(buf, bytes);
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6, j = lane >> 4, gl = lane & 15;
    const int row = wave * 4 + j;
    const size_t half = bytes / 2;
    uint32_t acc = 0;
    if (row >= rows) return;
    u32x2 q[PF];
    auto roff = [&](int T) -> uint32_t { const int x = T - j; return (x >= 0 && x < rowlen) ? (uint32_t)(((size_t)row * rowlen + x) * 128 + gl * 8) : 0xffffffffu; };
    if (MODE != 1) {
        #pragma unroll
        for (int i = 0; i < PF; i++) q[i] = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)roff(i), 0, 0);
    }
    for (int T0 = 0; T0 < rowlen + 4; T0 += PF) {
        #pragma unroll
        for (int i = 0; i < PF; i++) {
            const int T = T0 + i;
            if (MODE != 1) { acc ^= q[i].x ^ q[i].y; q[i] = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)roff(T + PF), 0, 0); }
            if (MODE != 2) {
                const uint32_t o = roff(T);
                u32x2 v; v.x = (uint32_t)T; v.y = (uint32_t)row;
                __builtin_amdgcn_raw_buffer_store_b64(v, rs, (int)(o == 0xffffffffu ? o : (uint32_t)(half + o)), 0, 2);
            }
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main()
{
    const size_t bytes = (size_t)1 << 30;
    uint8_t* a; uint32_t* sink;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&sink, 256) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipMemset(a, 1, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](auto&& f) { f(); hipEventRecord(e0); f(); f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 2; };
    printf("the band kernel's traffic mix by number of waves (half of a 1 GiB buffer read, the other half written; TB/s of the bytes moved)\n");
    printf("%8s %8s | %14s %14s %14s | %12s %12s\n", "waves", "row px", "mix, 1 deep", "mix, 4 deep", "mix, 16 deep", "write only", "read only 16");
    for (int rowlen : {1024, 512, 256, 128, 64, 32}) {
        const int rows = (int)(bytes / 2 / ((size_t)rowlen * 128)), waves = rows / 4, blocks = (waves + 3) / 4;
        const float m1 = timeit([&] { hipLaunchKernelGGL((k_mix<0, 1>), dim3(blocks), dim3(256), 0, 0, a, sink, bytes, rowlen, rows); });
        const float m4 = timeit([&] { hipLaunchKernelGGL((k_mix<0, 4>), dim3(blocks), dim3(256), 0, 0, a, sink, bytes, rowlen, rows); });
        const float m16 = timeit([&] { hipLaunchKernelGGL((k_mix<0, 16>), dim3(blocks), dim3(256), 0, 0, a, sink, bytes, rowlen, rows); });
        const float w = timeit([&] { hipLaunchKernelGGL((k_mix<1, 1>), dim3(blocks), dim3(256), 0, 0, a, sink, bytes, rowlen, rows); });
        const float r = timeit([&] { hipLaunchKernelGGL((k_mix<2, 16>), dim3(blocks), dim3(256), 0, 0, a, sink, bytes, rowlen, rows); });
        const double gb = bytes / 1e9, hb = gb / 2;
        printf("%8d %8d | %9.2f TB/s %9.2f TB/s %9.2f TB/s | %7.2f TB/s %7.2f TB/s\n", waves, rowlen, gb / m1, gb / m4, gb / m16, hb / w, hb / r);
    }
    return 0;
}
