// tools/probes/granule_bw.hip -- what the memory system gives the MGM band kernel's access pattern (dev probe, not shipped).
// A wave owns 4 rows (lane groups of 16 lanes x 8 bytes = one 128-byte pixel each) and sweeps them with a skew of one step per
// row, loading the pixel's 128 bytes of one volume (8 steps ahead) and storing 128 bytes to another (non-temporal), as
// k_mgm_bands does with C and e -- without its arithmetic, LDS traffic or hand-offs.  Layouts of the two volumes:
//   rows    row-major [row][x][128]: the wave's 4 pixels are 4 image rows apart (128 KB): the axis lattices today
//   il4     4 rows interleaved per pixel column [row / 4][x][row % 4][128]: the wave's 4 pixels of a step sit within 1.2 KB
//   diag    the wave's 4 pixels in one image row, 256 bytes apart: the diagonal lattices today
// hipcc --offload-arch=gfx950 -O3 -o tools/probes/granule_bw tools/probes/granule_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#define W 1024
#define PF 8
template <int LAYOUT>
__device__ __forceinline__ size_t addr(int row, int x)
{
    if (LAYOUT == 0) return ((size_t)row * W + x) * 128;
    if (LAYOUT == 1) return ((size_t)(row >> 2) * W + x) * 512 + (size_t)(row & 3) * 128;
    return ((size_t)(row >> 2) * 4 * W + (size_t)(x * 4 + (row & 3) * 2) % (4 * W)) * 128;      // 4 "rows" share one long image row
}
template <int LAYOUT, bool NTLOAD>
__global__ __launch_bounds__(256) void k_sweep(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int nrows)
{
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6, j = lane >> 4, gl = lane & 15;
    const int row = wave * 4 + j;
    if (row >= nrows) return;
    u32x2 q[PF];
    #pragma unroll
    for (int i = 0; i < PF; i++) {
        const int x = i - j;
        const u32x2* p = reinterpret_cast<const u32x2*>(src + addr<LAYOUT>(row, x < 0 ? 0 : x) + gl * 8);
        q[i] = NTLOAD ? __builtin_nontemporal_load(p) : *p;
    }
    for (int T = 0; T < W + 4; T += PF) {
        #pragma unroll
        for (int i = 0; i < PF; i++) {
            const int x = T + i - j;
            u32x2 v = q[i];
            const int xn = x + PF;
            const u32x2* p = reinterpret_cast<const u32x2*>(src + addr<LAYOUT>(row, xn < 0 ? 0 : (xn >= W ? W - 1 : xn)) + gl * 8);
            q[i] = NTLOAD ? __builtin_nontemporal_load(p) : *p;
            v.x = v.x * 3u + 1u; v.y ^= v.x;
            if (x >= 0 && x < W) __builtin_nontemporal_store(v, reinterpret_cast<u32x2*>(dst + addr<LAYOUT>(row, x) + gl * 8));
        }
    }
}
int main()
{
    const int nrows = 8192;                                    // 8192 rows x 1024 px x 128 B = 1 GiB per volume
    const size_t bytes = (size_t)nrows * W * 128;
    uint8_t *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, auto&& f) {
        for (int i = 0; i < 2; i++) f();
        hipEventRecord(e0); for (int i = 0; i < 5; i++) f(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("%-44s %.3f ms  %.2f TB/s (read + write)\n", name, ms, 2 * bytes / 1e9 / ms);
    };
    const int blocks = nrows / 16;                             // 4 waves x 4 rows per block: 512 blocks = 2 per CU
    timeit("rows  (4 image rows per wave-step)", [&] { hipLaunchKernelGGL((k_sweep<0, false>), dim3(blocks), dim3(256), 0, 0, a, b, nrows); });
    timeit("il4   (4 rows interleaved per column)", [&] { hipLaunchKernelGGL((k_sweep<1, false>), dim3(blocks), dim3(256), 0, 0, a, b, nrows); });
    timeit("diag  (4 pixels of one row, 256 B apart)", [&] { hipLaunchKernelGGL((k_sweep<2, false>), dim3(blocks), dim3(256), 0, 0, a, b, nrows); });
    timeit("rows, nt loads", [&] { hipLaunchKernelGGL((k_sweep<0, true>), dim3(blocks), dim3(256), 0, 0, a, b, nrows); });
    timeit("il4, nt loads", [&] { hipLaunchKernelGGL((k_sweep<1, true>), dim3(blocks), dim3(256), 0, 0, a, b, nrows); });
    return 0;
}
