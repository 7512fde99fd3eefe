// tools/probes/cumask_map.hip -- what a hipExtStreamCreateWithCUMask bit selects on this device (dev probe, not shipped).
// For a mask of the `lo` lowest and / or `hi` highest bits of the device's CU count: which (XCC, SE, CU) ran workgroups, how many
// distinct CUs per XCC, and the streaming read rate those CUs reach alone.
// hipcc --offload-arch=gfx950 -O3 -o tools/probes/cumask_map tools/probes/cumask_map.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <map>
#include <vector>

__global__ __launch_bounds__(64) void k_where(uint32_t* out, int spin)
{
    const uint32_t hw = __builtin_amdgcn_s_getreg(4 << 0 | 0 << 6 | 31 << 11);     // HW_REG_HW_ID
    const uint32_t xcc = __builtin_amdgcn_s_getreg(20 << 0 | 0 << 6 | 31 << 11);   // HW_REG_XCC_ID
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(8);   // stay resident so that later blocks spread
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_read(const u32x4* __restrict__ p, size_t n, u32x4* out)
{
    u32x4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc ^= __builtin_nontemporal_load(p + i);
    if (acc.x == 0x12345678u) out[0] = acc;
}

static hipStream_t masked_stream(int ncu, int lo, int hi)
{
    std::vector<uint32_t> m((ncu + 31) / 32, 0u);
    for (int i = 0; i < lo && i < ncu; i++) m[i / 32] |= 1u << (i % 32);
    for (int i = 0; i < hi && i < ncu; i++) { const int b = ncu - 1 - i; m[b / 32] |= 1u << (b % 32); }
    hipStream_t st = nullptr;
    if (hipExtStreamCreateWithCUMask(&st, (uint32_t)m.size(), m.data()) != hipSuccess) { printf("hipExtStreamCreateWithCUMask failed\n"); exit(1); }
    return st;
}

int main(int argc, char** argv)
{
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int ncu = pr.multiProcessorCount;
    printf("device: %s, %d CUs\n", pr.name, ncu);
    const size_t bytes = (size_t)1 << 30;
    u32x4 *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, 4096); hipMemset(a, 1, bytes);
    uint32_t* d_out; const int nblk = 8192; hipMalloc(&d_out, nblk * 8);
    std::vector<uint32_t> h(nblk * 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct Case { int lo, hi; };
    std::vector<Case> cases;
    if (argc >= 3) cases.push_back({atoi(argv[1]), atoi(argv[2])});
    else for (Case c : {Case{ncu, 0}, Case{8, 0}, Case{32, 0}, Case{64, 0}, Case{128, 0}, Case{192, 0}, Case{224, 0}, Case{0, 32}, Case{0, 64}, Case{0, 96}, Case{0, 128}, Case{1, 0}, Case{0, 1}, Case{9, 0}}) cases.push_back(c);
    for (Case c : cases) {
        hipStream_t st = masked_stream(ncu, c.lo, c.hi);
        hipMemsetAsync(d_out, 0xff, nblk * 8, st);
        hipLaunchKernelGGL(k_where, dim3(nblk), dim3(64), 0, st, d_out, 2000);    // 2000 ticks of the 100 MHz clock = 20 us per block
        hipMemcpyAsync(h.data(), d_out, nblk * 8, hipMemcpyDeviceToHost, st);
        hipStreamSynchronize(st);
        std::map<int, std::set<int>> per_xcc;
        for (int i = 0; i < nblk; i++) {
            const uint32_t hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
            const int cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
            per_xcc[xcc].insert(se << 8 | sh << 4 | cu);
        }
        int total = 0;
        printf("mask lo %3d hi %3d: CUs per XCC", c.lo, c.hi);
        for (auto& kv : per_xcc) { printf(" %d:%zu", kv.first, kv.second.size()); total += (int)kv.second.size(); }
        // streaming read under the mask
        for (int i = 0; i < 2; i++) hipLaunchKernelGGL(k_read, dim3(4096), dim3(256), 0, st, a, bytes / 16, b);
        hipEventRecord(e0, st);
        for (int i = 0; i < 5; i++) hipLaunchKernelGGL(k_read, dim3(4096), dim3(256), 0, st, a, bytes / 16, b);
        hipEventRecord(e1, st); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("  = %d distinct;  1 GiB nt read %.3f ms = %.2f TB/s\n", total, ms, bytes / 1e9 / ms);
        if (c.lo + c.hi <= 9) {
            printf("    (se, sh, cu) per XCC:");
            for (auto& kv : per_xcc) for (int v : kv.second) printf(" x%d:(%d,%d,%d)", kv.first, v >> 8, (v >> 4) & 1, v & 0xf);
            printf("\n");
        }
        hipStreamDestroy(st);
    }
    return 0;
}
