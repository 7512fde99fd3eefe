// tools/probes/pmc_calib.hip -- known-byte calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the access widths and cache
// policies the matcher's kernels use (dev probe, not shipped).  MI355X_MICROARCH.md (HBM section) calibrates FETCH_SIZE only for
// wide coalesced reads (16 B per lane: the counter reports 1/2 of the bytes) and calls every other width uncalibrated; k_mgm_bands
// loads its costs with raw_buffer_load_b64 (8 B per lane: 16 lanes x 8 B = one 128-byte pixel, 4 pixels of 4 different rows per
// wave-step) and stores its e-bytes with raw_buffer_store_b64 nt; the WTA kernels read them back with b64 nt loads.
//
// Every kernel below moves EXACTLY `bytes` (1 GiB: four times the 256 MiB Infinity Cache, so nothing is served on-die) once, with
// one of those instructions; tools/pmc_calib.sh runs the binary under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`
// (separate passes) and writes, per kernel, counter bytes / known bytes -- the factor bench.py's pmc_traffic() applies.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/pmc_calib tools/probes/pmc_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
#define BUF_FLAGS 0x00020000
#define W 1024                      // pixels per row of the band pattern; a pixel = 128 bytes (D = 128 u8 costs)
#define ROWS 8192                   // 8192 rows x 1024 px x 128 B = 1 GiB

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, size_t n) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)n, BUF_FLAGS); }

// (a) the guide's calibrated case: 16 B per lane, consecutive lanes consecutive addresses
__global__ __launch_bounds__(256) void calib_read_b128_stream(const uint8_t* src, uint32_t* sink, size_t bytes)
{
    const __amdgpu_buffer_rsrc_t rs = rsrc(src, bytes);
    uint32_t acc = 0;
    for (size_t o = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16; o < bytes; o += (size_t)gridDim.x * 256 * 16) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(uint32_t)o, 0, 0);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
// (b) 8 B per lane, consecutive lanes consecutive addresses (512 bytes per wave-load)
template <int AUX>
__device__ __forceinline__ void read_b64_stream(const uint8_t* src, uint32_t* sink, size_t bytes)
{
    const __amdgpu_buffer_rsrc_t rs = rsrc(src, bytes);
    uint32_t acc = 0;
    for (size_t o = ((size_t)blockIdx.x * 256 + threadIdx.x) * 8; o < bytes; o += (size_t)gridDim.x * 256 * 8) {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(uint32_t)o, 0, AUX);
        acc ^= v.x ^ v.y;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void calib_read_b64_stream(const uint8_t* src, uint32_t* sink, size_t bytes) { read_b64_stream<0>(src, sink, bytes); }
__global__ __launch_bounds__(256) void calib_read_b64_stream_nt(const uint8_t* src, uint32_t* sink, size_t bytes) { read_b64_stream<2>(src, sink, bytes); }
// (c) the band kernel's pattern: a wave owns 4 rows (lane groups of 16 lanes x 8 B = one 128-byte pixel each, 128 KB apart) and
//     sweeps them pixel by pixel with a skew of one step per row, 8 loads in flight
template <int AUX>
__device__ __forceinline__ void read_b64_band(const uint8_t* src, uint32_t* sink, size_t bytes)
{
    const __amdgpu_buffer_rsrc_t rs = rsrc(src, bytes);
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6, j = lane >> 4, gl = lane & 15;
    const int row = wave * 4 + j;
    uint32_t acc = 0;
    if (row < ROWS)
        for (int T = 0; T < W + 4; T++) {
            const int x = T - j;
            const uint32_t off = (x >= 0 && x < W) ? (uint32_t)(((size_t)row * W + x) * 128 + gl * 8) : 0xffffffffu;   // out of range: returns 0, moves nothing
            const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, 0, AUX);
            acc ^= v.x ^ v.y;
        }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void calib_read_b64_band(const uint8_t* src, uint32_t* sink, size_t bytes) { read_b64_band<0>(src, sink, bytes); }
// (d) the stores: 8 B per lane non-temporal in the band pattern (k_mgm_bands' e-stores), and 16 B per lane plain
__global__ __launch_bounds__(256) void calib_write_b64_band_nt(uint8_t* dst, size_t bytes)
{
    const __amdgpu_buffer_rsrc_t rs = rsrc(dst, bytes);
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6, j = lane >> 4, gl = lane & 15;
    const int row = wave * 4 + j;
    if (row < ROWS)
        for (int T = 0; T < W + 4; T++) {
            const int x = T - j;
            const uint32_t off = (x >= 0 && x < W) ? (uint32_t)(((size_t)row * W + x) * 128 + gl * 8) : 0xffffffffu;
            u32x2 v; v.x = (uint32_t)T; v.y = (uint32_t)row;
            __builtin_amdgcn_raw_buffer_store_b64(v, rs, (int)off, 0, 2);
        }
}
__global__ __launch_bounds__(256) void calib_write_b128_stream(uint8_t* dst, size_t bytes)
{
    const __amdgpu_buffer_rsrc_t rs = rsrc(dst, bytes);
    for (size_t o = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16; o < bytes; o += (size_t)gridDim.x * 256 * 16) {
        u32x4 v; v.x = (uint32_t)o; v.y = 1; v.z = 2; v.w = 3;
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)(uint32_t)o, 0, 0);
    }
}
// (f) round 6: the band kernel's own MIX -- per wave-step one 8 B-per-lane load of a 128-byte pixel and one non-temporal 8 B-per-lane store
//     of a 128-byte pixel, 4 skewed rows per wave, reads from the first half of the buffer and writes to the second (1/2 GiB each way =
//     1 GiB moved) -- and a plain streaming copy of the same bytes: what the memory system gives this read/write mix when nothing
//     computes and nothing waits for a neighbour.  (FETCH_SIZE / WRITE_SIZE over known: 0.25 / 0.5 -- half of the bytes go each way.)
__global__ __launch_bounds__(256) void calib_rw_b64_band(uint8_t* buf, uint32_t* sink, size_t bytes)
{
    const __amdgpu_buffer_rsrc_t rs = rsrc(buf, bytes);
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6, j = lane >> 4, gl = lane & 15;
    const int row = wave * 4 + j;                       // ROWS rows of W / 2 pixels in each half of the buffer: 2 048 waves, as many as the band kernel's
    uint32_t acc = 0;
    const size_t half = bytes / 2;
    if (row < ROWS)
        for (int T = 0; T < W / 2 + 4; T++) {
            const int x = T - j;
            const bool in = x >= 0 && x < W / 2;
            const uint32_t off = in ? (uint32_t)(((size_t)row * (W / 2) + x) * 128 + gl * 8) : 0xffffffffu;
            const uint32_t woff = in ? (uint32_t)(half + ((size_t)row * (W / 2) + x) * 128 + gl * 8) : 0xffffffffu;
            const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, 0, 0);
            u32x2 o; o.x = (uint32_t)T; o.y = (uint32_t)row;
            __builtin_amdgcn_raw_buffer_store_b64(o, rs, (int)woff, 0, 2);
            acc ^= v.x ^ v.y;
        }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void calib_copy_b128_stream(uint8_t* buf, size_t bytes)
{
    const __amdgpu_buffer_rsrc_t rs = rsrc(buf, bytes);
    const size_t half = bytes / 2;
    for (size_t o = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16; o < half; o += (size_t)gridDim.x * 256 * 16) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(uint32_t)o, 0, 2);
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)(uint32_t)(o + half), 0, 2);
    }
}
// (e) the hand-off rows: 16 B per lane write-through stores / L2-bypassing loads (sc0 sc1)
__global__ __launch_bounds__(256) void calib_write_b128_sc0sc1(uint8_t* dst, size_t bytes)
{
    const __amdgpu_buffer_rsrc_t rs = rsrc(dst, bytes);
    for (size_t o = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16; o < bytes; o += (size_t)gridDim.x * 256 * 16) {
        u32x4 v; v.x = (uint32_t)o; v.y = 1; v.z = 2; v.w = 3;
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)(uint32_t)o, 0, 17);
    }
}
__global__ __launch_bounds__(256) void calib_read_b128_sc0sc1(const uint8_t* src, uint32_t* sink, size_t bytes)
{
    const __amdgpu_buffer_rsrc_t rs = rsrc(src, bytes);
    uint32_t acc = 0;
    for (size_t o = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16; o < bytes; o += (size_t)gridDim.x * 256 * 16) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(uint32_t)o, 0, 17);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main()
{
    const size_t bytes = (size_t)ROWS * W * 128;
    uint8_t* a; uint32_t* sink;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&sink, 256) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipMemset(a, 1, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, auto&& f) {
        f();                                                    // (the PMC passes see both launches: the summary averages them)
        hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %8.3f ms  %6.2f TB/s  known_bytes %zu\n", name, ms, bytes / 1e9 / ms, bytes);
    };
    const int sblocks = 4096, bblocks = ROWS / 16;              // band pattern: 4 waves x 4 rows per block
    timeit("calib_read_b128_stream", [&] { hipLaunchKernelGGL(calib_read_b128_stream, dim3(sblocks), dim3(256), 0, 0, a, sink, bytes); });
    timeit("calib_read_b64_stream", [&] { hipLaunchKernelGGL(calib_read_b64_stream, dim3(sblocks), dim3(256), 0, 0, a, sink, bytes); });
    timeit("calib_read_b64_stream_nt", [&] { hipLaunchKernelGGL(calib_read_b64_stream_nt, dim3(sblocks), dim3(256), 0, 0, a, sink, bytes); });
    timeit("calib_read_b64_band", [&] { hipLaunchKernelGGL(calib_read_b64_band, dim3(bblocks), dim3(256), 0, 0, a, sink, bytes); });
    timeit("calib_read_b128_sc0sc1", [&] { hipLaunchKernelGGL(calib_read_b128_sc0sc1, dim3(sblocks), dim3(256), 0, 0, a, sink, bytes); });
    timeit("calib_write_b64_band_nt", [&] { hipLaunchKernelGGL(calib_write_b64_band_nt, dim3(bblocks), dim3(256), 0, 0, a, bytes); });
    timeit("calib_write_b128_stream", [&] { hipLaunchKernelGGL(calib_write_b128_stream, dim3(sblocks), dim3(256), 0, 0, a, bytes); });
    timeit("calib_write_b128_sc0sc1", [&] { hipLaunchKernelGGL(calib_write_b128_sc0sc1, dim3(sblocks), dim3(256), 0, 0, a, bytes); });
    timeit("calib_rw_b64_band", [&] { hipLaunchKernelGGL(calib_rw_b64_band, dim3(bblocks), dim3(256), 0, 0, a, sink, bytes); });
    timeit("calib_copy_b128_stream", [&] { hipLaunchKernelGGL(calib_copy_b128_stream, dim3(sblocks), dim3(256), 0, 0, a, bytes); });
    hipDeviceSynchronize();
    return 0;
}
