// s2p_amd/csrc/raster_kernels.hip -- DSM rasterisation for gfx950: the MI355X stand-in for `rasterize_cloud` of the
// plyflatten package (a pip dependency of the reference, not vendored: s2p/__init__.py:31, :462-466;
// tests/rasterization_test.py), SURVEY.md 8(f) rank 4.  Statement: oracle/rasterize_oracle.c, pinned bit for bit on
// the reference's own golden (input_ply/cloud.ply -> expected_output/plyflatten/dsm_40cm.tiff) for radius 0.
//
// The CPU code walks the points in input order and keeps, per cell, a running weighted mean in float32
//   avg = (v * weight + cnt * avg) / (weight + cnt);  cnt += weight
// which is order dependent: to return the same bits, every cell must see its points in input order.  So the GPU
// does not scatter with floating-point atomics; it builds the cell -> points lists (counting sort on the cell index:
// integer atomics only), puts each list back into input order, and lets one thread per cell run the recurrence:
//   k_raster_scatter<false>  per point: the cells of its disc that lie in the raster, atomicAdd 1 on each   (4 B / contribution)
//   scan             exclusive prefix sum of the per-cell counts (three small kernels)
//   k_raster_scatter<true>   per point again: slot = start[cell] + atomicAdd(fill[cell], 1); list[slot] = point index
//   k_raster_cells   per cell: its list sorted by point index (insertion sort for the usual handful of points, in-place
//                    heap sort for a crowded cell), then the recurrence over the bands, weights recomputed from the
//                    point and the cell centre; cells without points -> NaN
// Bytes: points are read 2 + (list length) times (24 + 8 nb B each), cells written once: bound by the gather of the
// last kernel (random 8 (2 + nb)-byte reads) -- microseconds for a tile's cloud, the launch overheads dominate.
#include "common.hpp"

#include <algorithm>

namespace s2p {

#define RASTER_MAX_BANDS 16

struct RasterArgs {
    const double* pts; int npts, nb;
    double xoff, yoff, res;
    int xsize, ysize, radius; float sigma;
    int* cnt; int* start; int* fill; uint32_t* list;
    float* raster;
};

// cell of a point; a point with a non-finite or far-away coordinate has none (the C code's int conversion of such
// a value is undefined: the oracle and this kernel both skip the point)
__device__ __forceinline__ bool point_cell(const RasterArgs& a, int k, int& i, int& j)
{
    const double* p = a.pts + (size_t)k * (2 + a.nb);
    const double fi = floor((p[0] - a.xoff) / a.res), fj = floor((-p[1] - (-a.yoff)) / a.res);
    if (!(fabs(fi) < 1e9) || !(fabs(fj) < 1e9)) return false;
    i = (int)fi; j = (int)fj;
    return true;
}

template <bool FILL>
__global__ __launch_bounds__(256) void k_raster_scatter(RasterArgs a)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= a.npts) return;
    int i, j;
    if (!point_cell(a, k, i, j)) return;
    const int r = a.radius;
    for (int k1 = -r; k1 <= r; k1++)
        for (int k2 = -r; k2 <= r; k2++) {
            if (k1 * k1 + k2 * k2 > r * r) continue;
            const int ii = i + k1, jj = j + k2;
            if (ii < 0 || jj < 0 || ii >= a.xsize || jj >= a.ysize) continue;
            const size_t c = (size_t)a.xsize * jj + ii;
            if (FILL) a.list[a.start[c] + atomicAdd(&a.fill[c], 1)] = (uint32_t)k;
            else atomicAdd(&a.cnt[c], 1);
        }
}

// ---- exclusive scan of n ints: 1024 per block ------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_scan_block(const int* __restrict__ in, int* __restrict__ out, int* __restrict__ bsum, size_t n)
{
    __shared__ int wsum[4];
    const size_t base = (size_t)blockIdx.x * 1024 + (size_t)threadIdx.x * 4;
    int v[4], t = 0;
    #pragma unroll
    for (int q = 0; q < 4; q++) { v[q] = base + q < n ? in[base + q] : 0; t += v[q]; }
    int incl = t;                                   // inclusive scan of the per-thread totals inside the wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    #pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int woff = 0;
    for (int q = 0; q < wave; q++) woff += wsum[q];
    int run = woff + incl - t;                       // exclusive prefix of this thread
    #pragma unroll
    for (int q = 0; q < 4; q++) { if (base + q < n) out[base + q] = run; run += v[q]; }
    if (threadIdx.x == 255) bsum[blockIdx.x] = woff + incl;
}
__global__ __launch_bounds__(256) void k_scan_sums(int* __restrict__ bsum, int nblocks)
{
    // one block walks the block sums sequentially in chunks of 256 (nblocks = cells / 1024: small)
    __shared__ int carry, tmp[256];
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nblocks; b0 += 256) {
        const int idx = b0 + threadIdx.x;
        tmp[threadIdx.x] = idx < nblocks ? bsum[idx] : 0;
        __syncthreads();
        if (threadIdx.x == 0) {
            int run = carry;
            for (int q = 0; q < 256; q++) { const int x = tmp[q]; tmp[q] = run; run += x; }
            carry = run;
        }
        __syncthreads();
        if (idx < nblocks) bsum[idx] = tmp[threadIdx.x];
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void k_scan_add(int* __restrict__ out, const int* __restrict__ bsum, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] += bsum[i >> 10];
}

// ---- one thread per cell: restore input order, run the reference's recurrence ------------------------------------
__global__ __launch_bounds__(256) void k_raster_cells(RasterArgs a)
{
    const size_t c = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t ncell = (size_t)a.xsize * a.ysize;
    if (c >= ncell) return;
    const int m = a.cnt[c], nb = a.nb;
    float* out = a.raster + c * nb;
    uint32_t* l = a.list + a.start[c];
    if (m <= 48) {
        for (int q = 1; q < m; q++) {               // insertion sort by point index: the usual case, a handful of points
            const uint32_t x = l[q];
            int p = q - 1;
            while (p >= 0 && l[p] > x) { l[p + 1] = l[p]; p--; }
            l[p + 1] = x;
        }
    } else {                                        // a crowded cell: in-place heap sort, O(m log m) whatever the input
        auto sift = [&](int root, int end) {
            for (;;) {
                int child = 2 * root + 1;
                if (child >= end) break;
                if (child + 1 < end && l[child] < l[child + 1]) child++;
                if (l[root] >= l[child]) break;
                const uint32_t t = l[root]; l[root] = l[child]; l[child] = t;
                root = child;
            }
        };
        for (int q = m / 2 - 1; q >= 0; q--) sift(q, m);
        for (int end = m - 1; end > 0; end--) {
            const uint32_t t = l[0]; l[0] = l[end]; l[end] = t;
            sift(0, end);
        }
    }
    float avg[RASTER_MAX_BANDS], cnt = 0.f;
    for (int b = 0; b < nb; b++) avg[b] = 0.f;
    const int ii = (int)(c % a.xsize), jj = (int)(c / a.xsize);
    const bool unweighted = isinf(a.sigma);
    for (int q = 0; q < m; q++) {
        const double* p = a.pts + (size_t)l[q] * (2 + nb);
        float weight = 1.0f;
        if (!unweighted) {
            const float dx = (float)(p[0] - (a.xoff + a.res * (0.5 + ii)));
            const float dy = (float)(p[1] - (a.yoff - a.res * (0.5 + jj)));
            const float d = sqrtf(dx * dx + dy * dy);
            weight = (float)exp((double)(-d * d / (2 * a.sigma * a.sigma)));
        }
        for (int b = 0; b < nb; b++) avg[b] = (float)((p[2 + b] * weight + cnt * avg[b]) / (weight + cnt));
        cnt += weight;
    }
    for (int b = 0; b < nb; b++) out[b] = cnt ? avg[b] : __builtin_nanf("");
}

size_t raster_disc_cells(int radius)
{
    size_t n = 0;
    for (int k1 = -radius; k1 <= radius; k1++) for (int k2 = -radius; k2 <= radius; k2++) if (k1 * k1 + k2 * k2 <= radius * radius) n++;
    return n;
}
size_t raster_workspace_bytes(int npts, int nb, int xsize, int ysize, int radius)
{
    const size_t ncell = (size_t)xsize * ysize, nblk = (ncell + 1023) / 1024;
    return align_up((size_t)npts * (2 + nb) * 8, 256) + 3 * align_up(ncell * 4, 256) + align_up(nblk * 4, 256) +
           align_up((size_t)npts * raster_disc_cells(radius) * 4, 256) + align_up(ncell * nb * 4, 256) + 4096;
}

// d_pts: npts x (2 + nb) doubles on the device; d_raster: ysize x xsize x nb float32.  The scratch comes from the ctx workspace.
int raster_enqueue(s2p_hip_ctx* ctx, const double* d_pts, int npts, int nb, double xoff, double yoff, double res,
                   int xsize, int ysize, int radius, float sigma, float* d_raster)
{
    hipStream_t st = ctx->stream;
    const size_t ncell = (size_t)xsize * ysize, nblk = (ncell + 1023) / 1024;
    RasterArgs a;
    a.pts = d_pts; a.npts = npts; a.nb = nb; a.xoff = xoff; a.yoff = yoff; a.res = res;
    a.xsize = xsize; a.ysize = ysize; a.radius = radius; a.sigma = sigma; a.raster = d_raster;
    a.cnt = (int*)ws_alloc(ctx, ncell * 4); a.start = (int*)ws_alloc(ctx, ncell * 4); a.fill = (int*)ws_alloc(ctx, ncell * 4);
    int* bsum = (int*)ws_alloc(ctx, nblk * 4);
    a.list = (uint32_t*)ws_alloc(ctx, std::max<size_t>((size_t)npts * raster_disc_cells(radius) * 4, 4));
    if (!a.cnt || !a.start || !a.fill || !bsum || !a.list) { set_last_error("plyflatten: workspace"); return S2P_HIP_RUNTIME_ERROR; }
    S2P_HIP_CHECK(hipMemsetAsync(a.cnt, 0, ncell * 4, st));
    S2P_HIP_CHECK(hipMemsetAsync(a.fill, 0, ncell * 4, st));
    const unsigned pb = (unsigned)((npts + 255) / 256);
    if (npts > 0) hipLaunchKernelGGL(k_raster_scatter<false>, dim3(pb), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_scan_block, dim3((unsigned)nblk), dim3(256), 0, st, a.cnt, a.start, bsum, ncell);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(256), 0, st, bsum, (int)nblk);
    hipLaunchKernelGGL(k_scan_add, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, st, a.start, bsum, ncell);
    if (npts > 0) hipLaunchKernelGGL(k_raster_scatter<true>, dim3(pb), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_raster_cells, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_last_error("kernel launch failed: %s", hipGetErrorString(e)); return S2P_HIP_RUNTIME_ERROR; }
    return S2P_HIP_OK;
}

}  // namespace s2p
