// s2p_amd/csrc/ccl.hpp -- connected-component speckle filter shared by both matchers.
#pragma once
#include "common.hpp"

namespace s2p {

// =============================================================================================
// K6: speckle filter (stereosgbm.cpp:872-967: serial flood fill) as parallel connected-component
// labelling.  Components of the 4-neighbour graph whose edges join valid pixels differing by
// <= maxDiff are well defined (symmetric relation, evaluated on the unmodified image), so any
// exact CCL reproduces the flood fill.  Run-based union-find:
//   rows   : every pixel learns the start of its horizontal run (segmented scan, one block per row)
//   vmerge : runs of adjacent rows that touch through a vertical edge are united (atomicMin hooks,
//            path halving); one union per distinct (run above, run below) contact
//   count  : the last pixel of every run adds the run length to its root (saturating: once a root is
//            known to exceed maxSize nobody adds any more -> no hot-address atomics)
//   apply  : pixels whose root stayed <= maxSize become INVALID
// =============================================================================================
__device__ __forceinline__ int uf_find(int* par, int i) {
    int p = par[i];
    while (p != i) {
        int gp = par[p];
        if (gp != p) par[i] = gp;     // path halving; racing writers only ever store ancestors
        i = p; p = gp;
    }
    return i;
}
__device__ __forceinline__ void uf_union(int* par, int a, int b) {
    for (;;) {
        a = uf_find(par, a); b = uf_find(par, b);
        if (a == b) return;
        if (a < b) { int t = a; a = b; b = t; }       // a > b: hook the larger root under the smaller
        int old = atomicMin(&par[a], b);
        if (old == a) return;
        a = old;
    }
}
// Pixel policies: what counts as a pixel, when two neighbours are joined, and the value a removed pixel takes.
struct SpeckleS16 {                      // filterSpeckles on a x16 disparity canvas (stereosgbm.cpp:872-967)
    typedef int16_t T;
    int newVal, maxDiff;
    __device__ __forceinline__ bool valid(T v) const { return (int)v != newVal; }
    __device__ __forceinline__ bool edge(T a, T b) const { return (int)a != newVal && (int)b != newVal && abs((int)a - (int)b) <= maxDiff; }
    __device__ __forceinline__ T removed() const { return (T)newVal; }
};
struct SmallCcF32 {                      // remove_small_cc on a float map (common.cargarse_basura, s2p/common.py:234): NaN = no pixel
    typedef float T;
    float thr;
    __device__ __forceinline__ bool valid(T v) const { return v == v; }
    __device__ __forceinline__ bool edge(T a, T b) const { return a == a && b == b && fabsf(a - b) < thr; }
    __device__ __forceinline__ T removed() const { return __builtin_nanf(""); }
};

// runstart[i] = linear index of the first pixel of i's horizontal run (-1 for INVALID pixels);
// par[i] = i at run starts; cnt[i] = 0
template <typename P>
static __global__ __launch_bounds__(256) void k_ccl_rows(const typename P::T* __restrict__ img, int w, P pol,
                                                  int* __restrict__ runstart, int* __restrict__ par, int* __restrict__ cnt)
{
    __shared__ int carry[256];
    const int y = blockIdx.x, t = threadIdx.x;
    const int chunk = (w + 255) / 256;
    const int xa = t * chunk, xb = min(xa + chunk, w);
    const typename P::T* row = img + (size_t)y * w;
    // outgoing run start of this chunk: >= 0 when defined inside the chunk, -1 = "inherits", -2 = "no open run"
    int open = -1;
    for (int x = xa; x < xb; x++) {
        const typename P::T v = row[x];
        if (!pol.valid(v)) open = -2;
        else if (x == 0 || !pol.edge(row[x - 1], v)) open = x;
        // else: continues the run of x-1 (open unchanged)
    }
    carry[t] = (xa < xb) ? open : -1;
    __syncthreads();
    if (t == 0) {   // serial exclusive scan over 256 chunk summaries
        int cur = -2;
        for (int i = 0; i < 256; i++) { int o = carry[i]; carry[i] = cur; if (o != -1) cur = o; }
    }
    __syncthreads();
    open = carry[t];
    for (int x = xa; x < xb; x++) {
        const typename P::T v = row[x];
        size_t i = (size_t)y * w + x;
        if (!pol.valid(v)) { open = -2; runstart[i] = -1; par[i] = -1; }
        else {
            if (x == 0 || !pol.edge(row[x - 1], v)) open = x;
            runstart[i] = y * w + open;
            par[i] = (int)i;          // only entries at run starts are ever used as union-find nodes
        }
        cnt[i] = 0;
    }
}

template <typename P>
static __global__ __launch_bounds__(256) void k_ccl_vmerge(const typename P::T* __restrict__ img, int w, int h, P pol,
                                                    const int* __restrict__ runstart, int* par)
{
    int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w || y + 1 >= h) return;
    int i = y * w + x;
    if (!pol.edge(img[i], img[i + w])) return;
    int ra = runstart[i], rb = runstart[i + w];
    // skip when the pixel to the left made the very same contact
    if (x > 0 && runstart[i - 1] == ra && runstart[i + w - 1] == rb && pol.edge(img[i - 1], img[i + w - 1])) return;
    uf_union(par, ra, rb);
}

static __global__ __launch_bounds__(256) void k_ccl_count(int w, int h, int maxSize, const int* __restrict__ runstart, int* par, int* cnt)
{
    int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    int i = y * w + x;
    int rs = runstart[i];
    if (rs < 0) return;
    if (x + 1 < w && runstart[i + 1] == rs) return;      // not the last pixel of its run
    int len = i - rs + 1;
    int r = uf_find(par, rs);
    // saturating count: values above maxSize are all equivalent
    if (__hip_atomic_load(&cnt[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= maxSize)
        atomicAdd(&cnt[r], min(len, maxSize + 1));
}

template <typename P>
static __global__ __launch_bounds__(256) void k_ccl_apply(typename P::T* img, int n, P pol, int maxSize,
                                                   const int* __restrict__ runstart, int* par, const int* __restrict__ cnt)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int rs = runstart[i];
    if (rs < 0) return;
    int r = uf_find(par, rs);
    if (cnt[r] <= maxSize) img[i] = pol.removed();
}


// Enqueue the filter on an image (w x h) of P::T: components of pixels joined by P::edge; components of <= maxSize
// pixels are set to P::removed().
template <typename P>
static void enqueue_small_cc(hipStream_t st, typename P::T* img, int w, int h, P pol, int maxSize, int* runstart, int* par, int* cnt)
{
    const int n = w * h, nb = (n + 255) / 256;
    hipLaunchKernelGGL(k_ccl_rows<P>, dim3(h), dim3(256), 0, st, img, w, pol, runstart, par, cnt);
    hipLaunchKernelGGL(k_ccl_vmerge<P>, dim3((w + 255) / 256, h), dim3(256), 0, st, img, w, h, pol, runstart, par);
    hipLaunchKernelGGL(k_ccl_count, dim3((w + 255) / 256, h), dim3(256), 0, st, w, h, maxSize, runstart, par, cnt);
    hipLaunchKernelGGL(k_ccl_apply<P>, dim3(nb), dim3(256), 0, st, img, n, pol, maxSize, runstart, par, cnt);
}
// int16 image: components of pixels != newVal joined by |difference| <= maxDiff; components of <= maxSize pixels -> newVal
static inline void enqueue_speckle(hipStream_t st, int16_t* img, int w, int h, int newVal, int maxSize, int maxDiff,
                                   int* runstart, int* par, int* cnt)
{
    SpeckleS16 pol; pol.newVal = newVal; pol.maxDiff = maxDiff;
    enqueue_small_cc<SpeckleS16>(st, img, w, h, pol, maxSize, runstart, par, cnt);
}

}  // namespace s2p
