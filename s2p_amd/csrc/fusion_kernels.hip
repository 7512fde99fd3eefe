// s2p_amd/csrc/fusion_kernels.hip -- pixelwise merge of n co-registered height maps for gfx950: the MI355X
// stand-in for fusion.merge_n (s2p/fusion.py:26-68), the tri-stereo tail of the path.
// The reference stacks the n images minus their offsets in a float64 (h, w, n) array and runs a Python
// function per pixel through np.apply_along_axis (:54-58); here one thread per pixel does the same float64
// arithmetic in registers: values x_i = (double)img_i - offset_i, then
//   average_if_close (s2p/fusion.py:16-23): nanmax - nanmin > threshold ? NaN : nanmedian
//   np.nanmedian / np.median / np.nanmean / np.mean / np.nanmin / np.nanmax / np.min / np.max
// with numpy's own evaluation order (pairwise_sum: below 8 terms a plain left-to-right sum starting from 0.,
// up to 128 terms eight interleaved accumulators; median of an even count = pairwise sum of the two middle
// values / 2), then + mean(offsets) and the cast to float32 (:61-68).  Statement: oracle/pyoracle.py
// oracle_merge_n (numpy itself) and tests/golden/fusion_stack.npz (the reference's own average_if_close).
#include "common.hpp"
#include "ccl.hpp"

namespace s2p {

#define MERGE_MAX_N 64

struct MergeArgs {
    const float* stack;      // n planes of h*w float32
    const double* offsets;   // n
    int n; size_t npx;
    int op; double threshold, mean_offset;
    float* out;
};

// numpy's pairwise_sum for n <= 128 (numpy/core/src/umath/loops_utils.h.src), n <= MERGE_MAX_N here
__device__ __forceinline__ double np_sum(const double* a, int n)
{
    if (n < 8) {
        double r = 0.0;
        for (int i = 0; i < n; i++) r += a[i];
        return r;
    }
    double r[8];
    #pragma unroll
    for (int j = 0; j < 8; j++) r[j] = a[j];
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
        #pragma unroll
        for (int j = 0; j < 8; j++) r[j] += a[i + j];
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
}

__device__ __forceinline__ double median_sorted(const double* s, int m)
{
    if (m == 0) return __builtin_nan("");
    if (m & 1) return 0.0 + s[m / 2];                        // np.mean of one value (-0.0 becomes +0.0)
    return ((0.0 + s[m / 2 - 1]) + s[m / 2]) / 2.0;        // np.mean of the two middle values
}

__global__ __launch_bounds__(256) void k_merge_n(MergeArgs a)
{
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= a.npx) return;
    double x[MERGE_MAX_N], s[MERGE_MAX_N];
    const int n = a.n;
    int m = 0;                               // finite-or-inf (non-NaN) count, s[0..m) kept sorted
    bool any_nan = false;
    for (int i = 0; i < n; i++) {
        const double v = (double)a.stack[(size_t)i * a.npx + p] - a.offsets[i];
        x[i] = v;
        if (v != v) { any_nan = true; continue; }
        int j = m++;
        while (j > 0 && s[j - 1] > v) { s[j] = s[j - 1]; j--; }
        s[j] = v;
    }
    const double nan = __builtin_nan("");
    double r;
    switch (a.op) {
    case 0:                                                  // average_if_close
        r = (m > 0 && s[m - 1] - s[0] > a.threshold) ? nan : median_sorted(s, m);
        break;
    case 1: r = median_sorted(s, m); break;                  // np.nanmedian
    case 2: r = any_nan ? nan : median_sorted(s, m); break;  // np.median
    case 3: {                                                // np.nanmean: NaN -> 0, sum / count
        for (int i = 0; i < n; i++) s[i] = x[i] != x[i] ? 0.0 : x[i];
        r = m > 0 ? np_sum(s, n) / (double)m : nan;
        break;
    }
    case 4: r = np_sum(x, n) / (double)n; break;             // np.mean
    case 5: r = m > 0 ? s[0] : nan; break;                   // np.nanmin
    case 6: r = m > 0 ? s[m - 1] : nan; break;               // np.nanmax
    case 7: r = any_nan || m == 0 ? nan : s[0]; break;       // np.min
    default: r = any_nan || m == 0 ? nan : s[m - 1]; break;  // np.max
    }
    a.out[p] = (float)(r + a.mean_offset);
}

int merge_enqueue(s2p_hip_ctx* ctx, const float* d_stack, const double* d_offsets, int n, size_t npx, int op,
                  double threshold, double mean_offset, float* d_out)
{
    MergeArgs a;
    a.stack = d_stack; a.offsets = d_offsets; a.n = n; a.npx = npx; a.op = op; a.threshold = threshold; a.mean_offset = mean_offset; a.out = d_out;
    hipLaunchKernelGGL(k_merge_n, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, ctx->stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_last_error("kernel launch failed: %s", hipGetErrorString(e)); return S2P_HIP_RUNTIME_ERROR; }
    return S2P_HIP_OK;
}


// ---- common.cargarse_basura (s2p/common.py:224-235): the outlier filter of heights_fusion ------------------------
// Reference: six subprocesses (4 x morphoop, plambda, remove_small_cc) and three temporary TIFFs per height map.  Here
// two stages on the resident map: (1) k_range5: 5 x 5 NaN-skipping local max - min (c/morphoop.c:139-181: square
// element centred at sz / 2, whole-sample symmetric boundary), NaN where it exceeds 5 -- the tile rows staged through
// LDS with their 2-pixel apron, separable (row extrema first); (2) the run-based union-find CCL of ccl.hpp with the
// float policy (4-connected, |difference| < 5, components of fewer than 200 pixels -> NaN).  Statement:
// oracle/cleanup_oracle.c (remove_small_cc's source is absent: unpinned, see there).
__device__ __forceinline__ int sym_idx(int n, int x) { if (x < 0) x = -x - 1; if (x >= n) x = -x + 2 * n - 1; return x; }

#define R5_TX 64
#define R5_TY 16
static __global__ __launch_bounds__(256) void k_range5(const float* __restrict__ in, int w, int h, float thr, float* __restrict__ out)
{
    __shared__ float lo[R5_TY + 4][R5_TX], hi[R5_TY + 4][R5_TX];
    const int x0 = blockIdx.x * R5_TX, y0 = blockIdx.y * R5_TY;
    // horizontal pass: rows y0 - 2 .. y0 + TY + 1, the 5 columns around every x of the tile
    for (int i = threadIdx.x; i < (R5_TY + 4) * R5_TX; i += 256) {
        const int ry = i / R5_TX, rx = i % R5_TX;
        const int y = sym_idx(h, min(y0 + ry - 2, h + 1)), x = x0 + rx;
        float mn = __builtin_inff(), mx = -__builtin_inff();
        if (x < w) {
            const float* row = in + (size_t)y * w;
            #pragma unroll
            for (int d = -2; d <= 2; d++) {
                const float v = row[sym_idx(w, x + d)];
                if (v == v) { mn = fminf(mn, v); mx = fmaxf(mx, v); }
            }
        }
        lo[ry][rx] = mn; hi[ry][rx] = mx;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < R5_TY * R5_TX; i += 256) {
        const int ty = i / R5_TX, tx = i % R5_TX;
        const int x = x0 + tx, y = y0 + ty;
        if (x >= w || y >= h) continue;
        float mn = __builtin_inff(), mx = -__builtin_inff();
        #pragma unroll
        for (int d = 0; d < 5; d++) { mn = fminf(mn, lo[ty + d][tx]); mx = fmaxf(mx, hi[ty + d][tx]); }
        float z = in[(size_t)y * w + x];
        if (mx >= mn && fabsf(mx - mn) > thr) z = __builtin_nanf("");     // mx < mn: no sample in the window (all NaN)
        out[(size_t)y * w + x] = z;
    }
}

// d_out: w*h float32 (filtered in place after the range stage); lab/par/cnt: w*h int32 each
int cargarse_basura_enqueue(s2p_hip_ctx* ctx, const float* d_in, int w, int h, float* d_out, int* lab, int* par, int* cnt)
{
    if (h < 3 || w < 3) {   // sym_idx reflects once: windows wider than the image would need repeated reflection
        set_last_error("cargarse_basura: maps smaller than 3 x 3 are not supported"); return S2P_HIP_UNSUPPORTED;
    }
    hipLaunchKernelGGL(k_range5, dim3((w + R5_TX - 1) / R5_TX, (h + R5_TY - 1) / R5_TY), dim3(256), 0, ctx->stream, d_in, w, h, 5.0f, d_out);
    SmallCcF32 pol; pol.thr = 5.0f;
    enqueue_small_cc<SmallCcF32>(ctx->stream, d_out, w, h, pol, 200 - 1, lab, par, cnt);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_last_error("kernel launch failed: %s", hipGetErrorString(e)); return S2P_HIP_RUNTIME_ERROR; }
    return S2P_HIP_OK;
}

}  // namespace s2p
