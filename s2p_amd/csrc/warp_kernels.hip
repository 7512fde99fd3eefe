// s2p_amd/csrc/warp_kernels.hip -- homography resampler for gfx950: the MI355X stand-in for the
// `homography` binary behind s2p.common.image_apply_homography (s2p/common.py:159-180), called twice
// per tile by rectification.rectify_pair (s2p/rectification.py:379-380).
// Algorithm statement and parity status: oracle/resample_oracle.c (quintic B-spline, Unser/Thevenaz
// recursive prefilter + 6x6 tensor-product taps, mirror boundary, NaN outside the source domain).
// Same float32 operation order as the oracle (no FMA contraction: -ffp-contract=off).
//
// Kernels: convert(+NaN mask) -> transpose -> IIR prefilter along lines (one thread per line, lines
// contiguous across threads => coalesced) -> transpose back -> IIR prefilter along columns -> gather.
#include "common.hpp"

#include <algorithm>

namespace s2p {

#define BS_Z1 (-0.43057534709997430f)
#define BS_Z2 (-0.04309628820326465f)
#define BS_HORIZON 40

template <typename T>
__global__ __launch_bounds__(256) void k_warp_convert(const T* __restrict__ src, size_t n, float* __restrict__ coef, uint8_t* __restrict__ bad)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v = (float)src[i];
    bool f = isfinite(v);
    bad[i] = f ? 0 : 1;
    coef[i] = f ? v : 0.0f;
}

// out[x][y] = in[y][x]   (in: rows x cols)
__global__ __launch_bounds__(256) void k_transpose(const float* __restrict__ in, int rows, int cols, float* __restrict__ out)
{
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
    const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
    #pragma unroll
    for (int j = 0; j < 32; j += 8) {
        int x = x0 + tx, y = y0 + ty + j;
        if (x < cols && y < rows) tile[ty + j][tx] = in[(size_t)y * cols + x];
    }
    __syncthreads();
    #pragma unroll
    for (int j = 0; j < 32; j += 8) {
        int y = y0 + tx, x = x0 + ty + j;
        if (x < cols && y < rows) out[(size_t)x * rows + y] = tile[tx][ty + j];
    }
}

__device__ __forceinline__ void prefilter_pole_dev(float* c, int n, size_t s, float z)
{
    if (n == 1) return;
    float zk = z, sum = c[0];
    const int hor = BS_HORIZON < n ? BS_HORIZON : n;
    for (int k = 1; k < hor; k++) { sum = sum + zk * c[(size_t)k * s]; zk = zk * z; }
    c[0] = sum;
    float prev = sum;
    for (int k = 1; k < n; k++) { prev = c[(size_t)k * s] + z * prev; c[(size_t)k * s] = prev; }
    // prev = c+[n-1]
    float last = (z / (z * z - 1.0f)) * (z * c[(size_t)(n - 2) * s] + prev);
    c[(size_t)(n - 1) * s] = last;
    float next = last;
    for (int k = n - 2; k >= 0; k--) { next = z * (next - c[(size_t)k * s]); c[(size_t)k * s] = next; }
}

// img: len samples per line, nlines lines; sample k of line l at img[k * nlines + l]
__global__ __launch_bounds__(64) void k_prefilter_lines(float* img, int nlines, int len)
{
    const int l = blockIdx.x * 64 + threadIdx.x;
    if (l >= nlines) return;
    float* c = img + l;
    const size_t s = (size_t)nlines;
    const float lambda = (1.0f - BS_Z1) * (1.0f - 1.0f / BS_Z1) * ((1.0f - BS_Z2) * (1.0f - 1.0f / BS_Z2));
    if (len > 1) for (int k = 0; k < len; k++) c[(size_t)k * s] = c[(size_t)k * s] * lambda;
    prefilter_pole_dev(c, len, s, BS_Z1);
    prefilter_pole_dev(c, len, s, BS_Z2);
}

__device__ __forceinline__ void bspline5_weights_dev(float w, float* o)
{
    float w2 = w * w;
    o[5] = (1.0f / 120.0f) * w * w2 * w2;
    w2 = w2 - w;
    float w4 = w2 * w2;
    w = w - 0.5f;
    float t = w2 * (w2 - 3.0f);
    o[0] = (1.0f / 24.0f) * (1.0f / 5.0f + w2 + w4) - o[5];
    float t0 = (1.0f / 24.0f) * (w2 * (w2 - 5.0f) + 46.0f / 5.0f);
    float t1 = (-1.0f / 12.0f) * w * (t + 4.0f);
    o[2] = t0 + t1;
    o[3] = t0 - t1;
    t0 = (1.0f / 16.0f) * (9.0f / 5.0f - t);
    t1 = (1.0f / 24.0f) * w * (w4 - w2 - 5.0f);
    o[1] = t0 + t1;
    o[4] = t0 - t1;
}
__device__ __forceinline__ int mirror_dev(int i, int n)
{
    if (n == 1) return 0;
    const int p = 2 * n - 2;
    i = i < 0 ? -i : i;
    i = i % p;
    return i >= n ? p - i : i;
}

struct WarpArgs { double Hi[9]; const float* coef; const uint8_t* bad; int sw, sh, w, h; float* dst; };

__global__ __launch_bounds__(256) void k_warp_sample(WarpArgs a)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= a.w) return;
    const double X = a.Hi[0] * x + a.Hi[1] * y + a.Hi[2], Y = a.Hi[3] * x + a.Hi[4] * y + a.Hi[5], Z = a.Hi[6] * x + a.Hi[7] * y + a.Hi[8];
    const double u = X / Z, v = Y / Z;
    float out = __builtin_nanf("");
    if (u >= -0.5 && u <= a.sw - 0.5 && v >= -0.5 && v <= a.sh - 0.5) {
        const double fu = floor(u), fv = floor(v);
        const int iu = (int)fu, iv = (int)fv;
        float wx[6], wy[6];
        bspline5_weights_dev((float)(u - fu), wx);
        bspline5_weights_dev((float)(v - fv), wy);
        int xi[6];
        #pragma unroll
        for (int i = 0; i < 6; i++) xi[i] = mirror_dev(iu - 2 + i, a.sw);
        float acc = 0.0f;
        int anybad = 0;
        #pragma unroll
        for (int j = 0; j < 6; j++) {
            const size_t ro = (size_t)mirror_dev(iv - 2 + j, a.sh) * a.sw;
            float row = 0.0f;
            #pragma unroll
            for (int i = 0; i < 6; i++) {
                anybad |= a.bad[ro + xi[i]];
                row = row + wx[i] * a.coef[ro + xi[i]];
            }
            acc = acc + wy[j] * row;
        }
        if (!anybad) out = acc;
    }
    a.dst[(size_t)y * a.w + x] = out;
}

static bool invert3x3(const double* H, double* I)
{
    double a = H[0], b = H[1], c = H[2], d = H[3], e = H[4], f = H[5], g = H[6], hh = H[7], i = H[8];
    double det = a * (e * i - f * hh) - b * (d * i - f * g) + c * (d * hh - e * g);
    if (det == 0.0) return false;
    double s = 1.0 / det;
    I[0] = (e * i - f * hh) * s; I[1] = (c * hh - b * i) * s; I[2] = (b * f - c * e) * s;
    I[3] = (f * g - d * i) * s;  I[4] = (a * i - c * g) * s;  I[5] = (c * d - a * f) * s;
    I[6] = (d * hh - e * g) * s; I[7] = (b * g - a * hh) * s; I[8] = (a * e - b * d) * s;
    return true;
}

size_t warp_workspace_bytes(int sw, int sh)
{
    const size_t n = (size_t)sw * sh;
    return align_up(n * 4, 256) * 2 + align_up(n, 256) + 4096;
}

// d_src: device pointer to the source raster (dtype: 0 = float32, 1 = uint16, 2 = uint8); d_dst: w*h float32.
// Uses the context workspace from `ws_offset` on (so that callers can keep their own I/O there).
int warp_enqueue(s2p_hip_ctx* ctx, const void* d_src, int dtype, int sw, int sh, const double H[9],
                 float* d_dst, int w, int h, char* scratch)
{
    hipStream_t st = ctx->stream;
    WarpArgs a;
    if (!invert3x3(H, a.Hi)) { set_last_error("warp: singular homography"); return S2P_HIP_BAD_ARGUMENT; }
    const size_t n = (size_t)sw * sh;
    float* coef = (float*)scratch;
    float* tmp = (float*)(scratch + align_up(n * 4, 256));
    uint8_t* bad = (uint8_t*)(scratch + 2 * align_up(n * 4, 256));
    StageScope total(ctx, "warp");
    const unsigned nb = (unsigned)((n + 255) / 256);
    if (dtype == 0) hipLaunchKernelGGL(k_warp_convert<float>, dim3(nb), dim3(256), 0, st, (const float*)d_src, n, coef, bad);
    else if (dtype == 1) hipLaunchKernelGGL(k_warp_convert<uint16_t>, dim3(nb), dim3(256), 0, st, (const uint16_t*)d_src, n, coef, bad);
    else if (dtype == 2) hipLaunchKernelGGL(k_warp_convert<uint8_t>, dim3(nb), dim3(256), 0, st, (const uint8_t*)d_src, n, coef, bad);
    else { set_last_error("warp: unknown source dtype %d", dtype); return S2P_HIP_BAD_ARGUMENT; }
    // rows: transpose so that the sh image rows become contiguous-across-threads lines of length sw
    hipLaunchKernelGGL(k_transpose, dim3((sw + 31) / 32, (sh + 31) / 32), dim3(256), 0, st, coef, sh, sw, tmp);
    hipLaunchKernelGGL(k_prefilter_lines, dim3((sh + 63) / 64), dim3(64), 0, st, tmp, sh, sw);
    hipLaunchKernelGGL(k_transpose, dim3((sh + 31) / 32, (sw + 31) / 32), dim3(256), 0, st, tmp, sw, sh, coef);
    // columns
    hipLaunchKernelGGL(k_prefilter_lines, dim3((sw + 63) / 64), dim3(64), 0, st, coef, sw, sh);
    a.coef = coef; a.bad = bad; a.sw = sw; a.sh = sh; a.w = w; a.h = h; a.dst = d_dst;
    hipLaunchKernelGGL(k_warp_sample, dim3((w + 255) / 256, h), dim3(256), 0, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_last_error("kernel launch failed: %s", hipGetErrorString(e)); return S2P_HIP_RUNTIME_ERROR; }
    return S2P_HIP_OK;
}

}  // namespace s2p
