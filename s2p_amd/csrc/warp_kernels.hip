// s2p_amd/csrc/warp_kernels.hip -- homography resampler for gfx950: the MI355X stand-in for the
// `homography` binary behind s2p.common.image_apply_homography (s2p/common.py:159-180), called twice
// per tile by rectification.rectify_pair (s2p/rectification.py:379-380).
// Algorithm statement and parity status: oracle/resample_oracle.c (quintic B-spline, Unser/Thevenaz
// recursive prefilter + 6x6 tensor-product taps, mirror boundary, NaN outside the source domain).
// Same float32 operation order as the oracle (no FMA contraction, -ffp-contract=off; the recursions use
// the explicit fmaf the oracle prescribes).
//
// Kernels: IIR prefilter of the rows (+ conversion to float32 and NaN mask) -> IIR prefilter of the columns (+ NaN poison
// of the coefficients at non-finite source pixels) -> 6x6 gather.
// The recursive prefilter is a serial float32 chain per line (4 dependent sweeps; the oracle fixes the
// rounding order, so a scan is not an option).  k_prefilter_lds keeps a block of L lines resident in LDS
// (up to 160 KB) for all four sweeps: global memory is touched once per direction, fully coalesced, and
// the chain runs at LDS/VALU latency instead of one L2 round trip per sample.  Lines too long for LDS
// (> 40960 samples) take the strided global-memory kernel k_prefilter_lines.
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
    for (int k = 1; k < n; k++) { prev = __builtin_fmaf(z, prev, c[(size_t)k * s]); c[(size_t)k * s] = prev; }
    // prev = c+[n-1]
    float last = (z / (z * z - 1.0f)) * (z * c[(size_t)(n - 2) * s] + prev);
    c[(size_t)(n - 1) * s] = last;
    float next = last;
    for (int k = n - 2; k >= 0; k--) { const float t = z * c[(size_t)k * s]; next = __builtin_fmaf(z, next, -t); c[(size_t)k * s] = next; }
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


// ---- LDS-resident prefilter ---------------------------------------------------------------------
// One workgroup = L lines, line l at lds[l * S ..], S = 4 * odd >= len: a lane walks its own line with
// 128-bit LDS accesses (4 samples per instruction; the single chain wave is LDS-issue bound, not VALU
// bound), and S / 4 odd keeps the 16-lane b128 phases on distinct banks.
__device__ __forceinline__ void prefilter_pole_lds(float* c, int n, float z)
{
    constexpr int B = 16;     // samples per register block; the next block's LDS reads are in flight while
                              // this block's dependent chain runs (two blocks, rotated by hand)
#define S2P_LD(v, k0) _Pragma("unroll") for (int j = 0; j < B / 4; j++) { const float4 t = ((const float4*)(c + (k0)))[j]; v[4 * j] = t.x; v[4 * j + 1] = t.y; v[4 * j + 2] = t.z; v[4 * j + 3] = t.w; }
#define S2P_ST(v, k0) _Pragma("unroll") for (int j = 0; j < B / 4; j++) ((float4*)(c + (k0)))[j] = make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3])
#define S2P_FWD(v, k0) { _Pragma("unroll") for (int j = 0; j < B; j++) { prev = __builtin_fmaf(z, prev, v[j]); v[j] = prev; } S2P_ST(v, k0); }
#define S2P_BWD(v, k0) { _Pragma("unroll") for (int j = B - 1; j >= 0; j--) { next = __builtin_fmaf(z, next, -(z * v[j])); v[j] = next; } S2P_ST(v, k0); }
    float zk = z, sum = c[0];
    const int hor = BS_HORIZON < n ? BS_HORIZON : n;
    for (int k = 1; k < hor; k++) { sum = sum + zk * c[k]; zk = zk * z; }
    c[0] = sum;
    float prev = sum;
    float a[B], b[B];
    {
        int k = 1;
        for (; k < n && (k & 3); k++) { prev = __builtin_fmaf(z, prev, c[k]); c[k] = prev; }
        const int k1 = k, nb = (n - k1) / B;
        if (nb > 0) S2P_LD(a, k1);
        int blk = 0;
        for (; blk + 2 <= nb; blk += 2) {
            k = k1 + blk * B;
            S2P_LD(b, k + B);
            __builtin_amdgcn_sched_barrier(0);            // keep the reads ahead of the chain they overlap
            S2P_FWD(a, k);
            if (blk + 2 < nb) S2P_LD(a, k + 2 * B);
            __builtin_amdgcn_sched_barrier(0);
            S2P_FWD(b, k + B);
        }
        if (blk < nb) S2P_FWD(a, k1 + blk * B);
        for (k = k1 + nb * B; k < n; k++) { prev = __builtin_fmaf(z, prev, c[k]); c[k] = prev; }
    }
    const float last = (z / (z * z - 1.0f)) * (z * c[n - 2] + prev);
    c[n - 1] = last;
    float next = last;
    {
        // samples n-2 .. 0: the ragged top quad, then aligned blocks [q - B, q), downwards
        const int q = (n - 1) & ~3;
        for (int k = n - 2; k >= q; k--) { next = __builtin_fmaf(z, next, -(z * c[k])); c[k] = next; }
        const int nb = q / B;
        if (nb > 0) S2P_LD(a, q - B);
        int blk = 0;
        for (; blk + 2 <= nb; blk += 2) {
            const int k = q - (blk + 1) * B;
            S2P_LD(b, k - B);
            __builtin_amdgcn_sched_barrier(0);
            S2P_BWD(a, k);
            if (blk + 2 < nb) S2P_LD(a, k - 2 * B);
            __builtin_amdgcn_sched_barrier(0);
            S2P_BWD(b, k - B);
        }
        if (blk < nb) S2P_BWD(a, q - (blk + 1) * B);
        for (int k = q - nb * B - 1; k >= 0; k--) { next = __builtin_fmaf(z, next, -(z * c[k])); c[k] = next; }
    }
#undef S2P_LD
#undef S2P_ST
#undef S2P_FWD
#undef S2P_BWD
}

__host__ __device__ inline int lds_line_stride(int len) { return 4 * (((len + 3) / 4) | 1); }

// img: W x H row-major.  COLS = false: lines are the rows (len = W); true: the columns (len = H).
// `poison` (columns pass only): pixels whose source sample was non-finite get a NaN coefficient, so the
// gather yields NaN exactly where one of its 36 taps is bad (oracle: `anybad`).
template <int L, bool COLS>
__global__ __launch_bounds__(256) void k_prefilter_lds(float* __restrict__ img, int W, int H, const uint8_t* __restrict__ poison,
                                                       const void* __restrict__ src, int dtype, uint8_t* __restrict__ bad_out)
{
    // rows pass with `src`: the conversion of the source raster (dtype 0 f32 / 1 u16 / 2 u8) to float32 and
    // the non-finite mask `bad_out` are folded into the fill (otherwise the fill reads img itself)
    constexpr int U = 8;                                      // global loads in flight per thread
    extern __shared__ float4 lds4[];
    float* lds = (float*)lds4;
    const int len = COLS ? H : W, nlines = COLS ? W : H;
    const int S = lds_line_stride(len);
    const int g0 = blockIdx.x * L;
    const int nl = nlines - g0 < L ? nlines - g0 : L;        // lines of this block
    const float lambda = len > 1 ? (1.0f - BS_Z1) * (1.0f - 1.0f / BS_Z1) * ((1.0f - BS_Z2) * (1.0f - 1.0f / BS_Z2)) : 1.0f;
    const int tid = threadIdx.x;
    // element e of the block <-> (global offset, LDS offset); COLS: e = k * L + l; rows: e = l * len + k
    // (the nl rows of a block are one contiguous chunk of the image)
    const int total = COLS ? len * L : nl * len;
    int rl = 0, rk = 0;                                       // rows: running (l, k) of element tid + 256 * step
    const int dl = 256 / len, dk = 256 - dl * len;
    if (!COLS) { rl = tid / len; rk = tid - rl * len; }
    auto locate = [&](int e, size_t& go, int& lo) -> bool {
        if (COLS) {
            const int k = e / L, l = e % L;
            go = (size_t)k * W + g0 + l; lo = l * S + k;
            return e < total && l < nl;
        }
        go = (size_t)g0 * W + e; lo = rl * S + rk;
        const bool ok = e < total;
        rl += dl; rk += dk;
        if (rk >= len) { rk -= len; rl++; }
        return ok;
    };
    for (int e0 = tid; e0 < total; e0 += 256 * U) {
        float v[U]; size_t go[U]; int lo[U]; bool ok[U];
        #pragma unroll
        for (int u = 0; u < U; u++) {
            ok[u] = locate(e0 + 256 * u, go[u], lo[u]);
            if (!COLS && src) {
                v[u] = 0.0f;
                if (ok[u]) v[u] = dtype == 0 ? ((const float*)src)[go[u]] : dtype == 1 ? (float)((const uint16_t*)src)[go[u]] : (float)((const uint8_t*)src)[go[u]];
            } else
                v[u] = ok[u] ? img[go[u]] : 0.0f;
        }
        #pragma unroll
        for (int u = 0; u < U; u++) if (ok[u]) {
            if (!COLS && src) {
                const bool f = isfinite(v[u]);
                bad_out[go[u]] = f ? 0 : 1;
                v[u] = f ? v[u] : 0.0f;
            }
            lds[lo[u]] = len > 1 ? v[u] * lambda : v[u];
        }
    }
    __syncthreads();
#ifndef S2P_WARP_NOCHAIN
    if (tid < nl && len > 1) {
        prefilter_pole_lds(lds + tid * S, len, BS_Z1);
        prefilter_pole_lds(lds + tid * S, len, BS_Z2);
    }
#endif
    __syncthreads();
    if (!COLS) { rl = tid / len; rk = tid - rl * len; }
    for (int e0 = tid; e0 < total; e0 += 256 * U) {
        size_t go[U]; int lo[U]; bool ok[U]; uint8_t bad[U];
        #pragma unroll
        for (int u = 0; u < U; u++) {
            ok[u] = locate(e0 + 256 * u, go[u], lo[u]);
            bad[u] = (COLS && poison && ok[u]) ? poison[go[u]] : 0;
        }
        #pragma unroll
        for (int u = 0; u < U; u++) if (ok[u]) img[go[u]] = bad[u] ? __builtin_nanf("") : lds[lo[u]];
    }
}

// ---- register-resident prefilter ----------------------------------------------------------------------
// The LDS kernel above is bound by the LDS issue rate of its single chain wave (20 cycles per sample-sweep against
// 4.4 for the bare dependent v_fma_f32, tools/probes/chain_probe.hip).  Here a line (a ROW of a row-major image)
// lives in REGISTERS, RCH = 64 samples per lane, its chunks on adjacent lanes; a wavefront holds 64 / nch lines and
// sweeps them chunk by chunk: in phase c the lanes that own chunk c run their 64 dependent FMAs on statically
// indexed registers (exec-masked, no guards), then hand the running value to the neighbouring lane.  The last,
// possibly partial chunk of every line is kept in LDS instead and walked by a rolled loop (<= 64 slow steps per
// sweep) so that no step of the register code needs a bounds test.  Same operations in the same order as the
// oracle.  Columns are filtered as rows of the transposed image (k_transpose / k_transpose_poison).
#define RCH 64
// value of the neighbouring lane (whole-wave DPP shift: one VALU move instead of a trip through the LDS crossbar)
__device__ __forceinline__ float lane_below(float v) { return __builtin_bit_cast(float, dpp_mov<DPP_WAVE_SHR1>(__builtin_bit_cast(uint32_t, v), 0u)); }
__device__ __forceinline__ float lane_above(float v) { return __builtin_bit_cast(float, dpp_mov<DPP_WAVE_SHL1>(__builtin_bit_cast(uint32_t, v), 0u)); }
__device__ __forceinline__ void reg_fwd(float (&r)[RCH], float& prev, const float z, const int k0)
{
    #pragma unroll
    for (int k = 0; k < RCH; k++) if (k >= k0) { prev = __builtin_fmaf(z, prev, r[k]); r[k] = prev; }   // k0 is a literal at both call sites
}
__device__ __forceinline__ void reg_bwd(float (&r)[RCH], float& next, const float z)
{
    #pragma unroll
    for (int k = RCH - 1; k >= 0; k--) { const float t = z * r[k]; next = __builtin_fmaf(z, next, -t); r[k] = next; }
}

// img: nlines rows of len samples (row-major, in place); RCH < len <= 64 * RCH.  nch = ceil(len / RCH) lanes per
// line (>= 2), lw = 64 / nch lines per wavefront (one wavefront per block).  SRC: the samples are read from an
// integer raster `src` of the same shape instead (uint16 / uint8: no NaN to track, no separate conversion pass).
template <typename SRC>
__global__ __launch_bounds__(64) void k_prefilter_reg(float* __restrict__ img, const SRC* __restrict__ src, int len, int nlines, int nch, int lw)
{
    __shared__ float tailbuf[64][RCH + 1];
    const int lane = threadIdx.x;
    const int ll = lane / nch;
    const int line = blockIdx.x * lw + ll;
    const bool live = ll < lw && line < nlines;
    const int c_me = live ? lane - ll * nch : -1;              // my chunk; dead lanes never become active
    const bool isreg = live && c_me < nch - 1, istail = live && c_me == nch - 1;
    const int kl = len - 1 - (nch - 1) * RCH;                  // last valid index inside the tail chunk (0 .. RCH-1)
    const size_t off = live ? (size_t)line * len + (size_t)c_me * RCH : 0;
    float* p = img + off;
    const SRC* q = src + off;
    float* tl = tailbuf[lane];
    const float lambda = (1.0f - BS_Z1) * (1.0f - 1.0f / BS_Z1) * ((1.0f - BS_Z2) * (1.0f - 1.0f / BS_Z2));
    float r[RCH];
    #pragma unroll
    for (int j = 0; j < RCH; j++) r[j] = 0.0f;
    if (isreg) {
        #pragma unroll
        for (int j = 0; j < RCH; j++) r[j] = (float)q[j] * lambda;
    }
    if (istail) {
        #pragma unroll 1
        for (int j = 0; j <= kl; j++) tl[j] = (float)q[j] * lambda;
    }
    #pragma unroll
    for (int pole = 0; pole < 2; pole++) {                     // unrolled: z is a literal operand of every FMA
        const float z = pole ? BS_Z2 : BS_Z1;
        float prev = 0.0f, cm2 = 0.0f, next = 0.0f;
        // ---- causal sweep: chunk 0 (holds the >= 40 samples of the initialisation), middle chunks, tail
        if (c_me == 0) {
            float zk = z, sum = r[0];
            #pragma unroll
            for (int k = 1; k < BS_HORIZON; k++) { sum = sum + zk * r[k]; zk = zk * z; }
            r[0] = sum;
            prev = sum;
            reg_fwd(r, prev, z, 1);
        }
        #pragma unroll 1
        for (int c = 1; c < nch - 1; c++) {
            const float pin = lane_below(prev);
            if (c_me == c) { prev = pin; reg_fwd(r, prev, z, 0); }
        }
        {
            const float pin = lane_below(prev);
            if (istail) {
                prev = pin;
                #pragma unroll 4
                for (int k = 0; k <= kl; k++) { cm2 = prev; prev = __builtin_fmaf(z, cm2, tl[k]); tl[k] = prev; }
                // cm2 = c+[n-2], prev = c+[n-1]: anticausal initialisation, then the tail backwards
                next = (z / (z * z - 1.0f)) * (z * cm2 + prev);
                tl[kl] = next;
                #pragma unroll 4
                for (int k = kl - 1; k >= 0; k--) { const float t = z * tl[k]; next = __builtin_fmaf(z, next, -t); tl[k] = next; }
            }
        }
        // ---- anticausal sweep through the register chunks, last one first
        #pragma unroll 1
        for (int c = nch - 2; c >= 0; c--) {
            const float nin = lane_above(next);
            if (c_me == c) { next = nin; reg_bwd(r, next, z); }
        }
    }
    if (isreg) {
        #pragma unroll
        for (int j = 0; j < RCH; j++) p[j] = r[j];
    }
    if (istail) {
        #pragma unroll 1
        for (int j = 0; j <= kl; j++) p[j] = tl[j];
    }
}

static bool prefilter_reg_ok(int len) { return len > RCH && len <= 64 * RCH; }
template <typename SRC>
static void prefilter_reg_rows(hipStream_t st, float* img, const SRC* src, int len, int nlines)
{
    const int nch = (len + RCH - 1) / RCH, lw = 64 / nch;
    hipLaunchKernelGGL(k_prefilter_reg<SRC>, dim3((nlines + lw - 1) / lw), dim3(64), 0, st, img, src, len, nlines, nch, lw);
}

// out[x][y] = bad[y][x] ? NaN : in[y][x]   (in: rows x cols; out and bad indexed in their own row-major layouts:
// `in` is the transposed coefficient image, `bad` the mask of the final, untransposed one)
__global__ __launch_bounds__(256) void k_transpose_poison(const float* __restrict__ in, int rows, int cols, const uint8_t* __restrict__ bad, float* __restrict__ out)
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
        if (x < cols && y < rows) {
            const size_t o = (size_t)x * rows + y;
            out[o] = bad[o] ? __builtin_nanf("") : tile[tx][ty + j];
        }
    }
}

__global__ __launch_bounds__(256) void k_poison(float* __restrict__ coef, const uint8_t* __restrict__ bad, size_t n)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n && bad[i]) coef[i] = __builtin_nanf("");
}

constexpr int LDS_BUDGET = 160 * 1024;
// Lines per workgroup.  The serial chain costs the same for 1 or 32 lines per wave, so blocks are made as
// small as still leaves about one block per CU (256), and never larger than fits in LDS; 0 = no fit.
static int lds_lines(int len, int nlines)
{
    int fit = 0;
    for (int L = 32; L >= 1 && !fit; L >>= 1)
        if ((size_t)lds_line_stride(len) * L * 4 <= (size_t)LDS_BUDGET) fit = L;
    int L = fit;
    while (L > 4 && (nlines + L / 2 - 1) / (L / 2) <= 256) L >>= 1;
    return L;
}
template <int L, bool COLS>
static void launch_prefilter_lds(hipStream_t st, float* img, int W, int H, const uint8_t* poison, const void* src, int dtype, uint8_t* bad_out)
{
    const int len = COLS ? H : W, nlines = COLS ? W : H;
    const size_t bytes = (size_t)lds_line_stride(len) * L * 4;
    if (bytes > 64 * 1024) hipFuncSetAttribute((const void*)k_prefilter_lds<L, COLS>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BUDGET);
    hipLaunchKernelGGL((k_prefilter_lds<L, COLS>), dim3((nlines + L - 1) / L), dim3(256), bytes, st, img, W, H, poison, src, dtype, bad_out);
}
template <bool COLS>
static bool prefilter_lds(hipStream_t st, float* img, int W, int H, const uint8_t* poison, const void* src = nullptr, int dtype = 0, uint8_t* bad_out = nullptr)
{
    switch (lds_lines(COLS ? H : W, COLS ? W : H)) {
    case 32: launch_prefilter_lds<32, COLS>(st, img, W, H, poison, src, dtype, bad_out); return true;
    case 16: launch_prefilter_lds<16, COLS>(st, img, W, H, poison, src, dtype, bad_out); return true;
    case 8: launch_prefilter_lds<8, COLS>(st, img, W, H, poison, src, dtype, bad_out); return true;
    case 4: launch_prefilter_lds<4, COLS>(st, img, W, H, poison, src, dtype, bad_out); return true;
    case 2: launch_prefilter_lds<2, COLS>(st, img, W, H, poison, src, dtype, bad_out); return true;
    case 1: launch_prefilter_lds<1, COLS>(st, img, W, H, poison, src, dtype, bad_out); return true;
    default: return false;
    }
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

struct WarpArgs { double Hi[9]; const float* coef; int sw, sh, w, h; float* dst; };

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
        #pragma unroll
        for (int j = 0; j < 6; j++) {
            const size_t ro = (size_t)mirror_dev(iv - 2 + j, a.sh) * a.sw;
            float row = 0.0f;
            #pragma unroll
            for (int i = 0; i < 6; i++) row = row + wx[i] * a.coef[ro + xi[i]];
            acc = acc + wy[j] * row;
        }
        if (acc == acc) out = acc;       // a poisoned (non-finite source) tap makes acc NaN: canonical NaN out
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
    if (dtype < 0 || dtype > 2) { set_last_error("warp: unknown source dtype %d", dtype); return S2P_HIP_BAD_ARGUMENT; }
    static const bool use_lds = getenv("S2P_WARP_LDS") != nullptr;     // A/B switch: the LDS-resident prefilter only
    const auto convert = [&]() {
        if (dtype == 0) hipLaunchKernelGGL(k_warp_convert<float>, dim3(nb), dim3(256), 0, st, (const float*)d_src, n, coef, bad);
        else if (dtype == 1) hipLaunchKernelGGL(k_warp_convert<uint16_t>, dim3(nb), dim3(256), 0, st, (const uint16_t*)d_src, n, coef, bad);
        else hipLaunchKernelGGL(k_warp_convert<uint8_t>, dim3(nb), dim3(256), 0, st, (const uint8_t*)d_src, n, coef, bad);
    };
    if (!use_lds && prefilter_reg_ok(sw) && prefilter_reg_ok(sh)) {
        // rows in registers; columns as rows of the transposed image; the transpose back also poisons the
        // coefficients of non-finite source pixels
        // (integer rasters are converted by the first pass itself and have nothing to poison)
        if (dtype == 0) { convert(); prefilter_reg_rows<float>(st, coef, coef, sw, sh); }
        else if (dtype == 1) prefilter_reg_rows<uint16_t>(st, coef, (const uint16_t*)d_src, sw, sh);
        else prefilter_reg_rows<uint8_t>(st, coef, (const uint8_t*)d_src, sw, sh);
        hipLaunchKernelGGL(k_transpose, dim3((sw + 31) / 32, (sh + 31) / 32), dim3(256), 0, st, coef, sh, sw, tmp);
        prefilter_reg_rows<float>(st, tmp, tmp, sh, sw);
        if (dtype == 0) hipLaunchKernelGGL(k_transpose_poison, dim3((sh + 31) / 32, (sw + 31) / 32), dim3(256), 0, st, tmp, sw, sh, bad, coef);
        else hipLaunchKernelGGL(k_transpose, dim3((sh + 31) / 32, (sw + 31) / 32), dim3(256), 0, st, tmp, sw, sh, coef);
    } else {
        if (!prefilter_lds<false>(st, coef, sw, sh, nullptr, d_src, dtype, bad)) {
            // rows too long for LDS: convert, then transpose so that they become contiguous-across-threads lines
            convert();
            hipLaunchKernelGGL(k_transpose, dim3((sw + 31) / 32, (sh + 31) / 32), dim3(256), 0, st, coef, sh, sw, tmp);
            hipLaunchKernelGGL(k_prefilter_lines, dim3((sh + 63) / 64), dim3(64), 0, st, tmp, sh, sw);
            hipLaunchKernelGGL(k_transpose, dim3((sh + 31) / 32, (sw + 31) / 32), dim3(256), 0, st, tmp, sw, sh, coef);
        }
        if (!prefilter_lds<true>(st, coef, sw, sh, bad)) {
            hipLaunchKernelGGL(k_prefilter_lines, dim3((sw + 63) / 64), dim3(64), 0, st, coef, sw, sh);
            hipLaunchKernelGGL(k_poison, dim3(nb), dim3(256), 0, st, coef, bad, n);
        }
    }
    a.coef = coef; a.sw = sw; a.sh = sh; a.w = w; a.h = h; a.dst = d_dst;
    hipLaunchKernelGGL(k_warp_sample, dim3((w + 255) / 256, h), dim3(256), 0, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_last_error("kernel launch failed: %s", hipGetErrorString(e)); return S2P_HIP_RUNTIME_ERROR; }
    return S2P_HIP_OK;
}

}  // namespace s2p
