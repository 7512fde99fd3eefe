// s2p_amd/csrc/tri_kernels.hip -- per-pixel triangulation on gfx950: disparity -> (lon, lat, alt) through
// two RPC camera models, the step that follows the matcher in the pipeline.
// Arithmetic of c/disp_to_h.c:14-140 (apply_homography, disp_to_lonlatalt) and c/rpc.c:279-516 (eval_pol20,
// eval_nrpci, eval_nrpc_iterative, eval_rpc, eval_rpci, eval_rpc_pair, rpc_height): float64, same operation
// order, no FMA contraction (-ffp-contract=off) => bit-exact against the reference's own functions
// (oracle/_ref/libdisp_to_h_ref.so) on the lon/lat/alt outputs.  One thread per rectified pixel; the two
// 181-double RPC structs are wave-uniform (scalar loads); the data-dependent loops (height refinement, <= 100
// iterations; iterative localisation, capped at 200 where the reference has no cap) diverge per lane.
#include "common.hpp"

#include <utility>

namespace s2p {

#define TRI_LOC_MAXIT 200

__device__ __forceinline__ double tri_pol20(const double* __restrict__ c, double x, double y, double z)
{
    const double col = y, lig = x, alt = z;                        // c/rpc.c:281-284 (x/y inversion)
    const double m[20] = {1, lig, col, alt, lig*col,
        lig*alt, col*alt, lig*lig, col*col, alt*alt,
        col*lig*alt, lig*lig*lig, lig*col*col, lig*alt*alt, lig*lig*col,
        col*col*col, col*alt*alt, lig*lig*alt, col*col*alt, alt*alt*alt};
    double r = 0;
    #pragma unroll
    for (int i = 0; i < 20; i++) r += c[i] * m[i];
    return r;
}

__device__ __forceinline__ void tri_nrpci(double* res, const s2p_rpc* __restrict__ p, double x, double y, double z)
{
    const double numx = tri_pol20(p->inumx, x, y, z), denx = tri_pol20(p->idenx, x, y, z);
    const double numy = tri_pol20(p->inumy, x, y, z), deny = tri_pol20(p->ideny, x, y, z);
    res[0] = numx / denx;
    res[1] = numy / deny;
}

__device__ void tri_nrpc_iterative(double* res, const s2p_rpc* __restrict__ p, double x, double y, double z)
{
    double a[2], x0[2], x1[2], x2[2];
    const double xf[2] = {x, y};
    double delta = 1.0;
    if (p->delta) delta = p->delta;
    double lon = -1 * delta, lat = -1 * delta, eps = 2 * delta;
    tri_nrpci(x0, p, lon, lat, z);
    tri_nrpci(x1, p, lon + eps, lat, z);
    tri_nrpci(x2, p, lon, lat + eps, z);
    for (int it = 0; it < TRI_LOC_MAXIT; it++) {
        const double d0 = x0[0] - xf[0], d1 = x0[1] - xf[1];
        if (!(d0 * d0 + d1 * d1 > 1e-18)) break;
        const double u[2] = {xf[0] - x0[0], xf[1] - x0[1]};
        const double e1[2] = {x1[0] - x0[0], x1[1] - x0[1]};
        const double e2[2] = {x2[0] - x0[0], x2[1] - x0[1]};
        const double det = e1[0] * e2[1] - e1[1] * e2[0];
        a[0] = e2[1] * u[0] - e2[0] * u[1];
        a[1] = -e1[1] * u[0] + e1[0] * u[1];
        a[0] /= det;
        a[1] /= det;
        lon += a[0] * eps;
        lat += a[1] * eps;
        eps = 0.1;
        tri_nrpci(x0, p, lon, lat, z);
        tri_nrpci(x1, p, lon + eps, lat, z);
        tri_nrpci(x2, p, lon, lat + eps, z);
    }
    res[0] = lon;
    res[1] = lat;
}

__device__ void tri_rpc_direct(double* res, const s2p_rpc* __restrict__ p, double x, double y, double z)
{
    const double nx = (x - p->offset[0]) / p->scale[0];
    const double ny = (y - p->offset[1]) / p->scale[1];
    const double nz = (z - p->offset[2]) / p->scale[2];
    double tmp[2];
    if (isfinite(p->numx[0])) {
        const double numx = tri_pol20(p->numx, nx, ny, nz), denx = tri_pol20(p->denx, nx, ny, nz);
        const double numy = tri_pol20(p->numy, nx, ny, nz), deny = tri_pol20(p->deny, nx, ny, nz);
        tmp[0] = numx / denx;
        tmp[1] = numy / deny;
    } else
        tri_nrpc_iterative(tmp, p, nx, ny, nz);
    res[0] = tmp[0] * p->iscale[0] + p->ioffset[0];
    res[1] = tmp[1] * p->iscale[1] + p->ioffset[1];
}

__device__ __forceinline__ void tri_rpc_inverse(double* res, const s2p_rpc* __restrict__ p, double x, double y, double z)
{
    const double nx = (x - p->ioffset[0]) / p->iscale[0];
    const double ny = (y - p->ioffset[1]) / p->iscale[1];
    const double nz = (z - p->ioffset[2]) / p->iscale[2];
    double tmp[2];
    tri_nrpci(tmp, p, nx, ny, nz);
    res[0] = tmp[0] * p->scale[0] + p->offset[0];
    res[1] = tmp[1] * p->scale[1] + p->offset[1];
}

__device__ __forceinline__ void tri_rpc_pair(double* xp, const s2p_rpc* a, const s2p_rpc* b, double x, double y, double z)
{
    double tmp[2];
    tri_rpc_direct(tmp, a, x, y, z);
    tri_rpc_inverse(xp, b, tmp[0], tmp[1], z);
}

__device__ double tri_rpc_height(const s2p_rpc* ra, const s2p_rpc* rb, double xa, double ya, double xb, double yb, double* outerr)
{
    double h = 0;
    for (int t = 0; t < 100; t++) {                                 // RPCH_MAXIT, c/rpc.c:475
        const double hstep = 1;
        double p[2], q[2];
        tri_rpc_pair(p, ra, rb, xa, ya, h);
        tri_rpc_pair(q, ra, rb, xa, ya, h + hstep);
        const double a[2] = {q[0] - p[0], q[1] - p[1]};
        const double b[2] = {xb - p[0], yb - p[1]};
        const double a2 = a[0] * a[0] + a[1] * a[1];
        const double lambda = (a[0] * b[0] + a[1] * b[1]) / a2;
        const double z[2] = {p[0] + lambda * a[0], p[1] + lambda * a[1]};
        *outerr = hypot(z[0] - xb, z[1] - yb);
        h += lambda * hstep;
        if (fabs(lambda) < 0.00001) break;                          // RPCH_LAMBDA_STOP
    }
    return h;
}

__device__ __forceinline__ void tri_apply_h(double y[2], const double* h, const double x[2])
{
    const double z = h[6] * x[0] + h[7] * x[1] + h[8];
    const double tmp = x[0];
    y[0] = (h[0] * x[0] + h[1] * x[1] + h[2]) / z;
    y[1] = (h[3] * tmp + h[4] * x[1] + h[5]) / z;
}

struct TriArgs {
    double ha_inv[9], hb_inv[9];
    const s2p_rpc* rpc;               // device: [0] = rpca, [1] = rpcb
    const float *dispx, *dispy, *msk, *msk_orig;
    int nx, ny, w, h;
    float bbox[4];
    double* lonlatalt; float* err;
};

__global__ __launch_bounds__(64) void k_disp_to_lonlatalt(TriArgs a)
{
    const int col = blockIdx.x * 64 + threadIdx.x, row = blockIdx.y;
    if (col >= a.nx) return;
    const int pix = col + a.nx * row;
    const double nan = __builtin_nan("");
    double o0 = nan, o1 = nan, o2 = nan;
    float oe = __builtin_nanf("");
    const float col_min = a.bbox[0], col_max = a.bbox[1], row_min = a.bbox[2], row_max = a.bbox[3];
    if (a.msk[pix]) {
        double p[2], q[2];
        const double c[2] = {(double)col, (double)row};
        tri_apply_h(p, a.ha_inv, c);
        const bool inside = !(round(p[0]) < col_min || round(p[0]) > col_max || round(p[1]) < row_min || round(p[1]) > row_max);
        bool keep = inside;
        if (inside) {
            const int x = (int)round(p[0]) - col_min;                // float arithmetic, as the reference (:120-121)
            const int y = (int)round(p[1]) - row_min;
            if ((x < a.w) && (y < a.h))
                if (!a.msk_orig[y * a.w + x]) keep = false;
        }
        if (keep) {
            const double dx = a.dispx[pix], dy = a.dispy ? (double)a.dispy[pix] : 0.0;
            const double b[2] = {col + dx, row + dy};
            tri_apply_h(q, a.hb_inv, b);
            double e = 0, lonlat[2];
            const double z = tri_rpc_height(&a.rpc[0], &a.rpc[1], p[0], p[1], q[0], q[1], &e);
            tri_rpc_direct(lonlat, &a.rpc[0], p[0], p[1], z);
            o0 = lonlat[0]; o1 = lonlat[1]; o2 = z; oe = (float)e;
        }
    }
    a.lonlatalt[3 * (size_t)pix + 0] = o0;
    a.lonlatalt[3 * (size_t)pix + 1] = o1;
    a.lonlatalt[3 * (size_t)pix + 2] = o2;
    a.err[pix] = oe;
}

static void invert_h(double o[9], const double i[9])               // c/disp_to_h.c:27-41
{
    double det = i[0]*i[4]*i[8] + i[2]*i[3]*i[7] + i[1]*i[5]*i[6]
               - i[2]*i[4]*i[6] - i[1]*i[3]*i[8] - i[0]*i[5]*i[7];
    o[0] = (i[4]*i[8] - i[5]*i[7]) / det;
    o[1] = (i[2]*i[7] - i[1]*i[8]) / det;
    o[2] = (i[1]*i[5] - i[2]*i[4]) / det;
    o[3] = (i[5]*i[6] - i[3]*i[8]) / det;
    o[4] = (i[0]*i[8] - i[2]*i[6]) / det;
    o[5] = (i[2]*i[3] - i[0]*i[5]) / det;
    o[6] = (i[3]*i[7] - i[4]*i[6]) / det;
    o[7] = (i[1]*i[6] - i[0]*i[7]) / det;
    o[8] = (i[0]*i[4] - i[1]*i[3]) / det;
}

// all pointers are device pointers; d_rpc holds the two structs back to back
int tri_enqueue(s2p_hip_ctx* ctx, const float* d_dispx, const float* d_dispy, const float* d_msk, int nx, int ny,
                const float* d_msk_orig, int w, int h, const double ha[9], const double hb[9], const s2p_rpc* d_rpc,
                const float bbox[4], double* d_lonlatalt, float* d_err)
{
    TriArgs a;
    invert_h(a.ha_inv, ha);
    invert_h(a.hb_inv, hb);
    a.rpc = d_rpc; a.dispx = d_dispx; a.dispy = d_dispy; a.msk = d_msk; a.msk_orig = d_msk_orig;
    a.nx = nx; a.ny = ny; a.w = w; a.h = h;
    for (int i = 0; i < 4; i++) a.bbox[i] = bbox[i];
    a.lonlatalt = d_lonlatalt; a.err = d_err;
    StageScope s(ctx, "triangulate");
    hipLaunchKernelGGL(k_disp_to_lonlatalt, dim3((nx + 63) / 64, ny), dim3(64), 0, ctx->stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_last_error("kernel launch failed: %s", hipGetErrorString(e)); return S2P_HIP_RUNTIME_ERROR; }
    return S2P_HIP_OK;
}

// ---- stereo_corresp_to_lonlatalt (c/disp_to_h.c:43-67): one 3-D point per keypoint match ------------
__global__ __launch_bounds__(64) void k_corresp_to_lonlatalt(const float* __restrict__ kpa, const float* __restrict__ kpb, int n,
                                                             const s2p_rpc* __restrict__ rpc, double* __restrict__ lonlatalt, float* __restrict__ err)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    double e, lonlat[2];
    const double xa = kpa[2 * i], ya = kpa[2 * i + 1];
    const double z = tri_rpc_height(&rpc[0], &rpc[1], xa, ya, kpb[2 * i], kpb[2 * i + 1], &e);
    tri_rpc_direct(lonlat, &rpc[0], xa, ya, z);
    lonlatalt[3 * i + 0] = lonlat[0]; lonlatalt[3 * i + 1] = lonlat[1]; lonlatalt[3 * i + 2] = z;
    err[i] = (float)e;
}

int corresp_enqueue(s2p_hip_ctx* ctx, const float* d_kpa, const float* d_kpb, int n, const s2p_rpc* d_rpc, double* d_lonlatalt, float* d_err)
{
    hipLaunchKernelGGL(k_corresp_to_lonlatalt, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, d_kpa, d_kpb, n, d_rpc, d_lonlatalt, d_err);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_last_error("kernel launch failed: %s", hipGetErrorString(e)); return S2P_HIP_RUNTIME_ERROR; }
    return S2P_HIP_OK;
}

// ---- triangulation.height_map_to_xyz before the CRS conversion (s2p/triangulation.py:165-219) -------------------------
// lon, lat of every pixel (c + off_x, r + off_y) of a float32 height map at its altitude, through the iterative
// localisation of c/rpc.c:378-439 (the reference calls rpcm's RPCModel.localization here: same equation, parity at the
// 1e-9 pixel level of both iterations' stopping rules, not bitwise).  NaN heights give NaN triples.
__global__ __launch_bounds__(64) void k_height_map_to_lonlatalt(const s2p_rpc* __restrict__ rpc, const float* __restrict__ hm, int w, int h,
                                                                int off_x, int off_y, double* __restrict__ lonlatalt)
{
    const int c = blockIdx.x * 64 + threadIdx.x, r = blockIdx.y;
    if (c >= w) return;
    const size_t i = (size_t)r * w + c;
    const float z = hm[i];
    const double nan = __builtin_nan("");
    double o0 = nan, o1 = nan, o2 = nan;
    if (z == z) {
        double ll[2];
        tri_rpc_direct(ll, rpc, (double)(c + off_x), (double)(r + off_y), (double)z);
        o0 = ll[0]; o1 = ll[1]; o2 = (double)z;
    }
    lonlatalt[3 * i + 0] = o0; lonlatalt[3 * i + 1] = o1; lonlatalt[3 * i + 2] = o2;
}

int height_map_localize_enqueue(s2p_hip_ctx* ctx, const s2p_rpc* d_rpc, const float* d_hm, int w, int h, int off_x, int off_y, double* d_lonlatalt)
{
    StageScope s(ctx, "localize");
    hipLaunchKernelGGL(k_height_map_to_lonlatalt, dim3((w + 63) / 64, h), dim3(64), 0, ctx->stream, d_rpc, d_hm, w, h, off_x, off_y, d_lonlatalt);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_last_error("kernel launch failed: %s", hipGetErrorString(e)); return S2P_HIP_RUNTIME_ERROR; }
    return S2P_HIP_OK;
}

// ---- count_3d_neighbors / remove_isolated_3d_points (c/disp_to_h.c:143-230) -----------------------------
// squared distance exactly as the reference: double differences rounded to float, float products and sums
__device__ __forceinline__ float sqdist3(const double* __restrict__ a, const double* __restrict__ b)
{
    const float x = (float)(a[0] - b[0]), y = (float)(a[1] - b[1]), z = (float)(a[2] - b[2]);
    return x * x + y * y + z * z;
}

__global__ __launch_bounds__(256) void k_count_3d_neighbors(const double* __restrict__ xyz, int nx, int ny, float r, int p, int* __restrict__ count)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= nx) return;
    const double* v = xyz + ((size_t)x + (size_t)nx * y) * 3;
    const int i0 = y > p ? -p : -y, i1 = y < ny - p ? p : ny - y - 1;
    const int j0 = x > p ? -p : -x, j1 = x < nx - p ? p : nx - x - 1;
    const float r2 = r * r;
    int c = 0;
    for (int i = i0; i <= i1; i++)
        for (int j = j0; j <= j1; j++)
            c += sqdist3(xyz + ((size_t)(x + j) + (size_t)nx * (y + i)) * 3, v) < r2 ? 1 : 0;
    count[x + nx * y] = c;
}

// The reference's "mercy" loop saves a rejected point as soon as one non-rejected point of its (2q+1)^2 window
// is closer than r, sweeping in raster order until nothing changes.  Its fixed point is order independent --
// the rejected points that stay rejected are exactly those no chain of close window-neighbours connects to an
// initially accepted point -- so parallel sweeps until no change give the same set.
__global__ __launch_bounds__(256) void k_reject_init(const int* __restrict__ count, size_t n, int minn, uint8_t* __restrict__ rejected)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) rejected[i] = count[i] < minn ? 1 : 0;
}
// In place: `rejected` only ever goes 1 -> 0, so a thread that reads a neighbour's stale 1 merely defers a
// rescue to the next sweep; the sweep after the last change sees every write (kernel boundary) and changes nothing.
__global__ __launch_bounds__(256) void k_mercy_sweep(const double* __restrict__ xyz, int nx, int ny, float r, int q,
                                                     uint8_t* rejected, int* __restrict__ changed)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= nx) return;
    const size_t pos = (size_t)x + (size_t)y * nx;
    if (!rejected[pos]) return;
    const float r2 = r * r;
    for (int yy = max(y - q, 0); yy <= min(y + q, ny - 1); yy++)
        for (int xx = max(x - q, 0); xx <= min(x + q, nx - 1); xx++) {
            const size_t o = (size_t)xx + (size_t)yy * nx;
            if (!rejected[o] && sqdist3(xyz + pos * 3, xyz + o * 3) < r2) { rejected[pos] = 0; *changed = 1; return; }
        }
}
__global__ __launch_bounds__(256) void k_apply_rejected(const uint8_t* __restrict__ rejected, size_t n, double* __restrict__ xyz)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n && rejected[i]) { const double nan = __builtin_nan(""); xyz[3 * i] = nan; xyz[3 * i + 1] = nan; xyz[3 * i + 2] = nan; }
}

int count3d_enqueue(s2p_hip_ctx* ctx, const double* d_xyz, int nx, int ny, float r, int p, int* d_count)
{
    hipLaunchKernelGGL(k_count_3d_neighbors, dim3((nx + 255) / 256, ny), dim3(256), 0, ctx->stream, d_xyz, nx, ny, r, p, d_count);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_last_error("kernel launch failed: %s", hipGetErrorString(e)); return S2P_HIP_RUNTIME_ERROR; }
    return S2P_HIP_OK;
}

// d_rej: nx*ny bytes, d_flag: one int.  The sweep loop is data dependent: 8 sweeps per host check.
int remove_isolated_enqueue(s2p_hip_ctx* ctx, double* d_xyz, int nx, int ny, float r, int p, int n, int q,
                            int* d_count, uint8_t* d_rej, int* d_flag)
{
    hipStream_t st = ctx->stream;
    const size_t npx = (size_t)nx * ny;
    const unsigned nb = (unsigned)((npx + 255) / 256);
    int rc = count3d_enqueue(ctx, d_xyz, nx, ny, r, p, d_count);
    if (rc) return rc;
    hipLaunchKernelGGL(k_reject_init, dim3(nb), dim3(256), 0, st, d_count, npx, n, d_rej);
    for (;;) {
        S2P_HIP_CHECK(hipMemsetAsync(d_flag, 0, sizeof(int), st));
        for (int k = 0; k < 8; k++)
            hipLaunchKernelGGL(k_mercy_sweep, dim3((nx + 255) / 256, ny), dim3(256), 0, st, d_xyz, nx, ny, r, q, d_rej, d_flag);
        int changed = 0;
        S2P_HIP_CHECK(hipMemcpyAsync(&changed, d_flag, sizeof(int), hipMemcpyDeviceToHost, st));
        S2P_HIP_CHECK(hipStreamSynchronize(st));
        if (!changed) break;
    }
    uint8_t* cur = d_rej;
    hipLaunchKernelGGL(k_apply_rejected, dim3(nb), dim3(256), 0, st, cur, npx, d_xyz);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_last_error("kernel launch failed: %s", hipGetErrorString(e)); return S2P_HIP_RUNTIME_ERROR; }
    return S2P_HIP_OK;
}

// ---- triangulation.height_map, second half (s2p/triangulation.py:376-389): the rectified altitude plane is carried to
// the grid of the original image with scipy's ndimage.affine_transform -- order 1 on nan_to_num(heights), order 0 on the
// NaN mask, a 3x3 binary dilation of that mask, NaN where it is set.  The kernels restate scipy's arithmetic in its own
// order (ni_interpolation.c NI_GeometricTransform, scipy 1.15): coordinate = (o0 m0 + o1 m1) + shift per axis;
// "constant" mode = cval 0 for a coordinate outside [0, len - 1]; linear weights w0 = 1 - frac, w1 = 1 - w0; the 4
// taps summed in row-major order as (coeff * w_axis0) * w_axis1 -- bit-identical float64 output
// (tests/test_gpu_triangulation.py compares with scipy itself, which is importable next to the tests).
// scipy works on the TRANSPOSED arrays (axis 0 = x): out[oy][ox] = A[H (ox, oy)] with A[a][b] = heights[b][a].
__global__ __launch_bounds__(256) void k_height_transfer(const double* __restrict__ hm, int wr, int hr, double m00, double m01, double sh0,
                                                         double m10, double m11, double sh1, int w, int h,
                                                         double* __restrict__ val, uint8_t* __restrict__ nanflag)
{
    const int ox = blockIdx.x * 256 + threadIdx.x, oy = blockIdx.y;
    if (ox >= w) return;
    const double o0 = (double)ox, o1 = (double)oy;
    double cc0 = 0.0 + o0 * m00; cc0 += o1 * m01; cc0 += sh0;
    double cc1 = 0.0 + o0 * m10; cc1 += o1 * m11; cc1 += sh1;
    const int n0 = wr, n1 = hr;
    const bool inb = !(cc0 < 0 || cc0 > n0 - 1 || cc1 < 0 || cc1 > n1 - 1);      // NaN coordinates compare false: scipy proceeds, we too
    double t = 0.0;
    bool isn = false;
    if (inb && cc0 == cc0 && cc1 == cc1) {
        const double f0 = floor(cc0), f1 = floor(cc1);
        const int s0 = (int)f0, s1 = (int)f1;
        const double x0 = cc0 - f0, x1 = cc1 - f1;
        const double w00 = 1.0 - x0, w01 = 1.0 - w00, w10 = 1.0 - x1, w11 = 1.0 - w10;
        auto tap = [&](int a, int b) -> double {
            if (a < 0 || a >= n0 || b < 0 || b >= n1) return 0.0;            // only reached with a zero weight
            const double v = hm[(size_t)b * wr + a];
            if (v != v) return 0.0;                                            // np.nan_to_num
            if (isinf(v)) return v > 0 ? 1.7976931348623157e308 : -1.7976931348623157e308;
            return v;
        };
        t = t + (tap(s0, s1) * w00) * w10;
        t = t + (tap(s0, s1 + 1) * w00) * w11;
        t = t + (tap(s0 + 1, s1) * w01) * w10;
        t = t + (tap(s0 + 1, s1 + 1) * w01) * w11;
        const int r0 = min(max((int)floor(cc0 + 0.5), 0), n0 - 1), r1 = min(max((int)floor(cc1 + 0.5), 0), n1 - 1);
        const double q = hm[(size_t)r1 * wr + r0];
        isn = q != q;
    }
    val[(size_t)oy * w + ox] = t;
    nanflag[(size_t)oy * w + ox] = isn ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_height_apply_nan(const double* __restrict__ val, const uint8_t* __restrict__ nanflag, int w, int h,
                                                          double* __restrict__ out)
{
    const int ox = blockIdx.x * 256 + threadIdx.x, oy = blockIdx.y;
    if (ox >= w) return;
    bool any = false;                                                         // binary_dilation, 3x3 ones, border 0
    for (int dy = -1; dy <= 1; dy++)
        for (int dx = -1; dx <= 1; dx++) {
            const int x = ox + dx, y = oy + dy;
            if (x >= 0 && x < w && y >= 0 && y < h) any |= nanflag[(size_t)y * w + x] != 0;
        }
    out[(size_t)oy * w + ox] = any ? __builtin_nan("") : val[(size_t)oy * w + ox];
}

// d_hm: hr x wr float64 (rectified grid); H: the 3x3 matrix handed to affine_transform (np.dot(H1, translation(x, y)),
// bottom row [0, 0, 1]); d_val / d_flag: scratch of w*h doubles / bytes; d_out: h x w float64
int height_transfer_enqueue(s2p_hip_ctx* ctx, const double* d_hm, int wr, int hr, const double H[9], int w, int h,
                            double* d_val, uint8_t* d_flag, double* d_out)
{
    hipStream_t st = ctx->stream;
    const dim3 grid((w + 255) / 256, h);
    hipLaunchKernelGGL(k_height_transfer, grid, dim3(256), 0, st, d_hm, wr, hr, H[0], H[1], H[2], H[3], H[4], H[5], w, h, d_val, d_flag);
    hipLaunchKernelGGL(k_height_apply_nan, grid, dim3(256), 0, st, d_val, d_flag, w, h, d_out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_last_error("kernel launch failed: %s", hipGetErrorString(e)); return S2P_HIP_RUNTIME_ERROR; }
    return S2P_HIP_OK;
}

}  // namespace s2p
