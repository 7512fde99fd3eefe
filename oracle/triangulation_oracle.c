/* oracle/triangulation_oracle.c -- TEST INFRASTRUCTURE (see oracle.h).
 *
 * CPU restatement of the per-pixel triangulation that follows the matcher in the pipeline
 * (SURVEY.md 8f rank 2): c/disp_to_h.c:14-140 (apply_homography, invert_homography,
 * disp_to_lonlatalt) and c/rpc.c:279-516 (eval_pol20, eval_nrpci, eval_nrpc_iterative, eval_rpc,
 * eval_rpci, eval_rpc_pair, rpc_height).  float64 throughout, same operation order, no FMA
 * contraction (-ffp-contract=off) => bit-exact against oracle/_ref/libdisp_to_h_ref.so, which is
 * built from exactly those line ranges of the reference files (oracle/Makefile, target ref_tri).
 * One deviation: the reference's `while (dist > 1e-18)` localisation loop has no iteration cap;
 * here (and in the HIP kernel) it stops after ORACLE_LOC_MAXIT iterations so that degenerate input
 * cannot hang a GPU.  On the fixtures it converges in 2-4 iterations.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <float.h>

#define ORACLE_LOC_MAXIT 200

/* c/rpc.c:279-298 (note the x/y inversion: col = y, lig = x) */
static double pol20(const double c[20], double x, double y, double z)
{
    double col = y, lig = x, alt = z;
    double m[20] = {1, lig, col, alt, lig*col,
        lig*alt, col*alt, lig*lig, col*col, alt*alt,
        col*lig*alt, lig*lig*lig, lig*col*col, lig*alt*alt, lig*lig*col,
        col*col*col, col*alt*alt, lig*lig*alt, col*col*alt, alt*alt*alt};
    double r = 0;
    for (int i = 0; i < 20; i++) r += c[i] * m[i];
    return r;
}

/* c/rpc.c:337-348 */
static void nrpci(double* res, const s2p_oracle_rpc* p, double x, double y, double z)
{
    double numx = pol20(p->inumx, x, y, z), denx = pol20(p->idenx, x, y, z);
    double numy = pol20(p->inumy, x, y, z), deny = pol20(p->ideny, x, y, z);
    res[0] = numx / denx;
    res[1] = numy / deny;
}

/* c/rpc.c:378-410 */
static void nrpc_iterative(double* res, const s2p_oracle_rpc* p, double x, double y, double z)
{
    double a[2], x0[2], x1[2], x2[2], xf[2] = {x, y};
    double delta = 1.0;
    if (p->delta) delta = p->delta;
    double lon = -1 * delta, lat = -1 * delta, eps = 2 * delta;
    nrpci(x0, p, lon, lat, z);
    nrpci(x1, p, lon + eps, lat, z);
    nrpci(x2, p, lon, lat + eps, z);
    for (int it = 0; it < ORACLE_LOC_MAXIT; it++) {
        double d0 = x0[0] - xf[0], d1 = x0[1] - xf[1];
        if (!(d0 * d0 + d1 * d1 > 1e-18)) break;                       /* :394 */
        double u[2] = {xf[0] - x0[0], xf[1] - x0[1]};
        double e1[2] = {x1[0] - x0[0], x1[1] - x0[1]};
        double e2[2] = {x2[0] - x0[0], x2[1] - x0[1]};
        double det = e1[0] * e2[1] - e1[1] * e2[0];                    /* :362-375 decompose_vector_basis */
        a[0] = e2[1] * u[0] - e2[0] * u[1];
        a[1] = -e1[1] * u[0] + e1[0] * u[1];
        a[0] /= det;
        a[1] /= det;
        lon += a[0] * eps;
        lat += a[1] * eps;
        eps = 0.1;
        nrpci(x0, p, lon, lat, z);
        nrpci(x1, p, lon + eps, lat, z);
        nrpci(x2, p, lon, lat + eps, z);
    }
    res[0] = lon;
    res[1] = lat;
}

/* c/rpc.c:414-439 eval_nrpc + eval_rpc */
static void rpc_direct(double* res, const s2p_oracle_rpc* p, double x, double y, double z)
{
    double nx = (x - p->offset[0]) / p->scale[0];
    double ny = (y - p->offset[1]) / p->scale[1];
    double nz = (z - p->offset[2]) / p->scale[2];
    double tmp[2];
    if (isfinite(p->numx[0])) {
        double numx = pol20(p->numx, nx, ny, nz), denx = pol20(p->denx, nx, ny, nz);
        double numy = pol20(p->numy, nx, ny, nz), deny = pol20(p->deny, nx, ny, nz);
        tmp[0] = numx / denx;
        tmp[1] = numy / deny;
    } else
        nrpc_iterative(tmp, p, nx, ny, nz);
    res[0] = tmp[0] * p->iscale[0] + p->ioffset[0];
    res[1] = tmp[1] * p->iscale[1] + p->ioffset[1];
}

/* c/rpc.c:442-452 */
static void rpc_inverse(double* res, const s2p_oracle_rpc* p, double x, double y, double z)
{
    double nx = (x - p->ioffset[0]) / p->iscale[0];
    double ny = (y - p->ioffset[1]) / p->iscale[1];
    double nz = (z - p->ioffset[2]) / p->iscale[2];
    double tmp[2];
    nrpci(tmp, p, nx, ny, nz);
    res[0] = tmp[0] * p->scale[0] + p->offset[0];
    res[1] = tmp[1] * p->scale[1] + p->offset[1];
}

/* c/rpc.c:455-462 */
static void rpc_pair(double* xp, const s2p_oracle_rpc* a, const s2p_oracle_rpc* b, double x, double y, double z)
{
    double tmp[2];
    rpc_direct(tmp, a, x, y, z);
    rpc_inverse(xp, b, tmp[0], tmp[1], z);
}

/* c/rpc.c:475-516 */
static double rpc_height(const s2p_oracle_rpc* ra, const s2p_oracle_rpc* rb, double xa, double ya, double xb, double yb, double* outerr)
{
    double h = 0;
    for (int t = 0; t < 100; t++) {
        double hstep = 1;
        double p[2], q[2];
        rpc_pair(p, ra, rb, xa, ya, h);
        rpc_pair(q, ra, rb, xa, ya, h + hstep);
        double a[2] = {q[0] - p[0], q[1] - p[1]};
        double b[2] = {xb - p[0], yb - p[1]};
        double a2 = a[0] * a[0] + a[1] * a[1];
        double lambda = (a[0] * b[0] + a[1] * b[1]) / a2;
        double z[2] = {p[0] + lambda * a[0], p[1] + lambda * a[1]};
        double err = hypot(z[0] - xb, z[1] - yb);
        *outerr = err;
        h += lambda * hstep;
        if (fabs(lambda) < 0.00001) break;
    }
    return h;
}

/* c/disp_to_h.c:14-24 */
static void apply_h(double y[2], const double h[9], const double x[2])
{
    double z = h[6] * x[0] + h[7] * x[1] + h[8];
    double tmp = x[0];
    y[0] = (h[0] * x[0] + h[1] * x[1] + h[2]) / z;
    y[1] = (h[3] * tmp + h[4] * x[1] + h[5]) / z;
}

/* c/disp_to_h.c:27-41 */
static void invert_h(double o[9], const double i[9])
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

/* c/disp_to_h.c:70-140 -- same argument list as the reference function */
void s2p_oracle_disp_to_lonlatalt(double* lonlatalt, float* err, const float* dispx, const float* dispy,
                                  const float* msk, int nx, int ny, const float* msk_orig, int w, int h,
                                  const double ha[9], const double hb[9],
                                  const s2p_oracle_rpc* rpca, const s2p_oracle_rpc* rpcb, const float bbox[4])
{
    double ha_inv[9], hb_inv[9];
    invert_h(ha_inv, ha);
    invert_h(hb_inv, hb);
    float col_min = bbox[0], col_max = bbox[1], row_min = bbox[2], row_max = bbox[3];
    for (int row = 0; row < ny; row++)
        for (int col = 0; col < nx; col++) {
            int pix = col + nx * row;
            err[pix] = NAN;
            for (int k = 0; k < 3; k++) lonlatalt[3 * pix + k] = NAN;
            if (!msk[pix]) continue;
            double p[2], q[2], lonlat[2], e = 0, z;
            double a[2] = {col, row};
            apply_h(p, ha_inv, a);
            if (round(p[0]) < col_min || round(p[0]) > col_max || round(p[1]) < row_min || round(p[1]) > row_max)
                continue;
            int x = (int)round(p[0]) - col_min;
            int y = (int)round(p[1]) - row_min;
            if ((x < w) && (y < h))
                if (!msk_orig[y * w + x]) continue;
            double dx = dispx[pix], dy = dispy[pix];
            double b[2] = {col + dx, row + dy};
            apply_h(q, hb_inv, b);
            z = rpc_height(rpca, rpcb, p[0], p[1], q[0], q[1], &e);
            rpc_direct(lonlat, rpca, p[0], p[1], z);
            lonlatalt[3 * pix + 0] = lonlat[0];
            lonlatalt[3 * pix + 1] = lonlat[1];
            lonlatalt[3 * pix + 2] = z;
            err[pix] = e;
        }
}

/* ---- stereo_corresp_to_lonlatalt (c/disp_to_h.c:43-67) ------------------------------------------------- */
void s2p_oracle_stereo_corresp_to_lonlatalt(double* lonlatalt, float* err, const float* kp_a, const float* kp_b, int n_kp,
                                            const s2p_oracle_rpc* rpca, const s2p_oracle_rpc* rpcb)
{
    for (int i = 0; i < n_kp; i++) {
        double lonlat[2], e;
        double z = rpc_height(rpca, rpcb, kp_a[2 * i], kp_a[2 * i + 1], kp_b[2 * i], kp_b[2 * i + 1], &e);
        rpc_direct(lonlat, rpca, kp_a[2 * i], kp_a[2 * i + 1], z);
        lonlatalt[3 * i + 0] = lonlat[0];
        lonlatalt[3 * i + 1] = lonlat[1];
        lonlatalt[3 * i + 2] = z;
        err[i] = e;
    }
}

/* ---- count_3d_neighbors / remove_isolated_3d_points (c/disp_to_h.c:143-230) ---------------------------- */
static float sqdist3(const double* a, const double* b)          /* :143-149: float differences, float products */
{
    float x = (a[0] - b[0]), y = (a[1] - b[1]), z = (a[2] - b[2]);
    return x * x + y * y + z * z;
}

void s2p_oracle_count_3d_neighbors(int* count, const double* xyz, int nx, int ny, float r, int p)   /* :152-174 */
{
    for (int y = 0; y < ny; y++)
        for (int x = 0; x < nx; x++) {
            const double* v = xyz + ((size_t)x + (size_t)nx * y) * 3;
            int c = 0;
            int i0 = y > p ? -p : -y, i1 = y < ny - p ? p : ny - y - 1;
            int j0 = x > p ? -p : -x, j1 = x < nx - p ? p : nx - x - 1;
            for (int i = i0; i <= i1; i++)
                for (int j = j0; j <= j1; j++)
                    if (sqdist3(xyz + ((size_t)(x + j) + (size_t)nx * (y + i)) * 3, v) < r * r) c++;
            count[x + nx * y] = c;
        }
}

/* :177-230.  Stated as the fixed point the reference's raster-order loop converges to: a rejected point is
 * saved iff a chain of window-(2q+1) neighbours, each closer than r to the next, links it to an accepted point.
 * Computed by a breadth-first flood from the accepted points (any sweep order reaches the same set). */
void s2p_oracle_remove_isolated_3d_points(double* xyz, int nx, int ny, float r, int p, int n, int q)
{
    size_t npx = (size_t)nx * ny;
    int* count = (int*)malloc(npx * sizeof(int));
    unsigned char* rejected = (unsigned char*)malloc(npx);
    size_t* queue = (size_t*)malloc(npx * sizeof(size_t));
    size_t head = 0, tail = 0;
    s2p_oracle_count_3d_neighbors(count, xyz, nx, ny, r, p);
    for (size_t i = 0; i < npx; i++) { rejected[i] = count[i] < n; if (!rejected[i]) queue[tail++] = i; }
    while (head < tail) {
        size_t o = queue[head++];
        int x = (int)(o % nx), y = (int)(o / nx);
        for (int yy = y - q; yy <= y + q; yy++) {
            if (yy < 0 || yy > ny - 1) continue;
            for (int xx = x - q; xx <= x + q; xx++) {
                if (xx < 0 || xx > nx - 1) continue;
                size_t t = (size_t)xx + (size_t)yy * nx;
                if (rejected[t] && sqdist3(xyz + t * 3, xyz + o * 3) < r * r) { rejected[t] = 0; queue[tail++] = t; }
            }
        }
    }
    for (size_t i = 0; i < npx; i++)
        if (rejected[i]) for (int c = 0; c < 3; c++) xyz[c + i * 3] = NAN;
    free(queue); free(rejected); free(count);
}


/* ---- triangulation.height_map_to_xyz (s2p/triangulation.py:165-219), the localisation of a height map ----------
 * For every pixel (c, r) of a w x h float32 height map whose altitude is not NaN: lon, lat = localisation of the
 * image point (c + off_x, r + off_y) at that altitude.  The reference calls rpcm's RPCModel.localization (a pip
 * dependency, not under /root/reference: an iterative inversion of the projection when the RPC has no direct
 * coefficients); here the inversion is the one of c/rpc.c:378-439 restated above (eval_rpc), which solves the same
 * equation to |residual|^2 <= 1e-18 in normalised image coordinates -- parity with rpcm is at that level, not bitwise.
 * lonlatalt: h x w x 3 float64, NaN where the height is NaN. */
void s2p_oracle_height_map_to_lonlatalt(const s2p_oracle_rpc* rpc, const float* heights, int w, int h, int off_x, int off_y,
                                        double* lonlatalt)
{
    for (int r = 0; r < h; r++)
        for (int c = 0; c < w; c++) {
            double* o = lonlatalt + 3 * ((size_t)r * w + c);
            float z = heights[(size_t)r * w + c];
            if (isnan(z)) { o[0] = o[1] = o[2] = NAN; continue; }
            double ll[2];
            rpc_direct(ll, rpc, (double)(c + off_x), (double)(r + off_y), (double)z);
            o[0] = ll[0]; o[1] = ll[1]; o[2] = (double)z;
        }
}
