/* oracle/rasterize_oracle.c -- TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke, bench cpu_baseline).
 *
 * CPU restatement of the DSM rasterisation s2p delegates to the `plyflatten` package
 * (s2p/__init__.py:31 import, :462-466 call `plyflatten_from_plyfiles_list(clouds, resolution=r, roi=roi,
 * radius=cfg['dsm_radius'], sigma=cfg['dsm_sigma'])`; tests/rasterization_test.py:13-28).
 *
 * plyflatten is a pip dependency (setup.py:52 `plyflatten>=0.2.0`), NOT vendored under /root/reference: the
 * algorithm below restates its published C core (`rasterize_cloud` of plyflatten 0.2.0):
 *   for every point, in input order:  i = floor((x - xoff) / res), j = floor((-y + yoff) / res);
 *     for k1, k2 in [-radius, radius]^2 with k1^2 + k2^2 <= radius^2, cell (i + k1, j + k2) inside the raster:
 *       weight = sigma == inf ? 1 : exp(-d^2 / (2 sigma^2)), d = distance of the point to the cell centre (float);
 *       per band: avg = (v * weight + cnt * avg) / (weight + cnt) with float avg / cnt / weight and C's promotion
 *       rules (v is double), then cnt += weight;
 *   cells never touched -> NaN.
 * PINNING: with radius 0 (s2p's default) this reproduces the reference's own golden -- input_ply/cloud.ply ->
 * expected_output/plyflatten/dsm_40cm.tiff (tests/rasterization_test.py) -- BIT FOR BIT on all 217 512 cells
 * (tests/golden/plyflatten_crop.npz holds a window of it; the reference's own bar is np.allclose).  The
 * radius > 0 / finite sigma branch has no golden in the reference tree: "parity unpinned" for that branch. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

int s2p_oracle_plyflatten(const double* pts, int npts, int nb, double xoff, double yoff, double res,
                          int xsize, int ysize, int radius, float sigma, float* raster)
{
    if (!pts || !raster || npts < 0 || nb <= 0 || xsize <= 0 || ysize <= 0 || radius < 0 || !(res > 0)) return -1;
    const size_t ncell = (size_t)xsize * ysize;
    float* avg = (float*)calloc(ncell * nb, sizeof(float));
    float* cnt = (float*)calloc(ncell, sizeof(float));
    if (!avg || !cnt) { free(avg); free(cnt); return -2; }
    const int unweighted = isinf(sigma);
    for (int k = 0; k < npts; k++) {
        const double* p = pts + (size_t)k * (2 + nb);
        const double xx = p[0], yy = p[1];
        const double fi = floor((xx - xoff) / res), fj = floor((-yy - (-yoff)) / res);
        if (!(fabs(fi) < 1e9) || !(fabs(fj) < 1e9)) continue;   /* non-finite / absurd coordinate: (int) would be undefined */
        const int i = (int)fi, j = (int)fj;
        for (int k1 = -radius; k1 <= radius; k1++)
            for (int k2 = -radius; k2 <= radius; k2++) {
                if (k1 * k1 + k2 * k2 > radius * radius) continue;
                const int ii = i + k1, jj = j + k2;
                if (ii < 0 || jj < 0 || ii >= xsize || jj >= ysize) continue;
                float weight = 1.0f;
                if (!unweighted) {
                    const float dx = (float)(xx - (xoff + res * (0.5 + ii)));
                    const float dy = (float)(yy - (yoff - res * (0.5 + jj)));
                    const float d = sqrtf(dx * dx + dy * dy);   /* (hypot in the C code: <= 1 ulp apart, and not reproducible across libms) */
                    weight = (float)exp((double)(-d * d / (2 * sigma * sigma)));
                }
                const size_t c = (size_t)xsize * jj + ii;
                for (int b = 0; b < nb; b++) {
                    float* a = &avg[c * nb + b];
                    *a = (float)((p[2 + b] * weight + cnt[c] * *a) / (weight + cnt[c]));
                }
                cnt[c] += weight;
            }
    }
    for (size_t c = 0; c < ncell; c++)
        for (int b = 0; b < nb; b++) raster[c * nb + b] = cnt[c] ? avg[c * nb + b] : NAN;
    free(avg); free(cnt);
    return 0;
}
