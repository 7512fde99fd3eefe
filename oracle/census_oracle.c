/* oracle/census_oracle.c -- TEST INFRASTRUCTURE (see oracle.h).
 *
 * PARITY UNPINNED AT SOURCE LEVEL: this is the CPU statement of the census / 8-path SGM matcher that
 * stands in for the reference's `mgm` and `mgm_multi` binaries (s2p/block_matching.py:155-188,
 * 269-310).  Their sources (gfacciol/mgm, branches master and multiscale; .gitmodules:5-8,13-17)
 * are un-vendored submodules -- the directories are empty in /root/reference and no commit pin is
 * recoverable -- so nothing here can follow the reference line by line.  What IS taken from the
 * reference: the call sites' parameters (census on a CENSUS_NCC_WIN^2 window, 8 directions, P1 = 8,
 * P2 = 32 (x stereo_regularity_multiplier), vfit sub-pixel, left-right test with TESTLRRL_TAU = 1,
 * MEDIAN = 1 for 'mgm', REMOVESMALLCC for 'mgm_multi', MINDIFF = -1), the file conventions
 * (float32, NaN = invalid, s2p sign convention) and the one stored mgm output tile
 * (tests/data/input_triangulation/pair_1/rectified_disp.tif), against which tests/ measure
 * STATISTICAL agreement.  The algorithm itself is the published semi-global matching recurrence
 * (Hirschmuller, PAMI 2008, eq. 12-14) on a census/Hamming cost (Zabih & Woodfill 1994), with the data term
 * counted once in the final sum (S = sum_r L_r - 7 C; Drory et al. 2014, mgm's TSGM_FIX_OVERCOUNT default:
 * 97.7 % -> 98.9 % of the stored tile within 0.5 px); MGM's 2-neighbour recursion (Facciolo, de Franchis,
 * Meinhardt, BMVC 2015; a prototype reaches 99.5 % with it) is NOT reproduced.
 * The HIP kernels must match THIS file bit for bit (integer pipeline + one IEEE division).
 *
 * `mgm_multi` (s2p/block_matching.py:269-310: `-S 6`, SUBPIX=2) adds two things on top, both UNPINNED (no artefact of
 * the reference pins anything that is specific to them) and stated here from the published multi-scale idea:
 *   scales > 1  coarse-to-fine: the pair is halved (2x2 mean of the finite samples) while the smaller side stays
 *               >= 128 px, at most scales - 1 times; the coarsest level is matched over the whole (halved) range; each
 *               finer level only admits, per pixel, the disparities within [2 min - 2, 2 max + 2] of the 3x3 coarse
 *               neighbourhood of its parent; the level is matched over the union of those ranges, which is also what a
 *               pixel without a parent estimate searches (no parent estimate anywhere: the whole halved range).  The
 *               stage dumps C / S of a multi-level call are laid out for that narrowed range;
 *   subpix = 2  candidates every half pixel: image 2 is also sampled half way between its columns (mean of the two
 *               neighbours) and census-transformed there; P1 / P2, the L-R threshold (in pixels) and the V fit apply
 *               to the half-pixel candidate grid.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define IMAX(a, b) ((a) > (b) ? (a) : (b))
#define IMIN(a, b) ((a) < (b) ? (a) : (b))
#define C_EXCLUDED 255       /* candidate outside image 2 / NaN pixel / padding: never wins */
#define CENSUS_MAX_BITS 24   /* largest Hamming distance of a valid candidate (5x5 census)     */

/* census transform on a win x win window (win in {3,5}), coordinates clamped to the image,
 * bit = neighbour < centre (row-major neighbour order, centre skipped); NaN compares false. */
void s2p_oracle_census(const float* im, int w, int h, int win, uint32_t* out)
{
    int r = win / 2;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float c = im[(size_t)y * w + x];
            uint32_t bits = 0;
            for (int dy = -r; dy <= r; dy++)
                for (int dx = -r; dx <= r; dx++) {
                    if (dx == 0 && dy == 0) continue;
                    int xx = IMIN(IMAX(x + dx, 0), w - 1), yy = IMIN(IMAX(y + dy, 0), h - 1);
                    bits = (bits << 1) | (im[(size_t)yy * w + xx] < c ? 1u : 0u);
                }
            out[(size_t)y * w + x] = bits;
        }
}

static int popc(uint32_t v) { return __builtin_popcount(v); }

/* ---- ZNCC cost (cost = 1): north_star's "census/ZNCC cost volume".  No call site of the reference selects it (both pass
 * `-t census`, s2p/block_matching.py:171,293), mgm's own `-t ncc` source is absent: UNPINNED, stated here so that the HIP
 * kernel has a bit-exact checker.  Window = the census window (win x win, coordinates clamped to the image), float32, fixed
 * operation order, no fused multiply-add (-ffp-contract=off):
 *   a'_k = a_k - mean(a)                      the reference window, centred (mean = (sum in raster order) / n)
 *   va   = sum a'_k^2 ,  vb(x2) = sum (b_k - mean(b))^2     both in raster order
 *   cov  = sum a'_k b_k                      (b raw: sum a'_k is zero up to rounding)
 *   zncc = cov / sqrtf(va vb)  if va vb > 0, else 0
 *   cost = clamp(floorf((1 - zncc) 12 + 0.5), 0, 24)        the census scale (0 .. 24), so that P1 / P2 keep their meaning
 * A NaN anywhere in either window excludes the candidate (255), as does a candidate outside image 2.  Half-pixel candidates (subpix = 2)
 * take the window, its variance and its validity from image 2 sampled half way between its columns (im2h of census_level), as the
 * census cost takes their signature from it. */
static void zncc_stats(const float* im, int w, int h, int win, int x, int y, float* centred, float* var, int* ok)
{
    const int r = win / 2, n = win * win;
    float v[25], sum = 0.0f;
    int k = 0, fin = 1;
    for (int dy = -r; dy <= r; dy++)
        for (int dx = -r; dx <= r; dx++) {
            const int xx = IMIN(IMAX(x + dx, 0), w - 1), yy = IMIN(IMAX(y + dy, 0), h - 1);
            v[k] = im[(size_t)yy * w + xx];
            if (!isfinite(v[k])) fin = 0;
            sum = sum + v[k];
            k++;
        }
    const float mean = sum / (float)n;
    float s2 = 0.0f;
    for (k = 0; k < n; k++) { const float c = v[k] - mean; if (centred) centred[k] = c; s2 = s2 + c * c; }
    *var = s2; *ok = fin;
}
static uint8_t zncc_cost(const float* ac, float va, const float* im2, int w, int h, int win, int x2, int y, float vb)
{
    const int r = win / 2;
    float cov = 0.0f;
    int k = 0;
    for (int dy = -r; dy <= r; dy++)
        for (int dx = -r; dx <= r; dx++) {
            const int xx = IMIN(IMAX(x2 + dx, 0), w - 1), yy = IMIN(IMAX(y + dy, 0), h - 1);
            cov = cov + ac[k] * im2[(size_t)yy * w + xx];
            k++;
        }
    const float den = va * vb;
    const float z = den > 0.0f ? cov / sqrtf(den) : 0.0f;
    float c = floorf((1.0f - z) * 12.0f + 0.5f);
    if (!(c >= 0.0f)) c = 0.0f;
    if (c > 24.0f) c = 24.0f;
    return (uint8_t)c;
}

/* 3x3 median over the finite values of the window (centre must be finite): element (n-1)/2 of the
 * sorted finite values; borders: window clipped to the image. */
static void median3x3_valid(const float* src, float* dst, int w, int h)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float c = src[(size_t)y * w + x];
            if (!isfinite(c)) { dst[(size_t)y * w + x] = c; continue; }
            float v[9]; int n = 0;
            for (int dy = -1; dy <= 1; dy++)
                for (int dx = -1; dx <= 1; dx++) {
                    int xx = x + dx, yy = y + dy;
                    if (xx < 0 || xx >= w || yy < 0 || yy >= h) continue;
                    float t = src[(size_t)yy * w + xx];
                    if (isfinite(t)) v[n++] = t;
                }
            for (int i = 1; i < n; i++) { float t = v[i]; int j = i - 1; while (j >= 0 && v[j] > t) { v[j + 1] = v[j]; j--; } v[j + 1] = t; }
            dst[(size_t)y * w + x] = v[(n - 1) / 2];
        }
}

static int floordiv(int a, int b) { int q = a / b; return (a % b != 0 && (a < 0)) ? q - 1 : q; }

/* One level: candidates j = 0 .. Dt - 1 stand for the disparities dmin + j / SP (SP = subpix); lo / hi (optional, per
 * pixel, in whole pixels) restrict the admissible ones. */
static int census_level(const float* im1, const float* im2, int w, int h, int dmin, int dmax,
                        const s2p_oracle_census_params* p, const int16_t* lo, const int16_t* hi,
                        float* odisp, float* oconf, uint8_t* omask, s2p_oracle_census_dump* dump)
{
    const int SP = p->subpix == 2 ? 2 : 1;
    const int Dt = SP * (dmax - dmin) + 1, D = (Dt + 15) / 16 * 16;
    const int P1 = p->P1, P2 = p->P2;
    const size_t npx = (size_t)w * h, vol = npx * D;
    uint32_t* c1 = (uint32_t*)malloc(npx * 4);
    uint32_t* c2 = (uint32_t*)malloc(npx * 4);
    uint32_t* c2h = NULL;
    float* im2h = NULL;
    s2p_oracle_census(im1, w, h, p->census_win, c1);
    s2p_oracle_census(im2, w, h, p->census_win, c2);
    if (SP == 2 || p->subpix_model == 2) {    /* image 2 half way between its columns, and its census transform */
        im2h = (float*)malloc(npx * 4);
        c2h = (uint32_t*)malloc(npx * 4);
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++)
                im2h[(size_t)y * w + x] = 0.5f * (im2[(size_t)y * w + x] + im2[(size_t)y * w + IMIN(x + 1, w - 1)]);
        s2p_oracle_census(im2h, w, h, p->census_win, c2h);
    }
    uint8_t* C = (uint8_t*)malloc(vol);
    if (p->cost == 1) {                        /* ZNCC; SP = 2: the odd candidates correlate with image 2 sampled half way between its columns (im2h) */
        float* vb = (float*)malloc(npx * 4);
        uint8_t* okb = (uint8_t*)malloc(npx);
        float* vbh = SP == 2 ? (float*)malloc(npx * 4) : NULL;
        uint8_t* okbh = SP == 2 ? (uint8_t*)malloc(npx) : NULL;
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                int ok;
                zncc_stats(im2, w, h, p->census_win, x, y, NULL, &vb[(size_t)y * w + x], &ok); okb[(size_t)y * w + x] = (uint8_t)ok;
                if (SP == 2) { zncc_stats(im2h, w, h, p->census_win, x, y, NULL, &vbh[(size_t)y * w + x], &ok); okbh[(size_t)y * w + x] = (uint8_t)ok; }
            }
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                uint8_t* c = C + ((size_t)y * w + x) * D;
                float ac[25], va; int ok1;
                zncc_stats(im1, w, h, p->census_win, x, y, ac, &va, &ok1);
                const int jlo = lo ? SP * ((int)lo[(size_t)y * w + x] - dmin) : 0;
                const int jhi = hi ? SP * ((int)hi[(size_t)y * w + x] - dmin) : Dt - 1;
                for (int i = 0; i < D; i++) {
                    const int d2 = SP * dmin + i, x2 = x + floordiv(d2, SP), ph = d2 - SP * floordiv(d2, SP);
                    const float* s2 = ph ? im2h : im2;
                    const float* v2 = ph ? vbh : vb;
                    const uint8_t* k2 = ph ? okbh : okb;
                    if (i >= Dt || i < jlo || i > jhi || !ok1 || x2 < 0 || x2 >= w || !k2[(size_t)y * w + x2]) c[i] = C_EXCLUDED;
                    else c[i] = zncc_cost(ac, va, s2, w, h, p->census_win, x2, y, v2[(size_t)y * w + x2]);
                }
            }
        free(vb); free(okb); free(vbh); free(okbh);
    } else
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            uint8_t* c = C + ((size_t)y * w + x) * D;
            int ok1 = isfinite(im1[(size_t)y * w + x]);
            const int jlo = lo ? SP * ((int)lo[(size_t)y * w + x] - dmin) : 0;
            const int jhi = hi ? SP * ((int)hi[(size_t)y * w + x] - dmin) : Dt - 1;
            for (int i = 0; i < D; i++) {
                const int d2 = SP * dmin + i, x2 = x + floordiv(d2, SP), ph = d2 - SP * floordiv(d2, SP);
                const float* s2 = ph ? im2h : im2;
                const uint32_t* g2 = ph ? c2h : c2;
                if (i >= Dt || i < jlo || i > jhi || !ok1 || x2 < 0 || x2 >= w || !isfinite(s2[(size_t)y * w + x2])) c[i] = C_EXCLUDED;
                else c[i] = (uint8_t)popc(c1[(size_t)y * w + x] ^ g2[(size_t)y * w + x2]);
            }
        }
    if (SP == 2 && p->subpix_model == 1 && p->cost == 0)     /* experiment: a half-pixel candidate costs the mean of its whole-pixel neighbours */
        for (size_t i0 = 0; i0 < npx; i0++) {
            uint8_t* c = C + i0 * D;
            for (int i = 1; i + 1 < Dt; i += 2)
                if (c[i] != C_EXCLUDED) c[i] = (c[i - 1] == C_EXCLUDED || c[i + 1] == C_EXCLUDED) ? C_EXCLUDED : (uint8_t)((c[i - 1] + c[i + 1] + 1) >> 1);
        }
    if (dump) { dump->dmin0 = dmin; dump->D0 = D; }          /* the range the volumes are laid out for (narrowed at the finest level of a multi-scale call) */
    if (dump && dump->C) memcpy(dump->C, C, vol);

    /* ND independent path sets (the first ND entries of the direction table: 4 = the axis directions, mgm's -O 4);
     * S = sum_r L_r (uint16) */
    const int ND = p->nb_dir;
    uint16_t* S = (uint16_t*)calloc(vol, 2);
    uint16_t* Lbest = (uint16_t*)calloc(npx * 16, 2);      /* per-direction argmin, for the confidence */
    /* 0..3 axis, 4..7 diagonal, 8..15 the knight's moves of nb_dir = 16 (cfg['mgm_nb_directions'], s2p/config.py:149: the binary's source
     * is absent, so which 16 and in which order is an ASSUMPTION -- the usual 16-path set of semi-global matching; UNPINNED).  Only the MGM
     * recursion takes them (s2p_oracle_census_sgm refuses nb_dir = 16 with recursion = 0): r_perp = (-dy, dx) as for the other 8 */
    static const int DX[16] = {1, -1, 0, 0, 1, -1, -1, 1, 2, -1, -2, 1, 1, -2, -1, 2}, DY[16] = {0, 0, 1, -1, 1, 1, -1, -1, 1, 2, -1, -2, 2, 1, -2, -1};
    int* Lp = (int*)malloc((size_t)(D + 2) * sizeof(int));
    int* Ln = (int*)malloc((size_t)(D + 2) * sizeof(int));
    if (p->recursion == 0)
    for (int r = 0; r < ND; r++) {
        int dx = DX[r], dy = DY[r];
        for (int sy = 0; sy < h; sy++)
            for (int sx = 0; sx < w; sx++) {
                /* path starts: pixels whose predecessor is outside */
                int px = sx - dx, py = sy - dy;
                if (px >= 0 && px < w && py >= 0 && py < h) continue;
                int x = sx, y = sy;
                for (int i = 0; i < D; i++) Lp[i + 1] = 0;    /* virtual predecessor: L = 0 => first L = C */
                int minLp = 0;
                while (x >= 0 && x < w && y >= 0 && y < h) {
                    const uint8_t* c = C + ((size_t)y * w + x) * D;
                    uint16_t* s = S + ((size_t)y * w + x) * D;
                    Lp[0] = Lp[D + 1] = 1 << 20;
                    int mn = 1 << 30, arg = 0;
                    for (int i = 0; i < D; i++) {
                        int m = IMIN(IMIN(Lp[i + 1], IMIN(Lp[i], Lp[i + 2]) + P1), minLp + P2);
                        int L = c[i] + m - minLp;
                        Ln[i + 1] = L;
                        if (L < mn) { mn = L; arg = i; }
                        s[i] = (uint16_t)(s[i] + L);
                    }
                    Lbest[((size_t)y * w + x) * 16 + r] = (uint16_t)arg;
                    { int* t = Lp; Lp = Ln; Ln = t; }
                    minLp = mn;
                    x += dx; y += dy;
                }
            }
    }
    else {
        /* MGM (Facciolo, de Franchis, Meinhardt, BMVC 2015, eq. 7), integer form: for every direction r the cost of a
         * pixel collects the messages of TWO predecessors, p - r and p - r_perp (r_perp = r rotated by 90 degrees),
         *     msg_q(d) = min(L(q,d), L(q,d-1) + P1, L(q,d+1) + P1, min_k L(q,k) + P2) - min_k L(q,k)   (0 outside the image)
         *     L_r(p,d) = C(p,d) + (msg_{p-r}(d) + msg_{p-r_perp}(d) + 1) >> 1
         * (the mean of the two messages, rounded half up so that the pipeline stays integral and L - C in [0, P2]).
         * Pixels are visited in the order of x (dx + ex) + y (dy + ey), for which both predecessors come first:
         * anti-diagonals for the 4 axis directions, rows / columns for the 4 diagonal ones, lines of slope 1/3 or 3 for a
         * knight's move (both predecessors sit dx^2 + dy^2 = 5 fronts back). */
        uint16_t* L = (uint16_t*)malloc(vol * 2);
        int* mnL = (int*)malloc(npx * sizeof(int));
        for (int r = 0; r < ND; r++) {
            const int dx = DX[r], dy = DY[r], ex = -dy, ey = dx;
            const int kx = dx + ex, ky = dy + ey;
            int kmin = IMIN(0, kx * (w - 1)) + IMIN(0, ky * (h - 1)), kmax = IMAX(0, kx * (w - 1)) + IMAX(0, ky * (h - 1));
            for (int key = kmin; key <= kmax; key++)
                for (int y = 0; y < h; y++) {
                    /* the pixels of row y on this front: all of them (kx == 0, ky y == key) or the single x = (key - ky y) / kx */
                    int xa = 0, xb = w - 1;
                    if (kx == 0) { if (ky * y != key) continue; }
                    else {
                        const int num = key - ky * y;
                        if (num % kx != 0) continue;
                        xa = xb = num / kx;
                        if (xa < 0 || xa >= w) continue;
                    }
                    for (int x = xa; x <= xb; x++) {
                        const size_t i0 = (size_t)y * w + x;
                        const uint8_t* c = C + i0 * D;
                        uint16_t* l = L + i0 * D;
                        /* recursion = 2 (TSGM = 3 of the 'mgm' call site, s2p/block_matching.py:158; the binary's source is absent, so what
                         * "3" adds is an ASSUMPTION: the third predecessor of the quadrant, p - r - r_perp): mean of three messages */
                        const int NQ = p->recursion == 2 ? 3 : 2;
                        const int qx[3] = {x - dx, x - ex, x - dx - ex}, qy[3] = {y - dy, y - ey, y - dy - ey};
                        for (int i = 0; i < D; i++) Ln[i + 1] = 0;                 /* sum of the messages */
                        for (int n = 0; n < NQ; n++) {
                            if (qx[n] < 0 || qx[n] >= w || qy[n] < 0 || qy[n] >= h) continue;
                            const size_t j = (size_t)qy[n] * w + qx[n];
                            const uint16_t* lq = L + j * D;
                            const int m0 = mnL[j];
                            Lp[0] = Lp[D + 1] = 1 << 20;
                            for (int i = 0; i < D; i++) Lp[i + 1] = lq[i];
                            for (int i = 0; i < D; i++)
                                Ln[i + 1] += IMIN(IMIN(Lp[i + 1], IMIN(Lp[i], Lp[i + 2]) + P1), m0 + P2) - m0;
                        }
                        int mn = 1 << 30, arg = 0;
                        uint16_t* s = S + i0 * D;
                        for (int i = 0; i < D; i++) {
                            int Lv = c[i] + (NQ == 2 ? (Ln[i + 1] + 1) >> 1 : (2 * Ln[i + 1] + 3) / 6);   /* mean, rounded half up */
                            l[i] = (uint16_t)Lv;
                            if (Lv < mn) { mn = Lv; arg = i; }
                            s[i] = (uint16_t)(s[i] + Lv);
                        }
                        mnL[i0] = mn;
                        Lbest[i0 * 16 + r] = (uint16_t)arg;
                    }
                }
        }
        free(L); free(mnL);
    }
    const int fixo = p->fix_overcount ? ND - 1 : 0;
    if (fixo) for (size_t i = 0; i < vol; i++) S[i] = (uint16_t)(S[i] - fixo * IMIN((int)C[i], CENSUS_MAX_BITS));
    const int s_excluded = ND * C_EXCLUDED - fixo * CENSUS_MAX_BITS;     /* every excluded candidate is >= this */
    if (dump && dump->S) memcpy(dump->S, S, vol * 2);

    /* WTA (first minimum), right view from the same S (min over the diagonal), vfit, L-R test */
    float* d0 = (float*)malloc(npx * 4);
    int* bestL = (int*)malloc(npx * sizeof(int));
    const int tau = (int)floorf(p->lr_tau * (float)SP);     /* in candidates */
    for (int y = 0; y < h; y++) {
        /* right view: every candidate competes for the (half-)pixel of image 2 it points at, slot = SP x + j */
        const int nslot = SP * w + D;
        uint32_t* rkey = (uint32_t*)malloc((size_t)nslot * 4);
        for (int x = 0; x < nslot; x++) rkey[x] = 0xffffffffu;
        for (int x = 0; x < w; x++) {
            const uint16_t* s = S + ((size_t)y * w + x) * D;
            int mn = 1 << 30, b = 0;
            for (int i = 0; i < D; i++) if (s[i] < mn) { mn = s[i]; b = i; }
            bestL[(size_t)y * w + x] = (mn >= s_excluded) ? -1 : b;
            for (int i = 0; i < Dt; i++) {
                uint32_t k = ((uint32_t)s[i] << 16) | (uint32_t)i;
                if (k < rkey[SP * x + i]) rkey[SP * x + i] = k;
            }
        }
        for (int x = 0; x < w; x++) {
            size_t i0 = (size_t)y * w + x;
            int b = bestL[i0];
            float out = NAN;
            if (b >= 0) {
                const uint16_t* s = S + i0 * D;
                int ok = 1;
                if (p->lr_check) {
                    int ir = (int)(rkey[SP * x + b] & 0xffffu);   /* the slot the winner points at (inside image 2, else it were excluded) */
                    if (abs(ir - b) > tau) ok = 0;
                }
                if (p->mindiff > 0) {     /* MINDIFF (cfg['mgm_mindiff_control'], s2p/config.py:158-160; the binary's source is absent: UNPINNED).  Adopted */
                    int s2 = 1 << 30;     /* statement: the winner must beat every candidate that is not its neighbour by at least `mindiff` units of S */
                    for (int i = 0; i < Dt; i++) if (abs(i - b) > 1 && s[i] < s2) s2 = s[i];
                    if (s2 != (1 << 30) && s2 - (int)s[b] < p->mindiff) ok = 0;
                }
                if (ok) {
                    float off = 0.0f;
                    int refined = 0;
                    if (p->subpix_model == 2 && SP == 1 && b > 0 && b < Dt - 1) {   /* experiment: refine the winner on the half-pixel grid */
                        const uint8_t* c = C + i0 * D;
                        const int x2 = x + dmin + b;
                        if (c[b - 1] != C_EXCLUDED && c[b + 1] != C_EXCLUDED && x2 - 1 >= 0 && x2 + 1 < w &&
                            isfinite(im2h[(size_t)y * w + x2 - 1]) && isfinite(im2h[(size_t)y * w + x2])) {
                            const int k = p->fix_overcount ? 1 : ND;
                            const int cm = popc(c1[i0] ^ c2h[(size_t)y * w + x2 - 1]), cp = popc(c1[i0] ^ c2h[(size_t)y * w + x2]);
                            const int hm = IMAX(0, ((s[b] + s[b - 1] + 1) >> 1) + k * (cm - ((c[b] + c[b - 1] + 1) >> 1)));
                            const int hp = IMAX(0, ((s[b] + s[b + 1] + 1) >> 1) + k * (cp - ((c[b] + c[b + 1] + 1) >> 1)));
                            int ch = 0, sm, s0, sp;                           /* chosen half-pixel offset: -1, 0, +1 */
                            if (hm < s[b] && hm <= hp) ch = -1; else if (hp < s[b]) ch = 1;
                            if (ch == 0) { sm = hm; s0 = s[b]; sp = hp; }
                            else if (ch < 0) { sm = s[b - 1]; s0 = hm; sp = s[b]; }
                            else { sm = s[b]; s0 = hp; sp = s[b + 1]; }
                            float o2 = 0.0f;
                            const int den = IMAX(sm - s0, sp - s0);
                            if (den > 0) o2 = 0.5f * ((float)(sm - sp) / (float)den);
                            off = 0.5f * ((float)ch + o2);
                            refined = 1;
                        }
                    }
                    if (!refined && b > 0 && b < Dt - 1) {    /* vfit: V-shaped interpolation */
                        int sm = s[b - 1], s0 = s[b], sp = s[b + 1];
                        int den = IMAX(sm - s0, sp - s0);
                        if (den > 0) off = 0.5f * ((float)(sm - sp) / (float)den);
                    }
                    out = SP == 1 ? (float)(dmin + b) + off : 0.5f * ((float)(2 * dmin + b) + off);
                }
            }
            d0[i0] = out;
        }
        free(rkey);
    }
    if (dump && dump->disp_raw) memcpy(dump->disp_raw, d0, npx * 4);

    float* d1 = (float*)malloc(npx * 4);
    if (p->median) median3x3_valid(d0, d1, w, h); else memcpy(d1, d0, npx * 4);
    if (dump && dump->disp_med) memcpy(dump->disp_med, d1, npx * 4);

    if (p->remove_small_cc > 0) {            /* components of valid pixels, |delta d| <= 1 px, size < N removed */
        int16_t* q = (int16_t*)malloc(npx * 2);
        const int INV = -32768;
        for (size_t i = 0; i < npx; i++) q[i] = isfinite(d1[i]) ? (int16_t)lrintf(d1[i] * 16.0f) : (int16_t)INV;
        s2p_oracle_speckle_s16(q, w, h, INV, p->remove_small_cc - 1, 16);
        for (size_t i = 0; i < npx; i++) if (q[i] == INV) d1[i] = NAN;
        free(q);
    }
    memcpy(odisp, d1, npx * 4);
    if (oconf)
        for (size_t i = 0; i < npx; i++) {
            int b = bestL[i], n = 0;
            if (b >= 0) for (int r = 0; r < ND; r++) n += abs((int)Lbest[i * 16 + r] - b) <= 1;
            oconf[i] = isfinite(d1[i]) ? (float)n / (float)ND : NAN;
        }
    if (omask) s2p_oracle_rejection_mask(d1, im1, im2, w, h, omask);
    free(c1); free(c2); free(c2h); free(im2h); free(C); free(S); free(Lbest); free(Lp); free(Ln); free(d0); free(d1); free(bestL);
    return 0;
}

/* ---- multi-scale driver ------------------------------------------------------------------------------------------ */
#define MS_MIN_DIM 128      /* a level is only added while the smaller side of the halved pair stays >= this */
#define MS_MARGIN 2         /* pixels added on both sides of the range a parent neighbourhood suggests       */

/* 2x2 mean of the finite samples (float32, summed in the order (0,0) (1,0) (0,1) (1,1), then one division); NaN if none */
void s2p_oracle_down2(const float* src, int w, int h, float* dst)
{
    const int w2 = (w + 1) / 2, h2 = (h + 1) / 2;
    for (int y = 0; y < h2; y++)
        for (int x = 0; x < w2; x++) {
            float sum = 0.0f; int n = 0;
            for (int dy = 0; dy < 2; dy++)
                for (int dx = 0; dx < 2; dx++) {
                    const int xx = 2 * x + dx, yy = 2 * y + dy;
                    if (xx >= w || yy >= h) continue;
                    const float v = src[(size_t)yy * w + xx];
                    if (isfinite(v)) { sum = sum + v; n++; }
                }
            dst[(size_t)y * w2 + x] = n ? sum / (float)n : NAN;
        }
}

/* admissible range of every pixel of a (w, h) level from the disparity map of its parent level ((w+1)/2, (h+1)/2) */
void s2p_oracle_range_from_coarse(const float* dc, int w, int h, int dmin, int dmax, int16_t* lo, int16_t* hi)
{
    const int wc = (w + 1) / 2, hc = (h + 1) / 2;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int cx = x >> 1, cy = y >> 1;
            int l = dmin, u = dmax;
            if (isfinite(dc[(size_t)cy * wc + cx])) {
                float mn = INFINITY, mx = -INFINITY;
                for (int dy = -1; dy <= 1; dy++)
                    for (int dx = -1; dx <= 1; dx++) {
                        const int xx = cx + dx, yy = cy + dy;
                        if (xx < 0 || xx >= wc || yy < 0 || yy >= hc) continue;
                        const float v = dc[(size_t)yy * wc + xx];
                        if (isfinite(v)) { mn = fminf(mn, v); mx = fmaxf(mx, v); }
                    }
                l = (int)floorf(2.0f * mn) - MS_MARGIN;
                u = (int)ceilf(2.0f * mx) + MS_MARGIN;
                l = IMIN(IMAX(l, dmin), dmax);
                u = IMIN(IMAX(u, dmin), dmax);
            }
            lo[(size_t)y * w + x] = (int16_t)l;
            hi[(size_t)y * w + x] = (int16_t)u;
        }
}

int s2p_oracle_census_levels(int w, int h, int scales)
{
    int L = 1;
    while (L < scales && IMIN((w + 1) / 2, (h + 1) / 2) >= MS_MIN_DIM) { L++; w = (w + 1) / 2; h = (h + 1) / 2; }
    return L;
}

int s2p_oracle_census_sgm(const float* im1, const float* im2, int w, int h, int dmin, int dmax,
                          const s2p_oracle_census_params* p, float* odisp, float* oconf, uint8_t* omask,
                          s2p_oracle_census_dump* dump)
{
    if (dmax < dmin) return 1;
    if (!(p->census_win == 3 || p->census_win == 5) || (p->nb_dir != 8 && p->nb_dir != 4 && !(p->nb_dir == 16 && p->recursion >= 1))) return 4;
    if (!(p->subpix == 0 || p->subpix == 1 || p->subpix == 2)) return 4;
    if (p->cost != 0 && p->cost != 1) return 4;                             /* 0 = census / Hamming, 1 = ZNCC */
    const int L = s2p_oracle_census_levels(w, h, p->scales);
    if (L <= 1) return census_level(im1, im2, w, h, dmin, dmax, p, NULL, NULL, odisp, oconf, omask, dump);
    float* a[16]; float* b[16]; int ws[16], hs[16], lo_[16], hi_[16];
    a[0] = (float*)im1; b[0] = (float*)im2; ws[0] = w; hs[0] = h; lo_[0] = dmin; hi_[0] = dmax;
    for (int k = 1; k < L; k++) {
        ws[k] = (ws[k - 1] + 1) / 2; hs[k] = (hs[k - 1] + 1) / 2;
        lo_[k] = floordiv(lo_[k - 1], 2); hi_[k] = -floordiv(-hi_[k - 1], 2);
        a[k] = (float*)malloc((size_t)ws[k] * hs[k] * 4); b[k] = (float*)malloc((size_t)ws[k] * hs[k] * 4);
        s2p_oracle_down2(a[k - 1], ws[k - 1], hs[k - 1], a[k]);
        s2p_oracle_down2(b[k - 1], ws[k - 1], hs[k - 1], b[k]);
    }
    float* dc = NULL;                         /* disparity of the level below the current one */
    int rc = 0;
    for (int k = L - 1; k >= 0 && !rc; k--) {
        const size_t n = (size_t)ws[k] * hs[k];
        int16_t* lo = NULL; int16_t* hi = NULL;
        if (dc) {
            lo = (int16_t*)malloc(n * 2); hi = (int16_t*)malloc(n * 2);
            s2p_oracle_range_from_coarse(dc, ws[k], hs[k], lo_[k], hi_[k], lo, hi);
            /* the level is matched over the union of its pixels' admissible ranges (a configured search range is usually
             * several times what the parent level found: the volumes shrink with it) */
            int gmin = hi_[k] + 1, gmax = lo_[k] - 1;
            const int wc = (ws[k] + 1) / 2;
            for (int y = 0; y < hs[k]; y++)
                for (int x = 0; x < ws[k]; x++)
                    if (isfinite(dc[(size_t)(y >> 1) * wc + (x >> 1)])) {
                        const size_t i = (size_t)y * ws[k] + x;
                        gmin = IMIN(gmin, lo[i]); gmax = IMAX(gmax, hi[i]);
                    }
            if (gmin <= gmax) {                  /* pixels without a parent estimate search what the level's other pixels search */
                for (int y = 0; y < hs[k]; y++)
                    for (int x = 0; x < ws[k]; x++)
                        if (!isfinite(dc[(size_t)(y >> 1) * wc + (x >> 1)])) {
                            const size_t i = (size_t)y * ws[k] + x;
                            lo[i] = (int16_t)gmin; hi[i] = (int16_t)gmax;
                        }
                lo_[k] = gmin; hi_[k] = gmax;
            }
        }
        if (k == 0) rc = census_level(a[0], b[0], w, h, lo_[0], hi_[0], p, lo, hi, odisp, oconf, omask, dump);
        else {
            float* d = (float*)malloc(n * 4);
            s2p_oracle_census_params pc = *p;              /* mgm_leftright_control = 2: the L-R test at the last scale only (s2p/config.py:155-157) */
            if (pc.lr_check == 2) pc.lr_check = 0;
            rc = census_level(a[k], b[k], ws[k], hs[k], lo_[k], hi_[k], &pc, lo, hi, d, NULL, NULL, NULL);
            free(dc); dc = d;
        }
        free(lo); free(hi);
    }
    free(dc);
    for (int k = 1; k < L; k++) { free(a[k]); free(b[k]); }
    return rc;
}


/* ---- s2p/masking.py:87-97 erosion: `morsi disk%d erosion msk out` when radius >= 2 (imscript's
 * morsi: source absent from the reference tree -> UNPINNED).  Structuring element adopted: the
 * integer offsets (i, j) with hypot(i, j) < radius (radius 2 -> the 3x3 square); erosion = minimum
 * over the element; offsets that fall outside the image are ignored. */
void s2p_oracle_erode_disk(const uint8_t* msk, int w, int h, int radius, uint8_t* out)
{
    int R = radius + 1;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int v = msk[(size_t)y * w + x];
            for (int j = -R; j <= R && v; j++)
                for (int i = -R; i <= R; i++) {
                    if (!(hypot((double)i, (double)j) < (double)radius)) continue;
                    int xx = x + i, yy = y + j;
                    if (xx < 0 || xx >= w || yy < 0 || yy >= h) continue;
                    if (!msk[(size_t)yy * w + xx]) { v = 0; break; }
                }
            out[(size_t)y * w + x] = (uint8_t)v;
        }
}
