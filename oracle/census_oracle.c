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

int s2p_oracle_census_sgm(const float* im1, const float* im2, int w, int h, int dmin, int dmax,
                          const s2p_oracle_census_params* p, float* odisp, float* oconf, uint8_t* omask,
                          s2p_oracle_census_dump* dump)
{
    if (dmax < dmin) return 1;
    if (!(p->census_win == 3 || p->census_win == 5) || p->nb_dir != 8) return 4;
    const int Dt = dmax - dmin + 1, D = (Dt + 15) / 16 * 16;
    const int P1 = p->P1, P2 = p->P2;
    const size_t npx = (size_t)w * h, vol = npx * D;
    uint32_t* c1 = (uint32_t*)malloc(npx * 4);
    uint32_t* c2 = (uint32_t*)malloc(npx * 4);
    s2p_oracle_census(im1, w, h, p->census_win, c1);
    s2p_oracle_census(im2, w, h, p->census_win, c2);
    uint8_t* C = (uint8_t*)malloc(vol);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            uint8_t* c = C + ((size_t)y * w + x) * D;
            int ok1 = isfinite(im1[(size_t)y * w + x]);
            for (int i = 0; i < D; i++) {
                int x2 = x + dmin + i;
                if (i >= Dt || !ok1 || x2 < 0 || x2 >= w || !isfinite(im2[(size_t)y * w + x2])) c[i] = C_EXCLUDED;
                else c[i] = (uint8_t)popc(c1[(size_t)y * w + x] ^ c2[(size_t)y * w + x2]);
            }
        }
    if (dump && dump->C) memcpy(dump->C, C, vol);

    /* 8 independent path sets; S = sum_r L_r (uint16) */
    uint16_t* S = (uint16_t*)calloc(vol, 2);
    uint16_t* Lbest = (uint16_t*)calloc(npx * 8, 2);       /* per-direction argmin, for the confidence */
    static const int DX[8] = {1, -1, 0, 0, 1, -1, -1, 1}, DY[8] = {0, 0, 1, -1, 1, 1, -1, -1};
    int* Lp = (int*)malloc((size_t)(D + 2) * sizeof(int));
    int* Ln = (int*)malloc((size_t)(D + 2) * sizeof(int));
    if (p->recursion == 0)
    for (int r = 0; r < 8; r++) {
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
                    Lbest[((size_t)y * w + x) * 8 + r] = (uint16_t)arg;
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
         * anti-diagonals for the 4 axis directions, rows / columns for the 4 diagonal ones. */
        uint16_t* L = (uint16_t*)malloc(vol * 2);
        int* mnL = (int*)malloc(npx * sizeof(int));
        for (int r = 0; r < 8; r++) {
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
                        const int qx[2] = {x - dx, x - ex}, qy[2] = {y - dy, y - ey};
                        for (int i = 0; i < D; i++) Ln[i + 1] = 0;                 /* sum of the two messages */
                        for (int n = 0; n < 2; n++) {
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
                            int Lv = c[i] + ((Ln[i + 1] + 1) >> 1);
                            l[i] = (uint16_t)Lv;
                            if (Lv < mn) { mn = Lv; arg = i; }
                            s[i] = (uint16_t)(s[i] + Lv);
                        }
                        mnL[i0] = mn;
                        Lbest[i0 * 8 + r] = (uint16_t)arg;
                    }
                }
        }
        free(L); free(mnL);
    }
    const int fixo = p->fix_overcount ? 7 : 0;
    if (fixo) for (size_t i = 0; i < vol; i++) S[i] = (uint16_t)(S[i] - fixo * IMIN((int)C[i], CENSUS_MAX_BITS));
    const int s_excluded = 8 * C_EXCLUDED - fixo * CENSUS_MAX_BITS;     /* every excluded candidate is >= this */
    if (dump && dump->S) memcpy(dump->S, S, vol * 2);

    /* WTA (first minimum), right view from the same S (min over the diagonal), vfit, L-R test */
    float* d0 = (float*)malloc(npx * 4);
    int* bestL = (int*)malloc(npx * sizeof(int));
    const int tau = (int)floorf(p->lr_tau);
    for (int y = 0; y < h; y++) {
        uint32_t* rkey = (uint32_t*)malloc((size_t)w * 4);
        for (int x = 0; x < w; x++) rkey[x] = 0xffffffffu;
        for (int x = 0; x < w; x++) {
            const uint16_t* s = S + ((size_t)y * w + x) * D;
            int mn = 1 << 30, b = 0;
            for (int i = 0; i < D; i++) if (s[i] < mn) { mn = s[i]; b = i; }
            bestL[(size_t)y * w + x] = (mn >= s_excluded) ? -1 : b;
            for (int i = 0; i < Dt; i++) {
                int x2 = x + dmin + i;
                if (x2 < 0 || x2 >= w) continue;
                uint32_t k = ((uint32_t)s[i] << 16) | (uint32_t)i;
                if (k < rkey[x2]) rkey[x2] = k;
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
                    int x2 = x + dmin + b;                    /* inside the image, else it were excluded */
                    int ir = (int)(rkey[x2] & 0xffffu);
                    if (abs(ir - b) > tau) ok = 0;
                }
                if (ok) {
                    float off = 0.0f;
                    if (b > 0 && b < Dt - 1) {                /* vfit: V-shaped interpolation */
                        int sm = s[b - 1], s0 = s[b], sp = s[b + 1];
                        int den = IMAX(sm - s0, sp - s0);
                        if (den > 0) off = 0.5f * ((float)(sm - sp) / (float)den);
                    }
                    out = (float)(dmin + b) + off;
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
            if (b >= 0) for (int r = 0; r < 8; r++) n += abs((int)Lbest[i * 8 + r] - b) <= 1;
            oconf[i] = isfinite(d1[i]) ? (float)n / 8.0f : NAN;
        }
    if (omask) s2p_oracle_rejection_mask(d1, im1, im2, w, h, omask);
    free(c1); free(c2); free(C); free(S); free(Lbest); free(Lp); free(Ln); free(d0); free(d1); free(bestL);
    return 0;
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
