/* oracle/sgbm_oracle.c -- TEST INFRASTRUCTURE (see oracle.h).
 *
 * CPU restatement of the reference `sgbm` matcher, the only matcher whose arithmetic is present
 * under /root/reference (vendored OpenCV-2.4 StereoSGBM driven by 3rdparty/sgbm/sgbm.cpp).
 * Integer arithmetic throughout => the contract is BIT-EXACTNESS against the reference, pinned by
 * tests/golden/sgbm_*.npz (generated from oracle/_ref/libsgbm_ref.so = the real reference sources)
 * and, in this container, directly against that library on random inputs.
 *
 * Definition adopted for the reference's uninitialised reads:
 * "uninitialised == 0" (what fresh mmap pages give the real binary on realistic tile sizes;
 * golden vectors are generated with a zero-filling malloc inside the reference build, ref_harness.cpp).
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>

#define IMAX(a, b) ((a) > (b) ? (a) : (b))
#define IMIN(a, b) ((a) < (b) ? (a) : (b))

/* ---- 3rdparty/sgbm/sgbm.cpp:23-42 (compare_floats, get_rminmax) ---- */
static int cmp_float(const void* a, const void* b)
{
    float x = *(const float*)a, y = *(const float*)b;
    return (x > y) - (x < y);
}

void s2p_oracle_rminmax(const float* x, size_t n, float* rmin, float* rmax)
{
    float* t = (float*)malloc((n ? n : 1) * sizeof(float));
    size_t N = 0;
    for (size_t i = 0; i < n; i++)
        if (!isnan(x[i])) t[N++] = x[i];
    if (N == 0) { *rmin = 0; *rmax = 0; free(t); return; }   /* reference: undefined behaviour */
    qsort(t, N, sizeof(float), cmp_float);
    size_t rb = N / 200;                                    /* 0.5 % and 99.5 % quantiles */
    *rmin = t[rb];
    *rmax = t[N - 1 - rb];
    free(t);
}

/* ---- sgbm.cpp:44-71 (qauto / qeasy): y = clamp(floor(255*(g-rmin)/(rmax-rmin)), 0, 255) in float32.
 * NaN input: every comparison is false and the float->uint8 conversion of NaN yields 0 on x86-64
 * (cvttss2si -> 0x80000000 -> low byte 0); we define it as 0. */
void s2p_oracle_quantize(const float* x, size_t n, float rmin, float rmax, uint8_t* y)
{
    for (size_t i = 0; i < n; i++) {
        float g = x[i];
        g = floorf(255 * (g - rmin) / (rmax - rmin));
        if (g < 0) g = 0;
        if (g > 255) g = 255;
        y[i] = isnan(g) ? 0 : (uint8_t)g;
    }
}

/* ---- smooth.cpp:246-270: 3x3 median, replicate borders (the sorting network computes the exact
 * median, so any exact median-of-9 is identical). */
static inline void srt(int* a, int* b) { if (*a > *b) { int t = *a; *a = *b; *b = t; } }
void s2p_oracle_median3x3_s16(const int16_t* src, int16_t* dst, int w, int h)
{
    for (int i = 0; i < h; i++) {
        const int16_t* r0 = src + (size_t)IMAX(i - 1, 0) * w;
        const int16_t* r1 = src + (size_t)i * w;
        const int16_t* r2 = src + (size_t)IMIN(i + 1, h - 1) * w;
        for (int j = 0; j < w; j++) {
            int j0 = j >= 1 ? j - 1 : j, j2 = j < w - 1 ? j + 1 : j;
            int p0 = r0[j0], p1 = r0[j], p2 = r0[j2];
            int p3 = r1[j0], p4 = r1[j], p5 = r1[j2];
            int p6 = r2[j0], p7 = r2[j], p8 = r2[j2];
            srt(&p1, &p2); srt(&p4, &p5); srt(&p7, &p8); srt(&p0, &p1);
            srt(&p3, &p4); srt(&p6, &p7); srt(&p1, &p2); srt(&p4, &p5);
            srt(&p7, &p8); srt(&p0, &p3); srt(&p5, &p8); srt(&p4, &p7);
            srt(&p3, &p6); srt(&p1, &p4); srt(&p2, &p5); srt(&p4, &p7);
            srt(&p4, &p2); srt(&p6, &p4); srt(&p4, &p2);
            dst[(size_t)i * w + j] = (int16_t)p4;
        }
    }
}

/* ---- stereosgbm.cpp:872-967 filterSpecklesImpl<short>: 4-connected flood fill over pixels
 * != newVal whose neighbour difference is <= maxDiff; regions of <= maxSize pixels -> newVal. */
void s2p_oracle_speckle_s16(int16_t* img, int w, int h, int newVal, int maxSize, int maxDiff)
{
    size_t np = (size_t)w * h;
    int* labels = (int*)calloc(np, sizeof(int));
    int* stack = (int*)malloc(np * sizeof(int));
    uint8_t* rtype = (uint8_t*)calloc(np + 1, 1);
    int cur = 0;
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
            size_t idx = (size_t)i * w + j;
            if (img[idx] == newVal) continue;
            if (labels[idx]) { if (rtype[labels[idx]]) img[idx] = (int16_t)newVal; continue; }
            int sp = 0, count = 0;
            int p = (int)idx;
            cur++;
            labels[idx] = cur;
            for (;;) {
                count++;
                int px = p % w, py = p / w, dp = img[p];
                if (px < w - 1 && !labels[p + 1] && img[p + 1] != newVal && abs(dp - img[p + 1]) <= maxDiff) { labels[p + 1] = cur; stack[sp++] = p + 1; }
                if (px > 0 && !labels[p - 1] && img[p - 1] != newVal && abs(dp - img[p - 1]) <= maxDiff) { labels[p - 1] = cur; stack[sp++] = p - 1; }
                if (py < h - 1 && !labels[p + w] && img[p + w] != newVal && abs(dp - img[p + w]) <= maxDiff) { labels[p + w] = cur; stack[sp++] = p + w; }
                if (py > 0 && !labels[p - w] && img[p - w] != newVal && abs(dp - img[p - w]) <= maxDiff) { labels[p - w] = cur; stack[sp++] = p - w; }
                if (sp == 0) break;
                p = stack[--sp];
            }
            if (count <= maxSize) { rtype[cur] = 1; img[idx] = (int16_t)newVal; } else rtype[cur] = 0;
        }
    free(labels); free(stack); free(rtype);
}

/* ---- stereosgbm.cpp:115-280 calcPixelCostBT, single channel (cn == 1), scalar path.
 * `flat` reproduces the reference's flat row scratch `tempBuf` (width*16 bytes, :375,390):
 *   [ v0 (width2) | v1 (width2) | prow1: prefiltered, raw (2*width) | prow2: prefiltered, raw (2*width, MIRRORED) | 0... ]
 * Out-of-range disparities index past a row into whatever follows it in that scratch (A.3); a
 * GUARD prefix of zeros stands for the bytes before tempBuf (tail of disp2ptr, still zero in pass 1). */
#define GUARD 4096
typedef struct {
    int width, height, minD, maxD, D, minX1, maxX1, width1, minX2, maxX2, width2;
    const uint8_t *img1, *img2;   /* canvases, width*height */
    uint8_t* flat_alloc;          /* GUARD + width*16 */
    uint8_t clip[256 + 1024 * 2];
} bt_ctx;

static void calc_pixel_cost_bt(bt_ctx* c, int y, int16_t* cost /* width1*D */)
{
    const int width = c->width, minD = c->minD, maxD = c->maxD, D = c->D;
    const int minX1 = c->minX1, maxX1 = c->maxX1, minX2 = c->minX2, maxX2 = c->maxX2, width2 = c->width2;
    const uint8_t* tab = c->clip + 1024;                           /* :127 tab += tabOfs */
    const uint8_t* row1 = c->img1 + (size_t)y * width;
    const uint8_t* row2 = c->img2 + (size_t)y * width;
    uint8_t* buffer = c->flat_alloc + GUARD;
    uint8_t* prow1 = buffer + width2 * 2;                           /* :125 */
    uint8_t* prow2 = prow1 + width * 2;
    for (int ch = 0; ch < 2; ch++)                                  /* :129-133 */
        prow1[width * ch] = prow1[width * ch + width - 1] =
        prow2[width * ch] = prow2[width * ch + width - 1] = tab[0];
    int n1 = y > 0 ? -width : 0, s1 = y < c->height - 1 ? width : 0;   /* :135-136 */
    for (int x = 1; x < width - 1; x++) {                           /* :140-147 */
        prow1[x] = tab[(row1[x + 1] - row1[x - 1]) * 2 + row1[x + n1 + 1] - row1[x + n1 - 1] + row1[x + s1 + 1] - row1[x + s1 - 1]];
        prow2[width - 1 - x] = tab[(row2[x + 1] - row2[x - 1]) * 2 + row2[x + n1 + 1] - row2[x + n1 - 1] + row2[x + s1 + 1] - row2[x + s1 - 1]];
        prow1[x + width] = row1[x];
        prow2[width - 1 - x + width] = row2[x];
    }
    memset(cost, 0, (size_t)c->width1 * D * sizeof(int16_t));     /* :174 */
    uint8_t* buf = buffer - minX2;                                  /* :176 */
    for (int ch = 0; ch < 2; ch++, prow1 += width, prow2 += width) {   /* :184 */
        int diff_scale = ch < 1 ? 0 : 2;
        for (int x = minX2; x < maxX2; x++) {                       /* :191-200 */
            int v = prow2[x];
            int vl = x > 0 ? (v + prow2[x - 1]) / 2 : v;
            int vr = x < width - 1 ? (v + prow2[x + 1]) / 2 : v;
            int v0 = IMIN(vl, vr); v0 = IMIN(v0, v);
            int v1 = IMAX(vl, vr); v1 = IMAX(v1, v);
            buf[x] = (uint8_t)v0;
            buf[x + width2] = (uint8_t)v1;
        }
        for (int x = minX1; x < maxX1; x++) {                       /* :202-247 */
            int u = prow1[x];
            int ul = x > 0 ? (u + prow1[x - 1]) / 2 : u;
            int ur = x < width - 1 ? (u + prow1[x + 1]) / 2 : u;
            int u0 = IMIN(ul, ur); u0 = IMIN(u0, u);
            int u1 = IMAX(ul, ur); u1 = IMAX(u1, u);
            int16_t* cx = cost + (size_t)(x - minX1) * D - minD;
            for (int d = minD; d < maxD; d++) {
                int v = prow2[width - x - 1 + d];
                int v0 = buf[width - x - 1 + d];
                int v1 = buf[width - x - 1 + d + width2];
                int c0 = IMAX(0, u - v1); c0 = IMAX(c0, v0 - u);
                int c1 = IMAX(0, v - u1); c1 = IMAX(c1, u0 - v);
                cx[d] = (int16_t)(cx[d] + (IMIN(c0, c1) >> diff_scale));
            }
        }
    }
}

static inline int16_t sat16(int v) { return (int16_t)(v < SHRT_MIN ? SHRT_MIN : v > SHRT_MAX ? SHRT_MAX : v); }

/* ---- stereosgbm.cpp:303-824 computeDisparitySGBM, fullDP (2 passes), scalar paths.
 * img1/img2: canvases width x height.  disp1/cost1: width x height int16 (cost1 zero where unwritten).
 * Cout/Sout: optional height*width1*D dumps. */
static void compute_disparity_sgbm(const uint8_t* img1, const uint8_t* img2, int width, int height,
                                   int minDisparity, int numberOfDisparities, int SADWindowSize,
                                   int P1in, int P2in, int disp12MaxDiffIn, int preFilterCap,
                                   int uniquenessRatioIn, int alias_oob, long* oob_count,
                                   int16_t* disp1, int16_t* cost1, int16_t* Cout, int16_t* Sout)
{
    enum { NR2 = 8, DISP_SHIFT = 4, DISP_SCALE = 16 };
    const int MAX_COST = SHRT_MAX;
    int minD = minDisparity, maxD = minD + numberOfDisparities;                       /* :328 */
    int SADw = SADWindowSize > 0 ? SADWindowSize : 5;                                   /* :330 */
    int ftzero = IMAX(preFilterCap, 15) | 1;                                            /* :331 */
    int uniquenessRatio = uniquenessRatioIn >= 0 ? uniquenessRatioIn : 10;              /* :332 */
    int disp12MaxDiff = disp12MaxDiffIn > 0 ? disp12MaxDiffIn : 1;                      /* :333 */
    int P1 = P1in > 0 ? P1in : 2, P2 = IMAX(P2in > 0 ? P2in : 5, P1 + 1);               /* :334 */
    int minX1 = IMAX(-maxD, 0), maxX1 = width + IMIN(minD, 0);                          /* :336 */
    int D = maxD - minD, width1 = maxX1 - minX1;                                        /* :337 */
    int INVALID_DISP_SCALED = (minD - 1) * DISP_SCALE;                                  /* :338 */
    int SW2 = SADw / 2, SH2 = SADw / 2;                                                 /* :339 */

    if (minX1 >= maxX1) {                                                               /* :347-351 */
        for (size_t i = 0; i < (size_t)width * height; i++) disp1[i] = (int16_t)INVALID_DISP_SCALED;
        return;
    }

    bt_ctx bt;
    bt.width = width; bt.height = height; bt.minD = minD; bt.maxD = maxD; bt.D = D;
    bt.minX1 = minX1; bt.maxX1 = maxX1; bt.width1 = width1;
    bt.minX2 = IMAX(minX1 - maxD, 0); bt.maxX2 = IMIN(maxX1 - minD, width);          /* :122 */
    bt.width2 = bt.maxX2 - bt.minX2;
    bt.img1 = img1; bt.img2 = img2;
    bt.flat_alloc = (uint8_t*)calloc(GUARD + (size_t)width * 16 + 4096, 1);
    for (int k = 0; k < 256 + 1024 * 2; k++)                                            /* :344-345 */
        bt.clip[k] = (uint8_t)(IMIN(IMAX(k - 1024, -ftzero), ftzero) + ftzero);

    const int D2 = D + 16, NRD2 = NR2 * D2;                                             /* :357 */
    const size_t costBufSize = (size_t)width1 * D;                                      /* :368 */
    const size_t minLrSize = (size_t)(width1 + 2) * NR2, LrSize = minLrSize * D2;       /* :370 */
    const int hsumBufNRows = SH2 * 2 + 2;                                               /* :371 */

    int16_t* Cbuf = (int16_t*)calloc(costBufSize * height, sizeof(int16_t));          /* fullDP: one C,S row per y */
    int16_t* Sbuf = (int16_t*)calloc(costBufSize * height, sizeof(int16_t));
    int16_t* hsumBuf = (int16_t*)calloc(costBufSize * hsumBufNRows, sizeof(int16_t));
    int16_t* pixDiff = (int16_t*)calloc(costBufSize, sizeof(int16_t));
    /* One block with the reference's relative layout (:386-390):
     *   [Lr[0] | Lr[1] | minLr[0] | minLr[1] | disp2cost (width) | disp2ptr (width)]
     * because the reference indexes disp2cost/disp2ptr with _x2 = x + minX1 - d - minD (:781), which is
     * NEGATIVE for pixels left of their match (x < d + minD): an out-of-bounds access that lands in the
     * tail of minLr[1] (disp2cost[-k]) / the tail of disp2cost (disp2ptr[-k]).  With alias_oob != 0 we
     * reproduce that aliasing bit-for-bit; with alias_oob == 0 ("padded" semantics) negative _x2
     * goes to a private pad instead, i.e. the write has no side effect. */
    int16_t* block = (int16_t*)calloc(LrSize * 2 + minLrSize * 2 + (size_t)width * 2, sizeof(int16_t));
    int16_t* LrAll = block;
    int16_t* minLrAll = block + LrSize * 2;
    int16_t* disp2cost = minLrAll + minLrSize * 2;
    int16_t* disp2ptr = disp2cost + width;
    int16_t* pad_alloc = (int16_t*)calloc((size_t)2 * (D + 16), sizeof(int16_t));
    int16_t* pad_cost = pad_alloc + (D + 16);          /* padded semantics: disp2cost[-k] -> pad_cost[-k] */
    int16_t* pad_ptr = pad_alloc + 2 * (D + 16);       /* (never read back for a real decision)           */

    for (size_t k = 0; k < costBufSize; k++) Cbuf[k] = (int16_t)P2;                   /* :392-394 (row 0 only) */

    for (int pass = 1; pass <= 2; pass++) {
        int x1, y1, x2, y2, dx, dy;
        if (pass == 1) { y1 = 0; y2 = height; dy = 1; x1 = 0; x2 = width1; dx = 1; }    /* :400-409 */
        else { y1 = height - 1; y2 = -1; dy = -1; x1 = width1 - 1; x2 = -1; dx = -1; }

        int16_t *Lr[2], *minLr[2];
        for (int k = 0; k < 2; k++) {                                                   /* :413-424 */
            Lr[k] = LrAll + LrSize * k + NRD2 + 8;
            memset(Lr[k] - NRD2 - 8, 0, LrSize * sizeof(int16_t));
            minLr[k] = minLrAll + minLrSize * k + NR2;
            memset(minLr[k] - NR2, 0, minLrSize * sizeof(int16_t));
        }

        for (int y = y1; y != y2; y += dy) {
            int x, d;
            int16_t* disp1ptr = disp1 + (size_t)y * width;
            int16_t* cost1ptr = cost1 + (size_t)y * width;
            int16_t* C = Cbuf + (size_t)y * costBufSize;
            int16_t* S = Sbuf + (size_t)y * costBufSize;

            if (pass == 1) {                                                            /* :434-516 */
                int dy1 = y == 0 ? 0 : y + SH2, dy2 = y == 0 ? SH2 : dy1;
                for (int k = dy1; k <= dy2; k++) {
                    int16_t* hsumAdd = hsumBuf + (size_t)(IMIN(k, height - 1) % hsumBufNRows) * costBufSize;
                    if (k < height) {
                        calc_pixel_cost_bt(&bt, k, pixDiff);
                        memset(hsumAdd, 0, D * sizeof(int16_t));
                        for (x = 0; x <= SW2 * D; x += D) {
                            int scale = x == 0 ? SW2 + 1 : 1;
                            for (d = 0; d < D; d++)
                                hsumAdd[d] = (int16_t)(hsumAdd[d] + pixDiff[x + d] * scale);
                        }
                        if (y > 0) {
                            const int16_t* hsumSub = hsumBuf + (size_t)(IMAX(y - SH2 - 1, 0) % hsumBufNRows) * costBufSize;
                            const int16_t* Cprev = C - costBufSize;
                            for (x = D; x < width1 * D; x += D) {                       /* Q1: x = 0 never written */
                                const int16_t* pixAdd = pixDiff + IMIN(x + SW2 * D, (width1 - 1) * D);
                                const int16_t* pixSub = pixDiff + IMAX(x - (SW2 + 1) * D, 0);
                                for (d = 0; d < D; d++) {
                                    int hv = hsumAdd[x + d] = (int16_t)(hsumAdd[x - D + d] + pixAdd[d] - pixSub[d]);
                                    C[x + d] = (int16_t)(Cprev[x + d] + hv - hsumSub[x + d]);
                                }
                            }
                        } else {
                            for (x = D; x < width1 * D; x += D) {
                                const int16_t* pixAdd = pixDiff + IMIN(x + SW2 * D, (width1 - 1) * D);
                                const int16_t* pixSub = pixDiff + IMAX(x - (SW2 + 1) * D, 0);
                                for (d = 0; d < D; d++)
                                    hsumAdd[x + d] = (int16_t)(hsumAdd[x - D + d] + pixAdd[d] - pixSub[d]);
                            }
                        }
                    }                                                                   /* Q2: k >= height: row skipped */
                    if (y == 0) {
                        int scale = k == 0 ? SH2 + 1 : 1;
                        for (x = 0; x < width1 * D; x++)
                            C[x] = (int16_t)(C[x] + hsumAdd[x] * scale);
                    }
                }
                memset(S, 0, costBufSize * sizeof(int16_t));
            }

            memset(Lr[0] - NRD2 - 8, 0, NRD2 * sizeof(int16_t));                      /* :518-522 */
            memset(Lr[0] + width1 * NRD2 - 8, 0, NRD2 * sizeof(int16_t));
            memset(minLr[0] - NR2, 0, NR2 * sizeof(int16_t));
            memset(minLr[0] + width1 * NR2, 0, NR2 * sizeof(int16_t));

            for (x = x1; x != x2; x += dx) {                                            /* :542-662 */
                int xm = x * NR2, xd = xm * D2;
                int delta0 = minLr[0][xm - dx * NR2] + P2, delta1 = minLr[1][xm - NR2 + 1] + P2;
                int delta2 = minLr[1][xm + 2] + P2, delta3 = minLr[1][xm + NR2 + 3] + P2;
                int16_t* Lr_p0 = Lr[0] + xd - dx * NRD2;
                int16_t* Lr_p1 = Lr[1] + xd - NRD2 + D2;
                int16_t* Lr_p2 = Lr[1] + xd + D2 * 2;
                int16_t* Lr_p3 = Lr[1] + xd + NRD2 + D2 * 3;
                Lr_p0[-1] = Lr_p0[D] = Lr_p1[-1] = Lr_p1[D] =
                Lr_p2[-1] = Lr_p2[D] = Lr_p3[-1] = Lr_p3[D] = (int16_t)MAX_COST;
                int16_t* Lr_p = Lr[0] + xd;
                const int16_t* Cp = C + (size_t)x * D;
                int16_t* Sp = S + (size_t)x * D;
                int minL0 = MAX_COST, minL1 = MAX_COST, minL2 = MAX_COST, minL3 = MAX_COST;
                for (d = 0; d < D; d++) {                                               /* :633-656 */
                    int Cpd = Cp[d], L0, L1, L2, L3;
                    L0 = Cpd + IMIN((int)Lr_p0[d], IMIN(Lr_p0[d - 1] + P1, IMIN(Lr_p0[d + 1] + P1, delta0))) - delta0;
                    L1 = Cpd + IMIN((int)Lr_p1[d], IMIN(Lr_p1[d - 1] + P1, IMIN(Lr_p1[d + 1] + P1, delta1))) - delta1;
                    L2 = Cpd + IMIN((int)Lr_p2[d], IMIN(Lr_p2[d - 1] + P1, IMIN(Lr_p2[d + 1] + P1, delta2))) - delta2;
                    L3 = Cpd + IMIN((int)Lr_p3[d], IMIN(Lr_p3[d - 1] + P1, IMIN(Lr_p3[d + 1] + P1, delta3))) - delta3;
                    Lr_p[d] = (int16_t)L0; minL0 = IMIN(minL0, L0);
                    Lr_p[d + D2] = (int16_t)L1; minL1 = IMIN(minL1, L1);
                    Lr_p[d + D2 * 2] = (int16_t)L2; minL2 = IMIN(minL2, L2);
                    Lr_p[d + D2 * 3] = (int16_t)L3; minL3 = IMIN(minL3, L3);
                    Sp[d] = sat16(Sp[d] + L0 + L1 + L2 + L3);
                }
                minLr[0][xm] = (int16_t)minL0; minLr[0][xm + 1] = (int16_t)minL1;
                minLr[0][xm + 2] = (int16_t)minL2; minLr[0][xm + 3] = (int16_t)minL3;
            }

            if (pass == 2) {                                                            /* :664-816 */
                for (x = 0; x < width; x++) {
                    disp1ptr[x] = disp2ptr[x] = (int16_t)INVALID_DISP_SCALED;
                    disp2cost[x] = (int16_t)MAX_COST;
                }
                for (x = width1 - 1; x >= 0; x--) {
                    int16_t* Sp = S + (size_t)x * D;
                    int minS = MAX_COST, bestDisp = -1;
                    for (d = 0; d < D; d++) {                                           /* :762-770 */
                        int Sval = Sp[d];
                        if (Sval < minS) { minS = Sval; bestDisp = d; }
                    }
                    for (d = 0; d < D; d++)                                             /* :773-779 */
                        if (Sp[d] * (100 - uniquenessRatio) < minS * 100 && abs(bestDisp - d) > 1) break;
                    if (d < D) continue;
                    d = bestDisp;
                    int _x2 = x + minX1 - d - minD;                                     /* :781-786 */
                    if (_x2 < 0) {
                        if (oob_count) (*oob_count)++;
                        if (!alias_oob) {
                            /* padded semantics: behave as if the arrays extended to the left, initialised
                             * like the rest of the row (cost MAX_COST), so the store always happens but is
                             * never observed again. */
                            (void)pad_cost; (void)pad_ptr;
                            goto subpixel;
                        }
                    }
                    if (disp2cost[_x2] > minS) {
                        disp2cost[_x2] = (int16_t)minS;
                        disp2ptr[_x2] = (int16_t)(d + minD);
                    }
                subpixel:
                    if (0 < d && d < D - 1) {                                           /* :788-797 */
                        int denom2 = IMAX(Sp[d - 1] + Sp[d + 1] - 2 * Sp[d], 1);
                        d = d * DISP_SCALE + ((Sp[d - 1] - Sp[d + 1]) * DISP_SCALE + denom2) / (denom2 * 2);
                    } else
                        d *= DISP_SCALE;
                    disp1ptr[x + minX1] = (int16_t)(d + minD * DISP_SCALE);
                    cost1ptr[x + minX1] = (int16_t)minS;
                }
                for (x = minX1; x < maxX1; x++) {                                       /* :802-816 */
                    int d1 = disp1ptr[x];
                    if (d1 == INVALID_DISP_SCALED) continue;
                    int _d = d1 >> DISP_SHIFT;
                    int d_ = (d1 + DISP_SCALE - 1) >> DISP_SHIFT;
                    int _x = x - _d, x_ = x - d_;
                    if (0 <= _x && _x < width && disp2ptr[_x] >= minD && abs(disp2ptr[_x] - _d) > disp12MaxDiff &&
                        0 <= x_ && x_ < width && disp2ptr[x_] >= minD && abs(disp2ptr[x_] - d_) > disp12MaxDiff)
                        disp1ptr[x] = (int16_t)INVALID_DISP_SCALED;
                }
            }
            { int16_t* t = Lr[0]; Lr[0] = Lr[1]; Lr[1] = t; }                           /* :819-820 */
            { int16_t* t = minLr[0]; minLr[0] = minLr[1]; minLr[1] = t; }
        }
    }
    if (Cout) memcpy(Cout, Cbuf, costBufSize * height * sizeof(int16_t));
    if (Sout) memcpy(Sout, Sbuf, costBufSize * height * sizeof(int16_t));
    free(Cbuf); free(Sbuf); free(hsumBuf); free(pixDiff); free(block); free(pad_alloc); free(bt.flat_alloc);
}

/* Test knobs (see compute_disparity_sgbm): 1 = reproduce the reference's out-of-bounds aliasing
 * bit-for-bit (default; this is what pins the oracle against the real reference), 0 = "padded"
 * semantics.  s2p_oracle_oob_count counts the out-of-bounds disp2 accesses of the last call. */
int s2p_oracle_alias_oob = 1;
long s2p_oracle_oob_count = 0;

/* ---- 3rdparty/sgbm/sgbm.cpp:139-241 main + StereoSGBM::operator() (stereosgbm.cpp:828-846) ---- */
int s2p_oracle_sgbm(const float* im1, const float* im2, int w, int h,
                    int dmin, int dmax, int win, int P1, int P2, int lr,
                    float* odisp, float* ocost, s2p_oracle_dump* dump)
{
    size_t n = (size_t)w * h;
    float rmin, rmax;
    uint8_t* q1 = (uint8_t*)malloc(n);
    uint8_t* q2 = (uint8_t*)malloc(n);
    s2p_oracle_rminmax(im1, n, &rmin, &rmax);                 /* sgbm.cpp:153-155: im2 uses im1's thresholds */
    s2p_oracle_quantize(im1, n, rmin, rmax, q1);
    s2p_oracle_quantize(im2, n, rmin, rmax, q2);
    if (dump) { dump->rminmax[0] = rmin; dump->rminmax[1] = rmax; }
    if (dump && dump->q1) memcpy(dump->q1, q1, n);
    if (dump && dump->q2) memcpy(dump->q2, q2, n);

    s2p_oracle_oob_count = 0;
    int maxdisp = -dmin, mindisp = -dmax;                     /* sgbm.cpp:166-168: sign flip to the OpenCV convention */
    if (mindisp >= maxdisp) { free(q1); free(q2); return 1; } /* sgbm.cpp:174-177 */
    int ndisp = (int)(16 * ceil((maxdisp - mindisp) / 16.0)); /* sgbm.cpp:181 */
    int x0 = IMAX(maxdisp, 0);                                /* sgbm.cpp:204-207 crop trick */
    int Wc = w + IMAX(-mindisp, 0) + IMAX(maxdisp, 0);
    uint8_t* uu1 = (uint8_t*)calloc((size_t)Wc * h, 1);       /* margins: "uninitialised == 0" */
    uint8_t* uu2 = (uint8_t*)calloc((size_t)Wc * h, 1);
    for (int y = 0; y < h; y++) {
        memcpy(uu1 + (size_t)y * Wc + x0, q1 + (size_t)y * w, w);
        memcpy(uu2 + (size_t)y * Wc + x0, q2 + (size_t)y * w, w);
    }
    int16_t* ddisp = (int16_t*)calloc((size_t)Wc * h, sizeof(int16_t));
    int16_t* ccost = (int16_t*)calloc((size_t)Wc * h, sizeof(int16_t));
    int16_t* tmp = (int16_t*)malloc((size_t)Wc * h * sizeof(int16_t));

    int minD = mindisp, maxD = minD + ndisp;
    int minX1 = IMAX(-maxD, 0), maxX1 = Wc + IMIN(minD, 0), width1 = maxX1 - minX1;
    if (width1 == 1) {   /* the reference reads past its one-column pixDiff row here (:447-451): undefined, refused */
        free(q1); free(q2); free(uu1); free(uu2); free(ddisp); free(ccost); free(tmp);
        return 4;
    }
    if (dump) {
        dump->geom[0] = Wc; dump->geom[1] = width1; dump->geom[2] = ndisp; dump->geom[3] = minD;
        dump->geom[4] = x0; dump->geom[5] = minX1; dump->geom[6] = maxX1; dump->geom[7] = (minD - 1) * 16;
    }
    /* sgbm.cpp:188-192: fullDP=1, preFilterCap=63, uniquenessRatio=10, speckleWindowSize=50, speckleRange=1 */
    compute_disparity_sgbm(uu1, uu2, Wc, h, mindisp, ndisp, win, P1, P2, lr, 63, 10,
                           s2p_oracle_alias_oob, &s2p_oracle_oob_count, ddisp, ccost, dump ? dump->C : NULL, dump ? dump->S : NULL);
    if (dump && dump->disp_raw) memcpy(dump->disp_raw, ddisp, (size_t)Wc * h * 2);
    if (dump && dump->cost_raw) memcpy(dump->cost_raw, ccost, (size_t)Wc * h * 2);
    memcpy(tmp, ddisp, (size_t)Wc * h * 2);                   /* smooth.cpp:448-451 in-place => copy */
    s2p_oracle_median3x3_s16(tmp, ddisp, Wc, h);              /* stereosgbm.cpp:842 */
    if (dump && dump->disp_med) memcpy(dump->disp_med, ddisp, (size_t)Wc * h * 2);
    s2p_oracle_speckle_s16(ddisp, Wc, h, (mindisp - 1) * 16, 50, 16 * 1);   /* stereosgbm.cpp:844-845 */
    if (dump && dump->disp_fin) memcpy(dump->disp_fin, ddisp, (size_t)Wc * h * 2);

    for (int y = 0; y < h; y++)                               /* sgbm.cpp:210-234 */
        for (int x = 0; x < w; x++) {
            int16_t dv = ddisp[(size_t)y * Wc + x0 + x];
            if (dv == -((-mindisp + 1) * 16)) {
                odisp[(size_t)y * w + x] = NAN;
                ocost[(size_t)y * w + x] = NAN;
            } else {
                odisp[(size_t)y * w + x] = (float)(-((float)dv) / 16.0);
                ocost[(size_t)y * w + x] = (float)ccost[(size_t)y * Wc + x0 + x];
            }
        }
    free(q1); free(q2); free(uu1); free(uu2); free(ddisp); free(ccost); free(tmp);
    return 0;
}

/* ---- s2p/block_matching.py:18-32 create_rejection_mask.
 * plambda "x 0 join" builds the flow (d, 0); backflow samples im2 at (x + d, y); the final plambda
 * multiplies the three isfinite() tests.  backflow's interpolator is not in the tree; the rule is
 * pinned on the one triple the reference's tests hold (rectified_ref.tif, img_02 warped by H_sec.txt,
 * rectified_disp.tif -> rectified_mask.png; tests/test_oracle_tile.py): that mask keeps the 17 pixels
 * whose sample x + d falls up to half a pixel outside the first / last column, so the sample is
 * taken as finite iff x + d lies in [-0.5, w - 0.5] and the two bilinear taps it uses (floor, and
 * floor + 1 when the fraction is non-zero; tap indices clamped to the row) are finite.  Beyond half a
 * pixel nothing is pinned (no stored disparity points there): rejected. */
void s2p_oracle_rejection_mask(const float* disp, const float* im1, const float* im2,
                               int w, int h, uint8_t* mask)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            size_t i = (size_t)y * w + x;
            int ok = isfinite(disp[i]) && isfinite(im1[i]);
            if (ok) {
                float xs = (float)x + disp[i];
                if (!(xs >= -0.5f && xs <= (float)w - 0.5f)) ok = 0;
                else {
                    int xi = (int)floorf(xs);
                    float fr = xs - (float)xi;
                    int t0 = xi < 0 ? 0 : (xi > w - 1 ? w - 1 : xi), t1 = xi + 1 < 0 ? 0 : (xi + 1 > w - 1 ? w - 1 : xi + 1);
                    ok = isfinite(im2[(size_t)y * w + t0]) && (fr == 0.0f || isfinite(im2[(size_t)y * w + t1]));
                }
            }
            mask[i] = (uint8_t)ok;
        }
}
