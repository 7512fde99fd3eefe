/* oracle/oracle.h -- TEST INFRASTRUCTURE.
 *
 * CPU restatement (plain C99) of the reference algorithms on the hot path, used ONLY as the checker
 * by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product
 * (s2p_amd/, libs2p_hip.so) never includes, links or calls anything declared here.
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 */
#ifndef S2P_ORACLE_H
#define S2P_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Intermediate dumps; identical layout to `s2p_ref_dump` in ref_harness.cpp so that the same
 * Python structure drives both.  All pointers optional (NULL = skip). */
typedef struct {
    uint8_t* q1;        /* w*h quantised im1                                  */
    uint8_t* q2;        /* w*h quantised im2                                  */
    int16_t* C;         /* h*width1*D block cost (+P2 bias), layout [y][x][d] */
    int16_t* S;         /* h*width1*D aggregated cost                         */
    int16_t* disp_raw;  /* h*Wc canvas disparity (x16) before median          */
    int16_t* disp_med;  /* h*Wc after 3x3 median                              */
    int16_t* disp_fin;  /* h*Wc after speckle filter                          */
    int16_t* cost_raw;  /* h*Wc canvas cost                                   */
    int geom[8];        /* out: Wc, width1, D, minD, x0, minX1, maxX1, INVALID_SCALED */
    float rminmax[2];   /* out */
} s2p_oracle_dump;

/* `sgbm im1 im2 disp cost dmin dmax win P1 P2 lr` (s2p/block_matching.py:116-132;
 * 3rdparty/sgbm/sgbm.cpp:139-241; 3rdparty/sgbm/stereosgbm.cpp:115-280,303-846,872-967).
 * Disparities in the s2p convention im1(x) <-> im2(x+d); invalid = NaN.  Returns 0, or 1 when
 * the reference binary would exit(1) (empty range). */
int s2p_oracle_sgbm(const float* im1, const float* im2, int w, int h,
                    int dmin, int dmax, int win, int P1, int P2, int lr,
                    float* odisp, float* ocost, s2p_oracle_dump* dump);

/* Individual stages, exposed so tests can pin them separately. */
void s2p_oracle_rminmax(const float* x, size_t n, float* rmin, float* rmax);      /* sgbm.cpp:30-42 */
void s2p_oracle_quantize(const float* x, size_t n, float rmin, float rmax, uint8_t* y); /* sgbm.cpp:44-71 */
void s2p_oracle_median3x3_s16(const int16_t* src, int16_t* dst, int w, int h);    /* smooth.cpp:207-292 */
void s2p_oracle_speckle_s16(int16_t* img, int w, int h, int newVal, int maxSize, int maxDiff); /* stereosgbm.cpp:872-967 */

/* create_rejection_mask (s2p/block_matching.py:18-32): mask = isfinite(disp) & isfinite(im1) &
 * isfinite(im2 sampled at (x+d, y)).  `backflow`/`plambda` sources are absent from the reference
 * (un-vendored imscript submodule); see DESIGN.md for the sampling rule adopted. */
void s2p_oracle_rejection_mask(const float* disp, const float* im1, const float* im2,
                               int w, int h, uint8_t* mask);

/* ---- census / 8-path SGM matcher standing in for `mgm` / `mgm_multi` (census_oracle.c; parity
 * unpinned at source level, see that file's header).  Range [dmin, dmax] INCLUSIVE like mgm's -r/-R. */
typedef struct {
    int census_win;        /* CENSUS_NCC_WIN (5); 3 or 5                                  */
    int P1, P2;            /* 8, 32 (x stereo_regularity_multiplier for mgm_multi)        */
    int nb_dir;            /* -O 8; 4 = the axis directions; 16 = + the knight's moves (recursion >= 1 only) */
    int lr_check;          /* TESTLRRL: 0 off, 1 on (every scale), 2 on at the finest scale only */
    float lr_tau;          /* TESTLRRL_TAU (1.0)                                          */
    int mindiff;           /* MINDIFF: <= 0 disabled (the reference's default -1); t > 0: a pixel is rejected when the smallest S among the */
                           /* candidates at least 2 away from the winner is less than t above the winner's (unpinned statement)           */
    int median;            /* MEDIAN=1 ('mgm' branch)                                     */
    int remove_small_cc;   /* REMOVESMALLCC ('mgm_multi' branch: 25), 0 = off             */
    int fix_overcount;     /* S = sum_r L_r - (8 - 1) C (mgm's TSGM_FIX_OVERCOUNT, default 1) */
    int recursion;         /* 0: 8 independent 1-D paths (SGM); 1: MGM's two-predecessor recursion; 2: three predecessors (TSGM = 3) */
    int scales;            /* mgm_multi's -S: <= 1 single scale; n: up to n - 1 halvings (while the smaller side stays >= 128) */
    int subpix;            /* mgm_multi's SUBPIX: 1 (or 0) whole-pixel candidates, 2 half-pixel candidates */
    int cost;              /* 0: census / Hamming (the reference's call sites: -t census); 1: ZNCC on the same window (north_star's */
                           /* "census/ZNCC"; no call site of the reference reaches it: unpinned), quantised to the census scale      */
    int subpix_model;      /* ORACLE-ONLY experiment switch (other readings of SUBPIX=2, judged on the end-to-end     */
                           /* rasters; the library implements model 0): 0 = half-pixel candidates from image 2 sampled half way between */
                           /* its columns (subpix = 2); 1 = subpix = 2 with the half-pixel candidates' COST interpolated between their   */
                           /* whole-pixel neighbours ((a + b + 1) >> 1); 2 = subpix = 1 aggregation, then the winner refined on the     */
                           /* half-pixel grid (census cost at d +- 1/2 against the half-sampled image, S interpolated + the cost          */
                           /* difference) before the V fit                                                                              */
} s2p_oracle_census_params;

typedef struct {
    uint8_t* C;            /* h*w*D Hamming cost, D = roundup(dmax-dmin+1, 16), 255 = excluded */
    uint16_t* S;           /* h*w*D sum of the 8 path costs                                     */
    float* disp_raw;       /* h*w after WTA / vfit / L-R                                         */
    float* disp_med;       /* h*w after the median                                               */
    int dmin0, D0;         /* out: first disparity and depth of the C / S layout (a multi-scale call matches its finest level   */
                           /* over the union of the admissible ranges: D0 <= D of the call, the arrays are filled to h*w*D0)    */
} s2p_oracle_census_dump;

void s2p_oracle_census(const float* im, int w, int h, int win, uint32_t* out);
void s2p_oracle_down2(const float* src, int w, int h, float* dst);                 /* pyramid step of the multi-scale mode */
void s2p_oracle_range_from_coarse(const float* dc, int w, int h, int dmin, int dmax, int16_t* lo, int16_t* hi);
int s2p_oracle_census_levels(int w, int h, int scales);
int s2p_oracle_census_sgm(const float* im1, const float* im2, int w, int h, int dmin, int dmax,
                          const s2p_oracle_census_params* p, float* odisp, float* oconf, uint8_t* omask,
                          s2p_oracle_census_dump* dump);

void s2p_oracle_erode_disk(const uint8_t* msk, int w, int h, int radius, uint8_t* out);   /* masking.py:87-97 */

/* ---- homography resampler standing in for the `homography` binary (resample_oracle.c; parity
 * unpinned at source level, pinned empirically on rectified_ref.tif). */
void s2p_oracle_bspline5_prefilter(float* img, int w, int h);
int s2p_oracle_warp_homography(const float* src, int sw, int sh, const double* H, float* dst, int w, int h);

/* ---- triangulation (triangulation_oracle.c): struct rpc of c/rpc.h:13-31 and disp_to_lonlatalt of
 * c/disp_to_h.c:70-140, same argument list. */
typedef struct {
    double numx[20], denx[20], numy[20], deny[20], scale[3], offset[3];
    double inumx[20], idenx[20], inumy[20], ideny[20], iscale[3], ioffset[3];
    double dmval[4], imval[4], delta;
} s2p_oracle_rpc;
void s2p_oracle_disp_to_lonlatalt(double* lonlatalt, float* err, const float* dispx, const float* dispy,
                                  const float* msk, int nx, int ny, const float* msk_orig, int w, int h,
                                  const double ha[9], const double hb[9],
                                  const s2p_oracle_rpc* rpca, const s2p_oracle_rpc* rpcb, const float bbox[4]);

/* the rest of lib/disp_to_h.so (c/disp_to_h.c:43-67, 143-230), same argument lists */
void s2p_oracle_stereo_corresp_to_lonlatalt(double* lonlatalt, float* err, const float* kp_a, const float* kp_b, int n_kp,
                                            const s2p_oracle_rpc* rpca, const s2p_oracle_rpc* rpcb);
void s2p_oracle_count_3d_neighbors(int* count, const double* xyz, int nx, int ny, float r, int p);
void s2p_oracle_remove_isolated_3d_points(double* xyz, int nx, int ny, float r, int p, int n, int q);

/* triangulation.height_map_to_xyz before the CRS conversion (s2p/triangulation.py:165-219): lon, lat of every pixel
 * (c + off_x, r + off_y) of a w x h float32 height map at its altitude; lonlatalt h x w x 3, NaN where the height is. */
void s2p_oracle_height_map_to_lonlatalt(const s2p_oracle_rpc* rpc, const float* heights, int w, int h, int off_x, int off_y,
                                        double* lonlatalt);

/* common.cargarse_basura (s2p/common.py:224-235; cleanup_oracle.c): 5 x 5 range filter (> 5 -> NaN) + removal of the
 * connected components of fewer than 200 pixels.  in / out: w x h float32.  Returns 0, -1 on allocation failure. */
int s2p_oracle_cargarse_basura(const float* in, int w, int h, float* out);

/* ---- DSM rasterisation (rasterize_oracle.c): `rasterize_cloud` of the plyflatten package (pip dependency of the
 * reference, s2p/__init__.py:462-466), same argument meaning; pts = npts x (2 + nb) doubles (x, y, nb values),
 * raster = ysize x xsize x nb float32.  Returns 0, or < 0 on a bad argument / allocation failure. */
int s2p_oracle_plyflatten(const double* pts, int npts, int nb, double xoff, double yoff, double res,
                          int xsize, int ysize, int radius, float sigma, float* raster);

#ifdef __cplusplus
}
#endif
#endif
