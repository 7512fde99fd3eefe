/* include/s2p_hip.h -- C ABI of libs2p_hip.so, the MI355X-native replacement for the subprocess
 * boundary of the s2p stereo hot path.
 *
 * What each entry point replaces in the reference (paths relative to centreborelli/s2p):
 *
 *   s2p_hip_sgbm*        the `sgbm` binary + the three `plambda`/`backflow` subprocesses of
 *                        create_rejection_mask, i.e. the whole `algo == 'sgbm'` branch of
 *                        s2p.block_matching.compute_disparity_map
 *                        (s2p/block_matching.py:116-134 and :18-32; binary = 3rdparty/sgbm/sgbm.cpp:139-241
 *                        + 3rdparty/sgbm/stereosgbm.cpp:303-846,872-967).
 *   s2p_hip_census_sgm*  the `mgm` / `mgm_multi` binaries (+ the same rejection mask):
 *                        s2p/block_matching.py:155-188,269-310.
 *   s2p_hip_warp*        the `homography` binary behind s2p.common.image_apply_homography
 *                        (s2p/common.py:159-180), called twice by rectification.rectify_pair
 *                        (s2p/rectification.py:379-380).
 *
 * Conventions follow the reference's existing ctypes libraries (lib/disp_to_h.so,
 * s2p/triangulation.py:117-145): plain pointers and sizes, caller-allocated C-contiguous row-major
 * buffers, no ownership transfer.  Unlike disp_to_h.so every function returns an int status
 * (the subprocess boundary it replaces reported errors through exit codes / timeouts).
 *
 * Disparity convention: s2p's, im1(x, y) <-> im2(x + d, y).  Invalid disparity = NaN.  Mask: 1 = keep.
 * Limits (S2P_HIP_UNSUPPORTED beyond them): at most 1024 disparity candidates (after the sgbm driver's
 * rounding up to a multiple of 16), cost volume h*w*D*sizeof(cost) < 4 GiB, and one image row of per-pixel
 * state in the 160 KiB of LDS of a CU: census tiles up to ~15000 px wide (10 w + 8 D + 16 bytes <= 156 KiB; ~11000 with
 * half-pixel candidates: 14 w + 8 D), sgbm canvases (w + |range|) up to 8192 px.
 *
 * Two flavours per operation:
 *   *_host : host pointers in, host pointers out (what the Python shim uses: it decodes TIFFs to
 *            numpy, calls this, encodes the outputs).  Does H2D, kernels, D2H, synchronises.
 *   *_dev  : device pointers in/out, enqueued on the context's HIP stream, asynchronous
 *            (tile schedulers, bench.py; inputs already resident in HBM).
 * No HIP call is made at load time: the runtime is initialised lazily by s2p_hip_ctx_create, so
 * the library is safe to import before multiprocessing's fork (s2p/parallel.py:80).
 */
#ifndef S2P_HIP_H
#define S2P_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* the library is built with -fvisibility=hidden: only the entry points below are exported */
#define S2P_API __attribute__((visibility("default")))

/* ---- status codes ------------------------------------------------------------------------- */
enum {
    S2P_HIP_OK            = 0,
    S2P_HIP_EMPTY_RANGE   = 1,  /* sgbm.cpp:174-177 exit(1) -> subprocess.CalledProcessError in the shim */
    S2P_HIP_TIMEOUT       = 2,  /* deadline exceeded -> subprocess.TimeoutExpired in the shim           */
    S2P_HIP_RUNTIME_ERROR = 3,  /* HIP error (see s2p_hip_last_error)                                   */
    S2P_HIP_UNSUPPORTED   = 4,  /* parameter outside what the kernels implement                         */
    S2P_HIP_BAD_ARGUMENT  = 5
};

typedef struct s2p_hip_ctx s2p_hip_ctx;   /* one per (process, device, stream): workspace + stream */

/* Create a context on `device`.  `stream` is a hipStream_t to enqueue on (NULL = the context creates
 * and owns a non-blocking stream).  Lazy: first HIP call of the process happens here.
 * At most S2P_HIP_MAX_PROCS_PER_DEVICE processes (environment; default 8, 0 = no limit) may hold contexts on one physical
 * device at a time: a further process gets S2P_HIP_UNSUPPORTED (the message names the way out: the device's broker, which
 * serves any number of Pool workers through one process).  A lost worker of the reference's Pool must surface as an
 * exception (s2p/parallel.py:100-105), and beyond the device's hardware queues the runtime time-slices whole processes. */
S2P_API int  s2p_hip_ctx_create(int device, void* stream, s2p_hip_ctx** out);
S2P_API void s2p_hip_ctx_destroy(s2p_hip_ctx* ctx);
S2P_API int  s2p_hip_ctx_sync(s2p_hip_ctx* ctx);
/* Opt-in hipGraph replay for the *_dev entry points: the kernel sequence of a call is stream-captured
 * the first time a (geometry, parameters, pointers) signature is seen and replayed afterwards
 * (one graph launch instead of 12-25 kernel launches: matters for small tiles, which are launch-bound).
 * Meant for schedulers that reuse their device buffers; at most 32 signatures are kept. */
S2P_API int  s2p_hip_ctx_use_graphs(s2p_hip_ctx* ctx, int on);
S2P_API const char* s2p_hip_last_error(void);
/* What this library is: "libs2p_hip gfx950 <compiler>" for the shipped build.  A PROBE BUILD (any measurement / tuning switch of
 * s2p_amd/csrc/probe_guard.hpp on the compiler's command line: results may be invalid) answers "... PROBE BUILD [<flags>]", prefixes
 * every s2p_hip_last_error() message with "[PROBE BUILD] " and carries the data symbol s2p_hip_probe_build_marker; s2p_amd/build.py
 * never writes one to s2p_amd/lib/, and s2p_amd/_lib.py refuses to load one from there. */
S2P_API const char* s2p_hip_build_info(void);
S2P_API int  s2p_hip_device_count(void);          /* 0 when no HIP device is visible; -1 in a process forked from one that had already
                                                   * used the GPU (HIP does not survive fork: s2p_hip_last_error says so) */

/* ---- page-locked host memory for the buffers handed to the *_host entry points ---------------------------------
 * The *_host entries issue their transfers with hipMemcpyAsync on the context's stream.  On pageable memory (a plain
 * numpy array) the runtime stages such a copy through its own bounce buffers on the CALLING thread, chunk by chunk: the
 * call blocks for the whole transfer and tiles in flight on other streams queue behind one staging path.  On
 * page-locked memory the same call is a DMA the copy engines run on their own, so the upload of one tile, the kernels of
 * a second and the download of a third overlap ("pinned staging").  These two functions hand
 * out such memory (hipHostMalloc, portable across the devices of the process); s2p_amd/_lib.py wraps it as numpy arrays
 * (pinned_empty) with a size-class free list, the tile scheduler and the file-level shim read their inputs into it and
 * receive their outputs in it.  Nothing changes in the entry points' contract: any host pointer is accepted. */
S2P_API int  s2p_hip_pinned_alloc(size_t bytes, void** out);
S2P_API void s2p_hip_pinned_free(void* p);
/* Page-lock memory the caller already owns (hipHostRegister, portable) -- the GPU broker (s2p_amd/broker.py) registers the
 * shared-memory arenas of its clients, the forked Pool workers of the reference (s2p/parallel.py:76-110), once per
 * connection, so that their tiles move by DMA straight from / to the worker's own pages.  A range the driver refuses to pin
 * (S2P_HIP_RUNTIME_ERROR) is still a valid argument of every *_host entry, as pageable memory. */
S2P_API int  s2p_hip_host_register(void* p, size_t bytes);
S2P_API void s2p_hip_host_unregister(void* p);

/* ---- sgbm (bit-exact OpenCV-2.4 StereoSGBM as driven by the s2p `sgbm` binary) --------------- */
typedef struct {
    int win;               /* SADWindowSize; the reference passes 3 (block_matching.py:125); only 3 is implemented */
    int P1, P2;            /* 8, 32 (block_matching.py:121-122); 0 < P1 < P2 <= 255                               */
    int lr;                /* disp12MaxDiff, 1 (block_matching.py:126)                                           */
    int prefilter_cap;     /* 63  (sgbm.cpp:189)  */
    int uniqueness_ratio;  /* 10  (sgbm.cpp:190)  */
    int speckle_window;    /* 50  (sgbm.cpp:191); 0 disables the speckle filter */
    int speckle_range;     /* 1   (sgbm.cpp:192)  */
} s2p_sgbm_params;

S2P_API void s2p_hip_sgbm_default_params(s2p_sgbm_params* p);

/* `sgbm im1 im2 disp cost dmin dmax win P1 P2 lr` followed by create_rejection_mask.
 * im1, im2: w*h float32 (NaN allowed).  disp, cost: w*h float32 out (cost may be NULL).
 * mask: w*h uint8 out (may be NULL).  timeout_s < 0: no deadline; otherwise S2P_HIP_TIMEOUT is
 * returned when the call cannot finish within timeout_s seconds (0 => always). */
S2P_API int s2p_hip_sgbm_host(s2p_hip_ctx* ctx, const float* im1, const float* im2, int w, int h,
                      int dmin, int dmax, const s2p_sgbm_params* params,
                      float* disp, float* cost, uint8_t* mask, double timeout_s);

/* Same with device pointers, asynchronous on the context stream. */
S2P_API int s2p_hip_sgbm_dev(s2p_hip_ctx* ctx, const float* d_im1, const float* d_im2, int w, int h,
                     int dmin, int dmax, const s2p_sgbm_params* params,
                     float* d_disp, float* d_cost, uint8_t* d_mask);

/* Intermediate stages for parity tests (host pointers, all optional; same layout as the oracle's
 * dump: tests/ compare them one by one).  Runs the same kernels as s2p_hip_sgbm_host. */
typedef struct {
    uint8_t* q1;        /* w*h quantised im1                                  */
    uint8_t* q2;        /* w*h quantised im2                                  */
    int16_t* C;         /* h*width1*D block cost (+P2 bias), layout [y][x][d] */
    int16_t* S;         /* h*width1*D aggregated cost (sum of the 8 paths)    */
    int16_t* disp_raw;  /* h*Wc canvas disparity (x16) before median          */
    int16_t* disp_med;  /* h*Wc after 3x3 median                              */
    int16_t* disp_fin;  /* h*Wc after speckle filter                          */
    int16_t* cost_raw;  /* h*Wc canvas cost                                   */
    int geom[8];        /* out: Wc, width1, D, minD, x0, minX1, maxX1, INVALID_SCALED */
    float rminmax[2];   /* out */
} s2p_hip_sgbm_dump;

S2P_API int s2p_hip_sgbm_debug(s2p_hip_ctx* ctx, const float* im1, const float* im2, int w, int h,
                       int dmin, int dmax, const s2p_sgbm_params* params,
                       float* disp, float* cost, uint8_t* mask, s2p_hip_sgbm_dump* dump);

/* Geometry helper (host only, no GPU): fills geom[8] as above.  Returns S2P_HIP_EMPTY_RANGE when
 * the binary would exit(1). */
S2P_API int s2p_hip_sgbm_geometry(int w, int dmin, int dmax, int geom[8]);

/* ---- census / 8-path SGM matcher: the GPU stand-in for `mgm` and `mgm_multi` -------------------
 * (s2p/block_matching.py:155-188 and :269-310: `mgm -r dmin -R dmax -s vfit -t census -O 8
 * -confidence_consensusL conf im1 im2 disp` with MEDIAN / CENSUS_NCC_WIN / TESTLRRL / TESTLRRL_TAU /
 * MINDIFF / REMOVESMALLCC in the environment).  The binaries' sources are not in the reference tree;
 * the algorithm implemented is stated in oracle/census_oracle.c and DESIGN_PARITY.md section 3.
 * Range [dmin, dmax] is INCLUSIVE (mgm's -r / -R). */
typedef struct {
    int census_win;        /* CENSUS_NCC_WIN, cfg['census_ncc_win'] = 5; 3 or 5                          */
    int P1, P2;            /* 8, 32 (x cfg['stereo_regularity_multiplier'] for mgm_multi); P1 < P2 <= 128 */
    int nb_dir;            /* -O, cfg['mgm_nb_directions'] = 8; 8, 4 = the axis directions, or 16 = + the 8 knight's moves (round 4;
                              recursion 1 / 2 only; which 16 the absent binary means is an assumption: UNPINNED) */
    int lr_check;          /* TESTLRRL, cfg['mgm_leftright_control']: 0 off, 1 on, 2 on at the finest scale only */
    float lr_tau;          /* TESTLRRL_TAU, cfg['mgm_leftright_threshold'] = 1.0                         */
    int mindiff;           /* MINDIFF, cfg['mgm_mindiff_control'] (s2p/config.py:158-160): <= 0 disabled (the reference's default -1); t > 0: a */
                           /* pixel is rejected when the smallest summed cost among the candidates at least 2 away from the winner is less  */
                           /* than t above the winner's ("conservative results"; the binary's source is absent: an UNPINNED statement)   */
    int median;            /* MEDIAN=1 in the 'mgm' branch                                               */
    int remove_small_cc;   /* REMOVESMALLCC = cfg['stereo_speckle_filter'] (25) in the 'mgm_multi' branch */
    int fix_overcount;     /* 1 (default): S = sum_r L_r - 7 C, the data term counted once (mgm's           */
                           /* TSGM_FIX_OVERCOUNT default); 0: the plain sum of the 8 path costs               */
    int recursion;         /* 2 (default): MGM's recursion with three predecessors per direction (p - r, p - r_perp, */
                           /* p - r - r_perp: the model of TSGM=3 of the 'mgm' call site, s2p/block_matching.py:158 -- */
                           /* what the shim runs and the mode that meets the parity bar: 99.58 % of the reference's     */
                           /* stored tile within 0.5 px, all three end-to-end rasters inside compare_dsm's tolerances;  */
                           /* P2 <= 127); 1: two predecessors (the published form; what 'mgm_multi' runs: 99.53 %);     */
                           /* 0: 8 independent 1-D paths (plain SGM: north_star's wording; a 2 x faster PREVIEW mode   */
                           /* below the parity bar, 98.9 %)                                                            */
    int scales;            /* mgm_multi's -S (block_matching.py:292 passes 6): <= 1 (default) single scale; n: the */
                           /* pair is halved up to n - 1 times (while its smaller side stays >= 128 px), the       */
                           /* coarsest level is matched over the whole halved range and every level restricts the  */
                           /* candidates of the next finer one per pixel ([2 min - 2, 2 max + 2] of the 3x3 parent */
                           /* neighbourhood); a level is matched over the union of those ranges, which is also what */
                           /* a pixel without a parent estimate searches.  A multi-level call synchronises the      */
                           /* stream once per level (8 bytes come back to the host to size the next level): the     */
                           /* _dev entry is then not fully asynchronous                                             */
    int subpix;            /* mgm_multi's SUBPIX (=2, block_matching.py:277): 1 (default, or 0) whole-pixel      */
                           /* candidates; 2: a candidate every half pixel (image 2 sampled half way between its    */
                           /* columns); at most 1024 candidates either way                                        */
    int cost;              /* 0 (default): census transform + Hamming distance (`-t census`: what both call sites of */
                           /* the reference pass, block_matching.py:171,293); 1: ZNCC on the same window (north_star's */
                           /* "census/ZNCC"; no reference call site reaches it: unpinned), float32, quantised to the   */
                           /* census scale clamp(floor((1 - zncc) 12 + 0.5), 0, 24); half-pixel candidates correlate   */
                           /* with image 2 sampled half way between its columns; tiles up to ~3600 px wide (~2300 with */
                           /* subpix = 2: the window rows of the images live in LDS)                                    */
} s2p_census_params;

S2P_API void s2p_hip_census_default_params(s2p_census_params* p);

/* disp: w*h float32 out (NaN = invalid).  conf: w*h float32 out, the `<disp>_confidence.tif` image
 * (fraction of the 8 directions whose own winner is within 1 of the final one; may be NULL).
 * mask: w*h uint8 rejection mask (may be NULL). */
S2P_API int s2p_hip_census_sgm_host(s2p_hip_ctx* ctx, const float* im1, const float* im2, int w, int h,
                            int dmin, int dmax, const s2p_census_params* params,
                            float* disp, float* conf, uint8_t* mask, double timeout_s);
S2P_API int s2p_hip_census_sgm_dev(s2p_hip_ctx* ctx, const float* d_im1, const float* d_im2, int w, int h,
                           int dmin, int dmax, const s2p_census_params* params,
                           float* d_disp, float* d_conf, uint8_t* d_mask);

/* A batch of n equal-shape tiles in one asynchronous call (n <= 64; arrays of n device pointers, d_conf / d_mask may be NULL or
 * hold NULLs): the same results as n calls of s2p_hip_census_sgm_dev, tile by tile.  In the MGM modes (recursion 1 / 2, single
 * scale) the tiles share ONE aggregation launch -- the lattices of all tiles under one ready queue, so that the dependency chains
 * of n tiles fill the chip without n streams: 0.57 ms of aggregation per 1024^2 x 128 tile in a launch of 8 against 1.05 ms for a
 * tile alone; other parameters run the tiles one after the other.  Workspace: n x (9 w h D + planes). */
S2P_API int s2p_hip_census_sgm_dev_batch(s2p_hip_ctx* ctx, int n, const float* const* d_im1, const float* const* d_im2, int w, int h,
                                 int dmin, int dmax, const s2p_census_params* params,
                                 float* const* d_disp, float* const* d_conf, uint8_t* const* d_mask);

/* The same batch from / to HOST buffers (arrays of n host pointers; conf / mask may be NULL or hold NULLs): n uploads, the
 * batched launch sequence above, n downloads, one synchronisation.  This is the call the GPU broker issues for the requests of
 * several Pool workers that are waiting at the same time (each worker's s2p.block_matching.compute_disparity_map call,
 * s2p/block_matching.py:155-188, becomes one slot of the batch): byte-identical to n calls of s2p_hip_census_sgm_host
 * (tests/test_gpu_broker.py).  timeout_s as in s2p_hip_census_sgm_host.
 * Transfers: a tile whose five planes lie BACK TO BACK in one block -- im1, im2 = im1 + s, disp = im1 + 2 s, conf = im1 + 3 s,
 * mask = im1 + 4 s with s = 4 w h rounded up to a multiple of 256 bytes, the layout of the broker's arenas -- travels in two copies
 * (inputs up, outputs down) instead of five; the outputs' copy then also writes the alignment gaps (< 256 bytes each) between those
 * planes.  Any other layout keeps one copy per plane and touches nothing but the planes. */
S2P_API int s2p_hip_census_sgm_host_batch(s2p_hip_ctx* ctx, int n, const float* const* im1, const float* const* im2, int w, int h,
                                  int dmin, int dmax, const s2p_census_params* params,
                                  float* const* disp, float* const* conf, uint8_t* const* mask, double timeout_s);
/* The same for n tiles of DIFFERENT sizes and disparity ranges (arrays w[n], h[n], dmin[n], dmax[n]; n <= 16) -- what the tiles of a
 * real job look like: rectified sizes a few pixels apart, a range per tile (s2p/__init__.py:166-196 reads disp_min_max.txt per tile).
 * In the single-scale MGM modes with P2 <= 115 the tiles still share ONE aggregation launch: the volumes get the depth of the widest
 * range (the others are padded with excluded candidates, as rounding the depth up to 16 / 64 always pads), the kernel takes per-tile geometry.
 * Byte-identical to n calls of s2p_hip_census_sgm_host (tests/test_gpu_batch.py); other parameters run the tiles one by one. */
S2P_API int s2p_hip_census_sgm_host_batch_v(s2p_hip_ctx* ctx, int n, const float* const* im1, const float* const* im2,
                                    const int* w, const int* h, const int* dmin, const int* dmax, const s2p_census_params* params,
                                    float* const* disp, float* const* conf, uint8_t* const* mask, double timeout_s);
/* Grow the context's workspace NOW to what a host batch of n such tiles needs (a later, smaller batch then finds it in place):
 * the workspace only ever grows, and growing it frees and reallocates gigabytes -- a device-wide synchronisation that the
 * broker takes once per shape and lane instead of at every new batch size. */
S2P_API int s2p_hip_census_sgm_host_batch_reserve(s2p_hip_ctx* ctx, int n, int w, int h, int dmin, int dmax, const s2p_census_params* params);

typedef struct {
    uint8_t* C;            /* h*w*D0 Hamming cost (buffers sized for D = roundup(subpix*(dmax-dmin)+1, 16) >= D0), 255 = excluded.    */
                           /* (A call WITH dumps lays its volumes out to this multiple of 16 -- the oracle's layout; without, the MGM */
                           /* modes round the depth up to 64 for P2 <= 115: whole lines per pixel, same maps.)                         */
    uint16_t* S;           /* h*w*D sum of the 8 path costs                                     */
    float* disp_raw;       /* h*w after WTA / vfit / L-R                                         */
    float* disp_med;       /* h*w after the median                                               */
    int dmin0, D0;         /* out: first disparity and depth of the C / S layout.  A multi-scale call matches its finest level over */
                           /* the union of its pixels' admissible ranges: D0 <= D of the call, C / S are filled to h*w*D0 entries   */
} s2p_hip_census_dump;

S2P_API int s2p_hip_census_sgm_debug(s2p_hip_ctx* ctx, const float* im1, const float* im2, int w, int h,
                             int dmin, int dmax, const s2p_census_params* params,
                             float* disp, float* conf, uint8_t* mask, s2p_hip_census_dump* dump);

/* ---- homography resampler: the `homography` binary behind s2p.common.image_apply_homography
 * (s2p/common.py:159-180): dst(x) = src(H^-1 x) on the w x h output grid, quintic B-spline.
 * src: sw*sh raster of `src_dtype` (S2P_HIP_F32 / U16 / U8: the GeoTIFF sample types s2p feeds it),
 * H: the 9 row-major coefficients the reference prints into the command line (:177), dst: w*h float32
 * (NaN outside the source domain).  The binary's source is not in the reference tree; algorithm and
 * parity status in oracle/resample_oracle.c. */
enum { S2P_HIP_F32 = 0, S2P_HIP_U16 = 1, S2P_HIP_U8 = 2 };
S2P_API int s2p_hip_warp_host(s2p_hip_ctx* ctx, const void* src, int src_dtype, int sw, int sh,
                      const double H[9], float* dst, int w, int h);
S2P_API int s2p_hip_warp_dev(s2p_hip_ctx* ctx, const void* d_src, int src_dtype, int sw, int sh,
                     const double H[9], float* d_dst, int w, int h);

/* ---- create_rejection_mask on its own (s2p/block_matching.py:18-32), host pointers ------------- */
S2P_API int s2p_hip_rejection_mask_host(s2p_hip_ctx* ctx, const float* disp, const float* im1, const float* im2,
                                int w, int h, uint8_t* mask);

/* ---- masking.erosion (s2p/masking.py:87-97: `morsi disk<radius> erosion msk out`, applied to the
 * rejection mask right after the matcher, s2p/__init__.py:189-190).  mask/out: w*h uint8 (0/1).
 * Structuring element: integer offsets with hypot(i,j) < radius (morsi's source is not in the
 * reference tree: unpinned, see oracle/census_oracle.c). */
S2P_API int s2p_hip_erode_mask_host(s2p_hip_ctx* ctx, const uint8_t* mask, int w, int h, int radius, uint8_t* out);

/* ---- triangulation: disparity map -> (lon, lat, alt) per pixel through two RPC camera models -------
 * Replaces `disp_to_lonlatalt` of the reference's own ctypes library lib/disp_to_h.so
 * (c/disp_to_h.c:70-140, bound at s2p/triangulation.py:117-145); arithmetic of c/rpc.c:279-516.
 * `s2p_rpc` is `struct rpc` of c/rpc.h:13-31 (= RPCStruct of s2p/triangulation.py:23-45), field for field.
 * lonlatalt: ny*nx*3 float64 out, err: ny*nx float32 out (NaN where masked); dispx/dispy/msk: ny*nx
 * float32 (dispy may be NULL = zeros); msk_orig: h*w float32; ha, hb: the rectifying homographies;
 * bbox: col_min, col_max, row_min, row_max of the image domain. */
typedef struct {
    double numx[20], denx[20], numy[20], deny[20], scale[3], offset[3];
    double inumx[20], idenx[20], inumy[20], ideny[20], iscale[3], ioffset[3];
    double dmval[4], imval[4], delta;
} s2p_rpc;

S2P_API int s2p_hip_disp_to_lonlatalt_host(s2p_hip_ctx* ctx, double* lonlatalt, float* err,
                                   const float* dispx, const float* dispy, const float* msk, int nx, int ny,
                                   const float* msk_orig, int w, int h, const double ha[9], const double hb[9],
                                   const s2p_rpc* rpca, const s2p_rpc* rpcb, const float bbox[4]);

/* The reference's exact symbol and argument list (c/disp_to_h.c:70-75), so that s2p/triangulation.py can
 * load this library in place of lib/disp_to_h.so without any other change.  Runs on a process-wide context
 * (device: S2P_HIP_DEVICE, else LOCAL_RANK, else pid mod device count); a failure aborts the process with
 * the HIP error on stderr (the void signature has no error channel, and there is no CPU fallback). */
S2P_API void disp_to_lonlatalt(double* lonlatalt, float* err, float* dispx, float* dispy, float* msk, int nx, int ny,
                       float* msk_orig, int w, int h, double ha[9], double hb[9],
                       s2p_rpc* rpca, s2p_rpc* rpcb, float orig_img_bounding_box[4]);

/* ---- the rest of lib/disp_to_h.so, so that this library replaces it entirely (the four symbols
 * s2p/triangulation.py binds: :117-145, :244-258, :292-299, :324-328) ------------------------------------
 * stereo_corresp_to_lonlatalt (c/disp_to_h.c:43-67): one 3-D point per keypoint match; kp_a, kp_b: n_kp x 2
 *   float32 (x, y); lonlatalt: n_kp x 3 float64; err: n_kp float32.
 * count_3d_neighbors (c/disp_to_h.c:152-174): per pixel of a gridded (ny, nx, 3) float64 cloud, the number of
 *   points of its (2p+1)^2 window closer than r (float32 squared distances, as the reference).
 * remove_isolated_3d_points (c/disp_to_h.c:177-230): in place; points with fewer than n such neighbours become
 *   NaN unless a chain of close (2q+1)^2-window neighbours links them to an accepted point. */
S2P_API int s2p_hip_stereo_corresp_to_lonlatalt_host(s2p_hip_ctx* ctx, double* lonlatalt, float* err, const float* kp_a,
                                                     const float* kp_b, int n_kp, const s2p_rpc* rpca, const s2p_rpc* rpcb);
S2P_API int s2p_hip_count_3d_neighbors_host(s2p_hip_ctx* ctx, int* count, const double* xyz, int nx, int ny, float r, int p);
S2P_API int s2p_hip_remove_isolated_3d_points_host(s2p_hip_ctx* ctx, double* xyz, int nx, int ny, float r, int p, int n, int q);
/* the reference's exact symbols and argument lists (process-wide context, abort on failure: see disp_to_lonlatalt) */
S2P_API void stereo_corresp_to_lonlatalt(double* lonlatalt, float* err, float* kp_a, float* kp_b, int n_kp,
                                         s2p_rpc* rpc_a, s2p_rpc* rpc_b);
S2P_API void count_3d_neighbors(int* count, double* xyz, int nx, int ny, float r, int p);
S2P_API void remove_isolated_3d_points(double* xyz, int nx, int ny, float r, int p, int n, int q);

/* ---- triangulation.height_map, the resampling half (s2p/triangulation.py:376-389) --------------------------
 * After disp_to_xyz the reference carries the altitude plane from the rectified grid to the grid of the original
 * image:  out = ndimage.affine_transform(np.nan_to_num(heights).T, H, output_shape=(w, h), order=1).T, then NaN where
 * the 3x3 binary dilation of the order-0 transform of isnan(heights) is set (H = np.dot(H1, translation(x, y))).
 * heights: hr x wr float64 (NaN = no altitude); H: that 3x3 matrix, bottom row [0, 0, 1] (scipy refuses others);
 * out: h x w float64.  scipy's arithmetic in scipy's order: bit-identical to scipy 1.15's float64 output. */
S2P_API int s2p_hip_height_transfer_host(s2p_hip_ctx* ctx, const double* heights, int wr, int hr, const double H[9],
                                         int w, int h, double* out);

/* ---- triangulation.height_map_to_xyz, the localisation (s2p/triangulation.py:165-219; called by heights_to_ply,
 * s2p/__init__.py:410-413, on the fused height map of a tri-stereo tile) --------------------------------------------
 * heights: h x w float32 (what height_map.tif holds; NaN = no altitude), sampled on the grid of the reference image
 * starting at (off_x, off_y).  lonlatalt: h x w x 3 float64 out: longitude, latitude of image point (c + off_x,
 * r + off_y) at its altitude, and the altitude; NaN triples where the height is NaN.  The reference calls rpcm's
 * RPCModel.localization (pip dependency, not in the tree); this entry inverts the projection with the iteration of
 * c/rpc.c:378-439, as disp_to_lonlatalt does: same equation, agreement at the 1e-9 pixel level, not bitwise.  The CRS
 * conversion that follows stays in Python (s2p_amd/geographiclib.py for UTM, pyproj otherwise). */
S2P_API int s2p_hip_height_map_to_lonlatalt_host(s2p_hip_ctx* ctx, const s2p_rpc* rpc, const float* heights, int w, int h,
                                                 int off_x, int off_y, double* lonlatalt);

/* ---- common.cargarse_basura (s2p/common.py:224-235), the outlier filter heights_fusion runs on every pair's height
 * map before merge_n when cfg['cargarse_basura'] is set (s2p/__init__.py:362-365): six subprocesses in the reference
 * (morphoop min / max with a 5 x 5 square, c/morphoop.c; plambda "x y - fabs 5 > nan z if"; remove_small_cc 200 5).
 * in / out: h x w float32 (may alias).  NaN where the 5 x 5 local range exceeds 5, then 4-connected components (edges:
 * |difference| < 5) of fewer than 200 pixels -> NaN.  remove_small_cc's source is not in the reference tree: that
 * stage is unpinned (oracle/cleanup_oracle.c).  Maps smaller than 3 x 3: S2P_HIP_UNSUPPORTED. */
S2P_API int s2p_hip_cargarse_basura_host(s2p_hip_ctx* ctx, const float* in, int w, int h, float* out);

/* ---- fusion.merge_n (s2p/fusion.py:26-68): pixelwise merge of n co-registered height maps -------------
 * inputs: n pointers to h*w float32 maps; offsets: n doubles subtracted before merging (their mean is added
 * back); op: 0 average_if_close (s2p/fusion.py:16-23, with `threshold`), 1 np.nanmedian, 2 np.median,
 * 3 np.nanmean, 4 np.mean, 5 np.nanmin, 6 np.nanmax, 7 np.min, 8 np.max; float64 arithmetic in numpy's
 * evaluation order, float32 out.  n <= 64. */
S2P_API int s2p_hip_merge_n_host(s2p_hip_ctx* ctx, const float* const* inputs, const double* offsets, int n, int w, int h,
                                 int op, double threshold, float* out);

/* ---- DSM rasterisation: `rasterize_cloud` of the plyflatten package ------------------------------------------
 * s2p rasterises the tiles' point clouds with plyflatten (s2p/__init__.py:31 import, :462-466
 * plyflatten_from_plyfiles_list(clouds, resolution, roi, radius, sigma); tests/rasterization_test.py:13-28).
 * plyflatten >= 0.2.0 is a pip dependency (setup.py:52), not vendored in the reference tree; its Python front end
 * concatenates the clouds to an (n, 2 + nb) float64 array of x, y and nb values per point and calls the C entry
 * `rasterize_cloud(input_buffer, raster, nb_points, nb_extra_columns, xoff, yoff, resolution, xsize, ysize, radius,
 * sigma)` of its libplyflatten.so through ctypes.  Both entries below take exactly those arguments; `raster` is
 * ysize x xsize x nb_extra_columns float32, cell (row j, column i) covering x in xoff + [i, i + 1) resolution and
 * y in yoff - (j, j + 1] resolution; cells without points get NaN.  Per cell the points are folded IN INPUT ORDER
 * into the running weighted float32 mean of the C code (weight 1 for sigma = inf), so the result has the bits of
 * the CPU code: pinned on the reference's golden dsm_40cm.tiff for radius 0 (oracle/rasterize_oracle.c).
 * nb_extra_columns <= 16, radius <= 64, fewer than 2^31 cells and point-cell contributions; a point whose x or y is
 * not finite contributes nothing. */
S2P_API int s2p_hip_plyflatten_host(s2p_hip_ctx* ctx, const double* cloud, int nb_points, int nb_extra_columns,
                                    double xoff, double yoff, double resolution, int xsize, int ysize,
                                    int radius, float sigma, float* raster);
/* drop-in for the symbol of plyflatten's own shared library (process-wide context; errors abort with a message as a
 * failing C library would) */
S2P_API void rasterize_cloud(double* input_buffer, float* raster, int nb_points, int nb_extra_columns,
                             double xoff, double yoff, double resolution, int xsize, int ysize, int radius, float sigma);

/* ---- one tile end to end in one call: rectify -> match -> mask/erode -> triangulate ---------------
 * the reference hands a tile from step to step through files
 * (rectified_ref/sec.tif -> rectified_disp.tif + rectified_mask.png -> the point cloud;
 * s2p/__init__.py:147-159,178-190,213-233), one subprocess or pool task per step.  This entry keeps the
 * tile in HBM between the steps; every intermediate is byte-identical to what the separate entry points
 * return (tests/test_gpu_tile_pipeline.py).  Host pointers in, host pointers out, one synchronisation.
 *   src1/src2   windows of the two images (dtype 0 f32 / 1 u16 / 2 u8), with H1/H2 mapping window
 *               coordinates to the rectified frame (what image_apply_homography passes to `homography`)
 *   algo        0 = sgbm (s2p_sgbm_params), 1 = census/SGM (s2p_census_params); NULL params = defaults
 *   erosion     masking.erosion radius applied to the rejection mask (s2p/__init__.py:189-190); 0 = none
 *   rpca/rpcb   both NULL = stop after the mask; otherwise disp_to_lonlatalt with ha, hb, msk_orig
 *               (oh x ow float32) and bbox as in s2p_hip_disp_to_lonlatalt_host
 * Outputs (each may be NULL = not copied back): rect1, rect2, disp: h*w float32; mask: h*w uint8 (after
 * erosion); lonlatalt: h*w*3 float64; err: h*w float32. */
typedef struct {
    const void* src1; int src1_dtype, sw1, sh1; double H1[9];
    const void* src2; int src2_dtype, sw2, sh2; double H2[9];
    int w, h, dmin, dmax;
    int algo;
    const s2p_sgbm_params* sgbm;
    const s2p_census_params* census;
    int erosion;
    const s2p_rpc* rpca; const s2p_rpc* rpcb;
    double ha[9], hb[9];
    const float* msk_orig; int ow, oh;
    float bbox[4];
} s2p_tile;
typedef struct { float* rect1; float* rect2; float* disp; uint8_t* mask; double* lonlatalt; float* err; } s2p_tile_out;
S2P_API int s2p_hip_tile_host(s2p_hip_ctx* ctx, const s2p_tile* tile, const s2p_tile_out* out, double timeout_s);
/* n tiles of ONE shape (same w, h, dmin, dmax, algo = 1 and census parameters; windows, homographies, erosion and the
 * triangulation inputs are per tile) in one call: every tile is rectified, the n pairs are matched by the batched
 * launch sequence of s2p_hip_census_sgm_dev_batch (one k_mgm_bands launch for the n tiles: the chip runs full instead
 * of following one tile's dependency chain), then masks / erosion / triangulation per tile and the copies back; one
 * synchronisation.  This is how a tile worker of the reference's Pool (s2p/parallel.py:58-110) becomes a worker that
 * takes several tiles of the queue at a time (s2p_amd/tiles.py: process_queue(batch=...)).  Every output is
 * byte-identical to n calls of s2p_hip_tile_host (tests/test_gpu_tile_batch.py).  n <= 64; tiles that differ in matcher or
 * parameters: S2P_HIP_BAD_ARGUMENT (group them on the caller's side).  Tiles of different sizes / disparity ranges are taken
 * (since round 4, n <= 16) where the matcher has a launch for them -- the single-scale MGM modes with P2 <= 115, see
 * s2p_hip_census_sgm_host_batch_v -- and refused with S2P_HIP_BAD_ARGUMENT otherwise; n = 1 is s2p_hip_tile_host. */
S2P_API int s2p_hip_tile_host_batch(s2p_hip_ctx* ctx, int n, const s2p_tile* tiles, const s2p_tile_out* outs, double timeout_s);

/* ---- per-kernel timing (HIP events on the context stream) ------------------------------------ */
/* When enabled, every stage of the next calls is bracketed by hipEvents recorded on the stream the
 * kernels are launched on.  s2p_hip_timing_get returns the accumulated milliseconds and launch
 * count of a stage since the last reset ("quantize", "cost", "aggregate", "wta", "median",
 * "speckle", "epilogue", "total"); it synchronises the stream. */
S2P_API int s2p_hip_timing_enable(s2p_hip_ctx* ctx, int on);
S2P_API int s2p_hip_timing_reset(s2p_hip_ctx* ctx);
S2P_API int s2p_hip_timing_get(s2p_hip_ctx* ctx, const char* stage, double* ms, int* launches);

#ifdef __cplusplus
}
#endif
#endif /* S2P_HIP_H */
