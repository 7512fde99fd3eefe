// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE, not product code.
//
// Builds the *real* reference matcher (vendored OpenCV-2.4 StereoSGBM + the s2p
// `sgbm` driver) into oracle/_ref/libsgbm_ref.so so that
//   (a) golden vectors under tests/golden/ can be generated from the reference itself, and
//   (b) oracle/sgbm_oracle.c (our own CPU restatement) can be pinned bit-for-bit against it.
//
// The reference sources are compiled WHERE THEY LIE under /root/reference/3rdparty/sgbm
// (see oracle/Makefile); nothing is copied into this repository.  This translation unit
// and ref_driver_tu.cpp textually include two reference files:
//   * sgbm.cpp       -- for qauto / qeasy / paste (3rdparty/sgbm/sgbm.cpp:30-71,133-137).
//                       Its `main` is renamed to an unused static function (ref_driver_tu.cpp), because
//                       it needs iio.c, which cannot be compiled here (tiffio.h / png.h / jpeglib.h
//                       are not installed).  The ~40 lines of argv/sign/crop glue of that main
//                       (sgbm.cpp:139-241) are therefore restated in ref_run() below, line-cited.
//   * stereosgbm.cpp -- the whole matcher, so that the `static` computeDisparitySGBM is reachable
//                       for intermediate dumps (C, S, raw disparity).
// No stand-in headers/libraries are used: system.cpp/parallel.cpp are compiled with
// `-U__linux__ -include unistd.h` (real system header) to skip their vestigial <sys/sysctl.h>.
//
// The reference reads uninitialised heap memory (canvas margins, never-written rows of C, the flat
// row scratch).  For realistic tile sizes those bytes come from fresh
// zero mmap pages; to make that deterministic at every size, every malloc() issued from inside this
// library (cv::fastMalloc in alloc.cpp, qauto/qeasy) is bound to the zero-filling wrapper below
// (-Wl,-Bsymbolic-functions => bound inside this .so only; the rest of the process keeps
// libc's malloc).  glibc's M_PERTURB is NOT reliable for this: tcache hits bypass alloc_perturb.
// "uninitialised == 0" is the definition every golden vector is generated with.

#include <stdlib.h>
#include <stdint.h>
#include <string.h>

#include "stereosgbm.cpp"   // reference matcher, textually, so `static computeDisparitySGBM` is reachable

using namespace cv;
// defined by the reference driver sgbm.cpp, compiled in ref_driver_tu.cpp
uint8_t *qauto(float *x, int w, int h, int pd, float *rmin, float *rmax);
uint8_t *qeasy(float *x, int w, int h, int pd, float black, float white);
void paste(Mat &dest, Mat &overlay, int px, int py);

extern "C" void* malloc(size_t n) { return calloc(1, n); }

extern "C" {

struct s2p_ref_dump {
    // all optional (NULL = skip); sizes in elements
    uint8_t* q1;        // w*h quantised im1
    uint8_t* q2;        // w*h quantised im2
    int16_t* C;         // height*width1*D block cost (+P2 bias), layout [y][x][d]
    int16_t* S;         // height*width1*D aggregated cost
    int16_t* disp_raw;  // height*Wc  canvas disparity (x16) before median
    int16_t* disp_med;  // height*Wc  after 3x3 median
    int16_t* disp_fin;  // height*Wc  after speckle filter
    int16_t* cost_raw;  // height*Wc  canvas cost
    int geom[8];        // out: Wc, width1, D, minD, x0, minX1, maxX1, INVALID_SCALED
    float rminmax[2];   // out
};

// Mirrors `sgbm im1 im2 disp cost dmin dmax win P1 P2 lr` (s2p/block_matching.py:128-132).
int s2p_ref_sgbm(const float* im1, const float* im2, int w, int h,
                 int dmin, int dmax, int SADwin, int P1, int P2, int LRdiff,
                 float* odisp, float* ocost, s2p_ref_dump* dump)
{
    int rc = 0;
    {
        // sgbm.cpp:153-155
        float rmin, rmax;
        uint8_t* im1_quantized = qauto((float*)im1, w, h, 1, &rmin, &rmax);
        uint8_t* im2_quantized = qeasy((float*)im2, w, h, 1, rmin, rmax);
        if (dump) { dump->rminmax[0] = rmin; dump->rminmax[1] = rmax; }
        if (dump && dump->q1) memcpy(dump->q1, im1_quantized, (size_t)w*h);
        if (dump && dump->q2) memcpy(dump->q2, im2_quantized, (size_t)w*h);
        // sgbm.cpp:158-159
        Mat u1(h, w, CV_8UC1, (void*)im1_quantized);
        Mat u2(h, w, CV_8UC1, (void*)im2_quantized);
        // sgbm.cpp:166-168 (argv[5] = dmin, argv[6] = dmax, both in s2p convention)
        int maxdisp = -dmin;
        int mindisp = -dmax;
        if (mindisp >= maxdisp) { rc = 1; goto done_q; }   // sgbm.cpp:174-177 (exit(1))
        {
            // sgbm.cpp:180-192
            StereoSGBM sgbm;
            sgbm.numberOfDisparities = (int)16 * ceil((maxdisp - mindisp) / 16.0);
            sgbm.minDisparity = mindisp;
            sgbm.SADWindowSize = SADwin;
            sgbm.P1 = P1;
            sgbm.P2 = P2;
            sgbm.disp12MaxDiff = LRdiff;
            sgbm.fullDP = 1;
            sgbm.preFilterCap = 63;
            sgbm.uniquenessRatio = 10;
            sgbm.speckleWindowSize = 50;
            sgbm.speckleRange = 1;
            // sgbm.cpp:204-207 ("crop trick")
            int x0 = max(maxdisp, 0);
            Mat uu1(u1.rows, (int)(u1.cols + max(-mindisp, 0) + max(maxdisp, 0)), u1.type());
            Mat uu2(u2.rows, (int)(u2.cols + max(-mindisp, 0) + max(maxdisp, 0)), u2.type());
            paste(uu1, u1, x0, 0);
            paste(uu2, u2, x0, 0);
            Mat ddisp, ccost;
            if (!dump) {
                sgbm(uu1, uu2, ddisp, ccost);   // sgbm.cpp:209
            } else {
                // Same statements as StereoSGBM::operator() (stereosgbm.cpp:828-846), unrolled so
                // that the intermediates can be copied out.
                Mat buffer;
                ddisp.create(uu1.size(), CV_16S);
                ccost.create(uu1.size(), CV_16S);
                computeDisparitySGBM(uu1, uu2, ddisp, ccost, sgbm, buffer);
                int minD = sgbm.minDisparity, maxD = minD + sgbm.numberOfDisparities;
                int Wc = uu1.cols;
                int minX1 = max(-maxD, 0), maxX1 = Wc + min(minD, 0);
                int D = maxD - minD, width1 = maxX1 - minX1;
                dump->geom[0] = Wc; dump->geom[1] = width1; dump->geom[2] = D; dump->geom[3] = minD;
                dump->geom[4] = x0; dump->geom[5] = minX1; dump->geom[6] = maxX1;
                dump->geom[7] = (minD - 1) * 16;
                if (width1 > 0 && buffer.data) {
                    size_t n = (size_t)width1 * D * h;
                    short* Cbuf = (short*)alignPtr(buffer.data, 16);   // stereosgbm.cpp:383-384
                    short* Sbuf = Cbuf + n;
                    if (dump->C) memcpy(dump->C, Cbuf, n * sizeof(short));
                    if (dump->S) memcpy(dump->S, Sbuf, n * sizeof(short));
                }
                size_t nc = (size_t)Wc * h;
                if (dump->disp_raw) for (int y = 0; y < h; y++) memcpy(dump->disp_raw + (size_t)y*Wc, ddisp.ptr<short>(y), Wc*sizeof(short));
                if (dump->cost_raw) for (int y = 0; y < h; y++) memcpy(dump->cost_raw + (size_t)y*Wc, ccost.ptr<short>(y), Wc*sizeof(short));
                medianBlur(ddisp, ddisp, 3);
                if (dump->disp_med) for (int y = 0; y < h; y++) memcpy(dump->disp_med + (size_t)y*Wc, ddisp.ptr<short>(y), Wc*sizeof(short));
                Mat buf2;
                filterSpeckles(ddisp, (sgbm.minDisparity - 1) * StereoSGBM::DISP_SCALE, sgbm.speckleWindowSize,
                               StereoSGBM::DISP_SCALE * sgbm.speckleRange, buf2);
                if (dump->disp_fin) for (int y = 0; y < h; y++) memcpy(dump->disp_fin + (size_t)y*Wc, ddisp.ptr<short>(y), Wc*sizeof(short));
                (void)nc;
            }
            // sgbm.cpp:210-211
            Mat disp = ddisp(Range::all(), Range(x0, x0 + u1.cols));
            Mat cost = ccost(Range::all(), Range(x0, x0 + u1.cols));
            // sgbm.cpp:220-234
            for (int y = 0; y < disp.rows; y++)
                for (int x = 0; x < disp.cols; x++)
                    if (disp.at<int16_t>(y, x) == -((-mindisp + 1) * 16)) {
                        odisp[x + y*disp.cols] = NAN;
                        ocost[x + y*disp.cols] = NAN;
                    } else {
                        odisp[x + y*disp.cols] = -((float)disp.at<int16_t>(y, x)) / 16.0;
                        ocost[x + y*disp.cols] = (float)cost.at<int16_t>(y, x);
                    }
        }
    done_q:
        free(im1_quantized);
        free(im2_quantized);
    }
    return rc;
}

} // extern "C"
