// s2p_amd/csrc/sgbm_kernels.hip -- hand-written gfx950 kernels for the `sgbm` matcher.
//
// Bit-exact MI355X re-design of what the reference computes in
//   3rdparty/sgbm/sgbm.cpp:30-71,139-241          (quantisation, crop trick, sign conventions)
//   3rdparty/sgbm/stereosgbm.cpp:115-280          (Birchfield-Tomasi pixel cost on Sobel-x prefiltered rows)
//   3rdparty/sgbm/stereosgbm.cpp:392-516          (3x3 block cost C)
//   3rdparty/sgbm/stereosgbm.cpp:518-662          (8-path semi-global aggregation)
//   3rdparty/sgbm/stereosgbm.cpp:664-816          (WTA, uniqueness, disp2, parabola, L-R check)
//   3rdparty/sgbm/smooth.cpp:207-292, stereosgbm.cpp:872-967   (3x3 median, speckle filter)
// NOT a translation of that code: the CPU runs two row sweeps that carry 4 directions each; here
// every one of the 8 directions is an independent set of 1-D recurrences ("paths"), all 8 sets run
// concurrently in ONE launch, a path is owned by a group of G lanes of a 64-wide wavefront (8
// disparities per lane, packed 2 x int16 per VGPR), the d+-1 neighbours travel by DPP row shifts
// and the per-pixel minimum by a DPP xor-butterfly.  No MFMA: this is a min/add recurrence.
//
// HBM layout (all row-major, d fastest):
//   C  [h][width1][D]      int16   block cost + P2 bias                       (2 B / candidate)
//   E_r[h][width1][D]      uint8   r = 0..7; e = C - L_r  in [0, P2]          (1 B / candidate / path)
// S = sum_r L_r = 8*C - sum_r e_r is never materialised: the WTA kernel rebuilds it on the fly.
// e fits a byte because L_r = C + min(Lp[d], Lp[d+-1]+P1, minLp+P2) - (minLp+P2) and the min is
// within [minLp, minLp+P2].
#include "common.hpp"
#include "agg.hpp"
#include "ccl.hpp"

#include <algorithm>

namespace s2p {

// =============================================================================================
// K0: exact order statistics of im1 (sgbm.cpp:30-42 get_rminmax: full qsort on the CPU).
// Radix select on order-preserving uint32 keys, 3 passes of 11/11/10 bits, two ranks at once.
// =============================================================================================
__device__ __forceinline__ uint32_t float_key(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_float(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

// pass: 0 -> bits [31:21] (2048 bins, single histogram), 1 -> bits [20:10] (2048 bins x2), 2 -> bits [9:0] (1024 x2)
template <int PASS>
__global__ __launch_bounds__(256) void k_select_hist(const float* __restrict__ im, size_t n,
                                                     const SelectState* __restrict__ st, uint32_t* __restrict__ hist)
{
    __shared__ uint32_t sh[2 * 2048];
    constexpr int NB = PASS == 2 ? 1024 : 2048;
    constexpr int NH = PASS == 0 ? 1 : 2;
    for (int i = threadIdx.x; i < NB * NH; i += 256) sh[i] = 0;
    __syncthreads();
    uint32_t p0 = 0, p1 = 0;
    if (PASS > 0) { p0 = st->prefix[0]; p1 = st->prefix[1]; }
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float f = im[i];
        if (f != f) continue;
        uint32_t k = float_key(f);
        if (PASS == 0) atomicAdd(&sh[k >> 21], 1u);
        else if (PASS == 1) {
            if ((k >> 21) == (p0 >> 21)) atomicAdd(&sh[(k >> 10) & 2047], 1u);
            if ((k >> 21) == (p1 >> 21)) atomicAdd(&sh[2048 + ((k >> 10) & 2047)], 1u);
        } else {
            if ((k >> 10) == (p0 >> 10)) atomicAdd(&sh[k & 1023], 1u);
            if ((k >> 10) == (p1 >> 10)) atomicAdd(&sh[1024 + (k & 1023)], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NB * NH; i += 256)
        if (sh[i]) atomicAdd(&hist[i], sh[i]);
}

template <int PASS>
__global__ __launch_bounds__(256) void k_select_pick(SelectState* st, uint32_t* hist)
{
    // one block; both ranks handled together: 256 threads x (NB/256) bins, LDS scan of the partials
    constexpr int NB = PASS == 2 ? 1024 : 2048;
    constexpr int SHIFT = PASS == 0 ? 21 : PASS == 1 ? 10 : 0;
    constexpr int PER = NB / 256;
    __shared__ uint32_t part[2][256];
    __shared__ uint32_t tot[2];
    __shared__ uint32_t s_n, s_rank[2], s_prefix[2];     // snapshot of the state: it is overwritten below
    const int t = threadIdx.x;
    if (t == 0) {
        s_n = PASS == 0 ? 0u : st->n;
        for (int s = 0; s < 2; s++) { s_rank[s] = PASS == 0 ? 0u : st->rank[s]; s_prefix[s] = PASS == 0 ? 0u : st->prefix[s]; }
    }
    uint32_t loc[2][PER];
    #pragma unroll
    for (int s = 0; s < 2; s++) {
        const uint32_t* h = hist + (PASS == 0 ? 0 : s * NB);
        uint32_t a = 0;
        #pragma unroll
        for (int j = 0; j < PER; j++) { loc[s][j] = h[t * PER + j]; a += loc[s][j]; }
        part[s][t] = a;
    }
    __syncthreads();
    if (t < 2) {   // exclusive scan of 256 partials (serial, LDS only)
        uint32_t c = 0;
        for (int i = 0; i < 256; i++) { uint32_t v = part[t][i]; part[t][i] = c; c += v; }
        tot[t] = c;
    }
    __syncthreads();
    #pragma unroll
    for (int s = 0; s < 2; s++) {
        const uint32_t n = PASS == 0 ? tot[0] : s_n;
        uint32_t rank;
        if (PASS == 0) { uint32_t rb = n / 200; rank = (s == 0) ? rb : (n ? n - 1 - rb : 0); }
        else rank = s_rank[s];
        if (n == 0) { if (t == 0) { st->n = 0; st->rank[s] = 0; st->prefix[s] = 0; st->rminmax[s] = 0.f; } continue; }
        uint32_t cum = part[s][t];
        #pragma unroll
        for (int j = 0; j < PER; j++) {
            if (rank >= cum && rank < cum + loc[s][j]) {          // exactly one (thread, j) matches
                uint32_t pre = s_prefix[s] | ((uint32_t)(t * PER + j) << SHIFT);
                st->rank[s] = rank - cum;
                st->prefix[s] = pre;
                if (PASS == 2) st->rminmax[s] = key_float(pre);
                if (PASS == 0 && s == 0) st->n = n;
            }
            cum += loc[s][j];
        }
    }
}

__global__ void k_zero_u32(uint32_t* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = 0; }

// =============================================================================================
// K1: 8-bit requantisation of both images with im1's thresholds, pasted into zero canvases
// (sgbm.cpp:44-71 qauto/qeasy; :204-207 crop trick).  float32 arithmetic exactly as the reference:
// floor(255 * (g - rmin) / (rmax - rmin)), IEEE division, no contraction.  NaN -> 0.
// =============================================================================================
__global__ __launch_bounds__(256) void k_quantize_paste(const float* __restrict__ im1, const float* __restrict__ im2,
                                                        int w, int h, int Wc, int x0, const SelectState* __restrict__ st,
                                                        uint8_t* __restrict__ uu1, uint8_t* __restrict__ uu2)
{
    int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= Wc) return;
    float rmin = st->rminmax[0], rmax = st->rminmax[1];
    float den = __fsub_rn(rmax, rmin);
    int xs = x - x0;
    uint8_t a = 0, b = 0;
    if (xs >= 0 && xs < w) {
        float g1 = im1[(size_t)y * w + xs], g2 = im2[(size_t)y * w + xs];
        float q1 = floorf(__fdiv_rn(__fmul_rn(255.0f, __fsub_rn(g1, rmin)), den));
        float q2 = floorf(__fdiv_rn(__fmul_rn(255.0f, __fsub_rn(g2, rmin)), den));
        if (q1 < 0) q1 = 0; if (q1 > 255) q1 = 255;
        if (q2 < 0) q2 = 0; if (q2 > 255) q2 = 255;
        a = (q1 != q1) ? 0 : (uint8_t)q1;
        b = (q2 != q2) ? 0 : (uint8_t)q2;
    }
    uu1[(size_t)y * Wc + x] = a;
    uu2[(size_t)y * Wc + x] = b;
}

// =============================================================================================
// K2a: per-row prefilter + Birchfield-Tomasi half-sample bounds (stereosgbm.cpp:125-147,188-205).
// One block per canvas row.  The reference keeps these in one flat byte scratch and, for
// disparities that point outside image 2, reads past a row into whatever follows it (SURVEY.md
// App. A.3).  The block rebuilds that flat scratch in LDS for both observation states (during the
// prefiltered channel's x-loop v0/v1 hold channel-0 values, during the raw channel's loop channel-1
// values) and emits, per row, arrays indexed by the reference's flat index idx = Wc-1-x+d with the
// two channels PACKED into one dword (lo half = prefiltered, hi half = raw channel, both <= 255):
//   vpk[y][0][idx] = v   vpk[y][1][idx] = v0   vpk[y][2][idx] = v1        idx in [0, NI)
//   upk[y][0][x]   = u   upk[y][1][x]   = u0   upk[y][2][x]   = u1        x   in [0, Wc)
// so that the cost kernel evaluates both channels with single packed-int16 instructions and
// reproduces every out-of-range read bit for bit.
// flat layout (byte offsets after `guard` zeros): [v0 (width2) | v1 (width2) | prow1 pref | prow1 raw |
// prow2 pref MIRRORED | prow2 raw MIRRORED | zeros up to fl].
// =============================================================================================
__device__ __forceinline__ int clip_tab(int v, int ftzero) { return min(max(v, -ftzero), ftzero) + ftzero; }

__global__ __launch_bounds__(256) void k_prefilter(const uint8_t* __restrict__ uu1, const uint8_t* __restrict__ uu2,
                                                   Geom g, int ftzero, int NI, uint32_t* __restrict__ vpk, uint32_t* __restrict__ upk)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t sm[];
    const int Wc = g.Wc, y = blockIdx.x;
    uint8_t* f0 = sm;                 // flat image observed during channel 0   [fl]
    uint8_t* f1 = sm + g.fl;          // flat image observed during channel 1   [fl]
    for (int i = threadIdx.x; i < 2 * g.fl; i += 256) sm[i] = 0;
    __syncthreads();
    const int P1OFF = g.guard + 2 * g.width2, P2OFF = P1OFF + 2 * Wc;
    const uint8_t* r1 = uu1 + (size_t)y * Wc;
    const uint8_t* r2 = uu2 + (size_t)y * Wc;
    const int n = y > 0 ? -Wc : 0, s = y < g.h - 1 ? Wc : 0;
    const int t0 = clip_tab(0, ftzero);
    for (int x = threadIdx.x; x < Wc; x += 256) {
        int a0, a1, b0, b1;
        if (x == 0 || x == Wc - 1) { a0 = a1 = b0 = b1 = t0; }      // :129-133 (both channels!)
        else {
            a0 = clip_tab((r1[x + 1] - r1[x - 1]) * 2 + r1[x + n + 1] - r1[x + n - 1] + r1[x + s + 1] - r1[x + s - 1], ftzero);
            b0 = clip_tab((r2[x + 1] - r2[x - 1]) * 2 + r2[x + n + 1] - r2[x + n - 1] + r2[x + s + 1] - r2[x + s - 1], ftzero);
            a1 = r1[x]; b1 = r2[x];
        }
        f0[P1OFF + x] = f1[P1OFF + x] = (uint8_t)a0;
        f0[P1OFF + Wc + x] = f1[P1OFF + Wc + x] = (uint8_t)a1;
        f0[P2OFF + Wc - 1 - x] = f1[P2OFF + Wc - 1 - x] = (uint8_t)b0;
        f0[P2OFF + Wc + Wc - 1 - x] = f1[P2OFF + Wc + Wc - 1 - x] = (uint8_t)b1;
    }
    __syncthreads();
    for (int c = 0; c < 2; c++) {                                      // :191-200
        uint8_t* f = c ? f1 : f0;
        const uint8_t* q2 = f + P2OFF + c * Wc;
        for (int i = g.minX2 + threadIdx.x; i < g.maxX2; i += 256) {
            int v = q2[i];
            int vl = i > 0 ? (v + q2[i - 1]) >> 1 : v;
            int vr = i < Wc - 1 ? (v + q2[i + 1]) >> 1 : v;
            f[g.guard + i - g.minX2] = (uint8_t)min(min(vl, vr), v);
            f[g.guard + i - g.minX2 + g.width2] = (uint8_t)max(max(vl, vr), v);
        }
    }
    __syncthreads();
    uint32_t* vrow = vpk + (size_t)y * 3 * NI;
    for (int idx = threadIdx.x; idx < NI; idx += 256) {
        const int j = g.guard + idx - g.minX2;                          // may reach into the guard (zeros)
        vrow[idx] = (uint32_t)f0[P2OFF + idx] | ((uint32_t)f1[P2OFF + Wc + idx] << 16);
        vrow[NI + idx] = (uint32_t)f0[j] | ((uint32_t)f1[j] << 16);
        vrow[2 * NI + idx] = (uint32_t)f0[j + g.width2] | ((uint32_t)f1[j + g.width2] << 16);
    }
    uint32_t* urow = upk + (size_t)y * 3 * Wc;
    for (int x = threadIdx.x; x < Wc; x += 256) {                      // :204-208
        uint32_t o[3];
        #pragma unroll
        for (int c = 0; c < 2; c++) {
            const uint8_t* q1 = f0 + P1OFF + c * Wc;
            int v = q1[x];
            int vl = x > 0 ? (v + q1[x - 1]) >> 1 : v;
            int vr = x < Wc - 1 ? (v + q1[x + 1]) >> 1 : v;
            int lo = min(min(vl, vr), v), hi = max(max(vl, vr), v);
            if (c == 0) { o[0] = v; o[1] = lo; o[2] = hi; }
            else { o[0] |= (uint32_t)v << 16; o[1] |= (uint32_t)lo << 16; o[2] |= (uint32_t)hi << 16; }
        }
        urow[x] = o[0]; urow[Wc + x] = o[1]; urow[2 * Wc + x] = o[2];
    }
}

// =============================================================================================
// K2b: BT pixel cost + 3x3 block sum -> C[y][x][d] (int16, +P2), fused: one block owns a strip of
// XS columns x YC rows, keeps a 3-row ring of pixel costs in LDS, never writes pixDiff to HBM.
// The pixel cost depends on image 1 only through x and on image 2 only through idx = Wc-1-x+d: a
// thread owns ONE idx (its v/v0/v1 triple lives in 3 VGPRs, both channels packed) and walks the
// strip's columns, reading the column's u/u0/u1 triple as an LDS broadcast; each evaluation is 7
// packed-int16 instructions for both channels (unsigned saturating sub, max, min).
// Reference quirks reproduced: C(y>0, x=0, .) = 0 (Q1), C(h-1, ., .) = 0 (Q2),
// replicate borders of the box sum, top row counted twice.
// =============================================================================================
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pku_subsat(uint32_t a, uint32_t b) {
    u16x2 r = __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b));
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t pku_max(uint32_t a, uint32_t b) {
    u16x2 r = __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b));
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t pku_min(uint32_t a, uint32_t b) {
    u16x2 r = __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b));
    return __builtin_bit_cast(uint32_t, r);
}

struct CostArgs {
    Geom g;
    const uint32_t* vpk; const uint32_t* upk; int NI;
    int16_t* C;
    int XS, YC, P2, WL;    // WL: LDS window length (XS + 2 + D, rounded up to 4)
};

__global__ __launch_bounds__(256) void k_block_cost(CostArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t sm[];
    const Geom& g = a.g;
    const int D = g.D, XS = a.XS, NXL = XS + 2, WL = a.WL;
    uint32_t* U = reinterpret_cast<uint32_t*>(sm);      // [NXL][4]: u, u0, u1 (both channels packed), pad
    uint32_t* V = U + NXL * 4;                           // [3][WL]: v, v0, v1 windows (both channels packed)
    // pixel-cost ring [3][NXL][QS] of uint2: the costs of 4 consecutive d as two dwords of 16-bit fields,
    // (d0, d2) and (d1, d3), so that the 3x3 block sum below is 18 plain dword adds with no unpacking.
    // QS = D/4 + 1 (odd): the column-per-lane 8-byte stores of the pixel-cost phase fall on distinct banks.
    uint2* P = reinterpret_cast<uint2*>(V + 3 * WL);
    const int QS = (D >> 2) + 1;
    const int xs0 = blockIdx.x * XS;                    // first column (width1 coordinates)
    const int y0 = blockIdx.y * a.YC;
    const int y1 = min(y0 + a.YC, g.h);                 // rows [y0, y1)
    const int xend = min(xs0 + XS, g.width1);
    // halo columns, clamped => replicate borders of the horizontal 3-sum (:449,461-462)
    const int xlo = max(xs0 - 1, 0), xhi = min(xend, g.width1 - 1);       // width1 coords
    const int idx_lo = g.Wc - 1 - (xhi + g.minX1) + g.minD;                 // flat index of (xhi, d=0)
    const int wl_used = (xhi - xlo) + D;
    const int kfirst = max(y0 - 1, 0), klast = min(y1, g.h - 1);
    const int tid = threadIdx.x;
    const int quarterD = D >> 2;

    for (int k = kfirst; k <= klast; k++) {
        __syncthreads();
        {   // ---- stage the u triples of the strip's columns and the v/v0/v1 windows of row k
            const uint32_t* u = a.upk + (size_t)k * 3 * g.Wc + g.minX1;
            for (int i = tid; i < NXL * 3; i += 256) {
                int xl = i / 3, q = i - xl * 3;
                int x = min(max(xs0 - 1 + xl, 0), g.width1 - 1);
                U[xl * 4 + q] = u[q * g.Wc + x];
            }
            const uint32_t* vrow = a.vpk + (size_t)k * 3 * a.NI + idx_lo;
            for (int i = tid; i < wl_used; i += 256) {
                V[i] = vrow[i]; V[WL + i] = vrow[a.NI + i]; V[2 * WL + i] = vrow[2 * a.NI + i];
            }
        }
        __syncthreads();
        // ---- pixel cost of row k for the strip + halo: P[k%3][xl][d / 4], 4 consecutive d per thread
        // (one conflict-free 8-byte store); cost(x, d) = f(U[x], V[(xhi - x) + d])
        uint2* Pk = P + (size_t)(k % 3) * NXL * QS;
        // lanes run over the columns (xl fastest): the v windows are then read at consecutive addresses
        // (index (xhi - x) + dq: stride -1 across lanes) instead of stride 4 (a 4-way bank conflict on each of
        // the 12 reads), and the odd row stride makes the stores conflict-free as well
        {
            int xl = tid % NXL, dqi = tid / NXL;
            const int dxl = 256 % NXL, ddq = 256 / NXL;
            for (; dqi < quarterD; ) {
                const int dq = dqi * 4;
                const int x = min(max(xs0 - 1 + xl, 0), g.width1 - 1);
                const int i0 = (xhi - x) + dq;
                const uint4 uu = *reinterpret_cast<const uint4*>(U + xl * 4);         // u, u0, u1
                uint32_t ev = 0, od = 0;
                #pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t v = V[i0 + j], v0 = V[WL + i0 + j], v1 = V[2 * WL + i0 + j];
                    const uint32_t c0 = pku_max(pku_subsat(uu.x, v1), pku_subsat(v0, uu.x));   // max(0, u - v1, v0 - u)
                    const uint32_t c1 = pku_max(pku_subsat(v, uu.z), pku_subsat(uu.y, v));     // max(0, v - u1, u0 - v)
                    const uint32_t m = pku_min(c0, c1);
                    const uint32_t c = (m & 0xffffu) + (m >> 18);                              // prefiltered + (raw >> 2)
                    if (j & 1) od |= c << (8 * (j - 1)); else ev |= c << (8 * j);
                }
                Pk[xl * QS + dqi] = make_uint2(ev, od);
                xl += dxl; dqi += ddq;
                if (xl >= NXL) { xl -= NXL; dqi++; }
            }
        }
        __syncthreads();
        // ---- emit C(y) for y = k-1 (needs rows y-1, y, y+1), and y = 0 when h == 1
        int y = (g.h == 1) ? 0 : k - 1;
        if (y < y0 || y >= y1 || (g.h > 1 && k == 0)) continue;
        if (g.h > 1 && y == g.h - 1) continue;                   // written as zeros below (Q2)
        const uint2* Pa = P + (size_t)(max(y - 1, 0) % 3) * NXL * QS;
        const uint2* Pb = P + (size_t)(y % 3) * NXL * QS;
        const uint2* Pc = P + (size_t)(min(y + 1, g.h - 1) % 3) * NXL * QS;
        int16_t* Crow = a.C + (size_t)y * g.width1 * D;
        const uint32_t P2pk = pk_dup(a.P2);
        for (int e = tid; e < (xend - xs0) * quarterD; e += 256) {
            const int xl = e / quarterD, dq = (e - xl * quarterD) * 4;
            const int x = xs0 + xl;
            uint32_t ev = 0, od = 0;                             // sums of d0,d2 | d1,d3 as packed u16
            if (!(y > 0 && x == 0)) {                            // Q1
                ev = P2pk; od = P2pk;
                #pragma unroll
                for (int kx = 0; kx < 3; kx++) {
                    const int o = (xl + kx) * QS + (dq >> 2);
                    const uint2 wa = Pa[o], wb = Pb[o], wc = Pc[o];
                    ev += (wa.x + wb.x) + wc.x;
                    od += (wa.y + wb.y) + wc.y;
                }
            }
            uint2 out;
            out.x = __builtin_amdgcn_perm(od, ev, 0x05040100u);  // (d0, d1)
            out.y = __builtin_amdgcn_perm(od, ev, 0x07060302u);  // (d2, d3)
            *reinterpret_cast<uint2*>(Crow + (size_t)x * D + dq) = out;
        }
    }
    // Q2: the last row is never written by the reference (zero under "uninitialised == 0")
    if (g.h > 1 && y1 == g.h) {
        int16_t* Crow = a.C + (size_t)(g.h - 1) * g.width1 * D;
        const int halfD = D >> 1;
        for (int e = tid; e < (xend - xs0) * halfD; e += 256)
            *reinterpret_cast<uint32_t*>(Crow + (size_t)xs0 * D + e * 2) = 0u;
    }
}

// debug/parity: S = sat16(sum_r L_r) = 8*C - sum_r e_r   (stereosgbm.cpp:655)
__global__ __launch_bounds__(256) void k_sum_S(const int16_t* __restrict__ C, const uint8_t* __restrict__ E, size_t vol,
                                               int16_t* __restrict__ S)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= vol) return;
    int c = C[i], s = 0;
    #pragma unroll
    for (int r = 0; r < 8; r++) s += c - (int)E[(size_t)r * vol + i];
    S[i] = (int16_t)min(max(s, -32768), 32767);
}

// =============================================================================================
// K4: winner-take-all + uniqueness + disp2 competition + parabola + left-right check
// (stereosgbm.cpp:664-816).  One block per canvas row; a group of G lanes owns one pixel at a time.
// "Padded" semantics for the reference's out-of-bounds disp2 store (_x2 < 0): no side effect
// (oracle/sgbm_oracle.c, alias_oob = 0).
// =============================================================================================
struct WtaArgs {
    const int16_t* C; const uint8_t* E; size_t vol;
    Geom g;
    int uniq, maxdiff;
    int16_t* disp; int16_t* cost;     // h * Wc
};

template <int G, int K, bool PAD>
__global__ __launch_bounds__(256) void k_wta(WtaArgs a)
{
    constexpr int DPL = 2 * K;          // disparities per lane
    extern __shared__ __attribute__((aligned(16))) uint8_t sm[];
    const Geom& g = a.g;
    const int Wc = g.Wc, D = g.D, width1 = g.width1, y = blockIdx.x;
    uint32_t* d2key = reinterpret_cast<uint32_t*>(sm);            // [Wc]  (cost+32768)<<16 | (0xffff - x)
    int16_t* d1 = reinterpret_cast<int16_t*>(d2key + Wc);        // [Wc]
    int16_t* c1 = d1 + Wc;                                       // [Wc]
    for (int x = threadIdx.x; x < Wc; x += 256) { d2key[x] = 0xffffffffu; d1[x] = (int16_t)g.invalid; c1[x] = 0; }
    __syncthreads();

    constexpr int NP = 64 / G;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gl = lane & (G - 1);
    const bool lane_ok = PAD ? (gl * DPL < D) : true;
    // software pipeline: the 9 loads (C + 8 e-volumes) of the NEXT pixel group are in flight while the
    // current one is reduced; bounds/padding lanes use an out-of-range buffer offset (loads return 0)
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(const_cast<int16_t*>(a.C), 0, (int)(a.vol * 2), S2P_BUF_FLAGS);
    __amdgpu_buffer_rsrc_t rsE[8];
    #pragma unroll
    for (int r = 0; r < 8; r++) rsE[r] = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.E) + (size_t)r * a.vol, 0, (int)a.vol, S2P_BUF_FLAGS);
    const uint32_t rowoff = (uint32_t)((size_t)y * width1 * D);
    typedef CostLoad<int16_t, K> CL;
    typedef EBytes<K> EL;
    struct Px { typename CL::raw_t c; typename EL::raw_t e[8]; };
    auto issue = [&](int xb) __attribute__((always_inline)) -> Px {
        const int x = xb + wave * NP + lane / G;
        const bool in = x < width1 && lane_ok;
        const uint32_t off = rowoff + (uint32_t)(x * D + gl * DPL);
        Px p;
        p.c = CL::load_last(rsC, in ? off * 2u : S2P_OOB - 32u);      // - 32: the second 16-byte half must stay out of range too
        #pragma unroll
        for (int r = 0; r < 8; r++) p.e[r] = EL::load(rsE[r], in ? off : S2P_OOB);
        return p;
    };
    Px cur = issue(0);
    for (int xb = 0; xb < width1; xb += 4 * NP) {
        const Px nxt = issue(xb + 4 * NP);
        const int x = xb + wave * NP + lane / G;
        const bool ok = x < width1 && lane_ok;
        int S[DPL];
        {
            costs_to_ints<int16_t, K>(cur.c, S);
            #pragma unroll
            for (int j = 0; j < DPL; j++) S[j] *= 8;
            #pragma unroll
            for (int r = 0; r < 8; r++) {
                int ev[DPL];
                EL::get(cur.e[r], ev);
                #pragma unroll
                for (int j = 0; j < DPL; j++) S[j] -= ev[j];
            }
        }
        cur = nxt;
        // first minimum over d ascending (:762-770): min over (S, d) keys
        uint32_t key = 0xffffffffu;
        #pragma unroll
        for (int j = 0; j < DPL; j++) {
            uint32_t k = ((uint32_t)(S[j] + 32768) << 16) | (uint32_t)(gl * DPL + j);
            key = (ok && k < key) ? k : key;
        }
        key = group_min_u32<G>(key);
        const int minS = (int)(key >> 16) - 32768, best = (int)(key & 0xffffu);
        // uniqueness (:773-779)
        int bad = 0;
        #pragma unroll
        for (int j = 0; j < DPL; j++) {
            int d = gl * DPL + j;
            bad |= (ok && S[j] * (100 - a.uniq) < minS * 100 && abs(best - d) > 1) ? 1 : 0;
        }
        bad = group_or_i32<G>(bad);
        // S[best-1], S[best+1] (:788-797): gather with a masked or-reduce (exactly one lane contributes)
        int sm1 = 0, sp1 = 0;
        #pragma unroll
        for (int j = 0; j < DPL; j++) {
            int d = gl * DPL + j;
            sm1 |= (ok && d == best - 1) ? (S[j] & 0xffff) : 0;
            sp1 |= (ok && d == best + 1) ? (S[j] & 0xffff) : 0;
        }
        const int packed = group_or_i32<G>(sm1 | (sp1 << 16));
        if (x < width1 && gl == 0 && !bad) {
            int d = best;
            int x2 = x + g.minX1 - d - g.minD;
            if (x2 >= 0)        // padded semantics: the reference's out-of-bounds store is dropped
                atomicMin(&d2key[x2], ((uint32_t)(minS + 32768) << 16) | (uint32_t)(0xffff - x));
            if (0 < d && d < D - 1) {
                int Sm = (int)(short)(packed & 0xffff), Sp = (int)(short)((uint32_t)packed >> 16);
                int denom2 = max(Sm + Sp - 2 * minS, 1);
                d = d * 16 + ((Sm - Sp) * 16 + denom2) / (denom2 * 2);
            } else
                d *= 16;
            d1[x + g.minX1] = (int16_t)(d + g.minD * 16);
            c1[x + g.minX1] = (int16_t)minS;
        }
    }
    __syncthreads();
    // left-right check (:802-816)
    for (int x = threadIdx.x; x < Wc; x += 256) {
        int dv = d1[x];
        if (x >= g.minX1 && x < g.maxX1 && dv != g.invalid) {
            int _d = dv >> 4, d_ = (dv + 15) >> 4;
            int _x = x - _d, x_ = x - d_;
            bool k1 = false, k2 = false;
            if (_x >= 0 && _x < Wc) {
                uint32_t kk = d2key[_x];
                int v = (kk == 0xffffffffu) ? g.invalid : ((0xffff - (int)(kk & 0xffffu)) + g.minX1 - _x);
                k1 = v >= g.minD && abs(v - _d) > a.maxdiff;
            }
            if (x_ >= 0 && x_ < Wc) {
                uint32_t kk = d2key[x_];
                int v = (kk == 0xffffffffu) ? g.invalid : ((0xffff - (int)(kk & 0xffffu)) + g.minX1 - x_);
                k2 = v >= g.minD && abs(v - d_) > a.maxdiff;
            }
            if (k1 && k2) dv = g.invalid;
        }
        a.disp[(size_t)y * Wc + x] = (int16_t)dv;
        a.cost[(size_t)y * Wc + x] = c1[x];
    }
}

// =============================================================================================
// K5: 3x3 median with replicate borders (smooth.cpp:246-270) on the int16 canvas, INVALID included.
// =============================================================================================
__device__ __forceinline__ void cswap(int& a, int& b) { int t = min(a, b); b = max(a, b); a = t; }

__global__ __launch_bounds__(256) void k_median3(const int16_t* __restrict__ src, int16_t* __restrict__ dst, int w, int h)
{
    int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const int16_t* r0 = src + (size_t)max(y - 1, 0) * w;
    const int16_t* r1 = src + (size_t)y * w;
    const int16_t* r2 = src + (size_t)min(y + 1, h - 1) * w;
    int j0 = max(x - 1, 0), j2 = min(x + 1, w - 1);
    int p0 = r0[j0], p1 = r0[x], p2 = r0[j2], p3 = r1[j0], p4 = r1[x], p5 = r1[j2], p6 = r2[j0], p7 = r2[x], p8 = r2[j2];
    cswap(p1, p2); cswap(p4, p5); cswap(p7, p8); cswap(p0, p1);
    cswap(p3, p4); cswap(p6, p7); cswap(p1, p2); cswap(p4, p5);
    cswap(p7, p8); cswap(p0, p3); cswap(p5, p8); cswap(p4, p7);
    cswap(p3, p6); cswap(p1, p4); cswap(p2, p5); cswap(p4, p7);
    cswap(p4, p2); cswap(p6, p4); cswap(p4, p2);
    dst[(size_t)y * w + x] = (int16_t)p4;
}

// =============================================================================================
// K7: epilogue (sgbm.cpp:210-234): crop the canvas, back to the s2p sign convention, /16, NaN for
// INVALID; fused with create_rejection_mask (s2p/block_matching.py:18-32).
// =============================================================================================
__global__ __launch_bounds__(256) void k_epilogue(const int16_t* __restrict__ dcan, const int16_t* __restrict__ ccan,
                                                  const float* __restrict__ im1, const float* __restrict__ im2,
                                                  int w, int h, int Wc, int x0, int invalid,
                                                  float* __restrict__ disp, float* __restrict__ cost, uint8_t* __restrict__ mask)
{
    int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    size_t i = (size_t)y * w + x, ic = (size_t)y * Wc + x0 + x;
    int dv = dcan[ic];
    float d, c;
    if (dv == invalid) { d = __builtin_nanf(""); c = d; }
    else { d = -((float)dv) / 16.0f; c = (float)ccan[ic]; }
    disp[i] = d;
    if (cost) cost[i] = c;
    if (mask) {
        bool ok = (dv != invalid) && isfinite(im1[i]);
        if (ok) {
            float xs = (float)x + d;
            if (!(xs >= -0.5f && xs <= (float)w - 0.5f)) ok = false;     // pinned on the reference's stored mask (oracle/sgbm_oracle.c)
            else {
                int xi = (int)floorf(xs);
                float fr = xs - (float)xi;
                const int t0 = min(max(xi, 0), w - 1), t1 = min(max(xi + 1, 0), w - 1);
                ok = isfinite(im2[(size_t)y * w + t0]) && (fr == 0.0f || isfinite(im2[(size_t)y * w + t1]));
            }
        }
        mask[i] = ok ? 1 : 0;
    }
}

// =============================================================================================
// host-side pipeline
// =============================================================================================
template <int G, int K>
static void launch_wta(hipStream_t st, int rows, size_t shmem, bool pad, const WtaArgs& a) {
    if (pad) hipLaunchKernelGGL((k_wta<G, K, true>), dim3(rows), dim3(256), shmem, st, a);
    else     hipLaunchKernelGGL((k_wta<G, K, false>), dim3(rows), dim3(256), shmem, st, a);
}

__global__ __launch_bounds__(256) void k_fill_invalid(size_t n, float* disp, float* cost, uint8_t* mask)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    disp[i] = __builtin_nanf("");
    if (cost) cost[i] = __builtin_nanf("");
    if (mask) mask[i] = 0;
}

__global__ __launch_bounds__(256) void k_fill_s16(int16_t* p, size_t n, int v)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = (int16_t)v;
}

int sgbm_read_rminmax(s2p_hip_ctx* ctx, const SgbmBuffers& b, float out[2])
{
    S2P_HIP_CHECK(hipMemcpyAsync(out, b.st->rminmax, 2 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    return S2P_HIP_OK;
}

// number of flat indices idx = Wc-1-x+d the cost kernel can touch: x >= minX1, d < maxD
static inline int sgbm_ni(const Geom& g) { return (int)align_up((size_t)g.Wc + std::max(g.maxD, 0) + 8, 8); }

static inline size_t sgbm_vol(const Geom& g) { return g.width1 > 0 ? (size_t)g.h * g.width1 * g.D : 0; }

size_t sgbm_workspace_bytes(const Geom& g, bool want_S)
{
    size_t vol = sgbm_vol(g);
    size_t n = 0;
    auto add = [&](size_t b) { n += align_up(b, 256); };
    add(sizeof(SelectState)); add(3 * 2 * 2048 * 4);
    add((size_t)g.Wc * g.h); add((size_t)g.Wc * g.h);
    add((size_t)g.h * 3 * sgbm_ni(g) * 4); add((size_t)g.h * 3 * g.Wc * 4);
    add(vol * 2); add(vol * 8); if (want_S) add(vol * 2);
    add((size_t)g.Wc * g.h * 2); add((size_t)g.Wc * g.h * 2); add((size_t)g.Wc * g.h * 2); add((size_t)g.Wc * g.h * 2);
    add((size_t)g.Wc * g.h * 4); add((size_t)g.Wc * g.h * 4); add((size_t)g.Wc * g.h * 4);
    return n + 4096;
}

static int carve(s2p_hip_ctx* ctx, const Geom& g, bool want_S, SgbmBuffers* b)
{
    size_t vol = sgbm_vol(g);
    ws_reset(ctx);
    #define CARVE(field, type, bytes) b->field = (type)ws_alloc(ctx, (bytes)); if (!b->field) return S2P_HIP_RUNTIME_ERROR;
    CARVE(st, SelectState*, sizeof(SelectState));
    CARVE(hist, uint32_t*, 3 * 2 * 2048 * 4);
    CARVE(uu1, uint8_t*, (size_t)g.Wc * g.h);
    CARVE(uu2, uint8_t*, (size_t)g.Wc * g.h);
    CARVE(vpk, uint32_t*, (size_t)g.h * 3 * sgbm_ni(g) * 4);
    CARVE(upk, uint32_t*, (size_t)g.h * 3 * g.Wc * 4);
    CARVE(C, int16_t*, vol * 2);
    CARVE(E, uint8_t*, vol * 8);
    b->S = nullptr;
    if (want_S) { CARVE(S, int16_t*, vol * 2); }
    CARVE(disp_raw, int16_t*, (size_t)g.Wc * g.h * 2);
    CARVE(cost_raw, int16_t*, (size_t)g.Wc * g.h * 2);
    CARVE(disp_med, int16_t*, (size_t)g.Wc * g.h * 2);
    CARVE(disp_fin, int16_t*, (size_t)g.Wc * g.h * 2);
    CARVE(lab, int*, (size_t)g.Wc * g.h * 4);
    CARVE(cnt, int*, (size_t)g.Wc * g.h * 4);
    CARVE(par, int*, (size_t)g.Wc * g.h * 4);
    #undef CARVE
    return S2P_HIP_OK;
}

// Enqueue the whole sgbm pipeline on ctx->stream.  All pointers are device pointers.
int sgbm_enqueue(s2p_hip_ctx* ctx, const Geom& g, const s2p_sgbm_params& p,
                 const float* d_im1, const float* d_im2, float* d_disp, float* d_cost, uint8_t* d_mask,
                 bool want_S, SgbmBuffers* out)
{
    hipStream_t st = ctx->stream;
    SgbmBuffers b;
    int rc = ws_reserve(ctx, sgbm_workspace_bytes(g, want_S));
    if (rc) return rc;
    rc = carve(ctx, g, want_S, &b);
    if (rc) return rc;
    if (out) *out = b;
    const size_t npx = (size_t)g.w * g.h, ncan = (size_t)g.Wc * g.h;
    const size_t vol = sgbm_vol(g);
    StageScope total(ctx, "total");

    {   // ---- K0/K1: rank select + quantise
        StageScope s(ctx, "quantize");
        const int nh = 3 * 2 * 2048;
        hipLaunchKernelGGL(k_zero_u32, dim3((nh + 255) / 256), dim3(256), 0, st, b.hist, nh);
        int nb = (int)std::min<size_t>((npx + 256 * 16 - 1) / (256 * 16), 1024);
        if (nb < 1) nb = 1;
        hipLaunchKernelGGL(k_select_hist<0>, dim3(nb), dim3(256), 0, st, d_im1, npx, b.st, b.hist);
        hipLaunchKernelGGL(k_select_pick<0>, dim3(1), dim3(256), 0, st, b.st, b.hist);
        hipLaunchKernelGGL(k_select_hist<1>, dim3(nb), dim3(256), 0, st, d_im1, npx, b.st, b.hist + 2 * 2048);
        hipLaunchKernelGGL(k_select_pick<1>, dim3(1), dim3(256), 0, st, b.st, b.hist + 2 * 2048);
        hipLaunchKernelGGL(k_select_hist<2>, dim3(nb), dim3(256), 0, st, d_im1, npx, b.st, b.hist + 4 * 2048);
        hipLaunchKernelGGL(k_select_pick<2>, dim3(1), dim3(256), 0, st, b.st, b.hist + 4 * 2048);
        hipLaunchKernelGGL(k_quantize_paste, dim3((g.Wc + 255) / 256, g.h), dim3(256), 0, st,
                           d_im1, d_im2, g.w, g.h, g.Wc, g.x0, b.st, b.uu1, b.uu2);
    }
    if (g.width1 <= 0) {   // stereosgbm.cpp:347-351: everything INVALID -> NaN after the epilogue
        hipLaunchKernelGGL(k_fill_invalid, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, st, npx, d_disp, d_cost, d_mask);
        const unsigned nbc = (unsigned)((ncan + 255) / 256);     // keep the stage dumps meaningful: constant INVALID canvases
        hipLaunchKernelGGL(k_fill_s16, dim3(nbc), dim3(256), 0, st, b.disp_raw, ncan, g.invalid);
        hipLaunchKernelGGL(k_fill_s16, dim3(nbc), dim3(256), 0, st, b.disp_med, ncan, g.invalid);
        hipLaunchKernelGGL(k_fill_s16, dim3(nbc), dim3(256), 0, st, b.disp_fin, ncan, g.invalid);
        hipMemsetAsync(b.cost_raw, 0, ncan * 2, st);
        return S2P_HIP_OK;
    }
    {   // ---- K2: prefilter + block cost
        StageScope s(ctx, "cost");
        const int NI = sgbm_ni(g);
        if ((size_t)2 * g.fl > 64 * 1024)      // wide canvases: the flat row scratch needs more than the default 64 KiB of dynamic LDS
            hipFuncSetAttribute((const void*)k_prefilter, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(k_prefilter, dim3(g.h), dim3(256), (size_t)2 * g.fl, st, b.uu1, b.uu2, g,
                           std::max(p.prefilter_cap, 15) | 1, NI, b.vpk, b.upk);
        CostArgs ca;
        ca.g = g; ca.vpk = b.vpk; ca.upk = b.upk; ca.NI = NI; ca.C = b.C; ca.P2 = p.P2;
        ca.XS = std::max(4, std::min(64, 4096 / g.D)); ca.YC = 32;
        ca.WL = (ca.XS + 2 + g.D + 3) & ~3;
        size_t shm = (size_t)(ca.XS + 2) * 16 + (size_t)3 * ca.WL * 4 + (size_t)3 * (ca.XS + 2) * (g.D / 4 + 1) * 8;
        hipLaunchKernelGGL(k_block_cost, dim3((g.width1 + ca.XS - 1) / ca.XS, (g.h + ca.YC - 1) / ca.YC), dim3(256), shm, st, ca);
    }
    const LaneLayout ll = lane_layout(g.D);
    const int G = ll.G;
    const bool pad = ll.pad;
    {   // ---- K3: aggregation, 8 directions in one launch (agg.hpp)
        StageScope s(ctx, "aggregate");
        enqueue_aggregate<int16_t>(st, b.C, b.E, g.width1, g.h, g.D, p.P1, p.P2, 0);
    }
    if (want_S) hipLaunchKernelGGL(k_sum_S, dim3((unsigned)((vol + 255) / 256)), dim3(256), 0, st, b.C, b.E, vol, b.S);
    {   // ---- K4: WTA row kernel
        StageScope s(ctx, "wta");
        WtaArgs wa;
        wa.C = b.C; wa.E = b.E; wa.vol = vol; wa.g = g; wa.uniq = p.uniqueness_ratio >= 0 ? p.uniqueness_ratio : 10;
        wa.maxdiff = p.lr > 0 ? p.lr : 1; wa.disp = b.disp_raw; wa.cost = b.cost_raw;
        size_t shm = (size_t)g.Wc * 8;
        if (ll.K == 8) launch_wta<64, 8>(st, g.h, shm, pad, wa);
        else switch (G) {
            case 2: launch_wta<2, 4>(st, g.h, shm, pad, wa); break;
            case 4: launch_wta<4, 4>(st, g.h, shm, pad, wa); break;
            case 8: launch_wta<8, 4>(st, g.h, shm, pad, wa); break;
            case 16: launch_wta<16, 4>(st, g.h, shm, pad, wa); break;
            case 32: launch_wta<32, 4>(st, g.h, shm, pad, wa); break;
            default: launch_wta<64, 4>(st, g.h, shm, pad, wa); break;
        }
    }
    {   // ---- K5: median
        StageScope s(ctx, "median");
        hipLaunchKernelGGL(k_median3, dim3((g.Wc + 255) / 256, g.h), dim3(256), 0, st, b.disp_raw, b.disp_med, g.Wc, g.h);
    }
    int16_t* fin = b.disp_fin;
    hipMemcpyAsync(b.disp_fin, b.disp_med, ncan * 2, hipMemcpyDeviceToDevice, st);
    if (p.speckle_window > 0) {   // ---- K6: speckle
        StageScope s(ctx, "speckle");
        enqueue_speckle(st, fin, g.Wc, g.h, g.invalid, p.speckle_window, 16 * p.speckle_range, b.lab, b.par, b.cnt);
    }
    {   // ---- K7: epilogue + rejection mask
        StageScope s(ctx, "epilogue");
        hipLaunchKernelGGL(k_epilogue, dim3((g.w + 255) / 256, g.h), dim3(256), 0, st, fin, b.cost_raw, d_im1, d_im2,
                           g.w, g.h, g.Wc, g.x0, g.invalid, d_disp, d_cost, d_mask);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_last_error("kernel launch failed: %s", hipGetErrorString(e)); return S2P_HIP_RUNTIME_ERROR; }
    return S2P_HIP_OK;
}

}  // namespace s2p
