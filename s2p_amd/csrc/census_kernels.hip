// s2p_amd/csrc/census_kernels.hip -- census / Hamming cost + 8-path SGM matcher for gfx950: the
// MI355X stand-in for the reference's `mgm` / `mgm_multi` binaries (s2p/block_matching.py:155-188,
// 269-310; sources absent from the reference tree, see oracle/census_oracle.c for the algorithm
// statement these kernels match bit for bit and DESIGN.md for the parity status).
//
// HBM layout (row-major, d fastest; D = roundup(dmax - dmin + 1, 16)):
//   cen1, cen2 [h][w]        uint32  census signatures (24 bits for 5x5)
//   C          [h][w][D]     uint8   Hamming cost, 255 = excluded candidate        (1 B / candidate)
//   E_r        [h][w][D]     uint8   r = 0..7, e = (C + P2) - L_r  in [0, P2]      (1 B / candidate / path)
// The aggregation is the shared wavefront-recurrence kernel of agg.hpp instantiated for uint8 costs.
#include "common.hpp"
#include "agg.hpp"
#include "ccl.hpp"
#include <cstdio>
#include <mutex>
#include <map>
#include "mgm_geom.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace s2p {

// both 16-bit fields shifted right by one (v_pk_lshrrev_b16)
__device__ __forceinline__ uint32_t pk_shr1(uint32_t a) {
    typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2_t, a) >> (unsigned short)1);
}

#define C_EXCLUDED 255
#define CENSUS_MAX_BITS 24      // largest Hamming distance of a valid candidate
#ifndef S2P_WTA_NT
#define S2P_WTA_NT 256         // threads per WTA block (one block = one image row); 512 / 1024 measured equal
#endif
#ifndef S2P_COST_KILLMASK
#define S2P_COST_KILLMASK 1      // (0: the per-candidate range tests of rounds 1-3, for A/B builds)
#endif
#ifndef S2P_WTA_PF
#define S2P_WTA_PF 2          // pixel groups in flight per wave in the packed WTA kernel
#endif

// ---- census transform: bit = neighbour < centre, row-major neighbours, clamped coordinates.
// Bit 31 flags a non-finite centre pixel (the signature itself uses <= 24 bits), so that the cost
// kernel needs the two signature images only. --------------------------------------------------------
#define CENSUS_INVALID 0x80000000u
template <int WIN>
__device__ __forceinline__ uint32_t census_signature(const float* __restrict__ im, int w, int h, int x, int y)
{
    constexpr int R = WIN / 2;
    const float c = im[(size_t)y * w + x];
    uint32_t bits = 0;
    #pragma unroll
    for (int dy = -R; dy <= R; dy++) {
        const float* row = im + (size_t)min(max(y + dy, 0), h - 1) * w;
        #pragma unroll
        for (int dx = -R; dx <= R; dx++) {
            if (dx == 0 && dy == 0) continue;
            bits = (bits << 1) | (row[min(max(x + dx, 0), w - 1)] < c ? 1u : 0u);
        }
    }
    return isfinite(c) ? bits : (bits | CENSUS_INVALID);
}
// ---- Hamming cost volume.  One block per image row: both signature rows are staged in LDS once, then every
// thread produces 8 consecutive candidates (one octet) of one pixel with one 8-byte store.  A wavefront covers
// PW consecutive pixels x OW consecutive octets (8 x 8 at D >= 64): for a fixed candidate j its 64 lanes read the
// signature words x + 8 o + j = 64 consecutive words (bank-conflict free; 16 lanes on one pixel would read words
// 8 apart, a 4-way conflict), and every pixel still receives OW * 8 contiguous bytes per store instruction.
// The image-2 row is extended by D invalid entries on both sides, so candidates that fall outside image 2 need no
// range test.  Candidates outside image 2, padding and NaN pixels get 255.
// SP = 2 (mgm_multi's SUBPIX): candidate j stands for the disparity dmin + j / 2; image 2 is also sampled half way
// between its columns (mean of the two neighbours) and census-transformed there, and the two signature rows are
// interleaved in LDS (entry 2 x + phase), so that consecutive candidates are again consecutive words.
// lo / hi (multi-scale mode): the admissible disparities of every pixel, in whole pixels; the others get 255. ------
static inline size_t census_cost_lds(int w, int D, int sp) { return (size_t)w * 4 + (size_t)(sp * w + 2 * D) * 4; }
// census signature of image `im` sampled half way between columns x and x + 1 (clamped), same bit order as above
template <int WIN>
__device__ __forceinline__ uint32_t census_signature_half(const float* __restrict__ im, int w, int h, int x, int y)
{
    constexpr int R = WIN / 2;
    auto at = [&](const float* row, int xx) -> float { return __fmul_rn(0.5f, __fadd_rn(row[xx], row[min(xx + 1, w - 1)])); };
    const float c = at(im + (size_t)y * w, x);
    uint32_t bits = 0;
    #pragma unroll
    for (int dy = -R; dy <= R; dy++) {
        const float* row = im + (size_t)min(max(y + dy, 0), h - 1) * w;
        #pragma unroll
        for (int dx = -R; dx <= R; dx++) {
            if (dx == 0 && dy == 0) continue;
            bits = (bits << 1) | (at(row, min(max(x + dx, 0), w - 1)) < c ? 1u : 0u);
        }
    }
    return isfinite(c) ? bits : (bits | CENSUS_INVALID);
}
// The signature rows are computed by the block itself from the two images (5 clamped rows each, L2-resident) and
// also written to cen1 / cen2 for the stage dumps when those are given.
template <int WIN, int SP>
__global__ __launch_bounds__(256) void k_census_cost(const float* __restrict__ im1, const float* __restrict__ im2, int h,
                                                     uint32_t* __restrict__ cen1, uint32_t* __restrict__ cen2,
                                                     int w, int dmin, int Dt, int D,
                                                     const int16_t* __restrict__ lo, const int16_t* __restrict__ hi,
                                                     uint8_t* __restrict__ C)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t sm[];
    uint32_t* s1 = reinterpret_cast<uint32_t*>(sm);     // [w]
    uint32_t* s2e = s1 + w;                              // [SP w + 2 D]: slot i = (half-)pixel i - D of image 2
    const int y = blockIdx.x, we = SP * w + 2 * D;
    for (int x = threadIdx.x; x < w; x += 256) {
        const uint32_t a = census_signature<WIN>(im1, w, h, x, y), b = census_signature<WIN>(im2, w, h, x, y);
        s1[x] = a; s2e[SP * x + D] = b;
        if (SP == 2) s2e[2 * x + 1 + D] = census_signature_half<WIN>(im2, w, h, x, y);
        if (cen1) { cen1[(size_t)y * w + x] = a; cen2[(size_t)y * w + x] = b; }
    }
    for (int i = threadIdx.x; i < 2 * D; i += 256) s2e[i < D ? i : SP * w + i] = CENSUS_INVALID;     // the two invalid margins
    __syncthreads();
    const int oct = D >> 3;
    const int OW = oct < 8 ? oct : 8, PW = 64 / OW;      // oct is even (D is a multiple of 16): OW in {2, 4, 6, 8}
    const int nog = (oct + OW - 1) / OW, npg = (w + PW - 1) / PW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lo_ = lane / PW, lp = lane - lo_ * PW;     // octet / pixel of this lane inside the wave's patch
    uint8_t* Crow = C + (size_t)y * w * D;
    for (int it = wave; it < npg * nog; it += 4) {
        const int pg = it / nog, og = it - pg * nog;
        const int x = pg * PW + lp, o = og * OW + lo_;
        if (x >= w || o >= oct || lo_ >= OW) continue;
        const uint32_t a = s1[x];
        // 8 slots from i0 on; a run entirely left (right) of the extended row is moved into the invalid margin
        const int i0 = min(max(SP * (x + dmin) + o * 8 + D, 0), we - 8);
        int jlim = Dt - o * 8, jlo = 0;                  // candidates j >= jlim are padding; j < jlo / j >= jlim outside the pixel's range
        if (lo) {
            jlo = SP * ((int)lo[(size_t)y * w + x] - dmin) - o * 8;
            jlim = min(jlim, SP * ((int)hi[(size_t)y * w + x] - dmin) - o * 8 + 1);
        }
        uint32_t lo32 = 0, hi32 = 0;
        #pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t b = s2e[i0 + j];
            uint32_t c = ((a | b) & CENSUS_INVALID) ? (uint32_t)C_EXCLUDED : (uint32_t)__popc(a ^ b);
#if !S2P_COST_KILLMASK
            c = (j < jlim && j >= jlo) ? c : (uint32_t)C_EXCLUDED;
#endif
            if (j < 4) lo32 |= c << (8 * j); else hi32 |= c << (8 * (j - 4));
        }
#if S2P_COST_KILLMASK
        {   // candidates outside [jlo, jlim) of this octet are excluded: all-ones bytes OR-ed over the 8 costs at once (two compares and a
            // select per CANDIDATE before; with tiles in flight the kernels share the SIMDs and every VALU instruction counts: DESIGN_KERNELS.md 4)
            const int ja = max(jlo, 0), jb = min(jlim, 8);
            const unsigned long long keep = jb > ja ? ((~0ull >> (8 * (8 - (jb - ja)))) << (8 * ja)) : 0ull;
            lo32 |= ~(uint32_t)keep; hi32 |= ~(uint32_t)(keep >> 32);
        }
#endif
        *reinterpret_cast<uint2*>(Crow + ((size_t)x * oct + o) * 8) = make_uint2(lo32, hi32);
    }
}

// ---- multi-scale mode (mgm_multi's -S; oracle: s2p_oracle_down2, s2p_oracle_range_from_coarse) ---------------------
// 2x2 mean of the finite samples, summed in the order (0,0) (1,0) (0,1) (1,1), one division; NaN if none
__global__ __launch_bounds__(256) void k_down2(const float* __restrict__ src, int w, int h, float* __restrict__ dst)
{
    const int w2 = (w + 1) >> 1, h2 = (h + 1) >> 1;
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w2 || y >= h2) return;
    float sum = 0.0f; int n = 0;
    #pragma unroll
    for (int dy = 0; dy < 2; dy++)
        #pragma unroll
        for (int dx = 0; dx < 2; dx++) {
            const int xx = 2 * x + dx, yy = 2 * y + dy;
            if (xx >= w || yy >= h) continue;
            const float v = src[(size_t)yy * w + xx];
            if (isfinite(v)) { sum = __fadd_rn(sum, v); n++; }
        }
    dst[(size_t)y * w2 + x] = n ? __fdiv_rn(sum, (float)n) : __builtin_nanf("");
}
#define MS_MIN_DIM 128      // a level is only added while the smaller side of the halved pair stays >= this
#define MS_MARGIN 2         // pixels added on both sides of the range a parent neighbourhood suggests
// admissible range of every pixel of a (w, h) level from the disparity map of its parent level
__global__ __launch_bounds__(256) void k_range_from_coarse(const float* __restrict__ dc, int w, int h, int dmin, int dmax,
                                                           int16_t* __restrict__ lo, int16_t* __restrict__ hi)
{
    const int wc = (w + 1) >> 1, hc = (h + 1) >> 1;
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const int cx = x >> 1, cy = y >> 1;
    int l = dmin, u = dmax;
    if (isfinite(dc[(size_t)cy * wc + cx])) {
        float mn = __builtin_inff(), mx = -__builtin_inff();
        #pragma unroll
        for (int dy = -1; dy <= 1; dy++)
            #pragma unroll
            for (int dx = -1; dx <= 1; dx++) {
                const int xx = cx + dx, yy = cy + dy;
                if (xx < 0 || xx >= wc || yy < 0 || yy >= hc) continue;
                const float v = dc[(size_t)yy * wc + xx];
                if (isfinite(v)) { mn = fminf(mn, v); mx = fmaxf(mx, v); }
            }
        l = (int)floorf(__fmul_rn(2.0f, mn)) - MS_MARGIN;
        u = (int)ceilf(__fmul_rn(2.0f, mx)) + MS_MARGIN;
        l = min(max(l, dmin), dmax);
        u = min(max(u, dmin), dmax);
    }
    lo[(size_t)y * w + x] = (int16_t)l;
    hi[(size_t)y * w + x] = (int16_t)u;
}
// union of the ranges of the pixels of a (w, h) level that have a parent estimate: mm[0] = min lo, mm[1] = max hi
// (mm preset to {INT_MAX, INT_MIN}).  A block walks rows y = blockIdx.x, blockIdx.x + gridDim.x, ... and ends with ONE pair of
// atomics: until round 4 every wave of a (w / 256, h) grid ended with its own pair -- 31 000 atomics on the same two words for a
// 1000 x 1000 level, which the L2 serialises: 229 us per call, 0.46 ms per 'mgm_multi' tile, a sixth of its kernel time
// (profiles/r04/ms_trace_kernel_stats.csv).
#define S2P_RANGE_UNION_BLOCKS 240
__global__ __launch_bounds__(256) void k_range_union(const float* __restrict__ dc, const int16_t* __restrict__ lo, const int16_t* __restrict__ hi,
                                                     int w, int h, int* __restrict__ mm)
{
    __shared__ int sa[4], sb[4];
    const int wc = (w + 1) >> 1;
    int a = 0x7fffffff, b = -0x7fffffff - 1;
    for (int y = blockIdx.x; y < h; y += gridDim.x)
        for (int x = threadIdx.x; x < w; x += 256)
            if (isfinite(dc[(size_t)(y >> 1) * wc + (x >> 1)])) { a = min(a, (int)lo[(size_t)y * w + x]); b = max(b, (int)hi[(size_t)y * w + x]); }
    #pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a = min(a, __shfl_xor(a, o)); b = max(b, __shfl_xor(b, o)); }
    if ((threadIdx.x & 63) == 0) { sa[threadIdx.x >> 6] = a; sb[threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = min(min(sa[0], sa[1]), min(sa[2], sa[3])); b = max(max(sb[0], sb[1]), max(sb[2], sb[3]));
        if (a <= b) { atomicMin(mm, a); atomicMax(mm + 1, b); }
    }
}
static inline dim3 range_union_grid(int h) { return dim3((unsigned)std::max(1, std::min(h, S2P_RANGE_UNION_BLOCKS))); }
// pixels without a parent estimate search what the level's other pixels search
__global__ __launch_bounds__(256) void k_range_fill(const float* __restrict__ dc, int w, int h, const int* __restrict__ mm,
                                                    int16_t* __restrict__ lo, int16_t* __restrict__ hi)
{
    const int wc = (w + 1) >> 1, x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w || isfinite(dc[(size_t)(y >> 1) * wc + (x >> 1)])) return;
    lo[(size_t)y * w + x] = (int16_t)mm[0];
    hi[(size_t)y * w + x] = (int16_t)mm[1];
}
int census_levels(int w, int h, int scales)
{
    int L = 1;
    while (L < scales && std::min((w + 1) / 2, (h + 1) / 2) >= MS_MIN_DIM) { L++; w = (w + 1) / 2; h = (h + 1) / 2; }
    return L;
}

__global__ __launch_bounds__(256) void k_sum_S_u8(const uint8_t* __restrict__ C, const uint8_t* __restrict__ E, size_t vol,
                                                  int P2, int fixo, int nd, uint16_t* __restrict__ S)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= vol) return;
    int c = (int)C[i] + P2, s = 0;
    for (int r = 0; r < nd; r++) s += c - (int)E[(size_t)r * vol + i];
    S[i] = (uint16_t)(s - fixo * min((int)C[i], CENSUS_MAX_BITS));
}

// ---- MGM recursion (oracle/census_oracle.c, recursion = 1): two predecessors per direction -----------------
// L_r(p) = C(p) + (msg_{p-r} + msg_{p-r_perp} + 1) >> 1 with the SGM message of each predecessor.  The dependencies
// no longer run along independent 1-D paths: a direction advances as a FRONT -- the anti-diagonals x +- y = t for
// the 4 axis directions, the rows / columns for the 4 diagonal ones -- and one launch does step t of all 8
// directions (<= 8 x max(w, h) pixels, one lane group per pixel as in the path kernel).  Only the previous front
// of every direction is kept (a two-line ring of int16 costs + their minima, L2-resident); what leaves the kernel
// is the same e = P2 - (L - C) byte volume per direction, so S, the WTA and the consensus are shared with the path
// mode.  w + h - 1 dependent launches per tile: this mode trades speed for fidelity to the `mgm` binary
// (99.5 % of the reference's stored tile within 0.5 px instead of 98.9 %).
struct MgmArgs {
    const uint8_t* C; uint8_t* E; size_t vol;
    int w, h, D, P1, P2, t, lmax, nd;
    uint16_t* Lbuf;       // [8][2][lmax][D]
    int* Mbuf;            // [8][2][lmax]   min_k L
};

template <int K> __device__ __forceinline__ void store_line(__amdgpu_buffer_rsrc_t rs, uint32_t off, const uint32_t (&n)[K]);
template <> __device__ __forceinline__ void store_line<4>(__amdgpu_buffer_rsrc_t rs, uint32_t off, const uint32_t (&n)[4]) {
    u32x4 v; v.x = n[0]; v.y = n[1]; v.z = n[2]; v.w = n[3];
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)off, 0, 0);
}
template <> __device__ __forceinline__ void store_line<8>(__amdgpu_buffer_rsrc_t rs, uint32_t off, const uint32_t (&n)[8]) {
    u32x4 v; v.x = n[0]; v.y = n[1]; v.z = n[2]; v.w = n[3];
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)off, 0, 0);
    v.x = n[4]; v.y = n[5]; v.z = n[6]; v.w = n[7];
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)(off + 16u), 0, 0);
}

template <int G, int K, bool PAD>
__global__ __launch_bounds__(256) void k_mgm_step(MgmArgs a)
{
    constexpr int DPL = 2 * K, NP = 64 / G;
    typedef CostLoad<int16_t, K> LL;
    typedef CostLoad<uint8_t, K> CL;
    const int r = blockIdx.y, w = a.w, h = a.h, D = a.D, t = a.t;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, gl = lane & (G - 1);
    const int li = ((int)blockIdx.x * 4 + wave) * NP + lane / G;        // index of the pixel along its front line
    const bool lane_ok = PAD ? (gl * DPL < D) : true;
    const bool is_first = gl == 0, is_last = gl == G - 1;
    int dx, dy, x, y;
    bool colidx = false;                                                 // line indexed by y (column sweeps) instead of x
    switch (r) {                                                         // same direction table as the path kernel / oracle
        case 0: dx = 1; dy = 0; x = li; y = t - li; break;
        case 1: dx = -1; dy = 0; x = li; y = (h - 1) - (t - (w - 1 - li)); break;
        case 2: dx = 0; dy = 1; x = li; y = t - (w - 1 - li); break;
        case 3: dx = 0; dy = -1; x = li; y = (h - 1) - (t - li); break;
        case 4: dx = 1; dy = 1; x = li; y = t; break;
        case 5: dx = -1; dy = 1; x = w - 1 - t; y = li; colidx = true; break;
        case 6: dx = -1; dy = -1; x = li; y = h - 1 - t; break;
        default: dx = 1; dy = -1; x = t; y = li; colidx = true; break;
    }
    const bool live = li >= 0 && x >= 0 && x < w && y >= 0 && y < h;
    if (!__any(live)) return;
    const int ex = -dy, ey = dx;                                         // r_perp
    const size_t line = (size_t)a.lmax * D * 2;                          // bytes per line of Lbuf
    const size_t lbytes = (size_t)16 * line;
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.C), 0, (int)a.vol, S2P_BUF_FLAGS);
    const __amdgpu_buffer_rsrc_t rsE = __builtin_amdgcn_make_buffer_rsrc(a.E + (size_t)r * a.vol, 0, (int)a.vol, S2P_BUF_FLAGS);
    const __amdgpu_buffer_rsrc_t rsL = __builtin_amdgcn_make_buffer_rsrc(a.Lbuf, 0, (int)lbytes, S2P_BUF_FLAGS);
    const uint32_t prevL = (uint32_t)((size_t)(r * 2 + ((t + 1) & 1)) * line), curL = (uint32_t)((size_t)(r * 2 + (t & 1)) * line);
    const int* Mprev = a.Mbuf + (size_t)(r * 2 + ((t + 1) & 1)) * a.lmax;
    int* Mcur = a.Mbuf + (size_t)(r * 2 + (t & 1)) * a.lmax;
    const uint32_t P1pk = pk_dup(a.P1), P2pk = pk_dup(a.P2);

    uint32_t msum[K];                                                    // msg_{p-r} + msg_{p-r_perp}
    #pragma unroll
    for (int j = 0; j < K; j++) msum[j] = 0;
    #pragma unroll
    for (int n = 0; n < 2; n++) {
        const int qx = x - (n ? ex : dx), qy = y - (n ? ey : dy);
        const bool in = live && t > 0 && qx >= 0 && qx < w && qy >= 0 && qy < h;
        const int qli = colidx ? qy : qx;
        // a predecessor outside the image sends no message: all-zero costs with minimum 0 give msg = 0
        typename LL::raw_t raw = LL::load(rsL, (in && lane_ok) ? prevL + (uint32_t)((qli * D + gl * DPL) * 2) : S2P_OOB - 32u);
        const int m0 = in ? Mprev[qli] : 0;
        uint32_t lq[K];
        LL::unpack(raw, lq);
        if (PAD) {
            #pragma unroll
            for (int j = 0; j < K; j++) lq[j] = lane_ok ? lq[j] : BIGPK;
        }
        const uint32_t below = group_from_below<G>(lq[K - 1], BIGPK, is_first);
        const uint32_t above = group_from_above<G>(lq[0], BIGPK, is_last);
        const uint32_t delta = pk_dup(m0 + a.P2), m0pk = pk_dup(m0);
        #pragma unroll
        for (int j = 0; j < K; j++) {
            const uint32_t dm1 = __builtin_amdgcn_alignbit(lq[j], j ? lq[j - 1] : below, 16);
            const uint32_t dp1 = __builtin_amdgcn_alignbit(j < K - 1 ? lq[j + 1] : above, lq[j], 16);
            const uint32_t v = pk_min(pk_min(pk_add(pk_min(dm1, dp1), P1pk), lq[j]), delta);
            msum[j] += pk_sub(v, m0pk);                                  // fields <= P2: plain dword add, no carry
        }
    }
    uint32_t c[K], nl[K], e[K];
    CL::unpack(CL::load(rsC, (live && lane_ok) ? (uint32_t)(((size_t)y * w + x) * D + gl * DPL) : S2P_OOB), c);
    #pragma unroll
    for (int j = 0; j < K; j++) {
        const uint32_t m = ((msum[j] + 0x00010001u) >> 1) & 0x7fff7fffu; // (a + b + 1) >> 1 on both 16-bit fields
        nl[j] = pk_add(c[j], m);
        e[j] = pk_sub(P2pk, m);
        if (PAD) nl[j] = lane_ok ? nl[j] : BIGPK;
    }
    const bool st = live && lane_ok;
    store_e<K>(rsE, st ? (uint32_t)(((size_t)y * w + x) * D + gl * DPL) : S2P_OOB, e);
    store_line<K>(rsL, st ? curL + (uint32_t)((li * D + gl * DPL) * 2) : S2P_OOB - 32u, nl);
    uint32_t mm = pk_min(pk_min(nl[0], nl[1]), pk_min(nl[2], nl[3]));
    #pragma unroll
    for (int j = 4; j < K; j += 4) mm = pk_min(mm, pk_min(pk_min(nl[j], nl[j + 1]), pk_min(nl[j + 2], nl[j + 3])));
    const int mn = group_min_i32<G>(min(pk_lo(mm), pk_hi(mm)));
    if (live && gl == 0) Mcur[li] = mn;
}

template <int G, int K>
static void launch_mgm_step(hipStream_t st, int nblocks, bool pad, const MgmArgs& a) {
    if (pad) hipLaunchKernelGGL((k_mgm_step<G, K, true>), dim3(nblocks, a.nd), dim3(256), 0, st, a);
    else     hipLaunchKernelGGL((k_mgm_step<G, K, false>), dim3(nblocks, a.nd), dim3(256), 0, st, a);
}
static void enqueue_mgm(hipStream_t st, const uint8_t* C, uint8_t* E, int w, int h, int D, int P1, int P2, uint16_t* Lbuf, int* Mbuf, int nd)
{
    MgmArgs a;
    a.nd = nd;
    a.C = C; a.E = E; a.vol = (size_t)w * h * D; a.w = w; a.h = h; a.D = D; a.P1 = P1; a.P2 = P2; a.lmax = std::max(w, h);
    a.Lbuf = Lbuf; a.Mbuf = Mbuf;
    const LaneLayout ll = lane_layout(D);
    const int per_block = 4 * (64 / ll.G), nblocks = (a.lmax + per_block - 1) / per_block;
    for (int t = 0; t < w + h - 1; t++) {
        a.t = t;
        if (ll.K == 8) launch_mgm_step<64, 8>(st, nblocks, ll.pad, a);
        else switch (ll.G) {
            case 2: launch_mgm_step<2, 4>(st, nblocks, ll.pad, a); break;
            case 4: launch_mgm_step<4, 4>(st, nblocks, ll.pad, a); break;
            case 8: launch_mgm_step<8, 4>(st, nblocks, ll.pad, a); break;
            case 16: launch_mgm_step<16, 4>(st, nblocks, ll.pad, a); break;
            case 32: launch_mgm_step<32, 4>(st, nblocks, ll.pad, a); break;
            default: launch_mgm_step<64, 4>(st, nblocks, ll.pad, a); break;
        }
    }
}


// ---- ZNCC cost volume (s2p_census_params.cost = 1): north_star's "census/ZNCC".  No call site of the reference selects it
// (`-t census` is hard-coded, s2p/block_matching.py:171,293): unpinned; the statement is oracle/census_oracle.c (zncc_stats /
// zncc_cost), and this kernel matches it bit for bit -- float32, the sums in raster order, separate multiply and add
// (-ffp-contract=off), correctly rounded division and square root.  One block per image row: the WIN rows of both images
// around it are staged in LDS with their replicated borders (the window clamps its coordinates to the image), every pixel of
// image 2 gets its window variance (negative = a NaN in the window), then work items of (pixel, 16 consecutive candidates)
// slide a WIN x WIN register window along the candidates (WIN new LDS reads per candidate) and store their 16 cost bytes with
// one 16-byte store.  Quantised to the census scale: clamp(floor((1 - zncc) 12 + 0.5), 0, 24).
// SP = 2 (half-pixel candidates, round 5): candidate i stands for dmin + i / 2; the odd ones correlate with image 2 sampled half way
// between its columns (0.5 (b[x] + b[min(x + 1, w - 1)]), the im2h of the oracle) -- its rows, window variances and validity are staged
// beside the whole-pixel ones, and a slice of 16 candidates slides TWO register windows, one step per pair of candidates.
static inline size_t zncc_cost_lds(int w, int win, int sp = 1) { return (size_t)(1 + sp) * win * (w + 2 * (win / 2)) * 4 + (size_t)sp * w * 4; }
template <int WIN, int SP>
__global__ __launch_bounds__(256) void k_zncc_cost(const float* __restrict__ im1, const float* __restrict__ im2, int h, int w,
                                                   int dmin, int Dt, int D, const int16_t* __restrict__ lo, const int16_t* __restrict__ hi,
                                                   uint8_t* __restrict__ C)
{
    constexpr int R = WIN / 2, N = WIN * WIN;
    extern __shared__ __attribute__((aligned(16))) uint8_t sm[];
    const int wp = w + 2 * R;                            // padded row: entry i = pixel clamp(i - R)
    float* ra = reinterpret_cast<float*>(sm);            // [WIN][wp] image 1
    float* rb = ra + WIN * wp;                           // [WIN][wp] image 2
    float* rh = rb + WIN * wp;                           // [WIN][wp] image 2 half way between its columns (SP == 2)
    float* vb = rh + (SP == 2 ? WIN * wp : 0);           // [w] window variance of image 2 (< 0: a NaN in the window)
    float* vh = vb + w;                                  // [w] ... of the half-sampled image (SP == 2)
    const int y = blockIdx.x;
    for (int i = threadIdx.x; i < WIN * wp; i += 256) {
        const int j = i / wp, c = i - j * wp;
        const int px = min(max(c - R, 0), w - 1);
        const size_t row = (size_t)min(max(y + j - R, 0), h - 1) * w;
        ra[i] = im1[row + px]; rb[i] = im2[row + px];
        if (SP == 2) rh[i] = __fmul_rn(0.5f, __fadd_rn(im2[row + px], im2[row + min(px + 1, w - 1)]));
    }
    __syncthreads();
    for (int xs = threadIdx.x; xs < SP * w; xs += 256) {
        const int x = xs % w;
        const float* src = xs < w ? rb : rh;
        float v[N], sum = 0.0f;
        bool fin = true;
        #pragma unroll
        for (int j = 0; j < WIN; j++)
            #pragma unroll
            for (int i = 0; i < WIN; i++) { const float t = src[j * wp + x + i]; v[j * WIN + i] = t; fin = fin && isfinite(t); sum = sum + t; }
        const float mean = sum / (float)N;
        float s2 = 0.0f;
        #pragma unroll
        for (int k = 0; k < N; k++) { const float c = v[k] - mean; s2 = s2 + c * c; }
        (xs < w ? vb : vh)[x] = fin ? s2 : -1.0f;
    }
    __syncthreads();
    const int nsl = D >> 4;                              // slices of 16 candidates
    uint8_t* Crow = C + (size_t)y * w * D;
    for (int it = threadIdx.x; it < w * nsl; it += 256) {
        const int x = it / nsl, s = it - x * nsl;
        float ac[N], sum = 0.0f;
        bool ok1 = true;
        #pragma unroll
        for (int j = 0; j < WIN; j++)
            #pragma unroll
            for (int i = 0; i < WIN; i++) { const float t = ra[j * wp + x + i]; ac[j * WIN + i] = t; ok1 = ok1 && isfinite(t); sum = sum + t; }
        const float mean = sum / (float)N;
        float va = 0.0f;
        #pragma unroll
        for (int k = 0; k < N; k++) { ac[k] = ac[k] - mean; va = va + ac[k] * ac[k]; }
        int jlo = 0, jhi = Dt - 1;
        if (lo) { jlo = SP * ((int)lo[(size_t)y * w + x] - dmin); jhi = min(jhi, SP * ((int)hi[(size_t)y * w + x] - dmin)); }
        uint32_t out[4] = {0, 0, 0, 0};
        float bw[WIN][WIN];                              // bw[j][i]: image-2 sample of window row j, column x2 - R + i
        float bh[SP == 2 ? WIN : 1][SP == 2 ? WIN : 1];  // the same of the half-sampled image
        const int c0 = s * 16;
        // candidate c0 + c stands for the disparity dmin + (c0 + c) / SP: SP dmin + c0 is a multiple of SP, so the phase of a candidate
        // is c mod SP and its column x + dmin + (c0 + c) / SP
        const int xfirst = x + dmin + c0 / SP;
        {   // the window(s) of the slice's first candidate minus the last column (loaded in the loop)
            #pragma unroll
            for (int j = 0; j < WIN; j++)
                #pragma unroll
                for (int i = 1; i < WIN; i++) {
                    const int col = j * wp + min(max(xfirst + i - 1, 0), wp - 1);
                    bw[j][i] = rb[col];
                    if constexpr (SP == 2) bh[j][i] = rh[col];
                }
        }
        #pragma unroll
        for (int cs = 0; cs < 16 / SP; cs++) {           // one column of image 2 per turn: SP candidates
            const int x2 = xfirst + cs;
            #pragma unroll
            for (int j = 0; j < WIN; j++) {
                #pragma unroll
                for (int k = 0; k < WIN - 1; k++) {
                    bw[j][k] = bw[j][k + 1];
                    if constexpr (SP == 2) bh[j][k] = bh[j][k + 1];
                }
                const int col = j * wp + min(max(x2 + 2 * R, 0), wp - 1);
                bw[j][WIN - 1] = rb[col];
                if constexpr (SP == 2) bh[j][WIN - 1] = rh[col];
            }
            const bool in2 = x2 >= 0 && x2 < w;
            const int xc = min(max(x2, 0), w - 1);
            #pragma unroll
            for (int ph = 0; ph < SP; ph++) {
                const int c = SP * cs + ph, i = c0 + c;
                uint32_t cost = C_EXCLUDED;
                const float vbx = ph ? vh[xc] : vb[xc];
                if (i <= jhi && i >= jlo && ok1 && in2 && vbx >= 0.0f) {
                    float cov = 0.0f;
                    #pragma unroll
                    for (int j = 0; j < WIN; j++)
                        #pragma unroll
                        for (int k = 0; k < WIN; k++) {
                            float bs = bw[j][k];
                            if constexpr (SP == 2) bs = ph ? bh[j][k] : bs;
                            cov = cov + ac[j * WIN + k] * bs;
                        }
                    const float den = va * vbx;
                    const float z = den > 0.0f ? __fdiv_rn(cov, __fsqrt_rn(den)) : 0.0f;
                    float q = floorf((1.0f - z) * 12.0f + 0.5f);
                    if (!(q >= 0.0f)) q = 0.0f;
                    if (q > 24.0f) q = 24.0f;
                    cost = (uint32_t)q;
                }
                out[c >> 2] |= cost << (8 * (c & 3));
            }
        }
        u32x4 v; v.x = out[0]; v.y = out[1]; v.z = out[2]; v.w = out[3];
        *reinterpret_cast<u32x4*>(Crow + (size_t)x * D + c0) = v;
    }
}

// ---- MGM recursion, band-pipelined: ONE launch per tile (mgm_bands.hpp) ------------------------------------------
}  // namespace s2p
#include "mgm_bands.hpp"
namespace s2p {
#ifndef S2P_MGM_BATCH_STAGGER
#define S2P_MGM_BATCH_STAGGER -1    // >= 0: a batch lets tile t + 1 in when lattice q of tile t is (x / 256) of its bands in; -1: every tile's
                                    // lattices are in the queue from the start (measured best: profiles/r03/batch_sweep*.txt)
#endif
#ifndef S2P_MGM_DEFAULT_BANDS
#define S2P_MGM_DEFAULT_BANDS 1       // 0: the front-by-front kernel (kept as the in-process cross-check of the tests)
#endif
// which implementation serves recursion = 1: "bands" (one launch) or "steps" (one launch per front); S2P_MGM_IMPL
// overrides the default for A/B measurements (read at every call: the tests flip it inside one process)
static int mgm_impl_bands() { const char* e = getenv("S2P_MGM_IMPL"); return e && *e ? (strcmp(e, "steps") != 0) : S2P_MGM_DEFAULT_BANDS; }
static size_t mgm_workspace_bytes(int w, int h, int D, int nd = 8) {
    const size_t lmax = (size_t)std::max(w, h);
    const size_t steps = align_up(16 * lmax * D * 2, 256) + align_up(16 * lmax * 4, 256) + 512;
    return std::max(steps, mgm_bands_workspace_bytes(w, h, D, 1, mgm_nlat(nd)));
}
// e-volumes of a tile: one per direction, 8 of them also when only the 4 axis directions are summed (the layout of round 1), 16 with
// the knight's moves (nb_dir = 16)
static inline size_t census_planes(int nd) { return nd > 8 ? 16 : 8; }
// ---- WTA + right view + vfit + left-right test (+ optional per-direction consensus) ---------------
struct CensusWtaArgs {
    const uint8_t* C; const uint8_t* E; size_t vol;
    int w, h, D, Dt, dmin, P2, lr_check, tau;   // tau in candidates
    int sp;               // candidates per pixel of disparity (1, or 2 = half-pixel grid)
    int fixo;             // nd - 1 with the overcount fix (S = sum_r L_r - (nd - 1) min(C, 24)), else 0
    int nd, sh;           // directions summed (8, or 4 = the axis ones: the e-volumes beyond are neither written nor read), log2(nd)
    float* disp;          // h*w, pre-median
    float* conf;          // h*w consensus / 8 (may be null when CONF == false)
    int mindiff;          // MINDIFF: > 0 = a winner must beat every non-neighbouring candidate by this much (in units of S), else NaN
    int byte_keys;        // host: the consensus kernel may use its (L << 8) | index keys -- census costs and P2 <= 63 (see the kernel)
    const int* win;       // null, or device {lo, hi}: the disparities this TILE's level really searches when the volume covers more
                          // (a batch of multi-scale tiles shares one volume shape, the hull of the tiles' ranges): candidates outside
                          // are excluded in the volume, and here they neither compete for the right view nor bound the V fit --
                          // the winner at lo / hi is the range's edge, exactly as if the volume ended there (lo > hi: no narrowing)
};

// On packed 16-bit fields (a first version on unpacked ints spent ~60 % of its issue slots unpacking bytes).
// Per lane and pixel:
//   * sum_r e_r: QUAD (P2 <= 63, so 4 bytes sum below 256) adds the e-dwords of 4 directions as plain
//     dwords before one v_perm split into 16-bit pairs; otherwise every dword is split first;
//   * S = 8 (C + P2) - sum e as dword arithmetic on the pairs (no field ever borrows: S >= 0, S <= 4080);
//   * lane arg-min on 16-bit keys (S << log2(DPL)) | j with v_pk_min_u16, then the group butterfly;
//   * the +-1 neighbours of the winner by a register select tree instead of 2*DPL compare/selects;
//   * right view in an LDS array indexed by x + i (= x2 - dmin): no per-candidate range test.
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) {
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    u16x2 r = __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b));
    return __builtin_bit_cast(uint32_t, r);
}
template <int K> __device__ __forceinline__ void raw_words(typename EBytes<K>::raw_t v, uint32_t (&w)[K / 2]);
template <> __device__ __forceinline__ void raw_words<4>(u32x2 v, uint32_t (&w)[2]) { w[0] = v.x; w[1] = v.y; }
template <> __device__ __forceinline__ void raw_words<6>(u32x3 v, uint32_t (&w)[3]) { w[0] = v.x; w[1] = v.y; w[2] = v.z; }
template <> __device__ __forceinline__ void raw_words<8>(u32x4 v, uint32_t (&w)[4]) { w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; }

// NE: e-volumes per tile (8; 16 with the knight's moves of nb_dir = 16 -- only built as the CONF variants)
template <int G, int K, bool PAD, bool QUAD, bool CONF, bool MD = false, int NE = 8>
__global__ __launch_bounds__(S2P_WTA_NT) void k_wta_census_pk(CensusWtaArgs a)
{
    constexpr int DPL = 2 * K, NW = K / 2, SH = K == 4 ? 3 : 4;   // disparities per lane, dwords per lane, bits of a lane-relative index (K = 6: 12 per lane)
    constexpr int NT = S2P_WTA_NT, NWV = NT / 64;                  // threads, waves per block (one block = one row)
    extern __shared__ __attribute__((aligned(16))) uint8_t sm[];
    const int w = a.w, D = a.D, y = blockIdx.x;
    const int sp = a.sp, nslot = sp * w + D;
    uint32_t* rkey = reinterpret_cast<uint32_t*>(sm);           // [sp w + D]  (S << 16) | i, at slot sp x + i: the (half-)pixel of image 2
    float* dsub = reinterpret_cast<float*>(rkey + nslot);       // [w]  left disparity incl. vfit offset
    int16_t* bl = reinterpret_cast<int16_t*>(dsub + w);         // [w]  left winner index or -1
    for (int x = threadIdx.x; x < nslot; x += NT) rkey[x] = 0xffffffffu;
    for (int x = threadIdx.x; x < w; x += NT) bl[x] = -1;
    __syncthreads();

    constexpr int NP = 64 / G;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gl = lane & (G - 1);
    const bool lane_ok = PAD ? (gl * DPL < D) : true;
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.C), 0, (int)a.vol, S2P_BUF_FLAGS);
    __amdgpu_buffer_rsrc_t rsE[NE];
    #pragma unroll
    for (int r = 0; r < NE; r++) rsE[r] = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.E) + (size_t)r * a.vol, 0, (int)a.vol, S2P_BUF_FLAGS);
    const uint32_t rowoff = (uint32_t)((size_t)y * w * D);
    typedef EBytes<K> EL;                                       // C and e are both DPL bytes per lane
    struct Px { typename EL::raw_t c; typename EL::raw_t e[NE]; };
    auto issue = [&](int xb) __attribute__((always_inline)) -> Px {
        const int x = xb + wave * NP + lane / G;
        const uint32_t off = (x < w && lane_ok) ? rowoff + (uint32_t)(x * D + gl * DPL) : S2P_OOB;
        Px p;
        p.c = EL::load(rsC, off);
        #pragma unroll
        for (int r = 0; r < NE; r++) p.e[r] = EL::load(rsE[r], r < a.nd ? off : S2P_OOB);     // (out of range: 0, no traffic)
        return p;
    };
    const uint32_t p2pk = pk_dup(a.P2), cmaxpk = pk_dup(CENSUS_MAX_BITS);
    const int s_excluded = a.nd * C_EXCLUDED - a.fixo * CENSUS_MAX_BITS;   // every excluded candidate sums to at least this
    const int sh = a.sh;
    int j0 = 0, j1 = a.Dt - 1;                                  // the tile's own candidate window inside the volume
    if (a.win) { const int lo = a.win[0], hi = a.win[1]; if (lo <= hi) { j0 = a.sp * (lo - a.dmin); j1 = a.sp * (hi - a.dmin); } }
    const int jlim = j1 + 1 - gl * DPL, jlo = j0 - gl * DPL;    // candidates j >= jlim (padding) or j < jlo of this lane are not the tile's
    uint32_t kill[DPL];                                         // ... as an OR mask per candidate of this lane: all ones = it does not compete
    #pragma unroll
    for (int j = 0; j < DPL; j++) kill[j] = (j < jlim && j >= jlo) ? 0u : 0xffffffffu;
    // software pipeline: the 9 loads (C + 8 e-volumes) of the next PFW pixel groups are in flight while the
    // current one is reduced (statically named register sets -> counted vmcnt waits)
    constexpr int PFW = S2P_WTA_PF;
    Px q[PFW];
    #pragma unroll
    for (int u = 0; u < PFW; u++) q[u] = issue(u * NWV * NP);
    auto process = [&](const Px& cur, const int xb) __attribute__((always_inline)) {
        const int x = xb + wave * NP + lane / G;
        const bool ok = x < w && lane_ok;
        uint32_t cw[NW], S[K];                                  // S[p] = (S(2p), S(2p + 1)) as 16-bit fields
        raw_words<K>(cur.c, cw);
        uint32_t ew[NE][NW];
        #pragma unroll
        for (int r = 0; r < NE; r++) raw_words<K>(cur.e[r], ew[r]);
        // CONF: per-direction arg-min of L_r = (C + P2) - e_r on the same 16-bit keys (the unpacked e pairs also feed
        // the sum, so QUAD is not used then)
        uint32_t dirmin[NE];
        if (CONF) {
            #pragma unroll
            for (int r = 0; r < NE; r++) dirmin[r] = 0xffffffffu;
        }
        #pragma unroll
        for (int i = 0; i < NW; i++) {
            uint32_t c0, c1, s0, s1;
            bytes_to_pairs(cw[i], c0, c1);
            if (CONF && QUAD) {
                // census costs (<= 24, or 255 = excluded) with P2 <= 63: L_r <= 87 for a valid candidate and min(C + P2, 255) - e_r >= 192 for an
                // excluded one, in the order of the true L_r among themselves -- so the per-direction keys fit (L << 8) | index, the byte e_r
                // goes straight to the high byte of its field with the one v_perm it needs anyway, and (x << 8 | j) - (e << 8) is the key (x >= e:
                // no borrow between the fields): v_perm + v_sub + v_pk_min per pair instead of v_perm + 2 adds ... + v_sub + v_lshl_or + v_pk_min;
                // the sum over the directions takes the byte-wise quad adds of the plain kernel
                const uint32_t x0 = pk_min_u16(c0 + p2pk, 0x00ff00ffu), x1 = pk_min_u16(c1 + p2pk, 0x00ff00ffu);
                const uint32_t base0 = (x0 << 8) | (uint32_t)((4 * i) | ((4 * i + 1) << 16)), base1 = (x1 << 8) | (uint32_t)((4 * i + 2) | ((4 * i + 3) << 16));
                #pragma unroll
                for (int r = 0; r < NE; r++) {
                    const uint32_t f0 = __builtin_amdgcn_perm(0u, ew[r][i], 0x010c000cu), f1 = __builtin_amdgcn_perm(0u, ew[r][i], 0x030c020cu);
                    dirmin[r] = pk_min_u16(dirmin[r], base0 - f0);
                    dirmin[r] = pk_min_u16(dirmin[r], base1 - f1);
                }
                s0 = 0; s1 = 0;
                #pragma unroll
                for (int g = 0; g < NE; g += 4) {
                    const uint32_t qg = (ew[g][i] + ew[g + 1][i]) + (ew[g + 2][i] + ew[g + 3][i]);
                    uint32_t a0, a1;
                    bytes_to_pairs(qg, a0, a1);
                    s0 += a0; s1 += a1;
                }
            } else if (CONF) {
                s0 = 0; s1 = 0;
                const uint32_t cc0 = c0 + p2pk, cc1 = c1 + p2pk;
                #pragma unroll
                for (int r = 0; r < NE; r++) {
                    uint32_t a0, a1;
                    bytes_to_pairs(ew[r][i], a0, a1);
                    s0 += a0; s1 += a1;
                    dirmin[r] = pk_min_u16(dirmin[r], ((cc0 - a0) << SH) | (uint32_t)((4 * i) | ((4 * i + 1) << 16)));
                    dirmin[r] = pk_min_u16(dirmin[r], ((cc1 - a1) << SH) | (uint32_t)((4 * i + 2) | ((4 * i + 3) << 16)));
                }
            } else if (QUAD) {
                const uint32_t q0 = (ew[0][i] + ew[1][i]) + (ew[2][i] + ew[3][i]), q1 = (ew[4][i] + ew[5][i]) + (ew[6][i] + ew[7][i]);
                uint32_t a0, a1, b0, b1;
                bytes_to_pairs(q0, a0, a1); bytes_to_pairs(q1, b0, b1);
                s0 = a0 + b0; s1 = a1 + b1;
                #pragma unroll
                for (int g = 8; g < NE; g += 4) {
                    const uint32_t qg = (ew[g][i] + ew[g + 1][i]) + (ew[g + 2][i] + ew[g + 3][i]);
                    bytes_to_pairs(qg, a0, a1);
                    s0 += a0; s1 += a1;
                }
            } else {
                s0 = 0; s1 = 0;
                #pragma unroll
                for (int r = 0; r < NE; r++) { uint32_t a0, a1; bytes_to_pairs(ew[r][i], a0, a1); s0 += a0; s1 += a1; }
            }
            S[2 * i] = ((c0 + p2pk) << sh) - s0;
            S[2 * i + 1] = ((c1 + p2pk) << sh) - s1;
            if (a.fixo) {       // data term counted once: - 7 min(C, 24) (exact for valid candidates, excluded ones stay on top)
                const uint32_t m0 = pk_min_u16(c0, cmaxpk), m1 = pk_min_u16(c1, cmaxpk);
                S[2 * i] -= (m0 << sh) - m0;
                S[2 * i + 1] -= (m1 << sh) - m1;
            }
        }
        // lane arg-min: 16-bit keys (S << SH) | j; ties -> smallest j (oracle: first minimum in d order)
        uint32_t m = 0xffffffffu;
        #pragma unroll
        for (int p = 0; p < K; p++) {
            // 16 directions: an excluded candidate may sum to 16 (255 + P2) > 4095, which does not fit a key with a 4-bit index;
            // capped there it still is >= s_excluded (<= 4080) and above every valid candidate (<= 16 (24 + P2) <= 2432)
            const uint32_t sk = (NE > 8 && SH == 4) ? pk_min_u16(S[p], 0x0fff0fffu) : S[p];
            m = pk_min_u16(m, (sk << SH) | (uint32_t)((2 * p) | ((2 * p + 1) << 16)));
        }
        const uint32_t m16 = min(m & 0xffffu, m >> 16);
        uint32_t key = ((m16 >> SH) << 16) | (uint32_t)(gl * DPL + (int)(m16 & ((1u << SH) - 1u)));
        key = ok ? key : 0xffffffffu;
        key = group_min_u32<G>(key);
        const int minS = (int)(key >> 16), best = (int)(key & 0xffffu);
        // right view: every candidate of the true range competes for its pixel of image 2 (slot x + i)
        if (ok) {
            uint32_t* slot = rkey + sp * x + gl * DPL;
            #pragma unroll
            for (int p = 0; p < K; p++) {
                const uint32_t k0 = (S[p] << 16) | (uint32_t)(gl * DPL + 2 * p), k1 = (S[p] & 0xffff0000u) | (uint32_t)(gl * DPL + 2 * p + 1);
                atomicMin(slot + 2 * p, k0 | kill[2 * p]);
                atomicMin(slot + 2 * p + 1, k1 | kill[2 * p + 1]);
            }
        }
        // S(best - 1), S(best + 1): each lives in one lane of the group, at a lane-relative index in [0, DPL)
        // (scalars selected by shifts and masks: a select chain over the S[] array is turned into a scratch array)
        const uint64_t w0 = (uint64_t)S[0] | ((uint64_t)S[1] << 32), w1 = (uint64_t)S[2] | ((uint64_t)S[3] << 32);
        const uint64_t w2 = K >= 6 ? (uint64_t)S[4 % K] | ((uint64_t)S[5 % K] << 32) : 0, w3 = K == 8 ? (uint64_t)S[6 % K] | ((uint64_t)S[7 % K] << 32) : 0;
        auto pick = [ok, w0, w1, w2, w3](int t) __attribute__((always_inline)) -> int {   // S at lane-relative index t, 0 if not ours
            const uint64_t m4 = 0 - (uint64_t)((t >> 2) & 1), m8 = 0 - (uint64_t)((t >> 3) & 1);
            const uint64_t a0 = (w0 & ~m4) | (w1 & m4), a1 = (w2 & ~m4) | (w3 & m4);
            uint64_t v = K >= 6 ? (a0 & ~m8) | (a1 & m8) : a0;
            v >>= 16 * (t & 3);
            return (ok && (unsigned)t < (unsigned)DPL) ? (int)(v & 0xffffu) : 0;
        };
        const int tb = best - gl * DPL;
        const int packed = group_or_i32<G>(pick(tb - 1) | (pick(tb + 1) << 16));
        int second = 0x7fffffff;                                 // MD: the smallest S of the tile's candidates that are not the winner or its neighbours
        if (MD) {
            uint32_t m2 = 0xffffffffu;
            #pragma unroll
            for (int p = 0; p < K; p++) {
                const int ja = 2 * p, jb = 2 * p + 1;
                const bool ua = kill[ja] == 0u && abs(ja - tb) > 1, ub = kill[jb] == 0u && abs(jb - tb) > 1;
                m2 = pk_min_u16(m2, (ua ? (S[p] & 0xffffu) : 0xffffu) | (ub ? (S[p] & 0xffff0000u) : 0xffff0000u));
            }
            const uint32_t s2 = ok ? min(m2 & 0xffffu, m2 >> 16) : 0xffffu;
            second = (int)group_min_u32<G>(s2);
        }
        int agree = 0;
        if (CONF) {
            #pragma unroll
            for (int r = 0; r < NE; r++) {
                const uint32_t m16r = min(dirmin[r] & 0xffffu, dirmin[r] >> 16);
                uint32_t kr = QUAD ? ((m16r >> 8) << 16) | (uint32_t)(gl * DPL + (int)(m16r & 0xffu))
                                   : ((m16r >> SH) << 16) | (uint32_t)(gl * DPL + (int)(m16r & ((1u << SH) - 1u)));
                kr = ok ? kr : 0xffffffffu;
                const int arg = (int)(group_min_u32<G>(kr) & 0xffffu);
                agree += (r < a.nd && abs(arg - best) <= 1) ? 1 : 0;
            }
        }
        if (x < w && gl == 0) {
            const bool valid = minS < s_excluded;
            float off = 0.0f;
            if (valid && best > j0 && best < j1) {
                const int smv = packed & 0xffff, spv = (int)((uint32_t)packed >> 16);
                const int den = max(smv - minS, spv - minS);
                if (den > 0) off = __fmul_rn(0.5f, __fdiv_rn((float)(smv - spv), (float)den));
            }
            const bool beaten = MD && second != 0xffff && second - minS < a.mindiff;     // (0xffff: no other candidate at all)
            bl[x] = (valid && !beaten) ? (int16_t)best : (int16_t)-1;
            dsub[x] = sp == 1 ? __fadd_rn((float)(a.dmin + best), off) : __fmul_rn(0.5f, __fadd_rn((float)(2 * a.dmin + best), off));
            if (CONF) a.conf[(size_t)y * w + x] = valid ? __fdiv_rn((float)agree, (float)a.nd) : __builtin_nanf("");
        }
    };
    for (int xb = 0; xb < w; xb += NWV * NP * PFW) {
        #pragma unroll
        for (int u = 0; u < PFW; u++) {
            const Px cur = q[u];
            q[u] = issue(xb + (u + PFW) * NWV * NP);
            process(cur, xb + u * NWV * NP);
        }
    }
    __syncthreads();
    for (int x = threadIdx.x; x < w; x += NT) {
        const int b = bl[x];
        float out = __builtin_nanf("");
        if (b >= 0) {
            bool keep = true;
            if (a.lr_check) keep = abs((int)(rkey[sp * x + b] & 0xffffu) - b) <= a.tau;   // (a valid winner points inside image 2)
            if (keep) out = dsub[x];
        }
        a.disp[(size_t)y * w + x] = out;
    }
}

// rejection mask (s2p/block_matching.py:18-32) + confidence masking of one pixel whose final disparity is d
__device__ __forceinline__ void census_epilogue_px(float d, int x, int y, int w, const float* __restrict__ im1, const float* __restrict__ im2,
                                                   float* __restrict__ conf, uint8_t* __restrict__ mask)
{
    const size_t i = (size_t)y * w + x;
    const bool fin = isfinite(d);
    if (conf && !fin) conf[i] = __builtin_nanf("");
    if (mask) {
        bool ok = fin && isfinite(im1[i]);
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

// ---- 3x3 median over the finite values of the window (centre finite), element (n-1)/2 -------------
// EPI: the median is the final disparity (no small-component filter after it): the epilogue is applied on the spot.
template <bool EPI>
__global__ __launch_bounds__(256) void k_median_valid(const float* __restrict__ src, float* __restrict__ dst, int w, int h,
                                                      const float* __restrict__ im1, const float* __restrict__ im2,
                                                      float* __restrict__ conf, uint8_t* __restrict__ mask)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const float c = src[(size_t)y * w + x];
    if (!isfinite(c)) { dst[(size_t)y * w + x] = c; if (EPI) census_epilogue_px(c, x, y, w, im1, im2, conf, mask); return; }
    float v[9];
    int n = 0;
    #pragma unroll
    for (int dy = -1; dy <= 1; dy++)
        #pragma unroll
        for (int dx = -1; dx <= 1; dx++) {
            const int xx = x + dx, yy = y + dy;
            float t = __builtin_inff();                          // +inf sorts behind every finite value
            if (xx >= 0 && xx < w && yy >= 0 && yy < h) { float s = src[(size_t)yy * w + xx]; if (isfinite(s)) { t = s; n++; } }
            v[(dy + 1) * 3 + dx + 1] = t;
        }
    // full sort of 9 with the optimal 25-comparator network (registers only), then pick element (n-1)/2
    #define SW(i, j) { float lo = fminf(v[i], v[j]), hi = fmaxf(v[i], v[j]); v[i] = lo; v[j] = hi; }
    SW(0, 3) SW(1, 7) SW(2, 5) SW(4, 8) SW(0, 7) SW(2, 4) SW(3, 8) SW(5, 6) SW(0, 2) SW(1, 3) SW(4, 5) SW(7, 8)
    SW(1, 4) SW(3, 6) SW(5, 7) SW(0, 1) SW(2, 4) SW(3, 5) SW(6, 8) SW(2, 3) SW(4, 5) SW(6, 7) SW(1, 2) SW(3, 4) SW(5, 6)
    #undef SW
    const int k = (n - 1) >> 1;
    float out = v[0];
    #pragma unroll
    for (int i = 1; i < 9; i++) out = (i == k) ? v[i] : out;
    dst[(size_t)y * w + x] = out;
    if (EPI) census_epilogue_px(out, x, y, w, im1, im2, conf, mask);
}

// ---- small-component removal on the float map through the shared int16 CCL ------------------------
#define Q_INVALID (-32768)
__global__ __launch_bounds__(256) void k_f32_to_q16(const float* __restrict__ d, size_t n, int16_t* __restrict__ q)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v = d[i];
    q[i] = isfinite(v) ? (int16_t)(int)rintf(__fmul_rn(v, 16.0f)) : (int16_t)Q_INVALID;
}
__global__ __launch_bounds__(256) void k_q16_apply(const int16_t* __restrict__ q, size_t n, float* __restrict__ d)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (q[i] == Q_INVALID) d[i] = __builtin_nanf("");
}

// ---- epilogue: rejection mask (s2p/block_matching.py:18-32) + confidence masking --------------------
__global__ __launch_bounds__(256) void k_census_epilogue(const float* __restrict__ disp, const float* __restrict__ im1,
                                                         const float* __restrict__ im2, int w, int h,
                                                         float* __restrict__ conf, uint8_t* __restrict__ mask)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    census_epilogue_px(disp[(size_t)y * w + x], x, y, w, im1, im2, conf, mask);
}

// ---- host side -----------------------------------------------------------------------------------------
// Depth of the cost / e-volumes of a range: the candidates rounded up to 16 (what the oracle lays out: oracle/census_oracle.c), the surplus
// excluded (cost 255).  Round 6: with the MGM recursion the depth is rounded up to 64 instead -- a pixel's candidates then start and end on
// 64-byte boundaries and every store of the band kernel writes whole lines.  Measured on 8 tiles of 1024^2 per launch
// (profiles/r06/pitch_probe.txt): 112 candidates in a volume of depth 112 take 6.3 ms, in a volume of depth 128 -- more bytes -- 4.1, as 128
// candidates do; partial lines (the lanes of the padding not storing) at depth 128 take 5.7-6.4.  The surplus candidates change no result
// for P2 <= 115: the argument of census_batches() below (an excluded candidate's L is at least every valid pixel's min L + P2), which the
// batches of tiles of different ranges have relied on since round 4; above that, and whenever the stage dumps are asked for (their
// layout is the oracle's), the depth stays at the multiple of 16.
int census_D(const s2p_census_params& p, int dmin, int dmax, bool dumps = false)
{
    const int sp = p.subpix == 2 ? 2 : 1;
    const int D = (sp * (dmax - dmin) + 1 + 15) / 16 * 16;
#ifdef S2P_CENSUS_DEPTH16            // (A/B build of tools/depth_probe.sh: the packed layout of rounds 1-5; same results)
    return D;
#endif
    if (dumps || p.recursion < 1 || p.P2 > 115 || D <= 32) return D;
    return (D + 63) / 64 * 64;
}
static size_t census_level_bytes(int w, int h, int D, bool want_S, int nd = 8)
{
    const size_t npx = (size_t)w * h, vol = npx * D;
    size_t n = 0;
    auto add = [&](size_t b) { n += align_up(b, 256); };
    add(npx * 4); add(npx * 4);            // census
    add(vol); add(vol * census_planes(nd)); // C, E
    if (want_S) add(vol * 2);
    add(npx * 4); add(npx * 4);            // disp_raw, disp_med
    add(npx * 2);                          // q16
    add(npx * 4); add(npx * 4); add(npx * 4);   // CCL
    return n + mgm_workspace_bytes(w, h, D, nd) + 4096;
}
// geometry of the pyramid of the multi-scale mode: level 0 = the tile itself
struct CensusPyramid { int L; int w[16], h[16], dmin[16], dmax[16]; };
static CensusPyramid census_pyramid(const s2p_census_params& p, int w, int h, int dmin, int dmax)
{
    CensusPyramid py;
    py.L = census_levels(w, h, p.scales);
    py.w[0] = w; py.h[0] = h; py.dmin[0] = dmin; py.dmax[0] = dmax;
    for (int k = 1; k < py.L; k++) {
        py.w[k] = (py.w[k - 1] + 1) / 2; py.h[k] = (py.h[k - 1] + 1) / 2;
        py.dmin[k] = (int)std::floor(py.dmin[k - 1] / 2.0); py.dmax[k] = (int)std::ceil(py.dmax[k - 1] / 2.0);
    }
    return py;
}
size_t census_workspace_bytes(const s2p_census_params& p, int w, int h, int dmin, int dmax, bool want_S)
{
    const CensusPyramid py = census_pyramid(p, w, h, dmin, dmax);
    size_t level = 0, extra = 1024;
    for (int k = 0; k < py.L; k++) {
        level = std::max(level, census_level_bytes(py.w[k], py.h[k], census_D(p, py.dmin[k], py.dmax[k]), want_S && k == 0, p.nb_dir));
        if (k == 0) level = std::max(level, census_level_bytes(py.w[k], py.h[k], census_D(p, py.dmin[k], py.dmax[k], true), want_S, p.nb_dir));   // (a call with stage dumps)
        const size_t n = (size_t)py.w[k] * py.h[k];
        if (k > 0) extra += 3 * align_up(n * 4, 256);          // the two halved images and the level's disparity
        if (k + 1 < py.L) extra += 2 * align_up(n * 2, 256);   // lo, hi
    }
    return level + extra + 4096;
}

template <int G, int K>
static void launch_wta_census_pk(hipStream_t st, int rows, const CensusWtaArgs& a) {
    const size_t shm = (size_t)(a.sp * a.w + a.D) * 4 + (size_t)a.w * 6 + 16;
    const bool pad = G * 2 * K != a.D, quad = a.P2 <= 63;
    // rows wider than ~6000 px: more than the default 64 KiB of dynamic LDS (a CU has 160)
    #define S2P_WTA_LAUNCH(PADV, QUADV, CONFV, ...) do { if (shm > 64 * 1024) hipFuncSetAttribute((const void*)k_wta_census_pk<G, K, PADV, QUADV, CONFV, ##__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, S2P_ROW_LDS_MAX); \
        hipLaunchKernelGGL((k_wta_census_pk<G, K, PADV, QUADV, CONFV, ##__VA_ARGS__>), dim3(rows), dim3(S2P_WTA_NT), shm, st, a); } while (0)
    if (a.nd > 8) {                                       // 16 e-volumes: the consensus variants only (conf is never null here either)
        if (a.mindiff > 0) { if (pad) S2P_WTA_LAUNCH(true, false, true, true, 16); else S2P_WTA_LAUNCH(false, false, true, true, 16); }
        else               { if (pad) S2P_WTA_LAUNCH(true, false, true, false, 16); else S2P_WTA_LAUNCH(false, false, true, false, 16); }
    }
    else if (a.mindiff > 0) { if (pad) S2P_WTA_LAUNCH(true, false, true, true); else S2P_WTA_LAUNCH(false, false, true, true); }   // (the MINDIFF variant is built on the CONF one: conf is never null here)
    else if (a.conf && a.byte_keys) { if (pad) S2P_WTA_LAUNCH(true, true, true); else S2P_WTA_LAUNCH(false, true, true); }
    else if (a.conf) { if (pad) S2P_WTA_LAUNCH(true, false, true); else S2P_WTA_LAUNCH(false, false, true); }
    else if (pad) { if (quad) S2P_WTA_LAUNCH(true, true, false); else S2P_WTA_LAUNCH(true, false, false); }
    else          { if (quad) S2P_WTA_LAUNCH(false, true, false); else S2P_WTA_LAUNCH(false, false, false); }
    #undef S2P_WTA_LAUNCH
}

// one level: buffers carved from the current position of the bump workspace (the caller reserved and placed it)
// `stages`: which parts run (a batch of tiles runs CS_CARVE | CS_COST per tile, ONE aggregation for all of them, then CS_POST
// per tile); `pre`: the tile's buffers, filled by CS_CARVE and read by the later stages; Cfix / Efix: where the cost and
// e-volumes of the tile live when the caller laid them out (a batch keeps them at a constant stride for its one MGM launch).
enum { CS_CARVE = 1, CS_COST = 2, CS_AGG = 4, CS_POST = 8, CS_ALL = 15 };
static int census_level_enqueue(s2p_hip_ctx* ctx, const s2p_census_params& p, const float* d_im1, const float* d_im2,
                                int w, int h, int dmin, int dmax, const int16_t* d_lo, const int16_t* d_hi,
                                float* d_disp, float* d_conf, uint8_t* d_mask, bool want_S, CensusBuffers* out,
                                int stages = CS_ALL, CensusBuffers* pre = nullptr, uint8_t* Cfix = nullptr, uint8_t* Efix = nullptr,
                                const int* d_win = nullptr, int D_force = 0)
{
    hipStream_t st = ctx->stream;
    const int sp = p.subpix == 2 ? 2 : 1;
    // D_force: the volume's depth when it is more than this tile's own candidates need (a batch of tiles of different ranges shares one
    // lane layout): candidates Dt .. D_force - 1 are padding, exactly like the up-to-15 that rounding Dt up to 16 always adds
    const int Dt = sp * (dmax - dmin) + 1, D = D_force > 0 ? D_force : census_D(p, dmin, dmax, out != nullptr);
    const size_t npx = (size_t)w * h, vol = npx * D;
    CensusBuffers b;
    if (stages & CS_CARVE) {
    #define CARVE(field, type, bytes) b.field = (type)ws_alloc(ctx, (bytes)); if (!b.field) return S2P_HIP_RUNTIME_ERROR;
    CARVE(cen1, uint32_t*, npx * 4); CARVE(cen2, uint32_t*, npx * 4);
    if (Cfix) { b.C = Cfix; b.E = Efix; } else { CARVE(C, uint8_t*, vol); CARVE(E, uint8_t*, vol * census_planes(p.nb_dir)); }
    b.S = nullptr;
    if (want_S) { CARVE(S, uint16_t*, vol * 2); }
    CARVE(disp_raw, float*, npx * 4); CARVE(disp_med, float*, npx * 4);
    CARVE(q16, int16_t*, npx * 2);
    CARVE(lab, int*, npx * 4); CARVE(cnt, int*, npx * 4); CARVE(par, int*, npx * 4);
    #undef CARVE
    b.dmin0 = dmin; b.D0 = D;
    if (pre) *pre = b;
    } else b = *pre;
    if (out) *out = b;
    if (stages & CS_COST) {
        StageScope s(ctx, "cost");
        uint32_t* c1 = out ? b.cen1 : nullptr; uint32_t* c2 = out ? b.cen2 : nullptr;        // signatures only leave the kernel for dumps
        if (p.cost == 1) {                                   // ZNCC on the census window
            const size_t zl = zncc_cost_lds(w, p.census_win, sp);
            #define S2P_ZNCC_LAUNCH(WINV, SPV) do { if (zl > 64 * 1024) hipFuncSetAttribute((const void*)k_zncc_cost<WINV, SPV>, hipFuncAttributeMaxDynamicSharedMemorySize, S2P_ROW_LDS_MAX); \
                hipLaunchKernelGGL((k_zncc_cost<WINV, SPV>), dim3(h), dim3(256), zl, st, d_im1, d_im2, h, w, dmin, Dt, D, d_lo, d_hi, b.C); } while (0)
            if (p.census_win == 3) { if (sp == 2) S2P_ZNCC_LAUNCH(3, 2); else S2P_ZNCC_LAUNCH(3, 1); }
            else                   { if (sp == 2) S2P_ZNCC_LAUNCH(5, 2); else S2P_ZNCC_LAUNCH(5, 1); }
            #undef S2P_ZNCC_LAUNCH
        } else {
        const size_t lds = census_cost_lds(w, D, sp);
        #define S2P_COST_LAUNCH(WINV, SPV) do { if (lds > 64 * 1024) hipFuncSetAttribute((const void*)k_census_cost<WINV, SPV>, hipFuncAttributeMaxDynamicSharedMemorySize, S2P_ROW_LDS_MAX); \
            hipLaunchKernelGGL((k_census_cost<WINV, SPV>), dim3(h), dim3(256), lds, st, d_im1, d_im2, h, c1, c2, w, dmin, Dt, D, d_lo, d_hi, b.C); } while (0)
        if (p.census_win == 3) { if (sp == 2) S2P_COST_LAUNCH(3, 2); else S2P_COST_LAUNCH(3, 1); }
        else                   { if (sp == 2) S2P_COST_LAUNCH(5, 2); else S2P_COST_LAUNCH(5, 1); }
        #undef S2P_COST_LAUNCH
        }
    }
    if (stages & CS_AGG) {
        StageScope s(ctx, "aggregate");
        if (p.recursion >= 1) {
            char* mws = (char*)ws_alloc(ctx, mgm_workspace_bytes(w, h, D, p.nb_dir));
            if (!mws) return S2P_HIP_RUNTIME_ERROR;
            if (p.recursion == 2 || p.nb_dir > 8 || mgm_impl_bands()) {  // (the front-by-front cross-check kernel keeps two fronts: two predecessors, 8 directions only)
                if (!enqueue_mgm_bands(band_fork(ctx), b.C, b.E, w, h, D, p.P1, p.P2, mws, ctx->mgm_abort, mgm_nlat(p.nb_dir), 0, 1, 0, 0,
                                       p.recursion == 2 ? 3 : 2)) {
                    set_last_error("census: tile too large for the MGM hand-off ring"); return S2P_HIP_BAD_ARGUMENT;
                }
                band_join(ctx);
                ctx->mgm_check = true;
            } else {
                const size_t lmax = (size_t)std::max(w, h);
                enqueue_mgm(st, b.C, b.E, w, h, D, p.P1, p.P2, (uint16_t*)mws, (int*)(mws + align_up(16 * lmax * D * 2, 256)), p.nb_dir);
            }
        } else
            enqueue_aggregate<uint8_t>(st, b.C, b.E, w, h, D, p.P1, p.P2, p.P2, p.nb_dir);
    }
    if (stages & CS_POST) {
    if (want_S) hipLaunchKernelGGL(k_sum_S_u8, dim3((unsigned)((vol + 255) / 256)), dim3(256), 0, st, b.C, b.E, vol, p.P2, p.fix_overcount ? p.nb_dir - 1 : 0, p.nb_dir, b.S);
    {
        StageScope s(ctx, "wta");
        CensusWtaArgs wa;
        wa.C = b.C; wa.E = b.E; wa.vol = vol; wa.w = w; wa.h = h; wa.D = D; wa.Dt = Dt; wa.dmin = dmin; wa.P2 = p.P2;
        wa.lr_check = p.lr_check; wa.tau = (int)floorf(p.lr_tau * (float)sp); wa.sp = sp; wa.disp = b.disp_raw; wa.conf = d_conf;
        wa.mindiff = p.mindiff > 0 ? p.mindiff : 0;
        wa.byte_keys = (p.cost == 0 && p.P2 <= 63) ? 1 : 0;
        if (const char* e = getenv("S2P_WTA_BYTE_KEYS")) wa.byte_keys = wa.byte_keys && atoi(e);   // (A/B probe)
        if ((wa.mindiff > 0 || p.nb_dir > 8) && !wa.conf) wa.conf = (float*)b.lab;           // the MINDIFF kernel is the consensus one: a scratch plane takes what nobody asked for
        wa.fixo = p.fix_overcount ? p.nb_dir - 1 : 0; wa.nd = p.nb_dir; wa.sh = p.nb_dir == 16 ? 4 : p.nb_dir == 8 ? 3 : 2;
        wa.win = d_win;
        const LaneLayout ll = lane_layout(D);
        // 128 < D <= 256: 16 candidates per lane on one DPP row (padded below 256) instead of 8 per lane on 32 lanes -- twice the pixels per wave:
        // 0.43 -> 0.33 ms on 1024^2 x 144 ... 240, 0.129 -> 0.092 on 512^2 x 192 (round 6, profiles/r06/midrange_probe.txt)
        // D = 192: 12 per lane fill the row
        if (ll.K == 8) launch_wta_census_pk<64, 8>(st, h, wa);
        else if (D == 192) launch_wta_census_pk<16, 6>(st, h, wa);
        else if (D > 128 && D <= 256) launch_wta_census_pk<16, 8>(st, h, wa);
        else switch (ll.G) {
            case 2: launch_wta_census_pk<2, 4>(st, h, wa); break;
            case 4: launch_wta_census_pk<4, 4>(st, h, wa); break;
            case 8: launch_wta_census_pk<8, 4>(st, h, wa); break;
            case 16: launch_wta_census_pk<16, 4>(st, h, wa); break;
            case 32: launch_wta_census_pk<32, 4>(st, h, wa); break;
            default: launch_wta_census_pk<64, 4>(st, h, wa); break;
        }
    }
    // median (or not) of the raw map.  With stage dumps requested (`out`) the intermediate stays in the workspace
    // and is copied; otherwise the median kernel writes the caller's plane directly (no device-to-device copy).
    const bool want_epi = d_conf != nullptr || d_mask != nullptr;
    const bool fuse_epilogue = p.median && !out && p.remove_small_cc <= 0;   // the median is the last word on the disparity
    if (p.median) {
        StageScope s(ctx, "median");
        float* dst = out ? b.disp_med : d_disp;
        const dim3 grid((w + 255) / 256, h);
        if (fuse_epilogue) hipLaunchKernelGGL(k_median_valid<true>, grid, dim3(256), 0, st, b.disp_raw, dst, w, h, d_im1, d_im2, d_conf, d_mask);
        else hipLaunchKernelGGL(k_median_valid<false>, grid, dim3(256), 0, st, b.disp_raw, dst, w, h, nullptr, nullptr, nullptr, nullptr);
        if (out) hipMemcpyAsync(d_disp, b.disp_med, npx * 4, hipMemcpyDeviceToDevice, st);
    } else {
        if (out) hipMemcpyAsync(b.disp_med, b.disp_raw, npx * 4, hipMemcpyDeviceToDevice, st);   // keep the dump layout uniform
        hipMemcpyAsync(d_disp, b.disp_raw, npx * 4, hipMemcpyDeviceToDevice, st);
    }
    if (p.remove_small_cc > 0) {
        StageScope s(ctx, "speckle");
        const unsigned nb = (unsigned)((npx + 255) / 256);
        hipLaunchKernelGGL(k_f32_to_q16, dim3(nb), dim3(256), 0, st, d_disp, npx, b.q16);
        enqueue_speckle(st, b.q16, w, h, Q_INVALID, p.remove_small_cc - 1, 16, b.lab, b.par, b.cnt);
        hipLaunchKernelGGL(k_q16_apply, dim3(nb), dim3(256), 0, st, b.q16, npx, d_disp);
    }
    if (!fuse_epilogue && want_epi) {
        StageScope s(ctx, "epilogue");
        hipLaunchKernelGGL(k_census_epilogue, dim3((w + 255) / 256, h), dim3(256), 0, st, d_disp, d_im1, d_im2, w, h, d_conf, d_mask);
    }
    }   // CS_POST
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_last_error("kernel launch failed: %s", hipGetErrorString(e)); return S2P_HIP_RUNTIME_ERROR; }
    return S2P_HIP_OK;
}

int census_enqueue(s2p_hip_ctx* ctx, const s2p_census_params& p, const float* d_im1, const float* d_im2,
                   int w, int h, int dmin, int dmax, float* d_disp, float* d_conf, uint8_t* d_mask,
                   bool want_S, CensusBuffers* out)
{
    hipStream_t st = ctx->stream;
    int rc = ws_reserve(ctx, census_workspace_bytes(p, w, h, dmin, dmax, want_S));
    if (rc) return rc;
    ws_reset(ctx);
    StageScope total(ctx, "total");
    // ctx->mgm_abort: one word for every MGM launch of the context, raised by a hand-off that timed out, zero since the context was
    // created and cleared only by the host after it has seen it (api.hip: check_mgm) -- a later call never hides an earlier abort
    const CensusPyramid py = census_pyramid(p, w, h, dmin, dmax);
    if (py.L <= 1) return census_level_enqueue(ctx, p, d_im1, d_im2, w, h, dmin, dmax, nullptr, nullptr, d_disp, d_conf, d_mask, want_S, out);

    // multi-scale (mgm_multi's -S): halve the pair, match the coarsest level over the whole halved range, then let every
    // level restrict the candidates of the next finer one per pixel.  Level buffers share one region of the workspace.
    const float* a1[16]; const float* a2[16]; float* dl[16]; int16_t* lo[16]; int16_t* hi[16];
    a1[0] = d_im1; a2[0] = d_im2; dl[0] = d_disp;
    for (int k = 0; k < py.L; k++) {
        const size_t n = (size_t)py.w[k] * py.h[k];
        if (k > 0) {
            float* p1 = (float*)ws_alloc(ctx, n * 4); float* p2 = (float*)ws_alloc(ctx, n * 4); dl[k] = (float*)ws_alloc(ctx, n * 4);
            if (!p1 || !p2 || !dl[k]) return S2P_HIP_RUNTIME_ERROR;
            a1[k] = p1; a2[k] = p2;
            const dim3 grid((py.w[k] + 255) / 256, py.h[k]);
            hipLaunchKernelGGL(k_down2, grid, dim3(256), 0, st, a1[k - 1], py.w[k - 1], py.h[k - 1], p1);
            hipLaunchKernelGGL(k_down2, grid, dim3(256), 0, st, a2[k - 1], py.w[k - 1], py.h[k - 1], p2);
        }
        lo[k] = hi[k] = nullptr;
        if (k + 1 < py.L) {
            lo[k] = (int16_t*)ws_alloc(ctx, n * 2); hi[k] = (int16_t*)ws_alloc(ctx, n * 2);
            if (!lo[k] || !hi[k]) return S2P_HIP_RUNTIME_ERROR;
        }
    }
    int* d_mm = (int*)ws_alloc(ctx, 256);
    if (!d_mm) return S2P_HIP_RUNTIME_ERROR;
    const size_t mark = ctx->ws_used;
    CensusPyramid lv = py;                                   // ranges as narrowed on the way down
    for (int k = py.L - 1; k >= 0; k--) {
        ctx->ws_used = mark;
        if (k + 1 < py.L) {
            hipLaunchKernelGGL(k_range_from_coarse, dim3((py.w[k] + 255) / 256, py.h[k]), dim3(256), 0, st, dl[k + 1], py.w[k], py.h[k],
                               py.dmin[k], py.dmax[k], lo[k], hi[k]);
            // The level is matched over the UNION of the admissible ranges of the pixels that have a parent estimate, and the pixels
            // without one search that union too (a configured search range is usually several times what the parent level found:
            // cost and e-volumes shrink with it).  The union decides the lane layout and the
            // kernel instances, so the host needs it: 8 bytes back and one stream synchronisation per level -- a multi-scale call
            // is therefore not fully asynchronous (and is never captured into a hipGraph).
            const int init[2] = {0x7fffffff, -0x7fffffff - 1};
            int got[2];
            S2P_HIP_CHECK(hipMemcpyAsync(d_mm, init, 8, hipMemcpyHostToDevice, st));
            const dim3 grid((py.w[k] + 255) / 256, py.h[k]);
            hipLaunchKernelGGL(k_range_union, range_union_grid(py.h[k]), dim3(256), 0, st, dl[k + 1], lo[k], hi[k], py.w[k], py.h[k], d_mm);
            S2P_HIP_CHECK(hipMemcpyAsync(got, d_mm, 8, hipMemcpyDeviceToHost, st));
            S2P_HIP_CHECK(hipStreamSynchronize(st));
            if (got[0] <= got[1] && !getenv("S2P_MS_NO_UNION")) {       // (the switch is the A/B of tools/config2_time.py: timing only)
                hipLaunchKernelGGL(k_range_fill, grid, dim3(256), 0, st, dl[k + 1], py.w[k], py.h[k], d_mm, lo[k], hi[k]);
                lv.dmin[k] = got[0]; lv.dmax[k] = got[1];
            }
        }
        if (getenv("S2P_MS_DEBUG")) fprintf(stderr, "mgm_multi level %d: %d x %d, range [%d, %d] of [%d, %d], D %d\n", k, py.w[k], py.h[k], lv.dmin[k], lv.dmax[k], py.dmin[k], py.dmax[k], census_D(p, lv.dmin[k], lv.dmax[k]));
        s2p_census_params pk = p;
        if (k > 0 && pk.lr_check == 2) pk.lr_check = 0;      // mgm_leftright_control = 2: the L-R test at the last scale only
        rc = census_level_enqueue(ctx, pk, a1[k], a2[k], py.w[k], py.h[k], lv.dmin[k], lv.dmax[k], lo[k], hi[k], dl[k],
                                  k == 0 ? d_conf : nullptr, k == 0 ? d_mask : nullptr, want_S && k == 0, k == 0 ? out : nullptr);
        if (rc) return rc;
    }
    return S2P_HIP_OK;
}

// ---- a batch of equal-shape tiles in one call: cost volumes tile by tile, ONE aggregation launch over all of them (MGM modes:
// the lattices of every tile under one ready queue; a staggered start -- tile t + 1 let in when tile t is part way, so that fewer
// cost volumes are being re-read at a time -- was measured and loses: 0.574 ms per tile of an 8-tile launch with all tiles at
// once, 0.59-0.63 staggered), then WTA / median / epilogue tile by tile.  Multi-scale parameters and the 8-path mode run the
// tiles one after the other (same results either way: tiles share nothing).
// Multi-scale tiles (mgm_multi's -S) batch as well (round 4): the levels run level by level for ALL tiles -- per level one
// read-back of the n unions (one synchronisation per level and batch instead of one per level and tile), one volume shape for the
// batch (the hull of the tiles' ranges; every tile keeps its own range through the per-pixel [lo, hi] planes and the WTA's window)
// and ONE aggregation launch.  A candidate that the hull adds to a tile is excluded (cost 255) and, for P2 <= 115, can neither win
// a minimum nor lower a pixel's min L (255 >= 24 + 2 P2: an excluded candidate's L is at least every valid pixel's min L + P2), so
// the tile's own candidates aggregate to the same bytes as in a volume of the tile's own range: tests/test_gpu_batch.py.
static bool census_batches(const s2p_census_params& p, int n, int w, int h) {
    if (n <= 1 || p.recursion < 1) return false;
    return census_levels(w, h, p.scales) == 1 || p.P2 <= 115;
}
static size_t mgm_bands_workspace_upto(int w, int h, int D, int n, int nd) {     // the lane layout (rows per band) changes with D: the largest need up to D
    size_t m = 0;
    for (int d = 16; d <= D; d += 16) m = std::max(m, mgm_bands_workspace_bytes(w, h, d, n, mgm_nlat(nd)));
    return m;
}
size_t census_batch_workspace_bytes(const s2p_census_params& p, int n, int w, int h, int dmin, int dmax)
{
    const size_t single = census_workspace_bytes(p, w, h, dmin, dmax, false);
    if (!census_batches(p, n, w, h)) return single;
    const CensusPyramid py = census_pyramid(p, w, h, dmin, dmax);
    if (py.L <= 1) {
        const int D = census_D(p, dmin, dmax);
        return std::max(single, (size_t)n * census_level_bytes(w, h, D, false, p.nb_dir) + mgm_bands_workspace_bytes(w, h, D, n, mgm_nlat(p.nb_dir)) + 8192);
    }
    size_t level = 0, extra = 4096 + align_up((size_t)n * 8, 256);
    for (int k = 0; k < py.L; k++) {
        const int D = census_D(p, py.dmin[k], py.dmax[k]);
        level = std::max(level, (size_t)n * census_level_bytes(py.w[k], py.h[k], D, false, p.nb_dir) + mgm_bands_workspace_upto(py.w[k], py.h[k], D, n, p.nb_dir) + 4096);
        const size_t npx = (size_t)py.w[k] * py.h[k];
        if (k > 0) extra += (size_t)n * 3 * align_up(npx * 4, 256);
        if (k + 1 < py.L) extra += (size_t)n * 2 * align_up(npx * 2, 256);
    }
    return std::max(single, level + extra + 8192);
}

static int census_batch_multiscale_enqueue(s2p_hip_ctx* ctx, const s2p_census_params& p, int n, const float* const* d_im1, const float* const* d_im2,
                                           int w, int h, int dmin, int dmax, float* const* d_disp, float* const* d_conf, uint8_t* const* d_mask)
{
    hipStream_t st = ctx->stream;
    StageScope total(ctx, "total");
    const CensusPyramid py = census_pyramid(p, w, h, dmin, dmax);
    struct Lv { const float* a1[16]; const float* a2[16]; float* dl[16]; int16_t* lo[16]; int16_t* hi[16]; };
    std::vector<Lv> T(n);
    for (int t = 0; t < n; t++) {
        Lv& L = T[t];
        L.a1[0] = d_im1[t]; L.a2[0] = d_im2[t]; L.dl[0] = d_disp[t];
        for (int k = 0; k < py.L; k++) {
            const size_t npx = (size_t)py.w[k] * py.h[k];
            if (k > 0) {
                float* p1 = (float*)ws_alloc(ctx, npx * 4); float* p2 = (float*)ws_alloc(ctx, npx * 4); L.dl[k] = (float*)ws_alloc(ctx, npx * 4);
                if (!p1 || !p2 || !L.dl[k]) return S2P_HIP_RUNTIME_ERROR;
                L.a1[k] = p1; L.a2[k] = p2;
                const dim3 grid((py.w[k] + 255) / 256, py.h[k]);
                hipLaunchKernelGGL(k_down2, grid, dim3(256), 0, st, L.a1[k - 1], py.w[k - 1], py.h[k - 1], p1);
                hipLaunchKernelGGL(k_down2, grid, dim3(256), 0, st, L.a2[k - 1], py.w[k - 1], py.h[k - 1], p2);
            }
            L.lo[k] = L.hi[k] = nullptr;
            if (k + 1 < py.L) {
                L.lo[k] = (int16_t*)ws_alloc(ctx, npx * 2); L.hi[k] = (int16_t*)ws_alloc(ctx, npx * 2);
                if (!L.lo[k] || !L.hi[k]) return S2P_HIP_RUNTIME_ERROR;
            }
        }
    }
    int* d_mm = (int*)ws_alloc(ctx, (size_t)n * 8);
    if (!d_mm) return S2P_HIP_RUNTIME_ERROR;
    const size_t mark = ctx->ws_used;
    std::vector<int> init(2 * n), got(2 * n);
    for (int t = 0; t < n; t++) { init[2 * t] = 0x7fffffff; init[2 * t + 1] = -0x7fffffff - 1; }
    for (int k = py.L - 1; k >= 0; k--) {
        ctx->ws_used = mark;
        const int wk = py.w[k], hk = py.h[k];
        const dim3 grid((wk + 255) / 256, hk);
        int c0 = py.dmin[k], c1 = py.dmax[k];                    // the volume's range at this level: the hull of the tiles' ranges
        const bool narrowed = k + 1 < py.L;
        if (narrowed) {
            S2P_HIP_CHECK(hipMemcpyAsync(d_mm, init.data(), (size_t)n * 8, hipMemcpyHostToDevice, st));
            for (int t = 0; t < n; t++) {
                hipLaunchKernelGGL(k_range_from_coarse, grid, dim3(256), 0, st, T[t].dl[k + 1], wk, hk, py.dmin[k], py.dmax[k], T[t].lo[k], T[t].hi[k]);
                hipLaunchKernelGGL(k_range_union, range_union_grid(hk), dim3(256), 0, st, T[t].dl[k + 1], T[t].lo[k], T[t].hi[k], wk, hk, d_mm + 2 * t);
            }
            S2P_HIP_CHECK(hipMemcpyAsync(got.data(), d_mm, (size_t)n * 8, hipMemcpyDeviceToHost, st));
            S2P_HIP_CHECK(hipStreamSynchronize(st));             // ONE read-back per level for the whole batch
            int lo = 0x7fffffff, hi = -0x7fffffff - 1;
            for (int t = 0; t < n; t++) {
                if (got[2 * t] <= got[2 * t + 1]) {
                    hipLaunchKernelGGL(k_range_fill, grid, dim3(256), 0, st, T[t].dl[k + 1], wk, hk, d_mm + 2 * t, T[t].lo[k], T[t].hi[k]);
                    lo = std::min(lo, got[2 * t]); hi = std::max(hi, got[2 * t + 1]);
                } else { lo = std::min(lo, py.dmin[k]); hi = std::max(hi, py.dmax[k]); }   // no pixel of this tile has a parent: the configured range
            }
            c0 = lo; c1 = hi;
        }
        if (getenv("S2P_MS_DEBUG")) fprintf(stderr, "mgm_multi batch of %d, level %d: %d x %d, hull [%d, %d] of [%d, %d], D %d\n", n, k, wk, hk, c0, c1, py.dmin[k], py.dmax[k], census_D(p, c0, c1));
        const int D = census_D(p, c0, c1);
        const size_t vol = (size_t)wk * hk * D;
        uint8_t* Call = (uint8_t*)ws_alloc(ctx, (size_t)n * vol);
        uint8_t* Eall = (uint8_t*)ws_alloc(ctx, (size_t)n * vol * census_planes(p.nb_dir));
        if (!Call || !Eall) return S2P_HIP_RUNTIME_ERROR;
        s2p_census_params pk = p;
        if (k > 0 && pk.lr_check == 2) pk.lr_check = 0;      // mgm_leftright_control = 2: the L-R test at the last scale only
        std::vector<CensusBuffers> bufs(n);
        int rc;
        for (int t = 0; t < n; t++) {
            rc = census_level_enqueue(ctx, pk, T[t].a1[k], T[t].a2[k], wk, hk, c0, c1, T[t].lo[k], T[t].hi[k], T[t].dl[k], nullptr, nullptr, false, nullptr,
                                      CS_CARVE | CS_COST, &bufs[t], Call + (size_t)t * vol, Eall + (size_t)t * vol * census_planes(p.nb_dir));
            if (rc) return rc;
        }
        {
            StageScope s(ctx, "aggregate");
            char* mws = (char*)ws_alloc(ctx, mgm_bands_workspace_bytes(wk, hk, D, n, mgm_nlat(p.nb_dir)));
            if (!mws) return S2P_HIP_RUNTIME_ERROR;
            if (!enqueue_mgm_bands(band_fork(ctx), Call, Eall, wk, hk, D, p.P1, p.P2, mws, ctx->mgm_abort, mgm_nlat(p.nb_dir), 0, n, vol, vol * census_planes(p.nb_dir),
                                   p.recursion == 2 ? 3 : 2, S2P_MGM_BATCH_STAGGER)) {
                set_last_error("census: tile too large for the MGM hand-off ring"); return S2P_HIP_BAD_ARGUMENT;
            }
            band_join(ctx);
            ctx->mgm_check = true;
        }
        for (int t = 0; t < n; t++) {
            rc = census_level_enqueue(ctx, pk, T[t].a1[k], T[t].a2[k], wk, hk, c0, c1, T[t].lo[k], T[t].hi[k], T[t].dl[k],
                                      k == 0 && d_conf ? d_conf[t] : nullptr, k == 0 && d_mask ? d_mask[t] : nullptr, false, nullptr,
                                      CS_POST, &bufs[t], nullptr, nullptr, narrowed ? d_mm + 2 * t : nullptr);
            if (rc) return rc;
        }
    }
    return S2P_HIP_OK;
}

int census_batch_enqueue(s2p_hip_ctx* ctx, const s2p_census_params& p, int n, const float* const* d_im1, const float* const* d_im2,
                         int w, int h, int dmin, int dmax, float* const* d_disp, float* const* d_conf, uint8_t* const* d_mask)
{
    if (!census_batches(p, n, w, h)) {
        for (int t = 0; t < n; t++) {
            int rc = census_enqueue(ctx, p, d_im1[t], d_im2[t], w, h, dmin, dmax, d_disp[t], d_conf ? d_conf[t] : nullptr,
                                    d_mask ? d_mask[t] : nullptr, false, nullptr);
            if (rc) return rc;
        }
        return S2P_HIP_OK;
    }
    if (census_levels(w, h, p.scales) > 1) {
        int rc = ws_reserve(ctx, census_batch_workspace_bytes(p, n, w, h, dmin, dmax));
        if (rc) return rc;
        ws_reset(ctx);
        return census_batch_multiscale_enqueue(ctx, p, n, d_im1, d_im2, w, h, dmin, dmax, d_disp, d_conf, d_mask);
    }
    hipStream_t st = ctx->stream;
    int rc = ws_reserve(ctx, census_batch_workspace_bytes(p, n, w, h, dmin, dmax));
    if (rc) return rc;
    ws_reset(ctx);
    StageScope total(ctx, "total");
    const int D = census_D(p, dmin, dmax);
    const size_t vol = (size_t)w * h * D;
    uint8_t* Call = (uint8_t*)ws_alloc(ctx, (size_t)n * vol);
    uint8_t* Eall = (uint8_t*)ws_alloc(ctx, (size_t)n * vol * census_planes(p.nb_dir));
    if (!Call || !Eall) return S2P_HIP_RUNTIME_ERROR;
    std::vector<CensusBuffers> bufs(n);
    for (int t = 0; t < n; t++) {
        rc = census_level_enqueue(ctx, p, d_im1[t], d_im2[t], w, h, dmin, dmax, nullptr, nullptr, d_disp[t], d_conf ? d_conf[t] : nullptr,
                                  d_mask ? d_mask[t] : nullptr, false, nullptr, CS_CARVE | CS_COST, &bufs[t], Call + (size_t)t * vol, Eall + (size_t)t * vol * census_planes(p.nb_dir));
        if (rc) return rc;
    }
    {
        StageScope s(ctx, "aggregate");
        char* mws = (char*)ws_alloc(ctx, mgm_bands_workspace_bytes(w, h, D, n, mgm_nlat(p.nb_dir)));
        if (!mws) return S2P_HIP_RUNTIME_ERROR;
        if (!enqueue_mgm_bands(band_fork(ctx), Call, Eall, w, h, D, p.P1, p.P2, mws, ctx->mgm_abort, mgm_nlat(p.nb_dir), 0, n, vol, vol * census_planes(p.nb_dir),
                               p.recursion == 2 ? 3 : 2, S2P_MGM_BATCH_STAGGER)) {
            set_last_error("census: tile too large for the MGM hand-off ring"); return S2P_HIP_BAD_ARGUMENT;
        }
        band_join(ctx);
        ctx->mgm_check = true;
    }
    for (int t = 0; t < n; t++) {
        rc = census_level_enqueue(ctx, p, d_im1[t], d_im2[t], w, h, dmin, dmax, nullptr, nullptr, d_disp[t], d_conf ? d_conf[t] : nullptr,
                                  d_mask ? d_mask[t] : nullptr, false, nullptr, CS_POST, &bufs[t]);
        if (rc) return rc;
    }
    return S2P_HIP_OK;
}

// ---- a batch of tiles of DIFFERENT sizes and ranges (what the tiles of a real job look like; round 4): one depth D for all of them
// (the largest tile's, the others' volumes are padded with excluded candidates), cost volumes tile by tile, ONE aggregation launch with
// per-tile geometry (enqueue_mgm_bands_hetero), WTA ... epilogue tile by tile with each tile's own candidate count.  Single-scale MGM
// modes with P2 <= 115 (the padding argument of census_batches); anything else runs tile by tile.
// number of pyramid levels the tiles of a mixed batch share, 0 if they do not
static int census_hetero_levels(const s2p_census_params& p, int n, const int* w, const int* h) {
    const int L = census_levels(w[0], h[0], p.scales);
    for (int t = 1; t < n; t++) if (census_levels(w[t], h[t], p.scales) != L) return 0;
    return L;
}
bool census_batches_hetero(const s2p_census_params& p, int n, const int* w, const int* h) {
    if (n <= 1 || n > S2P_MGM_HETERO_MAX || p.recursion < 1 || p.P2 > 115) return false;
    return census_hetero_levels(p, n, w, h) >= 1;             // single scale, or the same number of levels for every tile
}
static size_t mgm_bands_hetero_workspace_upto(int n, const int* w, const int* h, int D, int nd) {    // the lane layout changes with D: the largest need up to D
    size_t m = 0;
    for (int d = 16; d <= D; d += 16) m = std::max(m, mgm_bands_hetero_workspace_bytes(n, w, h, d, mgm_nlat(nd)));
    return m;
}
size_t census_batch_hetero_workspace_bytes(const s2p_census_params& p, int n, const int* w, const int* h, const int* dmin, const int* dmax)
{
    size_t need = 0;
    if (!census_batches_hetero(p, n, w, h)) {
        for (int t = 0; t < n; t++) need = std::max(need, census_workspace_bytes(p, w[t], h[t], dmin[t], dmax[t], false));
        return need;
    }
    const int L = census_hetero_levels(p, n, w, h);
    std::vector<CensusPyramid> py(n);
    for (int t = 0; t < n; t++) py[t] = census_pyramid(p, w[t], h[t], dmin[t], dmax[t]);
    size_t level = 0, extra = 4096 + align_up((size_t)n * 8, 256);
    std::vector<int> wk(n), hk(n);
    for (int k = 0; k < L; k++) {
        int D = 0;
        size_t lv = 0;
        for (int t = 0; t < n; t++) D = std::max(D, census_D(p, py[t].dmin[k], py[t].dmax[k]));
        for (int t = 0; t < n; t++) {
            wk[t] = py[t].w[k]; hk[t] = py[t].h[k];
            lv += census_level_bytes(wk[t], hk[t], D, false, p.nb_dir) + 1024;
            const size_t npx = (size_t)wk[t] * hk[t];
            if (k > 0) extra += 3 * align_up(npx * 4, 256);
            if (k + 1 < L) extra += 2 * align_up(npx * 2, 256);
        }
        level = std::max(level, lv + mgm_bands_hetero_workspace_upto(n, wk.data(), hk.data(), D, p.nb_dir) + 4096);
    }
    return level + extra + 8192;
}
// one level of n tiles of different sizes and ranges: cost volumes of ONE depth (the widest range's), one aggregation launch with
// per-tile geometry, WTA ... epilogue per tile.  The caller has reserved the workspace and placed the bump pointer.
static int census_hetero_level(s2p_hip_ctx* ctx, const s2p_census_params& p, int n, const float* const* d_im1, const float* const* d_im2,
                               const int* w, const int* h, const int* lo, const int* hi, int16_t* const* d_lo, int16_t* const* d_hi,
                               float* const* d_disp, float* const* d_conf, uint8_t* const* d_mask)
{
    hipStream_t st = ctx->stream;
    int D = 0, rc;
    for (int t = 0; t < n; t++) D = std::max(D, census_D(p, lo[t], hi[t]));
    std::vector<size_t> c_off(n), e_off(n);
    size_t csum = 0, esum = 0;
    for (int t = 0; t < n; t++) {
        const size_t vol = (size_t)w[t] * h[t] * D;
        c_off[t] = csum; e_off[t] = esum;
        csum += align_up(vol, 256); esum += align_up(vol * census_planes(p.nb_dir), 256);
    }
    uint8_t* Call = (uint8_t*)ws_alloc(ctx, csum);
    uint8_t* Eall = (uint8_t*)ws_alloc(ctx, esum);
    if (!Call || !Eall) return S2P_HIP_RUNTIME_ERROR;
    std::vector<CensusBuffers> bufs(n);
    for (int t = 0; t < n; t++) {
        rc = census_level_enqueue(ctx, p, d_im1[t], d_im2[t], w[t], h[t], lo[t], hi[t], d_lo ? d_lo[t] : nullptr, d_hi ? d_hi[t] : nullptr, d_disp[t],
                                  d_conf ? d_conf[t] : nullptr, d_mask ? d_mask[t] : nullptr, false, nullptr, CS_CARVE | CS_COST, &bufs[t],
                                  Call + c_off[t], Eall + e_off[t], nullptr, D);
        if (rc) return rc;
    }
    {
        StageScope s(ctx, "aggregate");
        char* mws = (char*)ws_alloc(ctx, mgm_bands_hetero_workspace_bytes(n, w, h, D, mgm_nlat(p.nb_dir)));
        if (!mws) return S2P_HIP_RUNTIME_ERROR;
        if (!enqueue_mgm_bands_hetero(band_fork(ctx), Call, Eall, n, w, h, D, p.P1, p.P2, c_off.data(), e_off.data(), mws, ctx->mgm_abort,
                                      mgm_nlat(p.nb_dir), p.recursion == 2 ? 3 : 2)) {
            set_last_error("census: tile too large for the MGM hand-off ring"); return S2P_HIP_BAD_ARGUMENT;
        }
        band_join(ctx);
        ctx->mgm_check = true;
    }
    for (int t = 0; t < n; t++) {
        rc = census_level_enqueue(ctx, p, d_im1[t], d_im2[t], w[t], h[t], lo[t], hi[t], d_lo ? d_lo[t] : nullptr, d_hi ? d_hi[t] : nullptr, d_disp[t],
                                  d_conf ? d_conf[t] : nullptr, d_mask ? d_mask[t] : nullptr, false, nullptr, CS_POST, &bufs[t], nullptr, nullptr, nullptr, D);
        if (rc) return rc;
    }
    return S2P_HIP_OK;
}
int census_batch_hetero_enqueue(s2p_hip_ctx* ctx, const s2p_census_params& p, int n, const float* const* d_im1, const float* const* d_im2,
                                const int* w, const int* h, const int* dmin, const int* dmax,
                                float* const* d_disp, float* const* d_conf, uint8_t* const* d_mask)
{
    if (!census_batches_hetero(p, n, w, h)) {
        for (int t = 0; t < n; t++) {
            int rc = census_enqueue(ctx, p, d_im1[t], d_im2[t], w[t], h[t], dmin[t], dmax[t], d_disp[t], d_conf ? d_conf[t] : nullptr,
                                    d_mask ? d_mask[t] : nullptr, false, nullptr);
            if (rc) return rc;
        }
        return S2P_HIP_OK;
    }
    hipStream_t st = ctx->stream;
    int rc = ws_reserve(ctx, census_batch_hetero_workspace_bytes(p, n, w, h, dmin, dmax));
    if (rc) return rc;
    ws_reset(ctx);
    StageScope total(ctx, "total");
    const int L = census_hetero_levels(p, n, w, h);
    if (L <= 1) return census_hetero_level(ctx, p, n, d_im1, d_im2, w, h, dmin, dmax, nullptr, nullptr, d_disp, d_conf, d_mask);

    // multi-scale tiles of different sizes (the same number of levels each): level by level for all tiles, one read-back of the n
    // unions per level; every tile's volume starts at its OWN range and has the batch's depth (padding at the top only)
    std::vector<CensusPyramid> py(n);
    struct Lv { const float* a1[16]; const float* a2[16]; float* dl[16]; int16_t* lo[16]; int16_t* hi[16]; };
    std::vector<Lv> T(n);
    for (int t = 0; t < n; t++) {
        py[t] = census_pyramid(p, w[t], h[t], dmin[t], dmax[t]);
        Lv& V = T[t];
        V.a1[0] = d_im1[t]; V.a2[0] = d_im2[t]; V.dl[0] = d_disp[t];
        for (int k = 0; k < L; k++) {
            const size_t npx = (size_t)py[t].w[k] * py[t].h[k];
            if (k > 0) {
                float* p1 = (float*)ws_alloc(ctx, npx * 4); float* p2 = (float*)ws_alloc(ctx, npx * 4); V.dl[k] = (float*)ws_alloc(ctx, npx * 4);
                if (!p1 || !p2 || !V.dl[k]) return S2P_HIP_RUNTIME_ERROR;
                V.a1[k] = p1; V.a2[k] = p2;
                const dim3 grid((py[t].w[k] + 255) / 256, py[t].h[k]);
                hipLaunchKernelGGL(k_down2, grid, dim3(256), 0, st, V.a1[k - 1], py[t].w[k - 1], py[t].h[k - 1], p1);
                hipLaunchKernelGGL(k_down2, grid, dim3(256), 0, st, V.a2[k - 1], py[t].w[k - 1], py[t].h[k - 1], p2);
            }
            V.lo[k] = V.hi[k] = nullptr;
            if (k + 1 < L) {
                V.lo[k] = (int16_t*)ws_alloc(ctx, npx * 2); V.hi[k] = (int16_t*)ws_alloc(ctx, npx * 2);
                if (!V.lo[k] || !V.hi[k]) return S2P_HIP_RUNTIME_ERROR;
            }
        }
    }
    int* d_mm = (int*)ws_alloc(ctx, (size_t)n * 8);
    if (!d_mm) return S2P_HIP_RUNTIME_ERROR;
    const size_t mark = ctx->ws_used;
    std::vector<int> init(2 * n), got(2 * n), wk(n), hk(n), lo(n), hi(n);
    std::vector<const float*> a1(n), a2(n);
    std::vector<float*> dl(n);
    std::vector<int16_t*> plo(n), phi(n);
    for (int t = 0; t < n; t++) { init[2 * t] = 0x7fffffff; init[2 * t + 1] = -0x7fffffff - 1; }
    for (int k = L - 1; k >= 0; k--) {
        ctx->ws_used = mark;
        const bool narrowed = k + 1 < L;
        for (int t = 0; t < n; t++) { wk[t] = py[t].w[k]; hk[t] = py[t].h[k]; lo[t] = py[t].dmin[k]; hi[t] = py[t].dmax[k]; }
        if (narrowed) {
            S2P_HIP_CHECK(hipMemcpyAsync(d_mm, init.data(), (size_t)n * 8, hipMemcpyHostToDevice, st));
            for (int t = 0; t < n; t++) {
                const dim3 grid((wk[t] + 255) / 256, hk[t]);
                hipLaunchKernelGGL(k_range_from_coarse, grid, dim3(256), 0, st, T[t].dl[k + 1], wk[t], hk[t], lo[t], hi[t], T[t].lo[k], T[t].hi[k]);
                hipLaunchKernelGGL(k_range_union, range_union_grid(hk[t]), dim3(256), 0, st, T[t].dl[k + 1], T[t].lo[k], T[t].hi[k], wk[t], hk[t], d_mm + 2 * t);
            }
            S2P_HIP_CHECK(hipMemcpyAsync(got.data(), d_mm, (size_t)n * 8, hipMemcpyDeviceToHost, st));
            S2P_HIP_CHECK(hipStreamSynchronize(st));
            for (int t = 0; t < n; t++)
                if (got[2 * t] <= got[2 * t + 1]) {
                    const dim3 grid((wk[t] + 255) / 256, hk[t]);
                    hipLaunchKernelGGL(k_range_fill, grid, dim3(256), 0, st, T[t].dl[k + 1], wk[t], hk[t], d_mm + 2 * t, T[t].lo[k], T[t].hi[k]);
                    lo[t] = got[2 * t]; hi[t] = got[2 * t + 1];
                }
        }
        s2p_census_params pk = p;
        if (k > 0 && pk.lr_check == 2) pk.lr_check = 0;      // mgm_leftright_control = 2: the L-R test at the last scale only
        for (int t = 0; t < n; t++) { a1[t] = T[t].a1[k]; a2[t] = T[t].a2[k]; dl[t] = T[t].dl[k]; plo[t] = T[t].lo[k]; phi[t] = T[t].hi[k]; }
        rc = census_hetero_level(ctx, pk, n, a1.data(), a2.data(), wk.data(), hk.data(), lo.data(), hi.data(), narrowed ? plo.data() : nullptr,
                                 narrowed ? phi.data() : nullptr, dl.data(), k == 0 ? d_conf : nullptr, k == 0 ? d_mask : nullptr);
        if (rc) return rc;
    }
    return S2P_HIP_OK;
}

// ---- masking.erosion (s2p/masking.py:87-97: `morsi disk%d erosion`): minimum over the offsets with
// hypot(i, j) < radius, offsets outside the image ignored (oracle: s2p_oracle_erode_disk; unpinned). ----
__global__ __launch_bounds__(256) void k_erode_disk(const uint8_t* __restrict__ msk, int w, int h, int radius, uint8_t* __restrict__ out)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    int v = msk[(size_t)y * w + x];
    const int R = radius + 1, r2 = radius * radius;
    for (int j = -R; j <= R; j++) {
        const int yy = y + j;
        if (yy < 0 || yy >= h) continue;
        for (int i = -R; i <= R; i++) {
            const int xx = x + i;
            if (i * i + j * j >= r2 || xx < 0 || xx >= w) continue;     // hypot(i, j) < radius on integers
            v = msk[(size_t)yy * w + xx] ? v : 0;
        }
    }
    out[(size_t)y * w + x] = (uint8_t)v;
}

int erode_enqueue(s2p_hip_ctx* ctx, const uint8_t* d_msk, int w, int h, int radius, uint8_t* d_out)
{
    hipLaunchKernelGGL(k_erode_disk, dim3((w + 255) / 256, h), dim3(256), 0, ctx->stream, d_msk, w, h, radius, d_out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_last_error("kernel launch failed: %s", hipGetErrorString(e)); return S2P_HIP_RUNTIME_ERROR; }
    return S2P_HIP_OK;
}

// standalone rejection mask on device buffers (file-level create_rejection_mask)
int rejection_mask_enqueue(s2p_hip_ctx* ctx, const float* d_disp, const float* d_im1, const float* d_im2, int w, int h, uint8_t* d_mask)
{
    hipLaunchKernelGGL(k_census_epilogue, dim3((w + 255) / 256, h), dim3(256), 0, ctx->stream, d_disp, d_im1, d_im2, w, h,
                       (float*)nullptr, d_mask);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_last_error("kernel launch failed: %s", hipGetErrorString(e)); return S2P_HIP_RUNTIME_ERROR; }
    return S2P_HIP_OK;
}

}  // namespace s2p
