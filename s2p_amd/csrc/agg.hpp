// s2p_amd/csrc/agg.hpp -- the hot kernel: 8-path semi-global aggregation as wavefront recurrences,
// generic over the cost element type (int16 for the sgbm matcher, uint8 for the census matcher).
// Reference arithmetic: 3rdparty/sgbm/stereosgbm.cpp:518-662 (L_r update, formula 13 of the paper).
#pragma once
#include "common.hpp"

namespace s2p {

struct AggArgs {
    const void* C;              // int16 (sgbm: the reference's +P2 bias already inside) or uint8 (census: raw Hamming)
    uint8_t* E;                 // 8 volumes, each vol elements: e = (C + bias) - L_r in [0, P2]
    size_t vol;                 // h * width1 * D
    int width1, h, D, P1, P2;
    int bias;                   // 0 for sgbm, P2 for census (see the kernel comment)
    int block_start[9];         // first block of direction r (prefix sums); blocks never mix directions
    int npaths[8];
};

#define BIGPK 0x3fff3fffu        // "MAX_COST" stand-in for Lr[-1], Lr[D]: any value that loses every min
#define S2P_BUF_FLAGS 0x00020000 // gfx9-family raw buffer descriptor word 3 (DATA_FORMAT = 32 bit)
#ifndef S2P_AGG_PF
#define S2P_AGG_PF 8
#endif
#define S2P_OOB 0xffffffffu      // buffer offset that is always out of range: loads return 0, stores are dropped

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// 8 costs of one lane -> 4 packed int16 pairs, through a bounds-checked raw buffer load (no branches)
template <typename CT> struct CostLoad;
template <> struct CostLoad<int16_t> {
    typedef u32x4 raw_t;
    static __device__ __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t r, uint32_t off) { return __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0); }
    static __device__ __forceinline__ void unpack(raw_t v, uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d) { a = v.x; b = v.y; c = v.z; d = v.w; }
};
template <> struct CostLoad<uint8_t> {
    typedef u32x2 raw_t;
    static __device__ __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t r, uint32_t off) { return __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0); }
    static __device__ __forceinline__ void unpack(raw_t v, uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d) {
        a = __builtin_amdgcn_perm(0u, v.x, 0x0c010c00u); b = __builtin_amdgcn_perm(0u, v.x, 0x0c030c02u);
        c = __builtin_amdgcn_perm(0u, v.y, 0x0c010c00u); d = __builtin_amdgcn_perm(0u, v.y, 0x0c030c02u);
    }
};

// One launch = all 8 directions.  A path (1-D recurrence along one direction) is owned by a group of
// G lanes, 8 disparities per lane as 4 packed int16 pairs; 64/G paths per wavefront advance in lock
// step.  The d+-1 neighbours come from DPP shifts inside the group, min_k L from a DPP xor butterfly.
// State kept in registers is L' = L - bias (bias = 0 when the stored costs already carry the
// reference's +P2, = P2 for raw census costs): the recurrence is invariant under that shift, so no
// per-step bias add is needed.  Memory goes through raw buffer descriptors: the C prefetch (PF steps
// ahead, statically named registers so that the compiler emits counted vmcnt waits) needs no
// predicate at all, masked lanes store to an out-of-range offset.  No branch inside a step.
//
// Diagonals are WRAPPED: lane group s walks pixel ((s + t*dx) mod width1, t) for every row t, i.e.
// the concatenation of the two image diagonals that share a start column modulo width1, with a
// state reset where a new diagonal enters at the image border.  Every path of every direction then
// has the same length (width1 steps for the 2 horizontal directions, h for the 6 others), there are
// exactly 2h + 6*width1 paths, adjacent groups touch adjacent pixels of one row at every step, and no
// lane ever idles on a path that has not started or already ended.
template <int G, bool PAD, typename CT, int PF, bool DIAG>
__device__ __forceinline__ void aggregate_paths(const AggArgs& a, const int r)
{
    typedef CostLoad<CT> CL;
    typedef typename CL::raw_t raw_t;
    constexpr int NP = 64 / G;                 // paths per wavefront
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane & (G - 1);              // lane inside its path group
    const int width1 = a.width1, h = a.h, D = a.D;
    const int path = ((int)blockIdx.x - a.block_start[r]) * (4 * NP) + wave * NP + lane / G;
    const bool path_ok = path < a.npaths[r];
    const bool lane_ok = PAD ? (g * 8 < D) : true;
    if (!__any(path_ok)) return;

    // path geometry: pixel(t) = (xs + t*dx [mod width1 on diagonals], ys + t*dy), t in [0, T)
    int xs, ys, dx, dy, T;
    switch (r) {
        case 0: xs = 0; ys = path; dx = 1; dy = 0; T = width1; break;
        case 1: xs = width1 - 1; ys = path; dx = -1; dy = 0; T = width1; break;
        case 2: xs = path; ys = 0; dx = 0; dy = 1; T = h; break;
        case 3: xs = path; ys = h - 1; dx = 0; dy = -1; T = h; break;
        case 4: xs = path; ys = 0; dx = 1; dy = 1; T = h; break;
        case 5: xs = path; ys = 0; dx = -1; dy = 1; T = h; break;
        case 6: xs = path; ys = h - 1; dx = -1; dy = -1; T = h; break;
        default: xs = path; ys = h - 1; dx = 1; dy = -1; T = h; break;
    }
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.C), 0, (int)(a.vol * sizeof(CT)), S2P_BUF_FLAGS);
    const __amdgpu_buffer_rsrc_t rsE = __builtin_amdgcn_make_buffer_rsrc(a.E + (size_t)r * a.vol, 0, (int)a.vol, S2P_BUF_FLAGS);
    const int stride = (dy * width1 + dx) * D;                            // elements per step (may be negative)
    const uint32_t base = (uint32_t)((ys * width1 + xs) * D + g * 8);    // element offset at t = 0
    const bool lane_live = path_ok && lane_ok;
    uint32_t offE = lane_live ? base : S2P_OOB;                           // byte offset into E_r at step t
    const uint32_t stepE = lane_live ? (uint32_t)stride : 0u;
    uint32_t offC = base * (uint32_t)sizeof(CT);                          // byte offset of the NEXT prefetch
    const uint32_t stepC = (uint32_t)stride * (uint32_t)sizeof(CT);
    // diagonal wrap: when the column leaves the image on one side it re-enters on the other one
    const int xedge = dx > 0 ? width1 : -1;                               // first column value outside
    const int xback = dx > 0 ? -width1 : width1;                          // what brings it back
    const uint32_t wrapE = lane_live ? (uint32_t)(xback * D) : 0u, wrapC = (uint32_t)(xback * D) * (uint32_t)sizeof(CT);
    int x = xs, xpf = xs;                                                  // column at step t / of the next prefetch

    const uint32_t P1pk = pk_dup(a.P1);
    const int P2 = a.P2;
    const bool is_first = g == 0, is_last = g == G - 1;
    const uint32_t initL = lane_ok ? pk_dup(-a.bias) : BIGPK;            // (virtual) predecessor: L = 0 (:421-423)
    const uint32_t initDelta = pk_dup(P2 - a.bias);
    uint32_t L0 = initL, L1 = initL, L2 = initL, L3 = initL;
    uint32_t delta = initDelta;                                            // min_k L'(pred, k) + P2, both halves

    auto step = [&](raw_t raw) __attribute__((always_inline)) {
        uint32_t c0, c1, c2, c3;
        CL::unpack(raw, c0, c1, c2, c3);
        // neighbours d-1 / d+1 of every packed pair (Lr_p[-1] = Lr_p[D] = MAX_COST, :554-555)
        const uint32_t below = group_from_below<G>(L3, BIGPK, is_first);
        const uint32_t above = group_from_above<G>(L0, BIGPK, is_last);
        const uint32_t m0 = pk_min(pk_min(pk_add(pk_min(__builtin_amdgcn_alignbit(L0, below, 16), __builtin_amdgcn_alignbit(L1, L0, 16)), P1pk), L0), delta);
        const uint32_t m1 = pk_min(pk_min(pk_add(pk_min(__builtin_amdgcn_alignbit(L1, L0, 16), __builtin_amdgcn_alignbit(L2, L1, 16)), P1pk), L1), delta);
        const uint32_t m2 = pk_min(pk_min(pk_add(pk_min(__builtin_amdgcn_alignbit(L2, L1, 16), __builtin_amdgcn_alignbit(L3, L2, 16)), P1pk), L2), delta);
        const uint32_t m3 = pk_min(pk_min(pk_add(pk_min(__builtin_amdgcn_alignbit(L3, L2, 16), __builtin_amdgcn_alignbit(above, L3, 16)), P1pk), L3), delta);
        const uint32_t e0 = pk_sub(delta, m0), e1 = pk_sub(delta, m1), e2 = pk_sub(delta, m2), e3 = pk_sub(delta, m3);
        uint32_t n0 = pk_sub(c0, e0), n1 = pk_sub(c1, e1), n2 = pk_sub(c2, e2), n3 = pk_sub(c3, e3);
        u32x2 ev;
        ev.x = __builtin_amdgcn_perm(e1, e0, 0x06040200u);
        ev.y = __builtin_amdgcn_perm(e3, e2, 0x06040200u);
        __builtin_amdgcn_raw_buffer_store_b64(ev, rsE, (int)offE, 0, 0);
        offE += stepE;
        if (PAD) {    // padding lanes (d >= D) stay at MAX_COST forever
            n0 = lane_ok ? n0 : BIGPK; n1 = lane_ok ? n1 : BIGPK; n2 = lane_ok ? n2 : BIGPK; n3 = lane_ok ? n3 : BIGPK;
        }
        const uint32_t mm = pk_min(pk_min(n0, n1), pk_min(n2, n3));
        const int mn = group_min_i32<G>(min(pk_lo(mm), pk_hi(mm)));
        uint32_t dl = pk_dup(mn + P2);
        if (DIAG) {
            // next pixel of this lane group: if the column leaves the image, a NEW diagonal starts at the
            // opposite border: its predecessor is virtual (state reset) and the offsets jump back one row width
            x += dx;
            const bool wrap = x == xedge;
            x = wrap ? x + xback : x;
            offE += wrap ? wrapE : 0u;
            n0 = wrap ? initL : n0; n1 = wrap ? initL : n1; n2 = wrap ? initL : n2; n3 = wrap ? initL : n3;
            dl = wrap ? initDelta : dl;
        }
        L0 = n0; L1 = n1; L2 = n2; L3 = n3;
        delta = dl;
    };
    auto prefetch = [&]() __attribute__((always_inline)) -> raw_t {
        raw_t v = CL::load(rsC, offC);      // beyond the path end this reads a neighbour / out of range: never consumed
        offC += stepC;
        if (DIAG) {
            xpf += dx;
            const bool wrap = xpf == xedge;
            xpf = wrap ? xpf + xback : xpf;
            offC += wrap ? wrapC : 0u;
        }
        return v;
    };

    raw_t q[PF];                 // statically indexed after full unrolling: PF named register sets
    #pragma unroll
    for (int u = 0; u < PF; u++) q[u] = prefetch();
    int t = 0;
    for (; t + PF <= T; t += PF) {
        #pragma unroll
        for (int u = 0; u < PF; u++) { step(q[u]); q[u] = prefetch(); }
    }
    const int rem = T - t;       // < PF, wave-uniform
    #pragma unroll
    for (int u = 0; u < PF - 1; u++)
        if (u < rem) step(q[u]);
}

template <int G, bool PAD, typename CT, int PF>
__global__ __launch_bounds__(256) void k_aggregate(AggArgs a)
{
    int r = 0;
    #pragma unroll
    for (int i = 1; i < 8; i++) if ((int)blockIdx.x >= a.block_start[i]) r = i;
    r = __builtin_amdgcn_readfirstlane(r);          // blocks never mix directions: r is scalar
    if (r >= 4) aggregate_paths<G, PAD, CT, PF, true>(a, r);
    else        aggregate_paths<G, PAD, CT, PF, false>(a, r);
}

template <int G, typename CT>
static void launch_agg_g(hipStream_t st, int nblocks, bool pad, const AggArgs& a) {
    constexpr int PF = S2P_AGG_PF;                  // C prefetch depth in steps (8 KiB / 4 KiB in flight per wave)
    if (pad) hipLaunchKernelGGL((k_aggregate<G, true, CT, PF>), dim3(nblocks), dim3(256), 0, st, a);
    else     hipLaunchKernelGGL((k_aggregate<G, false, CT, PF>), dim3(nblocks), dim3(256), 0, st, a);
}

// lane-group size for D disparities (8 per lane): smallest power of two G with 8*G >= D
static inline int group_lanes(int D) { int G = 2; while (G * 8 < D) G *= 2; return G; }

// Enqueue the 8-direction aggregation of a [h][width1][D] cost volume (CT) into 8 e-volumes.
// The volume must stay below 4 GiB (32-bit buffer offsets); callers validate.
template <typename CT>
static void enqueue_aggregate(hipStream_t st, const CT* C, uint8_t* E, int width1, int h, int D, int P1, int P2, int bias)
{
    AggArgs aa;
    aa.C = C; aa.E = E; aa.vol = (size_t)h * width1 * D; aa.width1 = width1; aa.h = h; aa.D = D;
    aa.P1 = P1; aa.P2 = P2; aa.bias = bias;
    const int G = group_lanes(D);
    const bool pad = (G * 8 != D);
    const int np[8] = {h, h, width1, width1, width1, width1, width1, width1};   // wrapped diagonals: one path per column
    const int per_block = 4 * (64 / G);
    int nblocks = 0;
    for (int r = 0; r < 8; r++) { aa.npaths[r] = np[r]; aa.block_start[r] = nblocks; nblocks += (np[r] + per_block - 1) / per_block; }
    aa.block_start[8] = nblocks;
    switch (G) {
        case 2: launch_agg_g<2, CT>(st, nblocks, pad, aa); break;
        case 4: launch_agg_g<4, CT>(st, nblocks, pad, aa); break;
        case 8: launch_agg_g<8, CT>(st, nblocks, pad, aa); break;
        case 16: launch_agg_g<16, CT>(st, nblocks, pad, aa); break;
        case 32: launch_agg_g<32, CT>(st, nblocks, pad, aa); break;
        default: launch_agg_g<64, CT>(st, nblocks, pad, aa); break;
    }
}

}  // namespace s2p
