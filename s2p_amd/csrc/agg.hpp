// s2p_amd/csrc/agg.hpp -- the hot kernel: 8-path semi-global aggregation as wavefront recurrences,
// generic over the cost element type (int16 for the sgbm matcher, uint8 for the census matcher).
#pragma once
#include "common.hpp"

namespace s2p {

// =============================================================================================
// K3: 8-path semi-global aggregation (stereosgbm.cpp:518-662), all directions in one launch.
// =============================================================================================
struct AggArgs {
    const void* C;              // int16 (sgbm: +P2 bias already inside) or uint8 (census: bias added on load)
    uint8_t* E;                 // 8 volumes, each vol elements
    size_t vol;                 // h * width1 * D
    int width1, h, D, P1, P2;
    int bias;                   // added to every loaded cost (0 for sgbm, P2 for census)
    int block_start[9];         // first block of direction r (prefix sums); blocks never mix directions
    int npaths[8];
};

#define BIGPK 0x3fff3fffu       // "MAX_COST" stand-in for Lr[-1], Lr[D]: any value that loses every min

// 8 costs of one lane -> 4 packed int16 pairs
__device__ __forceinline__ uint4 load_costs8(const int16_t* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ uint4 load_costs8(const uint8_t* p) {
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    uint4 r;
    r.x = __builtin_amdgcn_perm(0u, v.x, 0x0c010c00u); r.y = __builtin_amdgcn_perm(0u, v.x, 0x0c030c02u);
    r.z = __builtin_amdgcn_perm(0u, v.y, 0x0c010c00u); r.w = __builtin_amdgcn_perm(0u, v.y, 0x0c030c02u);
    return r;
}

template <int G, bool PAD, typename CT>
__global__ __launch_bounds__(256) void k_aggregate(AggArgs a)
{
    constexpr int NP = 64 / G;                 // paths per wavefront
    constexpr int PF = 4;                      // C prefetch depth (steps)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane & (G - 1);              // lane inside its path group
    int r = 0;
    #pragma unroll
    for (int i = 1; i < 8; i++) if ((int)blockIdx.x >= a.block_start[i]) r = i;
    r = __builtin_amdgcn_readfirstlane(r);
    const int width1 = a.width1, h = a.h, D = a.D;
    const int path = ((int)blockIdx.x - a.block_start[r]) * (4 * NP) + wave * NP + lane / G;
    const bool path_ok = path < a.npaths[r];
    const bool lane_ok = PAD ? (g * 8 < D) : true;

    // path geometry: pixel(t) = (xs + t*dx, ys + t*dy), t in [0, T)
    int xs, ys, dx, dy, T;
    const bool diag = r >= 4;
    switch (r) {
        case 0: xs = 0; ys = path; dx = 1; dy = 0; T = width1; break;
        case 1: xs = width1 - 1; ys = path; dx = -1; dy = 0; T = width1; break;
        case 2: xs = path; ys = 0; dx = 0; dy = 1; T = h; break;
        case 3: xs = path; ys = h - 1; dx = 0; dy = -1; T = h; break;
        case 4: xs = path - (h - 1); ys = 0; dx = 1; dy = 1; T = h; break;               // s = x - y
        case 5: xs = path; ys = 0; dx = -1; dy = 1; T = h; break;                        // s = x + y
        case 6: xs = path - (h - 1) + (h - 1); ys = h - 1; dx = -1; dy = -1; T = h; break;  // reverse of 4
        default: xs = path - (h - 1); ys = h - 1; dx = 1; dy = -1; T = h; break;          // reverse of 5
    }
    // wave-uniform trip range: union of the active ranges of this wave's paths
    int t0 = 0, t1 = T;
    if (diag) {
        // active(t) <=> 0 <= xs + t*dx < width1
        int lo, hi;   // this path's [lo, hi)
        if (dx > 0) { lo = max(0, -xs); hi = min(T, width1 - xs); }
        else        { lo = max(0, xs - (width1 - 1)); hi = min(T, xs + 1); }
        if (!path_ok || hi <= lo) { lo = T; hi = 0; }
        // wave-wide min/max
        for (int o = 32; o; o >>= 1) { lo = min(lo, __shfl_xor(lo, o)); hi = max(hi, __shfl_xor(hi, o)); }
        t0 = __builtin_amdgcn_readfirstlane(lo);
        t1 = __builtin_amdgcn_readfirstlane(hi);
        if (t1 <= t0) return;
    } else if (!__any(path_ok)) return;

    const long stride = ((long)dy * width1 + dx) * D;          // elements per step
    const long base = ((long)ys * width1 + xs) * D + g * 8;    // element offset at t = 0
    const CT* Cp = reinterpret_cast<const CT*>(a.C) + base;
    const uint32_t biaspk = pk_dup(a.bias);
    uint8_t* Ep = a.E + (size_t)r * a.vol + base;

    const uint32_t P1pk = pk_dup(a.P1);
    const int P2 = a.P2;
    const bool is_first = g == 0, is_last = g == G - 1;

    auto is_active = [&](int t) -> bool {
        if (!path_ok || !lane_ok) return false;
        if (!diag) return true;
        int x = xs + t * dx;
        return x >= 0 && x < width1;
    };
    auto load_c = [&](int t) -> uint4 {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (t < t1 && is_active(t)) {
            v = load_costs8(Cp + (long)t * stride);
            if (sizeof(CT) == 1) { v.x = pk_add(v.x, biaspk); v.y = pk_add(v.y, biaspk); v.z = pk_add(v.z, biaspk); v.w = pk_add(v.w, biaspk); }
        }
        return v;
    };

    uint32_t L0 = lane_ok ? 0u : BIGPK, L1 = L0, L2 = L0, L3 = L0;   // Lr of the (virtual) predecessor: 0 (:421-423)
    uint32_t delta = pk_dup(P2);                                       // minLr(pred) + P2, both halves

    uint4 cb[PF];
    #pragma unroll
    for (int u = 0; u < PF; u++) cb[u] = load_c(t0 + u);

    for (int tb = t0; tb < t1; tb += PF) {
        #pragma unroll
        for (int u = 0; u < PF; u++) {
            const int t = tb + u;
            if (t >= t1) break;
            const uint4 c4 = cb[u];
            cb[u] = load_c(t + PF);
            const bool act = is_active(t);
            // neighbours d-1 / d+1 of every packed pair (Lr_p[-1] = Lr_p[D] = MAX_COST, :554-555)
            const uint32_t below = group_from_below<G>(L3, BIGPK, is_first);
            const uint32_t above = group_from_above<G>(L0, BIGPK, is_last);
            const uint32_t m0 = pk_min(pk_min(pk_add(pk_min(__builtin_amdgcn_alignbit(L0, below, 16), __builtin_amdgcn_alignbit(L1, L0, 16)), P1pk), L0), delta);
            const uint32_t m1 = pk_min(pk_min(pk_add(pk_min(__builtin_amdgcn_alignbit(L1, L0, 16), __builtin_amdgcn_alignbit(L2, L1, 16)), P1pk), L1), delta);
            const uint32_t m2 = pk_min(pk_min(pk_add(pk_min(__builtin_amdgcn_alignbit(L2, L1, 16), __builtin_amdgcn_alignbit(L3, L2, 16)), P1pk), L2), delta);
            const uint32_t m3 = pk_min(pk_min(pk_add(pk_min(__builtin_amdgcn_alignbit(L3, L2, 16), __builtin_amdgcn_alignbit(above, L3, 16)), P1pk), L3), delta);
            const uint32_t e0 = pk_sub(delta, m0), e1 = pk_sub(delta, m1), e2 = pk_sub(delta, m2), e3 = pk_sub(delta, m3);
            uint32_t n0 = pk_sub(c4.x, e0), n1 = pk_sub(c4.y, e1), n2 = pk_sub(c4.z, e2), n3 = pk_sub(c4.w, e3);
            if (act) {
                uint2 ev;
                ev.x = __builtin_amdgcn_perm(e1, e0, 0x06040200u);
                ev.y = __builtin_amdgcn_perm(e3, e2, 0x06040200u);
                *reinterpret_cast<uint2*>(Ep + (long)t * stride) = ev;
            }
            if (PAD || diag) {
                // inactive (path not started): state stays the virtual predecessor; padded lanes stay BIG
                const uint32_t idle = lane_ok ? 0u : BIGPK;
                n0 = act ? n0 : idle; n1 = act ? n1 : idle; n2 = act ? n2 : idle; n3 = act ? n3 : idle;
            }
            L0 = n0; L1 = n1; L2 = n2; L3 = n3;
            const uint32_t mm = pk_min(pk_min(n0, n1), pk_min(n2, n3));
            const int mn = group_min_i32<G>(min(pk_lo(mm), pk_hi(mm)));
            delta = pk_dup(mn + P2);
        }
    }
}


template <int G, typename CT>
static void launch_agg_g(hipStream_t st, int nblocks, bool pad, const AggArgs& a) {
    if (pad) hipLaunchKernelGGL((k_aggregate<G, true, CT>), dim3(nblocks), dim3(256), 0, st, a);
    else     hipLaunchKernelGGL((k_aggregate<G, false, CT>), dim3(nblocks), dim3(256), 0, st, a);
}

// lane-group size for D disparities (8 per lane): smallest power of two G with 8*G >= D
static inline int group_lanes(int D) { int G = 2; while (G * 8 < D) G *= 2; return G; }

// Enqueue the 8-direction aggregation of a [h][width1][D] cost volume (CT) into 8 e-volumes.
template <typename CT>
static void enqueue_aggregate(hipStream_t st, const CT* C, uint8_t* E, int width1, int h, int D, int P1, int P2, int bias)
{
    AggArgs aa;
    aa.C = C; aa.E = E; aa.vol = (size_t)h * width1 * D; aa.width1 = width1; aa.h = h; aa.D = D;
    aa.P1 = P1; aa.P2 = P2; aa.bias = bias;
    const int G = group_lanes(D);
    const bool pad = (G * 8 != D);
    const int np[8] = {h, h, width1, width1, width1 + h - 1, width1 + h - 1, width1 + h - 1, width1 + h - 1};
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
