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
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));

// 2K costs of one lane -> K packed int16 pairs, through bounds-checked raw buffer loads (no branches).
// K = 4 (8 disparities per lane) serves D <= 512 with lane groups of up to 64 lanes; K = 8 (16 per lane)
// serves 512 < D <= 1024.
struct u32x8 { u32x4 a, b; };
// Cache policy of the e-volume traffic (CPol bits of the buffer instruction on gfx942/950: 1 = sc0, 2 = nt,
// 16 = sc1).  The 8 e-volumes (1.07 GB at 1024^2 x 128) are written once by the aggregation and read once by the
// WTA; with the default policy they sweep the 256 MB Infinity Cache and evict the cost volume that the 8
// directions re-read.  Non-temporal on both sides keeps C on-die and lets the WTA find the freshest e-lines
// still cached: measured 0.347 -> 0.308 ms (aggregation) and 0.253 -> 0.174 ms (WTA) on the census tile
// (tools/sweep_cpol.sh; sc0 / sc1 change nothing).
#ifndef S2P_C_LOAD_AUX
#define S2P_C_LOAD_AUX 0          // the cost volume is re-read by all 8 directions: keep it cached
#endif
#ifndef S2P_E_STORE_AUX
#define S2P_E_STORE_AUX 2
#endif
#ifndef S2P_E_LOAD_AUX
#define S2P_E_LOAD_AUX 2
#endif
template <typename CT, int K> struct CostLoad;
template <> struct CostLoad<int16_t, 4> {
    typedef u32x4 raw_t;
    static __device__ __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t r, uint32_t off) { return __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, S2P_C_LOAD_AUX); }
    static __device__ __forceinline__ raw_t load_last(__amdgpu_buffer_rsrc_t r, uint32_t off) { return __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, S2P_E_LOAD_AUX); }   // last use (WTA): streaming
    static __device__ __forceinline__ void unpack(raw_t v, uint32_t (&c)[4]) { c[0] = v.x; c[1] = v.y; c[2] = v.z; c[3] = v.w; }
};
template <> struct CostLoad<int16_t, 8> {
    typedef u32x8 raw_t;
    static __device__ __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t r, uint32_t off) {
        raw_t v;
        v.a = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, S2P_C_LOAD_AUX);
        v.b = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(off + 16u), 0, S2P_C_LOAD_AUX);   // off == OOB stays out of range (wraps to 15)
        return v;
    }
    static __device__ __forceinline__ raw_t load_last(__amdgpu_buffer_rsrc_t r, uint32_t off) {
        raw_t v;
        v.a = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, S2P_E_LOAD_AUX);
        v.b = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(off + 16u), 0, S2P_E_LOAD_AUX);
        return v;
    }
    static __device__ __forceinline__ void unpack(raw_t v, uint32_t (&c)[8]) {
        c[0] = v.a.x; c[1] = v.a.y; c[2] = v.a.z; c[3] = v.a.w; c[4] = v.b.x; c[5] = v.b.y; c[6] = v.b.z; c[7] = v.b.w;
    }
};
__device__ __forceinline__ void bytes_to_pairs(uint32_t w, uint32_t& lo, uint32_t& hi) {
    lo = __builtin_amdgcn_perm(0u, w, 0x0c010c00u); hi = __builtin_amdgcn_perm(0u, w, 0x0c030c02u);
}
template <> struct CostLoad<uint8_t, 4> {
    typedef u32x2 raw_t;
    // soff: wave-uniform byte offset (an SGPR operand of the instruction: no VALU add; not part of the range check)
    static __device__ __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t r, uint32_t off, uint32_t soff = 0) { return __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, (int)soff, S2P_C_LOAD_AUX); }
    static __device__ __forceinline__ void unpack(raw_t v, uint32_t (&c)[4]) { bytes_to_pairs(v.x, c[0], c[1]); bytes_to_pairs(v.y, c[2], c[3]); }
};
// 12 candidates per lane (round 6: D = 144 / 192 in the band kernel -- 16 lanes x 12 = 192, no lane of a DPP row idle at D = 192)
template <> struct CostLoad<uint8_t, 6> {
    typedef u32x3 raw_t;
    static __device__ __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t r, uint32_t off, uint32_t soff = 0) { return __builtin_amdgcn_raw_buffer_load_b96(r, (int)off, (int)soff, S2P_C_LOAD_AUX); }
    static __device__ __forceinline__ void unpack(raw_t v, uint32_t (&c)[6]) {
        bytes_to_pairs(v.x, c[0], c[1]); bytes_to_pairs(v.y, c[2], c[3]); bytes_to_pairs(v.z, c[4], c[5]);
    }
};
template <> struct CostLoad<uint8_t, 8> {
    typedef u32x4 raw_t;
    static __device__ __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t r, uint32_t off, uint32_t soff = 0) { return __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, (int)soff, S2P_C_LOAD_AUX); }
    static __device__ __forceinline__ void unpack(raw_t v, uint32_t (&c)[8]) {
        bytes_to_pairs(v.x, c[0], c[1]); bytes_to_pairs(v.y, c[2], c[3]); bytes_to_pairs(v.z, c[4], c[5]); bytes_to_pairs(v.w, c[6], c[7]);
    }
};
// the 2K e-values of a lane (each in [0, P2] <= 255) packed to bytes and stored.  AUX = cache policy of the store: non-temporal by default
// (chosen at D = 128, where the stores of a DPP row are whole lines); the band kernel passes 0 where a pixel is a quarter of a line or less
// (mgm_bands.hpp: e_store_aux).
template <int K, int AUX = S2P_E_STORE_AUX>
__device__ __forceinline__ void store_e(__amdgpu_buffer_rsrc_t rs, uint32_t off, const uint32_t (&e)[K], uint32_t soff = 0) {
    static_assert(K == 4 || K == 6 || K == 8, "8, 12 or 16 candidates per lane");
    if constexpr (K == 4) {
        u32x2 v;
        v.x = __builtin_amdgcn_perm(e[1], e[0], 0x06040200u); v.y = __builtin_amdgcn_perm(e[3], e[2], 0x06040200u);
        __builtin_amdgcn_raw_buffer_store_b64(v, rs, (int)off, (int)soff, AUX);
    } else if constexpr (K == 6) {
        u32x3 v;
        v.x = __builtin_amdgcn_perm(e[1], e[0], 0x06040200u); v.y = __builtin_amdgcn_perm(e[3], e[2], 0x06040200u); v.z = __builtin_amdgcn_perm(e[5], e[4], 0x06040200u);
        __builtin_amdgcn_raw_buffer_store_b96(v, rs, (int)off, (int)soff, AUX);
    } else {
        u32x4 v;
        v.x = __builtin_amdgcn_perm(e[1], e[0], 0x06040200u); v.y = __builtin_amdgcn_perm(e[3], e[2], 0x06040200u);
        v.z = __builtin_amdgcn_perm(e[5], e[4], 0x06040200u); v.w = __builtin_amdgcn_perm(e[7], e[6], 0x06040200u);
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)off, (int)soff, AUX);
    }
}

// the 2K e-bytes of one lane in one of the 8 e-volumes (WTA side)
template <int K> struct EBytes;
template <> struct EBytes<4> {
    typedef u32x2 raw_t;
    static __device__ __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t r, uint32_t off) { return __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, S2P_E_LOAD_AUX); }
    static __device__ __forceinline__ void get(raw_t e, int (&v)[8]) {
        v[0] = e.x & 255; v[1] = (e.x >> 8) & 255; v[2] = (e.x >> 16) & 255; v[3] = e.x >> 24;
        v[4] = e.y & 255; v[5] = (e.y >> 8) & 255; v[6] = (e.y >> 16) & 255; v[7] = e.y >> 24;
    }
};
template <> struct EBytes<6> {
    typedef u32x3 raw_t;
    static __device__ __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t r, uint32_t off) { return __builtin_amdgcn_raw_buffer_load_b96(r, (int)off, 0, S2P_E_LOAD_AUX); }
};
template <> struct EBytes<8> {
    typedef u32x4 raw_t;
    static __device__ __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t r, uint32_t off) { return __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, S2P_E_LOAD_AUX); }
    static __device__ __forceinline__ void get(raw_t e, int (&v)[16]) {
        const uint32_t w[4] = {e.x, e.y, e.z, e.w};
        #pragma unroll
        for (int i = 0; i < 4; i++) { v[4 * i] = w[i] & 255; v[4 * i + 1] = (w[i] >> 8) & 255; v[4 * i + 2] = (w[i] >> 16) & 255; v[4 * i + 3] = w[i] >> 24; }
    }
};
// the 2K costs of one lane as ints (WTA side)
template <typename CT, int K>
__device__ __forceinline__ void costs_to_ints(typename CostLoad<CT, K>::raw_t raw, int (&v)[2 * K]) {
    uint32_t c[K];
    CostLoad<CT, K>::unpack(raw, c);
    #pragma unroll
    for (int j = 0; j < K; j++) { v[2 * j] = pk_lo(c[j]); v[2 * j + 1] = pk_hi(c[j]); }
}

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
template <int G, int K, bool PAD, typename CT, int PF, bool DIAG>
__device__ __forceinline__ void aggregate_paths(const AggArgs& a, const int r)
{
    typedef CostLoad<CT, K> CL;
    constexpr int DPL = 2 * K;                 // disparities per lane
    typedef typename CL::raw_t raw_t;
    constexpr int NP = 64 / G;                 // paths per wavefront
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane & (G - 1);              // lane inside its path group
    const int width1 = a.width1, h = a.h, D = a.D;
    const int path = ((int)blockIdx.x - a.block_start[r]) * (4 * NP) + wave * NP + lane / G;
    const bool path_ok = path < a.npaths[r];
    const bool lane_ok = PAD ? (g * DPL < D) : true;
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
    const uint32_t base = (uint32_t)(ys * width1 + xs) * (uint32_t)D + (uint32_t)(g * DPL);  // element offset at t = 0 (32-bit unsigned: volumes up to 4 GiB)
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
    uint32_t L[K];
    #pragma unroll
    for (int j = 0; j < K; j++) L[j] = initL;
    uint32_t delta = initDelta;                                            // min_k L'(pred, k) + P2, both halves
    // G == 16: the edge lanes of a DPP row never receive a shifted value, so a register that starts as MAX_COST
    // everywhere and is used as the `old` operand of every row shift keeps MAX_COST there by induction: no
    // per-step re-initialisation of the fill value
    uint32_t nb_below = BIGPK, nb_above = BIGPK;

    auto step = [&](raw_t raw) __attribute__((always_inline)) {
        uint32_t c[K], e[K], n[K];
        CL::unpack(raw, c);
        // neighbours d-1 / d+1 of every packed pair (Lr_p[-1] = Lr_p[D] = MAX_COST, :554-555)
        if (G == 16) { nb_below = dpp_mov<DPP_ROW_SHR1>(L[K - 1], nb_below); nb_above = dpp_mov<DPP_ROW_SHL1>(L[0], nb_above); }
        const uint32_t below = G == 16 ? nb_below : group_from_below<G>(L[K - 1], BIGPK, is_first);
        const uint32_t above = G == 16 ? nb_above : group_from_above<G>(L[0], BIGPK, is_last);
        #pragma unroll
        for (int j = 0; j < K; j++) {
            const uint32_t dm1 = __builtin_amdgcn_alignbit(L[j], j ? L[j - 1] : below, 16);
            const uint32_t dp1 = __builtin_amdgcn_alignbit(j < K - 1 ? L[j + 1] : above, L[j], 16);
            const uint32_t m = pk_min(pk_min(pk_add(pk_min(dm1, dp1), P1pk), L[j]), delta);
            e[j] = pk_sub(delta, m);
            n[j] = pk_sub(c[j], e[j]);
        }
        store_e<K>(rsE, offE, e);
        offE += stepE;
        if (PAD) {    // padding lanes (d >= D) stay at MAX_COST forever
            #pragma unroll
            for (int j = 0; j < K; j++) n[j] = lane_ok ? n[j] : BIGPK;
        }
        uint32_t mm = pk_min(pk_min(n[0], n[1]), pk_min(n[2], n[3]));
        #pragma unroll
        for (int j = 4; j < K; j += 4) mm = pk_min(mm, pk_min(pk_min(n[j], n[j + 1]), pk_min(n[j + 2], n[j + 3])));
        const int mn = group_min_i32<G>(min(pk_lo(mm), pk_hi(mm)));
        uint32_t dl = pk_dup(mn + P2);
        if (DIAG) {
            // next pixel of this lane group: if the column leaves the image, a NEW diagonal starts at the
            // opposite border: its predecessor is virtual (state reset) and the offsets jump back one row width
            x += dx;
            const bool wrap = x == xedge;
            x = wrap ? x + xback : x;
            offE += wrap ? wrapE : 0u;
            #pragma unroll
            for (int j = 0; j < K; j++) n[j] = wrap ? initL : n[j];
            dl = wrap ? initDelta : dl;
        }
        #pragma unroll
        for (int j = 0; j < K; j++) L[j] = n[j];
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

template <int G, int K, bool PAD, typename CT, int PF>
__global__ __launch_bounds__(256) void k_aggregate(AggArgs a)
{
    int r = 0;
    #pragma unroll
    for (int i = 1; i < 8; i++) if ((int)blockIdx.x >= a.block_start[i]) r = i;
    r = __builtin_amdgcn_readfirstlane(r);          // blocks never mix directions: r is scalar
    if (r >= 4) aggregate_paths<G, K, PAD, CT, PF, true>(a, r);
    else        aggregate_paths<G, K, PAD, CT, PF, false>(a, r);
}

template <int G, int K, typename CT>
static void launch_agg_g(hipStream_t st, int nblocks, bool pad, const AggArgs& a) {
    constexpr int PF = K == 4 ? S2P_AGG_PF : S2P_AGG_PF / 2;    // C prefetch depth in steps (same bytes in flight)
    if (pad) hipLaunchKernelGGL((k_aggregate<G, K, true, CT, PF>), dim3(nblocks), dim3(256), 0, st, a);
    else     hipLaunchKernelGGL((k_aggregate<G, K, false, CT, PF>), dim3(nblocks), dim3(256), 0, st, a);
}

// Lane layout for D disparities: K packed dwords (2K disparities) per lane, groups of G lanes per pixel.
// D <= 512: K = 4 and the smallest power of two G with 8 G >= D; 512 < D <= 1024: K = 8, G = 64.
struct LaneLayout { int G, K; bool pad; };
static inline LaneLayout lane_layout(int D) {
    LaneLayout l;
    if (D > 512) { l.K = 8; l.G = 64; }
    else { l.K = 4; l.G = 2; while (l.G * 8 < D) l.G *= 2; }
    l.pad = (l.G * 2 * l.K != D);
    return l;
}
#define S2P_MAX_DISPARITIES 1024

// Enqueue the 8-direction aggregation of a [h][width1][D] cost volume (CT) into 8 e-volumes.
// The volume (in bytes) must stay below 4 GiB (32-bit unsigned buffer offsets); callers validate.
template <typename CT>
static void enqueue_aggregate(hipStream_t st, const CT* C, uint8_t* E, int width1, int h, int D, int P1, int P2, int bias, int nd = 8)
{
    AggArgs aa;
    aa.C = C; aa.E = E; aa.vol = (size_t)h * width1 * D; aa.width1 = width1; aa.h = h; aa.D = D;
    aa.P1 = P1; aa.P2 = P2; aa.bias = bias;
    // uint8 costs at D >= 128: 16 disparities per lane (K = 8, G = D / 16).  Measured on the census tile
    // (1024^2 x 128): 0.307 -> 0.282 ms -- 12 % fewer issue slots per candidate (the per-lane overheads of the
    // group reduction and the neighbour exchange are shared by twice the disparities).  int16 costs lose with
    // it (0.56 -> 0.64 ms: two 128-bit loads per step and twice the live registers), so they keep K = 4.
    LaneLayout ll = lane_layout(D);
    if (sizeof(CT) == 1 && D >= 128 && D <= 512) { ll.K = 8; ll.G = 8; while (ll.G * 16 < D) ll.G *= 2; ll.pad = ll.G * 16 != D; }
    const int G = ll.G;
    const bool pad = ll.pad;
    int np[8] = {h, h, width1, width1, width1, width1, width1, width1};         // wrapped diagonals: one path per column
    for (int r = nd; r < 8; r++) np[r] = 0;                                      // nd = 4: the axis directions only
    const int per_block = 4 * (64 / G);
    int nblocks = 0;
    for (int r = 0; r < 8; r++) { aa.npaths[r] = np[r]; aa.block_start[r] = nblocks; nblocks += (np[r] + per_block - 1) / per_block; }
    aa.block_start[8] = nblocks;
    if (ll.K == 8) {
        if (sizeof(CT) == 1) {
            switch (G) {
                case 8: launch_agg_g<8, 8, uint8_t>(st, nblocks, pad, aa); return;
                case 16: launch_agg_g<16, 8, uint8_t>(st, nblocks, pad, aa); return;
                case 32: launch_agg_g<32, 8, uint8_t>(st, nblocks, pad, aa); return;
                default: break;
            }
        }
        launch_agg_g<64, 8, CT>(st, nblocks, pad, aa);
        return;
    }
    switch (G) {
        case 2: launch_agg_g<2, 4, CT>(st, nblocks, pad, aa); break;
        case 4: launch_agg_g<4, 4, CT>(st, nblocks, pad, aa); break;
        case 8: launch_agg_g<8, 4, CT>(st, nblocks, pad, aa); break;
        case 16: launch_agg_g<16, 4, CT>(st, nblocks, pad, aa); break;
        case 32: launch_agg_g<32, 4, CT>(st, nblocks, pad, aa); break;
        default: launch_agg_g<64, 4, CT>(st, nblocks, pad, aa); break;
    }
}

}  // namespace s2p
