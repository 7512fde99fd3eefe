// s2p_amd/csrc/mgm_bands.hpp -- MGM recursion (oracle/census_oracle.c, recursion = 1), band-pipelined: ONE launch per tile.
// Included by census_kernels.hip (needs agg.hpp, mgm_geom.hpp, pk_shr1).
//
// The 12 quadrant lattices of mgm_geom.hpp (52 with 16 directions) are cut into BANDS of R = 256 / G consecutive v-rows; one 256-thread
// workgroup (+ its fetcher wave) owns a band and sweeps u with its R lane groups skewed by one step (row j is at u = T - j in step T), so
// that both predecessors of a point were produced one step earlier: (u - 1, v) by the group itself (registers),
// (u, v - 1) by the group of row j - 1.  What travels is the MESSAGE of a point (computed once by its producer, used by
// its two successors), never L.
//
// Round 1 ran the four waves of a band in lock step (LDS exchange + one s_barrier per step) and chained the bands
// through a flag protocol (write-through rows, drained progress counters, polled one chunk ahead).  Its per-band
// trace showed where the time went: a band alone on its CU took 0.33 us per step whether the step had 95 or 80
// VALU instructions -- the step was the latency chain ds_write -> s_barrier -> ds_read -> 22 dependent VALU, not
// issue -- and a band started 5.5 us (hand-off) + 15 steps (skew) after its predecessor, 64 times in a row.
// This version removes both serialisations:
//   * NO BARRIER IN THE SWEEP.  The rows of a band exchange messages through an LDS ring `chan[row][T & 7]` (entry
//     written in step T, read in step T + 1).  The rows of one wave read what the same wave wrote one instruction
//     earlier (DS operations of a wave execute in order: no wait in between); across waves the producer publishes
//     its step count in an LDS word right behind the data (same in-order queue: no wait either) and the consumer
//     polls that word only when its cached copy is not enough.  Back-pressure uses the same words (an entry is
//     rewritten 8 steps later).  Waves drift apart by a step or two instead of meeting 1 000 times per sweep.
//   * NO FLAGS BETWEEN BANDS.  The last row of band k goes to global memory as self-validating 16-byte granules:
//     messages are <= P2 <= 128, so the high byte of every 16-bit field is free and carries a tag (since round 4 the two
//     bytes of a dword together hold the 16-bit count 1 + (k >> 1); before: 1 + (k >> 1) mod
//     255; the two-slot row ring and the control block are zeroed by a memset node in front of every launch, so a
//     granule of an earlier launch or of band k - 2 never passes).  A FIFTH WAVE of band k + 1, the fetcher, does
//     nothing but bring that row in: it keeps two 1 KB loads (groups of 4 points at G = 16) in flight, stages
//     whatever prefix of the oldest group carries the right tags into the LDS ring of row 0 -- point by point, as
//     they arrive -- publishes the count in a progress word like any other wave (wave 0 waits on it exactly as
//     wave 1 waits on wave 0), and simply asks again for a group that is not complete: no producer-side drain, no
//     counter, no poll of a second location.  (Until the middle of round 2 wave 0 staged chunks of 8 points itself:
//     a band could only enter a chunk when its LAST point had arrived, 7 steps of extra distance per band, and the
//     retries sat on the critical wave; per-band trace: start-to-start 9.6 -> 8.1 us.)  Stores and loads are
//     write-through / L2-bypassing (sc0 sc1) on both sides, as MI355X_MICROARCH.md prescribes for cross-XCD data; a
//     tag sits in EVERY dword of a granule, so not even a torn 16-byte store could pass.
//     Overwrite safety needs no gate: band k + 2 writes (slot, u) only after its row 0 consumed band k + 1's last row
//     at u, which -- the in-image part of a lattice column is one interval, the lattices are convex -- descends from
//     band k + 1's row 0 at u, which consumed band k's granule from its LDS copy.  (Rows are only stored where the
//     point lies in the image; points outside send nothing and nobody waits for them.)
// Bands take their identity from an atomic ticket in band-major order, so a band only ever waits for a workgroup that
// already runs: no residency assumption.  Every wait is bounded; a timeout raises ctl[1] (checked by the host entry
// points) and lets the launch drain.
#pragma once

namespace s2p {

#define S2P_MGM_PF 16                 // cost prefetch depth in steps (= unroll of the sweep; a multiple of the LDS ring entries).  One tile alone
                                      // does not care (8 / 16 / 32: 1.050 / 1.060 / 1.071 ms); with the chip full the loaded memory latency is what a
                                      // step waits for: 8 tiles per launch 4.58 / 4.40 / 4.23 ms, whole tiles on two streams 0.735 / 0.68-0.70 / 0.739
                                      // (32 costs a wave per SIMD): tools/pf_probe.sh, profiles/r03/pf_probe.txt.  16 disparities per lane: 8.
// Build switches of this file (all three compile only under S2P_PROBE_BUILD, csrc/probe_guard.hpp): S2P_MGM_TRACE (per-band records),
// S2P_MGM_PROBE_NOPOLL and S2P_MGM_PROBE_NOMEM (timing probes, results invalid: the launch without its flow control / without its
// memory traffic -- the decomposition of profiles/r06/decompose_probe.txt).  Every other switch this file carried until round 5 had a
// final verdict and is gone; the verdicts and the files that hold them are listed in docs/notebook/10_round6_switches.md.
#define S2P_MGM_SLEEP 1               // s_sleep argument of the LDS polls (64 cycles each)
#define S2P_HANDOFF_ST_AUX 17         // sc0 | sc1: write-through stores ...
#define S2P_HANDOFF_LD_AUX 17         // ... and L1/L2-bypassing loads
// how many steps a wave may run ahead of the wave below it (<= ring length - 2).  The waves of a band carry different
// loads (the last wave stores the outgoing row), so they drift apart as far as they are
// allowed to -- and every step of drift is a step added to the distance the next band keeps.
#define S2P_MGM_LEAD 0                // 0 = ring length - 2
// Compute waves per band, by lane layout (R = waves * 64 / G rows per band; the fetcher comes on top).  More rows per band =
// fewer band-to-band hand-offs on the chain, but waves beyond 4 share SIMDs with each other and the rings of a band grow:
//   G <= 8  (D <= 64, 8+ rows per wave): 4 -- with 8 the bands get 64-256 rows, no hand-off left to save (1024^2 x 32: 0.87 vs 0.74 ms)
//   G = 16, 32 (D = 128, 256): 8 (1024^2 x 128: 4 / 8 waves tie at one tile, 8 wins with tiles in flight and at D = 256: 512^2 x 256 0.66 -> 0.57;
//                                  12 / 15 at G = 32 lose: 1000^2 x 256 1.43 / 1.48 / 1.69)
//   G = 64, K = 4 (256 < D <= 512, one row per wave): 15, all a workgroup holds (1000^2 x 512: 4 / 8 / 12 / 15 waves 2.80 / 3.50 / 3.01 / 2.77 ms)
//   G = 64, K = 8 (D > 512): 8 (147 KB of rings)
#define S2P_MGM_NW_WIDE 8
#define S2P_MGM_NW_NARROW 4
#define S2P_MGM_NW_G64 15
#define S2P_MGM_NW_G32 S2P_MGM_NW_WIDE
constexpr int mgm_waves(int G, int K) { return G >= 64 ? (K > 4 ? S2P_MGM_NW_WIDE : S2P_MGM_NW_G64) : G == 32 ? S2P_MGM_NW_G32 : G >= 16 ? S2P_MGM_NW_WIDE : S2P_MGM_NW_NARROW; }
// A batch (several tiles under one queue: the chip is full whatever the bands look like) at D = 128 runs 4-wave bands: one tile
// alone loses with them (launch 1.03 -> 1.18 ms: twice the hand-offs on its chain), 8 tiles per launch gain 5 % (4.32 -> 4.11 ms:
// two compute waves per SIMD instead of four; profiles/r03/nw4_probe.txt).  Only measured there, only used there.
#define S2P_MGM_NW_BATCH_G16 4
constexpr int mgm_waves(int G, int K, bool batch) { return (batch && G == 16 && K == 4) ? S2P_MGM_NW_BATCH_G16 : mgm_waves(G, K); }
#define S2P_MGM_FSLEEP 1              // s_sleep between two polls of the fetcher
#define S2P_MGM_SPIN_LIMIT (1u << 22)
// entries of every LDS ring (the sweep is unrolled by a multiple of it; a wave may lead the next by ring - 2 steps).  The
// rings are what limits how many bands are resident, and a lattice needs ~ U / (R + 5) of its bands in flight: 12 chains
// of a 1000-step sweep want 25 MB of rings at D = 128 but 98 MB at D = 512 -- the chip has 41 MB of LDS.  Rings of 4 for
// the wide rows (twice the bands in flight, tighter coupling) were measured: 1000^2 x 512 with 8-wave bands 3.50 -> 2.96 ms,
// but 15-wave bands with rings of 8 do better (2.77) and rings of 4 do not help those (3.55); at D = 256 they lose (1.41 -> 1.56).
#define S2P_MGM_RING4_FROM 4096       // LW = G * K (dwords per row message) from which the rings have 4 entries: never
#define S2P_MGM_RING16_UPTO 16        // LW up to which the rings have 16 entries (D <= 32: 4-wave bands of 64+ rows; a wave may lead by 14).  Until
                                      // round 6 also LW = 32 (D = 64), chosen on a tile alone; with the chip full rings of 8 there run the 8-tile launch of
                                      // 1024^2 in 2.70 instead of 3.34 ms and 512^2 in 0.75 instead of 0.94 (a tile alone 0.83 -> 0.85); at D = 32 / 16
                                      // rings of 8 lose (2.42 -> 2.82, 2.77 -> 3.01): profiles/r06/smalld_probe.txt
constexpr int mgm_ring(int LW) { return LW >= S2P_MGM_RING4_FROM ? 4 : LW <= S2P_MGM_RING16_UPTO ? 16 : 8; }
// cache policy of the e-stores.  Non-temporal (S2P_E_STORE_AUX = 2) was chosen at D = 128, where the stores of a DPP row are whole 128-byte
// lines.  At D <= 32 (LW <= 16) a pixel's e-bytes of one direction are a quarter or an eighth of a line, every store is a partial line by
// construction and the x-neighbours that complete the line are written by the neighbouring lattice rows a step or two later: plain stores let
// the L2 merge them before the line leaves (round 6, profiles/r06/cpol_small_probe.txt: the 8-tile launch of 1024^2 2.74 -> 1.31 ms at
// D = 16, 2.44 -> 1.89 at D = 32; at D = 48 ... 128 plain stores lose 2-6 % with calls in flight and stay non-temporal).
#ifndef S2P_MGM_E_PLAIN_UPTO
#define S2P_MGM_E_PLAIN_UPTO 16       // LW = G * K up to which the band kernel's e-stores are plain
#endif
constexpr int e_store_aux(int LW) { return LW <= S2P_MGM_E_PLAIN_UPTO ? 0 : S2P_E_STORE_AUX; }
// dwords from one band row's ring to the next in LDS.  A 16-byte-per-lane LDS access is served 16 lanes at a time; with fewer than 16 lanes
// per pixel those 16 lanes are 16 / G ROWS, and rings that start a multiple of 256 bytes apart put them on the same banks (G = 8: 2-way,
// G = 4: 4-way, G = 2: 8-way conflicts on every read and write of the step).  One entry of slack per row staggers them (round 6).
#ifndef S2P_MGM_LDS_SKEW
#define S2P_MGM_LDS_SKEW 1
#endif
constexpr int mgm_row_stride(int G, int LW) { return mgm_ring(LW) * LW + ((S2P_MGM_LDS_SKEW && G < 16) ? LW : 0); }

#define S2P_MGM_HETERO_MAX 16
struct MgmBandArgs {
    const uint8_t* C; uint8_t* E; size_t vol;
    int w, h, D, P1, P2;
    int nbands;           // max over the lattices of ceil(V / R)
    int nlat;             // lattices swept: 12, 4 = the axis directions only (nb_dir = 4), or 52 = 16 directions (mgm_geom.hpp)
    int upad;             // row length of the hand-off ring (max U rounded up to 8)
    uint32_t* rows;       // [max(nlat, 12)][2][upad][G * K] tagged messages of a band's last row
    uint32_t rows_bytes;
    uint32_t* ctl;        // [0] items popped so far, [1] items published so far; the published items follow at ctl + 64
    uint32_t* abortw;     // raised by a wait that timed out (one word per context, checked by the host entry points)
    // work items (bands) of the launch: `total` of them over `ntiles` tiles; the first `ninit` (band 0 of every lattice of every
    // tile) need no publication, every other band is published by its predecessor once that one is under way
    int total, ninit, ntiles;
    int stagger;                  // batch: < 0 every tile's lattices start at once; else band (nbands * stagger) >> 8 of lattice q of
                                  // tile t publishes band 0 of lattice q of tile t + 1 (only tile 0 is in the queue from the start)
    size_t c_stride, e_stride;    // byte distance between the cost volumes / the e-volume sets of consecutive tiles of a batch
    uint32_t* trace;              // -DS2P_MGM_TRACE: per-band records
    // a batch of tiles of DIFFERENT sizes (round 4; one lane layout, i.e. one D, for all of them): tile t is tw[t] x th[t], its cost
    // volume starts 256 * c_off[t] bytes into C, its e-volumes 256 * e_off[t] bytes into E, each tvol[t] bytes; w / h / vol / the
    // strides above are then unused.  The hand-off rings keep one (maximal) shape for every tile.
    int hetero;
    int tw[S2P_MGM_HETERO_MAX], th[S2P_MGM_HETERO_MAX];
    uint32_t c_off[S2P_MGM_HETERO_MAX], e_off[S2P_MGM_HETERO_MAX], tvol[S2P_MGM_HETERO_MAX];
};
// item = ((tile * 64 + lattice) << 12) | band  (52 lattices with 16 directions; tile < 8192)
#define S2P_MGM_ITEM(tile, q, band) ((((tile) * 64 + (q)) << 12) | (band))
#define S2P_MGM_TILES_MAX 8191
#define S2P_MGM_TRIG 16               // steps into its sweep at which a band publishes its successor (the successor's first
                                      // input is produced at step R - 1; its own prologue takes ~10 steps)

// wave-uniform bounded wait for an LDS progress word to reach `need`; returns the value seen (>= need), or `need` after
// a timeout / abort with `waiting` cleared (the caller stops waiting and drains)
__device__ __forceinline__ int mgm_wait_lds(int* p, int need, uint32_t* abortw, bool& waiting)
{
    int v = need;
    if (waiting) {
        for (uint32_t it = 0;; ++it) {
            v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            if (v >= need) break;
            if ((it & 255u) == 255u) {
                if (__hip_atomic_load(abortw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { waiting = false; v = need; break; }
                if (it > S2P_MGM_SPIN_LIMIT) { __hip_atomic_store(abortw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); waiting = false; v = need; break; }
            }
            __builtin_amdgcn_s_sleep(S2P_MGM_SLEEP);
        }
    }
    asm volatile("" ::: "memory");       // the data reads that follow stay behind the poll
    return v;
}

// K dwords of a lane to / from an LDS row: 16-byte accesses where the lane's slice is 16-byte aligned (K = 4, 8), 8-byte ones for K = 6
template <int K> __device__ __forceinline__ void lds_get(const uint32_t* p, uint32_t (&v)[K]) {
    if constexpr (K % 4 == 0) {
        #pragma unroll
        for (int i = 0; i < K; i += 4) { const u32x4 t = *reinterpret_cast<const u32x4*>(p + i); v[i] = t.x; v[i + 1] = t.y; v[i + 2] = t.z; v[i + 3] = t.w; }
    } else {
        #pragma unroll
        for (int i = 0; i < K; i += 2) { const u32x2 t = *reinterpret_cast<const u32x2*>(p + i); v[i] = t.x; v[i + 1] = t.y; }
    }
}
template <int K> __device__ __forceinline__ void lds_add(const uint32_t* p, uint32_t (&v)[K]) {
    if constexpr (K % 4 == 0) {
        #pragma unroll
        for (int i = 0; i < K; i += 4) { const u32x4 t = *reinterpret_cast<const u32x4*>(p + i); v[i] += t.x; v[i + 1] += t.y; v[i + 2] += t.z; v[i + 3] += t.w; }
    } else {
        #pragma unroll
        for (int i = 0; i < K; i += 2) { const u32x2 t = *reinterpret_cast<const u32x2*>(p + i); v[i] += t.x; v[i + 1] += t.y; }
    }
}
template <int K> __device__ __forceinline__ void lds_put(uint32_t* p, const uint32_t (&v)[K]) {
    if constexpr (K % 4 == 0) {
        #pragma unroll
        for (int i = 0; i < K; i += 4) { u32x4 t; t.x = v[i]; t.y = v[i + 1]; t.z = v[i + 2]; t.w = v[i + 3]; *reinterpret_cast<u32x4*>(p + i) = t; }
    } else {
        #pragma unroll
        for (int i = 0; i < K; i += 2) { u32x2 t; t.x = v[i]; t.y = v[i + 1]; *reinterpret_cast<u32x2*>(p + i) = t; }
    }
}

// (a + 1) / 3 on both 16-bit fields of `a1` = a + 0x00010001, for fields < 384: x * 171 >> 9 == x / 3 for x < 512 and the product
// stays below 2^16 for x <= 383 (3 messages <= P2 <= 127 each, + 1)
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_div3(uint32_t a1) {
    u16x2 v = __builtin_bit_cast(u16x2, a1);
    v = (v * (unsigned short)171) >> (unsigned short)9;
    return __builtin_bit_cast(uint32_t, v);
}

// NQ = predecessors of a lattice point whose messages are averaged: 2 = (u - 1, v), (u, v - 1) (recursion = 1);
// 3 = those and (u - 1, v - 1) (recursion = 2: TSGM = 3 of the 'mgm' call site).  The third message is the ring entry the
// row above wrote TWO steps ago, so a wave may lead the next one by one step less.
template <int G, int K, bool PAD, int NQ, int NW>
__global__ __launch_bounds__(64 * (NW + 1)) void k_mgm_bands(MgmBandArgs a)
{
    constexpr int NT = 64 * (NW + 1), DPL = 2 * K, NP = 64 / G, R = NW * NP, LW = G * K;
    constexpr int RING = mgm_ring(LW), PFW = K > 4 ? 8 : S2P_MGM_PF, PF = PFW > RING ? PFW : RING;
    constexpr int LEADMAX = RING - NQ;                                   // an entry is read for NQ - 1 steps after it was written
    constexpr int LEAD = (S2P_MGM_LEAD > 0 && S2P_MGM_LEAD < LEADMAX) ? S2P_MGM_LEAD : LEADMAX;
    constexpr int GPU = LW / 4;                                          // 16-byte granules per point of a row
    constexpr int EAUX = e_store_aux(LW);
    static_assert(PF % RING == 0 && (RING & (RING - 1)) == 0, "the sweep is unrolled by a multiple of the ring length");
    static_assert(NQ == 2 || NQ == 3, "two or three predecessors");
    static_assert(LEAD >= 0 && LEAD <= RING - NQ, "a ring entry is rewritten RING steps later");
    typedef CostLoad<uint8_t, K> CL;
    typedef typename CL::raw_t raw_t;
    // chan[row][entry][LW]: row 0 = messages of the previous band's last row (staged by wave 0), row j + 1 = output of band row j
    constexpr int RS = mgm_row_stride(G, LW);                            // dwords per band row in LDS (its ring + the bank stagger)
    __shared__ __attribute__((aligned(16))) uint32_t chan[(R + 1) * RS];
    __shared__ int s_prog[NW + 1];                                            // next step each wave will execute
    __shared__ int s_ticket, s_range[2];
  for (;;) {   // a workgroup is a WORKER: it takes band after band from the launch's queue until the queue is exhausted
    if (threadIdx.x == 0) {
        int item = -1;
        const uint32_t idx = atomicAdd(a.ctl, 1u);
        if (idx < (uint32_t)a.ninit) item = S2P_MGM_ITEM((int)idx / a.nlat, (int)idx % a.nlat, 0);
        else if (idx < (uint32_t)a.total) {
            // published by the band above it in the lattice, which runs (or ran): a bounded wait on a live producer
            const uint32_t* slot = a.ctl + 64 + (idx - (uint32_t)a.ninit);
            for (uint32_t it = 0;; ++it) {
                const uint32_t v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v) { item = (int)v - 1; break; }
                if ((it & 63u) == 63u) {
                    if (__hip_atomic_load(a.abortw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                    if (it > (S2P_MGM_SPIN_LIMIT >> 2)) { __hip_atomic_store(a.abortw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                }
                __builtin_amdgcn_s_sleep(8);
            }
        }
        s_ticket = item; s_range[0] = 0x7fffffff; s_range[1] = 0;
    }
    for (int i = threadIdx.x; i < (R + 1) * RS; i += NT) chan[i] = 0;
    __syncthreads();
    const int item = s_ticket;
    if (item < 0) return;                                                // queue exhausted (or the launch was aborted)
    const int band = item & 4095, q = (item >> 12) & 63, tile = item >> 18;
    const int tsel = a.hetero ? tile : 0;
    const int tile_w = a.hetero ? a.tw[tsel] : a.w, tile_h = a.hetero ? a.th[tsel] : a.h;
    const MgmLattice l = mgm_lattice(q, tile_w, tile_h);
    // a staggered batch: the item that stands for "lattice q of tile t is half way" lets lattice q of tile t + 1 in
    const bool chain_tile = a.stagger >= 0 && tile + 1 < a.ntiles;
    auto push_item = [&](int it_) __attribute__((always_inline)) {
        const uint32_t slot = atomicAdd(a.ctl + 1, 1u);
        __hip_atomic_store(a.ctl + 64 + slot, (uint32_t)it_ + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    if (l.U <= 0 || l.V <= 0 || band * R >= l.V) {                               // an empty lattice: its one item has nothing to do
        if (chain_tile && band == 0 && threadIdx.x == 0) push_item(S2P_MGM_ITEM(tile + 1, q, 0));
        __syncthreads(); continue;
    }
    const int nbq = (l.V + R - 1) / R;
    const bool opens_next_tile = chain_tile && band == min(nbq - 1, (nbq * a.stagger) >> 8);
    const bool has_next = (band + 1) * R < l.V;

    const int w = tile_w, h = tile_h, D = a.D, U = l.U;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), gl = lane & (G - 1);
    const int j = wave * NP + lane / G;                                  // band row of this lane group
    const int v = band * R + j;
#ifdef S2P_MGM_PROBE_DVALID         // timing probe (results invalid): only the lanes of the first S2P_MGM_PROBE_DVALID candidates load and store -- "a smaller range at this pixel pitch"
    const bool lane_ok = gl * DPL < S2P_MGM_PROBE_DVALID;
#else
    const bool lane_ok = PAD ? (gl * DPL < D) : true;
#endif
    const bool is_first = gl == 0, is_last = gl == G - 1;
    const int xb = l.x0 + v * l.xv, yb = l.y0 + v * l.yv;                // pixel of (u, v) = (xb + u xu, yb + u yu)
    // byte offsets in 32-bit unsigned arithmetic: exact for every in-image point (volumes stay below 4 GiB), harmless
    // wrap-around for the lattice points outside the image, which are never dereferenced
    const uint32_t stride = (uint32_t)(l.yu * w + l.xu) * (uint32_t)D;
    const uint32_t base = (uint32_t)(yb * w + xb) * (uint32_t)D + (uint32_t)(gl * DPL);
    const size_t vol_t = a.hetero ? (size_t)a.tvol[tsel] : a.vol;
    const size_t c_at = a.hetero ? (size_t)a.c_off[tsel] * 256u : (size_t)tile * a.c_stride, e_at = a.hetero ? (size_t)a.e_off[tsel] * 256u : (size_t)tile * a.e_stride;
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.C) + c_at, 0, (int)vol_t, S2P_BUF_FLAGS);
    const __amdgpu_buffer_rsrc_t rsE = __builtin_amdgcn_make_buffer_rsrc(a.E + e_at + (size_t)l.r * vol_t, 0, (int)vol_t, S2P_BUF_FLAGS);
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<uint8_t*>(a.rows) + (size_t)tile * a.rows_bytes, 0, (int)a.rows_bytes, S2P_BUF_FLAGS);
    const uint32_t row_bytes = (uint32_t)a.upad * LW * 4u;
    const uint32_t out_row = (uint32_t)(q * 2 + (band & 1)) * row_bytes, in_row = (uint32_t)(q * 2 + ((band + 1) & 1)) * row_bytes;
    uint32_t* const abortw = a.abortw;
    const uint32_t P1pk = pk_dup(a.P1), P2pk = pk_dup(a.P2);
    const bool consumer = wave == 0 && band > 0, producer = wave == NW - 1;
    // tags: the free high bytes of the two 16-bit fields of every dword (messages are <= P2 <= 128) hold ONE 16-bit count,
    // 1 + (band >> 1) <= 2048 -- its low byte above field 0, its high byte above field 1 -- so no two bands of a launch
    // (<= 4095) that use the same slot ever carry the same tag.  (Until round 3 a single byte, 1 + (band >> 1) mod 255,
    // sat above both fields: on a diagonal lattice a slot entry that the bands in between did not rewrite could have
    // passed for fresh 510 bands later -- ADVICE r03; reachable from ~8000-px tiles with 16-row bands.)
    const uint32_t c_out = 1u + ((uint32_t)band >> 1), c_in = 1u + ((uint32_t)(band - 1) >> 1);
    const uint32_t tag_out = ((c_out & 0xffu) << 8) | ((c_out >> 8) << 24);
    const uint32_t tag_in = ((c_in & 0xffu) << 8) | ((c_in >> 8) << 24);
#ifdef S2P_MGM_PROBE_NOPOLL
    bool waiting = false;                                                // (timing probe: nobody waits, the fetcher included)
#else
    bool waiting = true;                                                 // cleared by a timeout: drain without waiting
#endif

    // The points of a lattice row that lie in the image form ONE interval of u (mgm_row_interval).  On the diagonal
    // lattices the image is a diamond, so a band only sweeps the steps between the first and the last of its rows'
    // intervals instead of all U + R - 1.
    int ulo, uspan, plo, pspan;
    mgm_row_interval(l, w, h, v, &ulo, &uspan);
    mgm_row_interval(l, w, h, band * R - 1, &plo, &pspan);               // last row of the previous band (wave-uniform)
    if (wave < NW && gl == 0 && uspan > 0) { atomicMin(&s_range[0], ulo + j); atomicMax(&s_range[1], ulo + uspan + j); }
    __syncthreads();
    int s0 = s_range[0], s1 = s_range[1];                                // steps [s0, s1): row j is at u = T - j
    if (s1 <= s0) { s0 = 0; s1 = 1; }
    s0 &= ~(PF - 1);
    if (threadIdx.x <= NW) s_prog[threadIdx.x] = s0;
    __syncthreads();                                                     // last barrier of the kernel

    // ---- the fetcher wave: previous band's last row, global memory -> chan row 0, point by point ----
    // FP points per 1 KB load (granule g of point u sits at in_row + (u * GPU + g) * 16).  Two groups are in flight;
    // whatever prefix of the oldest group carries the previous band's tag is staged at once (entry (u - 1) & 7, read by
    // wave 0 in step T = u, last read in step u - 8: back-pressure on wave 0's word) and published in s_prog[NW]
    // (= number of points staged); an incomplete group is simply asked for again.
    if (wave == NW) {
      if (band > 0) {
        constexpr int FP = GPU >= 64 ? 1 : (64 / GPU > 4 ? 4 : 64 / GPU), NLF = (GPU + 63) / 64;
        const int sub = NLF == 1 ? lane / GPU : 0, gi0 = NLF == 1 ? lane % GPU : lane;
        const bool active = sub < FP;
        const int Ulim = min(U, s1);
        int* const fprog = &s_prog[NW];
        int seen0 = s0;
        u32x4 qa[NLF], qb[NLF];
        auto request = [&](int grp, u32x4 (&qq)[NLF]) __attribute__((always_inline)) {
            #pragma unroll
            for (int n = 0; n < NLF; n++) {
                const int pu = grp * FP + sub;
                qq[n] = __builtin_amdgcn_raw_buffer_load_b128(rsR, (active && pu < U) ? (int)(in_row + (uint32_t)(pu * GPU + gi0 + n * 64) * 16u) : (int)(S2P_OOB - 32u),
                                                              0, S2P_HANDOFF_LD_AUX);
            }
        };
        // stage group grp out of qq (re-requesting it while incomplete), then reuse qq for group grp + 2
        auto serve = [&](int grp, u32x4 (&qq)[NLF]) __attribute__((always_inline)) {
            const int pu = grp * FP + sub;
            bool need[NLF];
            #pragma unroll
            for (int n = 0; n < NLF; n++) {
                const int gi = gi0 + n * 64;
#ifdef S2P_MGM_PROBE_DVALID
                const bool lok = gi * 8 < S2P_MGM_PROBE_DVALID;
#else
                const bool lok = PAD ? (gi * 8 < D) : true;                 // a granule = 8 candidates (D is a multiple of 16: no granule straddles the end)
#endif
                need[n] = active && lok && pu < U && (uint32_t)(pu - plo) < (uint32_t)pspan;
            }
            int done = 0;
            for (uint32_t it = 0;; ++it) {
                bool bad = false;
                #pragma unroll
                for (int n = 0; n < NLF; n++) {
                    const uint32_t x = ((qq[n].x ^ tag_in) | (qq[n].y ^ tag_in)) | ((qq[n].z ^ tag_in) | (qq[n].w ^ tag_in));
                    bad = bad || (need[n] && (x & 0xff00ff00u) != 0u);
                }
                const unsigned long long bm = __ballot(bad);
                int nv = FP;
                if (bm && waiting) nv = NLF == 1 ? (int)(__builtin_ctzll(bm) / GPU) : 0;
                if (nv > done) {
                    const int needp = grp * FP + nv - RING + (NQ - 2);      // wave 0 finished the last step that reads the entry's previous point
                    if (seen0 < needp) seen0 = mgm_wait_lds(&s_prog[0], needp, abortw, waiting);
                    if (sub >= done && sub < nv && active) {
                        #pragma unroll
                        for (int n = 0; n < NLF; n++) {
                            u32x4 t = qq[n];
                            t.x = need[n] ? (t.x & 0x00ff00ffu) : 0u; t.y = need[n] ? (t.y & 0x00ff00ffu) : 0u;
                            t.z = need[n] ? (t.z & 0x00ff00ffu) : 0u; t.w = need[n] ? (t.w & 0x00ff00ffu) : 0u;
                            *reinterpret_cast<u32x4*>(&chan[((pu + RING - 1) & (RING - 1)) * LW + (gi0 + n * 64) * 4]) = t;
                        }
                    }
                    asm volatile("" ::: "memory");                       // the word follows the data in the wave's DS queue
                    if (lane == 0) __hip_atomic_store(fprog, grp * FP + nv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    asm volatile("" ::: "memory");
                    done = nv;
                }
                if (done >= FP) break;
                if ((it & 63u) == 63u) {
                    if (__hip_atomic_load(abortw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) waiting = false;
                    else if (it > (S2P_MGM_SPIN_LIMIT >> 4)) { __hip_atomic_store(abortw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); waiting = false; }
                }
                __builtin_amdgcn_s_sleep(S2P_MGM_FSLEEP);
                request(grp, qq);
            }
            if ((grp + 2) * FP < Ulim) request(grp + 2, qq);
        };
        // (NQ == 3: the band's first step also reads the point before s0 of the previous band's row -- the third predecessor
        // (u - 1, v - 1) of its row 0 -- so the staging starts one group earlier; s0 is a multiple of PF >= FP)
        int grp = s0 / FP - ((NQ == 3 && s0 >= FP) ? 1 : 0);
        request(grp, qa);
        request(grp + 1, qb);
        for (; grp * FP < Ulim; grp += 2) {
            serve(grp, qa);
            if ((grp + 1) * FP < Ulim) serve(grp + 1, qb);
        }
      }
    } else {

    int up_u = s0 - j;                                                   // u of the next prefetch
    uint32_t up_off = base + (uint32_t)up_u * stride;
    auto prefetch = [&]() __attribute__((always_inline)) -> raw_t {
        const bool in = (uint32_t)(up_u - ulo) < (uint32_t)uspan;
#ifdef S2P_MGM_PROBE_NOMEM          // timing probe (results invalid): the cost loads are issued out of range (same instructions, no bytes)
        const raw_t r = CL::load(rsC, (in && lane_ok && !(S2P_MGM_PROBE_NOMEM & 1)) ? up_off : S2P_OOB);
#else
        const raw_t r = CL::load(rsC, (in && lane_ok) ? up_off : S2P_OOB);
#endif
        up_u++; up_off += stride;
        return r;
    };

#ifdef S2P_MGM_TRACE
    unsigned long long t_gate = wall_clock64(), tr_wait = 0, tr_retries = 0;
    const unsigned long long c_start = __builtin_readcyclecounter(), w_start = wall_clock64();
    bool tr_started = false;
    unsigned long long tw_data = 0, tw_bp = 0, tn_data = 0, tn_bp = 0;
#endif
    uint32_t nb_below = BIGPK, nb_above = BIGPK;                         // G == 16: DPP fill registers (see the step)
    uint32_t msgl[K];                                                    // message of (u - 1, v): none before the row starts
    #pragma unroll
    for (int i = 0; i < K; i++) msgl[i] = 0;
    int u = s0 - j;
    uint32_t off = base + (uint32_t)u * stride;
    int seen_fetch = s0, seen_prev = s0, seen_next = s0;                 // cached progress words of the fetcher and of the neighbouring waves
    uint32_t* const rd_row = &chan[j * RS + gl * K];                     // + entry * LW
    uint32_t* const wr_row = &chan[(j + 1) * RS + gl * K];
    int* const my_prog = &s_prog[wave];

    // one step; I = T & 7 is static in the unrolled sweep, so every LDS address is a lane constant + an immediate
    auto step = [&](raw_t& rawq, const int T, const int I, const bool refill) __attribute__((always_inline)) {
        // -- flow control (wave-uniform; the cached words make these three compares in the steady state.  Folding them
        //    into one compare against a precomputed "safe until" step measured no faster (round 2), and a build WITHOUT the three
        //    tests runs the 8-tile launch slower, 4.5-5.1 against 4.13 ms: waves that drift apart lose the locality their
        //    coupled neighbours give the memory system, which is what bounds the launch (profiles/r06/decompose_probe.txt).
        //    ORDER MATTERS: a band runs nose to tail with the one above it, so what follows the arrival of the data is
        //    on the chain of the whole launch -- the back-pressure word is polled FIRST (while the data is still on
        //    its way), the data last. --
#ifdef S2P_MGM_TRACE
        const unsigned long long tc0 = __builtin_readcyclecounter();
        const bool tcd = (wave > 0 && seen_prev < T) || (consumer && T < U && seen_fetch < T + 1);
        const bool tcb = wave < NW - 1 && seen_next < T - LEAD;
#endif
#ifndef S2P_MGM_PROBE_NOPOLL        // timing probe (results invalid): the step without its three flow-control tests (= 2: without the progress word either)
        if (wave < NW - 1 && seen_next < T - LEAD) seen_next = mgm_wait_lds(&s_prog[wave + 1], T - LEAD, abortw, waiting);   // (entry T & 7 was read LEAD steps ago)
#endif
#ifdef S2P_MGM_TRACE
        const unsigned long long tc1 = __builtin_readcyclecounter();
#endif
#ifndef S2P_MGM_PROBE_NOPOLL
        if (wave > 0 && seen_prev < T) seen_prev = mgm_wait_lds(&s_prog[wave - 1], T, abortw, waiting);               // step T - 1 of the wave above is written
        if (consumer && T < U && seen_fetch < T + 1) seen_fetch = mgm_wait_lds(&s_prog[NW], T + 1, abortw, waiting);  // the point of the previous band's row this step reads is staged
#endif
#ifdef S2P_MGM_TRACE
        if (consumer && !tr_started) { tr_started = true; t_gate = wall_clock64(); }
        if (tcb) { tw_bp += tc1 - tc0; tn_bp++; }
        if (tcd) { tw_data += __builtin_readcyclecounter() - tc1; tn_data++; }
#endif
        asm volatile("" ::: "memory");                                   // the reads below stay behind the waits above
        // message of (u, v - 1): written one step ago by the group of row j - 1 (or staged from the previous band)
        uint32_t mu[K], c[K], nl[K], e[K], msg[K];
        {
            lds_get<K>(rd_row + ((I + RING - 1) & (RING - 1)) * LW, mu);
            if (NQ == 3)                                                 // message of (u - 1, v - 1): the entry of two steps ago; plain dword
                lds_add<K>(rd_row + ((I + RING - 2) & (RING - 1)) * LW, mu);   // adds (fields <= P2: no carry between them)
        }
        // independent work under the LDS latency: this step's costs out of their prefetch register, the next prefetch into it
        __builtin_amdgcn_sched_barrier(0);                               // (keeps the scheduler from hoisting that work above the read)
        const raw_t raw = rawq;
        if (refill) rawq = prefetch();
        const bool sends = ((uint32_t)(u - ulo) < (uint32_t)uspan) && lane_ok;   // a point outside the image sends no message
        CL::unpack(raw, c);
        #pragma unroll
        for (int i = 0; i < K; i++) {
            // mean of the messages, rounded half up: (a + b + 1) >> 1, or (a + b + c + 1) / 3, on both fields (sums <= 3 P2 + 1: no carry)
            const uint32_t m = NQ == 2 ? pk_shr1(msgl[i] + mu[i] + 0x00010001u) : pk_div3(msgl[i] + mu[i] + 0x00010001u);
            nl[i] = pk_add(c[i], m);
            e[i] = pk_sub(P2pk, m);
            if (PAD) nl[i] = lane_ok ? nl[i] : BIGPK;
        }
#ifdef S2P_MGM_PROBE_NOMEM          // timing probe (results invalid): the e-stores are issued out of range
        store_e<K, EAUX>(rsE, (sends && !(S2P_MGM_PROBE_NOMEM & 2)) ? off : S2P_OOB, e);
#else
#ifdef S2P_MGM_PROBE_FULLSTORE       // (with S2P_MGM_PROBE_DVALID: every lane of the pitch stores -- whole lines -- while only the valid ones load)
        store_e<K, EAUX>(rsE, ((uint32_t)(u - ulo) < (uint32_t)uspan) ? off : S2P_OOB, e);
#else
        store_e<K, EAUX>(rsE, sends ? off : S2P_OOB, e);
#endif
#endif
        uint32_t mm = pk_min(pk_min(nl[0], nl[1]), pk_min(nl[2], nl[3]));
        if constexpr (K % 4 == 0) {
            #pragma unroll
            for (int i = 4; i < K; i += 4) mm = pk_min(mm, pk_min(pk_min(nl[i], nl[i + 1]), pk_min(nl[i + 2], nl[i + 3])));
        } else {
            #pragma unroll
            for (int i = 4; i < K; i += 2) mm = pk_min(mm, pk_min(nl[i], nl[i + 1]));
        }
        const int m0 = group_min_i32<G>(min(pk_lo(mm), pk_hi(mm)));
        // G == 16: the edge lanes of a DPP row never receive a shifted value, so a register that starts as MAX_COST and
        // is the `old` operand of every row shift keeps MAX_COST there (as in the path kernel): no per-step refill
        if (G == 16) { nb_below = dpp_mov<DPP_ROW_SHR1>(nl[K - 1], nb_below); nb_above = dpp_mov<DPP_ROW_SHL1>(nl[0], nb_above); }
        const uint32_t below = G == 16 ? nb_below : group_from_below<G>(nl[K - 1], BIGPK, is_first);
        const uint32_t above = G == 16 ? nb_above : group_from_above<G>(nl[0], BIGPK, is_last);
        // a point outside the image sends no message: with the cap lowered from min L + P2 to min L every term of the
        // minimum is >= min L, so the message comes out as 0 without a select per register
        const uint32_t m0pk = pk_dup(m0), delta = sends ? m0pk + P2pk : m0pk;
        #pragma unroll
        for (int i = 0; i < K; i++) {
            const uint32_t dm1 = __builtin_amdgcn_alignbit(nl[i], i ? nl[i - 1] : below, 16);
            const uint32_t dp1 = __builtin_amdgcn_alignbit(i < K - 1 ? nl[i + 1] : above, nl[i], 16);
            const uint32_t t = pk_min(pk_min(pk_add(pk_min(dm1, dp1), P1pk), nl[i]), delta);
            msg[i] = pk_sub(t, m0pk);
            msgl[i] = msg[i];
        }
        lds_put<K>(wr_row + I * LW, msg);                                // row R (the band's last row) lands in a spare LDS row
        asm volatile("" ::: "memory");                                   // the progress word follows the data in the wave's DS queue
#if !defined(S2P_MGM_PROBE_NOPOLL) || S2P_MGM_PROBE_NOPOLL < 2
        if (lane == 0) __hip_atomic_store(my_prog, T + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (lane 0 alone: a store from all 64 lanes -- no exec change -- publishes LATER, launch +14 %)
#endif
        asm volatile("" ::: "memory");
        if (producer) {                                                  // wave-uniform: the wave that holds row R - 1
            // the band's last row also goes to the next band: tagged granules, write-through, no flag
            // (the fetcher checks the tag of every dword of a granule it loads, so a granule may come from two 8-byte stores: K = 6)
            if constexpr (K % 4 == 0) {
                #pragma unroll
                for (int i = 0; i < K; i += 4) {
                    u32x4 t; t.x = msg[i] | tag_out; t.y = msg[i + 1] | tag_out; t.z = msg[i + 2] | tag_out; t.w = msg[i + 3] | tag_out;
                    const uint32_t roff = (j == R - 1 && sends) ? out_row + (uint32_t)((u * LW + gl * K + i) * 4) : S2P_OOB - 32u;
                    __builtin_amdgcn_raw_buffer_store_b128(t, rsR, (int)roff, 0, S2P_HANDOFF_ST_AUX);
                }
            } else {
                #pragma unroll
                for (int i = 0; i < K; i += 2) {
                    u32x2 t; t.x = msg[i] | tag_out; t.y = msg[i + 1] | tag_out;
                    const uint32_t roff = (j == R - 1 && sends) ? out_row + (uint32_t)((u * LW + gl * K + i) * 4) : S2P_OOB - 32u;
                    __builtin_amdgcn_raw_buffer_store_b64(t, rsR, (int)roff, 0, S2P_HANDOFF_ST_AUX);
                }
            }
        }
        u++; off += stride;
    };

    // The compiler's s_waitcnt vmcnt(N) before a step's costs is the number of vector-memory instructions it can PROVE
    // were issued after their load on every path into the loop.  vmcnt counts stores too (gfx9 family) and the sweep
    // issues one e-store per load, but a prologue of PF back-to-back loads proves only the loads: N is 16 + i in step
    // i of the unrolled sweep, i.e. a step waits until all but the last 8-15 steps' loads AND STORES have completed.
    // Pairing each prologue load with a store that the range check drops makes N 32 in every step -- the full prefetch
    // distance, no wait for a recent write acknowledgement -- and measured no difference, alone or with the chip full
    // (round 3, profiles/r03/inner_probe_prologue_stores.txt): the sweep does not wait there.
    raw_t qr[PF];
    #pragma unroll
    for (int i = 0; i < PF; i++) qr[i] = prefetch();
    // The successor band is published once this one is S2P_MGM_TRIG steps into its sweep: a worker takes it, runs its prologue
    // and finds its first input (this band's last row, produced from step R - 1 on) about to arrive.  Publishing it at launch,
    // as a ticket per workgroup did until round 2, parked every band of a lattice on a CU from t = 0 although band k can only
    // start k x (R steps + hand-off) into the launch: on 1024^2 x 128 the resident bands were active 38 % of the time, and with
    // tiles in flight the waiting ones kept the slots the runnable ones needed.
    bool publish = (has_next || opens_next_tile) && wave == 0;
    auto push_next = [&]() __attribute__((always_inline)) {
        if (lane == 0) {
            if (has_next) push_item(S2P_MGM_ITEM(tile, q, band + 1));
            if (opens_next_tile) push_item(S2P_MGM_ITEM(tile + 1, q, 0));
        }
        publish = false;
    };
    int T = s0;
    for (; T + PF <= s1; T += PF) {
        if (publish && T >= s0 + S2P_MGM_TRIG) push_next();
        #pragma unroll
        for (int i = 0; i < PF; i++) step(qr[i], T + i, i & (RING - 1), true);
    }
    if (publish) push_next();
    const int rem = s1 - T;
    #pragma unroll
    for (int i = 0; i < PF - 1; i++)
        if (i < rem) step(qr[i], T + i, i & (RING - 1), false);
#ifdef S2P_MGM_TRACE
    if (lane == 0) {             // per wave: cycles and count of the steps that had to poll for data / for back-pressure, total cycles
        unsigned long long* tr = reinterpret_cast<unsigned long long*>(a.trace) + ((size_t)q * a.nbands + band) * 32;
        if (wave < 8) { tr[8 + wave] = tw_data | (tn_data << 40); tr[16 + wave] = tw_bp | (tn_bp << 40); tr[24 + wave] = __builtin_readcyclecounter() - c_start; }   // cycles | events << 40
    }
    if (threadIdx.x == 0) {      // [s0, s1, t_gate, t_end] per band, behind the control words (tools/mgm_trace.py)
        unsigned long long* tr = reinterpret_cast<unsigned long long*>(a.trace) + ((size_t)q * a.nbands + band) * 32;
        tr[0] = (unsigned long long)s0; tr[1] = (unsigned long long)s1; tr[2] = t_gate; tr[3] = wall_clock64();
        tr[7] = (__builtin_readcyclecounter() - c_start) * 1000ull / (wall_clock64() - w_start + 1);   // shader cycles per 10 us
        tr[4] = tr_wait; tr[5] = tr_retries; tr[6] = (unsigned long long)__builtin_amdgcn_s_getreg(20 << 0 | 0 << 6 | 31 << 11);   // HW_REG_XCC_ID
    }
#endif
    }   // compute waves
    __syncthreads();                                                     // the band is done: its rings may be zeroed for the next item
  }   // worker loop
}

// Workgroups per CU.  The launch is one dependency chain per lattice, and a wave of a chain that shares its SIMD with
// other chains' waves runs slower: FEWER bands per CU are better, even when that leaves bands waiting for a slot (the
// ticket order hands slots to the bands that are needed next).  Bands of a chain in flight at a time ~ sweep length /
// (R + hand-off), 12 chains: measured on 1024 x 1024 x 128 with 4-wave bands, cap 2 -> 0.96 ms, cap 1, 3, 4 or none
// 1.08-1.10; 512 x 512: cap 1 -> 0.44, 2 or more 0.475; 1536 / 2048: cap 2 best by 0-4 % (tools/percu_probe.sh).
// The band shapes this file ships hold 66-135 KB of LDS rings, so at most two fit a CU as they are and the padding below
// is idle (a cap of one was re-measured with them: no difference at any size); it stays as the guard of that property
// should a layout's rings shrink.
static size_t mgm_lds_static(int G, int K, int NW) {
    const int LW = G * K, ring = mgm_ring(LW);
    return (size_t)(NW * (64 / G) + 1) * mgm_row_stride(G, LW) * 4 + 64;
}
static size_t mgm_lds_pad(int G, int K, int NW, int per_cu) {
    if (per_cu <= 0) return 0;
    const size_t stat = mgm_lds_static(G, K, NW), want = (size_t)163840 / (per_cu + 1) + 2048;   // per_cu + 1 of them do not fit
    return want > stat ? (want - stat + 255) & ~(size_t)255 : 0;
}
template <int G, int K, int NQ, int NW>
static bool launch_mgm_bands(hipStream_t st, int nblocks, bool pad, const MgmBandArgs& a, int per_cu) {
    const size_t dyn = mgm_lds_pad(G, K, NW, per_cu);
    {   // per instantiation AND per device: totals beyond 64 KB need the attribute (only reached with S2P_MGM_PER_CU overrides)
        static std::mutex mu;
        static std::map<int, size_t> allowed;
        std::lock_guard<std::mutex> lock(mu);
        int dev = 0;
        hipGetDevice(&dev);
        if (dyn > allowed[dev]) {
            if (hipFuncSetAttribute((const void*)k_mgm_bands<G, K, true, NQ, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) != hipSuccess ||
                hipFuncSetAttribute((const void*)k_mgm_bands<G, K, false, NQ, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) != hipSuccess) return false;
            allowed[dev] = dyn;
        }
    }
    if (pad) hipLaunchKernelGGL((k_mgm_bands<G, K, true, NQ, NW>), dim3(nblocks), dim3(64 * (NW + 1)), dyn, st, a);
    else     hipLaunchKernelGGL((k_mgm_bands<G, K, false, NQ, NW>), dim3(nblocks), dim3(64 * (NW + 1)), dyn, st, a);
    return hipGetLastError() == hipSuccess;
}
#ifdef S2P_MGM_TRACE
int g_mgm_trace_nbands = 0;
uint32_t* g_mgm_trace_ctl = nullptr;
#endif
struct MgmBandPlan { int nbands, upad, items; size_t ctl_bytes, rows_bytes, trace_off; };
// lane layout of the band kernel: the path kernel's.  Both alternatives were built and measured on 1024 x 1024 x 128:
// 16 disparities per lane at D >= 128 (half the bands, 1.5x longer steps) loses, and so does 4 per lane (twice the lanes
// per row, 52 instead of 75 VALU per step: launch 1.13 vs 0.975 ms) -- the step is bound by its fixed part (message
// exchange, progress polls, the reduction's dependency chain), not by its arithmetic.
// 16 disparities per lane (K = 8, G = D / 16: twice the rows per wave, ~30 % fewer instructions per pixel) where it pays.  With the
// chip full the band kernel is bound inside the SIMDs (round 3's timing builds without the cost loads / the e-stores: the 8-tile launch
// hardly moves when half of its memory traffic is removed, and 256 workers run it as fast as 512; 0.95 instructions per SIMD per
// 4 cycles: DESIGN_KERNELS.md 1), on the step's dependent chain: less chain per candidate buys throughput -- VALU work beside the chain
// does not (round 3's unmasked inner blocks) -- while a tile alone is the sum of its steps, where longer steps lose.  Measured (profiles/r03/k8_probe.txt):
//   D = 128: loses both ways (launch 1.03 -> 1.19 ms, 8 tiles 4.35 -> 5.14)           -> K = 4
//   D = 256, 1000^2: 8 tiles per launch 8.89 -> 7.10 ms, three streams 1.37 -> 1.22 ms per tile, one tile alone 1.52 -> 1.60  -> K = 8
//   D = 256, 512^2: one tile alone 0.61 -> 0.80, 8 per launch 0.350 -> 0.326 ms per tile  -> K = 4 below 768 px
//   D = 512: 2.98 -> 2.73 alone, 2.97 -> 2.80 in flight                               -> K = 8
//   128 < D < 256 (round 6, profiles/r06/midrange_probe.txt): the K = 4 layout pads such a range to 32 lanes per pixel -- two rows per wave, up to
//   44 % of the lanes idle --, 16 candidates per lane keep a pixel on ONE DPP row (padded), four rows per wave.  Batches gain (1024^2, 8 tiles
//   per launch: D = 144 / 160 / 192 / 224: 9.4 / 9.2 / 9.4 / 9.9 -> 8.0 / 7.5 / 7.2 / 8.7 ms; 512^2 x 192: 2.44 -> 2.08), a tile alone loses
//   (1024^2 x 192: 1.56 -> 1.65 ms, 512^2: 0.63 -> 0.84: longer steps on its chain)                                  -> K = 8 in batches only
//   D = 32 / 64 in batches (round 6, after the e-stores at D <= 32 went plain; profiles/r06/smalld_shape_probe.txt): K = 8 on 2 / 4 lanes with 2-wave bands (the
//   same rows per band) is bit-exact and slower: 8 tiles of 1024^2 per launch 1.89 -> 2.21 ms, 2.71 -> 3.33                                       -> K = 4
#define S2P_MGM_K8_FROM 256
static LaneLayout mgm_lane_layout(int D, int w, int h, bool batch) {
    LaneLayout ll = lane_layout(D);
    const bool k8 = (D >= S2P_MGM_K8_FROM && D <= 512 && (D > 256 || std::min(w, h) >= 768)) || (batch && D > 128 && D < 256);
    if (k8) { ll.K = 8; ll.G = 8; while (ll.G * 16 < D) ll.G *= 2; ll.pad = ll.G * 16 != D; }
    // D = 192: 12 candidates per lane fill the 16 lanes of a DPP row (16 per lane would leave four of them idle): 8 tiles of 1024^2 per launch
    // 7.15 -> 6.2-6.4 ms, 512^2 2.07 -> 1.78-1.87 (profiles/r06/k6_probe.txt; the same layout at D = 96 / 48 on 8 / 4 lanes loses 5-8 % there)
    // A tile alone: 1024^2 1.57 -> 1.41 ms, 512^2 0.63 -> 0.71 (profiles/r06/depth_probe.txt) -> from 768 px, as for K = 8.
    if (D == 192 && (batch || std::min(w, h) >= 768)) { ll.K = 6; ll.G = 16; ll.pad = false; }
    return ll;
}
// per tile: `items` bands over the `nlat` lattices (an empty lattice counts as one item that does nothing); a batch of
// `ntiles` tiles shares one control block (queue of ntiles * items entries) and has one row ring per tile
static MgmBandPlan mgm_band_plan(int w, int h, int D, int nlat = MGM_LATTICES, int ntiles = 1) {
    const LaneLayout ll = mgm_lane_layout(D, w, h, ntiles > 1);
    const int R = 64 * mgm_waves(ll.G, ll.K, ntiles > 1) / ll.G;
    MgmBandPlan p; p.nbands = 0; p.items = 0;
    int umax = 0;
    const int NL = std::max(nlat, (int)MGM_LATTICES);                    // (4 directions keep the ring shape of 8)
    for (int q = 0; q < NL; q++) {
        const MgmLattice l = mgm_lattice(q, w, h);
        const int nb = (l.U <= 0 || l.V <= 0) ? 0 : (l.V + R - 1) / R;
        if (q < nlat) p.items += std::max(nb, 1);
        p.nbands = std::max(p.nbands, nb);
        umax = std::max(umax, l.U);
    }
    p.upad = (umax + 7) / 8 * 8;
    p.ctl_bytes = align_up(256 + (size_t)p.items * ntiles * 4, 256);
    p.trace_off = p.ctl_bytes;
#ifdef S2P_MGM_TRACE
    p.ctl_bytes += align_up((size_t)NL * p.nbands * 256, 256);
#endif
    p.rows_bytes = align_up((size_t)NL * 2 * p.upad * ll.G * ll.K * 4, 256);
    return p;
}
static size_t mgm_bands_workspace_bytes(int w, int h, int D, int ntiles = 1, int nlat = MGM_LATTICES) {
    const MgmBandPlan p = mgm_band_plan(w, h, D, nlat, ntiles);
    return p.ctl_bytes + p.rows_bytes * ntiles + 512;
}
// Workers (workgroups) of a launch.  Every worker is busy or about to be: a band enters the queue when its input is about
// to exist, so the workers needed = bands of a lattice under way at a time (~ sweep length / (R + hand-off) ~ 20 at
// 1024^2 x 128) x chains.  One tile gets at most 256 (measured: more than its chains can feed; the rest of the chip stays
// free for the launches of other streams / processes), a batch up to two per CU (what fits by LDS).
#define S2P_MGM_WORKERS_1 256
#define S2P_MGM_WORKERS_MAX 512
// the kernel instance of a lane layout (batch: several tiles under one queue)
static bool mgm_launch_for_layout(hipStream_t st, int nblocks, const LaneLayout& ll, const MgmBandArgs& a, int per_cu, int nq, bool batch)
{
    bool ok = false;
    #define S2P_MGM_LAUNCH_NW(GV, KV, NWV) (nq == 3 ? launch_mgm_bands<GV, KV, 3, NWV>(st, nblocks, ll.pad, a, per_cu) : launch_mgm_bands<GV, KV, 2, NWV>(st, nblocks, ll.pad, a, per_cu))
    #define S2P_MGM_LAUNCH(GV, KV) S2P_MGM_LAUNCH_NW(GV, KV, mgm_waves(GV, KV))
    if (ll.K == 6) ok = S2P_MGM_LAUNCH(16, 6);
    else if (ll.K == 8) switch (ll.G) {
        case 16: ok = S2P_MGM_LAUNCH(16, 8); break;
        case 32: ok = S2P_MGM_LAUNCH(32, 8); break;
        default: ok = S2P_MGM_LAUNCH(64, 8); break;
    }
    else switch (ll.G) {
        case 2: ok = S2P_MGM_LAUNCH(2, 4); break;
        case 4: ok = S2P_MGM_LAUNCH(4, 4); break;
        case 8: ok = S2P_MGM_LAUNCH(8, 4); break;
        case 16: ok = batch ? S2P_MGM_LAUNCH_NW(16, 4, mgm_waves(16, 4, true)) : S2P_MGM_LAUNCH(16, 4); break;
        case 32: ok = S2P_MGM_LAUNCH(32, 4); break;
        default: ok = S2P_MGM_LAUNCH(64, 4); break;
    }
    #undef S2P_MGM_LAUNCH
    #undef S2P_MGM_LAUNCH_NW
    return ok;
}

// false on a bad size (*abortw != 0 after the launch = a hand-off wait timed out).  ntiles > 1: a batch -- tile t has its
// cost volume at C + t * c_stride, its 8 e-volumes at E + t * e_stride, all of shape [h][w][D]; `ws` holds
// mgm_bands_workspace_bytes(w, h, D, ntiles).
static bool enqueue_mgm_bands(hipStream_t st, const uint8_t* C, uint8_t* E, int w, int h, int D, int P1, int P2, void* ws, uint32_t* abortw,
                              int nlat = MGM_LATTICES, int per_cu = 0, int ntiles = 1, size_t c_stride = 0, size_t e_stride = 0, int nq = 2,
                              int stagger = -1)
{
    if (per_cu == 0) per_cu = 2;                                         // see mgm_lds_pad
    if (const char* e = getenv("S2P_MGM_PER_CU")) per_cu = atoi(e);     // (probe: 0 = no cap)
    const MgmBandPlan p = mgm_band_plan(w, h, D, nlat, ntiles);
    if (p.rows_bytes >= ((size_t)1 << 31) || p.nbands <= 0 || p.nbands > 4095 || ntiles < 1 || ntiles > S2P_MGM_TILES_MAX || nlat > 64) return false;
    MgmBandArgs a;
    a.hetero = 0;
    a.C = C; a.E = E; a.vol = (size_t)w * h * D; a.w = w; a.h = h; a.D = D; a.P1 = P1; a.P2 = P2;
    a.nbands = p.nbands; a.nlat = nlat; a.upad = p.upad; a.ctl = (uint32_t*)ws; a.rows = (uint32_t*)((char*)ws + p.ctl_bytes);
    a.rows_bytes = (uint32_t)p.rows_bytes; a.abortw = abortw;
    if (const char* e = getenv("S2P_MGM_STAGGER")) stagger = atoi(e);   // (probe)
    if (ntiles == 1) stagger = -1;
    a.stagger = stagger;
    a.total = p.items * ntiles; a.ninit = stagger >= 0 ? nlat : nlat * ntiles; a.ntiles = ntiles; a.c_stride = c_stride; a.e_stride = e_stride;
    a.trace = (uint32_t*)((char*)ws + p.trace_off);
    hipMemsetAsync(ws, 0, p.ctl_bytes + p.rows_bytes * ntiles, st);      // the queue and every tag: every call
    const LaneLayout ll = mgm_lane_layout(D, w, h, ntiles > 1);
    int workers = ntiles == 1 ? S2P_MGM_WORKERS_1 : std::min(S2P_MGM_WORKERS_MAX, S2P_MGM_WORKERS_1 * ntiles);
    if (const char* e = getenv("S2P_MGM_WORKERS")) workers = atoi(e);   // (probe)
    const int nblocks = std::max(1, std::min(a.total, workers));
    const bool ok = mgm_launch_for_layout(st, nblocks, ll, a, per_cu, nq, ntiles > 1);
#ifdef S2P_MGM_TRACE
    g_mgm_trace_nbands = p.nbands; g_mgm_trace_ctl = a.trace;
#endif
    return ok;
}


// A batch of tiles of different sizes under one queue (round 4): one lane layout -- chosen for D and the smallest tile side --, per-tile
// geometry and volume offsets in the kernel arguments, hand-off rings of one (maximal) shape.  C / E: base pointers; c_off / e_off: byte
// offsets of tile t's volumes (multiples of 256), each tile's volumes [h_t][w_t][D].  Returns false on a bad size.
static size_t mgm_bands_hetero_workspace_bytes(int n, const int* w, const int* h, int D, int nlat = MGM_LATTICES) {
    int wmin = 1 << 30, hmin = 1 << 30;
    for (int t = 0; t < n; t++) { wmin = std::min(wmin, w[t]); hmin = std::min(hmin, h[t]); }
    const LaneLayout ll = mgm_lane_layout(D, wmin, hmin, n > 1);
    const int R = 64 * mgm_waves(ll.G, ll.K, n > 1) / ll.G;
    int items = 0, umax = 0;
    const int NL = std::max(nlat, (int)MGM_LATTICES);
    for (int t = 0; t < n; t++)
        for (int q = 0; q < NL; q++) {
            const MgmLattice l = mgm_lattice(q, w[t], h[t]);
            items += std::max((l.U <= 0 || l.V <= 0) ? 0 : (l.V + R - 1) / R, 1);
            umax = std::max(umax, l.U);
        }
    const int upad = (umax + 7) / 8 * 8;
    return align_up(256 + (size_t)items * 4, 256) + (size_t)n * align_up((size_t)NL * 2 * upad * ll.G * ll.K * 4, 256) + 512;
}
static bool enqueue_mgm_bands_hetero(hipStream_t st, const uint8_t* C, uint8_t* E, int n, const int* w, const int* h, int D, int P1, int P2,
                                     const size_t* c_off, const size_t* e_off, void* ws, uint32_t* abortw, int nlat, int nq)
{
#ifdef S2P_MGM_TRACE
    return false;
#endif
    if (n < 1 || n > S2P_MGM_HETERO_MAX) return false;
    int wmin = 1 << 30, hmin = 1 << 30;
    for (int t = 0; t < n; t++) { wmin = std::min(wmin, w[t]); hmin = std::min(hmin, h[t]); }
    const LaneLayout ll = mgm_lane_layout(D, wmin, hmin, n > 1);
    const int R = 64 * mgm_waves(ll.G, ll.K, n > 1) / ll.G;
    MgmBandArgs a;
    memset(&a, 0, sizeof(a));
    int items = 0, umax = 0, nbands = 0;
    const int NL = std::max(nlat, (int)MGM_LATTICES);
    if (nlat > 64) return false;
    for (int t = 0; t < n; t++) {
        for (int q = 0; q < NL; q++) {
            const MgmLattice l = mgm_lattice(q, w[t], h[t]);
            const int nb = (l.U <= 0 || l.V <= 0) ? 0 : (l.V + R - 1) / R;
            if (q < nlat) items += std::max(nb, 1);
            nbands = std::max(nbands, nb);
            umax = std::max(umax, l.U);
        }
        const size_t vol = (size_t)w[t] * h[t] * D;
        if (vol >= ((size_t)1 << 32) - 65536 || (c_off[t] & 255) || (e_off[t] & 255) || (c_off[t] >> 40) || (e_off[t] >> 40)) return false;
        a.tw[t] = w[t]; a.th[t] = h[t]; a.tvol[t] = (uint32_t)vol; a.c_off[t] = (uint32_t)(c_off[t] >> 8); a.e_off[t] = (uint32_t)(e_off[t] >> 8);
    }
    const int upad = (umax + 7) / 8 * 8;
    const size_t ctl_bytes = align_up(256 + (size_t)items * 4, 256);
    const size_t rows_bytes = align_up((size_t)NL * 2 * upad * ll.G * ll.K * 4, 256);
    if (rows_bytes >= ((size_t)1 << 31) || nbands <= 0 || nbands > 4095) return false;
    a.C = C; a.E = E; a.D = D; a.P1 = P1; a.P2 = P2; a.hetero = 1;
    a.nbands = nbands; a.nlat = nlat; a.upad = upad; a.ctl = (uint32_t*)ws; a.rows = (uint32_t*)((char*)ws + ctl_bytes);
    a.rows_bytes = (uint32_t)rows_bytes; a.abortw = abortw;
    a.stagger = -1; a.total = items; a.ninit = nlat * n; a.ntiles = n;
    a.trace = (uint32_t*)((char*)ws + ctl_bytes);
    hipMemsetAsync(ws, 0, ctl_bytes + rows_bytes * n, st);
    int workers = n == 1 ? S2P_MGM_WORKERS_1 : std::min(S2P_MGM_WORKERS_MAX, S2P_MGM_WORKERS_1 * n);
    if (const char* e = getenv("S2P_MGM_WORKERS")) workers = atoi(e);   // (probe)
    return mgm_launch_for_layout(st, std::max(1, std::min(a.total, workers)), ll, a, 2, nq, n > 1);
}

}  // namespace s2p
