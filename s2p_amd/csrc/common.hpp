// s2p_amd/csrc/common.hpp -- shared host/device helpers for libs2p_hip.so (gfx950 / CDNA4 only).
#pragma once
#include "probe_guard.hpp"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <vector>
#include <map>

#include "../../include/s2p_hip.h"

namespace s2p {

// ---------------------------------------------------------------------------------------------
// error handling
// ---------------------------------------------------------------------------------------------
void set_last_error(const char* fmt, ...);

#define S2P_HIP_CHECK(expr)                                                                      \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            s2p::set_last_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return S2P_HIP_RUNTIME_ERROR;                                                        \
        }                                                                                        \
    } while (0)

// ---------------------------------------------------------------------------------------------
// geometry of the sgbm driver (3rdparty/sgbm/sgbm.cpp:166-207, stereosgbm.cpp:328-339, :122)
// ---------------------------------------------------------------------------------------------
struct Geom {
    int w, h;            // tile
    int Wc, x0;          // canvas width, paste offset ("crop trick")
    int minD, maxD, D;   // OpenCV-convention disparity range [minD, maxD), D % 16 == 0
    int minX1, maxX1, width1;
    int minX2, maxX2, width2;
    int invalid;         // INVALID_DISP_SCALED = (minD-1)*16
    // flat row scratch emulation (see k_prefilter)
    int guard;           // zero bytes in front of the reference's tempBuf image
    int fl;              // bytes per (row, channel) of the flat image, incl. guard and zero tail
};
int make_geom(int w, int h, int dmin, int dmax, Geom* g);

// radix-select state of the quantiser (device memory)
struct SelectState {
    uint32_t n;            // non-NaN count
    uint32_t rank[2];      // remaining rank inside the current prefix bucket
    uint32_t prefix[2];    // key bits fixed so far (left aligned)
    float rminmax[2];
};

// device buffers of one sgbm call (carved from the context workspace)
struct SgbmBuffers {
    SelectState* st; uint32_t* hist;
    uint8_t *uu1, *uu2;
    uint32_t *vpk, *upk;       // per-row packed BT operands (k_prefilter)
    int16_t* C; uint8_t* E; int16_t* S;
    int16_t *disp_raw, *cost_raw, *disp_med, *disp_fin;
    int *lab, *cnt, *par;      // speckle CCL: run start per pixel, component size, union-find parent
};

// device buffers of one census call
struct CensusBuffers {
    uint32_t *cen1, *cen2;
    uint8_t* C; uint8_t* E; uint16_t* S;
    float *disp_raw, *disp_med;
    int16_t* q16;
    int *lab, *cnt, *par;
    int dmin0, D0;             // the range the volumes of the call's finest level are laid out for (stage dumps)
};

// ---------------------------------------------------------------------------------------------
// context: device, stream, grow-only workspace, optional per-stage event timing
// ---------------------------------------------------------------------------------------------
struct StageTiming { double ms = 0; int launches = 0; };
#define S2P_ROW_LDS_MAX (156 * 1024)      // dynamic LDS a row-state kernel may ask for (160 KiB per CU, minus its static words)

}  // namespace s2p

struct s2p_hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // CU partitioning (round 6; s2p_hip_ctx_create reads S2P_HIP_CU_BAND / S2P_HIP_CU_ROWS): the band-pipelined MGM launches go to
    // `band_stream`, a stream confined to one CU mask, the row kernels stay on `stream` (confined to the other); band_fork /
    // band_join order the two with one event each.  nullptr = one stream, the whole device (the shipped default).
    hipStream_t band_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // grow-only workspace (bump allocated per call)
    char* ws = nullptr;
    size_t ws_size = 0, ws_used = 0;
    // hipGraph replay of the *_dev pipelines (opt-in: s2p_hip_ctx_use_graphs); key = call signature
    bool use_graphs = false;
    struct Graph { hipGraphExec_t exec; bool mgm_check; };   // + whether its replay runs band-pipelined MGM launches
    std::map<std::string, Graph> graphs;
    uint32_t* mgm_abort = nullptr; // device word raised by a band-pipelined MGM launch whose hand-off wait timed out
    bool mgm_check = false;        // such a launch was enqueued since the last check (host entry points / ctx_sync read the word)
    // timing
    bool timing = false;
    std::vector<std::pair<std::string, std::pair<hipEvent_t, hipEvent_t>>> pending;
    std::map<std::string, s2p::StageTiming> stages;
    std::vector<hipEvent_t> event_pool;
};

namespace s2p {

int ws_reserve(s2p_hip_ctx* ctx, size_t bytes);          // ensure capacity (may sync + realloc)
void* ws_alloc(s2p_hip_ctx* ctx, size_t bytes);          // bump; nullptr if over capacity
inline void ws_reset(s2p_hip_ctx* ctx) { ctx->ws_used = 0; }
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct StageScope {   // RAII: brackets a stage with hipEvents on the ctx stream when timing is on
    s2p_hip_ctx* ctx; const char* name; hipEvent_t e0 = nullptr, e1 = nullptr;
    StageScope(s2p_hip_ctx* c, const char* n);
    ~StageScope();
};
int timing_collect(s2p_hip_ctx* ctx);   // sync + fold pending events into ctx->stages

// the stream a band-pipelined MGM launch goes to, ordered after everything enqueued on ctx->stream so far / ctx->stream made to wait for it.
// (A failing event call never costs the ordering: the launch then goes to ctx->stream itself, or the host waits for the band stream.)
inline hipStream_t band_fork(s2p_hip_ctx* ctx) {
    if (!ctx->band_stream) return ctx->stream;
    if (hipEventRecord(ctx->ev_fork, ctx->stream) != hipSuccess || hipStreamWaitEvent(ctx->band_stream, ctx->ev_fork, 0) != hipSuccess) {
        (void)hipGetLastError();
        return ctx->stream;
    }
    return ctx->band_stream;
}
inline void band_join(s2p_hip_ctx* ctx) {
    if (!ctx->band_stream) return;
    if (hipEventRecord(ctx->ev_join, ctx->band_stream) != hipSuccess || hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipStreamSynchronize(ctx->band_stream);
    }
}

// ---------------------------------------------------------------------------------------------
// device helpers: packed int16 math and DPP lane exchange (wave64)
// ---------------------------------------------------------------------------------------------
#ifdef __HIPCC__
typedef short s16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b) {
    s16x2 r = __builtin_elementwise_min(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b));
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) {
    s16x2 r = __builtin_bit_cast(s16x2, a) + __builtin_bit_cast(s16x2, b);
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) {
    s16x2 r = __builtin_bit_cast(s16x2, a) - __builtin_bit_cast(s16x2, b);
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ int pk_lo(uint32_t a) { return (int)(short)(a & 0xffffu); }
__device__ __forceinline__ int pk_hi(uint32_t a) { return (int)(short)(a >> 16); }
__device__ __forceinline__ uint32_t pk_dup(int v) { return ((uint32_t)v & 0xffffu) * 0x10001u; }

// DPP controls (GFX9 encoding)
enum : int {
    DPP_QUAD_XOR1 = 0xB1,        // quad_perm:[1,0,3,2]
    DPP_QUAD_XOR2 = 0x4E,        // quad_perm:[2,3,0,1]
    DPP_ROW_SHL1 = 0x101, DPP_ROW_SHR1 = 0x111,
    DPP_WAVE_SHL1 = 0x130, DPP_WAVE_SHR1 = 0x138,
    DPP_ROW_MIRROR = 0x140, DPP_ROW_HALF_MIRROR = 0x141
};
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t v, uint32_t fill) {   // out-of-range lanes get `fill`
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, CTRL, 0xf, 0xf, false);
}
template <int CTRL>
__device__ __forceinline__ int dpp_perm(int v) {                            // permutations: every lane valid
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}

// value of the lane one position below / above inside a group of G lanes; `fill` at the group edge.
template <int G>
__device__ __forceinline__ uint32_t group_from_below(uint32_t v, uint32_t fill, bool is_first) {
    if (G == 16) return dpp_mov<DPP_ROW_SHR1>(v, fill);          // DPP rows are exactly the groups
    uint32_t r = dpp_mov<DPP_WAVE_SHR1>(v, fill);
    return (G == 64) ? r : (is_first ? fill : r);
}
template <int G>
__device__ __forceinline__ uint32_t group_from_above(uint32_t v, uint32_t fill, bool is_last) {
    if (G == 16) return dpp_mov<DPP_ROW_SHL1>(v, fill);
    uint32_t r = dpp_mov<DPP_WAVE_SHL1>(v, fill);
    return (G == 64) ? r : (is_last ? fill : r);
}

// lane l <-> lane l ^ 16 / l ^ 32 exchanges of the xor butterfly: gfx950's row-swap instructions (v_permlane16_swap
// exchanges the odd 16-lane rows of one register with the even rows of another, v_permlane32_swap the halves) applied
// to two copies of the value give both partners in VGPRs -- 3 VALU instead of a ds_bpermute round trip through the LDS
// crossbar on the dependency chain of every step.
typedef uint32_t u32pair __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32pair swap16(uint32_t t) { return __builtin_amdgcn_permlane16_swap(t, t, false, false); }
__device__ __forceinline__ u32pair swap32(uint32_t t) { return __builtin_amdgcn_permlane32_swap(t, t, false, false); }

// all-reduce (signed min / unsigned min) over aligned groups of G lanes (xor butterfly)
template <int G>
__device__ __forceinline__ int group_min_i32(int t) {
    if (G >= 2) t = min(t, dpp_perm<DPP_QUAD_XOR1>(t));
    if (G >= 4) t = min(t, dpp_perm<DPP_QUAD_XOR2>(t));
    if (G >= 8) t = min(t, dpp_perm<DPP_ROW_HALF_MIRROR>(t));
    if (G >= 16) t = min(t, dpp_perm<DPP_ROW_MIRROR>(t));
    if (G >= 32) { const u32pair r = swap16((uint32_t)t); t = min((int)r.x, (int)r.y); }
    if (G >= 64) { const u32pair r = swap32((uint32_t)t); t = min((int)r.x, (int)r.y); }
    return t;
}
template <int G>
__device__ __forceinline__ uint32_t group_min_u32(uint32_t t) {
    if (G >= 2) t = min(t, (uint32_t)dpp_perm<DPP_QUAD_XOR1>((int)t));
    if (G >= 4) t = min(t, (uint32_t)dpp_perm<DPP_QUAD_XOR2>((int)t));
    if (G >= 8) t = min(t, (uint32_t)dpp_perm<DPP_ROW_HALF_MIRROR>((int)t));
    if (G >= 16) t = min(t, (uint32_t)dpp_perm<DPP_ROW_MIRROR>((int)t));
    if (G >= 32) { const u32pair r = swap16(t); t = min(r.x, r.y); }
    if (G >= 64) { const u32pair r = swap32(t); t = min(r.x, r.y); }
    return t;
}
template <int G>
__device__ __forceinline__ int group_or_i32(int t) {
    if (G >= 2) t |= dpp_perm<DPP_QUAD_XOR1>(t);
    if (G >= 4) t |= dpp_perm<DPP_QUAD_XOR2>(t);
    if (G >= 8) t |= dpp_perm<DPP_ROW_HALF_MIRROR>(t);
    if (G >= 16) t |= dpp_perm<DPP_ROW_MIRROR>(t);
    if (G >= 32) { const u32pair r = swap16((uint32_t)t); t = (int)(r.x | r.y); }
    if (G >= 64) { const u32pair r = swap32((uint32_t)t); t = (int)(r.x | r.y); }
    return t;
}
#endif  // __HIPCC__

}  // namespace s2p
