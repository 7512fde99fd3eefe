// s2p_amd/csrc/api.hip -- C ABI of libs2p_hip.so (declared in include/s2p_hip.h).
#include "common.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cerrno>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <unistd.h>
#include <fcntl.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <map>
#include <string>
#include <mutex>

namespace s2p {

static thread_local char g_err[512] = "";
#ifdef S2P_PROBE_BUILD
#define S2P_ERR_BANNER "[PROBE BUILD] "
#else
#define S2P_ERR_BANNER ""
#endif
void set_last_error(const char* fmt, ...) {
    const size_t b = sizeof(S2P_ERR_BANNER) - 1;
    memcpy(g_err, S2P_ERR_BANNER, b);
    va_list ap; va_start(ap, fmt); vsnprintf(g_err + b, sizeof(g_err) - b, fmt, ap); va_end(ap);
}

// ---- geometry (sgbm.cpp:166-207; stereosgbm.cpp:122,328-339) -----------------------------------
int make_geom(int w, int h, int dmin, int dmax, Geom* g) {
    int maxdisp = -dmin, mindisp = -dmax;                 // sign flip to the OpenCV convention
    if (mindisp >= maxdisp) return S2P_HIP_EMPTY_RANGE;   // sgbm.cpp:174-177
    int ndisp = (int)(16 * std::ceil((maxdisp - mindisp) / 16.0));
    g->w = w; g->h = h;
    g->x0 = std::max(maxdisp, 0);
    g->Wc = w + std::max(-mindisp, 0) + std::max(maxdisp, 0);
    g->minD = mindisp; g->maxD = mindisp + ndisp; g->D = ndisp;
    g->minX1 = std::max(-g->maxD, 0); g->maxX1 = g->Wc + std::min(g->minD, 0);
    g->width1 = g->maxX1 - g->minX1;
    g->minX2 = std::max(g->minX1 - g->maxD, 0); g->maxX2 = std::min(g->maxX1 - g->minD, g->Wc);
    g->width2 = g->maxX2 - g->minX2;
    g->invalid = (g->minD - 1) * 16;
    g->guard = (int)align_up((size_t)std::max(g->minX2, 0) + 16, 16);
    g->fl = (int)align_up((size_t)g->guard + 2 * (size_t)std::max(g->width2, 0) + 4 * (size_t)g->Wc + std::max(g->maxD, 0) + 64, 16);
    return S2P_HIP_OK;
}

// ---- workspace ---------------------------------------------------------------------------------
static void drop_graphs(s2p_hip_ctx* ctx) {
    for (auto& kv : ctx->graphs) hipGraphExecDestroy(kv.second.exec);
    ctx->graphs.clear();
}

int ws_reserve(s2p_hip_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->ws_size) return S2P_HIP_OK;
    S2P_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    drop_graphs(ctx);                                   // captured graphs hold pointers into the old workspace
    if (ctx->ws) { S2P_HIP_CHECK(hipFree(ctx->ws)); ctx->ws = nullptr; ctx->ws_size = 0; }
    size_t want = align_up(bytes + bytes / 8, (size_t)1 << 20);
    S2P_HIP_CHECK(hipMalloc((void**)&ctx->ws, want));
    ctx->ws_size = want;
    return S2P_HIP_OK;
}
void* ws_alloc(s2p_hip_ctx* ctx, size_t bytes) {
    size_t off = align_up(ctx->ws_used, 256);
    if (off + bytes > ctx->ws_size) { set_last_error("workspace overflow (%zu + %zu > %zu)", off, bytes, ctx->ws_size); return nullptr; }
    ctx->ws_used = off + bytes;
    return ctx->ws + off;
}

// ---- timing ------------------------------------------------------------------------------------
static hipEvent_t get_event(s2p_hip_ctx* ctx) {
    if (!ctx->event_pool.empty()) { hipEvent_t e = ctx->event_pool.back(); ctx->event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    hipEventCreate(&e);
    return e;
}
StageScope::StageScope(s2p_hip_ctx* c, const char* n) : ctx(c), name(n) {
    if (!ctx->timing) return;
    e0 = get_event(ctx); e1 = get_event(ctx);
    hipEventRecord(e0, ctx->stream);
}
StageScope::~StageScope() {
    if (!ctx->timing) return;
    hipEventRecord(e1, ctx->stream);
    ctx->pending.push_back({name, {e0, e1}});
}
int timing_collect(s2p_hip_ctx* ctx) {
    S2P_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (auto& p : ctx->pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.second.first, p.second.second) == hipSuccess) {
            auto& s = ctx->stages[p.first]; s.ms += ms; s.launches += 1;
        }
        ctx->event_pool.push_back(p.second.first); ctx->event_pool.push_back(p.second.second);
    }
    ctx->pending.clear();
    return S2P_HIP_OK;
}

// implemented in sgbm_kernels.hip
int sgbm_enqueue(s2p_hip_ctx* ctx, const Geom& g, const s2p_sgbm_params& p,
                 const float* d_im1, const float* d_im2, float* d_disp, float* d_cost, uint8_t* d_mask,
                 bool want_S, SgbmBuffers* out);
size_t sgbm_workspace_bytes(const Geom& g, bool want_S);
int sgbm_read_rminmax(s2p_hip_ctx* ctx, const SgbmBuffers& b, float out[2]);

// implemented in census_kernels.hip
int census_enqueue(s2p_hip_ctx* ctx, const s2p_census_params& p, const float* d_im1, const float* d_im2,
                   int w, int h, int dmin, int dmax, float* d_disp, float* d_conf, uint8_t* d_mask,
                   bool want_S, CensusBuffers* out);
size_t census_workspace_bytes(const s2p_census_params& p, int w, int h, int dmin, int dmax, bool want_S);
int census_batch_enqueue(s2p_hip_ctx* ctx, const s2p_census_params& p, int n, const float* const* d_im1, const float* const* d_im2,
                         int w, int h, int dmin, int dmax, float* const* d_disp, float* const* d_conf, uint8_t* const* d_mask);
size_t census_batch_workspace_bytes(const s2p_census_params& p, int n, int w, int h, int dmin, int dmax);
size_t census_batch_hetero_workspace_bytes(const s2p_census_params& p, int n, const int* w, const int* h, const int* dmin, const int* dmax);
int census_batch_hetero_enqueue(s2p_hip_ctx* ctx, const s2p_census_params& p, int n, const float* const* d_im1, const float* const* d_im2,
                                const int* w, const int* h, const int* dmin, const int* dmax,
                                float* const* d_disp, float* const* d_conf, uint8_t* const* d_mask);
bool census_batches_hetero(const s2p_census_params& p, int n, const int* w, const int* h);
int census_D(const s2p_census_params& p, int dmin, int dmax, bool dumps = false);
int census_levels(int w, int h, int scales);
int erode_enqueue(s2p_hip_ctx* ctx, const uint8_t* d_msk, int w, int h, int radius, uint8_t* d_out);
int rejection_mask_enqueue(s2p_hip_ctx* ctx, const float* d_disp, const float* d_im1, const float* d_im2, int w, int h, uint8_t* d_mask);

// implemented in warp_kernels.hip
size_t warp_workspace_bytes(int sw, int sh);
int warp_enqueue(s2p_hip_ctx* ctx, const void* d_src, int dtype, int sw, int sh, const double H[9],
                 float* d_dst, int w, int h, char* scratch);

// implemented in fusion_kernels.hip
int height_transfer_enqueue(s2p_hip_ctx* ctx, const double* d_hm, int wr, int hr, const double H[9], int w, int h,
                            double* d_val, uint8_t* d_flag, double* d_out);
// implemented in raster_kernels.hip
int raster_enqueue(s2p_hip_ctx* ctx, const double* d_pts, int npts, int nb, double xoff, double yoff, double res,
                   int xsize, int ysize, int radius, float sigma, float* d_raster);
size_t raster_workspace_bytes(int npts, int nb, int xsize, int ysize, int radius);
size_t raster_disc_cells(int radius);
int merge_enqueue(s2p_hip_ctx* ctx, const float* d_stack, const double* d_offsets, int n, size_t npx, int op,
                  double threshold, double mean_offset, float* d_out);
int cargarse_basura_enqueue(s2p_hip_ctx* ctx, const float* d_in, int w, int h, float* d_out, int* lab, int* par, int* cnt);

// implemented in tri_kernels.hip
int tri_enqueue(s2p_hip_ctx* ctx, const float* d_dispx, const float* d_dispy, const float* d_msk, int nx, int ny,
                const float* d_msk_orig, int w, int h, const double ha[9], const double hb[9], const s2p_rpc* d_rpc,
                const float bbox[4], double* d_lonlatalt, float* d_err);
int height_map_localize_enqueue(s2p_hip_ctx* ctx, const s2p_rpc* d_rpc, const float* d_hm, int w, int h, int off_x, int off_y, double* d_lonlatalt);
int corresp_enqueue(s2p_hip_ctx* ctx, const float* d_kpa, const float* d_kpb, int n, const s2p_rpc* d_rpc, double* d_lonlatalt, float* d_err);
int count3d_enqueue(s2p_hip_ctx* ctx, const double* d_xyz, int nx, int ny, float r, int p, int* d_count);
int remove_isolated_enqueue(s2p_hip_ctx* ctx, double* d_xyz, int nx, int ny, float r, int p, int n, int q,
                            int* d_count, uint8_t* d_rej, int* d_flag);

static int check_census_params(const s2p_census_params& p, int w, int h, int dmin, int dmax) {
    if (dmax < dmin) { set_last_error("census: empty disparity range [%d, %d]", dmin, dmax); return S2P_HIP_EMPTY_RANGE; }
    if (!(p.subpix == 0 || p.subpix == 1 || p.subpix == 2)) { set_last_error("census: subpix %d not implemented (1 or 2)", p.subpix); return S2P_HIP_UNSUPPORTED; }
    if (p.scales < 0 || p.scales > 16) { set_last_error("census: scales %d out of range (0..16)", p.scales); return S2P_HIP_BAD_ARGUMENT; }
    const int sp = p.subpix == 2 ? 2 : 1;
    const int D = census_D(p, dmin, dmax);
    if ((double)w * h * D >= 4294967296.0 - 65536.0) {
        set_last_error("census: cost volume exceeds 4 GiB (32-bit buffer offsets); use smaller tiles");
        return S2P_HIP_UNSUPPORTED;
    }
    if (!(p.census_win == 3 || p.census_win == 5)) { set_last_error("census: window %d not implemented (3 or 5)", p.census_win); return S2P_HIP_UNSUPPORTED; }
    if (p.nb_dir != 8 && p.nb_dir != 4 && p.nb_dir != 16) { set_last_error("census: 4, 8 or 16 directions are implemented (got %d)", p.nb_dir); return S2P_HIP_UNSUPPORTED; }
    if (p.nb_dir == 16 && p.recursion < 1) { set_last_error("census: 16 directions run with the MGM recursion (recursion = 1 or 2), not as 1-D paths"); return S2P_HIP_UNSUPPORTED; }
    if (!(p.P1 > 0 && p.P2 > p.P1 && p.P2 <= 128)) { set_last_error("census: need 0 < P1 < P2 <= 128 (got %d, %d)", p.P1, p.P2); return S2P_HIP_UNSUPPORTED; }
    if (p.mindiff > 16383) { set_last_error("census: MINDIFF %d out of range (<= 0 disabled, up to 16383 units of the summed cost)", p.mindiff); return S2P_HIP_BAD_ARGUMENT; }
    if (p.cost != 0 && p.cost != 1) { set_last_error("census: cost %d unknown (0 = census, 1 = zncc)", p.cost); return S2P_HIP_BAD_ARGUMENT; }
    // ZNCC: the window rows of image 1, image 2 (and, with half-pixel candidates, of image 2 sampled half way between its columns) + the
    // per-pixel window variances of a row live in LDS (zncc_cost_lds of census_kernels.hip)
    if (p.cost == 1 && (size_t)(1 + sp) * p.census_win * (w + 2 * (p.census_win / 2)) * 4 + (size_t)sp * w * 4 > S2P_ROW_LDS_MAX) {
        set_last_error("census: tile too wide (%d px) for the ZNCC cost kernel's row windows", w); return S2P_HIP_UNSUPPORTED;
    }
    if (p.recursion < 0 || p.recursion > 2) { set_last_error("census: recursion %d unknown (0 = SGM paths, 1 = MGM with two predecessors, 2 = with three)", p.recursion); return S2P_HIP_BAD_ARGUMENT; }
    if (p.recursion == 2 && p.P2 > 127) { set_last_error("census: the three-predecessor recursion needs P2 <= 127 (packed 16-bit mean of three messages; got %d)", p.P2); return S2P_HIP_UNSUPPORTED; }
    if (sp * (dmax - dmin) + 1 > 1024) { set_last_error("census: %d disparity candidates > 1024 not implemented", sp * (dmax - dmin) + 1); return S2P_HIP_UNSUPPORTED; }
    // one image row of per-pixel state lives in LDS (64 KiB launches): the WTA kernel keeps the right-view competition and the
    // left winners, (4 sp + 6) w + 4 D + 16 bytes; the cost kernel the two signature rows, (4 + 4 sp) w + 8 D
    if (std::max((size_t)w * (4 * sp + 6) + (size_t)D * 4 + 16, (size_t)w * (4 + 4 * sp) + (size_t)D * 8) > S2P_ROW_LDS_MAX) {
        set_last_error("census: tile too wide (%d px) for the per-row LDS state; use tiles up to ~%d px wide", w, sp == 2 ? 11000 : 15000);
        return S2P_HIP_UNSUPPORTED;
    }
    return S2P_HIP_OK;
}

static int check_params(const s2p_sgbm_params& p, const Geom& g) {
    if (p.win != 3) { set_last_error("sgbm: only SADWindowSize == 3 is implemented (got %d)", p.win); return S2P_HIP_UNSUPPORTED; }
    if (!(p.P1 > 0 && p.P2 > p.P1 && p.P2 <= 255)) { set_last_error("sgbm: need 0 < P1 < P2 <= 255 (got %d, %d)", p.P1, p.P2); return S2P_HIP_UNSUPPORTED; }
    if (g.D > 1024) { set_last_error("sgbm: disparity range %d > 1024 not implemented", g.D); return S2P_HIP_UNSUPPORTED; }
    if ((size_t)g.Wc * 8 > 64 * 1024) { set_last_error("sgbm: canvas too wide (%d px) for the per-row LDS state; use tiles up to ~8000 px wide", g.Wc); return S2P_HIP_UNSUPPORTED; }
    if (g.width1 == 1) {   // the reference's 3-column block sum reads past its one-column cost row (stereosgbm.cpp:447-451): undefined
        set_last_error("sgbm: degenerate geometry (1 usable column for range [%d, %d] on width %d)", -g.maxD, -g.minD, g.w);
        return S2P_HIP_UNSUPPORTED;
    }
    if ((double)g.h * std::max(g.width1, 0) * g.D * 2.0 >= 4294967296.0 - 65536.0) {
        set_last_error("sgbm: cost volume %dx%dx%d exceeds 4 GiB (32-bit buffer offsets); use smaller tiles", g.width1, g.h, g.D);
        return S2P_HIP_UNSUPPORTED;
    }
    if ((size_t)g.fl * 2 > 150 * 1024) { set_last_error("sgbm: canvas too wide for the per-row LDS scratch (%d)", g.Wc); return S2P_HIP_UNSUPPORTED; }
    if (p.uniqueness_ratio > 100 || p.speckle_range < 0) { set_last_error("sgbm: bad uniqueness/speckle parameters"); return S2P_HIP_UNSUPPORTED; }
    return S2P_HIP_OK;
}

static double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// wait for the stream, honouring a deadline (absolute seconds; < 0 = none)
static int wait_stream_raw(s2p_hip_ctx* ctx, double deadline);
// after the stream has drained: did the band-pipelined MGM launch of this call give up on a hand-off?
static int check_mgm(s2p_hip_ctx* ctx) {
    if (!ctx->mgm_check) return S2P_HIP_OK;
    ctx->mgm_check = false;
    uint32_t ab = 0;
    S2P_HIP_CHECK(hipMemcpy(&ab, ctx->mgm_abort, 4, hipMemcpyDeviceToHost));
    if (ab) S2P_HIP_CHECK(hipMemset(ctx->mgm_abort, 0, 4));   // the word is only ever cleared here, after the host has seen it
#ifdef S2P_MGM_TRACE
    {
        extern int g_mgm_trace_nbands;
        extern uint32_t* g_mgm_trace_ctl;
        const int nb = g_mgm_trace_nbands;
        std::vector<unsigned long long> tr((size_t)12 * nb * 32);
        hipMemcpy(tr.data(), g_mgm_trace_ctl, tr.size() * 8, hipMemcpyDeviceToHost);
        for (int q = 0; q < 12; q++) for (int b = 0; b < nb; b++) {
            const unsigned long long* t = &tr[((size_t)q * nb + b) * 32];
            if (t[3]) {
                fprintf(stderr, "MGMTRACE %d %d %llu %llu %llu %llu %llu %llu %llu %llu\n", q, b, t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7]);
                fprintf(stderr, "MGMWAVES %d %d", q, b);
                for (int i = 8; i < 28; i++) fprintf(stderr, " %llu", t[i]);
                fprintf(stderr, "\n");
            }
        }
        fprintf(stderr, "MGMTRACE_END\n");
    }
#endif
    if (ab) { set_last_error("census: MGM band hand-off timed out (results invalid)"); return S2P_HIP_RUNTIME_ERROR; }
    return S2P_HIP_OK;
}
static int wait_stream(s2p_hip_ctx* ctx, double deadline) {
    const int rc = wait_stream_raw(ctx, deadline);
    const int rm = check_mgm(ctx);
    return rc ? rc : rm;
}
static int wait_stream_raw(s2p_hip_ctx* ctx, double deadline) {
    if (deadline < 0) { S2P_HIP_CHECK(hipStreamSynchronize(ctx->stream)); return S2P_HIP_OK; }
    // With a deadline the stream is polled, not spun on: the orchestrator runs one worker process per core
    // (s2p/parallel.py:76-98), and a worker that burns its core while the GPU works starves the others.  The sleeps grow
    // with the time already waited (a tile takes ~1 ms; 20 us steps add a few percent at most).  The enqueued kernels
    // cannot be cancelled and write into the caller's buffers, so after the deadline the stream is still drained -- the
    // call then reports S2P_HIP_TIMEOUT.
    const double t0 = now_s();
    bool late = false;
    for (;;) {
        hipError_t e = hipStreamQuery(ctx->stream);
        if (e == hipSuccess) return late ? S2P_HIP_TIMEOUT : S2P_HIP_OK;
        if (e != hipErrorNotReady) { set_last_error("hipStreamQuery: %s", hipGetErrorString(e)); return S2P_HIP_RUNTIME_ERROR; }
        const double t = now_s();
        if (t > deadline && !late) { late = true; set_last_error("deadline exceeded (the enqueued kernels are left to drain)"); }
        const double waited = t - t0;
        usleep(waited < 2e-3 ? 20 : waited < 50e-3 ? 200 : 1000);
    }
}

// Run `enqueue` directly, or (graphs on, timing off) capture it once per signature and replay it.
template <typename F>
static int run_or_replay(s2p_hip_ctx* ctx, const std::string& key, size_t ws_bytes, F enqueue) {
    if (!ctx->use_graphs || ctx->timing) return enqueue();
    int rc = ws_reserve(ctx, ws_bytes);               // no allocation / synchronisation may happen inside a capture
    if (rc) return rc;
    auto it = ctx->graphs.find(key);
    if (it == ctx->graphs.end()) {
        if (ctx->graphs.size() >= 32) drop_graphs(ctx);
        hipGraph_t graph = nullptr;
        const bool pending = ctx->mgm_check;           // of earlier calls nobody synchronised on yet
        ctx->mgm_check = false;
        S2P_HIP_CHECK(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
        rc = enqueue();
        hipError_t e = hipStreamEndCapture(ctx->stream, &graph);
        const bool captured_mgm = ctx->mgm_check;
        ctx->mgm_check = pending;
        if (rc) { if (graph) hipGraphDestroy(graph); return rc; }
        if (e != hipSuccess) { set_last_error("hipStreamEndCapture: %s", hipGetErrorString(e)); return S2P_HIP_RUNTIME_ERROR; }
        hipGraphExec_t exec = nullptr;
        e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        hipGraphDestroy(graph);
        if (e != hipSuccess) { set_last_error("hipGraphInstantiate: %s", hipGetErrorString(e)); return S2P_HIP_RUNTIME_ERROR; }
        it = ctx->graphs.emplace(key, s2p_hip_ctx::Graph{exec, captured_mgm}).first;
    }
    ctx->mgm_check |= it->second.mgm_check;            // every replay re-arms the hand-off timeout check of s2p_hip_ctx_sync; a check
                                                       // still pending from an earlier, unsynchronised call is never dropped
    S2P_HIP_CHECK(hipGraphLaunch(it->second.exec, ctx->stream));
    return S2P_HIP_OK;
}

template <typename P>
static std::string call_key(const char* kind, const P& p, int w, int h, int dmin, int dmax, const void* a, const void* b,
                            const void* c, const void* d, const void* e) {
    std::string k(kind);
    k.append(reinterpret_cast<const char*>(&p), sizeof(P));
    const int dims[4] = {w, h, dmin, dmax};
    k.append(reinterpret_cast<const char*>(dims), sizeof(dims));
    const void* ptrs[5] = {a, b, c, d, e};
    k.append(reinterpret_cast<const char*>(ptrs), sizeof(ptrs));
    const char* impl = getenv("S2P_MGM_IMPL");         // selects the kernels that were captured
    k.append(impl ? impl : "");
    return k;
}

static int sgbm_host_impl(s2p_hip_ctx* ctx, const float* im1, const float* im2, int w, int h, int dmin, int dmax,
                          const s2p_sgbm_params* params, float* disp, float* cost, uint8_t* mask,
                          double timeout_s, s2p_hip_sgbm_dump* dump)
{
    if (!ctx || !im1 || !im2 || !disp || w <= 0 || h <= 0) { set_last_error("bad argument"); return S2P_HIP_BAD_ARGUMENT; }
    const double deadline = timeout_s < 0 ? -1.0 : now_s() + timeout_s;
    if (timeout_s == 0) { set_last_error("timeout of 0 s: nothing was enqueued"); return S2P_HIP_TIMEOUT; }
    s2p_sgbm_params p;
    if (params) p = *params; else s2p_hip_sgbm_default_params(&p);
    Geom g;
    int rc = make_geom(w, h, dmin, dmax, &g);
    if (rc) { set_last_error("sgbm: empty disparity range [%d, %d]", dmin, dmax); return rc; }
    rc = check_params(p, g);
    if (rc) return rc;
    S2P_HIP_CHECK(hipSetDevice(ctx->device));

    const size_t npx = (size_t)w * h;
    // user I/O lives in a separate device allocation from the bump workspace
    const size_t io_bytes = align_up(npx * 4, 256) * 4 + align_up(npx, 256);
    const bool want_S = dump && dump->S;
    rc = ws_reserve(ctx, sgbm_workspace_bytes(g, want_S) + io_bytes + 4096);
    if (rc) return rc;
    char* io = ctx->ws + ctx->ws_size - io_bytes;      // carve I/O from the top of the workspace
    float* d_im1 = (float*)io;
    float* d_im2 = (float*)(io + align_up(npx * 4, 256));
    float* d_disp = (float*)(io + 2 * align_up(npx * 4, 256));
    float* d_cost = (float*)(io + 3 * align_up(npx * 4, 256));
    uint8_t* d_mask = (uint8_t*)(io + 4 * align_up(npx * 4, 256));

    S2P_HIP_CHECK(hipMemcpyAsync(d_im1, im1, npx * 4, hipMemcpyHostToDevice, ctx->stream));
    S2P_HIP_CHECK(hipMemcpyAsync(d_im2, im2, npx * 4, hipMemcpyHostToDevice, ctx->stream));
    SgbmBuffers b;
    rc = sgbm_enqueue(ctx, g, p, d_im1, d_im2, d_disp, d_cost, d_mask, want_S, &b);
    if (rc) return rc;
    S2P_HIP_CHECK(hipMemcpyAsync(disp, d_disp, npx * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (cost) S2P_HIP_CHECK(hipMemcpyAsync(cost, d_cost, npx * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (mask) S2P_HIP_CHECK(hipMemcpyAsync(mask, d_mask, npx, hipMemcpyDeviceToHost, ctx->stream));
    if (dump) {
        dump->geom[0] = g.Wc; dump->geom[1] = g.width1; dump->geom[2] = g.D; dump->geom[3] = g.minD;
        dump->geom[4] = g.x0; dump->geom[5] = g.minX1; dump->geom[6] = g.maxX1; dump->geom[7] = g.invalid;
        {
            const size_t vol = g.width1 > 0 ? (size_t)h * g.width1 * g.D : 0, ncan = (size_t)g.Wc * h;
            // canvases -> cropped q1/q2
            if (dump->q1) S2P_HIP_CHECK(hipMemcpy2DAsync(dump->q1, w, b.uu1 + g.x0, g.Wc, w, h, hipMemcpyDeviceToHost, ctx->stream));
            if (dump->q2) S2P_HIP_CHECK(hipMemcpy2DAsync(dump->q2, w, b.uu2 + g.x0, g.Wc, w, h, hipMemcpyDeviceToHost, ctx->stream));
            if (dump->C && vol) S2P_HIP_CHECK(hipMemcpyAsync(dump->C, b.C, vol * 2, hipMemcpyDeviceToHost, ctx->stream));
            if (dump->S && vol) S2P_HIP_CHECK(hipMemcpyAsync(dump->S, b.S, vol * 2, hipMemcpyDeviceToHost, ctx->stream));
            if (dump->disp_raw) S2P_HIP_CHECK(hipMemcpyAsync(dump->disp_raw, b.disp_raw, ncan * 2, hipMemcpyDeviceToHost, ctx->stream));
            if (dump->cost_raw) S2P_HIP_CHECK(hipMemcpyAsync(dump->cost_raw, b.cost_raw, ncan * 2, hipMemcpyDeviceToHost, ctx->stream));
            if (dump->disp_med) S2P_HIP_CHECK(hipMemcpyAsync(dump->disp_med, b.disp_med, ncan * 2, hipMemcpyDeviceToHost, ctx->stream));
            if (dump->disp_fin) S2P_HIP_CHECK(hipMemcpyAsync(dump->disp_fin, b.disp_fin, ncan * 2, hipMemcpyDeviceToHost, ctx->stream));
            rc = sgbm_read_rminmax(ctx, b, dump->rminmax);
            if (rc) return rc;
        }
    }
    return wait_stream(ctx, deadline);
}

static int census_host_impl(s2p_hip_ctx* ctx, const float* im1, const float* im2, int w, int h, int dmin, int dmax,
                            const s2p_census_params* params, float* disp, float* conf, uint8_t* mask,
                            double timeout_s, s2p_hip_census_dump* dump)
{
    if (!ctx || !im1 || !im2 || !disp || w <= 0 || h <= 0) { set_last_error("bad argument"); return S2P_HIP_BAD_ARGUMENT; }
    const double deadline = timeout_s < 0 ? -1.0 : now_s() + timeout_s;
    if (timeout_s == 0) { set_last_error("timeout of 0 s: nothing was enqueued"); return S2P_HIP_TIMEOUT; }
    s2p_census_params p;
    if (params) p = *params; else s2p_hip_census_default_params(&p);
    int rc = check_census_params(p, w, h, dmin, dmax);
    if (rc) return rc;
    S2P_HIP_CHECK(hipSetDevice(ctx->device));
    const int D = census_D(p, dmin, dmax);
    const size_t npx = (size_t)w * h, a4 = align_up(npx * 4, 256);
    const size_t io_bytes = a4 * 4 + align_up(npx, 256);
    const bool want_S = dump && dump->S;
    rc = ws_reserve(ctx, census_workspace_bytes(p, w, h, dmin, dmax, want_S) + io_bytes + 4096);
    if (rc) return rc;
    char* io = ctx->ws + ctx->ws_size - io_bytes;
    float* d_im1 = (float*)io; float* d_im2 = (float*)(io + a4);
    float* d_disp = (float*)(io + 2 * a4); float* d_conf = (float*)(io + 3 * a4);
    uint8_t* d_mask = (uint8_t*)(io + 4 * a4);
    S2P_HIP_CHECK(hipMemcpyAsync(d_im1, im1, npx * 4, hipMemcpyHostToDevice, ctx->stream));
    S2P_HIP_CHECK(hipMemcpyAsync(d_im2, im2, npx * 4, hipMemcpyHostToDevice, ctx->stream));
    CensusBuffers b;
    rc = census_enqueue(ctx, p, d_im1, d_im2, w, h, dmin, dmax, d_disp, conf ? d_conf : nullptr, d_mask, want_S, dump ? &b : nullptr);
    if (rc) return rc;
    S2P_HIP_CHECK(hipMemcpyAsync(disp, d_disp, npx * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (conf) S2P_HIP_CHECK(hipMemcpyAsync(conf, d_conf, npx * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (mask) S2P_HIP_CHECK(hipMemcpyAsync(mask, d_mask, npx, hipMemcpyDeviceToHost, ctx->stream));
    if (dump) {
        const size_t vol = npx * (size_t)b.D0;         // a multi-scale call lays its finest level out for the narrowed range: D0 <= D
        dump->dmin0 = b.dmin0; dump->D0 = b.D0;
        if (dump->C) S2P_HIP_CHECK(hipMemcpyAsync(dump->C, b.C, vol, hipMemcpyDeviceToHost, ctx->stream));
        if (dump->S) S2P_HIP_CHECK(hipMemcpyAsync(dump->S, b.S, vol * 2, hipMemcpyDeviceToHost, ctx->stream));
        if (dump->disp_raw) S2P_HIP_CHECK(hipMemcpyAsync(dump->disp_raw, b.disp_raw, npx * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (dump->disp_med) S2P_HIP_CHECK(hipMemcpyAsync(dump->disp_med, b.disp_med, npx * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    return wait_stream(ctx, deadline);
}

}  // namespace s2p

using namespace s2p;

extern "C" {

const char* s2p_hip_last_error(void) { return g_err; }
#ifdef S2P_PROBE_BUILD
extern "C" __attribute__((visibility("default"))) const char s2p_hip_probe_build_marker[] = "PROBE BUILD [" S2P_PROBE_BUILD "]";
const char* s2p_hip_build_info(void) { return "libs2p_hip gfx950 (hipcc " __VERSION__ ") PROBE BUILD [" S2P_PROBE_BUILD "]: measurement switches are on, results may be invalid"; }
#else
const char* s2p_hip_build_info(void) { return "libs2p_hip gfx950 (hipcc " __VERSION__ ")"; }
#endif

// The HIP runtime does not survive fork(): a child of a process that already initialised it hangs on its first HIP call.
// The library remembers which process first touched the runtime and refuses, loudly, in any other one that inherited
// that state (the orchestrator forks its workers BEFORE any of them touches the GPU: s2p/parallel.py:76-98).
static int g_hip_pid = 0;
static bool hip_usable_here() {
    const int pid = (int)getpid();
    int seen = __atomic_load_n(&g_hip_pid, __ATOMIC_ACQUIRE);
    if (seen == 0) {
        int expected = 0;
        __atomic_compare_exchange_n(&g_hip_pid, &expected, pid, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE);
        seen = __atomic_load_n(&g_hip_pid, __ATOMIC_ACQUIRE);
    }
    if (seen != pid) {
        set_last_error("the HIP runtime was initialised in process %d before this process (%d) was forked from it: "
                       "use the GPU only after the fork (no context, no device query in the parent)", seen, pid);
        return false;
    }
    return true;
}

int s2p_hip_device_count(void) {
    if (!hip_usable_here()) return -1;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---- the process fence of a device --------------------------------------------------------------------------
// k_mgm_bands is a persistent-worker kernel whose workgroups wait for each other inside one launch; that is sound within a process
// (a band only ever waits for a workgroup that already runs) but once more processes drive a device than it has hardware queues to
// give them (~8), the runtime time-slices whole queues: launches of 16 direct-mode Pool workers then stretched 4 -> 47 ms per call,
// and about one such Pool in twelve lost its results (profiles/r04/pool_direct_sweep_run2...json; round 5 found the cause on the
// Python side: a worker's HipError -- a bounded hand-off wait timing out under that time-slicing -- could not be unpickled in the
// parent, s2p_amd/_lib.py HipError.__reduce__).  The time-outs themselves remain possible, so the library REFUSES instead of
// time-slicing: every process takes one of
// S2P_HIP_MAX_PROCS_PER_DEVICE (default 8; 0 = no fence) advisory slots per physical device (keyed by its PCI bus id) at its first
// context on that device -- a flock()ed file under /dev/shm, released by the kernel when the process ends however it ends -- and a
// process that finds none gets S2P_HIP_UNSUPPORTED with the way out in the message: the device's broker (s2p_amd/broker.py), which
// serves any number of Pool workers through ONE process.  The contract of s2p/parallel.py:100-105 is kept: a worker that cannot
// run surfaces as an exception in r.get(), not as a silent time-out.
static std::mutex g_slot_mutex;
struct DeviceSlot { int fd; int contexts; };
static std::map<std::string, DeviceSlot> g_slot;     // device key -> the descriptor whose lock this process holds while it has a context on that device
static int g_slot_pid = 0;
static std::string device_key(int device) {
    char bus[64];
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess) snprintf(bus, sizeof bus, "dev%d", device);
    for (char* c = bus; *c; ++c) if (*c == ':' || *c == '/') *c = '_';
    return bus;
}
// The slot directory lives under a world-writable place (/dev/shm or /tmp): it is only used when it is a real directory (not a
// symlink) that belongs to this user with mode 0700, and the slot files are opened relative to it with O_NOFOLLOW and used only
// when they are regular files of this user -- nothing another local user planted there is ever followed, truncated or written
// (ADVICE r05).  Anything else: no fence (it is advisory).
static int acquire_device_slot(int device) {
    int maxp = 8;
    if (const char* e = getenv("S2P_HIP_MAX_PROCS_PER_DEVICE")) maxp = atoi(e);
    if (maxp <= 0) return S2P_HIP_OK;
    const std::string bus = device_key(device);
    std::lock_guard<std::mutex> lock(g_slot_mutex);
    if (g_slot_pid != (int)getpid()) { g_slot.clear(); g_slot_pid = (int)getpid(); }      // (a forked child shares its parent's descriptors, not its slots)
    auto held = g_slot.find(bus);
    if (held != g_slot.end()) { held->second.contexts++; return S2P_HIP_OK; }
    const char* base = getenv("S2P_HIP_SLOT_DIR");
    struct stat st;
    if (!base) base = (stat("/dev/shm", &st) == 0 && S_ISDIR(st.st_mode)) ? "/dev/shm" : "/tmp";
    char dir[512];
    snprintf(dir, sizeof dir, "%s/s2p_hip_slots_%d", base, (int)getuid());
    if (mkdir(dir, 0700) != 0 && errno != EEXIST) return S2P_HIP_OK;      // no place for the slots: no fence
    const int dfd = open(dir, O_RDONLY | O_DIRECTORY | O_NOFOLLOW | O_CLOEXEC);
    if (dfd < 0) return S2P_HIP_OK;
    if (fstat(dfd, &st) != 0 || !S_ISDIR(st.st_mode) || st.st_uid != getuid() || (st.st_mode & 0777) != 0700) { close(dfd); return S2P_HIP_OK; }
    bool any = false;
    for (int i = 0; i < maxp; ++i) {
        char name[128];
        snprintf(name, sizeof name, "%s.%d", bus.c_str(), i);
        const int fd = openat(dfd, name, O_CREAT | O_RDWR | O_CLOEXEC | O_NOFOLLOW, 0600);
        if (fd < 0) continue;
        struct stat fs;
        if (fstat(fd, &fs) != 0 || !S_ISREG(fs.st_mode) || fs.st_uid != getuid()) { close(fd); continue; }
        any = true;
        if (flock(fd, LOCK_EX | LOCK_NB) == 0) {
            g_slot[bus] = DeviceSlot{fd, 1};
            close(dfd);
            return S2P_HIP_OK;
        }
        close(fd);
    }
    close(dfd);
    if (!any) return S2P_HIP_OK;
    set_last_error("%d processes already drive device %d (%s; S2P_HIP_MAX_PROCS_PER_DEVICE = %d): hand the tiles to the device's broker instead "
                   "(S2P_HIP_BROKER=1, the default of the file-level mirrors; s2p_amd/broker.py) -- beyond that many processes the runtime "
                   "time-slices their queues and a launch whose workgroups wait for each other is no longer bounded in wall time", maxp, device, bus.c_str(), maxp);
    return S2P_HIP_UNSUPPORTED;
}
// the process's last context on the device is gone: the slot is free for another process (closing the descriptor drops the lock)
static void release_device_slot(int device) {
    const std::string bus = device_key(device);
    std::lock_guard<std::mutex> lock(g_slot_mutex);
    if (g_slot_pid != (int)getpid()) return;
    auto held = g_slot.find(bus);
    if (held == g_slot.end()) return;
    if (--held->second.contexts <= 0) { close(held->second.fd); g_slot.erase(held); }
}

// a stream confined to the `lo` lowest and the `hi` highest bits of the device's CU mask
static hipError_t create_masked_stream(int device, int lo, int hi, hipStream_t* out) {
    int ncu = 0;
    hipError_t e = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device);
    if (e != hipSuccess) return e;
    if (ncu <= 0 || lo + hi <= 0) return hipErrorInvalidValue;
    std::vector<uint32_t> m((size_t)(ncu + 31) / 32, 0u);
    for (int i = 0; i < lo && i < ncu; i++) m[i / 32] |= 1u << (i % 32);
    for (int i = 0; i < hi && i < ncu; i++) { const int b = ncu - 1 - i; m[b / 32] |= 1u << (b % 32); }
    return hipExtStreamCreateWithCUMask(out, (uint32_t)m.size(), m.data());
}

int s2p_hip_ctx_create(int device, void* stream, s2p_hip_ctx** out) {
    if (!out) return S2P_HIP_BAD_ARGUMENT;
    *out = nullptr;
    if (!hip_usable_here()) return S2P_HIP_RUNTIME_ERROR;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) { set_last_error("no HIP device visible (%s)", hipGetErrorString(e)); return S2P_HIP_RUNTIME_ERROR; }
    if (device < 0 || device >= n) { set_last_error("device %d out of range (%d visible)", device, n); return S2P_HIP_BAD_ARGUMENT; }
    S2P_HIP_CHECK(hipSetDevice(device));
    if (int rc = acquire_device_slot(device)) return rc;
    s2p_hip_ctx* c = new s2p_hip_ctx();
    c->device = device;
    // CU partitioning (measured in profiles/r06/cumask_sweep.txt, off by default): S2P_HIP_CU_BAND = n confines the
    // band-pipelined MGM launches of this context to the n lowest bits of the device's CU mask (a second stream), S2P_HIP_CU_ROWS = m the
    // row kernels (cost, WTA, median, epilogue) to the m highest.  The driver deals mask bits round-robin over the 8 XCDs
    // (tools/probes/cumask_map.hip prints the mapping), so multiples of 8 are XCD-balanced.
    int cu_band = 0, cu_rows = 0;
    if (const char* e = getenv("S2P_HIP_CU_BAND")) cu_band = atoi(e);
    if (const char* e = getenv("S2P_HIP_CU_ROWS")) cu_rows = atoi(e);
    if (stream) { c->stream = (hipStream_t)stream; c->own_stream = false; cu_rows = 0; }
    else {
        hipError_t es = cu_rows > 0 ? create_masked_stream(device, 0, cu_rows, &c->stream) : hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (es != hipSuccess) { set_last_error("hipStreamCreate: %s", hipGetErrorString(es)); release_device_slot(device); delete c; return S2P_HIP_RUNTIME_ERROR; }
        c->own_stream = true;
    }
    if (cu_band <= 0 && cu_rows > 0) cu_band = 1 << 20;          // rows confined, bands everywhere: a second, unmasked stream
    if (cu_band > 0) {
        if (create_masked_stream(device, cu_band, 0, &c->band_stream) != hipSuccess || hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) {
            set_last_error("S2P_HIP_CU_BAND=%d: could not create the CU-masked stream", cu_band);
            if (c->band_stream) hipStreamDestroy(c->band_stream);
            if (c->own_stream) hipStreamDestroy(c->stream);
            release_device_slot(device); delete c; return S2P_HIP_RUNTIME_ERROR;
        }
    }
    if (hipMalloc((void**)&c->mgm_abort, 256) != hipSuccess) { set_last_error("hipMalloc failed"); s2p_hip_ctx_destroy(c); return S2P_HIP_RUNTIME_ERROR; }
    if (hipMemset(c->mgm_abort, 0, 256) != hipSuccess) { set_last_error("hipMemset failed"); s2p_hip_ctx_destroy(c); return S2P_HIP_RUNTIME_ERROR; }
    *out = c;
    return S2P_HIP_OK;
}

int s2p_hip_pinned_alloc(size_t bytes, void** out) {
    if (!out || bytes == 0) { set_last_error("pinned_alloc: bad argument"); return S2P_HIP_BAD_ARGUMENT; }
    *out = nullptr;
    if (!hip_usable_here()) return S2P_HIP_RUNTIME_ERROR;
    S2P_HIP_CHECK(hipHostMalloc(out, bytes, hipHostMallocPortable));
    return S2P_HIP_OK;
}
void s2p_hip_pinned_free(void* p) {
    if (p && (int)getpid() == __atomic_load_n(&g_hip_pid, __ATOMIC_ACQUIRE)) hipHostFree(p);   // a forked child must not touch the parent's runtime state
}

int s2p_hip_host_register(void* p, size_t bytes) {
    if (!p || bytes == 0) { set_last_error("host_register: bad argument"); return S2P_HIP_BAD_ARGUMENT; }
    if (!hip_usable_here()) return S2P_HIP_RUNTIME_ERROR;
    S2P_HIP_CHECK(hipHostRegister(p, bytes, hipHostRegisterPortable));
    return S2P_HIP_OK;
}
void s2p_hip_host_unregister(void* p) {
    if (p && (int)getpid() == __atomic_load_n(&g_hip_pid, __ATOMIC_ACQUIRE)) hipHostUnregister(p);
}

int s2p_hip_ctx_use_graphs(s2p_hip_ctx* ctx, int on) {
    if (!ctx) return S2P_HIP_BAD_ARGUMENT;
    ctx->use_graphs = on != 0;
    if (!on) { S2P_HIP_CHECK(hipStreamSynchronize(ctx->stream)); drop_graphs(ctx); }
    return S2P_HIP_OK;
}

void s2p_hip_ctx_destroy(s2p_hip_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    drop_graphs(c);
    for (auto& p : c->pending) { hipEventDestroy(p.second.first); hipEventDestroy(p.second.second); }
    for (auto e : c->event_pool) hipEventDestroy(e);
    if (c->ws) hipFree(c->ws);
    if (c->mgm_abort) hipFree(c->mgm_abort);
    if (c->band_stream) { hipStreamDestroy(c->band_stream); hipEventDestroy(c->ev_fork); hipEventDestroy(c->ev_join); }
    if (c->own_stream) hipStreamDestroy(c->stream);
    release_device_slot(c->device);
    delete c;
}

int s2p_hip_ctx_sync(s2p_hip_ctx* c) {
    if (!c) return S2P_HIP_BAD_ARGUMENT;
    S2P_HIP_CHECK(hipStreamSynchronize(c->stream));
    return check_mgm(c);        // the device entries are asynchronous: this is where a hand-off timeout surfaces
}

void s2p_hip_sgbm_default_params(s2p_sgbm_params* p) {
    if (!p) return;
    p->win = 3; p->P1 = 8; p->P2 = 32; p->lr = 1;            // s2p/block_matching.py:121-126
    p->prefilter_cap = 63; p->uniqueness_ratio = 10;           // sgbm.cpp:189-190
    p->speckle_window = 50; p->speckle_range = 1;              // sgbm.cpp:191-192
}

int s2p_hip_sgbm_geometry(int w, int dmin, int dmax, int geom[8]) {
    Geom g;
    int rc = make_geom(w, 1, dmin, dmax, &g);
    if (rc) return rc;
    geom[0] = g.Wc; geom[1] = g.width1; geom[2] = g.D; geom[3] = g.minD;
    geom[4] = g.x0; geom[5] = g.minX1; geom[6] = g.maxX1; geom[7] = g.invalid;
    return S2P_HIP_OK;
}

int s2p_hip_sgbm_host(s2p_hip_ctx* ctx, const float* im1, const float* im2, int w, int h, int dmin, int dmax,
                      const s2p_sgbm_params* params, float* disp, float* cost, uint8_t* mask, double timeout_s) {
    return sgbm_host_impl(ctx, im1, im2, w, h, dmin, dmax, params, disp, cost, mask, timeout_s, nullptr);
}

int s2p_hip_sgbm_debug(s2p_hip_ctx* ctx, const float* im1, const float* im2, int w, int h, int dmin, int dmax,
                       const s2p_sgbm_params* params, float* disp, float* cost, uint8_t* mask, s2p_hip_sgbm_dump* dump) {
    return sgbm_host_impl(ctx, im1, im2, w, h, dmin, dmax, params, disp, cost, mask, -1.0, dump);
}

int s2p_hip_sgbm_dev(s2p_hip_ctx* ctx, const float* d_im1, const float* d_im2, int w, int h, int dmin, int dmax,
                     const s2p_sgbm_params* params, float* d_disp, float* d_cost, uint8_t* d_mask) {
    if (!ctx || !d_im1 || !d_im2 || !d_disp || w <= 0 || h <= 0) { set_last_error("bad argument"); return S2P_HIP_BAD_ARGUMENT; }
    s2p_sgbm_params p;
    if (params) p = *params; else s2p_hip_sgbm_default_params(&p);
    Geom g;
    int rc = make_geom(w, h, dmin, dmax, &g);
    if (rc) { set_last_error("sgbm: empty disparity range [%d, %d]", dmin, dmax); return rc; }
    rc = check_params(p, g);
    if (rc) return rc;
    S2P_HIP_CHECK(hipSetDevice(ctx->device));
    return run_or_replay(ctx, call_key("sgbm", p, w, h, dmin, dmax, d_im1, d_im2, d_disp, d_cost, d_mask),
                         sgbm_workspace_bytes(g, false),
                         [&]() { return sgbm_enqueue(ctx, g, p, d_im1, d_im2, d_disp, d_cost, d_mask, false, nullptr); });
}

void s2p_hip_census_default_params(s2p_census_params* p) {
    if (!p) return;
    p->census_win = 5; p->P1 = 8; p->P2 = 32; p->nb_dir = 8;     // s2p/config.py:139,149; mgm defaults
    p->lr_check = 1; p->lr_tau = 1.0f; p->mindiff = -1;          // s2p/config.py:153-160
    p->median = 1; p->remove_small_cc = 0;                       // 'mgm' branch (block_matching.py:156)
    p->fix_overcount = 1;                                        // mgm's TSGM_FIX_OVERCOUNT default (see oracle/census_oracle.c)
    p->recursion = 2;                                            // what the 'mgm' call site runs (TSGM=3 as modelled: three predecessors), the mode that
                                                                 // meets the parity bar; 1 = two predecessors; 0 = 8 independent path sets (preview mode)
    p->scales = 1; p->subpix = 1;                                // single scale, whole-pixel candidates ('mgm'); mgm_multi: -S 6, SUBPIX=2
    p->cost = 0;                                                 // census / Hamming (`-t census`)
}

int s2p_hip_census_sgm_host(s2p_hip_ctx* ctx, const float* im1, const float* im2, int w, int h, int dmin, int dmax,
                            const s2p_census_params* params, float* disp, float* conf, uint8_t* mask, double timeout_s) {
    return census_host_impl(ctx, im1, im2, w, h, dmin, dmax, params, disp, conf, mask, timeout_s, nullptr);
}

int s2p_hip_census_sgm_debug(s2p_hip_ctx* ctx, const float* im1, const float* im2, int w, int h, int dmin, int dmax,
                             const s2p_census_params* params, float* disp, float* conf, uint8_t* mask, s2p_hip_census_dump* dump) {
    return census_host_impl(ctx, im1, im2, w, h, dmin, dmax, params, disp, conf, mask, -1.0, dump);
}

int s2p_hip_census_sgm_dev(s2p_hip_ctx* ctx, const float* d_im1, const float* d_im2, int w, int h, int dmin, int dmax,
                           const s2p_census_params* params, float* d_disp, float* d_conf, uint8_t* d_mask) {
    if (!ctx || !d_im1 || !d_im2 || !d_disp || w <= 0 || h <= 0) { set_last_error("bad argument"); return S2P_HIP_BAD_ARGUMENT; }
    s2p_census_params p;
    if (params) p = *params; else s2p_hip_census_default_params(&p);
    int rc = check_census_params(p, w, h, dmin, dmax);
    if (rc) return rc;
    S2P_HIP_CHECK(hipSetDevice(ctx->device));
    if (census_levels(w, h, p.scales) > 1)      // the multi-scale mode synchronises between its levels (census_enqueue): never captured
        return census_enqueue(ctx, p, d_im1, d_im2, w, h, dmin, dmax, d_disp, d_conf, d_mask, false, nullptr);
    return run_or_replay(ctx, call_key("census", p, w, h, dmin, dmax, d_im1, d_im2, d_disp, d_conf, d_mask),
                         census_workspace_bytes(p, w, h, dmin, dmax, false),
                         [&]() { return census_enqueue(ctx, p, d_im1, d_im2, w, h, dmin, dmax, d_disp, d_conf, d_mask, false, nullptr); });
}

int s2p_hip_census_sgm_dev_batch(s2p_hip_ctx* ctx, int n, const float* const* d_im1, const float* const* d_im2, int w, int h, int dmin, int dmax,
                                 const s2p_census_params* params, float* const* d_disp, float* const* d_conf, uint8_t* const* d_mask) {
    if (!ctx || n <= 0 || n > 64 || !d_im1 || !d_im2 || !d_disp || w <= 0 || h <= 0) { set_last_error("bad argument"); return S2P_HIP_BAD_ARGUMENT; }
    for (int t = 0; t < n; t++) if (!d_im1[t] || !d_im2[t] || !d_disp[t]) { set_last_error("bad argument"); return S2P_HIP_BAD_ARGUMENT; }
    s2p_census_params p;
    if (params) p = *params; else s2p_hip_census_default_params(&p);
    int rc = check_census_params(p, w, h, dmin, dmax);
    if (rc) return rc;
    if ((double)n * w * h * census_D(p, dmin, dmax) * 9.0 > 6.0e10) { set_last_error("census batch: more than 60 GB of volumes; use smaller batches"); return S2P_HIP_UNSUPPORTED; }
    S2P_HIP_CHECK(hipSetDevice(ctx->device));
    return census_batch_enqueue(ctx, p, n, d_im1, d_im2, w, h, dmin, dmax, d_disp, d_conf, d_mask);
}

// ---- transfers of a host batch ------------------------------------------------------------------------------------------------------
// A caller that keeps a tile's five planes BACK TO BACK in one block of memory -- im1, im2, disp, conf, mask at multiples of the plane size
// rounded up to 256 bytes, as the broker's arenas do (s2p_amd/broker.py: match) -- gets two transfers per tile instead of five: the device
// slot is laid out at the same stride, the two inputs go up as one copy, the three outputs come down as one.  A copy of a megabyte costs
// about as much in set-up as in bytes, and a call of 8 tiles queues 40 of them in front of and behind its kernels.  The outputs' copy
// also writes the (< 256-byte) alignment gaps between the caller's planes; a layout with anything wider between its planes -- room for
// somebody else's data -- does not qualify and keeps the five transfers (include/s2p_hip.h says so at the entry points).
static size_t common_plane_stride(const float* im1, const float* im2, const float* disp, const float* conf, const uint8_t* mask, size_t npx) {
    if (!im1 || !im2 || !disp || !conf || !mask) return 0;
    const uintptr_t a = (uintptr_t)im1, b = (uintptr_t)im2;
    if (b <= a) return 0;
    const size_t s = (size_t)(b - a);
    if (s < npx * 4 || s - npx * 4 >= 256 || (s & 255)) return 0;
    if ((uintptr_t)disp != a + 2 * s || (uintptr_t)conf != a + 3 * s || (uintptr_t)mask != a + 4 * s) return 0;
    return s;
}

// workspace of a host batch of n tiles: the batched launch sequence's volumes (or one tile's, where the parameters have no batched
// form) + n slots of the five planes that travel
static size_t census_host_batch_bytes(const s2p_census_params& p, int n, int w, int h, int dmin, int dmax, size_t* io_bytes_out) {
    const size_t npx = (size_t)w * h, a4 = align_up(npx * 4, 256);
    const size_t io_bytes = (a4 * 4 + align_up(npx, 256)) * n;
    if (io_bytes_out) *io_bytes_out = io_bytes;
    return census_batch_workspace_bytes(p, n, w, h, dmin, dmax) + io_bytes + 4096;      // (one tile's workspace where the parameters have no batched form)
}

int s2p_hip_census_sgm_host_batch_reserve(s2p_hip_ctx* ctx, int n, int w, int h, int dmin, int dmax, const s2p_census_params* params) {
    if (!ctx || n <= 0 || n > 64 || w <= 0 || h <= 0) { set_last_error("bad argument"); return S2P_HIP_BAD_ARGUMENT; }
    s2p_census_params p;
    if (params) p = *params; else s2p_hip_census_default_params(&p);
    int rc = check_census_params(p, w, h, dmin, dmax);
    if (rc) return rc;
    if ((double)n * w * h * census_D(p, dmin, dmax) * 9.0 > 6.0e10) { set_last_error("census batch: more than 60 GB of volumes; use smaller batches"); return S2P_HIP_UNSUPPORTED; }
    S2P_HIP_CHECK(hipSetDevice(ctx->device));
    return ws_reserve(ctx, census_host_batch_bytes(p, n, w, h, dmin, dmax, nullptr));
}

int s2p_hip_census_sgm_host_batch(s2p_hip_ctx* ctx, int n, const float* const* im1, const float* const* im2, int w, int h, int dmin, int dmax,
                                  const s2p_census_params* params, float* const* disp, float* const* conf, uint8_t* const* mask, double timeout_s) {
    if (!ctx || n <= 0 || n > 64 || !im1 || !im2 || !disp || w <= 0 || h <= 0) { set_last_error("bad argument"); return S2P_HIP_BAD_ARGUMENT; }
    for (int t = 0; t < n; t++) if (!im1[t] || !im2[t] || !disp[t]) { set_last_error("bad argument"); return S2P_HIP_BAD_ARGUMENT; }
    if (n == 1) return census_host_impl(ctx, im1[0], im2[0], w, h, dmin, dmax, params, disp[0], conf ? conf[0] : nullptr, mask ? mask[0] : nullptr, timeout_s, nullptr);
    const double deadline = timeout_s < 0 ? -1.0 : now_s() + timeout_s;
    if (timeout_s == 0) { set_last_error("timeout of 0 s: nothing was enqueued"); return S2P_HIP_TIMEOUT; }
    s2p_census_params p;
    if (params) p = *params; else s2p_hip_census_default_params(&p);
    int rc = check_census_params(p, w, h, dmin, dmax);
    if (rc) return rc;
    if ((double)n * w * h * census_D(p, dmin, dmax) * 9.0 > 6.0e10) { set_last_error("census batch: more than 60 GB of volumes; use smaller batches"); return S2P_HIP_UNSUPPORTED; }
    S2P_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t npx = (size_t)w * h;
    // one plane stride for the whole batch: the callers' own when every tile keeps its planes at the same one, else the packed 256-byte one
    size_t a4 = align_up(npx * 4, 256);
    bool coalesce = conf && mask;
    for (int t = 0; t < n && coalesce; t++) {
        const size_t st = common_plane_stride(im1[t], im2[t], disp[t], conf[t], mask[t], npx);
        if (!st || (t > 0 && st != a4)) coalesce = false; else a4 = st;
    }
    if (!coalesce) a4 = align_up(npx * 4, 256);
    const size_t slot = a4 * 4 + align_up(npx, 256);
    const size_t io_bytes = slot * n;
    rc = ws_reserve(ctx, census_batch_workspace_bytes(p, n, w, h, dmin, dmax) + io_bytes + 4096);
    if (rc) return rc;
    char* io = ctx->ws + ctx->ws_size - io_bytes;
    std::vector<const float*> d1(n), d2(n);
    std::vector<float*> dd(n), dc(n);
    std::vector<uint8_t*> dm(n);
    for (int t = 0; t < n; t++) {
        char* s = io + slot * t;
        d1[t] = (float*)s; d2[t] = (float*)(s + a4); dd[t] = (float*)(s + 2 * a4);
        dc[t] = (conf && conf[t]) ? (float*)(s + 3 * a4) : nullptr; dm[t] = (uint8_t*)(s + 4 * a4);
        if (coalesce) S2P_HIP_CHECK(hipMemcpyAsync((void*)d1[t], im1[t], a4 + npx * 4, hipMemcpyHostToDevice, ctx->stream));
        else {
            S2P_HIP_CHECK(hipMemcpyAsync((void*)d1[t], im1[t], npx * 4, hipMemcpyHostToDevice, ctx->stream));
            S2P_HIP_CHECK(hipMemcpyAsync((void*)d2[t], im2[t], npx * 4, hipMemcpyHostToDevice, ctx->stream));
        }
    }
    rc = census_batch_enqueue(ctx, p, n, d1.data(), d2.data(), w, h, dmin, dmax, dd.data(), dc.data(), dm.data());
    if (rc) return rc;
    for (int t = 0; t < n; t++) {
        if (coalesce) {
            if (a4 > npx * 4) {                             // the alignment gaps behind disp and conf travel too: nothing of the workspace's past rides in them
                S2P_HIP_CHECK(hipMemsetAsync((char*)dd[t] + npx * 4, 0, a4 - npx * 4, ctx->stream));
                S2P_HIP_CHECK(hipMemsetAsync((char*)dd[t] + a4 + npx * 4, 0, a4 - npx * 4, ctx->stream));
            }
            S2P_HIP_CHECK(hipMemcpyAsync(disp[t], dd[t], 2 * a4 + npx, hipMemcpyDeviceToHost, ctx->stream));
            continue;
        }
        S2P_HIP_CHECK(hipMemcpyAsync(disp[t], dd[t], npx * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (dc[t]) S2P_HIP_CHECK(hipMemcpyAsync(conf[t], dc[t], npx * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (mask && mask[t]) S2P_HIP_CHECK(hipMemcpyAsync(mask[t], dm[t], npx, hipMemcpyDeviceToHost, ctx->stream));
    }
    return wait_stream(ctx, deadline);
}

int s2p_hip_census_sgm_host_batch_v(s2p_hip_ctx* ctx, int n, const float* const* im1, const float* const* im2, const int* w, const int* h,
                                    const int* dmin, const int* dmax, const s2p_census_params* params,
                                    float* const* disp, float* const* conf, uint8_t* const* mask, double timeout_s) {
    if (!ctx || n <= 0 || n > 16 || !im1 || !im2 || !disp || !w || !h || !dmin || !dmax) { set_last_error("bad argument"); return S2P_HIP_BAD_ARGUMENT; }
    for (int t = 0; t < n; t++) if (!im1[t] || !im2[t] || !disp[t] || w[t] <= 0 || h[t] <= 0) { set_last_error("bad argument"); return S2P_HIP_BAD_ARGUMENT; }
    if (n == 1) return census_host_impl(ctx, im1[0], im2[0], w[0], h[0], dmin[0], dmax[0], params, disp[0], conf ? conf[0] : nullptr, mask ? mask[0] : nullptr, timeout_s, nullptr);
    const double deadline = timeout_s < 0 ? -1.0 : now_s() + timeout_s;
    if (timeout_s == 0) { set_last_error("timeout of 0 s: nothing was enqueued"); return S2P_HIP_TIMEOUT; }
    s2p_census_params p;
    if (params) p = *params; else s2p_hip_census_default_params(&p);
    double cand = 0;
    int Dmax = 0;
    for (int t = 0; t < n; t++) {
        int rc = check_census_params(p, w[t], h[t], dmin[t], dmax[t]);
        if (rc) return rc;
        Dmax = std::max(Dmax, census_D(p, dmin[t], dmax[t]));
    }
    for (int t = 0; t < n; t++) {
        if ((double)w[t] * h[t] * Dmax >= 4294967296.0 - 65536.0) { set_last_error("census batch: a tile's volume at the batch's depth exceeds 4 GiB"); return S2P_HIP_UNSUPPORTED; }
        cand += (double)w[t] * h[t] * Dmax;
    }
    if (cand * 9.0 > 6.0e10) { set_last_error("census batch: more than 60 GB of volumes; use smaller batches"); return S2P_HIP_UNSUPPORTED; }
    S2P_HIP_CHECK(hipSetDevice(ctx->device));
    size_t io_bytes = 0;
    std::vector<size_t> slot(n), st4(n);
    std::vector<char> one(n);
    for (int t = 0; t < n; t++) {                          // per tile: its caller's own plane stride when it has one (two transfers), else packed (five)
        const size_t npx = (size_t)w[t] * h[t];
        const size_t cs = (conf && mask) ? common_plane_stride(im1[t], im2[t], disp[t], conf[t], mask[t], npx) : 0;
        one[t] = cs != 0; st4[t] = cs ? cs : align_up(npx * 4, 256);
        slot[t] = io_bytes; io_bytes += st4[t] * 4 + align_up(npx, 256);
    }
    int rc = ws_reserve(ctx, census_batch_hetero_workspace_bytes(p, n, w, h, dmin, dmax) + io_bytes + 4096);
    if (rc) return rc;
    char* io = ctx->ws + ctx->ws_size - io_bytes;
    std::vector<const float*> d1(n), d2(n);
    std::vector<float*> dd(n), dc(n);
    std::vector<uint8_t*> dm(n);
    for (int t = 0; t < n; t++) {
        const size_t npx = (size_t)w[t] * h[t], a4 = st4[t];
        char* s = io + slot[t];
        d1[t] = (float*)s; d2[t] = (float*)(s + a4); dd[t] = (float*)(s + 2 * a4);
        dc[t] = (conf && conf[t]) ? (float*)(s + 3 * a4) : nullptr; dm[t] = (uint8_t*)(s + 4 * a4);
        if (one[t]) S2P_HIP_CHECK(hipMemcpyAsync((void*)d1[t], im1[t], a4 + npx * 4, hipMemcpyHostToDevice, ctx->stream));
        else {
            S2P_HIP_CHECK(hipMemcpyAsync((void*)d1[t], im1[t], npx * 4, hipMemcpyHostToDevice, ctx->stream));
            S2P_HIP_CHECK(hipMemcpyAsync((void*)d2[t], im2[t], npx * 4, hipMemcpyHostToDevice, ctx->stream));
        }
    }
    rc = census_batch_hetero_enqueue(ctx, p, n, d1.data(), d2.data(), w, h, dmin, dmax, dd.data(), dc.data(), dm.data());
    if (rc) return rc;
    for (int t = 0; t < n; t++) {
        const size_t npx = (size_t)w[t] * h[t];
        if (one[t]) {
            if (st4[t] > npx * 4) {                         // (as above: zeroed gaps)
                S2P_HIP_CHECK(hipMemsetAsync((char*)dd[t] + npx * 4, 0, st4[t] - npx * 4, ctx->stream));
                S2P_HIP_CHECK(hipMemsetAsync((char*)dd[t] + st4[t] + npx * 4, 0, st4[t] - npx * 4, ctx->stream));
            }
            S2P_HIP_CHECK(hipMemcpyAsync(disp[t], dd[t], 2 * st4[t] + npx, hipMemcpyDeviceToHost, ctx->stream));
            continue;
        }
        S2P_HIP_CHECK(hipMemcpyAsync(disp[t], dd[t], npx * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (dc[t]) S2P_HIP_CHECK(hipMemcpyAsync(conf[t], dc[t], npx * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (mask && mask[t]) S2P_HIP_CHECK(hipMemcpyAsync(mask[t], dm[t], npx, hipMemcpyDeviceToHost, ctx->stream));
    }
    return wait_stream(ctx, deadline);
}

int s2p_hip_warp_dev(s2p_hip_ctx* ctx, const void* d_src, int src_dtype, int sw, int sh, const double H[9],
                     float* d_dst, int w, int h) {
    if (!ctx || !d_src || !H || !d_dst || sw <= 0 || sh <= 0 || w <= 0 || h <= 0) { set_last_error("bad argument"); return S2P_HIP_BAD_ARGUMENT; }
    S2P_HIP_CHECK(hipSetDevice(ctx->device));
    int rc = ws_reserve(ctx, warp_workspace_bytes(sw, sh));
    if (rc) return rc;
    ws_reset(ctx);
    return warp_enqueue(ctx, d_src, src_dtype, sw, sh, H, d_dst, w, h, ctx->ws);
}

int s2p_hip_warp_host(s2p_hip_ctx* ctx, const void* src, int src_dtype, int sw, int sh, const double H[9],
                      float* dst, int w, int h) {
    if (!ctx || !src || !H || !dst || sw <= 0 || sh <= 0 || w <= 0 || h <= 0) { set_last_error("bad argument"); return S2P_HIP_BAD_ARGUMENT; }
    if (src_dtype < 0 || src_dtype > 2) { set_last_error("warp: unknown source dtype %d", src_dtype); return S2P_HIP_BAD_ARGUMENT; }
    S2P_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t esz = src_dtype == S2P_HIP_F32 ? 4 : src_dtype == S2P_HIP_U16 ? 2 : 1;
    const size_t nsrc = (size_t)sw * sh * esz, ndst = (size_t)w * h * 4;
    const size_t wsb = warp_workspace_bytes(sw, sh);
    int rc = ws_reserve(ctx, wsb + align_up(nsrc, 256) + align_up(ndst, 256) + 4096);
    if (rc) return rc;
    ws_reset(ctx);
    char* d_src = ctx->ws + align_up(wsb, 256);
    float* d_dst = (float*)(d_src + align_up(nsrc, 256));
    S2P_HIP_CHECK(hipMemcpyAsync(d_src, src, nsrc, hipMemcpyHostToDevice, ctx->stream));
    rc = warp_enqueue(ctx, d_src, src_dtype, sw, sh, H, d_dst, w, h, ctx->ws);
    if (rc) return rc;
    S2P_HIP_CHECK(hipMemcpyAsync(dst, d_dst, ndst, hipMemcpyDeviceToHost, ctx->stream));
    S2P_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return S2P_HIP_OK;
}

int s2p_hip_rejection_mask_host(s2p_hip_ctx* ctx, const float* disp, const float* im1, const float* im2, int w, int h, uint8_t* mask) {
    if (!ctx || !disp || !im1 || !im2 || !mask || w <= 0 || h <= 0) { set_last_error("bad argument"); return S2P_HIP_BAD_ARGUMENT; }
    S2P_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t npx = (size_t)w * h, a4 = align_up(npx * 4, 256);
    int rc = ws_reserve(ctx, a4 * 3 + align_up(npx, 256) + 4096);
    if (rc) return rc;
    ws_reset(ctx);
    float* d_d = (float*)ws_alloc(ctx, npx * 4); float* d_a = (float*)ws_alloc(ctx, npx * 4); float* d_b = (float*)ws_alloc(ctx, npx * 4);
    uint8_t* d_m = (uint8_t*)ws_alloc(ctx, npx);
    if (!d_d || !d_a || !d_b || !d_m) return S2P_HIP_RUNTIME_ERROR;
    S2P_HIP_CHECK(hipMemcpyAsync(d_d, disp, npx * 4, hipMemcpyHostToDevice, ctx->stream));
    S2P_HIP_CHECK(hipMemcpyAsync(d_a, im1, npx * 4, hipMemcpyHostToDevice, ctx->stream));
    S2P_HIP_CHECK(hipMemcpyAsync(d_b, im2, npx * 4, hipMemcpyHostToDevice, ctx->stream));
    rc = rejection_mask_enqueue(ctx, d_d, d_a, d_b, w, h, d_m);
    if (rc) return rc;
    S2P_HIP_CHECK(hipMemcpyAsync(mask, d_m, npx, hipMemcpyDeviceToHost, ctx->stream));
    S2P_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return S2P_HIP_OK;
}

int s2p_hip_disp_to_lonlatalt_host(s2p_hip_ctx* ctx, double* lonlatalt, float* err, const float* dispx, const float* dispy,
                                   const float* msk, int nx, int ny, const float* msk_orig, int w, int h,
                                   const double ha[9], const double hb[9], const s2p_rpc* rpca, const s2p_rpc* rpcb,
                                   const float bbox[4]) {
    if (!ctx || !lonlatalt || !err || !dispx || !msk || !msk_orig || !ha || !hb || !rpca || !rpcb || !bbox ||
        nx <= 0 || ny <= 0 || w <= 0 || h <= 0) { set_last_error("bad argument"); return S2P_HIP_BAD_ARGUMENT; }
    S2P_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t n = (size_t)nx * ny, no = (size_t)w * h;
    const size_t a4 = align_up(n * 4, 256);
    int rc = ws_reserve(ctx, 4 * a4 + align_up(no * 4, 256) + align_up(n * 24, 256) + 2 * sizeof(s2p_rpc) + 8192);
    if (rc) return rc;
    ws_reset(ctx);
    float* d_dx = (float*)ws_alloc(ctx, n * 4); float* d_dy = (float*)ws_alloc(ctx, n * 4); float* d_m = (float*)ws_alloc(ctx, n * 4);
    float* d_e = (float*)ws_alloc(ctx, n * 4); float* d_mo = (float*)ws_alloc(ctx, no * 4);
    double* d_l = (double*)ws_alloc(ctx, n * 24); s2p_rpc* d_r = (s2p_rpc*)ws_alloc(ctx, 2 * sizeof(s2p_rpc));
    if (!d_dx || !d_dy || !d_m || !d_e || !d_mo || !d_l || !d_r) return S2P_HIP_RUNTIME_ERROR;
    S2P_HIP_CHECK(hipMemcpyAsync(d_dx, dispx, n * 4, hipMemcpyHostToDevice, ctx->stream));
    if (dispy) S2P_HIP_CHECK(hipMemcpyAsync(d_dy, dispy, n * 4, hipMemcpyHostToDevice, ctx->stream));
    S2P_HIP_CHECK(hipMemcpyAsync(d_m, msk, n * 4, hipMemcpyHostToDevice, ctx->stream));
    S2P_HIP_CHECK(hipMemcpyAsync(d_mo, msk_orig, no * 4, hipMemcpyHostToDevice, ctx->stream));
    S2P_HIP_CHECK(hipMemcpyAsync(d_r, rpca, sizeof(s2p_rpc), hipMemcpyHostToDevice, ctx->stream));
    S2P_HIP_CHECK(hipMemcpyAsync(d_r + 1, rpcb, sizeof(s2p_rpc), hipMemcpyHostToDevice, ctx->stream));
    rc = tri_enqueue(ctx, d_dx, dispy ? d_dy : nullptr, d_m, nx, ny, d_mo, w, h, ha, hb, d_r, bbox, d_l, d_e);
    if (rc) return rc;
    S2P_HIP_CHECK(hipMemcpyAsync(lonlatalt, d_l, n * 24, hipMemcpyDeviceToHost, ctx->stream));
    S2P_HIP_CHECK(hipMemcpyAsync(err, d_e, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    S2P_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return S2P_HIP_OK;
}

// Process-wide context of the four entry points that keep the reference's own (context-free, void) signatures.
// One workspace and one stream: calls from several threads of a process (ctypes releases the GIL) are serialised by g_global_mutex,
// which every wrapper holds for its whole call.
static std::mutex g_global_mutex;
static s2p_hip_ctx* global_ctx(const char* who) {
    static s2p_hip_ctx* g_ctx = nullptr;               // created on first use (after any fork)
    static int g_pid = -1;
    if (!g_ctx || g_pid != (int)getpid()) {
        int n = s2p_hip_device_count(), dev = 0;
        const char* e = getenv("S2P_HIP_DEVICE");
        if (!e) e = getenv("LOCAL_RANK");
        if (e) dev = atoi(e); else if (n > 0) dev = (int)(getpid() % n);
        if (s2p_hip_ctx_create(dev, nullptr, &g_ctx) != S2P_HIP_OK) {
            fprintf(stderr, "libs2p_hip: %s: %s (no CPU fallback)\n", who, s2p_hip_last_error());
            abort();
        }
        g_pid = (int)getpid();
    }
    return g_ctx;
}
static void global_check(const char* who, int rc) {
    if (rc != S2P_HIP_OK) {
        fprintf(stderr, "libs2p_hip: %s failed with status %d: %s\n", who, rc, s2p_hip_last_error());
        abort();
    }
}

void disp_to_lonlatalt(double* lonlatalt, float* err, float* dispx, float* dispy, float* msk, int nx, int ny,
                       float* msk_orig, int w, int h, double ha[9], double hb[9],
                       s2p_rpc* rpca, s2p_rpc* rpcb, float orig_img_bounding_box[4]) {
    std::lock_guard<std::mutex> lock(g_global_mutex);
    global_check("disp_to_lonlatalt", s2p_hip_disp_to_lonlatalt_host(global_ctx("disp_to_lonlatalt"), lonlatalt, err, dispx, dispy, msk,
                                                                     nx, ny, msk_orig, w, h, ha, hb, rpca, rpcb, orig_img_bounding_box));
}

int s2p_hip_stereo_corresp_to_lonlatalt_host(s2p_hip_ctx* ctx, double* lonlatalt, float* err, const float* kp_a, const float* kp_b,
                                             int n_kp, const s2p_rpc* rpca, const s2p_rpc* rpcb) {
    if (!ctx || !lonlatalt || !err || !kp_a || !kp_b || !rpca || !rpcb || n_kp < 0) { set_last_error("bad argument"); return S2P_HIP_BAD_ARGUMENT; }
    if (n_kp == 0) return S2P_HIP_OK;
    S2P_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t n = (size_t)n_kp;
    int rc = ws_reserve(ctx, 3 * align_up(n * 8, 256) + align_up(n * 24, 256) + 2 * sizeof(s2p_rpc) + 8192);
    if (rc) return rc;
    ws_reset(ctx);
    float* d_a = (float*)ws_alloc(ctx, n * 8); float* d_b = (float*)ws_alloc(ctx, n * 8); float* d_e = (float*)ws_alloc(ctx, n * 4);
    double* d_l = (double*)ws_alloc(ctx, n * 24); s2p_rpc* d_r = (s2p_rpc*)ws_alloc(ctx, 2 * sizeof(s2p_rpc));
    if (!d_a || !d_b || !d_e || !d_l || !d_r) return S2P_HIP_RUNTIME_ERROR;
    S2P_HIP_CHECK(hipMemcpyAsync(d_a, kp_a, n * 8, hipMemcpyHostToDevice, ctx->stream));
    S2P_HIP_CHECK(hipMemcpyAsync(d_b, kp_b, n * 8, hipMemcpyHostToDevice, ctx->stream));
    S2P_HIP_CHECK(hipMemcpyAsync(d_r, rpca, sizeof(s2p_rpc), hipMemcpyHostToDevice, ctx->stream));
    S2P_HIP_CHECK(hipMemcpyAsync(d_r + 1, rpcb, sizeof(s2p_rpc), hipMemcpyHostToDevice, ctx->stream));
    rc = corresp_enqueue(ctx, d_a, d_b, n_kp, d_r, d_l, d_e);
    if (rc) return rc;
    S2P_HIP_CHECK(hipMemcpyAsync(lonlatalt, d_l, n * 24, hipMemcpyDeviceToHost, ctx->stream));
    S2P_HIP_CHECK(hipMemcpyAsync(err, d_e, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    S2P_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return S2P_HIP_OK;
}

void stereo_corresp_to_lonlatalt(double* lonlatalt, float* err, float* kp_a, float* kp_b, int n_kp, s2p_rpc* rpc_a, s2p_rpc* rpc_b) {
    std::lock_guard<std::mutex> lock(g_global_mutex);
    global_check("stereo_corresp_to_lonlatalt", s2p_hip_stereo_corresp_to_lonlatalt_host(global_ctx("stereo_corresp_to_lonlatalt"), lonlatalt, err,
                                                                                          kp_a, kp_b, n_kp, rpc_a, rpc_b));
}

static int filter3d_impl(s2p_hip_ctx* ctx, int* count, double* xyz, int nx, int ny, float r, int p, int n, int q, bool remove) {
    if (!ctx || !xyz || (!remove && !count) || nx <= 0 || ny <= 0 || p < 0 || q < 0) { set_last_error("bad argument"); return S2P_HIP_BAD_ARGUMENT; }
    S2P_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t npx = (size_t)nx * ny;
    int rc = ws_reserve(ctx, align_up(npx * 24, 256) + align_up(npx * 4, 256) + align_up(npx, 256) + 8192);
    if (rc) return rc;
    ws_reset(ctx);
    double* d_xyz = (double*)ws_alloc(ctx, npx * 24); int* d_cnt = (int*)ws_alloc(ctx, npx * 4);
    uint8_t* d_rej = (uint8_t*)ws_alloc(ctx, npx); int* d_flag = (int*)ws_alloc(ctx, 256);
    if (!d_xyz || !d_cnt || !d_rej || !d_flag) return S2P_HIP_RUNTIME_ERROR;
    S2P_HIP_CHECK(hipMemcpyAsync(d_xyz, xyz, npx * 24, hipMemcpyHostToDevice, ctx->stream));
    if (!remove) {
        rc = count3d_enqueue(ctx, d_xyz, nx, ny, r, p, d_cnt);
        if (rc) return rc;
        S2P_HIP_CHECK(hipMemcpyAsync(count, d_cnt, npx * 4, hipMemcpyDeviceToHost, ctx->stream));
    } else {
        rc = remove_isolated_enqueue(ctx, d_xyz, nx, ny, r, p, n, q, d_cnt, d_rej, d_flag);
        if (rc) return rc;
        S2P_HIP_CHECK(hipMemcpyAsync(xyz, d_xyz, npx * 24, hipMemcpyDeviceToHost, ctx->stream));
    }
    S2P_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return S2P_HIP_OK;
}
int s2p_hip_count_3d_neighbors_host(s2p_hip_ctx* ctx, int* count, const double* xyz, int nx, int ny, float r, int p) {
    return filter3d_impl(ctx, count, const_cast<double*>(xyz), nx, ny, r, p, 0, 0, false);
}
int s2p_hip_remove_isolated_3d_points_host(s2p_hip_ctx* ctx, double* xyz, int nx, int ny, float r, int p, int n, int q) {
    return filter3d_impl(ctx, nullptr, xyz, nx, ny, r, p, n, q, true);
}
void count_3d_neighbors(int* count, double* xyz, int nx, int ny, float r, int p) {
    std::lock_guard<std::mutex> lock(g_global_mutex);
    global_check("count_3d_neighbors", s2p_hip_count_3d_neighbors_host(global_ctx("count_3d_neighbors"), count, xyz, nx, ny, r, p));
}
void remove_isolated_3d_points(double* xyz, int nx, int ny, float r, int p, int n, int q) {
    std::lock_guard<std::mutex> lock(g_global_mutex);
    global_check("remove_isolated_3d_points", s2p_hip_remove_isolated_3d_points_host(global_ctx("remove_isolated_3d_points"), xyz, nx, ny, r, p, n, q));
}

int s2p_hip_erode_mask_host(s2p_hip_ctx* ctx, const uint8_t* mask, int w, int h, int radius, uint8_t* out) {
    if (!ctx || !mask || !out || w <= 0 || h <= 0 || radius < 0 || radius > 64) { set_last_error("bad argument"); return S2P_HIP_BAD_ARGUMENT; }
    S2P_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t npx = (size_t)w * h;
    int rc = ws_reserve(ctx, 2 * align_up(npx, 256) + 4096);
    if (rc) return rc;
    ws_reset(ctx);
    uint8_t* d_in = (uint8_t*)ws_alloc(ctx, npx); uint8_t* d_out = (uint8_t*)ws_alloc(ctx, npx);
    if (!d_in || !d_out) return S2P_HIP_RUNTIME_ERROR;
    S2P_HIP_CHECK(hipMemcpyAsync(d_in, mask, npx, hipMemcpyHostToDevice, ctx->stream));
    rc = erode_enqueue(ctx, d_in, w, h, radius, d_out);
    if (rc) return rc;
    S2P_HIP_CHECK(hipMemcpyAsync(out, d_out, npx, hipMemcpyDeviceToHost, ctx->stream));
    S2P_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return S2P_HIP_OK;
}

int s2p_hip_merge_n_host(s2p_hip_ctx* ctx, const float* const* inputs, const double* offsets, int n, int w, int h,
                         int op, double threshold, float* out) {
    if (!ctx || !inputs || !offsets || !out || n <= 0 || n > 64 || w <= 0 || h <= 0 || op < 0 || op > 8) { set_last_error("merge_n: bad argument"); return S2P_HIP_BAD_ARGUMENT; }
    for (int i = 0; i < n; i++) if (!inputs[i]) { set_last_error("merge_n: bad argument"); return S2P_HIP_BAD_ARGUMENT; }
    S2P_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t npx = (size_t)w * h, a4 = align_up(npx * 4, 256);
    int rc = ws_reserve(ctx, (size_t)(n + 1) * a4 + 4096);
    if (rc) return rc;
    ws_reset(ctx);
    float* d_stack = (float*)ws_alloc(ctx, (size_t)n * npx * 4);
    float* d_out = (float*)ws_alloc(ctx, npx * 4);
    double* d_off = (double*)ws_alloc(ctx, (size_t)n * 8);
    if (!d_stack || !d_out || !d_off) return S2P_HIP_RUNTIME_ERROR;
    for (int i = 0; i < n; i++) S2P_HIP_CHECK(hipMemcpyAsync(d_stack + (size_t)i * npx, inputs[i], npx * 4, hipMemcpyHostToDevice, ctx->stream));
    S2P_HIP_CHECK(hipMemcpyAsync(d_off, offsets, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    double so[64];
    for (int i = 0; i < n; i++) so[i] = offsets[i];
    // np.mean(offsets) (s2p/fusion.py:61): numpy's pairwise sum / n
    double tot;
    if (n < 8) { tot = 0.0; for (int i = 0; i < n; i++) tot += so[i]; }
    else {
        double r[8]; for (int j = 0; j < 8; j++) r[j] = so[j];
        int i = 8;
        for (; i < n - (n % 8); i += 8) for (int j = 0; j < 8; j++) r[j] += so[i + j];
        tot = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) tot += so[i];
    }
    rc = merge_enqueue(ctx, d_stack, d_off, n, npx, op, threshold, tot / (double)n, d_out);
    if (rc) return rc;
    S2P_HIP_CHECK(hipMemcpyAsync(out, d_out, npx * 4, hipMemcpyDeviceToHost, ctx->stream));
    S2P_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return S2P_HIP_OK;
}

// ---- common.cargarse_basura (include/s2p_hip.h: s2p_hip_cargarse_basura_host) ---------------------------------------
int s2p_hip_cargarse_basura_host(s2p_hip_ctx* ctx, const float* in, int w, int h, float* out) {
    if (!ctx || !in || !out || w <= 0 || h <= 0) { set_last_error("cargarse_basura: bad argument"); return S2P_HIP_BAD_ARGUMENT; }
    S2P_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t npx = (size_t)w * h, a4 = align_up(npx * 4, 256);
    int rc = ws_reserve(ctx, 5 * a4 + 4096);
    if (rc) return rc;
    ws_reset(ctx);
    float* d_in = (float*)ws_alloc(ctx, npx * 4); float* d_out = (float*)ws_alloc(ctx, npx * 4);
    int* lab = (int*)ws_alloc(ctx, npx * 4); int* par = (int*)ws_alloc(ctx, npx * 4); int* cnt = (int*)ws_alloc(ctx, npx * 4);
    if (!d_in || !d_out || !lab || !par || !cnt) return S2P_HIP_RUNTIME_ERROR;
    S2P_HIP_CHECK(hipMemcpyAsync(d_in, in, npx * 4, hipMemcpyHostToDevice, ctx->stream));
    rc = cargarse_basura_enqueue(ctx, d_in, w, h, d_out, lab, par, cnt);
    if (rc) return rc;
    S2P_HIP_CHECK(hipMemcpyAsync(out, d_out, npx * 4, hipMemcpyDeviceToHost, ctx->stream));
    S2P_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return S2P_HIP_OK;
}

// ---- triangulation.height_map_to_xyz, the localisation (include/s2p_hip.h: s2p_hip_height_map_to_lonlatalt_host) ----
int s2p_hip_height_map_to_lonlatalt_host(s2p_hip_ctx* ctx, const s2p_rpc* rpc, const float* heights, int w, int h, int off_x, int off_y,
                                         double* lonlatalt) {
    if (!ctx || !rpc || !heights || !lonlatalt || w <= 0 || h <= 0) { set_last_error("height_map_to_lonlatalt: bad argument"); return S2P_HIP_BAD_ARGUMENT; }
    S2P_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t npx = (size_t)w * h;
    int rc = ws_reserve(ctx, align_up(npx * 4, 256) + align_up(npx * 24, 256) + sizeof(s2p_rpc) + 4096);
    if (rc) return rc;
    ws_reset(ctx);
    float* d_hm = (float*)ws_alloc(ctx, npx * 4); double* d_l = (double*)ws_alloc(ctx, npx * 24);
    s2p_rpc* d_r = (s2p_rpc*)ws_alloc(ctx, sizeof(s2p_rpc));
    if (!d_hm || !d_l || !d_r) return S2P_HIP_RUNTIME_ERROR;
    S2P_HIP_CHECK(hipMemcpyAsync(d_hm, heights, npx * 4, hipMemcpyHostToDevice, ctx->stream));
    S2P_HIP_CHECK(hipMemcpyAsync(d_r, rpc, sizeof(s2p_rpc), hipMemcpyHostToDevice, ctx->stream));
    rc = height_map_localize_enqueue(ctx, d_r, d_hm, w, h, off_x, off_y, d_l);
    if (rc) return rc;
    S2P_HIP_CHECK(hipMemcpyAsync(lonlatalt, d_l, npx * 24, hipMemcpyDeviceToHost, ctx->stream));
    S2P_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return S2P_HIP_OK;
}

// ---- triangulation.height_map's resampling (include/s2p_hip.h: s2p_hip_height_transfer_host) ----------------------
int s2p_hip_height_transfer_host(s2p_hip_ctx* ctx, const double* heights, int wr, int hr, const double H[9], int w, int h, double* out) {
    if (!ctx || !heights || !H || !out || wr <= 0 || hr <= 0 || w <= 0 || h <= 0) { set_last_error("height_transfer: bad argument"); return S2P_HIP_BAD_ARGUMENT; }
    if (H[6] != 0.0 || H[7] != 0.0 || H[8] != 1.0) {       // scipy raises ValueError on such a matrix
        set_last_error("height_transfer: the bottom row of H must be [0, 0, 1] (affine_transform takes an affine map)"); return S2P_HIP_BAD_ARGUMENT;
    }
    S2P_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t nin = (size_t)wr * hr, nout = (size_t)w * h;
    int rc = ws_reserve(ctx, align_up(nin * 8, 256) + 2 * align_up(nout * 8, 256) + align_up(nout, 256) + 4096);
    if (rc) return rc;
    ws_reset(ctx);
    double* d_hm = (double*)ws_alloc(ctx, nin * 8);
    double* d_val = (double*)ws_alloc(ctx, nout * 8);
    double* d_out = (double*)ws_alloc(ctx, nout * 8);
    uint8_t* d_flag = (uint8_t*)ws_alloc(ctx, nout);
    if (!d_hm || !d_val || !d_out || !d_flag) return S2P_HIP_RUNTIME_ERROR;
    S2P_HIP_CHECK(hipMemcpyAsync(d_hm, heights, nin * 8, hipMemcpyHostToDevice, ctx->stream));
    rc = height_transfer_enqueue(ctx, d_hm, wr, hr, H, w, h, d_val, d_flag, d_out);
    if (rc) return rc;
    S2P_HIP_CHECK(hipMemcpyAsync(out, d_out, nout * 8, hipMemcpyDeviceToHost, ctx->stream));
    S2P_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return S2P_HIP_OK;
}

// ---- DSM rasterisation (include/s2p_hip.h: s2p_hip_plyflatten_host, rasterize_cloud) -----------------------------
int s2p_hip_plyflatten_host(s2p_hip_ctx* ctx, const double* cloud, int nb_points, int nb_extra_columns, double xoff, double yoff,
                            double resolution, int xsize, int ysize, int radius, float sigma, float* raster) {
    if (!ctx || !raster || (!cloud && nb_points > 0) || nb_points < 0 || nb_extra_columns <= 0 || nb_extra_columns > 16 ||
        xsize <= 0 || ysize <= 0 || radius < 0 || radius > 64 || !(resolution > 0) || sigma != sigma) {
        set_last_error("plyflatten: bad argument"); return S2P_HIP_BAD_ARGUMENT;
    }
    const size_t ncell = (size_t)xsize * ysize, nb = (size_t)nb_extra_columns;
    if (ncell >= ((size_t)1 << 31) || (size_t)nb_points * raster_disc_cells(radius) >= ((size_t)1 << 31)) {
        set_last_error("plyflatten: more than 2^31 cells or point-cell contributions"); return S2P_HIP_UNSUPPORTED;
    }
    S2P_HIP_CHECK(hipSetDevice(ctx->device));
    int rc = ws_reserve(ctx, raster_workspace_bytes(nb_points, nb_extra_columns, xsize, ysize, radius));
    if (rc) return rc;
    ws_reset(ctx);
    const size_t pbytes = (size_t)nb_points * (2 + nb) * 8;
    double* d_pts = (double*)ws_alloc(ctx, std::max<size_t>(pbytes, 8));
    float* d_raster = (float*)ws_alloc(ctx, ncell * nb * 4);
    if (!d_pts || !d_raster) return S2P_HIP_RUNTIME_ERROR;
    if (pbytes) S2P_HIP_CHECK(hipMemcpyAsync(d_pts, cloud, pbytes, hipMemcpyHostToDevice, ctx->stream));
    rc = raster_enqueue(ctx, d_pts, nb_points, nb_extra_columns, xoff, yoff, resolution, xsize, ysize, radius, sigma, d_raster);
    if (rc) return rc;
    S2P_HIP_CHECK(hipMemcpyAsync(raster, d_raster, ncell * nb * 4, hipMemcpyDeviceToHost, ctx->stream));
    S2P_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return S2P_HIP_OK;
}

// the symbol plyflatten's Python binds in its own libplyflatten.so: same argument list, process-wide context
void rasterize_cloud(double* input_buffer, float* raster, int nb_points, int nb_extra_columns, double xoff, double yoff,
                     double resolution, int xsize, int ysize, int radius, float sigma) {
    std::lock_guard<std::mutex> lock(g_global_mutex);
    global_check("rasterize_cloud", s2p_hip_plyflatten_host(global_ctx("rasterize_cloud"), input_buffer, nb_points, nb_extra_columns,
                                                            xoff, yoff, resolution, xsize, ysize, radius, sigma, raster));
}

// ---- one tile end to end (include/s2p_hip.h: s2p_hip_tile_host) --------------------------------------
static __global__ __launch_bounds__(256) void k_mask_u8_to_f32(const uint8_t* __restrict__ m, size_t n, float* __restrict__ out)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (float)m[i];
}

// The persistent buffers of one tile (top of the workspace; each step's scratch is carved from the bottom).
struct TileSlot {
    float *r1, *r2, *disp, *mf, *err, *mo;
    uint8_t *mask, *mask_e;
    char *s1, *s2;
    double* lla;
    s2p_rpc* rpc;
};
static const size_t k_src_esz[3] = {4, 2, 1};
static bool tile_args_ok(const s2p_tile* t, const s2p_tile_out* o) {
    if (!t || !o || !t->src1 || !t->src2 || t->w <= 0 || t->h <= 0 || t->sw1 <= 0 || t->sh1 <= 0 || t->sw2 <= 0 || t->sh2 <= 0 ||
        t->src1_dtype < 0 || t->src1_dtype > 2 || t->src2_dtype < 0 || t->src2_dtype > 2 || (t->algo != 0 && t->algo != 1) ||
        t->erosion < 0 || t->erosion > 64 || ((t->rpca == nullptr) != (t->rpcb == nullptr))) { set_last_error("tile: bad argument"); return false; }
    if (t->rpca && (!t->msk_orig || t->ow <= 0 || t->oh <= 0)) { set_last_error("tile: triangulation needs msk_orig"); return false; }
    return true;
}
static size_t tile_slot_bytes(const s2p_tile* t) {
    const bool tri = t->rpca != nullptr;
    const size_t npx = (size_t)t->w * t->h, a4 = align_up(npx * 4, 256), a1 = align_up(npx, 256);
    const size_t s1 = align_up((size_t)t->sw1 * t->sh1 * k_src_esz[t->src1_dtype], 256), s2 = align_up((size_t)t->sw2 * t->sh2 * k_src_esz[t->src2_dtype], 256);
    const size_t no = tri ? (size_t)t->ow * t->oh : 0;
    return 5 * a4 + 2 * a1 + s1 + s2 + align_up(no * 4, 256) + align_up(tri ? npx * 24 : 0, 256) + align_up(2 * sizeof(s2p_rpc), 256);
}
static TileSlot tile_slot_at(const s2p_tile* t, char* io) {
    const bool tri = t->rpca != nullptr;
    const size_t npx = (size_t)t->w * t->h, a4 = align_up(npx * 4, 256), a1 = align_up(npx, 256);
    const size_t s1 = align_up((size_t)t->sw1 * t->sh1 * k_src_esz[t->src1_dtype], 256), s2 = align_up((size_t)t->sw2 * t->sh2 * k_src_esz[t->src2_dtype], 256);
    const size_t no = tri ? (size_t)t->ow * t->oh : 0;
    TileSlot L;
    L.r1 = (float*)io; L.r2 = (float*)(io + a4); L.disp = (float*)(io + 2 * a4); L.mf = (float*)(io + 3 * a4); L.err = (float*)(io + 4 * a4);
    L.mask = (uint8_t*)(io + 5 * a4); L.mask_e = L.mask + a1;
    L.s1 = (char*)(L.mask_e + a1); L.s2 = L.s1 + s1;
    L.mo = (float*)(L.s2 + s2);
    L.lla = (double*)((char*)L.mo + align_up(no * 4, 256));
    L.rpc = (s2p_rpc*)((char*)L.lla + align_up(tri ? npx * 24 : 0, 256));
    return L;
}
// windows (+ the triangulation's inputs) to the device, both images resampled into the rectified frame
static int tile_upload_rectify(s2p_hip_ctx* ctx, const s2p_tile* t, const TileSlot& L) {
    hipStream_t st = ctx->stream;
    S2P_HIP_CHECK(hipMemcpyAsync(L.s1, t->src1, (size_t)t->sw1 * t->sh1 * k_src_esz[t->src1_dtype], hipMemcpyHostToDevice, st));
    S2P_HIP_CHECK(hipMemcpyAsync(L.s2, t->src2, (size_t)t->sw2 * t->sh2 * k_src_esz[t->src2_dtype], hipMemcpyHostToDevice, st));
    if (t->rpca) {
        S2P_HIP_CHECK(hipMemcpyAsync(L.mo, t->msk_orig, (size_t)t->ow * t->oh * 4, hipMemcpyHostToDevice, st));
        S2P_HIP_CHECK(hipMemcpyAsync(L.rpc, t->rpca, sizeof(s2p_rpc), hipMemcpyHostToDevice, st));
        S2P_HIP_CHECK(hipMemcpyAsync(L.rpc + 1, t->rpcb, sizeof(s2p_rpc), hipMemcpyHostToDevice, st));
    }
    int rc = warp_enqueue(ctx, L.s1, t->src1_dtype, t->sw1, t->sh1, t->H1, L.r1, t->w, t->h, ctx->ws);
    if (rc) return rc;
    return warp_enqueue(ctx, L.s2, t->src2_dtype, t->sw2, t->sh2, t->H2, L.r2, t->w, t->h, ctx->ws);
}
// erosion, triangulation and the copies back, after the matcher has filled L.disp / L.mask
static int tile_finish(s2p_hip_ctx* ctx, const s2p_tile* t, const s2p_tile_out* o, const TileSlot& L) {
    hipStream_t st = ctx->stream;
    const bool tri = t->rpca != nullptr;
    const int w = t->w, h = t->h;
    const size_t npx = (size_t)w * h;
    int rc;
    const uint8_t* d_mfin = L.mask;
    if (t->erosion > 0) {
        rc = erode_enqueue(ctx, L.mask, w, h, t->erosion, L.mask_e);
        if (rc) return rc;
        d_mfin = L.mask_e;
    }
    if (tri) {
        hipLaunchKernelGGL(k_mask_u8_to_f32, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, st, d_mfin, npx, L.mf);
        rc = tri_enqueue(ctx, L.disp, nullptr, L.mf, w, h, L.mo, t->ow, t->oh, t->ha, t->hb, L.rpc, t->bbox, L.lla, L.err);
        if (rc) return rc;
    }
    if (o->rect1) S2P_HIP_CHECK(hipMemcpyAsync(o->rect1, L.r1, npx * 4, hipMemcpyDeviceToHost, st));
    if (o->rect2) S2P_HIP_CHECK(hipMemcpyAsync(o->rect2, L.r2, npx * 4, hipMemcpyDeviceToHost, st));
    if (o->disp) S2P_HIP_CHECK(hipMemcpyAsync(o->disp, L.disp, npx * 4, hipMemcpyDeviceToHost, st));
    if (o->mask) S2P_HIP_CHECK(hipMemcpyAsync(o->mask, d_mfin, npx, hipMemcpyDeviceToHost, st));
    if (tri && o->lonlatalt) S2P_HIP_CHECK(hipMemcpyAsync(o->lonlatalt, L.lla, npx * 24, hipMemcpyDeviceToHost, st));
    if (tri && o->err) S2P_HIP_CHECK(hipMemcpyAsync(o->err, L.err, npx * 4, hipMemcpyDeviceToHost, st));
    return S2P_HIP_OK;
}

int s2p_hip_tile_host(s2p_hip_ctx* ctx, const s2p_tile* t, const s2p_tile_out* o, double timeout_s) {
    if (!ctx || !tile_args_ok(t, o)) { if (!ctx) set_last_error("tile: bad argument"); return S2P_HIP_BAD_ARGUMENT; }
    const double deadline = timeout_s < 0 ? -1.0 : now_s() + timeout_s;
    if (timeout_s == 0) { set_last_error("timeout of 0 s: nothing was enqueued"); return S2P_HIP_TIMEOUT; }
    const int w = t->w, h = t->h;
    s2p_sgbm_params ps; s2p_census_params pc; Geom g;
    size_t match_ws;
    int rc;
    if (t->algo == 0) {
        if (t->sgbm) ps = *t->sgbm; else s2p_hip_sgbm_default_params(&ps);
        rc = make_geom(w, h, t->dmin, t->dmax, &g);
        if (rc) { set_last_error("sgbm: empty disparity range [%d, %d]", t->dmin, t->dmax); return rc; }
        rc = check_params(ps, g);
        if (rc) return rc;
        match_ws = sgbm_workspace_bytes(g, false);
    } else {
        if (t->census) pc = *t->census; else s2p_hip_census_default_params(&pc);
        rc = check_census_params(pc, w, h, t->dmin, t->dmax);
        if (rc) return rc;
        match_ws = census_workspace_bytes(pc, w, h, t->dmin, t->dmax, false);
    }
    S2P_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t io_bytes = tile_slot_bytes(t);
    const size_t scratch = std::max(match_ws, std::max(warp_workspace_bytes(t->sw1, t->sh1), warp_workspace_bytes(t->sw2, t->sh2)));
    rc = ws_reserve(ctx, scratch + io_bytes + 4096);
    if (rc) return rc;
    const TileSlot L = tile_slot_at(t, ctx->ws + ctx->ws_size - io_bytes);
    rc = tile_upload_rectify(ctx, t, L);
    if (rc) return rc;
    if (t->algo == 0) rc = sgbm_enqueue(ctx, g, ps, L.r1, L.r2, L.disp, nullptr, L.mask, false, nullptr);
    else rc = census_enqueue(ctx, pc, L.r1, L.r2, w, h, t->dmin, t->dmax, L.disp, nullptr, L.mask, false, nullptr);
    if (rc) return rc;
    rc = tile_finish(ctx, t, o, L);
    if (rc) return rc;
    return wait_stream(ctx, deadline);
}

// N tiles of one shape in one call: the N pairs go through ONE batched matcher launch sequence (census_batch_enqueue)
int s2p_hip_tile_host_batch(s2p_hip_ctx* ctx, int n, const s2p_tile* tiles, const s2p_tile_out* outs, double timeout_s) {
    if (!ctx || n <= 0 || n > 64 || !tiles || !outs) { set_last_error("tile batch: bad argument"); return S2P_HIP_BAD_ARGUMENT; }
    if (n == 1) return s2p_hip_tile_host(ctx, tiles, outs, timeout_s);
    s2p_census_params pc;
    if (tiles[0].census) pc = *tiles[0].census; else s2p_hip_census_default_params(&pc);
    bool uniform = true;
    std::vector<int> tw(n), th(n), tlo(n), thi(n);
    for (int k = 0; k < n; k++) {
        const s2p_tile* t = tiles + k;
        if (!tile_args_ok(t, outs + k)) return S2P_HIP_BAD_ARGUMENT;
        s2p_census_params pk;
        if (t->census) pk = *t->census; else s2p_hip_census_default_params(&pk);
        if (t->algo != 1 || memcmp(&pk, &pc, sizeof(pc)) != 0) {
            set_last_error("tile batch: tile %d differs from tile 0 in matcher or parameters (census / SGM tiles of one parameter set only)", k);
            return S2P_HIP_BAD_ARGUMENT;
        }
        if (t->w != tiles[0].w || t->h != tiles[0].h || t->dmin != tiles[0].dmin || t->dmax != tiles[0].dmax) uniform = false;
        tw[k] = t->w; th[k] = t->h; tlo[k] = t->dmin; thi[k] = t->dmax;
    }
    if (!uniform && (n > 16 || !census_batches_hetero(pc, n, tw.data(), th.data()))) {
        set_last_error("tile batch: tiles of different sizes / ranges share a call only in the single-scale MGM modes with P2 <= 115, 16 at most "
                       "(group the others by shape on the caller's side)");
        return S2P_HIP_BAD_ARGUMENT;
    }
    const double deadline = timeout_s < 0 ? -1.0 : now_s() + timeout_s;
    if (timeout_s == 0) { set_last_error("timeout of 0 s: nothing was enqueued"); return S2P_HIP_TIMEOUT; }
    const int w = tiles[0].w, h = tiles[0].h, dmin = tiles[0].dmin, dmax = tiles[0].dmax;
    int rc = S2P_HIP_OK;
    double cand = 0;
    int Dmax = 0;
    for (int k = 0; k < n; k++) {
        rc = check_census_params(pc, tw[k], th[k], tlo[k], thi[k]);
        if (rc) return rc;
        Dmax = std::max(Dmax, census_D(pc, tlo[k], thi[k]));
    }
    for (int k = 0; k < n; k++) cand += (double)tw[k] * th[k] * Dmax;
    if (cand * 9.0 > 6.0e10) { set_last_error("tile batch: more than 60 GB of volumes; use smaller batches"); return S2P_HIP_UNSUPPORTED; }
    S2P_HIP_CHECK(hipSetDevice(ctx->device));
    size_t scratch = uniform ? census_batch_workspace_bytes(pc, n, w, h, dmin, dmax)
                             : census_batch_hetero_workspace_bytes(pc, n, tw.data(), th.data(), tlo.data(), thi.data()), io_total = 0;
    for (int k = 0; k < n; k++) {
        scratch = std::max(scratch, std::max(warp_workspace_bytes(tiles[k].sw1, tiles[k].sh1), warp_workspace_bytes(tiles[k].sw2, tiles[k].sh2)));
        io_total += tile_slot_bytes(tiles + k);
    }
    rc = ws_reserve(ctx, scratch + io_total + 4096);
    if (rc) return rc;
    std::vector<TileSlot> L(n);
    std::vector<const float*> im1(n), im2(n);
    std::vector<float*> disp(n);
    std::vector<uint8_t*> mask(n);
    char* io = ctx->ws + ctx->ws_size - io_total;
    for (int k = 0; k < n; k++) {
        L[k] = tile_slot_at(tiles + k, io);
        io += tile_slot_bytes(tiles + k);
        rc = tile_upload_rectify(ctx, tiles + k, L[k]);
        if (rc) return rc;
        im1[k] = L[k].r1; im2[k] = L[k].r2; disp[k] = L[k].disp; mask[k] = L[k].mask;
    }
    rc = uniform ? census_batch_enqueue(ctx, pc, n, im1.data(), im2.data(), w, h, dmin, dmax, disp.data(), nullptr, mask.data())
                 : census_batch_hetero_enqueue(ctx, pc, n, im1.data(), im2.data(), tw.data(), th.data(), tlo.data(), thi.data(), disp.data(), nullptr, mask.data());
    if (rc) return rc;
    for (int k = 0; k < n; k++) {
        rc = tile_finish(ctx, tiles + k, outs + k, L[k]);
        if (rc) return rc;
    }
    return wait_stream(ctx, deadline);
}

int s2p_hip_timing_enable(s2p_hip_ctx* ctx, int on) {
    if (!ctx) return S2P_HIP_BAD_ARGUMENT;
    ctx->timing = on != 0;
    return S2P_HIP_OK;
}
int s2p_hip_timing_reset(s2p_hip_ctx* ctx) {
    if (!ctx) return S2P_HIP_BAD_ARGUMENT;
    int rc = timing_collect(ctx);
    ctx->stages.clear();
    return rc;
}
int s2p_hip_timing_get(s2p_hip_ctx* ctx, const char* stage, double* ms, int* launches) {
    if (!ctx || !stage) return S2P_HIP_BAD_ARGUMENT;
    int rc = timing_collect(ctx);
    if (rc) return rc;
    auto it = ctx->stages.find(stage);
    if (ms) *ms = it == ctx->stages.end() ? 0.0 : it->second.ms;
    if (launches) *launches = it == ctx->stages.end() ? 0 : it->second.launches;
    return S2P_HIP_OK;
}

}  // extern "C"
