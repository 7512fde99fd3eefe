// tools/probes/issue_rate.hip -- how many instructions a gfx950 SIMD issues per cycle, by class and by waves per SIMD (dev probe, not shipped).
// The band kernel's steps are in-order chains of ~150 instructions (VALU 60 %, SALU + branches 30 %, LDS / VMEM / waits the rest); whether
// the scalar instructions of one wave ride beside the vector instructions of ANOTHER wave of the same SIMD decides what removing them buys.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/issue_rate tools/probes/issue_rate.hip && tools/probes/issue_rate
// Every workgroup is 256 threads = one wave per SIMD of its CU; `wps` workgroups per CU are made resident by the LDS they ask for.
// Each wave times its own loop with s_memtime (shader clock) and reports instructions per cycle; the table prints the mean per wave and
// the sum per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

enum { M_VALU = 0, M_SALU, M_MIXED_IN_WAVE, M_PK, M_DPP, M_VALU_DEP, M_SPLIT_WAVES, M_SNOP, M_BRANCH, M_LDS, M_SPLIT_PK_SALU, NMODES };
static const char* mode_name[NMODES] = {
    "v_add_u32, 4 independent chains", "s_add_u32, 4 independent chains", "v_add / s_add alternating in ONE wave", "v_pk_add_u16, 4 chains",
    "v_mov_dpp row_shr:1, 4 chains", "v_add_u32, ONE dependent chain", "odd waves v_add, even waves s_add (same SIMD)", "s_nop 0",
    "s_cbranch_scc0 (not taken) + s_cmp", "ds_read_b128 (independent)", "odd waves v_pk_add, even waves s_add + s_cbranch"};

template <int MODE>
__global__ __launch_bounds__(256) void k_issue(unsigned long long* out, int iters)
{
    extern __shared__ uint32_t lds[];
    const int wave = threadIdx.x >> 6;
    uint32_t a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3;
    uint32_t sa = blockIdx.x, sb = sa + 1, sc = sa + 2, sd = sa + 3;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    int n_per_iter = 64;
    // waves of a workgroup land on the 4 SIMDs of a CU (one each); with several workgroups per CU the waves of a SIMD come from
    // different workgroups, so "odd / even" below is by workgroup index
    const bool odd = blockIdx.x & 1;
    for (int it = 0; it < iters; it++) {
        if (MODE == M_VALU) { asm volatile(REP16("v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
        else if (MODE == M_SALU) { asm volatile(REP16("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n") : "+s"(sa), "+s"(sb), "+s"(sc), "+s"(sd) : : "scc"); }
        else if (MODE == M_MIXED_IN_WAVE) { asm volatile(REP16("v_add_u32 %0, %0, 1\n s_add_u32 %2, %2, 1\n v_add_u32 %1, %1, 1\n s_add_u32 %3, %3, 1\n") : "+v"(a), "+v"(b), "+s"(sa), "+s"(sb) : : "scc"); }
        else if (MODE == M_PK) { asm volatile(REP16("v_pk_add_u16 %0, %0, %0\n v_pk_add_u16 %1, %1, %1\n v_pk_add_u16 %2, %2, %2\n v_pk_add_u16 %3, %3, %3\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
        else if (MODE == M_DPP) { asm volatile(REP16("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
        else if (MODE == M_VALU_DEP) { asm volatile(REP64("v_add_u32 %0, %0, 1\n") : "+v"(a)); }
        else if (MODE == M_SPLIT_WAVES) {
            if (odd) asm volatile(REP16("v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            else asm volatile(REP16("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n") : "+s"(sa), "+s"(sb), "+s"(sc), "+s"(sd) : : "scc");
        }
        else if (MODE == M_SNOP) { asm volatile(REP64("s_nop 0\n")); }
        else if (MODE == M_BRANCH) { asm volatile(REP16("s_cmp_eq_u32 %0, %0\n s_cbranch_scc0 1f\n s_cmp_eq_u32 %1, %1\n s_cbranch_scc0 1f\n") "1:\n" : "+s"(sa), "+s"(sb) : : "scc"); }
        else if (MODE == M_LDS) {
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            u32x4 r0, r1, r2, r3;
            const uint32_t addr = (threadIdx.x & 63) * 16 + wave * 1024;
            n_per_iter = 16;
            asm volatile(REP4("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:4096\n ds_read_b128 %2, %4\n ds_read_b128 %3, %4 offset:4096\n") "s_waitcnt lgkmcnt(0)\n"
                         : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(addr));
            a += r0.x + r1.x + r2.x + r3.x;
        }
        else if (MODE == M_SPLIT_PK_SALU) {
            if (odd) asm volatile(REP16("v_pk_add_u16 %0, %0, %0\n v_pk_add_u16 %1, %1, %1\n v_pk_add_u16 %2, %2, %2\n v_pk_add_u16 %3, %3, %3\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            else asm volatile(REP16("s_add_u32 %0, %0, 1\n s_cmp_eq_u32 %1, %1\n s_cbranch_scc0 1f\n s_add_u32 %1, %1, 1\n") "1:\n" : "+s"(sa), "+s"(sb) : : "scc");
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * 4 + wave;
        out[2 * w] = t1 - t0;
        out[2 * w + 1] = ((unsigned long long)(a + b + c + d + sa + sb + sc + sd) & 1ull) | ((unsigned long long)iters * n_per_iter << 1) | ((unsigned long long)odd << 63);
    }
}

template <int MODE>
static void run(int ncu, int wps, unsigned long long* d_out, std::vector<unsigned long long>& h)
{
    const int nblk = ncu * wps, iters = 2000;
    // wps workgroups per CU: each asks for just under 160 KB / wps of LDS, so exactly wps fit
    size_t shm = (size_t)(160 * 1024 / wps) - 2048;
    if (shm > 64 * 1024) { hipFuncSetAttribute((const void*)k_issue<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); }
    hipLaunchKernelGGL((k_issue<MODE>), dim3(nblk), dim3(256), shm, 0, d_out, iters);
    hipLaunchKernelGGL((k_issue<MODE>), dim3(nblk), dim3(256), shm, 0, d_out, iters);
    hipMemcpy(h.data(), d_out, (size_t)nblk * 4 * 16, hipMemcpyDeviceToHost);
    double sum_ipc[2] = {0, 0}; int cnt[2] = {0, 0};
    for (int w = 0; w < nblk * 4; w++) {
        const double cyc = (double)h[2 * w], n = (double)((h[2 * w + 1] & 0x7fffffffffffffffull) >> 1);
        const int o = (int)(h[2 * w + 1] >> 63);
        sum_ipc[o] += n / cyc; cnt[o]++;
    }
    const double per_wave = (sum_ipc[0] + sum_ipc[1]) / (cnt[0] + cnt[1]);
    printf("  %-52s %d wave(s)/SIMD: %.3f instr/cycle per wave, %.3f per SIMD", mode_name[MODE], wps, per_wave, per_wave * wps);
    if (MODE == M_SPLIT_WAVES || MODE == M_SPLIT_PK_SALU) printf("   [vector waves %.3f, scalar waves %.3f per wave]", cnt[1] ? sum_ipc[1] / cnt[1] : 0.0, cnt[0] ? sum_ipc[0] / cnt[0] : 0.0);
    printf("\n");
}

int main()
{
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int ncu = pr.multiProcessorCount;
    printf("issue rates on %d CUs (cycles = s_memtime shader clock); a workgroup = 4 waves = one per SIMD\n", ncu);
    unsigned long long* d_out; hipMalloc(&d_out, (size_t)ncu * 8 * 4 * 16);
    std::vector<unsigned long long> h((size_t)ncu * 8 * 4 * 2);
    for (int wps : {1, 2, 4}) {
        run<M_VALU>(ncu, wps, d_out, h); run<M_VALU_DEP>(ncu, wps, d_out, h); run<M_PK>(ncu, wps, d_out, h); run<M_DPP>(ncu, wps, d_out, h);
        run<M_SALU>(ncu, wps, d_out, h); run<M_SNOP>(ncu, wps, d_out, h); run<M_BRANCH>(ncu, wps, d_out, h); run<M_LDS>(ncu, wps, d_out, h);
        run<M_MIXED_IN_WAVE>(ncu, wps, d_out, h);
        if (wps > 1) { run<M_SPLIT_WAVES>(ncu, wps, d_out, h); run<M_SPLIT_PK_SALU>(ncu, wps, d_out, h); }
        printf("\n");
    }
    return 0;
}
