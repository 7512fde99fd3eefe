// s2p_amd/csrc/probe_guard.hpp -- the quarantine of probe builds.  Included first by common.hpp.
//
// The kernels carry build switches that exist to MEASURE, not to ship: timing probes whose results are invalid (no cost loads, every
// 4th e-store dropped, lattices skipped ...), trace instrumentation, and tunables whose shipped values were chosen by measurement.
// Any of them given on the command line makes the build a PROBE BUILD, and a probe build must say so:
//   * it must be compiled with -DS2P_PROBE_BUILD="\"<flags>\"" -- s2p_amd/build.py adds that by itself whenever
//     S2P_HIP_EXTRA_FLAGS is set, and then writes build/variants/libs2p_hip_<name>.so, NEVER s2p_amd/lib/libs2p_hip.so; a switch
//     without the umbrella is a compile error (below), so no hand-run hipcc line can produce an unmarked probe library;
//   * s2p_hip_build_info() answers "... PROBE BUILD [<flags>] ...", every s2p_hip_last_error() message starts with "[PROBE BUILD] ",
//     the library carries the symbol s2p_hip_probe_build_marker, and s2p_amd/_lib.py refuses to load such a library from the shipped
//     path (tests/test_abi.py checks the shipped .so for all three).
#pragma once

// results-invalid probes and instrumentation (defined or not)
#if defined(S2P_MGM_PROBE_NOMEM) || defined(S2P_MGM_PROBE_NOPOLL) || defined(S2P_MGM_TRACE) || defined(S2P_WARP_NOCHAIN) || \
    defined(S2P_MGM_PROBE_DVALID) || defined(S2P_MGM_PROBE_FULLSTORE)
#define S2P_PROBE_SWITCH_SEEN 1
#endif
// tunables of the other kernels: the sources define them when the command line does not
#if defined(S2P_WTA_PF) || defined(S2P_WTA_NT) || defined(S2P_MGM_DEFAULT_BANDS) || defined(S2P_MGM_BATCH_STAGGER) || \
    defined(S2P_COST_KILLMASK) || defined(S2P_E_STORE_AUX) || defined(S2P_E_LOAD_AUX) || defined(S2P_C_LOAD_AUX) || defined(S2P_AGG_PF) || \
    defined(S2P_CENSUS_DEPTH16) || defined(S2P_MGM_LDS_SKEW) || defined(S2P_MGM_E_PLAIN_UPTO)
#define S2P_PROBE_SWITCH_SEEN 1
#endif
// switches that rounds 1-5 carried and round 6 removed (their verdicts: docs/notebook/10_round6_switches.md): naming one is an error,
// not a silent no-op
#if defined(S2P_MGM_PROBE_NO_C) || defined(S2P_MGM_PROBE_NO_E) || defined(S2P_PROBE_E34) || defined(S2P_MGM_IL4_PROBE) || \
    defined(S2P_MGM_ONLY_AXIS) || defined(S2P_MGM_ONLY_Q0) || defined(S2P_MGM_ONLY_DIAG) || defined(S2P_MGM_PROBE_NOP) || \
    defined(S2P_MGM_PROBE_VMOV) || defined(S2P_PROBE_FAKE_CONF) || defined(S2P_MGM_FPRIO) || defined(S2P_MGM_PF) || \
    defined(S2P_MGM_PROLOGUE_STORES) || defined(S2P_MGM_INNER) || defined(S2P_MGM_PRIO) || defined(S2P_MGM_SLEEP) || \
    defined(S2P_HANDOFF_ST_AUX) || defined(S2P_HANDOFF_LD_AUX) || defined(S2P_MGM_LEAD) || defined(S2P_MGM_NW_WIDE) || \
    defined(S2P_MGM_NW_NARROW) || defined(S2P_MGM_NW_G64) || defined(S2P_MGM_NW_G32) || defined(S2P_MGM_NW_BATCH_G16) || \
    defined(S2P_MGM_ORDER) || defined(S2P_MGM_FSLEEP) || defined(S2P_MGM_RING4_FROM) || defined(S2P_MGM_RING16_UPTO) || \
    defined(S2P_MGM_TRIG) || defined(S2P_MGM_WORKERS_MAX) || defined(S2P_MGM_WORKERS_1) || defined(S2P_MGM_K8_FROM)
#error "this switch was removed in round 6 (its verdict is final: docs/notebook/10_round6_switches.md)"
#endif

#if defined(S2P_PROBE_SWITCH_SEEN) && !defined(S2P_PROBE_BUILD)
#error "a probe / tuning switch was given without S2P_PROBE_BUILD: build through s2p_amd/build.py with S2P_HIP_EXTRA_FLAGS (the result goes to build/variants/, never to s2p_amd/lib/)"
#endif
