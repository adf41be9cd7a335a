// bb_inst.h -- lookup of the resident kernels' template instances.  The instances (about 140 of them, ~1.5 s of hipcc each) are
// spread over several translation units (bb_inst_*.hip) that __graft_entry__.build_hip compiles in parallel; bb_engine.hip only
// calls these functions.  (Experiment builds, -DBB_FAST_BUILD, keep a handful of instances inside bb_engine.hip itself.)
#pragma once
#include "bb_resident.h"
#include "bb_stream.h"

#ifndef BB_EMU
typedef void (*bb_persist_kernel)(const DevModel*, const DevState*, const BBLds*, RunArgs, int, int);
typedef void (*bb_res_kernel)(const DevModel*, const DevState*, const BRLay*, RunArgs, int, int);
#define BB_INST __attribute__((visibility("hidden")))
// (nm, where given: the selected instance as text, template arguments in declaration order -- bb_kernel_name)
BB_INST bb_persist_kernel bb_persist_instance(int kind, int P, int nthr, bool xg, const char** nm = nullptr);
// k_res<KIND, P, NT, XG, TT, AP, MS>: T = the model's common time-point count (0: replicates differ), compile-time in the BASELINE
// shapes; ap = br_any_parity(model); ms = several MC samples per step and / or ELBO recording (generic-T instances; round 4: the cross-GPU ones too)
BB_INST bb_res_kernel bb_res_instance_k0(int P, int nthr, bool xg, int T, bool ap, bool ms, const char** nm = nullptr);
BB_INST bb_res_kernel bb_res_instance_k1(int P, int nthr, bool xg, int T, bool ap, bool ms, const char** nm = nullptr);
BB_INST bb_res_kernel bb_res_instance_k2(int P, int nthr, bool xg, int T, bool ap, bool ms, const char** nm = nullptr);
BB_INST bb_res_kernel bb_res_instance_k3(int P, int nthr, bool xg, int T, bool ap, bool ms, const char** nm = nullptr);
BB_INST bb_res_kernel bb_res_instance_k4(int P, int nthr, bool xg, int T, bool ap, bool ms, const char** nm = nullptr);
// k_stream<KIND, NT, TT>: all five kinds, 1024 or 512 threads, T = 8, 6 or 4 (every replicate the same)
BB_INST bb_stream_kernel bb_stream_instance(int kind, int nthr, int T, const char** nm = nullptr);
// ... their MS form (several MC samples per step and / or the ELBO trace): k_stream<KIND, NT, TT, true>, T = 8 or 6
BB_INST bb_stream_kernel bb_stream_instance_ms(int kind, int nthr, int T, const char** nm = nullptr);
#endif
