// knobs.hpp -- lab-bench switches of the kernel dispatch (tile overrides, "run without fusion X", ...).
//
// A shipped libsummertts_hip.so reads exactly four environment variables, all documented in INTEGRATION.md:
//   STS_CONV_MATH (engine.hip), SUMMERTTS_HIP_DEVICE and SUMMERTTS_FRONTEND_LIB (synthesizer_trn.hip), and STS_TEST_HOOKS (multi.hip: the
//   gate of the test-only entry sts_multi_set_rccl_library).
// Everything below exists only in a build with -DSTS_EXPERIMENTS (`make -C summertts_amd/csrc exp`, tools/bench_variants.sh);
// in the default build the three helpers are constants, the compiler folds every knob away and the variable names do not
// even appear in the binary (tests/test_abi_cpu.py checks `strings`).
#pragma once
#include <stdlib.h>

namespace sts {
#ifdef STS_EXPERIMENTS
inline const char* exp_env(const char* name) { return getenv(name); }
#else
inline const char* exp_env(const char*) { return nullptr; }
#endif
inline int exp_int(const char* name, int dflt) { const char* v = exp_env(name); return v ? atoi(v) : dflt; }
inline bool exp_flag(const char* name) { return exp_env(name) != nullptr; }
}  // namespace sts
