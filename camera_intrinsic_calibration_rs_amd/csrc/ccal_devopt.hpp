// Developer switches.  The PRODUCT library (libccal_hip.so) reads three environment variables a user may need - CCAL_RCCL_LIB
// (ccal_rccl.hip), CCAL_MULTI_TRANSPORT (ccal_multi.hip) and, in the Python binding, CCAL_LIB - and nothing else: every A/B
// switch below is compiled out (dev_env is a constant NULL, the branches behind it disappear).  The SECOND build of the library
// (libccal_hip_legacy.so: -DCCAL_DEV_SWITCHES -DCCAL_LEGACY_KERNELS -DCCAL_TEST_HOOKS; tests and A/B tools load it by name)
// reads them from the environment: they select among implementations that the parity tests hold to the same results.
// All call sites live in the three translation units that are compiled twice (ccal_kernels_fused, ccal_kernels_normal,
// ccal_solver); choices that concern other translation units travel as arguments (FusedArgs::lpf_force, NormalWs::schurq_slots).
#pragma once
#include <cstdlib>

namespace ccal {
#ifdef CCAL_DEV_SWITCHES
inline const char* dev_env(const char* name) { return std::getenv(name); }
#else
constexpr const char* dev_env(const char*) { return nullptr; }
#endif
inline int dev_env_int(const char* name, int dflt) { const char* e = dev_env(name); return e ? std::atoi(e) : dflt; }
}  // namespace ccal
