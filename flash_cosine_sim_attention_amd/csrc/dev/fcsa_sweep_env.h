// dev/fcsa_sweep_env.h -- SWEEP BUILDS ONLY (-DFCSA_VAR_SPLIT_ENV, tools/split_sweep.py): launch parameters from the environment, read
// per call, so that one process can time every split count / form of a shape.  The product build never includes this file: no entry
// point of libfcsa_hip.so reads the environment (tests/test_cabi_cpu.py scans the product sources for it).
#pragma once
#include <cstdlib>
namespace fcsa_dev {
inline int env_int(const char* name) {          // -1: not set
  const char* e = std::getenv(name);
  return e == nullptr ? -1 : std::atoi(e);
}
}  // namespace fcsa_dev
