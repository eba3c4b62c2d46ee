// Host-only unit test of fcsa_kernels.h's per-device bookkeeping (ensure_dynamic_lds, cu_count) against STUBBED HIP runtime entry
// points: the per-device dynamic-LDS path can only execute for real on a node with several GPUs, which a 1-GPU lease never shows
// (round 4 review).  Built and run by tests/test_dynamic_lds_mask_cpu.py with g++; not part of the library.
#include <cstdio>
#include <vector>
#include "fcsa_kernels.h"

static int g_device = 0, g_cus = 256, g_fail_device = -1, g_get_device_fails = 0;
static std::vector<int> g_set_calls;      // device of every hipFuncSetAttribute call

extern "C" hipError_t hipGetDevice(int* d) { if (g_get_device_fails) return hipErrorNoDevice; *d = g_device; return hipSuccess; }
extern "C" hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute attr, int) {
  if (attr != hipFuncAttributeMaxDynamicSharedMemorySize) return hipErrorInvalidValue;
  g_set_calls.push_back(g_device);
  return g_device == g_fail_device ? hipErrorInvalidValue : hipSuccess;
}
extern "C" hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t a, int) {
  if (g_get_device_fails) return hipErrorNoDevice;
  if (a != hipDeviceAttributeMultiprocessorCount) return hipErrorInvalidValue;
  *v = g_cus; return hipSuccess;
}
extern "C" hipError_t hipGetLastError(void) { return hipSuccess; }

static void kernel_stub() {}
#define CHECK(c) do { if (!(c)) { std::printf("FAIL line %d: %s\n", __LINE__, #c); return 1; } } while (0)

int main() {
  using fcsa::ensure_dynamic_lds;
  std::atomic<uint64_t> done{0};
  // first launch on device 0 sets the attribute, the second does not
  g_device = 0;
  CHECK(ensure_dynamic_lds(&kernel_stub, 100000, done) == hipSuccess && g_set_calls.size() == 1 && done.load() == 1ull);
  CHECK(ensure_dynamic_lds(&kernel_stub, 100000, done) == hipSuccess && g_set_calls.size() == 1);
  // another device of the same process: the attribute is raised THERE too, once; device 0 stays marked
  g_device = 5;
  CHECK(ensure_dynamic_lds(&kernel_stub, 100000, done) == hipSuccess && g_set_calls.size() == 2 && g_set_calls.back() == 5);
  CHECK(done.load() == ((1ull << 5) | 1ull));
  CHECK(ensure_dynamic_lds(&kernel_stub, 100000, done) == hipSuccess && g_set_calls.size() == 2);
  g_device = 0;
  CHECK(ensure_dynamic_lds(&kernel_stub, 100000, done) == hipSuccess && g_set_calls.size() == 2);
  // the highest device the mask can hold
  g_device = 63;
  CHECK(ensure_dynamic_lds(&kernel_stub, 100000, done) == hipSuccess && g_set_calls.size() == 3 && (done.load() >> 63) == 1ull);
  CHECK(ensure_dynamic_lds(&kernel_stub, 100000, done) == hipSuccess && g_set_calls.size() == 3);
  // devices the mask cannot hold: set on every launch, mask untouched
  const uint64_t before = done.load();
  g_device = 64;
  CHECK(ensure_dynamic_lds(&kernel_stub, 100000, done) == hipSuccess && g_set_calls.size() == 4);
  CHECK(ensure_dynamic_lds(&kernel_stub, 100000, done) == hipSuccess && g_set_calls.size() == 5 && done.load() == before);
  // a failing hipFuncSetAttribute is reported and NOT remembered: the next launch tries again
  g_device = 7; g_fail_device = 7;
  CHECK(ensure_dynamic_lds(&kernel_stub, 100000, done) == hipErrorInvalidValue && (done.load() & (1ull << 7)) == 0);
  g_fail_device = -1;
  CHECK(ensure_dynamic_lds(&kernel_stub, 100000, done) == hipSuccess && (done.load() & (1ull << 7)) != 0);
  // an independent instantiation has its own mask
  std::atomic<uint64_t> other{0};
  g_device = 0;
  const size_t n = g_set_calls.size();
  CHECK(ensure_dynamic_lds(&kernel_stub, 65536, other) == hipSuccess && g_set_calls.size() == n + 1 && other.load() == 1ull);
  // hipGetDevice failing is passed through
  g_get_device_fails = 1;
  CHECK(ensure_dynamic_lds(&kernel_stub, 65536, other) == hipErrorNoDevice);
  // cu_count: a host without a device answers 256 and does NOT remember it; a device's count is cached per device id
  // (tests/test_dynamic_lds_mask_cpu.py runs this binary a second time with FCSA_STUB_CUS)
  if (const char* e = std::getenv("FCSA_STUB_CUS")) {
    g_get_device_fails = 0; g_device = 0; g_cus = std::atoi(e); CHECK(fcsa::cu_count() == g_cus);
    g_cus = 1; CHECK(fcsa::cu_count() == std::atoi(e));                  // device 0: cached
    g_device = 3; g_cus = 64; CHECK(fcsa::cu_count() == 64);            // another device of the process: its own answer
    g_device = 0; CHECK(fcsa::cu_count() == std::atoi(e));
  } else {
    CHECK(fcsa::cu_count() == 256);                                      // no device: the fallback ...
    g_get_device_fails = 0; g_device = 0; g_cus = 304; CHECK(fcsa::cu_count() == 304);      // ... is not cached
    g_get_device_fails = 1; CHECK(fcsa::cu_count() == 256);
  }
  std::printf("OK %zu set-attribute calls\n", g_set_calls.size());
  return 0;
}
