// api.cpp -- error reporting and version of libaabr_hip.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include "common.h"

namespace aabr {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

#define AABR_KNOB_NAME(n) #n,
static const char *const g_knob_names[] = {AABR_KNOB_LIST(AABR_KNOB_NAME)};
#undef AABR_KNOB_NAME
static_assert(sizeof(g_knob_names) / sizeof(*g_knob_names) == K_COUNT, "one name per knob");
static std::atomic<int> g_knob_val[K_COUNT];
static std::atomic<int> g_knob_state[K_COUNT];   // 0: environment not read yet, 1: value final
int knob(Knob k) {
  if (g_knob_state[k].load(std::memory_order_acquire) == 0) {
    char name[64];
    snprintf(name, sizeof(name), "AABR_%s", g_knob_names[k]);
    const char *v = getenv(name);
    g_knob_val[k].store(v && *v ? atoi(v) : kKnobUnset, std::memory_order_relaxed);
    g_knob_state[k].store(1, std::memory_order_release);
  }
  return g_knob_val[k].load(std::memory_order_relaxed);
}
} // namespace aabr

extern "C" int aabr_set_knob(const char *name, int value, int unset) {
  if (!name) { aabr::set_error("aabr_set_knob: null name"); return AABR_EINVAL; }
  for (int k = 0; k < aabr::K_COUNT; ++k)
    if (strcmp(name, aabr::g_knob_names[k]) == 0) {
      aabr::g_knob_val[k].store(unset ? aabr::kKnobUnset : value, std::memory_order_relaxed);
      aabr::g_knob_state[k].store(1, std::memory_order_release);
      return AABR_OK;
    }
  aabr::set_error("aabr_set_knob: unknown knob %s", name);
  return AABR_EINVAL;
}

extern "C" const char *aabr_last_error(void) { return aabr::g_err; }
extern "C" int aabr_version(void) { return AABR_ABI_VERSION; }
// bit 0: built with `make DEV=1` -- the timing-experiment variants and the measured-slower A/B kernels (row-stationary
// bf16 convolution, fp32 on the bf16 pipe by a three-term split, the four-waves-per-SIMD form of k_conv_cs) are in
extern "C" int aabr_build_flags(void) {
#ifdef AABR_DEV
  return 1;
#else
  return 0;
#endif
}
