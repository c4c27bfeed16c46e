// api.cpp -- error reporting and version of libaabr_hip.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include "../../include/aabr_hip.h"

namespace aabr {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

enum Knob { K_CONV_WLDS, K_CONV_SMALL, K_CONV_NBW, K_CONV_WPB, K_CONV_RS, K_RS_UNIT, K_WIDE_ROWS, K_CONV_WIDE,
            K_WIDE_NBUF, K_CONV_WIDE_BF16, K_VOXEL_MEAN, K_WIDE_NCB, K_BN_SMALL, K_CONV_X3, K_X3_FORM, K_WIDE_PRIO, K_PLAN_SIDE_BATCH, K_PLAN_SIDE_PRIO, K_COUNT };
constexpr int kKnobUnset = -2147483647 - 1;
static const char *const g_knob_names[K_COUNT] = {"CONV_WLDS", "CONV_SMALL", "CONV_NBW", "CONV_WPB", "CONV_RS", "RS_UNIT",
                                                  "WIDE_ROWS", "CONV_WIDE", "WIDE_NBUF", "CONV_WIDE_BF16", "VOXEL_MEAN", "WIDE_NCB", "BN_SMALL", "CONV_X3", "X3_FORM", "WIDE_PRIO", "PLAN_SIDE_BATCH", "PLAN_SIDE_PRIO"};
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
extern "C" int aabr_version(void) { return 100; }
