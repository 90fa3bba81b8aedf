// eval_multibody.h — per-knot evaluation of multibody stages (placeholder until the kernel lands).
#pragma once
#include <stdexcept>
#include "eval_common.h"

static inline void check_multibody_model(const int32_t*, int) {}
static inline size_t multibody_work_doubles(const Layout&) { return 0; }
static inline void launch_eval_multibody(hipStream_t, const SolverArgs&, const Layout&, double*, double*, size_t, bool) {
  throw std::runtime_error("multibody stage evaluation kernel is not built into this library yet");
}
