// oracle/cpu_port/capi_port.cpp — MEASUREMENT INFRASTRUCTURE: libmpc_cpu.so, the CPU port bench.py times as its cpu_baseline
// (kind "port").  The C-ABI of include/mpc_abi.h on top of the oracle's solver (oracle/solver.hpp: BCL / linesearch / proximal
// Riccati, serial and in legs, OpenMP over knots and legs) with the whole-body stage evaluation swapped for the closed-form port of
// the HIP stage kernel (eval_closed_form.hpp) and everything compiled -O3 -march=native.  Not the checker (that is
// ../libmpc_oracle.so with forward-mode AD) and never loaded by the product.
#include "eval_closed_form.hpp"
#define ORC_EVAL_MULTIBODY cpu_port::eval_multibody_cf
#define ORC_BACKEND_NAME "cpu-port"
#include "../capi.cpp"
