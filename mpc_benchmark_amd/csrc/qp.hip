// qp.hip — translation unit of the batched dense QP solver (include/mpc_qp_abi.h): kernels in qp_kernel.h and
// qp_assemble.h, host side in qp_host.h.  Part of libmpc_hip.so.
#include <hip/hip_runtime.h>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#define HIP_OK(expr)                                                                                  \
  do {                                                                                                \
    hipError_t e_ = (expr);                                                                           \
    if (e_ != hipSuccess) throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

#include "qp_host.h"
