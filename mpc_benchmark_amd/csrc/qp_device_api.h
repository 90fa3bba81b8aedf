// qp_device_api.h — what the two translation units of libmpc_hip.so share about a QP handle (qp.hip owns mpc_qp_solver; mpc_hip.hip strings the
// inverse-dynamics QP between the plan's feedback terms and the simulator step in mpc_qp_low_level_steps, pipeline_glue.h).  Internal: not part of the C-ABI.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/mpc_qp_abi.h"

struct QpIdBuffers {
  hipStream_t stream;
  double *xrob, *acc, *f;   // inputs of the assembly kernel: [B][nq + nv], [B][nv], [B][6 nk]
  int32_t* cs;              // [B][nk]
  double* sol;              // [B][n] = (da, df, tau)
  mpc_qp_info* info;        // [B]
  int B, n, nq, nv, nk, device;
};
void qp_id_prepare(mpc_qp_solver* s, int32_t nk, const int32_t* frames, const double* weights, const double* cone);  // throws
QpIdBuffers qp_id_buffers(mpc_qp_solver* s);
void qp_id_enqueue(mpc_qp_solver* s, const mpc_qp_settings* S, double kd);   // assembly (+ zeroed start unless warm_start) on the handle's stream
void qp_launch_solve(mpc_qp_solver* s, const mpc_qp_settings* S);            // the solve kernel on the handle's stream
double* qp_scratch(mpc_qp_solver* s, size_t doubles);                         // device scratch owned by the handle
void qp_set_error(mpc_qp_solver* s, const char* what);
