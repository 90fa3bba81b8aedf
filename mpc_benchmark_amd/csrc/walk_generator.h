// walk_generator.h — the swing-foot reference generator of the walking loops for one robot, as plain functions both libraries compile
// (the HIP kernel k_walk_refs in solver_kernels.h, and oracle/capi.cpp with DEV = static inline — like se3_math.h): forward kinematics of a
// frame from the model tables, the foothold rules of talos_utils.py:210-246 (footTrajectory.updateTrajectory), the swing curve of
// talos_utils.py:276-317 (defineBezier / foot_trajectory).  The numpy generator of mpc_benchmark_amd/references.py (FootTrajectoryBatch), which the
// drop-in fixtures hold to the reference's own code, is what these are tested against (tests/test_walk_generator.py).
#pragma once

// placement of model frame `frame` at configuration q (free-flyer root: p, quaternion x y z w ; then one angle per revolute joint)
DEV void walk_frame_placement(const int32_t* mi, const double* md, const double* q, int frame, M3& R, V3& p) {
  const int nj = mi[0];
  const int32_t* jw = mi + MPC_MODEL_HEADER_WORDS;
  const double* jd = md + MPC_MODEL_HEADER_DOUBLES;
  const int joint = jw[MPC_MODEL_JOINT_WORDS * nj + frame];
  const double* fd = jd + MPC_MODEL_JOINT_DOUBLES * nj + MPC_MODEL_FRAME_DOUBLES * frame;
  R = ldm3(fd); p = ldv3(fd + 9);  // frame placement in its joint
  for (int i = joint; i >= 0; i = jw[MPC_MODEL_JOINT_WORDS * i]) {  // compose towards the root: oMf = placement_i * joint_i(q) * (...)
    const int kind = jw[MPC_MODEL_JOINT_WORDS * i + 1], iq = jw[MPC_MODEL_JOINT_WORDS * i + 2];
    M3 Rj;
    V3 pj = v3(0, 0, 0);
    if (kind == MPC_JOINT_FREEFLYER) { Rj = quat_to_rot(q + iq + 3); pj = v3(q[iq], q[iq + 1], q[iq + 2]); }
    else {
      const double th = q[iq], cs = cos(th), sn = sin(th);
      const int ax = kind - MPC_JOINT_RX, b1 = (ax + 1) % 3, b2 = (ax + 2) % 3;
      for (int e = 0; e < 9; ++e) Rj.m[e] = (e % 4 == 0) ? 1.0 : 0.0;
      Rj.m[3 * b1 + b1] = cs; Rj.m[3 * b1 + b2] = -sn; Rj.m[3 * b2 + b1] = sn; Rj.m[3 * b2 + b2] = cs;
    }
    const M3 Rp = ldm3(jd + MPC_MODEL_JOINT_DOUBLES * i);
    const V3 pp = ldv3(jd + MPC_MODEL_JOINT_DOUBLES * i + 9);
    // (Rp, pp) * (Rj, pj) * (R, p)
    p = mul(Rp, mul(Rj, p) + pj) + pp;
    R = mul(Rp, mul(Rj, R));
  }
}

// pose = 12 doubles: R row-major, then p
DEV void walk_pose_store(double* o, const M3& R, V3 p) { for (int e = 0; e < 9; ++e) o[e] = R.m[e]; o[9] = p.x; o[10] = p.y; o[11] = p.z; }
DEV void walk_pose_copy(double* o, const double* s) { for (int e = 0; e < 12; ++e) o[e] = s[e]; }

// the foothold beside `pose`: translation moved by `offset` in the pose's yaw frame, rotation turned by rot_diff when asked (talos_utils.py:221-225)
// floor_z: no foothold is planned below this height (mpc_walk_config.floor_z ; <= -1e300: no floor)
DEV void walk_beside(double* o, const double* pose, const double* offset, const double* rot_diff, bool rotate, double floor_z) {
  const double yaw = atan2(pose[3], pose[0]), c = cos(yaw), s = sin(yaw);
  o[9] = pose[9] + c * offset[0] - s * offset[1];
  o[10] = pose[10] + s * offset[0] + c * offset[1];
  o[11] = pose[11] + offset[2];
  if (floor_z > -1e300 && o[11] < floor_z) o[11] = floor_z;
  if (rotate) { const M3 R2 = mul(ldm3(rot_diff), ldm3(pose)); for (int e = 0; e < 9; ++e) o[e] = R2.m[e]; }
  else for (int e = 0; e < 9; ++e) o[e] = pose[e];
}

// footTrajectory.updateTrajectory's rules on the state st = [start_L | final_L | start_R | final_R] (12 doubles each) from the measured sole poses
DEV void walk_plan(double* st, const double* LF, const double* RF, int takeoff_RF, int takeoff_LF, int land_RF, int land_LF, int T_ds,
                   const double* t_left, const double* t_right, const double* rot_diff, double floor_z) {
  double *sL = st, *fL = st + 12, *sR = st + 24, *fR = st + 36;
  if (land_LF < 0) { walk_pose_copy(sL, LF); walk_pose_copy(fL, LF); }
  if (land_RF < 0) { walk_pose_copy(sR, RF); walk_pose_copy(fR, RF); }
  if (takeoff_RF >= 0 && takeoff_RF < T_ds) {  // the right foot next to the left one, then the left foot next to that foothold
    walk_pose_copy(sR, RF); walk_beside(fR, LF, t_right, rot_diff, true, floor_z);
    walk_pose_copy(sL, LF); walk_beside(fL, fR, t_left, rot_diff, false, floor_z);
  }
  if (takeoff_LF >= 0 && takeoff_LF < T_ds) {
    walk_pose_copy(sL, LF); walk_beside(fL, RF, t_left, rot_diff, false, floor_z);
    walk_pose_copy(sR, RF); walk_beside(fR, fL, t_right, rot_diff, true, floor_z);
  }
}

// reference of the knot j ticks ahead for a foot with `land` ticks to its landing (talos_utils.py:297-318): the start pose before the swing, the final
// pose after it, in between the degree-8 Bezier curve (4 x start, the lifted point 3/4 start + 1/4 end, 4 x end) and the geodesic between the rotations
DEV void walk_ref(double* o, const double* start, const double* fin, int land, int j, int T_ss, double apex) {
  if (land <= -1) { walk_pose_copy(o, start); return; }
  const int ts = land - j;
  if (ts <= 0) { walk_pose_copy(o, fin); return; }
  if (ts > T_ss) { walk_pose_copy(o, start); return; }
  const double s = (double)(T_ss - ts) / (double)T_ss, r = 1.0 - s;
  const double binom[9] = {1, 8, 28, 56, 70, 56, 28, 8, 1};
  double sp[9], rp[9];
  sp[0] = 1.0; rp[0] = 1.0;
  for (int i = 1; i < 9; ++i) { sp[i] = sp[i - 1] * s; rp[i] = rp[i - 1] * r; }
  double b0 = 0.0, b4, b1 = 0.0;  // weights of the start point (control points 0..3), the lifted point (4) and the end point (5..8)
  for (int i = 0; i < 4; ++i) b0 += binom[i] * sp[i] * rp[8 - i];
  b4 = binom[4] * sp[4] * rp[4];
  for (int i = 5; i < 9; ++i) b1 += binom[i] * sp[i] * rp[8 - i];
  for (int c = 0; c < 3; ++c) {
    const double mid = 0.75 * start[9 + c] + 0.25 * fin[9 + c] + (c == 2 ? apex : 0.0);
    o[9 + c] = b0 * start[9 + c] + b4 * mid + b1 * fin[9 + c];
  }
  const M3 R0 = ldm3(start), R1 = ldm3(fin);
  const V3 w = log3(tmul(R0, R1));
  if (w.x == 0.0 && w.y == 0.0 && w.z == 0.0) { for (int e = 0; e < 9; ++e) o[e] = start[e]; }
  else { const M3 Rk = mul(R0, exp3(s * w)); for (int e = 0; e < 9; ++e) o[e] = Rk.m[e]; }
}
