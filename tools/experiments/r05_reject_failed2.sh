for rej in 0 1; do
  MPC_HIP_REJECT_FAILED=$rej GENERATOR=device WINDOW=0 REFINES=0 python tools/robustness_matrix.py 999 frozen,instance 2>&1 | grep references | sed "s/^/reject $rej plain warm start + corrector: /" | cut -c1-260
  MPC_HIP_REJECT_FAILED=$rej GENERATOR=device CORRECTOR=0 REFINES=0 python tools/robustness_matrix.py 999 frozen,instance 2>&1 | grep references | sed "s/^/reject $rej exact budget: /" | cut -c1-260
  MPC_HIP_REJECT_FAILED=$rej CLOSED=1 GENERATOR=device WINDOW=8 REFINES=3 python tools/robustness_matrix.py 999 instance 2>&1 | grep references | sed "s/^/reject $rej closed loop: /" | cut -c1-260
done
