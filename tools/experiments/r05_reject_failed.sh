for seed in 1 4 7 20250304 2 3; do
  for rej in 0 1; do
    MPC_HIP_REJECT_FAILED=$rej SEED=$seed GENERATOR=device WINDOW=8 REFINES=3 python tools/robustness_matrix.py 999 instance 2>&1 | grep references | sed "s/^/reject $rej seed $seed: /" | cut -c1-250
  done
done
