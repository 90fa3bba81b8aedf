for seed in 1 4 7 20250304; do
  for g in "" 1; do
    MPC_WALK_GROUND=$g SEED=$seed GENERATOR=host WINDOW=8 REFINES=3 python tools/robustness_matrix.py 999 instance 2>&1 | grep references | sed "s/^/ground '$g' seed $seed: /" | cut -c1-330
  done
done
MPC_WALK_GROUND=1 GENERATOR=host WINDOW=0 REFINES=0 python tools/robustness_matrix.py 999 instance 2>&1 | grep references | sed "s/^/ground 1 plain warm start: /" | cut -c1-330
MPC_WALK_GROUND=1 GENERATOR=host CORRECTOR=0 REFINES=0 python tools/robustness_matrix.py 999 instance 2>&1 | grep references | sed "s/^/ground 1 exact budget: /" | cut -c1-330
