# the floor under the measured soles with the generator in the library: eight other ensembles, bench.py's settings
for seed in 1 2 3 4 5 6 7 8 20250304; do
  FLOOR=1 SEED=$seed GENERATOR=device WINDOW=8 REFINES=3 python tools/robustness_matrix.py 999 instance 2>&1 | grep references | sed "s/^/floor, seed $seed: /" | cut -c1-330
done
FLOOR=1 GENERATOR=device CLOSED=1 WINDOW=8 REFINES=3 python tools/robustness_matrix.py 999 instance 2>&1 | grep references | sed "s/^/floor, closed loop: /" | cut -c1-330
FLOOR=1 GENERATOR=device WINDOW=0 REFINES=0 python tools/robustness_matrix.py 999 instance 2>&1 | grep references | sed "s/^/floor, plain warm start: /" | cut -c1-330
