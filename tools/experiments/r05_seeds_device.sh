#!/bin/bash
# eight other random ensembles over the whole schedule with bench.py's settings (refinement 3, corrector 20 on the 8 ticks after a pattern change, per-instance references generated in the library)
for seed in 1 2 3 4 5 6 7 8; do
  SEED=$seed GENERATOR=device WINDOW=8 REFINES=3 python tools/robustness_matrix.py 999 instance 2>&1 | grep references | sed "s/^/seed $seed: /"
done
