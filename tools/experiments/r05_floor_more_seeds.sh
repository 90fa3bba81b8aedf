for seed in 9 10 11 12 13 14 15 16; do
  FLOOR=1 SEED=$seed GENERATOR=device WINDOW=8 REFINES=3 python tools/robustness_matrix.py 999 instance 2>&1 | grep references | sed "s/^/floor, seed $seed: /" | cut -c1-330
done
