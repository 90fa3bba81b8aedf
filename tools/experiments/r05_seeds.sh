# the headline settings (refine 3, corrector 20 on the 8 ticks after a pattern change) and the every-tick corrector with OTHER random ensembles than the benchmark's
for seed in 1 2 3 4; do for w in 8 0; do echo "== seed $seed, refine 3, corrector 20, window $w"; SEED=$seed WINDOW=$w REFINES=3 CORRECTOR=20 timeout 300 python tools/robustness_matrix.py 999 frozen,instance; done; done
