for w in 8 40; do echo "== refine 3, corrector 20, window $w"; WINDOW=$w REFINES=3 CORRECTOR=20 timeout 300 python tools/robustness_matrix.py 999 frozen,shared,instance; done
