echo "== refine 3, corrector 20, no alpha trigger"; MPC_X_ALPHA=0 REFINES=3 CORRECTOR=20 timeout 300 python tools/robustness_matrix.py 999 frozen,instance
echo "== refine 3, corrector 20, alpha <= 0.25"; MPC_X_ALPHA=0.3 REFINES=3 CORRECTOR=20 timeout 300 python tools/robustness_matrix.py 999 frozen,instance
echo "== refine 0, corrector 20, budget 2"; MPC_X_BUDGET=2 REFINES=0 CORRECTOR=20 timeout 300 python tools/robustness_matrix.py 999 frozen,instance
echo "== refine 0, corrector 20, budget 3"; MPC_X_BUDGET=3 REFINES=0 CORRECTOR=20 timeout 300 python tools/robustness_matrix.py 999 frozen,instance
echo "== refine 0, corrector 20, alpha <= 0.25"; MPC_X_ALPHA=0.3 REFINES=0 CORRECTOR=20 timeout 300 python tools/robustness_matrix.py 999 frozen,instance
