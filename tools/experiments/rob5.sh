cd /root/repo
for cfg in "DYN_SCALE=1" "DYN_SCALE=1e-2" "ARMIJO=1e-2" "REG_INIT=1e-6" "MU_INIT=1e-6" "DYN_SCALE=1e-5"; do
  echo "=== $cfg"; env $cfg SIGMA=1 DOFS=all timeout 600 python tools/robustness_probe.py 400 32 2>&1 | grep -v "^  tick" | tail -9
done
