cd /root/repo
for cfg in "SIGMA=1 DOFS=all" "SIGMA=1 DOFS=upper" "SIGMA=0 DOFS=all" "SIGMA=0.25 DOFS=all"; do
  echo "=== $cfg"; env $cfg timeout 300 python tools/robustness_probe.py 400 32 2>&1 | tail -22
done
