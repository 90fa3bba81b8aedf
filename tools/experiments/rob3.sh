cd /root/repo
for cfg in "EXTRA=1 SIGMA=1 DOFS=all" "EXTRA=1 SIGMA=1 DOFS=all WALK=1"; do
  echo "=== $cfg"; env $cfg timeout 900 python tools/robustness_probe.py 1000 64 2>&1 | tail -24
done
