cd /root/repo
for cfg in "EXTRA=4 SIGMA=1 DOFS=all" "ITERS=2 SIGMA=1 DOFS=all WALK=1"; do
  echo "=== $cfg"; env $cfg timeout 900 python tools/robustness_probe.py 1000 64 2>&1 | tail -24
done
