cd /root/repo
for cfg in "ITERS=2 WALK=1 PERINST=1" "ITERS=3 WALK=1 PERINST=1" "ITERS=2 WALK=1 PERINST=1 ISOLATE=1"; do
  echo "=== $cfg"; env $cfg SIGMA=1 DOFS=all timeout 900 python tools/robustness_probe.py 1000 64 2>&1 | grep -v "^  tick" | tail -24
done
