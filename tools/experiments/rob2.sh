cd /root/repo
for cfg in "NOSETUP=1 SIGMA=1 DOFS=all" "NOSETUP=1 SIGMA=1 DOFS=all WALK=1" "ITERS=2 SIGMA=1 DOFS=all"; do
  echo "=== $cfg"; env $cfg timeout 600 python tools/robustness_probe.py 1000 32 2>&1 | tail -24
done
