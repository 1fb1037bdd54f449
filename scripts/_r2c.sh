mkdir -p gpurun_out/r2c
for cfg in "DRN_SIDE_LANE=0" "DRN_SIDE_LANE=1"; do
  echo "== $cfg"
  env $cfg python scripts/graph_probe.py 2>/dev/null
done
