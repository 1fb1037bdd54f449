cd /root/repo
for st in 3 4; do echo "== stages=$st"; DRN_TN3_STAGES=$st timeout 120 python scripts/bench_wgrad3.py 2>&1 | grep -v amdgpu.ids | cut -c1-150; done
