#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for shape in prop_fc wgrad_nt; do
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  rm -rf /tmp/pm; rocprofv3 --kernel-trace --pmc $c -d /tmp/pm -o p -- python $R/scripts/pmc_gemm.py $shape > /dev/null 2>&1
  echo "== $shape $c"; python $R/scripts/pmc_show.py /tmp/pm/p_results.db 2>&1 | tail -4
done
done
