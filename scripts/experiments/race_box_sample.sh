# sample a box: the bit-identity test N times per dtype in fresh processes; if any run fails, stay in this box and dig (more runs, the
# failure reports, variants).  Output: gpurun_out/flake4_<serial>.log
N=${N:-10}
mkdir -p gpurun_out
SER=$(rocm-smi --showserial 2>/dev/null | grep -i "serial number:" | head -1 | awk '{print $NF}')
LOG=gpurun_out/flake4_$SER.log
echo "box $SER $(date +%T)" | tee $LOG
T="tests/test_graph_gpu.py::test_forked_graph_step_is_bit_identical"
run() {  # $1 = dtype id, rest = env assignments
  local dt=$1; shift
  env DRN_TEST_ISOLATED=1 "$@" timeout 300 python -m pytest -x -q -p no:cacheprovider "$T[$dt]" > /tmp/one.log 2>&1
  local rc=$?
  if [ $rc -ne 0 ]; then echo "FAIL $dt $*" >> $LOG; grep "^E  " /tmp/one.log | head -6 >> $LOG; fi
  return $rc
}
fails=0
for i in $(seq 1 $N); do for dt in dtype0 dtype1; do run $dt || fails=$((fails+1)); done; done
echo "first pass: $fails of $((2*N)) failed" | tee -a $LOG
if [ $fails -gt 0 ]; then
  for v in "X=1" "AMD_SERIALIZE_KERNEL=3" "DRN_FORK_MAIN_FIRST=1" "HIP_LAUNCH_BLOCKING=1"; do
    f=0; for i in $(seq 1 $((2*N))); do for dt in dtype0 dtype1; do run $dt $v || f=$((f+1)); done; done
    echo "variant $v: $f of $((4*N)) failed" | tee -a $LOG
  done
fi
