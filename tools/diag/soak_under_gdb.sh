# the soak test under rocgdb: prints the native backtrace of the thread that faults (dev tool, GPU box)
cd $GRAFT_REPO_ROOT
export SS_SOAK_SECONDS=${SS_SOAK_SECONDS:-90}
for i in 1 2 3; do
  /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex "handle SIGUSR1 nostop noprint" -ex run -ex "bt 25" -ex "info threads" --args python -m pytest tests/test_gpu_lifetime.py -q -m gpu -k soak > gpurun_out/soak_gdb_$i.log 2>&1
  if grep -q "SIGSEGV\|SIGABRT\|SIGBUS" gpurun_out/soak_gdb_$i.log; then echo "run $i: fault"; grep -n "SIGSEGV\|SIGABRT\|SIGBUS" -A30 gpurun_out/soak_gdb_$i.log | head -60; break; else echo "run $i: $(tail -1 gpurun_out/soak_gdb_$i.log)"; fi
done
