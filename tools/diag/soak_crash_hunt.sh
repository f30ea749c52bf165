# The soak test, repeated, with the library's crash hook on (SS_CRASH_BACKTRACE=1: native frames of the faulting thread on SIGSEGV / SIGBUS / SIGABRT):
# stops at the first run that dies of a signal and prints its frames (dev tool, GPU box).
#   gpurun --timeout 1200 -- 'bash tools/diag/soak_crash_hunt.sh'
# To look for container overruns in the host side, point SS_LIB_PATH at a build of engine.cpp / capi.cpp / model.cpp / whisper_compat.cpp with
# -D_GLIBCXX_ASSERTIONS (bounds-checked std::vector / std::string) linked with the product's kernels_*.hip.o:
#   for s in engine.cpp capi.cpp model.cpp whisper_compat.cpp; do hipcc --offload-arch=gfx950 -O2 -g -std=c++17 -fPIC -D_GLIBCXX_ASSERTIONS -x hip -c speaksense_amd/csrc/$s -o /tmp/a/$s.o; done
#   hipcc --offload-arch=gfx950 -shared -fPIC -o speaksense_amd/libspeaksense_hip_assert.so /tmp/a/*.o speaksense_amd/build/*.hip.o -lpthread
# (ROCm's AddressSanitizer runtime intercepts HSA allocations and did not come up on this image; rocgdb changed the timing enough to hide the fault.)
cd $GRAFT_REPO_ROOT
export SS_SOAK_SECONDS=${SS_SOAK_SECONDS:-90} SS_CRASH_BACKTRACE=1
for i in 1 2 3 4 5; do
  python -m pytest tests/test_gpu_lifetime.py -q -m gpu -k soak -s -p no:faulthandler > gpurun_out/soak_hunt_$i.log 2>&1; rc=$?
  echo "run $i rc=$rc: $(tail -1 gpurun_out/soak_hunt_$i.log | cut -c1-120)"
  if [ $rc -ge 128 ] || grep -a -q "fatal signal" gpurun_out/soak_hunt_$i.log; then grep -a -n "fatal signal" -B4 -A30 gpurun_out/soak_hunt_$i.log | head -60; break; fi
done
