# the soak test, repeated, against a build of the host side with -D_GLIBCXX_ASSERTIONS (bounds-checked std::vector / std::string) and the library's crash
# hook on: stops at the first run that dies of a signal and prints its native frames (dev tool, GPU box).  Build (dev container):
#   for s in engine.cpp capi.cpp model.cpp whisper_compat.cpp: hipcc --offload-arch=gfx950 -O2 -g -std=c++17 -fPIC -D_GLIBCXX_ASSERTIONS -x hip -c ... -o /tmp/assert_obj/$s.o
#   hipcc --offload-arch=gfx950 -shared -fPIC -o speaksense_amd/libspeaksense_hip_assert.so /tmp/assert_obj/*.o speaksense_amd/build/*.hip.o -lpthread
cd $GRAFT_REPO_ROOT
export SS_SOAK_SECONDS=${SS_SOAK_SECONDS:-90} SS_CRASH_BACKTRACE=1
for i in 1 2 3 4 5; do
  python -m pytest tests/test_gpu_lifetime.py -q -m gpu -k soak -s -p no:faulthandler > gpurun_out/soak_hunt_$i.log 2>&1; rc=$?
  echo "run $i rc=$rc: $(tail -1 gpurun_out/soak_hunt_$i.log | cut -c1-120)"
  if [ $rc -ge 128 ] || grep -a -q "fatal signal" gpurun_out/soak_hunt_$i.log; then grep -a -n "fatal signal" -B4 -A30 gpurun_out/soak_hunt_$i.log | head -60; break; fi
done
