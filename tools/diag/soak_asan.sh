# the soak test against an AddressSanitizer build of the library's host side (dev tool, GPU box).  Build (in the dev container):
#   for s in engine.cpp capi.cpp model.cpp whisper_compat.cpp: hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fsanitize=address -fno-omit-frame-pointer -x hip -c ...
#   hipcc -shared -fPIC -fsanitize=address -shared-libasan -o speaksense_amd/libspeaksense_hip_asan.so <those>.o speaksense_amd/build/*.hip.o
cd $GRAFT_REPO_ROOT
export SS_SOAK_SECONDS=${SS_SOAK_SECONDS:-120} SS_LIB_PATH=$PWD/speaksense_amd/libspeaksense_hip_asan.so
export LD_PRELOAD=/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.asan-x86_64.so
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0:halt_on_error=1:symbolize=1:log_path=$PWD/gpurun_out/asan
export ASAN_SYMBOLIZER_PATH=/opt/rocm/lib/llvm/bin/llvm-symbolizer
python -m pytest tests/test_gpu_lifetime.py -q -m gpu -k soak -s -p no:faulthandler > gpurun_out/soak_asan.log 2>&1; echo "rc=$?"
tail -5 gpurun_out/soak_asan.log | cut -c1-200
ls gpurun_out/asan* 2>/dev/null | head; for f in gpurun_out/asan.*; do [ -f "$f" ] && head -60 "$f" | cut -c1-220; done
