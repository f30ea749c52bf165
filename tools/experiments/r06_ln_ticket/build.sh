# builds gpurun_ab/libln_ticket.so: the product with -DSS_EXP_LN_TICKET in kernels_decode.hip and engine.cpp (round-6 experiment)
set -e
cd /root/repo
python -c "from speaksense_amd import build; build.build()"
mkdir -p gpurun_ab
for src in kernels_decode.hip engine.cpp; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -x hip -DSS_EXP_LN_TICKET -Ispeaksense_amd/csrc -c speaksense_amd/csrc/$src -o gpurun_ab/exp_$src.o &
done
wait
objs=""
for o in speaksense_amd/build/*.o; do b=$(basename $o); [ "$b" = whisper_compat_post154.o ] && continue
  if [ "$b" = kernels_decode.hip.o ] || [ "$b" = engine.cpp.o ]; then objs="$objs gpurun_ab/exp_${b%.o}.o"; else objs="$objs $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_ab/libln_ticket.so $objs -lpthread
ls -la gpurun_ab/libln_ticket.so
