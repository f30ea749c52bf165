# builds gpurun_ab/libprio.so = the product library with an extra -D in the translation unit that holds the cross-attention kernels:
#   bash tools/experiments/r04_chain_prio/build_prio.sh -DSS_CROSS_PRIO=1
set -e
cd /root/repo
mkdir -p gpurun_ab
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -x hip $@"
/opt/rocm/bin/hipcc $F -c speaksense_amd/csrc/kernels_decode.hip -o gpurun_ab/kernels_decode.hip.o
objs=""
for o in speaksense_amd/build/*.o; do b=$(basename $o); if [ "$b" = "kernels_decode.hip.o" ]; then objs="$objs gpurun_ab/$b"; else objs="$objs $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_ab/libprio.so $objs -lpthread
ls -la gpurun_ab/libprio.so
