# build an A/B variant of one translation unit: tools/build_variant.sh <src in csrc> <extra flags...>  -> gpurun_ab/libvariant.so
set -e
cd /root/repo
src=$1; shift
mkdir -p gpurun_ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -x hip "$@" -c speaksense_amd/csrc/$src -o gpurun_ab/$src.o
objs=""
for o in speaksense_amd/build/*.o; do b=$(basename $o); [ "$b" = whisper_compat_post154.o ] && continue; if [ "$b" = "$src.o" ]; then objs="$objs gpurun_ab/$src.o"; else objs="$objs $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_ab/libvariant.so $objs -lpthread
ls -la gpurun_ab/libvariant.so
