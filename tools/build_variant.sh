#!/bin/bash
# usage: build_variant.sh NAME FILE.hip "-DFOO=1 ..." [f16]   (experiment build: -DDYF_EXPERIMENT_BUILD admits the wrong-results timing switches, csrc/common.h)
#   -> tools/variants/libvar_NAME.so (git-ignored; travels with gpurun; use with DYF_LIB=... / DYF_LIB_F16=... for the f16 build) with FILE.hip
#      compiled with the extra flags and the other objects of the product build (lib/obj or lib/obj_f16)
set -e
mkdir -p /root/repo/tools/variants
cd /root/repo/dyffusion_amd/csrc
base=$(basename $2 .hip)
objdir=../lib/obj; extra=""
if [ "$4" = "f16" ]; then objdir=../lib/obj_f16; extra="-DDYF_F16=1"; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment -Wno-unsequenced -DDYF_EXPERIMENT_BUILD $extra $3 -c $2 -o /root/repo/tools/variants/var_$1.o
objs=$(ls $objdir/*.o | grep -v "/$base.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/tools/variants/libvar_$1.so $objs /root/repo/tools/variants/var_$1.o
ls -la /root/repo/tools/variants/libvar_$1.so
