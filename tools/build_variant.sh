#!/bin/bash
# usage: build_variant.sh NAME FILE.hip "-DFOO=1 ..."   (experiment build: -DDYF_EXPERIMENT_BUILD admits the wrong-results timing switches, csrc/common.h)
#   -> tools/variants/libvar_NAME.so (git-ignored; travels with gpurun; use with DYF_LIB=...) (bf16) with FILE.hip compiled with the extra flags
set -e
mkdir -p /root/repo/tools/variants
cd /root/repo/dyffusion_amd/csrc
base=$(basename $2 .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment -Wno-unsequenced -DDYF_EXPERIMENT_BUILD $3 -c $2 -o /root/repo/tools/variants/var_$1.o
objs=$(ls ../lib/obj/*.o | grep -v "/$base.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/tools/variants/libvar_$1.so $objs /root/repo/tools/variants/var_$1.o
ls -la /root/repo/tools/variants/libvar_$1.so
