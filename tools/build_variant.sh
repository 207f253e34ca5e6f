#!/bin/bash
# Kernel experiments: build/variants/libsf_hip_<tag>.so = the current objects with ONE source recompiled with extra -D flags.
#   tools/build_variant.sh <tag> <source.hip> [-DNAME=VALUE ...]      then:  SF_HIP_LIB=$PWD/build/variants/libsf_hip_<tag>.so python ...
set -e
cd "$(dirname "$0")/.."
tag=$1; src=$2; shift 2
mkdir -p build/variants
python -m sample_factory_amd.build >/dev/null
fp=""; [ "$src" != "sf_nn.hip" ] && [ "$src" != "sf_dp.hip" ] && fp="-ffp-contract=off"
obj=build/variants/${src%.hip}_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Iinclude -Isample_factory_amd/csrc $fp "$@" \
    -c sample_factory_amd/csrc/$src -o $obj
objs=""
for s in sf_rl sf_nn sf_rnn sf_dp; do
    if [ "$s.hip" == "$src" ]; then objs="$objs $obj"; else objs="$objs sample_factory_amd/csrc/$s.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/libsf_hip_$tag.so $objs -ldl
echo build/variants/libsf_hip_$tag.so
