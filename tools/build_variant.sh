#!/bin/bash
# Build libmgf_hip.so with extra preprocessor flags into mgf_amd/variants/ for an A/B experiment (load it with MGF_AMD_LIB=...).
#   tools/build_variant.sh NAME -DFOO=1 ...
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p mgf_amd/variants
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -fvisibility=hidden -Wno-unused-function -Wno-unused-result"
/opt/rocm/bin/hipcc $F "$@" -c mgf_amd/csrc/mgf_hip.hip -o mgf_amd/variants/mgf_hip_$name.o
[ -f mgf_amd/csrc/prims.o ] || python -m mgf_amd.build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o mgf_amd/variants/libmgf_hip_$name.so mgf_amd/variants/mgf_hip_$name.o mgf_amd/csrc/prims.o
rm -f mgf_amd/variants/mgf_hip_$name.o
echo built mgf_amd/variants/libmgf_hip_$name.so
