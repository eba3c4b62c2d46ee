#!/bin/bash
# Development variant of libfcsa_hip.so: fcsa_fwd.hip and fcsa_bwd.hip rebuilt with -DFCSA_DEV_ONLY (bf16, one head dim: seconds
# instead of minutes) plus the given flags, linked with the norm / C-ABI objects built the same way.   tools/build_dev.sh <tag> "<flags>"
set -e
TAG=$1; FLAGS=$2
cd "$(dirname "$0")/../flash_cosine_sim_attention_amd/csrc"
mkdir -p build_var
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form -DFCSA_DEV_ONLY $FLAGS"
$CC -c fcsa_fwd.hip -o build_var/fcsa_fwd_$TAG.o &
$CC -c fcsa_fwd3.hip -o build_var/fcsa_fwd3_$TAG.o &
$CC -c fcsa_bwd.hip -o build_var/fcsa_bwd_$TAG.o &
$CC -c fcsa_norm.hip -o build_var/fcsa_norm_$TAG.o &
$CC -c fcsa_capi.hip -o build_var/fcsa_capi_$TAG.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libfcsa_hip_$TAG.so build_var/fcsa_fwd_$TAG.o build_var/fcsa_fwd3_$TAG.o build_var/fcsa_bwd_$TAG.o build_var/fcsa_norm_$TAG.o build_var/fcsa_capi_$TAG.o
echo built libfcsa_hip_$TAG.so
