#!/bin/bash
# Build an A/B variant of libfcsa_hip.so that differs only in ONE translation unit (default fcsa_bwd.hip):
#   tools/build_variant.sh <tag> "<extra hipcc flags>" [fcsa_bwd|fcsa_fwd|...]
# -> flash_cosine_sim_attention_amd/libfcsa_hip_<tag>.so  (use with FCSA_LIB=...).  The other objects come from csrc/build.
set -e
TAG=$1; FLAGS=$2; TU=${3:-fcsa_bwd}
cd "$(dirname "$0")/../flash_cosine_sim_attention_amd/csrc"
mkdir -p build_var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form $FLAGS -c $TU.hip -o build_var/${TU}_$TAG.o
OBJS=""
for f in fcsa_fwd fcsa_fwd3 fcsa_bwd fcsa_norm fcsa_capi; do
  if [ "$f" = "$TU" ]; then OBJS="$OBJS build_var/${TU}_$TAG.o"; else OBJS="$OBJS build/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libfcsa_hip_$TAG.so $OBJS
echo built ../libfcsa_hip_$TAG.so
