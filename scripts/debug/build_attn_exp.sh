#!/bin/bash
# experimental builds of the attention kernels (scripts/bench_attn.py with MISEG_HIP_LIB=...): build_attn_exp.sh <tag> <-D flags...>
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
B=$R/mi-seg_amd/csrc/build
TAG=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I $R/include '-DMISEG_COMPILED_ARCH="gfx950"' "$@" -c $R/mi-seg_amd/csrc/attention.hip -o /tmp/attention_$TAG.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/scripts/micro/libmiseg_attn_$TAG.so $(ls $B/*.o | grep -v common_prof | grep -v attention.hip.o) /tmp/attention_$TAG.o
echo built $R/scripts/micro/libmiseg_attn_$TAG.so
