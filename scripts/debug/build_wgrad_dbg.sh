#!/bin/bash
# debug build of the library with in-kernel stamps in the conv weight-gradient kernels (scripts/debug_wgrad_stamps.py)
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
B=$R/mi-seg_amd/csrc/build
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I $R/include '-DMISEG_COMPILED_ARCH="gfx950"' -DMISEG_WGRAD_STAMPS -c $R/mi-seg_amd/csrc/conv3d.hip -o /tmp/conv3d_dbg.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/scripts/micro/libmiseg_hip_dbg.so $(ls $B/*.o | grep -v common_prof | grep -v conv3d.hip.o) /tmp/conv3d_dbg.o
echo built $R/scripts/micro/libmiseg_hip_dbg.so
