#!/bin/bash
# dev: builds ablation variants of libgandtr_hip.so (conv3x3_halo_c.hip with -DGDT_C_ABL=n) into tmpbin/
set -e
cd "$(dirname "$0")/../gandtr_amd/csrc"
mkdir -p ../../tmpbin
OBJS=$(ls *.o | grep -v conv3x3_halo_c.o)
for n in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -DGDT_C_ABL=$n -c conv3x3_halo_c.hip -o /tmp/halo_c_abl$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tmpbin/libgandtr_abl$n.so $OBJS /tmp/halo_c_abl$n.o
done
