#!/bin/bash
# Profiling-build variants of the library with other ring depths of the split-context attention role (csrc/layer_attn.h, Q4_ATT_RING):
# only the three layer_attn*.hip units are recompiled, everything else is the profiling build's objects.
#   tools/lab/build_ring_variants.sh 2 6 8   ->  llama_cu_awq_amd/libllama2_q4_ring<D>.so   (A/B: Q4_LIB_OVERRIDE=... tools/lab/sweep_knob.py 7b 14 0,1)
set -e
cd "$(dirname "$0")/../../llama_cu_awq_amd/csrc"
make -s -j8 ../libllama2_q4_prof.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include -Wall -Wno-unused-function -Wno-unused-value -mllvm -amdgpu-kernarg-preload-count=16 -DQ4_PROFILING"
for D in "$@"; do
    mkdir -p build_ring$D
    for f in layer_attn layer_attn_h64 layer_attn_h256; do
        /opt/rocm/bin/hipcc $FLAGS -DQ4_ATT_RING=$D -c $f.hip -o build_ring$D/$f.o &
    done
    wait
    OBJS=$(ls build_prof/*.o | grep -v layer_attn)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=exports.map -o ../libllama2_q4_ring$D.so $OBJS build_ring$D/*.o build/q4_host.o
    echo built libllama2_q4_ring$D.so
done
