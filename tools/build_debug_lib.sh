#!/bin/bash
# usage: tools/build_debug_lib.sh [noguard]   (noguard: the time MLP is evaluated once, without the evaluate-twice guard)
# tools/bin/libtortoise_mi355x_dbg.so: the engine with -DTTS_DEBUG_CHECKSUM in diffusion.hip (a hash of every launch's output on stderr), for tools/determinism_trace.py
cd "$(dirname "$0")/../tortoise.cpp_amd" || exit 1
make >/dev/null || exit 1
mkdir -p ../tools/bin
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -Icsrc -I../include -mllvm -amdgpu-mfma-vgpr-form -DTTS_DEBUG_CHECKSUM $([ "$1" = noguard ] && echo -DTTS_DEBUG_NO_TIME_GUARD) -c csrc/diffusion.hip -o /tmp/diffusion_dbg.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../tools/bin/libtortoise_mi355x_dbg.so csrc/ar.o /tmp/diffusion_dbg.o csrc/extras.o csrc/vocoder.o csrc/api.o csrc/host_logic.o
