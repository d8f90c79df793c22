#!/bin/bash
# tuning build (float grids only): scripts/devbuild.sh [name] [extra flags]  ->  variants/<name>.so
name=${1:-dev}; shift
root="$(cd "$(dirname "$0")/.." && pwd)"; mkdir -p "$root/variants"
cd "$root/ttcr_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math \
  -Wall -Wno-unused-result -DFSM_DEV_F32_ONLY "$@" -c fsm_capi.hip -o "$root/variants/$name.o" 2>&1 | grep -E "error|warning" -A3 | head -40
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "$root/variants/$name.o" _obj/fsm_fast.o -o "$root/variants/$name.so" && echo "variants/$name.so"
