#!/bin/bash
# tuning build of the pipelined kernel's translation unit: scripts/piped_variant.sh <name> [extra flags]  ->  variants/<name>.so
# (links against the fsm_capi.o of the last library build: run `python -m ttcr_amd.build` first)
name=${1:-dev}; shift
root="$(cd "$(dirname "$0")/.." && pwd)"; mkdir -p "$root/variants"
cd "$root/ttcr_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-result \
  "$@" -c fsm_piped.hip -o "$root/variants/$name.o" 2>&1 | grep -E "error|warning" -A3 | head -40
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC _obj/fsm_capi.o "$root/variants/$name.o" -o "$root/variants/$name.so" && echo "variants/$name.so"
