#!/bin/bash
# tuning build of ONE translation unit (seconds instead of minutes): scripts/tubuild.sh <fsm_slab|fsm_fast> [name] [extra flags]
#   -> variants/<name>.so, linked with the other objects of the last full build (ttcr_amd/csrc/_obj/*.o)
tu=$1; name=${2:-$tu}; shift; shift
root="$(cd "$(dirname "$0")/.." && pwd)"; mkdir -p "$root/variants"
cd "$root/ttcr_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math \
  -Wall -Wno-unused-result "$@" -c $tu.hip -o "$root/variants/$name.o" 2>&1 | grep -E "error|warning" -A3 | head -40
objs=""; for o in _obj/*.o; do [ "$o" != "_obj/$tu.o" ] && objs="$objs $o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "$root/variants/$name.o" $objs -o "$root/variants/$name.so" && echo "variants/$name.so"
