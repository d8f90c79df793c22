#!/bin/bash
# tuning build (float grids only): scripts/devbuild.sh [name] [extra flags]  ->  variants/<name>.so
name=${1:-dev}; shift
cd "$(dirname "$0")/../ttcr_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math \
  -Wall -Wno-unused-result -DFSM_DEV_F32_ONLY "$@" fsm_capi.hip -o ../../variants/$name.so 2>&1 | grep -E "error|warning" -A3 | head -40
