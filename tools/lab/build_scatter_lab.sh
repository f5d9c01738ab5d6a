#!/bin/bash
# builds tools/lab/scatter_lab (the element-parallel scatter assembly with FP64 atomics, measured against the library's gather)
cd "$(dirname "$0")" || exit 1
hipcc -O3 --offload-arch=gfx950 -std=c++17 -munsafe-fp-atomics -I../../include -I../../fem-shell_amd/csrc scatter_lab.hip -o scatter_lab 2>&1 | grep -E "error|warning: v" | head
ls -la scatter_lab | cut -c1-80
