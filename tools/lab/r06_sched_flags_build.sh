#!/bin/bash
# Round 6: does the compiler's scheduling strategy move the assembly kernel (latency-bound at two waves per SIMD)?  Builds
# fem-shell_amd/libfemshell_<tag>.so with kernels.hip compiled under each flag set (the other objects as they are); r06_sched_flags_run.sh
# times them alternately on the GPU box.  Run here (no GPU needed), then gpurun the other script.
cd fem-shell_amd/csrc
make -s || exit 1
build() { # tag, flags...
  tag=$1; shift
  mkdir -p build_$tag
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Wall -Wno-unused-result "$@" --offload-arch=gfx950 -I../../include -I/opt/rocm/include -c kernels.hip -o build_$tag/kernels.o || return 1
  objs=$(ls build/*.o | grep -v "build/kernels.o")
  /opt/rocm/bin/hipcc -shared --offload-arch=gfx950 -o ../libfemshell_$tag.so build_$tag/kernels.o $objs -L/opt/rocm/lib -lroctx64 -ldl -lpthread
}
build maxilp -mllvm -amdgpu-sched-strategy=max-ilp &
build maxclause -mllvm -amdgpu-sched-strategy=max-memory-clause &
build bias0 -mllvm -amdgpu-schedule-metric-bias=0 &
build relaxed -mllvm -amdgpu-schedule-relaxed-occupancy &
wait
ls -la ../libfemshell_*.so
