#!/bin/bash
# builds tools/lab/asm_lab and prints the registers of the assembly kernels
cd "$(dirname "$0")" || exit 1
hipcc -O3 --offload-arch=gfx950 -std=c++17 -Rpass-analysis=kernel-resource-usage -I../../include -I../../fem-shell_amd/csrc asm_lab.hip ../../fem-shell_amd/csrc/plan.cpp -o asm_lab 2>&1 |
  grep -E "error|Function Name: _ZN8femshell(15k_assemble_pipeILi0|10k_assembleILi2ELi0ELb0)" -A6 | grep -E "error|Name|VGPRs:|Scratch"
