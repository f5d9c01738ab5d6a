// FEM-shell-precice: coupled program, twin of src/fem-shell/preCICE/fem-shell_precice.cpp main().
#include <iostream>

#include "coupling.hpp"

int main(int argc, char **argv) { return femshell_host::fem_shell_precice_main(argc, argv, std::cout, std::cerr); }
