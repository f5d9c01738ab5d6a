// FEM-shell: stand-alone program, twin of the reference's src/fem-shell/fem-shell.cpp main().
#include <iostream>

#include "shell_system.hpp"

int main(int argc, char **argv) { return femshell_host::fem_shell_main(argc, argv, std::cout, std::cerr); }
