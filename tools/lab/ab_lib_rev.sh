#!/bin/bash
# as ab_lib.sh, in the order prev / new / prev (the first run of a call sees a colder box)
set -u
cp fem-shell_amd/libfemshell.so /tmp/libfemshell_new.so && cp tools/lab/libfemshell_prev.so fem-shell_amd/libfemshell.so
echo "== prev"; python3 "$@"
cp /tmp/libfemshell_new.so fem-shell_amd/libfemshell.so
echo "== new"; python3 "$@"
cp tools/lab/libfemshell_prev.so fem-shell_amd/libfemshell.so
echo "== prev again"; python3 "$@"
cp /tmp/libfemshell_new.so fem-shell_amd/libfemshell.so
