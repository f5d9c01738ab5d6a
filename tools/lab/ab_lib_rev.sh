#!/bin/bash
# as ab_lib.sh, in the order prev / new / prev (the first run of a call sees a colder box)
set -u
if [ ! -f tools/lab/libfemshell_prev.so ]; then
  echo "ab_lib_rev.sh: tools/lab/libfemshell_prev.so is missing -- copy the build to compare against there first (it is not tracked)" >&2
  exit 2
fi
cp fem-shell_amd/libfemshell.so /tmp/libfemshell_new.so && cp tools/lab/libfemshell_prev.so fem-shell_amd/libfemshell.so
echo "== prev"; python3 "$@"
cp /tmp/libfemshell_new.so fem-shell_amd/libfemshell.so
echo "== new"; python3 "$@"
cp tools/lab/libfemshell_prev.so fem-shell_amd/libfemshell.so
echo "== prev again"; python3 "$@"
cp /tmp/libfemshell_new.so fem-shell_amd/libfemshell.so
