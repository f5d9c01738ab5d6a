#!/bin/bash
# A/B of two builds of libfemshell.so on one box:  tools/lab/ab_lib.sh <probe.py> [args]   (tools/lab/libfemshell_prev.so = the other build)
set -u
echo "== new"; python3 "$@"
cp fem-shell_amd/libfemshell.so /tmp/libfemshell_new.so && cp tools/lab/libfemshell_prev.so fem-shell_amd/libfemshell.so
echo "== prev"; python3 "$@"
cp /tmp/libfemshell_new.so fem-shell_amd/libfemshell.so
echo "== new again"; python3 "$@"
