#!/bin/bash
# A/B of two builds of the library on one GPU box (boxes differ by 3-5 %, so both builds run on the same one, twice,
# alternating): A = the tree's libzkhip.so, B = tools/ab/libzkhip_b.so (built by hand with other -D flags).
# usage: bash tools/ab/run_ab.sh '<command>' ['<command>' ...]   (each command under its own time limit)
set -e
run() { for c in "$@"; do timeout -k 10 200 bash -c "$c" 2>&1 | tail -2; done; }
cp zkevm-circuits_amd/libzkhip.so /tmp/libzkhip_a.so
trap 'cp /tmp/libzkhip_a.so zkevm-circuits_amd/libzkhip.so' EXIT
for round in 1 2; do
  echo "== A"; cp /tmp/libzkhip_a.so zkevm-circuits_amd/libzkhip.so; run "$@"
  echo "== B"; cp tools/ab/libzkhip_b.so zkevm-circuits_amd/libzkhip.so; run "$@"
done
