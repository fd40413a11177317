#!/bin/bash
# A/B of differently BUILT libraries (source experiments): runs tools/prefill_ab.py once per library per round,
# interleaved, each in its own process (SFA_LIB_PATH).  usage: tools/lib_ab.sh "impl args" lib1.so lib2.so ...
args=$1; shift
for rnd in 1 2 3; do
  for lib in "$@"; do
    echo "## round $rnd $lib"
    SFA_LIB_PATH=$PWD/starflashattention_amd/lib/$lib timeout -k 10 120 python -u tools/prefill_ab.py $args || exit 1
  done
done
