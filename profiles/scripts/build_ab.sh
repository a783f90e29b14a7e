#!/bin/bash
# A/B build of the library from the tree's sources with extra compiler flags: profiles/bin/ab/<name>.so  (loaded through OMM_AMD_LIBRARY by tests/ommtest.py)
# usage: bash profiles/scripts/build_ab.sh <name> "<extra flags>"      e.g.  build_ab.sh fold_noinline "-DOMMX_ONLY_U8 -DOMMX_FOLD_ATTR=__attribute__((noinline))"
name=$1; shift
R=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $R/profiles/bin/ab
make -j8 -C $R/omm_amd/csrc OUT=$R/profiles/bin/ab/$name.so BUILD=/tmp/ab_build_$name EXTRA="$*" 2>&1 | grep -E "error|Error" ; ls -la $R/profiles/bin/ab/$name.so
