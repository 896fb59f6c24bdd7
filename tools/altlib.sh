#!/bin/bash
# usage: tools/altlib.sh NAME "-DFLAG=1 ..." file1.hip [file2.hip ...]
# Builds build_alt/libw2s_NAME.so = the in-tree objects with the named sources recompiled under extra flags (A/B kernel experiments:
# run with W2S_LIB=build_alt/libw2s_NAME.so).  The compile command of each source is the Makefile's own (make -n: per-file flags included).
set -e
cd "$(dirname "$0")/../wav2sleep_amd/csrc"
NAME=$1; FLAGS=$2; shift 2
OUT=../../build_alt/$NAME; mkdir -p $OUT
OBJS=""
for f in *.hip; do
  o=${f%.hip}.o
  if [[ " $* " == *" $f "* ]]; then
    cmd=$(make -n -W $f $o | grep -- "-c $f" | head -1)
    cmd=${cmd/-o $o/-o $OUT/$o}
    eval "$cmd $FLAGS" &
    OBJS="$OBJS $OUT/$o"
  else
    OBJS="$OBJS $o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_alt/libw2s_$NAME.so $OBJS
echo built build_alt/libw2s_$NAME.so
