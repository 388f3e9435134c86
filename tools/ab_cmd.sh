#!/bin/bash
# tools/ab_cmd.sh "LIB_A LIB_B ..." SCRIPT [ARGS] -- alternate builds of librupphash_hip.so (RPH_LIB_PATH) under any python script of this repo on
# the SAME GPU box, three rounds; prints the script's lines that contain "kernel 4"
LIBS=$1; shift
for i in 1 2 3; do
  for L in $LIBS; do
    echo "$(basename $L): $(RPH_LIB_PATH=$L RPH_NO_CHECK=1 python "$@" 2>/dev/null | grep 'kernel 4')"
  done
done
