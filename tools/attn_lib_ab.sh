#!/bin/bash
# ABAB on one box: the product library against a variant (default: -DZH_ATTN_ACC_INIT=0, the round-5 softmax with its per-score fma)
VAR=${1:-$PWD/tools/_abl/libzh_attn_noacci.so}   # (the round-6 ACC_INIT experiment: profiles/r06_attn_acc_init.patch applies it)
for i in 1 2 3; do
  ZUTIS_HIP_LIB=$VAR python3 tools/attn_lib_ab.py 2>&1 | grep " us "
  python3 tools/attn_lib_ab.py 2>&1 | grep " us "
done
