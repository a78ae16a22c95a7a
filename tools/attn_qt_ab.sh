#!/bin/bash
# ABAB on one box: the product kernel (one query tile per wave) against ZH_ATTN_QT=2
for i in 1 2 3; do
  ZH_ATTN_QT=1 python3 tools/attn_qt_ab.py 2>&1 | grep QT=
  ZH_ATTN_QT=2 python3 tools/attn_qt_ab.py 2>&1 | grep QT=
done
