#!/bin/bash
# usage: sweep_igemm.sh <op> cin cout kd kh kw D H W -- cfg1 cfg2 ...
args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
for c in "$@"; do
  E2_IGEMM_FORCE=$c timeout -k 5 60 python tools/one_layer.py "${args[@]}" 20 2>&1 | tail -1
done
