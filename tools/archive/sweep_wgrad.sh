#!/bin/bash
# usage: sweep_wgrad.sh "<layer args>" cfg1 cfg2 ...
L="$1"; shift
for f in "$@"; do E2_WGRAD_FORCE=$f python tools/one_layer.py wgrad $L 10 2>&1 | grep -E "TF|rror:" | sed 's/.*wgrad/wgrad/' ; done
