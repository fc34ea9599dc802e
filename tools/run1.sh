#!/bin/bash
cd /root/repo
export E2_WGRAD_STAMPS=1
for f in 7,2,1,256,8 4,4,1,256,8; do
E2_MFMA_DTYPE=bf16 E2_WGRAD_FORCE=$f timeout -k 10 120 python tools/one_layer.py wgradp 200 200 1 3 3 10 39 39 3 2>&1 | tail -2 | cut -c1-330
done
E2_WGRAD_FORCE=7,2,1,128,8 timeout -k 10 120 python tools/one_layer.py wgradp 200 200 1 3 3 10 39 39 3 2>&1 | tail -2 | cut -c1-330
E2_WGRAD_FORCE=3,2,14,256,30 timeout -k 10 120 python tools/one_layer.py wgradp 20 40 3 3 3 23 90 90 3 2>&1 | tail -2 | cut -c1-330
