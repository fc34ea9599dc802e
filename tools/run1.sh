#!/bin/bash
cd /root/repo
export E2_MFMA_DTYPE=bf16 E2_WGRAD_STAMPS=1
cp elektronn2_amd/libe2hip.so /tmp/orig.so
for v in orig abl1 abl2; do
  if [ $v != orig ]; then cp elektronn2_amd/libe2hip_$v.so elektronn2_amd/libe2hip.so; fi
  echo "== $v"
  for f in 7,2,1,256,8 2,2,1,256,9; do
  E2_WGRAD_FORCE=$f timeout -k 10 120 python tools/one_layer.py wgradp 200 200 1 3 3 10 39 39 3 2>&1 | tail -2 | head -1 | cut -c1-260
  done
done
cp /tmp/orig.so elektronn2_amd/libe2hip.so
