#!/bin/bash
# tile sweep for the transformer / SequenceCNN GEMMs (tuning aid)
for mt in 4 2; do for nt in 8 4; do
  echo "== MT=$mt NT=$nt"
  BF=1 W2S_NO_SHRINK=1 W2S_FORCE_MT=$mt W2S_FORCE_NT=$nt python tools/kbench.py qkv ff1 proj ff2 seq1 seq32 f128 d128 f64 --iters 20 2>&1 | grep -v amdgpu.ids
done; done
echo "== default"; BF=1 python tools/kbench.py qkv ff1 proj ff2 seq1 seq32 f128 d128 f64 --iters 20 2>&1 | grep -v amdgpu.ids
