#!/bin/bash
cd $GRAFT_REPO_ROOT
for occ in 2 3; do for blocks in 256 512 768 1024 1536; do
  echo -n "OCC=$occ BLOCKS=$blocks: "; DPMN_SKMLP_OCC=$occ DPMN_SKMLP_BLOCKS=$blocks timeout 120 python tools/prof_skmlp.py 2>&1 | grep -v amdgpu.ids
done; done
