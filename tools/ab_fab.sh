#!/bin/bash
# build-variant sweep of the fused attention backward (tools/build_variants.sh attn_fused_bwd "<flags>" fab_<tag>)
cd $GRAFT_REPO_ROOT
python tools/prof_attn_bwd.py 1 2>&1 | grep ^shift
for v in tools/variants/libdpmn_fab_*.so; do echo -n "$(basename $v) "; DPMN_HIP_LIB=$v python tools/prof_attn_bwd.py 1 2>&1 | grep ^shift; done
