# usage: bash tools/fa_run.sh [variant ...]   ("main" = dpmn_amd/lib/libdpmn_hip.so)
for v in "$@"; do echo "== $v"; if [ "$v" = main ]; then python tools/bench_fused_attn.py 48 2>&1 | tail -2; else DPMN_HIP_LIB=tools/variants/libdpmn_$v.so python tools/bench_fused_attn.py 48 2>&1 | tail -2; fi; done
