#!/bin/bash
# per-layer conv TFLOP/s under forced split-K / tile settings (experiment knobs of csrc/conv.hip): tools/sweep_convs.sh > out.txt
for cfg in "0 0 768" "0 0 512" "0 0 1024" "0 0 1536" "3 0 768" "4 0 768" "6 0 768" "8 0 768" "12 0 768" "16 0 768" "24 0 768" "32 0 768" "48 0 768" "1 64 768" "2 64 768" "4 64 768" "8 64 768" "16 64 768"; do
  set -- $cfg
  DPMN_CONV_S=$1 DPMN_CONV_TILE=$2 DPMN_CONV_TARGET=$3 python tools/prof_convs.py 2>/dev/null | grep -E "^in .*cout +(128|256|512) |^convT" | sed -n 10,28p | awk -v c="S$1_T$2_G$3" '{ for (i=1;i<=NF;i++) if ($i=="TF") tf=$(i-1); if ($1=="convT") key="convT_"$3"_"$5; else key=$2"_c"$4"_k"$6; print key, c, tf }'
done
