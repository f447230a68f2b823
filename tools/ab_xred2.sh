#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in "XRED=0" "XRED=1" "XRED=1 ABL=9" "XRED=1 ABL=17" "XRED=1 ABL=33" "XRED=1 ABL=1" "XRED=1 ABL=4" "XRED=1 ABL=3"; do
  x=0; abl=0; unset DPMN_XRED_ORDER DPMN_XRED_GROUP DPMN_XRED_CSTRIDE
  for kv in $cfg; do case $kv in XRED=*) x=${kv#XRED=};; ABL=*) abl=${kv#ABL=};; ORDER=*) export DPMN_XRED_ORDER=${kv#ORDER=};; GROUP=*) export DPMN_XRED_GROUP=${kv#GROUP=};; CSTRIDE=*) export DPMN_XRED_CSTRIDE=${kv#CSTRIDE=};; esac; done
  echo "== $cfg"
  DPMN_CONV_XRED=$x DPMN_XRED_ABLATE=$abl timeout 200 python tools/prof_layer.py 2>&1 | grep -v amdgpu.ids | awk '{printf "%s %s;", $1, $2}'; echo
done
