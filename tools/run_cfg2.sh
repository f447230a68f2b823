cd $GRAFT_REPO_ROOT
for cfg in "A=1" "DPMN_SKMLP_TRAIN=0" "DPMN_DET_SMALL=0" "DPMN_SKMLP_TRAIN=0 DPMN_DET_SMALL=0"; do
  echo "== $cfg"
  env $cfg python -m pytest tests/test_gpu_bench_shapes.py -q -k "test_cfg2_training_step_tatt_3p3_vs_oracle_autograd and 4-False" > /dev/null 2>&1
  python - <<P
import json
d=json.load(open("gpurun_out/parity_errors.json"))
print([ (m["metric"][:8], "%.2e" % m["value"]) for m in d["records"] if m["test"].startswith("cfg2_step_tatt3p3_B4")])
P
done
