cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pwp
cat > /tmp/rk.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from dpmn_amd import ops
from dpmn_amd.utils import synth
B = 48
dev = torch.device("cuda:0")
g = synth.uniform("rf_g", (B, 1024, 384), -1, 1, 5).to(dev)
w = synth.uniform("rf_w", (384, 384), -0.1, 0.1, 5).to(dev)
b = synth.uniform("rf_b", (384,), -0.1, 0.1, 5).to(dev)
mode = sys.argv[1]
if mode == "fixedout":
    from dpmn_amd._abi import lib, dptr, check, stream
    z = torch.empty(B, 1024, 384, device=dev)
    for _ in range(300):
        check(lib.dpmn_pointwise_f32(dptr(g), dptr(w), dptr(b), dptr(z), B, 384, 1024, stream()))
elif mode == "zeros":
    g.zero_(); w.zero_()
    for _ in range(300):
        ops.pointwise(g, w, b)
else:
    for _ in range(300):
        ops.pointwise(g, w, b)
torch.cuda.synchronize()
PY
for mode in normal; do
rm -rf /tmp/pwp
rocprofv3 --kernel-trace --output-format csv -d /tmp/pwp -- python /tmp/rk.py $mode > /dev/null 2>&1
f=$(find /tmp/pwp -name "*kernel_trace.csv" | head -1)
python - "$f" $mode <<'PY'
import csv,sys
d=[(int(r['Start_Timestamp']), int(r['End_Timestamp'])-int(r['Start_Timestamp'])) for r in csv.DictReader(open(sys.argv[1])) if 'k_gemm_pw' in r['Kernel_Name']]
d.sort()
us=[x[1]/1e3 for x in d]
print(" ".join("%d:%.0f"%(i,us[i]) for i in range(0,len(us),8)))
t0=d[0][0]
print("elapsed ms at idx: ", " ".join("%d:%.1f"%(i,(d[i][0]-t0)/1e6) for i in range(0,len(us),40)))
PY
done
