# per-launch timeline of ONE single-image SISR x4 forward (LR 64^2 -> 256^2): through gpurun
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/prof_sisr_n1; mkdir -p $R/gpurun_out/prof_sisr_n1
cat > /tmp/sisr_n1.py <<PY
import sys, torch
sys.path.insert(0, "$R")
from bench import build_net
from virnet_amd.utils.synth import synth_images
dev = torch.device("cuda", 0)
net, sd = build_net(dev, "sisr"); net.load_state_dict(sd, strict=True); net = net.to(dev).eval()
x = synth_images(1, 3, 64, 64).to(dev)
with torch.no_grad():
    for _ in range(20):
        net(x, 4)
    torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_sisr_n1 -o t --output-format csv -- python /tmp/sisr_n1.py > $R/gpurun_out/prof_sisr_n1/log.txt 2>&1
f=$(find $R/gpurun_out/prof_sisr_n1 -name "*kernel_trace.csv" | head -1)
python - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# one forward = from a conv_head_s4 ... find the last two occurrences of the first kernel of a forward
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "FillFunctor<int>" in n or "fill" in n.lower() and "int" in n]
start = idx[-2] if len(idx) >= 2 else len(rows) - 110
end = idx[-1] if len(idx) >= 2 else len(rows)
prev = None; tot = 0
for r in rows[start:end]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0 if prev is None else s - prev
    prev = e; tot += e - s
    name = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:48]
    print(f"{name:50s} grid {int(r['Grid_Size_X'])//max(1,int(r['Workgroup_Size_X'])):6d} wg {int(r['Workgroup_Size_X']):4d}  {(e-s)/1000:7.1f} us  gap {gap/1000:6.1f}")
print("launches", end - start, "sum of kernels %.1f us" % (tot / 1000))
PY
