# kernel resource usage of one .hip unit: tools/kres.sh attention_fused.hip [extra hipcc flags]
f="$1"; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function \
  -Rpass-analysis=kernel-resource-usage "$@" -c "$(dirname "$0")/../openseg3d_amd/csrc/$f" -o /tmp/kres.o 2>&1 | python3 -c '
import sys,re
cur=None
for line in sys.stdin:
    if "error" in line or "warning" in line: print(line.rstrip()); continue
    m=re.search(r"Function Name: (\S+)",line)
    if m:
        import subprocess
        cur=subprocess.run(["c++filt",m.group(1)],capture_output=True,text=True).stdout.strip().split("(")[0][-60:]; vals={}
        continue
    for key in ("VGPRs:","AGPRs:","ScratchSize [bytes/lane]:","Occupancy [waves/SIMD]:","LDS Size [bytes/block]:","SGPRs:"):
        if key in line and "Spill" not in line and "Total" not in line:
            vals[key]=line.split(key)[1].split()[0]
    if "LDS Size" in line and cur:
        print(f"{cur:60s} vgpr {vals.get(\"VGPRs:\")} agpr {vals.get(\"AGPRs:\")} scratch {vals.get(\"ScratchSize [bytes/lane]:\")} occ {vals.get(\"Occupancy [waves/SIMD]:\")} lds {vals.get(\"LDS Size [bytes/block]:\")}")
'
