# usage: resusage.sh <file.hip> [extra cflags]  -- per-kernel register / spill / LDS summary from hipcc's resource-usage remarks
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $f -o /dev/null -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c "
import sys,re
cur=None; rows={}
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur=m.group(1); rows[cur]={}
    for k in ('VGPRs:','AGPRs','SGPRs:','ScratchSize','Occupancy','SGPRs Spill','VGPRs Spill','LDS Size'):
        m=re.search(k+r'[^:]*:?\s*(\d+)',l)
        if m and cur: rows[cur][k]=m.group(1)
import subprocess
for k,v in rows.items():
    name=subprocess.run(['c++filt',k],capture_output=True,text=True).stdout.strip()
    print(name[:110].ljust(110), v)
"
