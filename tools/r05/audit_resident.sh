#!/bin/bash
# Build container: per-kernel register counts, spilled VGPRs and v_accvgpr_write / _read counts of the weight-stationary MCEM chain kernels
# (csrc/mcem_resident.hip: 32-frame tiles, csrc/mcem_resident16.hip: 16-frame tiles, csrc/mcem_resident4.hip: exact fp32 on 4-frame tiles), to compare after a compiler or flag change (the hazard
# table in mcem_resident.hip says what the numbers guard)
cd "$(dirname "$0")/../.."
for src in mcem_resident mcem_resident16 mcem_resident4; do
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -S disentangled-vae_amd/csrc/$src.hip -o /tmp/$src.s --cuda-device-only 2>/dev/null
python3 - $src <<'PY'
import re, sys
txt = open("/tmp/%s.s" % sys.argv[1]).read()
for m in re.finditer(r"^(_ZN4dvae5fused\d+mcem_resident\w*_kernel\w+):", txt, re.M):
    n = m.group(1)
    body = txt[m.end():]
    end = body.index("s_endpgm")
    tail = body[end:end + 6000]
    body = body[:end]
    g = lambda k: re.search(r"; %s: (\d+)" % k, tail).group(1)
    mp = re.search(r"Pol\w+?E", n)
    pol = mp.group(0)[:-1] if mp else "PolF32"
    yp = re.search(r"Li(\d+)E", n).group(1)
    print("%-16s %-8s label rows %-4s VGPRs %s AGPRs %s scratch %s B  v_accvgpr_write %d  v_accvgpr_read %d  mfma %d" % (
        sys.argv[1], pol, yp, g("NumVgprs"), g("NumAgprs"), g("ScratchSize"), body.count("v_accvgpr_write"), body.count("v_accvgpr_read"), body.count("v_mfma")))
PY
done
