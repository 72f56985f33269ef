#!/bin/bash
# Build container: per-kernel VGPR spill counts and v_accvgpr_write / _read counts of the weight-stationary MCEM chain kernels (csrc/mcem_resident.hip),
# to compare after a compiler or flag change (the hazard table in that file says what the numbers guard)
cd "$(dirname "$0")/../.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -S disentangled-vae_amd/csrc/mcem_resident.hip -o /tmp/mcem_resident.s --cuda-device-only 2>/dev/null
python3 - <<'PY'
import re
txt = open("/tmp/mcem_resident.s").read()
names = re.findall(r"^(_ZN4dvae5fused20mcem_resident_kernel\w+):", txt, re.M)
for n in names:
    body = txt[txt.index(n + ":"):]
    body = body[:body.index("s_endpgm")]
    meta = txt[txt.index(".name:           " + n):]
    spill = re.search(r"\.vgpr_spill_count: (\d+)", meta).group(1)
    print(n[29:60], "spilled VGPRs", spill, "v_accvgpr_write", body.count("v_accvgpr_write"), "v_accvgpr_read", body.count("v_accvgpr_read"), "mfma", body.count("v_mfma"))
PY
