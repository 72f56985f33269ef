cd $GRAFT_REPO_ROOT
V=disentangled-vae_amd/build/variants; mkdir -p $V
DVAE_CFLAGS="-DR2_STASH_SC=1" python disentangled-vae_amd/build.py --force > /dev/null 2>&1; cp disentangled-vae_amd/libdvae_hip.so $V/sc1.so
DVAE_CFLAGS="-DR2_STASH_SC=0" python disentangled-vae_amd/build.py --force > /dev/null 2>&1; cp disentangled-vae_amd/libdvae_hip.so $V/sc0.so
DVAE_LIB=$PWD/$V/sc0.so python tests/diag/sc_check2.py /tmp/sc0.npz 2>&1 | grep -v amdgpu
DVAE_LIB=$PWD/$V/sc1.so python tests/diag/sc_check2.py /tmp/sc1.npz 2>&1 | grep -v amdgpu
python - <<'PY'
import numpy as np
a=np.load('/tmp/sc0.npz'); b=np.load('/tmp/sc1.npz')
for k in ('ws0','ws1'):
    x, y = a[k], b[k]
    d = np.nonzero(x != y)[0]
    print(k, 'bytes', x.size, 'differing bytes', d.size, 'first', d[:5], 'last', d[-5:] if d.size else None)
    if d.size:
        # histogram of differing offsets in MB
        h, edges = np.histogram(d, bins=20, range=(0, x.size))
        print('  per-20th histogram', h.tolist())
for k in ('slab0','slab1'):
    print(k, 'max abs diff', float(np.nanmax(np.abs(a[k]-b[k]))), 'nonfinite in sc1', int((~np.isfinite(b[k])).sum()))
print('sc0 ws0 vs ws1 identical:', bool((a['ws0']==a['ws1']).all()), ' sc1 ws0 vs ws1 identical:', bool((b['ws0']==b['ws1']).all()))
PY
