# usage: ab_flags.sh "<cflags A>" "<cflags B>" [bench args] -- build both flag sets into variants/, alternate them on the same box
cd $GRAFT_REPO_ROOT
V=disentangled-vae_amd/build/variants; mkdir -p $V
FA="$1"; FB="$2"; shift; shift
DVAE_CFLAGS="$FA" python disentangled-vae_amd/build.py --force > /dev/null 2>&1; cp disentangled-vae_amd/libdvae_hip.so $V/a.so
DVAE_CFLAGS="$FB" python disentangled-vae_amd/build.py --force > /dev/null 2>&1; cp disentangled-vae_amd/libdvae_hip.so $V/b.so
echo "A = '$FA'   B = '$FB'"
tools/ab.sh $V/a.so $V/b.so "$@"
