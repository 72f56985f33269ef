"""Build tests/golden/stft_ref_fixture.npz from the reference's OWN data fixtures.

Run in the build container only (needs /root/reference and the h5py-enabled
interpreter /opt/conda/bin/python3.9 to read HDF5):

    python tests/golden/make_stft_golden.py

What it captures (data only, no reference source):
  * the first frames of X_train / X_validation / Y_train / Y_validation of
    data/subset/processed/ntcd_timit/Clean_ibm_labels_upsampled.h5 -- these
    are OUTPUTS of the reference's librosa-based pipeline
    (scripts/create_train_set.py:132-194) and pin the forward STFT;
  * the int16 samples of the matching wav heads (enough for those frames) and
    the whole-file peak used by the reference's normalisation
    (scripts/create_train_set.py:137);
  * (wav length -> frame count) pairs confirmed by the reference's per-utterance
    label files (*_vad_labels.h5), which pin the end-pad rule.
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np
from scipy.io import wavfile

REF = "/root/reference"
SUB = REF + "/data/subset"
H5 = SUB + "/processed/ntcd_timit/Clean_ibm_labels_upsampled.h5"
NFRAMES = 8          # frames kept per utterance
NFFT, HOP = 1024, 256

DUMP = r"""
import h5py, numpy as np, sys, json, glob, os
out = sys.argv[1]
f = h5py.File(%r, 'r')
d = {k: f[k][:] for k in f}
np.savez(out + '/main.npz', **d)
counts = {}
for p in sorted(glob.glob(%r + '/processed/ntcd_timit/Clean/*/*/*_vad_labels.h5')):
    with h5py.File(p, 'r') as g:
        counts[os.path.relpath(p, %r)] = [int(s) for s in g['Y'].shape]
json.dump(counts, open(out + '/counts.json', 'w'))
""" % (H5, SUB, SUB)


def main():
    tmp = tempfile.mkdtemp()
    subprocess.check_call(["/opt/conda/bin/python3.9", "-c", DUMP, tmp])
    main_h5 = np.load(tmp + "/main.npz")
    counts = json.load(open(tmp + "/counts.json"))

    fix = {}
    # X_train = 01M sa1|sa2|si462 (67 frames each), X_validation = 08F sa1|sa2|si519
    layout = {"train": ("01M", ["sa1", "sa2", "si462"]), "validation": ("08F", ["sa1", "sa2", "si519"])}
    nsamp = NFFT + HOP * (NFRAMES - 1)
    for split, (spk, utts) in layout.items():
        X = main_h5["X_" + split]
        Y = main_h5["Y_" + split]
        per = X.shape[1] // len(utts)
        for i, u in enumerate(utts):
            fs, w = wavfile.read(f"{SUB}/raw/ntcd_timit/Clean/volunteers/{spk}/straightcam/{u}.wav")
            assert fs == 16000 and w.dtype == np.int16
            key = f"{spk}_{u}"
            fix[key + "_wav_head_i16"] = w[:nsamp].copy()
            fix[key + "_peak_i16"] = np.int64(np.max(np.abs(w.astype(np.int64))))
            fix[key + "_n"] = np.int64(len(w))
            fix[key + "_X"] = X[:, i * per:i * per + NFRAMES].astype(np.float32)
            fix[key + "_Y_ibm"] = Y[:, i * per:i * per + NFRAMES].astype(np.float32)
    fix["X_train_mean_head"] = main_h5["X_train_mean"][:16, 0]
    fix["X_train_std_head"] = main_h5["X_train_std"][:16, 0]

    # (wav length -> frames) known answers from the reference's label files
    kat = []
    for rel, shape in counts.items():
        parts = rel.split("/")            # processed/ntcd_timit/Clean/<split>/<spk>/<utt>_vad_labels.h5
        spk, utt = parts[-2], parts[-1].replace("_vad_labels.h5", "")
        wav = f"{SUB}/raw/ntcd_timit/Clean/volunteers/{spk}/straightcam/{utt}.wav"
        if not os.path.exists(wav):
            continue
        fs, w = wavfile.read(wav)
        kat.append((len(w), shape[-1]))
    fix["kat_len_frames"] = np.array(sorted(set(kat)), dtype=np.int64)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stft_ref_fixture.npz")
    np.savez_compressed(out, **fix)
    print("wrote", out, os.path.getsize(out), "bytes;", len(fix), "arrays; KATs:", fix["kat_len_frames"].tolist())


if __name__ == "__main__":
    sys.exit(main())
