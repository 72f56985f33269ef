"""Build tests/golden/labels_fixture.npz from the reference's OWN data fixtures: whole-utterance wavs of
data/subset and the per-utterance label files the reference's pipeline wrote for them
(scripts/create_video_train_files.py:181-288 -> *_vad_labels.h5, *_ibm_labels.h5).  These are OUTPUTS of the
reference's librosa-based label makers (packages/processing/target.py:5-70) and pin the oracle restatement
(librosa itself is not importable here).  Build container only (needs /root/reference and the h5py-enabled
interpreter /opt/conda/bin/python3.9):

    python tests/golden/make_labels_golden.py
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np
from scipy.io import wavfile

SUB = "/root/reference/data/subset"
UTTS = [("dev", "08F", "sa2"), ("train", "01M", "sa1"), ("dev", "08F", "si519")]

DUMP = r"""
import h5py, numpy as np, sys
out = sys.argv[1]
d = {}
for split, spk, utt in %r:
    for lab in ('vad', 'ibm'):
        with h5py.File('%s/processed/ntcd_timit/Clean/%%s/%%s/%%s_%%s_labels.h5' %% (split, spk, utt, lab), 'r') as f:
            d['%%s_%%s_%%s' %% (spk, utt, lab)] = f['Y'][:]
np.savez(out + '/labels.npz', **d)
""" % (UTTS, SUB)


def main():
    tmp = tempfile.mkdtemp()
    subprocess.check_call(["/opt/conda/bin/python3.9", "-c", DUMP, tmp])
    lab = np.load(tmp + "/labels.npz")
    fix = {}
    for split, spk, utt in UTTS:
        fs, w = wavfile.read(f"{SUB}/raw/ntcd_timit/Clean/volunteers/{spk}/straightcam/{utt}.wav")
        assert fs == 16000 and w.dtype == np.int16
        key = f"{spk}_{utt}"
        vad, ibm = lab[key + "_vad"], lab[key + "_ibm"]
        assert set(np.unique(vad)) <= {0.0, 1.0} and set(np.unique(ibm)) <= {0.0, 1.0}
        fix[key + "_wav_i16"] = w
        fix[key + "_vad_shape"] = np.array(vad.shape)
        fix[key + "_vad_bits"] = np.packbits(vad.astype(np.uint8).ravel())
        fix[key + "_ibm_shape"] = np.array(ibm.shape)
        fix[key + "_ibm_bits"] = np.packbits(ibm.astype(np.uint8).ravel())
        print(key, "samples", len(w), "vad", vad.shape, float(vad.mean()), "ibm", ibm.shape, float(ibm.mean()))
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "labels_fixture.npz")
    np.savez_compressed(out, **fix)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    sys.exit(main())
