"""NTCD-TIMIT file listings the reconstruct / evaluate scripts ask for (reference packages/dataset/ntcd_timit.py:98-146,
386-474): pure path arithmetic over the corpus layout

    <raw>/ntcd_timit/matlab_raw/{train,dev,test}/<speaker>/<utt>.mat          lip-region recordings (one per utterance)
    <processed>/ntcd_timit/Clean/{train,dev,test}/<speaker>/<utt>_<labels>[_upsampled].h5

Written on pathlib; same return values (strings with forward slashes, sorted like the reference's recursive globs).
Only the functions the hot-path callers import are provided (scripts/reconstruct_*.py, scripts/evaluate_ntcd_*.py:
`speech_list`, `proc_noisy_clean_pair_dict`; plus `video_list`, which the dataset builders share).
"""
import os
from pathlib import Path

_SPLIT_DIR = {"train": "train", "validation": "dev", "test": "test"}
_NOISES = {"complete": (["Babble", "Cafe", "Car", "LR", "Street", "White"], ["-5", "0", "5", "10", "15"]),
           "subset": (["Babble", "LR"], ["-5"])}


def _split_root(base, kind, dataset_type):
    """<base>ntcd_timit/<kind>/[<split>/]: an unknown split name lists every split, as string concatenation did."""
    return str(base) + "ntcd_timit/" + kind + "/" + (_SPLIT_DIR[dataset_type] + "/" if dataset_type in _SPLIT_DIR else "")


def _find(root, suffix):
    """Every file below `root` whose name ends with `suffix`, sorted by full path (the order of sorted(glob('**/*suffix')))."""
    hits = []
    for d, _, files in os.walk(root):
        hits += [os.path.join(d, f) for f in files if f.endswith(suffix)]
    return sorted(hits)


def video_list(input_video_dir, dataset_type='train', labels='vad_labels', upsampled=False):
    """Lip-region .mat files of a split, relative to `input_video_dir`."""
    return [os.path.relpath(p, input_video_dir) for p in _find(_split_root(input_video_dir, "matlab_raw", dataset_type), ".mat")]


def speech_list(input_speech_dir, dataset_type='train'):
    """-> (input wav paths, output wav paths), both relative, one per recorded utterance of the split:
    'ntcd_timit/Clean/volunteers/<speaker>/straightcam/<utt>.wav' and 'ntcd_timit/Clean/<split dir>/<speaker>/<utt>.wav'."""
    mats = [Path(p) for p in _find(_split_root(input_speech_dir, "matlab_raw", dataset_type), ".mat")]
    inputs = ["ntcd_timit/Clean/volunteers/{}/straightcam/{}.wav".format(m.parent.name, m.stem) for m in mats]
    outputs = [os.path.join("ntcd_timit/Clean/" + str(Path(*m.parts[-3:]).with_suffix(".wav"))) for m in mats]
    return inputs, outputs


def proc_noisy_clean_pair_dict(input_speech_dir, dataset_type='train', dataset_size='complete', labels='vad_labels', upsampled=False):
    """{noisy wav path: clean label-file path} over every (noise type, SNR) of the data set size; keys
    'ntcd_timit/Noisy/<noise>/<snr>/<split dir>/<speaker>/<utt>.wav', values relative to `input_speech_dir`."""
    tail = "_" + labels + ("_upsampled" if upsampled else "")
    files = _find(_split_root(input_speech_dir, "Clean", dataset_type), labels + ("_upsampled" if upsampled else "") + ".h5")
    short = []
    for f in files:
        p = Path(*Path(f).parts[-3:]).with_suffix("")
        short.append(str(p).replace(tail, "") + ".wav")
    clean = [os.path.relpath(f, input_speech_dir) for f in files]
    noises, snrs = _NOISES["subset" if dataset_size == "subset" else "complete"]
    pairs = {}
    for noise in noises:
        for snr in snrs:
            for s, c in zip(short, clean):
                pairs[os.path.join("ntcd_timit", "Noisy", noise, snr, s)] = c
    return pairs
