"""The reference's training loop (scripts/training_M1.py / training_M2.py / training_M2_info_vad.py: epochs of train batches, a validation pass,
one checkpoint per epoch named `<M>_epoch_{:03d}_vloss_{:.2f}.pt`, the same two log files) on the MI355X-native path:
GPU-resident frame store (disentangled-vae_amd/frames.py) + fused train step (disentangled-vae_amd/trainer.py).

    python examples/train_fused.py --model M2 --labels ibm_labels --h5 data/complete/processed/ntcd_timit/Clean_ibm_labels_upsampled.h5
    python examples/train_fused.py --model M2 --synthetic 200000            # no dataset at hand

Checkpoints are plain state_dicts with the reference's keys: the reference's evaluate / reconstruct scripts load them.
Differences from the scripts, all deliberate: the batch size is a flag (the fused step is meant for thousands of
frames per step; `--batch 128` reproduces the scripts' setting), std_norm is not offered (off in every script).
"""
import argparse
import importlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
Trainer = importlib.import_module("disentangled-vae_amd.trainer").Trainer
DeviceFrames = importlib.import_module("disentangled-vae_amd.frames").DeviceFrames


def synthetic(n, y_dim, seed):
    rng = np.random.default_rng(seed)
    scale = np.exp(np.random.default_rng(0).standard_normal((513, 1)) - 1)     # per-bin level shared by all splits
    X = (rng.standard_normal((513, n)) ** 2 * scale).astype(np.float32)
    Y = (rng.random((y_dim, n)) > 0.5).astype(np.float32) if y_dim else None
    return X, Y


def run_epoch(data, trainers, train, shuffle):   # trainers: [main, optional fork for the last, shorter batch]
    """One pass; returns mean (ELBO, recon, KL) over batches like the scripts (sum of batch means / number of batches).
    The sums are kept on the device by the step itself: no host sync, no extra kernels per step."""
    tot = torch.zeros(8, dtype=torch.float64, device=data.device)
    nb = 0
    for rows in data.index_batches(trainers[0].B, shuffle=shuffle):     # the rows kernel gathers the frames itself
        tr = trainers[0]
        if rows.shape[0] != tr.B:                               # last, shorter batch: same parameters, its own plan
            if len(trainers) == 1 or trainers[1].B != rows.shape[0]:
                trainers[1:] = [tr.fork(rows.shape[0])]
            tr = trainers[1]
        tr.accumulate_losses(tot)
        tr.step(data.x, data.y, rows=rows) if train else tr.evaluate(data.x, data.y, rows=rows)
        nb += 1
    for tr in trainers:
        tr.accumulate_losses(None)
    return (tot[:3] / nb).cpu().tolist()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", choices=["M1", "M2", "M2_info"], default="M2")
    ap.add_argument("--labels", choices=["vad_labels", "ibm_labels"], default="ibm_labels")
    ap.add_argument("--h5", default=None)
    ap.add_argument("--synthetic", type=int, default=0, help="number of synthetic training frames instead of --h5")
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--precision", choices=["fp32", "bf16"], default="fp32")
    ap.add_argument("--out", default="models/fused_run")
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    if a.model == "M2_info":
        a.labels = "vad_labels"                     # scripts/training_M2_info_vad.py: VAD labels, alpha 0 / beta 10 / gamma 1
    y_dim = 0 if a.model == "M1" else (1 if a.labels == "vad_labels" else 513)
    if a.h5:
        train, valid = DeviceFrames.from_hdf5(a.h5, "train"), DeviceFrames.from_hdf5(a.h5, "validation")
    else:
        n = a.synthetic or 100000
        train, valid = DeviceFrames(*synthetic(n, y_dim, 1)), DeviceFrames(*synthetic(max(n // 10, 1), y_dim, 2))
    if a.model == "M1":
        train.y = valid.y = None
    dims = dict(x_dim=513, y_dim=y_dim, z_dim=16, h_dim=(128, 128))
    tr = Trainer(a.model, dims, batch=min(a.batch, len(train)), precision=a.precision, lr=a.lr, seed=a.seed)
    tv = tr.fork(min(a.batch, len(valid)))
    train_trainers, valid_trainers = [tr], [tv]
    os.makedirs(a.out, exist_ok=True)
    open(os.path.join(a.out, "output_epoch.log"), "w").close()
    print(f"{a.model}: {len(train)} training frames, {len(valid)} validation frames, batch {tr.B}, {a.precision}")
    for epoch in range(a.epochs):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        elbo, rec, kl = run_epoch(train, train_trainers, True, True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        velbo, vrec, vkl = run_epoch(valid, valid_trainers, False, False)
        lines = [f"Epoch: {epoch}",
                 "[Train]\t\t ELBO: {:.2f}, Recon.: {:.2f}, KL: {:.2f}".format(elbo, rec, kl),
                 "[Validation]\t ELBO: {:.2f}, Recon.: {:.2f}, KL: {:.2f}".format(velbo, vrec, vkl)]
        with open(os.path.join(a.out, "output_epoch.log"), "a") as f:
            f.write("\n".join(lines) + "\n")
        print("\n".join(lines), f"   ({len(train) / dt / 1e6:.1f} M frames/s incl. shuffle)")
        torch.save({k: v.cpu() for k, v in tr.state_dict().items()},
                   os.path.join(a.out, "{}_epoch_{:03d}_vloss_{:.2f}.pt".format(a.model, epoch, velbo)))


if __name__ == "__main__":
    main()
