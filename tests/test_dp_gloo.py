"""CPU, world_size 2 over gloo: the data-parallel exchange (disentangled-vae_amd/dp.py) reproduces the
single-process step on the concatenated batch: shard rows, local gradients (numpy oracle), one SUM
all-reduce of the flat buffer, 1/world scaling, identical Adam on every rank."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    import golden_util as gu
    from oracle import vae_oracle as vo
    dp = importlib.import_module("disentangled-vae_amd.dp")
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dims = dict(x_dim=37, y_dim=5, z_dim=4, h_dim=(16, 16))
    Bg = 24
    params = gu.make_params("M2", dims, 1)
    p = {k: v.astype(np.float64) for k, v in params.items()}
    opt = vo.AdamState(list(p))
    names = list(p)
    for step in range(3):
        x, y, e = gu.make_batch(dims, Bg, 10 + step)
        lo, hi = dp.shard_rows(Bg, rank, world)
        out, grads = vo.vae_loss_and_grads("M2", p, x[lo:hi].astype(np.float64), y[lo:hi].astype(np.float64), e[lo:hi].astype(np.float64))
        flat = torch.from_numpy(np.concatenate([np.asarray(grads[n], np.float64).ravel() for n in names]))
        dp.allreduce_flat_(flat)
        flat /= world
        g, o = {}, 0
        for n in names:
            g[n] = flat[o:o + p[n].size].numpy().reshape(p[n].shape); o += p[n].size
        opt.step(p, g)
        loss = dp.mean_scalars_(torch.tensor([out["loss"]], dtype=torch.float64), world)
    q.put((rank, {k: v.copy() for k, v in p.items()}, float(loss)))
    dist.destroy_process_group()


def test_two_rank_gloo_step_equals_single_process():
    sys.path.insert(0, HERE)
    import golden_util as gu
    from oracle import vae_oracle as vo
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    dims = dict(x_dim=37, y_dim=5, z_dim=4, h_dim=(16, 16))
    p = {k: v.astype(np.float64) for k, v in gu.make_params("M2", dims, 1).items()}
    opt = vo.AdamState(list(p))
    for step in range(3):
        x, y, e = gu.make_batch(dims, 24, 10 + step)
        out, _ = vo.train_step_vae("M2", p, opt, x.astype(np.float64), y.astype(np.float64), e.astype(np.float64))
    for k in p:
        np.testing.assert_allclose(res[0][1][k], p[k], rtol=1e-9, atol=1e-12)     # W-rank result == single process
        np.testing.assert_array_equal(res[0][1][k], res[1][1][k])                 # replicas stay identical
    np.testing.assert_allclose(res[0][2], out["loss"], rtol=1e-12)


def _info_worker(rank, world, port, q):
    """M2_info (scripts/training_M2_info_vad.py:159-198): two parameter groups, ONE exchange.  The flat buffer holds what
    enc_loss.backward() leaves in every .grad plus what aux_loss.backward() adds to the auxiliary net (quirk Q4: the auxiliary
    group ends up with (gamma - beta) dBCE); it is reduced once after both backward passes -- legal because the two optimizers
    touch disjoint parameters and the step between the two backward passes changes nothing the second one reads."""
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    import golden_util as gu
    from oracle import vae_oracle as vo
    dp = importlib.import_module("disentangled-vae_amd.dp")
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dims = dict(x_dim=37, y_dim=1, z_dim=4, h_dim=(16, 16))
    Bg = 24
    p = {k: v.astype(np.float64) for k, v in gu.make_params("M2_info", dims, 2).items()}
    names = list(p)
    edc = [n for n in names if n.startswith("enc_dec_clf.")]
    aux = [n for n in names if n.startswith("auxiliary.")]
    opt_edc, opt_aux = vo.AdamState(edc), vo.AdamState(aux)
    for step in range(3):
        x, y, e = gu.make_batch(dims, Bg, 20 + step)
        lo, hi = dp.shard_rows(Bg, rank, world)
        out, g1, g2 = vo.m2info_losses_and_grads(p, x[lo:hi].astype(np.float64), y[lo:hi].astype(np.float64), e[lo:hi].astype(np.float64),
                                                 0.5, 10.0, 1.0)
        tot = {n: np.asarray(g1[n], np.float64) + (np.asarray(g2[n], np.float64) if n in g2 else 0.0) for n in names}
        flat = torch.from_numpy(np.concatenate([tot[n].ravel() for n in names]))
        dp.allreduce_flat_(flat)                                      # both groups in one message
        flat /= world
        g, o = {}, 0
        for n in names:
            g[n] = flat[o:o + p[n].size].numpy().reshape(p[n].shape); o += p[n].size
        opt_edc.step(p, {n: g[n] for n in edc})
        opt_aux.step(p, {n: g[n] for n in aux})
        losses = dp.mean_scalars_(torch.tensor([out["enc_loss"], out["aux_loss"]], dtype=torch.float64), world)
    q.put((rank, {k: v.copy() for k, v in p.items()}, losses.numpy().copy()))
    dist.destroy_process_group()


def test_two_rank_gloo_m2info_two_groups_equal_single_process():
    sys.path.insert(0, HERE)
    import golden_util as gu
    from oracle import vae_oracle as vo
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_info_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    dims = dict(x_dim=37, y_dim=1, z_dim=4, h_dim=(16, 16))
    p = {k: v.astype(np.float64) for k, v in gu.make_params("M2_info", dims, 2).items()}
    edc = [n for n in p if n.startswith("enc_dec_clf.")]
    aux = [n for n in p if n.startswith("auxiliary.")]
    opt_edc, opt_aux = vo.AdamState(edc), vo.AdamState(aux)
    for step in range(3):
        x, y, e = gu.make_batch(dims, 24, 20 + step)
        out, _, _, _ = vo.train_step_m2info(p, opt_edc, opt_aux, x.astype(np.float64), y.astype(np.float64), e.astype(np.float64),
                                            alpha=0.5, beta=10.0, gamma=1.0)
    for k in p:
        np.testing.assert_allclose(res[0][1][k], p[k], rtol=1e-9, atol=1e-12, err_msg=k)     # 2 ranks == single process on the whole batch
        np.testing.assert_array_equal(res[0][1][k], res[1][1][k])                              # replicas stay identical
    np.testing.assert_allclose(res[0][2], [out["enc_loss"], out["aux_loss"]], rtol=1e-12)


def test_shard_rows():
    dp = importlib.import_module("disentangled-vae_amd.dp")
    assert [dp.shard_rows(64, r, 8) for r in (0, 7)] == [(0, 8), (56, 64)]
    with pytest.raises(ValueError):
        dp.shard_rows(10, 0, 4)
