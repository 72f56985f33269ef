"""Frame-minibatch data parallelism: frames are i.i.d. and the loss is a batch mean,
so rank k takes rows [k*B, (k+1)*B) of the global batch, every rank holds a full replica of the
0.17-0.30 M parameters, and the only exchange per step is ONE all-reduce (sum) of the flat fp32
gradient buffer, scaled by 1/world before an identical Adam step on every rank.  On the GPU node the
process group is RCCL over xGMI (backend "nccl"); the same code runs on gloo for CPU tests."""
import torch.distributed as dist


def shard_rows(global_rows, rank, world):
    """Rows [lo, hi) of the global minibatch owned by `rank` (equal shards; global_rows % world == 0)."""
    if global_rows % world:
        raise ValueError("global batch must divide evenly over the ranks")
    per = global_rows // world
    return rank * per, (rank + 1) * per


def allreduce_flat_(flat, group=None):
    """In-place SUM all-reduce of the flat gradient buffer (one message per step, <= 1.21 MB)."""
    if not flat.is_contiguous():
        raise ValueError("flat gradient buffer must be contiguous")
    if flat.is_cuda and dist.get_backend(group) == "gloo":
        # test rigs only (several ranks sharing one GPU cannot use RCCL): stage through the host
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        flat.copy_(host)
        return flat
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


def mean_scalars_(t, world, group=None):
    """Average a small tensor of per-rank loss scalars over the ranks (reporting only)."""
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    t.div_(world)
    return t
