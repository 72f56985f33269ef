"""Frame-minibatch data parallelism: frames are i.i.d. and the loss is a batch mean,
so rank k takes rows [k*B, (k+1)*B) of the global batch, every rank holds a full replica of the
0.17-0.30 M parameters, and the only exchange per step is ONE all-reduce (sum) of the flat fp32
gradient buffer, scaled by 1/world before an identical Adam step on every rank.  On the GPU node the
process group is RCCL over xGMI (backend "nccl"); the same code runs on gloo for CPU tests."""
import ctypes
import os

import torch.distributed as dist

IPC_HANDLE_BYTES = 128         # include/dvae_train.h: DVAE_IPC_HANDLE_BYTES


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


def allreduce_flat_async_(flat, group=None):
    """SUM all-reduce of a part of the flat gradient that does NOT block the launch stream: returns a handle whose .wait() makes the current
    stream wait for the result (RCCL: the collective runs on the process group's own stream behind an event of the current one, so kernels
    launched on the current stream meanwhile overlap with it).  gloo test rigs (ranks sharing one GPU): staged through the host, done at once."""
    if not flat.is_contiguous():
        raise ValueError("flat gradient buffer must be contiguous")
    if flat.is_cuda and dist.get_backend(group) == "gloo":
        allreduce_flat_(flat, group)

        class _Done:
            def wait(self):
                return True
        return _Done()
    return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True)


def mean_scalars_(t, world, group=None):
    """Average a small tensor of per-rank loss scalars over the ranks (reporting only)."""
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    t.div_(world)
    return t


def exchange_mode():
    """DVAE_ALLREDUCE=rccl (default: torch.distributed all_reduce on the process group's backend) | direct (dvae_allreduce_flat: the
    library's own reduce-scatter / all-gather over hipIpc-mapped peer buffers, stream-ordered, fused with the slab sum)."""
    m = os.environ.get("DVAE_ALLREDUCE", "rccl")
    if m not in ("rccl", "direct"):
        raise ValueError("DVAE_ALLREDUCE must be rccl or direct")
    return m


class DirectExchange:
    """dvae_comm_* / dvae_allreduce_flat (include/dvae_train.h, csrc/allreduce.hip) for one flat gradient of n floats.

    The process group is used ONCE, on the host, to pass the hipIpc handles of the ranks' exchange buffers around; after that a step's
    exchange is one kernel launch per rank on the step's own stream.  Every rank must call allreduce() the same number of times.
    UNMEASURED on multi-GPU hardware (the build box has one GPU); bit-identical to the process-group path at world 2 (tests)."""

    def __init__(self, n_floats, group=None):
        from . import native as N
        self.N, self.lib = N, N.load()
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.n = int(n_floats)
        self.handle = ctypes.c_void_p()
        mine = (ctypes.c_ubyte * IPC_HANDLE_BYTES)()
        N.check(self.lib.dvae_comm_create(self.rank, self.world, self.n, ctypes.byref(self.handle), mine), "dvae_comm_create")
        gathered = [None] * self.world
        dist.all_gather_object(gathered, bytes(mine), group=group)
        blob = (ctypes.c_ubyte * (IPC_HANDLE_BYTES * self.world)).from_buffer_copy(b"".join(gathered))
        N.check(self.lib.dvae_comm_connect(self.handle, blob), "dvae_comm_connect")
        dist.barrier(group=group)                      # every rank has mapped every buffer before the first launch

    @classmethod
    def try_create(cls, n_floats, group=None):
        """-> (exchange, None) or (None, reason): like the constructor, but a rank whose create / connect step fails does not leave the
        others waiting in a collective -- every stage's outcome is agreed on by all ranks before the next one starts."""
        from . import native as N
        self = cls.__new__(cls)
        self.N, self.lib = N, N.load()
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.n = int(n_floats)
        self.handle = ctypes.c_void_p()
        mine = (ctypes.c_ubyte * IPC_HANDLE_BYTES)()
        err = None
        try:
            N.check(self.lib.dvae_comm_create(self.rank, self.world, self.n, ctypes.byref(self.handle), mine), "dvae_comm_create")
        except Exception as e:                          # noqa: BLE001 -- any local failure is reported, never raised past the collective
            err = f"rank {self.rank}: {e}"
        gathered = [None] * self.world
        dist.all_gather_object(gathered, (err, bytes(mine)), group=group)
        errs = [g[0] for g in gathered if g[0]]
        if not errs:
            try:
                blob = (ctypes.c_ubyte * (IPC_HANDLE_BYTES * self.world)).from_buffer_copy(b"".join(g[1] for g in gathered))
                N.check(self.lib.dvae_comm_connect(self.handle, blob), "dvae_comm_connect")
            except Exception as e:                      # noqa: BLE001
                err = f"rank {self.rank}: {e}"
            gathered = [None] * self.world
            dist.all_gather_object(gathered, err, group=group)
            errs = [g for g in gathered if g]
        if errs:
            self.close()
            return None, "; ".join(errs)
        return self, None

    def allreduce(self, slabs, n_slabs, slab_stride, out):
        """out[i] = sum over ranks of sum_k slabs[k * slab_stride + i]  (fp32 CUDA tensors; `out` may be slab 0); enqueued on the
        current stream, returns at once."""
        N = self.N
        N.check(self.lib.dvae_allreduce_flat(self.handle, N.ptr(slabs), int(n_slabs), int(slab_stride), N.ptr(out), N.stream()),
                "dvae_allreduce_flat")

    def set_timeout_ms(self, ms):
        """Wall-time bound of every in-kernel wait for a peer in the launches that follow (default 20 s, env DVAE_COMM_TIMEOUT_MS): generous
        enough for a peer that checkpoints or validates between two steps, finite so that a dead peer ends the launch."""
        self.N.check(self.lib.dvae_comm_set_timeout_ms(self.handle, int(ms)), "dvae_comm_set_timeout_ms")

    def failed(self):
        """Synchronises; True when a bounded wait for a peer expired on ANY rank (every rank's reduced gradient was then filled with NaN: the
        failure is also in-band -- parameters and losses turn NaN from that step on)."""
        f = ctypes.c_int(0)
        self.N.check(self.lib.dvae_comm_status(self.handle, ctypes.byref(f)), "dvae_comm_status")
        return bool(f.value)

    def close(self):
        if self.handle:
            self.lib.dvae_comm_destroy(self.handle)
            self.handle = ctypes.c_void_p()
