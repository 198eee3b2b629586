"""Frame-sharded data parallelism for the photometric fit (new in this build: the reference is
single-process, single-GPU -- SURVEY.md section 2 rows 16-17, section 8(e)).

One process per GPU.  Every rank holds a replica of all parameters and works on a contiguous slice of the frame batch (monocular:
frames; NeRSemble: views).  Per Adam step of the captured step (tracker.GraphedStep):
  1. one scalar all-reduce for the batch-global photometric normaliser #(alpha > 0) (tracker.py:439 divides by the count over the
     WHOLE batch), between the forward and the backward plan;
  2. the texture gradient (50.3 MB at T = 2048): reduce-scatter (ReduceOp.AVG) of its folded level 0 -> each rank finishes and
     Adam-updates ITS rows with its slice of the optimiser state -> all-gather of the updated rows; all three on a communication
     stream of the step's own, beside the geometry backward (all-reduce + replicated update as the fallback: VHAP_TEX_SHARDED=0);
  3. one small all-reduce (average) of the arena slice holding every other trained gradient (~90 KB).
The eager `optimize_iter` averages one flat bucket of all gradients instead (`average_gradients`).
The per-rank energy is defined so that its mean over ranks equals the single-GPU energy (equal shards): batch means become local
means, the photometric numerator is multiplied by world_size through the normaliser.  Replicas stay bit-identical because every rank
applies the same averaged gradient with the same Adam state.  The colour disturbance draws its pools from the local shard only.
Backend: 'nccl' (= RCCL over xGMI on ROCm) on GPUs, 'gloo' in the CPU tests.
"""
import os

import numpy as np
import torch
import torch.distributed as dist


class FrameShardContext:
    def __init__(self, group=None):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world_size = dist.get_world_size(group)
        # a ONE-rank group can still be made to take the sharded step (four plans with the collectives in between) -- VHAP_FORCE_DIST=1:
        # the RCCL calls (ReduceOp.AVG reduce-scatter, all-gather, asynchronous handles) then run for real on the one GPU a developer has
        self.force = os.environ.get("VHAP_FORCE_DIST", "0") == "1"
        self._flat = None

    @property
    def sharded(self):
        """True when the step runs in its frame-sharded form (collectives between the plans)."""
        return self.world_size > 1 or self.force

    # ---- sharding ----
    def shard_slice(self, n):
        """Contiguous split of n frames; the remainder goes to the first ranks."""
        base, rem = divmod(n, self.world_size)
        start = self.rank * base + min(self.rank, rem)
        return slice(start, start + base + (1 if self.rank < rem else 0))

    def shard_sample(self, sample):
        """This rank's contiguous slice of a batch.  A RAGGED batch (the last one of a sequence or of a shuffled pass: its size is not a
        multiple of the number of ranks) is not split at all: every rank fits the whole of it.  The per-rank energy is built so that its
        mean over ranks is the single-process energy for EQUAL shards (local means, the photometric sum normalised by the pixel count of
        all ranks times the number of ranks) -- which replicas of the same batch satisfy trivially, unequal shards do not; the price is
        redundant work on that one batch, the averaged gradient (and hence every replica) stays exact."""
        n = len(sample["timestep_index"])
        if n % self.world_size:
            return dict(sample)
        sl = self.shard_slice(n)
        out = {}
        for k, v in sample.items():
            if isinstance(v, (torch.Tensor, np.ndarray)) and len(v) == n:
                out[k] = v[sl]
            else:
                out[k] = v
        return out

    def probe(self):
        """Try every collective the sharded texture update issues, once, on small tensors, BEFORE a step is captured: reduce-scatter
        with ReduceOp.AVG, all-gather into a tensor, an asynchronous averaged all-reduce.  A collective library that refuses one of them
        (an unsupported op is refused on every rank at the same call, before anything is on the wire; the verdict is agreed by an
        all-reduce all the same) must not end the job -- the step then takes the all-reduce + replicated
        finish form (VHAP_TEX_SHARDED=0).  -> True when the sharded texture update can be used."""
        ok = 1
        try:
            dev = "cuda" if dist.get_backend(self.group) == "nccl" else "cpu"
            n = self.world_size
            full = torch.arange(16 * n, dtype=torch.float32, device=dev)
            out = torch.empty(16, dtype=torch.float32, device=dev)
            self.reduce_scatter_mean(full, out, async_op=True).wait()
            planes = torch.zeros(2, 4 * n, 8, dtype=torch.float32, device=dev)
            planes[:, 4 * self.rank:4 * self.rank + 4] = float(self.rank + 1)
            for w in self.all_gather_rows(planes, 4 * self.rank, 4, async_op=True):
                w.wait()
            ones = torch.ones(4, dtype=torch.float32, device=dev)
            w = self.all_reduce_mean_(ones, async_op=True)
            if w is not None:
                w.wait()
            if dev == "cuda":
                torch.cuda.synchronize()
            want = torch.arange(16 * n, dtype=torch.float32).view(n, 16)[self.rank]
            if not (torch.allclose(out.cpu(), want) and all(float(planes[0, 4 * r, 0]) == r + 1 for r in range(n)) and
                    torch.allclose(ones.cpu(), torch.ones(4))):
                ok = 0
        except Exception as e:                                   # noqa: BLE001 -- whatever the library raises
            import sys
            print(f"[vhap_amd.dist] rank {self.rank}: collective probe failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
            ok = 0
        try:
            flag = torch.tensor([ok], dtype=torch.int32, device="cuda" if dist.get_backend(self.group) == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
            ok = int(flag.item())
        except Exception:                                        # noqa: BLE001
            ok = 0
        self.tex_sharded_ok = bool(ok)                            # (on the context, not in the environment: round-5 advisor)
        return bool(ok)

    def tex_sharded_usable(self):
        """probe()'s verdict, probing on first use: GraphedStep asks before it captures a sharded texture path, so that library users
        (optimize_stage under torchrun) get the documented fallback instead of a refusal inside the first replay.  Every rank reaches
        this at the same point of the same call sequence."""
        if getattr(self, "tex_sharded_ok", None) is None:
            self.probe()
        return self.tex_sharded_ok

    # ---- collectives ----
    def all_reduce_sum(self, t):
        t = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def all_reduce_sum_(self, t):
        """in place (no clone): t = sum over ranks of t"""
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def broadcast_int(self, value, src=0):
        """The same Python int on every rank (rank `src`'s)."""
        dev = "cuda" if dist.get_backend(self.group) == "nccl" else "cpu"
        t = torch.tensor([int(value)], dtype=torch.int64, device=dev)
        dist.broadcast(t, src=src, group=self.group)
        return int(t.item())

    def all_reduce_mean_(self, t, async_op=False):
        """In-place mean over ranks of a contiguous tensor (RCCL averages in the collective; gloo sums, then scales).  With
        async_op the collective is only enqueued (it still waits for the work already on the current stream) and the returned handle's
        wait() makes the current stream wait for it -- kernels launched in between overlap with it."""
        if dist.get_backend(self.group) == "nccl":
            work = dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group, async_op=async_op)
            return work if async_op else None
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        t.mul_(1.0 / self.world_size)
        return None

    class _Done:
        """handle of a collective that has already completed (gloo: synchronous)"""
        def wait(self):
            return True

    def reduce_scatter_mean(self, full, out, async_op=False):
        """out (this rank's 1/N slice, contiguous) = mean over ranks of the matching slice of `full` (contiguous, N equal slices).  RCCL:
        ONE reduce-scatter (ReduceOp.AVG) -- half the wire traffic of the all-reduce the slice would otherwise come from; gloo has no
        reduce-scatter: all-reduce, scale, copy the slice (the CPU / single-GPU tests: same numbers).  -> handle with wait()."""
        n = self.world_size
        assert full.is_contiguous() and out.is_contiguous() and full.numel() == n * out.numel()
        if dist.get_backend(self.group) == "nccl":
            work = dist.reduce_scatter_tensor(out, full, op=dist.ReduceOp.AVG, group=self.group, async_op=async_op)
            return work if async_op else self._Done()
        dist.all_reduce(full, op=dist.ReduceOp.SUM, group=self.group)
        out.copy_(full.view(n, -1)[self.rank].view_as(out)).mul_(1.0 / n)
        return self._Done()

    def all_gather_rows(self, planes, row0, nrows, async_op=False):
        """planes [C, T, W] (contiguous): every rank owns the rows [r * nrows, (r + 1) * nrows) of each plane and has just updated ITS rows
        [row0, row0 + nrows); afterwards every rank holds all rows.  One all-gather per plane straight into the plane (a plane's row
        strips are its N equal contiguous chunks).  -> list of handles with wait()."""
        assert planes.is_contiguous() and row0 == self.rank * nrows and planes.shape[1] == nrows * self.world_size
        works = []
        nccl = dist.get_backend(self.group) == "nccl"
        for c in range(planes.shape[0]):
            # RCCL gathers IN PLACE when the input is the rank's own slot of the output (sendbuff == recvbuff + rank * count): no staging
            # copy of the strip; gloo is handed a copy
            src = planes[c, row0:row0 + nrows]
            if not nccl:
                src = src.clone()
            w = dist.all_gather_into_tensor(planes[c], src, group=self.group, async_op=async_op)
            works.append(w if async_op and w is not None else self._Done())
        return works

    def average_gradients(self, params):
        """One flat-bucket all-reduce of every gradient (missing grads count as zero)."""
        params = [p for p in params if p.requires_grad]
        total = sum(p.numel() for p in params)
        if total == 0:
            return
        if self._flat is None or self._flat.numel() != total or self._flat.device != params[0].device:
            self._flat = torch.empty(total, dtype=torch.float32, device=params[0].device)
        flat, o = self._flat, 0
        for p in params:
            n = p.numel()
            if p.grad is None:
                flat[o:o + n].zero_()
            else:
                flat[o:o + n].copy_(p.grad.reshape(-1))
            o += n
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        flat.mul_(1.0 / self.world_size)
        o = 0
        for p in params:
            n = p.numel()
            if p.grad is None:
                p.grad = flat[o:o + n].view_as(p).clone()
            else:
                p.grad.copy_(flat[o:o + n].view_as(p))
            o += n


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun); returns (rank, world, local_rank).  A single process
    gets no group -- unless VHAP_FORCE_DIST=1, which creates a world-size-1 group (RCCL on a GPU) so that the sharded step and its
    collectives can be run on one device (tests/test_dist_gpu.py, bench.py)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    force = os.environ.get("VHAP_FORCE_DIST", "0") == "1"
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1 and "MASTER_PORT" not in os.environ:
            import socket
            s = socket.socket()
            s.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(s.getsockname()[1])
            s.close()
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def attach(tracker, group=None):
    """Turn a tracker into one rank of a frame-sharded job."""
    tracker.dist = FrameShardContext(group)
    return tracker.dist
