"""Process-group bootstrap for one-process-per-GPU runs over RCCL/xGMI (replaces ref src/distributed.py).

Keeps the reference's helper names (`init_distributed_mode`, `is_dist_avail_and_initialized`, `get_rank`,
`is_main_process`, `save_on_master`, `mkdir`, `_get_first_node`) and its SLURM -> env mapping (ref distributed.py:41-60),
and adds what the reference leaves to Lightning: `setup_process_group()` (torchrun- or SLURM-launched, backend "nccl" = RCCL on
ROCm, gloo on CPU for tests) and a bucketed gradient all-reduce restricted to the parameters that actually received
gradients in this sub-step (the reference's DDP reduces every registered parameter, used or not: configs/trainer/ddp.yaml:12)."""
import errno
import os

import torch
import torch.distributed as dist


def _get_first_node():
    """First hostname in SLURM_JOB_NODELIST ('a[1-3,7]' -> 'a1', 'a,b' -> 'a', 'a' -> 'a')."""
    nodelist = os.getenv('SLURM_JOB_NODELIST')
    lb = nodelist.find("[")
    if lb >= 0 and "]" in nodelist[lb:]:
        inner = nodelist[lb + 1: nodelist.find("]", lb)]
        return nodelist[:lb] + inner.split(",")[0].split("-")[0]
    return nodelist.split(",")[0]


def init_distributed_mode(port=12354):
    """Export WORLD_SIZE / RANK / LOCAL_RANK / MASTER_ADDR / MASTER_PORT from SLURM variables (ref distributed.py:41-60).
    A torchrun launch already provides them; then this is a no-op."""
    if os.getenv('SLURM_NTASKS') is None:
        if "RANK" in os.environ and "WORLD_SIZE" in os.environ:
            return
        raise TypeError("init_distributed_mode needs SLURM_* variables (or a torchrun environment)")
    os.environ['WORLD_SIZE'] = os.getenv('SLURM_NTASKS')
    os.environ['RANK'] = os.getenv('SLURM_PROCID')
    os.environ['LOCAL_RANK'] = os.getenv('SLURM_LOCALID')
    master_addr = _get_first_node()
    if os.getenv('SYSTEMNAME', '') in ['juwels', 'juwelsbooster', 'jureca']:
        master_addr = master_addr + 'i'      # InfiniBand hostname suffix on the JSC machines
    os.environ['MASTER_ADDR'] = master_addr
    os.environ['MASTER_PORT'] = str(port)


def setup_process_group(backend=None):
    """One process per GPU.  Returns (rank, world_size, local_rank); initialises torch.distributed when world_size > 1."""
    # the host driver of this pool only supports dmabuf IPC (RCCL / cross-process tensor sharing fail with hipIpcGetMemHandle otherwise);
    # must be in the environment BEFORE the first call that initialises the HIP runtime, so it is the first thing done here (launchers
    # -- bench.py's parent process, torchrun -- export it as well)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if torch.cuda.is_available():
        torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:       # "nccl" is RCCL on ROCm; ONEPROT_DIST_BACKEND=gloo lets several ranks share one GPU (rehearsals, tests)
            backend = os.environ.get("ONEPROT_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def is_main_process():
    return get_rank() == 0


def save_on_master(*args, **kwargs):
    if is_main_process():
        torch.save(*args, **kwargs)


def mkdir(path):
    try:
        os.makedirs(path)
    except OSError as e:
        if e.errno != errno.EEXIST:
            raise


class ExchangeTimer:
    """Per-rank time spent in the exchange steps of the data path, measured with events on the launch stream: the packed feature all-gather,
    its reduce-scatter backward, and the part of the gradient all-reduce that is NOT hidden under the backward (from entering
    allreduce_gradients until every handle has been waited for).  bench.py --gpus N prints it per rank so that a scaling curve explains itself.
    Off by default (enable(): a handful of event records per step)."""
    active = None

    def __init__(self):
        self.spans = {}

    @classmethod
    def enable(cls):
        cls.active = cls()
        return cls.active

    @classmethod
    def disable(cls):
        cls.active = None

    class _Span:
        def __init__(self, timer, key):
            self.timer, self.key = timer, key

        def __enter__(self):
            if self.timer is not None and torch.cuda.is_available():
                self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                self.e0.record()
            return self

        def __exit__(self, *exc):
            if self.timer is not None and torch.cuda.is_available():
                self.e1.record()
                self.timer.spans.setdefault(self.key, []).append((self.e0, self.e1))

    @classmethod
    def span(cls, key):
        return cls._Span(cls.active, key)

    def totals_ms(self):
        torch.cuda.synchronize()
        return {k: sum(a.elapsed_time(b) for a, b in v) for k, v in self.spans.items()}


def grad_comm_dtype(name=None):
    """wire dtype of the gradient all-reduce: None / "fp32" (default; the reference's DDP reduces fp32 gradients) or "bf16" (half the bytes over
    xGMI: buckets are rounded to bf16, reduced, and written back into the fp32 gradient -- an option, not the default: the sum then runs in
    bf16).  `name` None reads ONEPROT_GRAD_COMM_DTYPE."""
    name = (name or os.environ.get("ONEPROT_GRAD_COMM_DTYPE") or "fp32").lower()
    if name in ("fp32", "float32", "f32"):
        return None
    if name in ("bf16", "bfloat16"):
        return torch.bfloat16
    raise ValueError(f"gradient communication dtype {name!r}: fp32 or bf16")


def grad_overlap_enabled(value=None):
    """ONEPROT_GRAD_OVERLAP (default 1): issue the arena-gradient all-reduce in ranges from inside the backward (GradOverlap) -- 0: one bucketed
    all-reduce after the backward, nothing co-resident with the backward's kernels.  Why a switch: every hot kernel of the step is ONE persistent
    work-group per CU with a static tile list, so a collective's channel work-groups that hold a few CUs through a GEMM make the displaced
    work-groups run after the others (tools/ab/cu_occupier.py, profiles/r05_cu_occupier.txt measures that with a stand-in on one GPU); which side
    wins on a real 8-GPU node is for the first multi-GPU run to decide with this switch -- bench.py prints the setting in its `exchange` block."""
    v = os.environ.get("ONEPROT_GRAD_OVERLAP", "1") if value is None else str(value)
    if v not in ("0", "1"):
        raise ValueError(f"ONEPROT_GRAD_OVERLAP={v!r}: 0 or 1")
    return v == "1"


class _Pending:
    """an asynchronous all-reduce of `src` (a gradient range, or its bf16 copy `wire` that is written back into `src` once the collective is done)"""

    def __init__(self, handle, src=None, wire=None):
        self.handle, self.src, self.wire = handle, src, wire

    def wait(self):
        self.handle.wait()
        if self.wire is not None:
            self.src.copy_(self.wire)
            self.wire = None


def _reduce_async(chunk, world, use_avg, average, wire_dtype):
    """mean (or sum) all-reduce of the fp32 range `chunk`, asynchronous; returns a _Pending"""
    op = dist.ReduceOp.AVG if (use_avg and average) else dist.ReduceOp.SUM
    if wire_dtype is None:
        if average and not use_avg:
            chunk.div_(world)
        return _Pending(dist.all_reduce(chunk, op=op, async_op=True))
    wire = (chunk / world if (average and not use_avg) else chunk).to(wire_dtype)
    return _Pending(dist.all_reduce(wire, op=op, async_op=True), chunk, wire)


class GradOverlap:
    """Asynchronous mean-all-reduce of arena-gradient ranges issued from inside an encoder's backward (one RCCL call per ~6 layers), so
    the reduction of the upper layers runs under the backward of the lower ones.  `allreduce_gradients` waits for the handles of a
    parameter that was reduced this way instead of reducing it again.  Attach with `attach(encoder.transformer)`."""

    def __init__(self, comm_dtype=None):
        self.world = get_world_size()
        self.use_avg = is_dist_avail_and_initialized() and dist.get_backend() == "nccl"
        self.wire = grad_comm_dtype(comm_dtype)
        self.calls = 0

    def attach(self, transformer):
        transformer._grad_overlap = self if self.world > 1 else None

    def reduce_range(self, param, gflat, lo, hi):
        chunk = gflat[lo:hi]
        self.calls += 1
        # the reduction operator is chosen once from the backend (RCCL has ncclAvg; gloo does not): an asynchronous collective reports
        # its errors at wait(), so there is nothing to catch and fall back from here
        h = _reduce_async(chunk, self.world, self.use_avg, True, self.wire)
        pend = getattr(param, "_oneprot_pending_reduce", None)
        if pend is None:
            pend = param._oneprot_pending_reduce = []
        pend.append(h)


def allreduce_gradients(parameters, bucket_bytes=256 << 20, average=True, comm_dtype=None):
    """Mean-all-reduce the gradients of `parameters` (only those with a .grad) in large flat buckets.
    The encoder arena gradient is already one contiguous tensor, so the 148 M-parameter encoder is reduced in place with a
    handful of RCCL calls sized for the per-link xGMI bandwidth (bucket_bytes), issued asynchronously and waited once."""
    if not is_dist_avail_and_initialized() or get_world_size() == 1:
        return
    with ExchangeTimer.span("grad_allreduce_exposed"):
        _allreduce_gradients(parameters, bucket_bytes, average, grad_comm_dtype(comm_dtype))


def _allreduce_gradients(parameters, bucket_bytes, average, wire=None):
    world = get_world_size()
    use_avg = average and dist.get_backend() == "nccl"           # RCCL: ncclAvg inside the collective; gloo: divide, then SUM
    op = dist.ReduceOp.AVG if use_avg else dist.ReduceOp.SUM
    handles, small = [], []
    for p in parameters:
        g = p.grad
        if g is None:
            continue
        pend = getattr(p, "_oneprot_pending_reduce", None)
        if pend:                                   # already being reduced (GradOverlap): just collect the handles
            handles.extend(pend)
            p._oneprot_pending_reduce = []
            continue
        if g.numel() * g.element_size() >= (1 << 20) and g.is_contiguous():
            flat = g.view(-1)
            step = max(bucket_bytes // g.element_size(), 1)
            for o in range(0, flat.numel(), step):
                handles.append(_reduce_async(flat[o:o + step], world, use_avg, average, wire))
        else:
            small.append(g)
    if small:
        buf = torch.cat([g.reshape(-1) for g in small])
        if average and not use_avg:
            buf.div_(world)
        dist.all_reduce(buf, op=op)
        o = 0
        for g in small:
            g.copy_(buf[o:o + g.numel()].view_as(g))
            o += g.numel()
    for h in handles:
        h.wait()
