"""Multi-GPU glue: one process per GPU, collectives BEHIND the C-ABI (csrc/dist.{h,cpp}: RCCL over xGMI through dlopen, or the
host-staged "shm" transport that lets several ranks share one GPU in tests).  No torch anywhere on this path: `python -m
torch.distributed.run` is only the launcher whose environment (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT)
vnrAmdDistInitFromEnv reads.  The reference is single-GPU (no collectives anywhere), so this is new work per
SURVEY.md 8(e):

  * rendering shards the image by interleaved 8-scanline tile rows (global pixel indices keep the random sequences and the
    accumulation exact); each rank renders straight into its slot of a [world][share] buffer and ONE in-place all-gather
    per frame plus a de-interleaving kernel give every rank the whole frame (Renderer::set_distributed);
  * training is data parallel: every rank draws its own sample batch; the gradient travels as fp16, range by range while the
    backward pass and the optimizer update of other ranges run (NeuralVolume::train_data_parallel).

This module is a thin mirror of those entry points plus the pure sharding maths the CPU tests check.
"""
import ctypes as C
import os

import numpy as np

from . import api
from ._lib import check, lib

SUM, MAX, MIN, AVG = 0, 1, 2, 3
F32, F16, U8 = 0, 1, 2


# ------------------------------------------------------------------------------------------------ sharding maths (pure python; CPU-testable)
def interleave_layout(n_pixels, block, n_parts):
    """-> (n_blocks_total, blocks_per_part, n_local): the share layout of csrc/dist.cpp share_layout"""
    n_blocks = (n_pixels + block - 1) // block
    per_part = (n_blocks + n_parts - 1) // n_parts
    return n_blocks, per_part, per_part * block


def local_to_global(i, block, n_parts, part):
    blk, off = divmod(i, block)
    return (blk * n_parts + part) * block + off


def pack_share(frame_flat, block, n_parts, part, n_pixels):
    """frame_flat [n_pixels, 4] -> the pixels of `part` in local order [n_local, 4], zero padded past the image end
    (numpy restatement of what a rank's slot of the gathered buffer holds)"""
    n_blocks, per_part, n_local = interleave_layout(n_pixels, block, n_parts)
    f = np.zeros((n_blocks * block, frame_flat.shape[1]), frame_flat.dtype)
    f[:n_pixels] = frame_flat[:n_pixels]
    mine = f.reshape(n_blocks, block, -1)[part::n_parts]
    out = np.zeros((per_part, block, frame_flat.shape[1]), frame_flat.dtype)
    out[:mine.shape[0]] = mine
    return out.reshape(n_local, -1)


def assemble_shares(gathered, block, n_parts, n_pixels):
    """gathered [n_parts, n_local, 4] -> [n_pixels, 4] (numpy restatement of assemble_shares_kernel)"""
    n_blocks, per_part, n_local = interleave_layout(n_pixels, block, n_parts)
    g = np.asarray(gathered).reshape(n_parts, per_part, block, -1)
    return g.transpose(1, 0, 2, 3).reshape(per_part * n_parts * block, -1)[:n_pixels]


# ------------------------------------------------------------------------------------------------ process group
class Context:
    def __init__(self, rank=0, world=1, local_rank=0, transport=None):
        self.rank, self.world, self.local_rank = rank, world, local_rank
        self.transport = transport

    @property
    def distributed(self):
        """collectives are in play (a one-rank group counts: VNR_AMD_DIST_FORCE, tests)"""
        return self.transport is not None


def init_from_env(transport=None):
    """reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun contract); binds this process to its GPU and, with more
    than one rank, meets the others (vnrAmdDistInitFromEnv).  transport: "rccl" (default) or "shm"."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0")))
    L = lib()
    if transport:
        os.environ["VNR_AMD_DIST_TRANSPORT"] = transport
    if world > 1 or os.environ.get("VNR_AMD_DIST_FORCE"):
        check(L.vnrAmdDistInitFromEnv())
        return Context(L.vnrAmdDistRank(), L.vnrAmdDistWorldSize(), local_rank, L.vnrAmdDistTransport().decode())
    check(L.vnrAmdInit(local_rank if "LOCAL_RANK" in os.environ else -1))
    return Context(0, 1, local_rank, None)


def finalize():
    check(lib().vnrAmdDistFinalize())


def barrier(ctx=None):
    """device synchronize + host barrier over the ranks"""
    L = lib()
    if ctx is None or ctx.distributed:
        check(L.vnrAmdDistBarrier())
    else:
        check(L.vnrAmdSynchronize())


def self_test(deadline_s=30.0):
    """every collective of the sharded paths once on patterned buffers (vnrAmdDistSelfTest) -> (ok, report or error text)"""
    buf = C.create_string_buffer(2048)
    rc = lib().vnrAmdDistSelfTest(float(deadline_s), buf, len(buf))
    if rc != 0:
        return False, lib().vnrAmdGetLastError().decode(errors="replace")
    return True, buf.value.decode(errors="replace")


def all_reduce_host(values, op=SUM):
    """a few doubles over the control plane (the bench's MAX / SUM over ranks)"""
    a = (C.c_double * len(values))(*[float(v) for v in values])
    check(lib().vnrAmdDistAllReduceHost(a, len(values), op))
    return [float(v) for v in a]


# ------------------------------------------------------------------------------------------------ data-parallel training
def train_data_parallel(ctx, nv, steps, fast_mode=True):
    """`steps` optimisation steps, each equal to one step on the concatenated batch of all ranks"""
    if ctx.transport is None:
        api.vnrNeuralVolumeTrain(nv, steps, fast_mode)
        return
    check(lib().vnrAmdNeuralVolumeTrainDataParallel(nv.h, int(steps), 1 if fast_mode else 0))


def train_data_parallel_by_hand(ctx, nv, steps, fast_mode=True):
    """the same step spelled out with the TrainBegin / AllReduceGradients / TrainEnd entry points (one exchange of the whole
    blob per step, nothing overlapped): what an application that drives the step itself would write"""
    L = lib()
    check(L.vnrAmdNeuralVolumeSyncReplicas(nv.h))
    for _ in range(steps):
        check(L.vnrAmdNeuralVolumeTrainBegin(nv.h))
        check(L.vnrAmdNeuralVolumeAllReduceGradients(nv.h))
        check(L.vnrAmdNeuralVolumeTrainEnd(nv.h, 1.0 / ctx.world, 1 if fast_mode else 0))


def params_checksum(nv):
    p = api.neural_get_params_fp16(nv).view(np.uint16).astype(np.uint64)
    return int((p * (np.arange(p.size, dtype=np.uint64) % np.uint64(65521) + np.uint64(1))).sum() & np.uint64(0xFFFFFFFFFFFF))


# ------------------------------------------------------------------------------------------------ tile-sharded rendering
class ShardedRenderer:
    """wraps a vnrRenderer in distributed mode: `render()` is the pipeline of depth one (enqueue frame k, gather frame k - 1
    meanwhile, complete frame k, return the assembled frame k - 1), `flush()` returns the frame still in flight.  Returned
    values are device pointers (ctypes void pointers) of width x height vec4f frames, or None."""

    def __init__(self, ctx, renderer, width, height):
        self.ctx, self.r = ctx, renderer
        self.width, self.height = width, height
        self.n_pixels = width * height
        api.vnrRendererSetOutputAsDeviceFramebuffer(renderer, True)
        if ctx.distributed:
            check(lib().vnrAmdRendererSetDistributed(renderer.h, 1))

    def render(self):
        out = C.c_void_p()
        check(lib().vnrAmdRendererRenderPipelined(self.r.h, C.byref(out)))
        return out if out.value else None

    def flush(self):
        out = C.c_void_p()
        check(lib().vnrAmdRendererFlushPipeline(self.r.h, C.byref(out)))
        return out if out.value else None

    def completed_stats(self):
        """statistics of the frame completed last (does not complete the frame in flight)"""
        from ._lib import FrameStats
        st = FrameStats()
        check(lib().vnrAmdRendererGetCompletedFrameStats(self.r.h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in FrameStats._fields_}

    def download(self, ptr):
        """a frame returned by render() / flush() as a [height, width, 4] numpy array"""
        if ptr is None:
            return None
        out = np.empty((self.height, self.width, 4), np.float32)
        p = ptr if isinstance(ptr, C.c_void_p) else C.c_void_p(ptr)
        check(lib().vnrAmdMemcpyD2H(out.ctypes.data_as(C.c_void_p), p, out.nbytes))
        return out
