"""Multi-GPU glue: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm, "gloo" on CPU
for tests).  The reference is single-GPU (no collectives anywhere), so this is new work per
SURVEY.md §8(e):

  * rendering shards the image by interleaved pixel blocks (global pixel indices keep RNG seeds and accumulation
    exact) and gathers the framebuffer shares with ONE all_gather per frame;
  * training is data parallel: every rank draws its own sample batch, gradients of the whole parameter blob live
    in ONE fp32 buffer that is all-reduced once per step, then every rank applies the identical optimizer update.

torch is plumbing only (device tensors aliasing the library's buffers + the collectives).
"""
import ctypes as C

import numpy as np

from . import api
from ._lib import check, check_ptr, lib, require_torch_loaded_first


# ------------------------------------------------------------------------------------------------ sharding maths (pure python; CPU-testable)
def interleave_layout(n_pixels, block, n_parts):
    """-> (n_blocks_total, blocks_per_part, n_local) for vnrAmdRendererSetPixelInterleave"""
    n_blocks = (n_pixels + block - 1) // block
    per_part = (n_blocks + n_parts - 1) // n_parts
    return n_blocks, per_part, per_part * block


def local_to_global(i, block, n_parts, part):
    blk, off = divmod(i, block)
    return (blk * n_parts + part) * block + off


def pack_share(frame_flat, block, n_parts, part, n_pixels):
    """frame_flat: [n_pixels, 4] tensor/array -> this rank's pixels [n_local, 4] (zero padded past the image end)"""
    import torch
    n_blocks, per_part, n_local = interleave_layout(n_pixels, block, n_parts)
    pad = n_blocks * block - n_pixels
    f = frame_flat
    if pad:
        f = torch.cat([f, f.new_zeros((pad, f.shape[1]))], 0)
    f = f.view(n_blocks, block, f.shape[1])
    mine = f[part::n_parts]
    if mine.shape[0] < per_part:
        mine = torch.cat([mine, mine.new_zeros((per_part - mine.shape[0], block, f.shape[2]))], 0)
    return mine.reshape(n_local, f.shape[2]).contiguous()


def share_view(frame_flat, block, n_parts, part):
    """this rank's pixels as a strided VIEW of the frame ([per_part, block, 4]); only when the blocks divide evenly"""
    n_pixels = frame_flat.shape[0]
    n_blocks, per_part, _ = interleave_layout(n_pixels, block, n_parts)
    if n_blocks * block != n_pixels or per_part * n_parts != n_blocks:
        return None
    return frame_flat.view(per_part, n_parts, block, frame_flat.shape[1])[:, part]


def assemble_shares_into(full_flat, gathered, block, n_parts):
    """gathered [n_parts, n_local, 4] -> full_flat [n_pixels, 4] in ONE strided copy (even division only)"""
    per_part = gathered.shape[1] // block
    full_flat.view(per_part, n_parts, block, gathered.shape[-1]).copy_(
        gathered.view(n_parts, per_part, block, gathered.shape[-1]).permute(1, 0, 2, 3))
    return full_flat


def assemble_shares(gathered, block, n_parts, n_pixels):
    """gathered: [n_parts, n_local, 4] -> [n_pixels, 4] full frame"""
    n_blocks, per_part, n_local = interleave_layout(n_pixels, block, n_parts)
    g = gathered.view(n_parts, per_part, block, gathered.shape[-1])
    full = g.permute(1, 0, 2, 3).reshape(per_part * n_parts * block, gathered.shape[-1])
    return full[:n_pixels]


# ------------------------------------------------------------------------------------------------ device tensor aliasing
class _CudaArray:
    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def as_torch(ptr, shape, typestr="<f4", device=None):
    require_torch_loaded_first()   # a device pointer of the library exists, so the library is loaded: torch must have come first
    import torch
    return torch.as_tensor(_CudaArray(ptr, shape, typestr), device=device if device is not None else torch.cuda.current_device())


# ------------------------------------------------------------------------------------------------ process group
class Context:
    def __init__(self, rank=0, world=1, local_rank=0, backend=None):
        self.rank, self.world, self.local_rank = rank, world, local_rank
        self.backend = backend

    @property
    def distributed(self):
        return self.world > 1


def init_from_env(device_backend="nccl"):
    """reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun contract); binds this process to its GPU"""
    import os
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        # torch BEFORE the first use of the library: its wheel bundles its own ROCm runtime, and two runtimes cannot
        # share the GPU in one process (_lib.require_torch_loaded_first explains; tools/repro_torch_*_lib.py reproduce)
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
    check(lib().vnrAmdInit(local_rank))
    if world > 1:
        require_torch_loaded_first()
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group(backend=device_backend, rank=rank, world_size=world)
    return Context(rank, world, local_rank, device_backend if world > 1 else None)


def barrier(ctx):
    check(lib().vnrAmdSynchronize())
    if ctx.distributed:
        import torch
        import torch.distributed as dist
        dist.barrier()
        torch.cuda.synchronize()


# ------------------------------------------------------------------------------------------------ data-parallel training
def train_data_parallel(ctx, nv, steps, fast_mode=True):
    """`steps` optimisation steps; with world > 1 gradients are all-reduced (sum) and scaled by 1/world, so the
    step equals one step on the concatenated batch (tcnn normalises the loss by the batch size)."""
    L = lib()
    if not ctx.distributed:
        api.vnrNeuralVolumeTrain(nv, steps, fast_mode)
        return
    import torch.distributed as dist
    grads = None
    for _ in range(steps):
        check(L.vnrAmdNeuralVolumeTrainBegin(nv.h))
        if grads is None:
            n = C.c_size_t()
            p = check_ptr(L.vnrAmdNeuralVolumeGradients(nv.h, C.byref(n)))
            grads = as_torch(p, (n.value,))
        check(L.vnrAmdSynchronize())          # library stream -> torch stream hand-off
        dist.all_reduce(grads, op=dist.ReduceOp.SUM)
        import torch
        torch.cuda.current_stream().synchronize()
        check(L.vnrAmdNeuralVolumeTrainEnd(nv.h, 1.0 / ctx.world, 1 if fast_mode else 0))


def params_checksum(nv):
    p = api.neural_get_params_fp16(nv).view(np.uint16).astype(np.uint64)
    return int((p * (np.arange(p.size, dtype=np.uint64) % np.uint64(65521) + np.uint64(1))).sum() & np.uint64(0xFFFFFFFFFFFF))


# ------------------------------------------------------------------------------------------------ tile-sharded rendering
class ShardedRenderer:
    """wraps a vnrRenderer: each rank renders its interleaved pixel blocks, one all_gather assembles the frame"""

    def __init__(self, ctx, renderer, width, height, block_rows=8):
        self.ctx, self.r = ctx, renderer
        self.width, self.height = width, height
        self.n_pixels = width * height
        self.block = block_rows * width
        api.vnrRendererSetOutputAsDeviceFramebuffer(renderer, True)
        if ctx.distributed:
            api.vnrRendererSetPixelInterleave(renderer, self.block, ctx.world, ctx.rank)
            import torch
            n_blocks, per_part, n_local = interleave_layout(self.n_pixels, self.block, ctx.world)
            self.gathered = torch.empty((ctx.world, n_local, 4), dtype=torch.float32, device="cuda")
            # blocks divide evenly among the ranks (1024 scanlines / 8 / 8 GPUs do): three device operations per frame, on
            # buffers and views made once: strided copy of the share, all_gather, strided copy into the frame
            self.even = n_blocks * self.block == self.n_pixels and per_part * ctx.world == n_blocks
            if self.even:
                self.share = torch.empty((n_local, 4), dtype=torch.float32, device="cuda")
                self.full_buf = torch.empty((self.n_pixels, 4), dtype=torch.float32, device="cuda")
            self._views = {}
            check(lib().vnrAmdRendererSetAsync(renderer.h, 1))
        self._prev = None
        self._gathered_event = None
        self.full = None

    def render(self):
        """Renders one frame.  Undistributed: returns the device pointer of that frame.  Distributed: a pipeline of depth one.
        The call enqueues frame k (asynchronous frames, vnrAmdRendererSetAsync), issues the gather of frame k - 1 while the GPU
        renders (the host side of three torch operations and one RCCL call is 0.1-0.15 ms, a sixth of what a rank's share of
        the bench frame takes on 8 GPUs), completes frame k and returns the assembled frame k - 1 ([n_pixels, 4] torch tensor;
        None on the first call).  `flush()` gathers the frame still in the pipeline."""
        if not self.ctx.distributed:
            api.vnrRender(self.r)
            return api.vnrRendererMapFrame(self.r)   # syncs the render stream
        if self._gathered_event is not None:
            # frame k overwrites the framebuffer frame k - 2 was gathered from (the renderer double-buffers)
            self._gathered_event.synchronize()
        api.vnrRender(self.r)                        # returns once the predicted iterations are enqueued
        out = self._gather(self._prev) if self._prev is not None else None
        self._prev = api.vnrRendererMapFrame(self.r)   # completes frame k (more iterations if rays are still alive)
        return out

    def flush(self):
        """gathers the frame still in the pipeline and returns it (None if there is none)"""
        if not self.ctx.distributed or self._prev is None:
            return None
        out = self._gather(self._prev)
        self._prev = None
        return out

    def _gather(self, ptr):
        import torch
        import torch.distributed as dist
        if self.even:
            key = int(ptr) if not hasattr(ptr, "value") else int(ptr.value)
            view = self._views.get(key)
            if view is None:   # the renderer double-buffers: two frame pointers, aliased once each
                view = self._views[key] = share_view(as_torch(ptr, (self.n_pixels, 4)), self.block, self.ctx.world, self.ctx.rank)
            self.share.view(view.shape).copy_(view)
            dist.all_gather_into_tensor(self.gathered.view(-1), self.share.view(-1))
            self.full = assemble_shares_into(self.full_buf, self.gathered, self.block, self.ctx.world)
        else:
            frame = as_torch(ptr, (self.n_pixels, 4))
            share = pack_share(frame, self.block, self.ctx.world, self.ctx.rank, self.n_pixels)
            dist.all_gather_into_tensor(self.gathered.view(-1), share.view(-1))
            self.full = assemble_shares(self.gathered, self.block, self.ctx.world, self.n_pixels)
        if self._gathered_event is None:
            self._gathered_event = torch.cuda.Event()
        self._gathered_event.record()
        return self.full
