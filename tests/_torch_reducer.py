""" Test helper (not a test module, not part of the product): the host-driven all-reduce of homonim_amd.split_norm's
``reducer`` protocol over a torch.distributed group -- for ranks that share one GPU over gloo, where RCCL cannot run. """


class TorchReducer:
    """ All-reduce (SUM) over a torch.distributed process group of a float64 device buffer owned by a torch tensor.
    backend nccl (= RCCL on ROCm): in place on the device; gloo (ranks sharing one GPU in tests): through the host. """

    def __init__(self, n_doubles: int, device_index: int, group=None):
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError('TorchReducer needs an initialised torch.distributed process group (init_torch_group)')
        self._torch, self._dist, self._group = torch, dist, group
        self.world_size = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.buf = torch.zeros(n_doubles, dtype=torch.float64, device=torch.device('cuda', device_index))
        torch.cuda.synchronize(device_index)
        self.ptr = int(self.buf.data_ptr())
        self.n_doubles = int(n_doubles)
        self._device_index = device_index

    def __call__(self):
        if self.backend == 'nccl':
            self._dist.all_reduce(self.buf, op=self._dist.ReduceOp.SUM, group=self._group)
        else:
            host = self.buf.cpu()
            self._dist.all_reduce(host, op=self._dist.ReduceOp.SUM, group=self._group)
            self.buf.copy_(host)
        self._torch.cuda.synchronize(self._device_index)  # the library's stream reads the buffer next


def init_torch_group(local_rank: int) -> str:
    """ Join the torch.distributed process group the launcher's environment names (test infrastructure: the product's own ranks
    meet over loopback TCP, homonim_amd/dist.py).  HOMONIM_AMD_DIST_BACKEND=host / gloo: gloo (ranks sharing one GPU), else
    nccl = RCCL with one device per rank.  -> the backend's name """
    import os
    import torch
    import torch.distributed as dist
    want = os.environ.get('HOMONIM_AMD_DIST_BACKEND')
    backend = 'gloo' if want in ('host', 'gloo') or not torch.cuda.is_available() else 'nccl'
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if torch.cuda.is_available():
        torch.cuda.init()   # torch's HIP runtime opens the device now, before the library's does
    if backend == 'nccl':
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend='nccl', device_id=torch.device('cuda', local_rank))
    else:
        dist.init_process_group(backend='gloo')
    return backend
