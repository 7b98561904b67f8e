"""One process per GPU from a single call: the mirror of the reference's `launch()`
(focoos/utils/distributed/dist.py:38-95, worker :98-135) used by `FocoosModel.train` when `num_gpus > 1`
(focoos/models/focoos_model.py:255-274) - same arguments, same semantics:

* world_size = num_machines * num_gpus_per_machine; world_size == 1 runs `main_func(*args)` in-process;
* `dist_url="auto"` picks a free port on 127.0.0.1 (single machine only);
* every worker initialises the default process group (backend `nccl` - RCCL on ROCm - when a GPU is visible, `gloo`
  otherwise), pins its GPU (`torch.cuda.set_device(local_rank)`), synchronises, runs `main_func(*args)`, synchronises.

Differences that are deliberate: workers export RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT so code written
against `torchrun`'s environment contract (bench.py) runs unchanged under either launcher, and the backend can be forced
(`backend="gloo"`: CPU plumbing tests on a GPU box).  One process per GPU, RCCL over xGMI between them; nothing here touches a
device before the fork-free `spawn`, so each child owns its HIP context.
"""
from __future__ import annotations

import os
import socket
from datetime import timedelta
from typing import Optional

DEFAULT_TIMEOUT = timedelta(minutes=60)


def _find_free_port() -> int:
    sock = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    sock.bind(("127.0.0.1", 0))  # port 0: the OS picks a free one (another process may still grab it before we bind again)
    port = sock.getsockname()[1]
    sock.close()
    return port


def launch(main_func, num_gpus_per_machine: int, num_machines: int = 1, machine_rank: int = 0, dist_url: Optional[str] = None, args=(),
           timeout: timedelta = DEFAULT_TIMEOUT, backend: Optional[str] = None):
    world_size = num_machines * num_gpus_per_machine
    if world_size <= 1:
        return main_func(*args)
    if dist_url in (None, "auto"):
        if num_machines != 1:
            raise ValueError("dist_url=auto is not supported in multi-machine jobs")
        dist_url = f"tcp://127.0.0.1:{_find_free_port()}"
    from torch.multiprocessing.spawn import start_processes

    start_processes(_distributed_worker, nprocs=num_gpus_per_machine, daemon=False, start_method="spawn",
                    args=(main_func, world_size, num_gpus_per_machine, machine_rank, dist_url, args, timeout, backend))


def _distributed_worker(local_rank: int, main_func, world_size: int, num_gpus_per_machine: int, machine_rank: int, dist_url: str, args,
                        timeout: timedelta = DEFAULT_TIMEOUT, backend: Optional[str] = None):
    import torch
    import torch.distributed as dist

    has_gpu = torch.cuda.is_available()
    if backend is None:
        backend = "nccl" if has_gpu else "gloo"
    if backend == "nccl":
        if num_gpus_per_machine > torch.cuda.device_count():
            raise RuntimeError(f"launch: {num_gpus_per_machine} processes per machine but only {torch.cuda.device_count()} GPUs visible")
        torch.cuda.set_device(local_rank)
    global_rank = machine_rank * num_gpus_per_machine + local_rank
    if dist_url.startswith("tcp://"):
        host, port = dist_url[len("tcp://"):].rsplit(":", 1)
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = host, port
    os.environ["RANK"], os.environ["LOCAL_RANK"], os.environ["WORLD_SIZE"] = str(global_rank), str(local_rank), str(world_size)
    os.environ["LOCAL_WORLD_SIZE"] = str(num_gpus_per_machine)
    dist.init_process_group(backend=backend, init_method=dist_url, world_size=world_size, rank=global_rank, timeout=timeout)
    try:
        dist.barrier()  # the reference synchronises here so that slow starters do not trip the first collective's timeout
        main_func(*args)
        if dist.is_initialized():  # main_func may tear the group down itself (bench.py does under torchrun)
            dist.barrier()
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
