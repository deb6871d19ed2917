#!/usr/bin/env python
"""Entry point with the reference's command line (reference main.py:43-61): ``python main.py --config X.yaml``.

Single GPU: runs in-process.  ``--ddp True --world-size N`` (or the yaml keys): one process per GPU is spawned
(reference main.py:55 ``mp.spawn``), each calls ``init_process_group("nccl")`` = RCCL over xGMI.  Also works under
``python -m torch.distributed.run`` (RANK/WORLD_SIZE already set)."""
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import agcn_amd  # noqa: E402,F401
from agcn_amd.processor import Processor, load_args  # noqa: E402


def _run(rank, world_size, argv):
    if world_size > 1 and 'RANK' not in os.environ:
        os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world_size))
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '8020')
    Processor(load_args(argv)).start()


def main(argv=None):
    arg = load_args(argv)
    if arg.ddp and arg.world_size > 1 and 'RANK' not in os.environ:
        import torch.multiprocessing as mp
        mp.spawn(_run, args=(arg.world_size, argv), nprocs=arg.world_size, join=True)
    else:
        _run(0, 1, argv)


if __name__ == '__main__':
    main()
