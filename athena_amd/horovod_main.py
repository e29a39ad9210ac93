"""Data-parallel training entry point (mirror of the reference's athena/horovod_main.py): one
process per GPU under ``torchrun --nproc-per-node N -m athena_amd.horovod_main config.json``; the
ranks average their gradients with RCCL where the reference uses Horovod."""
import logging
import sys

import torch
import torch.distributed as dist

from .main import parse_jsonfile, train
from .solver import HorovodSolver


def main(argv=None):
    argv = sys.argv if argv is None else argv
    logging.basicConfig(level=logging.INFO)
    if len(argv) < 2:
        logging.warning("Usage: python %s config_json_file", argv[0])
        return 1
    torch.manual_seed(1)
    parse_jsonfile(argv[1])
    HorovodSolver.initialize_devices()
    size = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    train(argv[1], HorovodSolver, size, rank)
    if dist.is_initialized():
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
