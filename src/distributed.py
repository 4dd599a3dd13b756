"""ref src/distributed.py surface -> oneprot_amd.distributed (RCCL over xGMI, torchrun or SLURM launched)."""
from oneprot_amd.distributed import (_get_first_node, allreduce_gradients, get_rank, get_world_size, init_distributed_mode,  # noqa: F401
                                     is_dist_avail_and_initialized, is_main_process, mkdir, save_on_master, setup_process_group)
