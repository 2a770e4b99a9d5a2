"""Entry point with the reference's modes and keys (add_gym/main.py:47-203): `python -m add_gym_amd.main
[mode=train|test] [a.b=c ...]`; under torchrun each rank masks to its own GPU and joins an RCCL process group."""
import os
import sys
from pathlib import Path

import torch


def _init_distributed():
    """main.py:128-176: auto-detect torchrun, bind the rank to its GPU, init RCCL."""
    if "RANK" not in os.environ or "WORLD_SIZE" not in os.environ or int(os.environ["WORLD_SIZE"]) == 1:
        return False
    from . import launch

    launch.bind_device(int(os.environ.get("LOCAL_RANK", 0)))  # one process per GPU; a LOCAL_RANK past the visible devices is an error
    torch.distributed.init_process_group(backend=launch.backend())  # "nccl" is RCCL on ROCm
    return True


def run_training(cfg, distributed):
    from .learning.add_agent import ADDAgent

    rank0 = (not distributed) or torch.distributed.get_rank() == 0
    log_dir = Path(cfg.get("log_dir", "logs")) / str(cfg["experiment_name"])
    if rank0:
        (log_dir / "intermediate_outputs").mkdir(parents=True, exist_ok=True)
    agent = ADDAgent(cfg, distributed=distributed)
    out_model_file = log_dir / "model.pt"
    if distributed:
        torch.distributed.barrier()
    if out_model_file.exists():  # spot-instance resume (main.py:92-98)
        agent.load(str(out_model_file))
    elif cfg.get("resume_path"):
        agent.load(cfg["resume_path"])
    agent.train_model(out_model_file=str(out_model_file), int_output_dir=str(log_dir / "intermediate_outputs"), log_file=str(log_dir / "log.txt"))


def run_test(cfg):
    from .learning.add_agent import ADDAgent

    agent = ADDAgent(cfg)
    if cfg.get("resume_path"):
        agent.load(cfg["resume_path"])
    print(agent.test_model(100))


def main(argv=None):
    from .config import load_config

    argv = list(sys.argv[1:] if argv is None else argv)
    name = "train"
    gpus = 1
    for a in list(argv):
        if a.startswith("--config-name="):
            name = a.split("=", 1)[1]
            argv.remove(a)
        elif a.startswith("--gpus="):  # `python -m add_gym_amd.main --gpus=8 ...` == torchrun --nproc_per_node=8 (sagemaker-entrypoint.sh:139-147)
            gpus = int(a.split("=", 1)[1])
            argv.remove(a)
    if gpus > 1 and "RANK" not in os.environ:
        from . import launch

        launch.check_world_fits(gpus, torch.cuda.device_count())  # counting devices does not initialise the GPU
        raise SystemExit(launch.spawn_ranks(["-m", "add_gym_amd.main", f"--config-name={name}"] + argv, gpus))
    cfg = load_config(name, argv)
    if cfg["mode"] == "train":
        distributed = _init_distributed()
        run_training(cfg, distributed)
        if distributed:
            torch.distributed.destroy_process_group()
    elif cfg["mode"] == "test":
        run_test(cfg)
    else:
        raise ValueError(f"Unknown mode: {cfg['mode']}. Please choose 'train' or 'test'.")


if __name__ == "__main__":
    main()
