"""Two training iterations for a rocprofv3 --kernel-trace run (the in-situ timeline of the update phase).
usage: rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 tools/trace_update.py [precision] [envs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, add_gym_amd
from add_gym_amd.config import load_config
from add_gym_amd.learning.add_agent import ADDAgent
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
ag = ADDAgent(load_config("train", [f"engine.num_envs={N}", f"agent.matmul_precision={prec}", "task.motion_file=synthetic:1x3600"]))
ag.reset_all_envs(); ag._init_train()
for it in range(2):
    ag._train_iter(); ag._iter += 1
torch.cuda.synchronize()
