"""Child of tests/test_launch.py: one CPU rank started by add_gym_amd.launch.spawn_ranks.  Joins a gloo group from the
environment the launcher prepared, all-reduces its rank, and (rank 0) prints one JSON line on stdout.  `fail` as first argument:
rank 1 exits with code 7 before the rendezvous and rank 0 must be stopped by the launcher."""
import json
import os
import sys
import time

if len(sys.argv) > 1 and sys.argv[1] == "fail":
    if os.environ["RANK"] == "1":
        sys.exit(7)
    time.sleep(600)
if len(sys.argv) > 2 and sys.argv[1] == "hang":  # every rank records its PID and sleeps: the launcher is then signalled by the test
    with open(os.path.join(sys.argv[2], "pid.%s" % os.environ["RANK"]), "w") as f:
        f.write(str(os.getpid()))
    time.sleep(600)
import torch
import torch.distributed as dist

# as bench.py does: libraries print banners on fd 1 (gloo's "connected to N peer ranks"); the result line goes to the saved descriptor
sys.stdout.flush()
json_fd = os.dup(1)
os.dup2(2, 1)

dist.init_process_group(backend="gloo")
t = torch.tensor([float(dist.get_rank() + 1)])
dist.all_reduce(t)
print("noise from rank %s" % os.environ["RANK"], file=sys.stderr)
if dist.get_rank() == 0:
    os.write(json_fd, (json.dumps({"group_size": dist.get_world_size(), "sum": float(t.item()), "local_rank": os.environ["LOCAL_RANK"],
                                   "master": os.environ["MASTER_ADDR"]}) + "\n").encode())
else:
    os.write(json_fd, b"rank %d stdout must not reach the launcher's stdout\n" % dist.get_rank())
dist.destroy_process_group()
