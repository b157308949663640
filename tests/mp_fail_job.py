"""Child of tests/test_gpu_dist.py::test_multi_process_job_with_a_dying_rank_ends_instead_of_hanging -- NOT a test module.

Run under torch.distributed.run with GSS_COMM_BACKEND=host: every rank builds its shard and its native sharded plan the way
trainer.py does, trains a few steps; rank 1 leaves the job without a word in the middle of step 3.  The surviving ranks must come
back from their collective with an error (exit code 3 here) instead of waiting for the dead peer forever."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import gcn_drug_repurposing_amd as pkg  # noqa: E402
from gcn_drug_repurposing_amd._lib import GssError  # noqa: E402
from gcn_drug_repurposing_amd.dist import job_comm, job_device  # noqa: E402
from gcn_drug_repurposing_amd.shards import RmatSource, build_shard, gaussian_rows, shard_engine  # noqa: E402


def main():
    import datetime
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(job_device(int(os.environ.get("LOCAL_RANK", "0"))))
    dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=30))
    pkg.load()
    comm = job_comm(world, rank)
    n, m, d, B = 20000, 200000, 32, 256
    shard = build_shard(RmatSource(n, m, seed=3, device="cuda"), comm, need_transpose=True, device="cuda")
    lo, hi = shard.part.rows(rank)
    w = np.eye(d, dtype=np.float32)
    params = {"W1": w, "b1": np.zeros(d, np.float32), "W2": w.copy(), "b2": np.zeros(d, np.float32)}
    eng = shard_engine(shard, gaussian_rows(lo, hi, d, 5), params, comm, num_layers=2, layer_decay=0.3, alpha=1.0, lr=3e-4, max_batch=B)
    rng = np.random.RandomState(1)
    try:
        for step in range(6):
            idx = torch.from_numpy(rng.permutation(n)[:B].astype(np.int32)).cuda()
            if step == 3 and rank == 1:
                os._exit(17)                       # no abort, no goodbye: the process is simply gone
            eng.step(idx, 0.25)
            comm.sync(20.0)
            print(f"rank {rank} step {step} loss {eng.loss.item():.6f}", flush=True)
    except GssError as e:
        print(f"rank {rank}: collective failed as it should: {e}", file=sys.stderr, flush=True)
        os._exit(3)
    os._exit(0)


if __name__ == "__main__":
    main()
