"""Two RCCL ranks, one of which never joins a halo exchange (tests/test_gpu_dist.py, two GPUs): rank 0 enqueues the grouped
ncclSend / ncclRecv and waits; its watchdog thread calls Comm.abort() one second later.  abort() must return although rank 0's main thread
is inside (or behind) RCCL, and rank 0 must come back with an error.  Rank 1 just sleeps and exits."""
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch.distributed as dist
    from gcn_drug_repurposing_amd import GssError
    from gcn_drug_repurposing_amd.dist import rccl_comm
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = rccl_comm(world, rank)
    if rank == 1:
        time.sleep(8.0)
        comm.abort()
        print("rank 1: never joined", flush=True)
        os._exit(0)

    def watchdog():
        time.sleep(1.0)
        t0 = time.time()
        comm.abort()
        print(f"rank 0: abort returned after {time.time() - t0:.2f} s", flush=True)

    threading.Thread(target=watchdog, daemon=True).start()
    d = 16
    send, recv = torch.ones(4, d, device="cuda"), torch.zeros(4, d, device="cuda")
    off = np.array([0, 0, 4], dtype=np.int64)          # nothing for myself, four rows to / from rank 1
    try:
        comm.exchange_rows(d, send, off, recv, off)
        comm.sync(30.0)
        print("rank 0: the exchange completed?", flush=True)
        os._exit(1)
    except GssError as e:
        print(f"rank 0: released with {e}", flush=True)
    os._exit(0)


if __name__ == "__main__":
    main()
