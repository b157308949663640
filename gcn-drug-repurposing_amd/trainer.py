"""The train.py entry point of the reference (train.py:12-193), same flags, same files, HIP inside.

    python train.py --emb-file F --num-layers 2 --hidden-units 128 --k 5 --kq 5 --epochs 20 --lr 0.0003 \\
                    --graph-mode descriptor --beta-percentile 98 --batch-size 2048

reads F ('.embs.txt'), trains the GSS graph-convolution embedding and writes ./graph_embs.txt.
Flags the reference parses but never uses (--dataset, --data-path, --report-hard, --regularizer-scale, --kq)
are accepted and ignored.  New optional flags: --adj-file (weighted edgelist / .sif adjacency instead of the
kNN graph), --out, --cache-layer1, --batch-file (replay recorded batches), --log-loss, --checkpoint / --resume.
"""
from __future__ import annotations

import argparse
import os
import time

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from . import embio
from .engine import GssEngine
from .graph import GssGraph, edgelist_adj, knn_descriptor_adj_device
from .model import ResidualGraphConvolutionalNetwork


COLLECTIVE_TIMEOUT_S = 300.0   # sharded runs: a stream that does not drain for this long means a peer stopped taking part in a collective;
                               # gss_comm_sync then aborts the RCCL communicator and the rank exits non-zero instead of hanging


def build_parser():
    p = argparse.ArgumentParser(description="GSS-GCN embedding trainer on MI355X (train.py of gcn-drug-repurposing)")
    p.add_argument('--kq', type=int, default=5, help='Top k number for the query graph (unused, as in the reference).')
    p.add_argument('--k', type=int, default=5, help='Top k number for the index graph.')
    p.add_argument('--alpha', type=float, default=1, help='Parameter alpha for gss loss.')
    p.add_argument('--beta', type=float, default=None, help='Parameter beta for gss loss.')
    p.add_argument('--beta-percentile', type=float, default=None,
                   help="Select beta by the percentile of the similarity matrix's distribution.")
    p.add_argument('--seed', type=int, default=None, help='Random seed.')
    p.add_argument('--epochs', type=int, default=200, help='Number of epochs to train.')
    p.add_argument('--batch-size', type=int, default=0, help='Batch size; 0 trains on all samples at once.')
    p.add_argument('--hidden-units', type=int, default=128, help='Number of units in hidden layer')
    p.add_argument('--num-layers', type=int, default=2, help='Number of layers')
    p.add_argument('--loss', type=str, default='gss', help='Loss type (only gss is live in the reference).')
    p.add_argument('--lr', type=float, default=0.0001, help='Learning rate.')
    p.add_argument('--init-weights', type=float, default=1e-5, help='Std of the off-diagonal weight init (epsilon).')
    p.add_argument('--regularizer-scale', type=float, default=1e-5, help='(unused)')
    p.add_argument('--layer-decay', type=float, default=0.3, help='Residual GCN layer decay.')
    p.add_argument('--dataset', type=str, default='roxford5k', help='(unused)')
    p.add_argument('--emb-file', type=str, default=None, help='embedding file name.')
    p.add_argument('--data-path', type=str, default=None, help='(unused)')
    p.add_argument('--gpu-id', type=int, default=None, help='Which GPU to use (default: LOCAL_RANK or 0).')
    p.add_argument('--report-hard', action='store_true', help='(unused)')
    p.add_argument('--graph-mode', type=str, default='descriptor', choices=['descriptor', 'ransac', 'approx_ransac'])
    # additions
    p.add_argument('--adj-file', type=str, default=None,
                   help="weighted edgelist ('u v w') or .sif adjacency over the nodes of --emb-file; replaces the kNN graph")
    p.add_argument('--allow-nan', action='store_true',
                   help="train on when A + I has a row sum <= 0 (negative similarities): NaN embeddings, as the reference writes "
                        "them (helpers/helper.py:85 does not guard); without it such a graph is an error that names the node")
    p.add_argument('--out', type=str, default='graph_embs.txt', help='output file (reference: ./graph_embs.txt)')
    p.add_argument('--cache-layer1', dest='cache_layer1', action='store_true', default=True,
                   help="keep layer 1's two SpMM results across steps (their inputs, A_hat and X, never change; bitwise neutral): the default")
    p.add_argument('--no-cache-layer1', dest='cache_layer1', action='store_false', help="recompute them every step, as the reference does")
    p.add_argument('--batch-file', type=str, default=None, help='.npz with batches/batch_sizes to replay instead of sampling')
    p.add_argument('--log-loss', action='store_true', help='print the last loss of every epoch')
    p.add_argument('--checkpoint', default=None, help='write the training state (weights, Adam moments, step, beta, sampler RNG) '
                   'to this .npz after every epoch')
    p.add_argument('--resume', default=None, help='continue from a --checkpoint file: same flags, remaining epochs')
    p.add_argument('--ngpus', type=int, default=None,
                   help='node-range shards over N GPUs of this node; launch with `python -m torch.distributed.run '
                        '--nproc-per-node N train.py ...` (defaults to WORLD_SIZE)')
    return p


class _IndexDataset(Dataset):
    """item == index (method/dataset.py:5-23)"""

    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return i


def epoch_batches(loader):
    """One epoch of the reference sampler (method/dataset.py:25-28, train.py:127-131): a shuffled
    DataLoader over node ids.  Drawn from torch's global RNG exactly like the reference, so the same
    --seed gives the same batches."""
    return [b for b in loader]


def main(argv=None):
    args = build_parser().parse_args(argv)
    for key in vars(args):
        print(key + ":" + str(vars(args)[key]))
    if args.beta is not None and args.beta_percentile is not None:
        raise Exception('beta and beta_percentile can not be used at the same time!')
    if args.beta is None and args.beta_percentile is None:
        raise Exception('At least one of beta and beta_percentile should be set!')
    if args.loss != 'gss':
        raise Exception("only --loss gss is supported (tri_loss is dead code in the reference, modules/model.py:223-240)")
    if args.graph_mode != 'descriptor':
        raise Exception("--graph-mode ransac/approx_ransac need the image-retrieval RANSAC graphs the reference never ships")
    if not torch.cuda.is_available():
        raise RuntimeError("no GPU visible: this trainer has no CPU path (the reference's CPU path is the oracle, not the product)")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if args.ngpus is not None and args.ngpus != world:
        raise Exception(f"--ngpus {args.ngpus} but WORLD_SIZE={world}: launch with python -m torch.distributed.run "
                        f"--nproc-per-node {args.ngpus} --master-addr 127.0.0.1 train.py ...")
    sharded = world > 1 or os.environ.get("GSS_FORCE_SHARDED") == "1"
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # GSS_COMM_BACKEND=host: collectives staged through host memory over gloo, ranks may share a GPU (dist.job_comm) -- how a
    # one-GPU box runs the multi-process job; the default is RCCL with one GPU per rank
    host_backend = os.environ.get("GSS_COMM_BACKEND", "rccl").lower() == "host"
    if host_backend:
        from .dist import job_device
        local = job_device(local)
    dev = torch.device('cuda', args.gpu_id if (args.gpu_id is not None and world == 1) else local)
    torch.cuda.set_device(dev)
    if sharded:
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:          # single forced-sharded rank: any free port, so concurrent runs do not collide
                import socket
                with socket.socket() as s_:
                    s_.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(s_.getsockname()[1])
            import datetime
            if host_backend:
                dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=2 * COLLECTIVE_TIMEOUT_S))
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(seconds=2 * COLLECTIVE_TIMEOUT_S))

    if args.seed:                       # seed 0 / None leaves the RNGs unseeded, like train.py:74-76
        torch.manual_seed(args.seed)
        np.random.seed(args.seed)

    names, X = embio.read_embs(args.emb_file)                 # train.py:79-81
    n, d = X.shape
    if d != args.hidden_units:
        raise Exception(f"--hidden-units {args.hidden_units} must equal the embedding width {d} (modules/model.py:142)")

    t0 = time.time()
    # all but test-sized graphs go through the shard builder even on one GPU: it relabels the nodes hub-first for gather
    # locality (shards.build_shard: -2 % of a step at N = 30k, -30 % at 10M; invisible in the results)
    from .shards import RELABEL_MIN_NODES
    shard_path = sharded or n >= RELABEL_MIN_NODES
    graph = None
    source = None
    if shard_path:
        # a ROW SOURCE instead of a matrix: every rank assembles the rows of A + I (and of its transpose) it owns, nothing more --
        # the kNN top-k of its own row window on its GPU (train.py:93 -> helper.py:39-53), or its rows of the edgelist
        from .shards import EdgelistSource, KnnSource
        if args.adj_file:
            src, dst, w, _ = embio.read_edgelist(args.adj_file, names)
            source = EdgelistSource(src, dst, w, n, device=dev)
        else:
            source = KnnSource(X, args.k, device=dev)
    else:
        if args.adj_file:
            src, dst, w, _ = embio.read_edgelist(args.adj_file, names)
            adj = edgelist_adj(src, dst, w, n)
        else:
            adj = knn_descriptor_adj_device(X, args.k, device=dev)   # train.py:93 -> helper.py:39-53, similarity + top-k on device
        graph = GssGraph(adj, device=dev, need_transpose=args.num_layers > 1, allow_nan=args.allow_nan,
                         name_of=lambda i: names[i])                              # train.py:100-101
        print('Created G with [k={}] [shape=[{}, {}]] [nnz(A_hat)={}] in {:.2f}s'.format(args.k, n, n, graph.nnz, time.time() - t0))

    # the kernels want a feature width that is a multiple of 16; the reference takes any --hidden-units.  Zero
    # padding is exact: padded columns of X, W (rows and columns) and b are zero, so AX, AM, P, ELU(P), the row
    # norms, every gradient and every Adam update are identically zero there; they are stripped on output.
    d_pad = (d + 15) // 16 * 16
    bsz = args.batch_size if args.batch_size > 0 else n
    model = ResidualGraphConvolutionalNetwork(train_batch_size=bsz, val_batch_size=n, num_layers=args.num_layers,
                                              hidden_units=args.hidden_units, init_weights=args.init_weights,
                                              layer_decay=args.layer_decay).to(dev)   # train.py:111-121
    host_params = {k: p.detach().cpu().numpy() for k, p in zip(("W1", "b1", "W2", "b2"), model._params())}
    X32 = X.astype(np.float32)
    if d_pad != d:
        X32 = np.concatenate([X32, np.zeros((n, d_pad - d), np.float32)], axis=1)
        for k in ("W1", "W2"):
            w = np.zeros((d_pad, d_pad), np.float32)
            w[:d, :d] = host_params[k]
            host_params[k] = w
        for k in ("b1", "b2"):
            host_params[k] = np.concatenate([host_params[k], np.zeros(d_pad - d, np.float32)])
    if shard_path:
        # one process per GPU, node-range shards: a native gss_plan per rank that owns the RCCL communicator and enqueues
        # kernels and collectives from C++ (dist.sharded_plan_engine); same step semantics
        from .dist import job_comm, local_comms
        from .shards import build_shard, row_weight_for, shard_engine, shard_rows
        comm = job_comm(world, rank) if sharded else local_comms(1)[0]
        shard = build_shard(source, comm, need_transpose=args.num_layers > 1, device=dev, allow_nan=args.allow_nan, name_of=lambda i: names[i],
                            row_weight=row_weight_for(d_pad, args.num_layers))
        engine = shard_engine(shard, shard_rows(shard, X32), host_params, comm, num_layers=args.num_layers, layer_decay=args.layer_decay,
                              alpha=args.alpha, lr=args.lr, max_batch=min(bsz, n), cache_layer1=args.cache_layer1)
        if rank == 0:
            print('Created G with [k={}] [shape=[{}, {}]] [nnz(A_hat)={}] in {:.2f}s; {} node-range shards'.format(
                args.k, n, n, engine.global_nnz, time.time() - t0, world))

        def watch():
            # every host-side wait of a sharded run goes through the communicator's watchdog first (RCCL error poll + deadline)
            if sharded:
                comm.sync(COLLECTIVE_TIMEOUT_S)

        def full_embeddings():
            out = engine.gather_embeddings()
            watch()
            return out

        def percentile(q):
            watch()
            return engine.percentile(q)
    else:
        feats = torch.from_numpy(X32).to(dev)                    # method/dataset.py:13
        params = [torch.from_numpy(host_params[k]).to(dev) for k in ("W1", "b1", "W2", "b2")]
        engine = GssEngine(graph, feats, params, num_layers=args.num_layers, layer_decay=args.layer_decay, alpha=args.alpha,
                           lr=args.lr, max_batch=min(bsz, n), cache_layer1=args.cache_layer1)
        full_embeddings = engine.gather_embeddings
        percentile = engine.percentile

        def watch():
            pass
    loader = DataLoader(_IndexDataset(n), batch_size=bsz, shuffle=True, num_workers=0, drop_last=False)

    replay = None
    if args.batch_file:
        z = np.load(args.batch_file)
        offs = np.concatenate([[0], np.cumsum(z['batch_sizes'])])
        replay = [np.asarray(z['batches'][offs[i]:offs[i + 1]], dtype=np.int64) for i in range(len(z['batch_sizes']))]
        # replayed ids index device buffers directly: the kernels assume what the reference's sampler guarantees
        # (a batch is a slice of a permutation of 0..N-1, method/dataset.py:25-28), so check it on the host
        for i, b in enumerate(replay):
            if b.size == 0 or b.size > min(bsz, n):
                raise Exception(f"{args.batch_file}: batch {i} has {b.size} ids, expected 1..{min(bsz, n)} (--batch-size)")
            if b.min() < 0 or b.max() >= n:
                raise Exception(f"{args.batch_file}: batch {i} holds node ids outside [0, {n})")
            if np.unique(b).size != b.size:
                raise Exception(f"{args.batch_file}: batch {i} repeats a node id (batches must be duplicate-free)")
    steps_per_epoch = (n + bsz - 1) // bsz

    beta_score = args.beta
    itr = 0
    step_no = 0
    # what a resumed run must share with the run that wrote the checkpoint to be its continuation
    hyper = {"lr": args.lr, "alpha": args.alpha, "layer_decay": args.layer_decay, "batch_size": float(bsz),
             "seed": float(args.seed or 0), "init_weights": args.init_weights}
    if args.resume:
        z = np.load(args.resume)
        if int(z["n"]) != n or int(z["d"]) != d_pad or int(z["num_layers"]) != args.num_layers:
            raise Exception(f"{args.resume} was written for N={int(z['n'])}, d={int(z['d'])}, L={int(z['num_layers'])}")
        hyper_saved = {k: float(z["hp_" + k]) for k in hyper} if "hp_lr" in z.files else None
        if hyper_saved is not None and any(abs(hyper_saved[k] - hyper[k]) > 1e-12 * max(1.0, abs(hyper[k])) for k in hyper):
            diff = {k: (hyper_saved[k], hyper[k]) for k in hyper if hyper_saved[k] != hyper[k]}
            raise Exception(f"{args.resume} was written with different hyper-parameters (saved, now): {diff}")
        if int(z["epoch"]) >= args.epochs:
            raise Exception(f"{args.resume} already holds {int(z['epoch'])} epochs, --epochs {args.epochs} leaves nothing to train "
                            f"(a finished run writes the embeddings from BEFORE its last optimizer step; they cannot be rebuilt "
                            f"from the post-step state)")
        engine.load_state_dict(z)
        itr, step_no, beta_score = int(z["epoch"]), int(z["step"]), float(z["beta"])
        torch.set_rng_state(torch.from_numpy(z["torch_rng"].copy()))     # the sampler continues its permutation stream
        engine.forward()                                     # the embeddings a finished run writes come from a forward
        print(f"resumed from {args.resume}: {itr} epochs done, beta {beta_score}")
    while itr < args.epochs:                                  # train.py:151
        start_time = time.time()
        if replay is not None:
            batches = [torch.as_tensor(b) for b in replay[itr * steps_per_epoch:(itr + 1) * steps_per_epoch]]
        else:
            batches = epoch_batches(loader)
        sizes = [int(b.numel()) for b in batches]
        idx32 = torch.cat(batches).to(torch.int32).to(dev)
        if sharded and world > 1:
            import torch.distributed as dist
            if host_backend:                 # a gloo group carries host tensors
                host_idx = idx32.cpu()
                dist.broadcast(host_idx, src=0)
                idx32 = host_idx.to(dev)
            else:
                dist.broadcast(idx32, src=0)     # every rank trains on rank 0's batches even when no --seed is given
        off = 0
        for batch_id, b in enumerate(sizes):
            if itr == 0 and batch_id == 0:
                engine.forward()                              # train.py:158-161
                if args.beta_percentile is not None:
                    beta_score = percentile(args.beta_percentile)   # train.py:165-167
                    print(f"selected beta:{beta_score}")
                engine.loss_backward(idx32, beta_score, count=b, offset=off)   # train.py:175,183
                engine.adam()                                 # train.py:184
            elif itr == args.epochs - 1 and batch_id == len(sizes) - 1:
                engine.step(idx32, beta_score, count=b, offset=off)          # the run's last forward: all rows (they are written out)
            else:
                # every other step reads B rows of its top layer (the loss, train.py:175): gss_plan_step_lazy computes those -- the
                # same loss, gradients and parameters bit for bit (the reference computes all N rows every step, train.py:158-161)
                engine.step_lazy(idx32, beta_score, count=b, offset=off)
            off += b
            step_no += 1
        watch()
        itr += 1
        if args.checkpoint and rank == 0:          # weights and optimizer state are replicated: rank 0's copy is the job's
            sd = engine.state_dict()
            tmp = args.checkpoint + ".tmp.npz"
            np.savez(tmp, epoch=itr, beta=float(beta_score), n=n, d=d_pad, num_layers=args.num_layers,
                     torch_rng=torch.get_rng_state().numpy(), **{"hp_" + k: v for k, v in hyper.items()}, **sd)
            os.replace(tmp, args.checkpoint)
        if args.log_loss:
            print(f"iter {itr} loss {float(engine.loss.item()):.8f} time {time.time() - start_time:.4f}s")
        else:
            print(f"iter {itr}")
    torch.cuda.synchronize()
    # embeddings of the last forward, i.e. before the last optimizer step (train.py:158,193)
    final = full_embeddings().cpu().numpy()[:, :d]
    if rank == 0:
        embio.write_graph_embs(args.out, final)
    if sharded and world > 1:
        import torch.distributed as dist
        dist.barrier()
    return engine


if __name__ == '__main__':
    main()
