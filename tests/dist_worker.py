"""One rank of the on-card data-parallel rehearsal (tests/test_gpu_distributed.py): trains a small latent-NeRF with the
real Trainer on the HIP path, several ranks sharing ONE card through the gloo backend (LNERF_DIST_BACKEND=gloo), and
writes checksums of its replica.  Launched with RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT set."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "latent-nerf-test_amd")):
    sys.path.insert(0, p)


def digest(t):
    t = t.detach().contiguous().cpu()
    if t.dtype == torch_bf16():
        t = t.view(__import__("torch").int16)
    return hashlib.sha256(t.numpy().tobytes()).hexdigest()


def torch_bf16():
    import torch
    return torch.bfloat16


def main():
    out_dir, groups, steps, precision = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    save = len(sys.argv) > 5 and sys.argv[5] == "save"
    from src.latent_nerf.training.distributed import init_distributed
    rank, world, dev = init_distributed()          # before anything touches the GPU
    import torch
    from src.latent_nerf.configs.train_config import TrainConfig, apply_overrides
    from src.latent_nerf.training.trainer import Trainer
    cfg = apply_overrides(TrainConfig(), {
        "log.exp_name": "dp", "log.exp_root": os.path.join(out_dir, "exp"), "render.train_h": 32, "render.train_w": 32,
        "render.eval_h": 32, "render.eval_w": 32, "render.grid_size": 64, "optim.iters": steps, "optim.lr": 5e-3,
        "log.save_interval": 10000, "log.eval_size": 1, "log.full_eval_size": 2, "optim.fp16": precision == "bf16",
        "guide.text": "a lego man",
        "optim.views_per_step": world * int(os.environ.get("LNERF_TEST_VIEWS_PER_RANK", "1")), "optim.exchange_groups": groups,
        "optim.graph_collectives": os.environ.get("LNERF_TEST_GRAPH_COLLECTIVES", "1") != "0",
        "optim.shard_table_optimizer": os.environ.get("LNERF_TEST_SHARD", "0") == "1"})
    tr = Trainer(cfg, device=dev)
    table0 = tr.nerf.encoder.embeddings.detach().clone()
    if save:   # the table after the FIRST step too (Adam's first step is lr * sign(g): an exact comparison point)
        import numpy as np
        tr.full_eval = lambda: None
        tr.train(iters=1)
        torch.cuda.synchronize()
        np.save(os.path.join(out_dir, "table1_rank%d.npy" % rank), tr.nerf.encoder.embeddings.detach().cpu().numpy())
    tr.train()
    torch.cuda.synchronize()
    res = {"rank": rank, "world": world, "pipelined": bool(tr.pipelined), "steps": tr.train_step,
           "iter_density": tr.nerf.iter_density,
           "table": digest(tr.nerf.encoder.embeddings), "bitfield": digest(tr.nerf.density_bitfield),
           "density_grid": digest(tr.nerf.density_grid), "mean_density": float(tr.nerf.mean_density_dev),
           "mlp": [digest(getattr(tr.nerf, k)) for k in ("w1", "b1", "w2", "b2", "w3", "b3")],
           "table_moved": float((tr.nerf.encoder.embeddings.detach() - table0).abs().max()),
           "finite": bool(torch.isfinite(tr.nerf.encoder.embeddings).all()),
           "bits_set": int(tr.nerf.density_bitfield.count_nonzero()),
           "noise_seed": int(cfg.render.noise_seed)}
    sh = tr.nerf.encoder.shadow()
    res["shadow"] = digest(sh) if sh is not None else None
    res["sharded"] = bool(getattr(tr, "sharded", False))
    if res["sharded"]:
        # the f32 master is only current on each row's owner until it is gathered -- which the checkpoint does (collective:
        # every rank calls it); the file rank 0 wrote must then hold the table every rank now has
        res["table_before_gather"] = res["table"]
        path = tr.save_checkpoint(full=True)
        torch.cuda.synchronize()
        res["table"] = digest(tr.nerf.encoder.embeddings)
        res["moments"] = [digest(tr.optimizer.big[0][1]), digest(tr.optimizer.big[0][2])]
        if rank == 0:
            state = torch.load(path, map_location="cpu", weights_only=True)
            res["ckpt_table"] = hashlib.sha256(state["model"]["encoder.embeddings"].contiguous().numpy().tobytes()).hexdigest()
            res["ckpt_m"] = hashlib.sha256(state["optimizer"]["exp_avg"][0].contiguous().numpy().tobytes()).hexdigest()
    res["views"] = list(tr.views)
    res["exchange"] = bool(tr.exchange)
    res["capture_exchange"] = bool(tr.capture_exchange)
    res["graph_stats"] = dict(tr.graph_stats)
    if save:   # the tensors themselves, for comparisons to a tolerance (forced single-rank RCCL against the fused step)
        import numpy as np
        np.save(os.path.join(out_dir, "table_rank%d.npy" % rank), tr.nerf.encoder.embeddings.detach().cpu().numpy())
        np.save(os.path.join(out_dir, "table0_rank%d.npy" % rank), table0.cpu().numpy())
        np.save(os.path.join(out_dir, "w2_rank%d.npy" % rank), tr.nerf.w2.detach().cpu().numpy())
    with open(os.path.join(out_dir, "rank%d.json" % rank), "w") as f:
        json.dump(res, f)
    import torch.distributed as dist
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
