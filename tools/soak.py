#!/usr/bin/env python3
"""GPU box: soak of the captured trainer step (closing scatter pass, whole-step graph): N steps with random views, then
the invariants a lost arrival or a stale counter would break: the device step counter equals the number of steps + 1,
every arrival counter of the scatter workspace is zero, parameters are finite, the table moved.
    python3 tools/soak.py [steps] [views per step]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "latent-nerf-test_amd")):
    sys.path.insert(0, _p)
import torch
from src.latent_nerf.configs.train_config import TrainConfig, apply_overrides
from src.latent_nerf.models import encoding as E
from src.latent_nerf.training.trainer import Trainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
views = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
with tempfile.TemporaryDirectory() as d:
    cfg = apply_overrides(TrainConfig(), {"log.exp_name": "soak", "log.exp_root": d, "optim.iters": steps, "optim.fp16": True,
                                          "log.save_interval": 10 ** 9, "log.eval_size": 1, "log.full_eval_size": 1,
                                          "log.quiet": True, "guide.text": "a lego man", "optim.views_per_step": views})
    tr = Trainer(cfg, device=dev)
    tr.full_eval = lambda: None
    t0 = tr.nerf.encoder.embeddings.detach().clone()
    tr.train()
    torch.cuda.synchronize()
    opt = tr.optimizer
    ws = E._scatter_ws[str(dev)]
    head = [w[:65536].view(torch.int32) for w in ws]
    # header: level maxima (first 4096 B: cleared by the closing pass), then item count (non-zero), then arrival counters
    arrive_ok = all(int(h[(4096 + 128) // 4:(4096 + 128 + 9 * 128) // 4].abs().sum()) == 0 for h in head)
    slice_ok = all(int(h[(4096 + 128 + 9 * 128) // 4:(4096 + 128 + 9 * 128 + 32 * 256 * 4) // 4].abs().sum()) == 0 for h in head)
    res = {"steps": tr.train_step, "graph_stats": tr.graph_stats, "step_dev": opt.step_dev.tolist(), "host_step": opt.step_no,
           "tail_arrivals_zero": arrive_ok, "slice_arrivals_zero": slice_ok,
           "finite": bool(all(torch.isfinite(p).all() for p in tr.nerf.parameters())),
           "table_moved": float((tr.nerf.encoder.embeddings.detach() - t0).abs().max())}
    print(res)
    ok = (res["step_dev"][0] == steps + 1 and res["step_dev"][1] == 0 and res["host_step"] == steps and arrive_ok and slice_ok
          and res["finite"] and res["table_moved"] > 0)
    print("SOAK", "OK" if ok else "FAILED")
    sys.exit(0 if ok else 1)
