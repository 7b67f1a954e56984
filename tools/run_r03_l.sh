# GPU box, round 3 step L: occupancy update + mean fused, trainer companions (fixed view / random views), host profile
set -u
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_render.py tests/test_gpu_trainer.py tests/test_gpu_config3_teddy.py -x -q -m gpu -k "occ or extra_state or refresh or trainer or teddy or occupancy" > gpurun_out/r03l_tests.log 2>&1
rc=$?; tail -5 gpurun_out/r03l_tests.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python3 bench.py --no-cpu-baseline > gpurun_out/r03l_bench.json 2> gpurun_out/r03l_bench.err; rc=$?; [ $rc -ne 0 ] && { tail -20 gpurun_out/r03l_bench.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/r03l_bench.json')); print('value', d['value'], 'no-refresh', d['refresh']['value_without_refresh'], 'refresh ms', d['refresh']['ms_per_refresh']); print('trainer', d['trainer']); print('trainer_random', d['trainer_random_views'])"
timeout -k 10 300 python3 - > gpurun_out/r03l_hostprof.txt 2>&1 <<'PY'
import cProfile, pstats, sys, os, tempfile, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "latent-nerf-test_amd"))
import torch, bench
from src.latent_nerf.configs.train_config import TrainConfig, apply_overrides
from src.latent_nerf.training.trainer import Trainer
root = tempfile.mkdtemp()
cfg = apply_overrides(TrainConfig(), {"log.exp_name": "p", "log.exp_root": root, "render.train_h": 64, "render.train_w": 64, "render.grid_size": 128,
    "render.eval_h": 8, "render.eval_w": 8, "log.eval_size": 1, "log.full_eval_size": 1, "log.save_interval": 10**9, "log.quiet": True,
    "optim.lr": 1e-7, "optim.fp16": True, "guide.text": "b", "optim.iters": 40})
cfg.render.train_pose = (60.0, 0.0, 1.25, 55.0); cfg.render.max_samples = bench.BENCH_CAPACITY
tr = Trainer(cfg, device=torch.device("cuda:0")); bench.sphere_scene(tr.nerf); tr.nerf.iter_density = 16; tr.full_eval = lambda: None
tr.train(); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable(); t0 = time.perf_counter()
tr.train(iters=440); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
pr.disable()
print("host ms/step %.4f  total ms/step %.4f" % ((t1 - t0) * 1e3 / 400, (t2 - t0) * 1e3 / 400))
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
PY
head -60 gpurun_out/r03l_hostprof.txt
exit 0
