# GPU box (one card): functional rehearsal of the N > 1 bench path with two gloo ranks sharing the card --
# pipelined exchange (default), single-collective sink, f32 transport.  Throughput numbers of this run mean nothing
# (two processes on one GPU, collectives staged through the host); bench.py's own checks (finite, table moved, replicas'
# checksums bit-identical) are what is being exercised.
set -u
R=$GRAFT_REPO_ROOT
cd $R
export LNERF_DIST_BACKEND=gloo
i=0
for extra in "" "--exchange-groups 0" "--grad-transport f32" "--graph 0"; do
  i=$((i+1))
  timeout -k 10 200 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29510+i)) bench.py --gpus 2 --steps 12 --warmup 4 --no-cpu-baseline $extra > gpurun_out/dp_rehearsal_$i.json 2> gpurun_out/dp_rehearsal_$i.err
  rc=$?; echo "variant '$extra' rc=$rc"; tail -c 600 gpurun_out/dp_rehearsal_$i.json; echo; [ $rc -ne 0 ] && { tail -15 gpurun_out/dp_rehearsal_$i.err; exit $rc; }
done
exit 0
