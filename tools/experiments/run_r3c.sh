set -x
mkdir -p gpurun_out/r3c
timeout 600 python -m pytest tests/test_gpu_sgan.py -m gpu -x -q 2>&1 | tail -3
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r3c/bench_default.json 2> gpurun_out/r3c/bench_default.err; echo rc=$?
bash tools/sweep_scenes.sh r3c 0 > gpurun_out/r3c/sweep0.log 2>&1
bash tools/sweep_scenes.sh r3c_v2 524288 > gpurun_out/r3c/sweep2.log 2>&1
bash tools/sweep_scenes.sh r3c_v3 786432 > gpurun_out/r3c/sweep3.log 2>&1
# strong-scaling code path, 2 ranks on one GPU over gloo (numbers mean nothing)
TNP_BENCH_BACKEND=gloo TNP_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --config directional --global-scenes 256 --no-cpu-baseline --no-traffic > gpurun_out/r3c/bench_strong2_gloo.json 2> gpurun_out/r3c/bench_strong2.err; echo rc=$?
timeout 600 python bench.py --steps 10 --warmup 3 --config directional --global-scenes 256 --no-cpu-baseline --no-traffic > gpurun_out/r3c/bench_strong1.json 2> gpurun_out/r3c/bench_strong1.err; echo rc=$?
grep -h "scene-steps/s," gpurun_out/r3c*_sweep_scenes.md
tail -c 600 gpurun_out/r3c/bench_strong2_gloo.json; tail -5 gpurun_out/r3c/bench_strong2.err
