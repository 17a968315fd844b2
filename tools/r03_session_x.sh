cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03_x
python bench.py --total-proofs 65536 --steps 10 --no-cpu-baseline --no-secondary > gpurun_out/r03_x/b16.json 2>gpurun_out/r03_x/b16.err
BPPP_NO_SMALL_KERNELS=1 python bench.py --total-proofs 65536 --steps 10 --no-cpu-baseline --no-secondary > gpurun_out/r03_x/b16_nosmall.json 2>gpurun_out/r03_x/b16_nosmall.err
python bench.py --total-proofs 32768 --steps 10 --no-cpu-baseline --no-secondary > gpurun_out/r03_x/b15.json 2>gpurun_out/r03_x/b15.err
python bench.py --total-proofs 131072 --steps 10 --no-cpu-baseline --no-secondary > gpurun_out/r03_x/b17.json 2>gpurun_out/r03_x/b17.err
python tools/show_bench.py gpurun_out/r03_x/b16.json gpurun_out/r03_x/b16_nosmall.json gpurun_out/r03_x/b15.json gpurun_out/r03_x/b17.json | grep -v "roofline\|setup"
