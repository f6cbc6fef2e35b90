set -e
cd $GRAFT_REPO_ROOT
python tools/bnn_mlp_bench.py --states 40960 | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print('base', round(d['fused_ms'],3), round(d['fused_frac_of_f32_matrix_peak'],3))"
for e in 1 2 3 4; do
  touch pddp_amd/csrc/bnn_mlp.hip
  make -C pddp_amd/csrc FLAGS_bnn_mlp=-DPDDP_MLP_EXP=$e > /dev/null 2>&1
  python tools/bnn_mlp_bench.py --states 40960 | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print('exp$e', round(d['fused_ms'],3), round(d['fused_frac_of_f32_matrix_peak'],3))"
done
