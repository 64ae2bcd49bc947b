# round 5, final tree: determinism repeats (stress case, small-system path), graph-builder fuzz, VerletGraph fuzz
cd $GRAFT_REPO_ROOT
python tools/stress_determinism.py > gpurun_out/r05_stress_determinism.txt 2>&1; cat gpurun_out/r05_stress_determinism.txt
python - <<'PY' > gpurun_out/r05_small_determinism.txt 2>&1
import sys, torch
sys.path[:0] = ['.', 'torch-m3gnet_amd', 'tests']
from torch_m3gnet.data.synthetic import fcc_cu_graph
from torch_m3gnet.data import MaterialGraphKey as K
from torch_m3gnet.model.build import build_model
torch.manual_seed(0)
model = build_model(5.0, 4.0, 3, 3, 95, 64, 3).cuda()
for n in (2, 3, 6):
    g = fcc_cu_graph(n, n, n).to('cuda')
    out = model(g)
    ref = {k: out[k].clone() for k in (K.TOTAL_ENERGY, K.FORCES, K.STRESSES)}
    bad = 0
    for i in range(300):
        o = model(g)
        bad += any(not torch.equal(o[k], v) for k, v in ref.items())
    print(f'{4 * n ** 3} atoms: mismatching repeats: {bad} of 300', flush=True)
PY
cat gpurun_out/r05_small_determinism.txt
timeout -k 10 600 python tests/checkers/fuzz_graph_build.py 300 > gpurun_out/r05_fuzz_graph_build.txt 2>&1; tail -1 gpurun_out/r05_fuzz_graph_build.txt
